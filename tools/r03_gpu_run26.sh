#!/bin/bash
# round 3, GPU call 26: final build of the round (transpose-read dW) — full -m gpu suite, bench (+ share), kernel trace, PMC, scaling proxies, all configs, JIT create times
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
tail -n 8 $O/tests_gpu.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 120 python bench.py --points 8192 --no-cpu-baseline --steps 200 > $O/bench_8192.json 2> $O/bench_8192.err
R=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof8k -o bench -- python $R/bench.py --no-cpu-baseline --points 8192 --events none > $R/$O/bench8k_under_rocprof.json 2> $R/$O/rocprof8k.err)
python profiles/rocpd_stats.py $O/prof/bench_results.db > $O/kernel_stats.txt 2>&1
python profiles/rocpd_stats.py $O/prof8k/bench_results.db > $O/kernel_stats_8k.txt 2>&1
python profiles/rocpd_timeline.py $O/prof8k/bench_results.db 24 > $O/timeline_8k.txt 2>&1
python profiles/rocpd_timeline.py $O/prof/bench_results.db 12 > $O/timeline.txt 2>&1
find $O -name "*.db" -size +20M -delete
timeout 600 bash tools/pmc_profile.sh $O/pmc > $O/pmc.log 2>&1
find $O -name "*.db" -size +8M -delete
timeout 900 python tools/scaling_proxy.py --out $O/scaling_proxy.json > $O/scaling_proxy.txt 2>&1
timeout 600 python tools/bench_configs.py cfg1 cfg2 cfg3 cfg4 cfg5 > $O/all_configs.txt 2>&1
timeout 900 python tools/jit_create_time.py > $O/jit_create_time.txt 2>&1
head -8 $O/kernel_stats.txt; cat $O/scaling_proxy.txt | grep N=; grep -v "^    " $O/all_configs.txt; tail -3 $O/pmc.log; grep create $O/jit_create_time.txt
