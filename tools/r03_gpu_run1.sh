#!/bin/bash
# round 3, GPU call 1: new-feature tests on hardware, A/B of the launch structure, bench, kernel trace
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mirror.py -q -x -k "merged or one_kernel or loss_only or chained or lbfgs or dgm_burgers_parity or two_stage" > $O/tests_new.log 2>&1; echo "rc=$?" >> $O/tests_new.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_engine_comm.py -q -x -m gpu > $O/tests_parity.log 2>&1; echo "rc=$?" >> $O/tests_parity.log
timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_cfg2.log 2>&1
timeout 300 python tools/ab_env.py --cfg cfg3 --points 262144 32768 --steps 100 > $O/ab_env_cfg3.log 2>&1
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 120 python bench.py --points 8192 --no-cpu-baseline --steps 200 > $O/bench_8192.json 2> $O/bench_8192.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o bench -- python $OLDPWD/bench.py --no-cpu-baseline > $OLDPWD/$O/bench_under_rocprof.json 2> $OLDPWD/$O/rocprof.err)
ls -R $O/prof | head -30
for db in $(find $O/prof -name "*.db"); do python profiles/rocpd_stats.py $db > $O/kernel_stats.txt 2>&1; python profiles/rocpd_timeline.py $db 2>/dev/null | tail -40 > $O/timeline_tail.txt; done
find $O/prof -name "*.db" -size +20M -delete
cat $O/kernel_stats.txt | head -20
tail -5 $O/tests_new.log $O/tests_parity.log
cat $O/ab_env_cfg2.log $O/ab_env_cfg3.log
cat $O/bench.json | head -c 3000
