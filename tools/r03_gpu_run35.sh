#!/bin/bash
# round 3, GPU call 35: k_reduce_one with all sixteen loads of a chunk in flight — kernel trace of the bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zk
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err)
python profiles/rocpd_stats.py $O/prof/bench_results.db > $O/kernel_stats.txt 2>&1
find $O -name "*.db" -delete
head -6 $O/kernel_stats.txt | cut -c1-170
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['value'],d['loss_terms'])"
timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -n 2
