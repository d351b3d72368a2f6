#!/bin/bash
# round 3, GPU call 4: loss-only merged launch, scaling proxies for cfg2..cfg5, all-config timings (+ cfg4 without records)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mirror.py -q -x -k "merged or loss_only or one_kernel" > $O/tests_new.log 2>&1; echo "rc=$?" >> $O/tests_new.log
tail -n 4 $O/tests_new.log
timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_cfg2.log 2>&1
timeout 900 python tools/scaling_proxy.py --out $O/scaling_proxy.json > $O/scaling_proxy.txt 2>&1
timeout 600 python tools/bench_configs.py cfg1 cfg2 cfg3 cfg4 cfg5 > $O/all_configs.txt 2>&1
PINN_REC_GB=0 timeout 300 python tools/bench_configs.py cfg4 > $O/cfg4_norec.txt 2>&1
cat $O/ab_env_cfg2.log $O/scaling_proxy.txt; grep -v "^    " $O/all_configs.txt; grep -v "^    " $O/cfg4_norec.txt
