#!/bin/bash
# round 3, GPU call 20: level 3 (transpose-read dW) on the 128-wide kernels — goldens, A/B against the staged fp32 dW (libpinn_h128off)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03v
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -m gpu -x > $O/tests_parity.log 2>&1; echo "rc=$?" >> $O/tests_parity.log
tail -n 6 $O/tests_parity.log
timeout 600 python tools/ab_compare.py --cfg cfg4 head h128off > $O/ab_cfg4.txt 2>&1
timeout 600 python tools/ab_compare.py --cfg cfg5 head h128off > $O/ab_cfg5.txt 2>&1
grep round $O/ab_cfg4.txt $O/ab_cfg5.txt | cut -c1-400
timeout 600 python tools/bench_configs.py cfg4 cfg5 > $O/configs.txt 2>&1
grep -v "^    " $O/configs.txt | tail -20
