#!/bin/bash
# round 3, GPU call 21: forward image as the first a-jet operand; read-ahead request in the dA + dW region — parity + A/B
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03w
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -m gpu -x > $O/tests_parity.log 2>&1; echo "rc=$?" >> $O/tests_parity.log
tail -n 4 $O/tests_parity.log
timeout 300 python tools/ab_compare.py head ahead0 fwdimg0 > $O/ab_cfg2.txt 2>&1
timeout 300 python tools/ab_compare.py --cfg cfg3 head ahead0 fwdimg0 > $O/ab_cfg3.txt 2>&1
grep round $O/ab_cfg2.txt $O/ab_cfg3.txt | cut -c1-200
timeout 600 python tools/bench_configs.py cfg4 cfg5 > $O/configs.txt 2>&1
grep -v "^    " $O/configs.txt | tail -4
