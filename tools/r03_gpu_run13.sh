#!/bin/bash
# round 3, GPU call 14: two-stage dW staging: the six-group 128-wide kernel (cfg5) on split-operand GEMMs
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/golden_check.py cfg4_full cfg5_full > $O/golden_new.txt 2>&1
timeout 300 python tools/ab_env.py --cfg cfg5 --points 1000000 --steps 20 > $O/ab_cfg5_new.txt 2>&1
timeout 300 python tools/bench_configs.py cfg4 cfg5 > $O/configs.txt 2>&1
timeout 900 python -m pytest tests -q -m gpu -x -k "128 or cfg4 or cfg5 or cavity or heat or coupled or golden" > $O/tests_h128.log 2>&1; echo "rc=$?" >> $O/tests_h128.log
grep -v "^$" $O/golden_new.txt | tail -3
grep -h "merged \|chained \|loss-only \|==" $O/ab_cfg5_new.txt | cut -c1-200
grep -v "^    " $O/configs.txt
tail -3 $O/tests_h128.log
