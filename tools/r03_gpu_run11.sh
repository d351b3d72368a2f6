#!/bin/bash
# round 3, GPU call 11: 128-wide kernels with split-operand forward / dA GEMMs vs their fp32 build: cfg4 / cfg5 timings, full-size goldens, the H = 128 tests
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
V=$(pwd)/neuralpde.jl_amd/csrc/abl/libpinn_h128f32.so
timeout 600 python tools/golden_check.py cfg4_full cfg5_full > $O/golden_new.txt 2>&1
PINN_LIB=$V timeout 600 python tools/golden_check.py cfg4_full cfg5_full > $O/golden_f32.txt 2>&1
for r in 1 2; do
timeout 300 python tools/ab_env.py --cfg cfg4 --points 1048576 --steps 30 > $O/ab_cfg4_new_$r.txt 2>&1
PINN_LIB=$V timeout 300 python tools/ab_env.py --cfg cfg4 --points 1048576 --steps 30 > $O/ab_cfg4_f32_$r.txt 2>&1
done
timeout 300 python tools/ab_env.py --cfg cfg5 --points 1000000 --steps 20 > $O/ab_cfg5_new.txt 2>&1
PINN_LIB=$V timeout 300 python tools/ab_env.py --cfg cfg5 --points 1000000 --steps 20 > $O/ab_cfg5_f32.txt 2>&1
timeout 900 python -m pytest tests -q -m gpu -x -k "128 or cfg4 or cfg5 or cavity or heat or coupled or golden" > $O/tests_h128.log 2>&1; echo "rc=$?" >> $O/tests_h128.log
cat $O/golden_new.txt $O/golden_f32.txt | grep -v "^$" | tail -8
grep -h "merged \|chained \|loss-only \|==" $O/ab_cfg4_*.txt $O/ab_cfg5_*.txt | cut -c1-200
tail -3 $O/tests_h128.log
