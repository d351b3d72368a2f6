#!/bin/bash
# round 3, GPU call 12: 128-wide split-operand kernels with whole-layer fragment prefetch and dA before dW, vs the fp32 build
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
V=$(pwd)/neuralpde.jl_amd/csrc/abl/libpinn_h128f32.so
timeout 600 python tools/golden_check.py cfg4_full cfg5_full > $O/golden_new.txt 2>&1
timeout 300 python tools/ab_env.py --cfg cfg4 --points 262144 --steps 40 > $O/ab_cfg4_new.txt 2>&1
PINN_LIB=$V timeout 300 python tools/ab_env.py --cfg cfg4 --points 262144 --steps 40 > $O/ab_cfg4_f32.txt 2>&1
timeout 300 python tools/ab_env.py --cfg cfg5 --points 1000000 --steps 20 > $O/ab_cfg5_new.txt 2>&1
timeout 300 python tools/bench_configs.py cfg4 cfg5 > $O/configs.txt 2>&1
grep -v "^$" $O/golden_new.txt | tail -3
grep -h "merged \|loss-only \|==" $O/ab_cfg4_*.txt $O/ab_cfg5_*.txt | cut -c1-200
grep -v "^    " $O/configs.txt
