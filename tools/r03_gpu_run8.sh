#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03h
mkdir -p $O
PINN_LIB=neuralpde.jl_amd/csrc/abl/libpinn_bfx.so timeout 300 python tools/golden_check.py cfg2_full cfg3_full > $O/golden_bfx.log 2>&1
timeout 300 python tools/golden_check.py cfg2_full cfg3_full > $O/golden_head.log 2>&1
PINN_LIB=neuralpde.jl_amd/csrc/abl/libpinn_bfx.so timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_bfx.log 2>&1
timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_head.log 2>&1
PINN_LIB=neuralpde.jl_amd/csrc/abl/libpinn_bfx.so timeout 300 python tools/ab_env.py --cfg cfg3 --points 262144 --steps 100 > $O/ab_env_cfg3_bfx.log 2>&1
cat $O/golden_bfx.log $O/golden_head.log; grep -E "==|merged  |loss-only" $O/ab_env_bfx.log $O/ab_env_head.log $O/ab_env_cfg3_bfx.log
