#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03i
mkdir -p $O
for v in bfx2 bfx; do
PINN_LIB=neuralpde.jl_amd/csrc/abl/libpinn_$v.so timeout 300 python tools/golden_check.py cfg2_full cfg3_full > $O/golden_$v.log 2>&1
PINN_LIB=neuralpde.jl_amd/csrc/abl/libpinn_$v.so timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_$v.log 2>&1
done
PINN_LIB=neuralpde.jl_amd/csrc/abl/libpinn_bfx2.so timeout 300 python tools/ab_env.py --cfg cfg3 --points 262144 --steps 100 > $O/ab_env_cfg3_bfx2.log 2>&1
cat $O/golden_*.log; grep -E "==|merged  |loss-only" $O/ab_env_*.log
