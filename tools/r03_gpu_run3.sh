#!/bin/bash
# round 3, GPU call 3: ping-pong A/B, fp-contract A/B, loss-only test, bench
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mirror.py -q -x -k "ping_pong or merged or loss_only or one_kernel or dgm_burgers_parity" > $O/tests_new.log 2>&1; echo "rc=$?" >> $O/tests_new.log
tail -n 4 $O/tests_new.log
timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_cfg2.log 2>&1
PINN_LIB=neuralpde.jl_amd/csrc/abl/libpinn_fastc.so timeout 300 python tools/ab_env.py --points 65536 > $O/ab_env_cfg2_fastc.log 2>&1
timeout 300 python tools/ab_env.py --cfg cfg3 --points 262144 --steps 100 > $O/ab_env_cfg3.log 2>&1
PINN_PP=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_pp.json 2> $O/bench_pp.err
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
PINN_PP=1 FULL=1 timeout 300 python tools/determinism_check.py cfg2 12 > $O/determinism_pp.log 2>&1
cat $O/ab_env_cfg2.log $O/ab_env_cfg2_fastc.log $O/ab_env_cfg3.log; tail -n 5 $O/determinism_pp.log
