#!/bin/bash
# round 3, GPU call 5: theta-order weights (no pack kernel) A/B against the packed fragment images, full -m gpu suite, bench, traces, PMC
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_cfg2.log 2>&1
PINN_LIB=neuralpde.jl_amd/csrc/abl/libpinn_packw.so timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_cfg2_packw.log 2>&1
timeout 300 python tools/ab_env.py --points 65536 8192 > $O/ab_env_cfg2_again.log 2>&1
cat $O/ab_env_cfg2.log $O/ab_env_cfg2_packw.log $O/ab_env_cfg2_again.log | grep -E "==|merged  |loss-only"
timeout 1200 python -m pytest tests -q -x -m gpu > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
tail -n 6 $O/tests_gpu.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 120 python bench.py --points 8192 --no-cpu-baseline --steps 200 > $O/bench_8192.json 2> $O/bench_8192.err
