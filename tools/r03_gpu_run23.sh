#!/bin/bash
# round 3, GPU call 23: one workgroup per CU (one wave per SIMD) against the product's two — how much do two waves per SIMD overlap?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03z
mkdir -p $O
PINN_WG_PER_CU=1 timeout 300 python tools/ab_compare.py head > $O/ab_cfg2_wg1.txt 2>&1
timeout 300 python tools/ab_compare.py head > $O/ab_cfg2_wg2.txt 2>&1
PINN_WG_PER_CU=1 timeout 300 python tools/ab_compare.py --points 8192 head > $O/ab_8192_wg1.txt 2>&1
grep "round\|rror" $O/*.txt | cut -c1-200
# N = 2 functional check of bench.py's multi-process path on ONE GPU (gloo control + data plane, two ranks folded onto the device)
PINN_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
tail -c 600 $O/bench_n2_gloo.json; tail -n 5 $O/bench_n2_gloo.err
