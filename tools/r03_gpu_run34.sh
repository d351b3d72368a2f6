#!/bin/bash
# round 3, GPU call 34: bench.py --gpus 2 on ONE GPU with the default (RCCL) data plane: both RCCL paths must fail cleanly ("Duplicate GPU") and the
# run must fall back to the host reduction over gloo and still print its line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zj
mkdir -p $O
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_n2_fallback.json 2> $O/bench_n2_fallback.err
echo "rc=$?"
tail -c 400 $O/bench_n2_fallback.json; echo; grep -i "bench rank\|error\|duplicate" $O/bench_n2_fallback.err | head -8
