#!/bin/bash
# round 3, GPU call 30: end-to-end training runs on the final build (resident-theta Adam): 2-D Poisson bench workload, config-5 heat inverse problem
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zf
mkdir -p $O
timeout 600 python examples/poisson2d_train.py > $O/train_poisson2d.txt 2>&1
tail -n 6 $O/train_poisson2d.txt
timeout 900 python examples/heat_inverse_train.py > $O/train_heat_inverse.txt 2>&1
tail -n 6 $O/train_heat_inverse.txt
