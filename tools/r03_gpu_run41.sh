#!/bin/bash
# round 3, GPU call 41: adjoint pieces pinned between the dW MFMA groups by an ordered use (pin_value) — parity + A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zq
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/ab_compare.py head pin0 > $O/ab_cfg2.txt 2>&1
timeout 300 python tools/ab_compare.py --cfg cfg3 head pin0 > $O/ab_cfg3.txt 2>&1
grep "round\|rror" $O/ab*.txt | sed 's/group1 -1000.0 us//' | cut -c1-160
timeout 300 python -m pytest tests/test_gpu_full_size.py -q -m gpu -x 2>&1 | tail -n 2
