#!/bin/bash
# round 3, GPU call 36: compiler SLP packing of the element-wise phases on / off (-fno-slp-vectorize) — A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zl
mkdir -p $O
timeout 300 python tools/ab_compare.py base noslp > $O/ab_cfg2.txt 2>&1
timeout 300 python tools/ab_compare.py --cfg cfg3 base noslp > $O/ab_cfg3.txt 2>&1
grep "round\|rror" $O/ab*.txt | sed 's/group1 -1000.0 us//' | cut -c1-160
