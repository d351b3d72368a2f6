#!/bin/bash
# round 3, GPU call 28: scheduling knobs on the transpose-read kernel — asymmetric GEMM priority, no priorities at all, no forward read-ahead request
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zd
mkdir -p $O
timeout 300 python tools/ab_compare.py base asym noprio ahead0 > $O/ab_cfg2.txt 2>&1
timeout 300 python tools/ab_compare.py --cfg cfg3 base asym noprio ahead0 > $O/ab_cfg3.txt 2>&1
grep "round\|rror" $O/ab*.txt | sed 's/group1 -1000.0 us//' | cut -c1-160
