#!/bin/bash
# round 3, GPU call 22: three workgroups per CU for the transpose-read kernels (LDS 49 KB per workgroup, <= 168 registers) — A/B
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03y
mkdir -p $O
timeout 300 python tools/ab_compare.py head occ3 occ3slab > $O/ab_cfg2.txt 2>&1
timeout 300 python tools/ab_compare.py --cfg cfg3 head occ3 occ3slab > $O/ab_cfg3.txt 2>&1
timeout 300 python tools/ab_compare.py --points 8192 head occ3 occ3slab > $O/ab_cfg2_8192.txt 2>&1
grep "round\|rror" $O/ab_cfg2.txt $O/ab_cfg3.txt $O/ab_cfg2_8192.txt | cut -c1-200
