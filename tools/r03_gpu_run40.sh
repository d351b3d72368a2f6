#!/bin/bash
# round 3, GPU call 40: per-phase stamps of BOTH members (interior C = 4 / PG = 1 and boundary C = 1 / PG = 4), un-merged and un-chained so that each has its own slabs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zp
mkdir -p $O
PINN_NO_MERGE=1 PINN_NO_CHAIN=1 timeout 300 python tools/stamp_report.py 3 > $O/stamps_both.txt 2>&1
grep -v "^$" $O/stamps_both.txt | head -44
