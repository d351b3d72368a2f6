#!/bin/bash
# round 3, GPU call 19: per-phase cycle stamps of the level-3 interior kernel (un-merged, stamped variant)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03u
mkdir -p $O
PINN_NO_MERGE=1 timeout 300 python tools/stamp_report.py 3 > $O/stamps.txt 2>&1
cat $O/stamps.txt
