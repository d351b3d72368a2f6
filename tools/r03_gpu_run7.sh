#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03g
mkdir -p $O
PINN_NO_MERGE=1 PINN_NO_CHAIN=1 timeout 300 python tools/stamp_report.py > $O/stamps.txt 2>&1
cat $O/stamps.txt
