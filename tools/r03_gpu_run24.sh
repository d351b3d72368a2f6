#!/bin/bash
# round 3, GPU call 24: per-phase stamps of the interior kernel with ONE workgroup per CU (no second wave on the SIMDs) and with two
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03za
mkdir -p $O
PINN_NO_MERGE=1 PINN_WG_PER_CU=1 timeout 300 python tools/stamp_report.py 3 > $O/stamps_wg1.txt 2>&1
PINN_NO_MERGE=1 timeout 300 python tools/stamp_report.py 3 > $O/stamps_wg2.txt 2>&1
cat $O/stamps_wg1.txt $O/stamps_wg2.txt | grep -v "group 1\|^  phase.*w0 *w1\|^$" | head -60
