#!/bin/bash
# round 3, GPU call 37: is the LDS the limiter?  Variant that reads every forward / dA B-operand fragment twice (+576 ds_read_b64 per tile, +50 % LDS read traffic)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zm
mkdir -p $O
timeout 300 python tools/ab_compare.py base ldsprobe > $O/ab_cfg2.txt 2>&1
PINN_WG_PER_CU=1 timeout 300 python tools/ab_compare.py base ldsprobe > $O/ab_cfg2_wg1.txt 2>&1
grep "round\|rror" $O/ab*.txt | sed 's/group1 -1000.0 us//' | cut -c1-160
