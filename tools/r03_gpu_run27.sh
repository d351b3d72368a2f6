#!/bin/bash
# round 3, GPU call 27: cfg4 with recompute (MODE_FWD + MODE_GRADIN, PINN_REC_GB=0) against records in HBM (MODE_FWDREC + MODE_GRADREC) on the final kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zc
mkdir -p $O
timeout 300 python tools/bench_configs.py cfg4 > $O/cfg4_records.txt 2>&1
PINN_REC_GB=0 timeout 300 python tools/bench_configs.py cfg4 > $O/cfg4_recompute.txt 2>&1
grep -v "^    " $O/cfg4_records.txt | tail -2; grep -v "^    " $O/cfg4_recompute.txt | tail -2
