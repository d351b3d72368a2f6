#!/bin/bash
# round 3, GPU call 38: activation adjoint issued between the MFMA groups of the dW GEMM (PINN_F2_ADJ_IL) — parity + A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zn
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -m gpu -x > $O/tests_parity.log 2>&1; echo "rc=$?" >> $O/tests_parity.log
tail -n 3 $O/tests_parity.log
timeout 300 python tools/ab_compare.py head adj0 > $O/ab_cfg2.txt 2>&1
timeout 300 python tools/ab_compare.py --cfg cfg3 head adj0 > $O/ab_cfg3.txt 2>&1
PINN_WG_PER_CU=1 timeout 300 python tools/ab_compare.py head adj0 > $O/ab_cfg2_wg1.txt 2>&1
grep "round\|rror" $O/ab*.txt | sed 's/group1 -1000.0 us//' | cut -c1-160
