#!/bin/bash
# round 3, GPU call 39: activation adjoint between the dW MFMA groups in the 128-wide 8-wave kernels too (PINN_F2_ADJ_IL=2) — parity + A/B against H = 64 only
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zo
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -m gpu -x > $O/tests_parity.log 2>&1; echo "rc=$?" >> $O/tests_parity.log
tail -n 3 $O/tests_parity.log
timeout 600 python tools/ab_compare.py --cfg cfg4 head adj1 > $O/ab_cfg4.txt 2>&1
timeout 600 python tools/ab_compare.py --cfg cfg5 head adj1 > $O/ab_cfg5.txt 2>&1
grep "round\|rror" $O/ab*.txt | sed 's/group1 -1000.0 us//' | cut -c1-330
