#!/bin/bash
# round 3, GPU call 18: dW by LDS transpose reads (PINN_F2_BF16X=3) — parity on hardware, A/B against level 1 and against merged ds_read2 operand reads
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03t
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -m gpu -x > $O/tests_parity.log 2>&1; echo "rc=$?" >> $O/tests_parity.log
tail -n 6 $O/tests_parity.log
timeout 300 python tools/ab_compare.py head l1 l3m > $O/ab_cfg2.txt 2>&1
timeout 300 python tools/ab_compare.py --cfg cfg3 head l1 l3m > $O/ab_cfg3.txt 2>&1
timeout 300 python tools/ab_compare.py --points 8192 head l1 l3m > $O/ab_cfg2_8192.txt 2>&1
grep round $O/ab_cfg2.txt $O/ab_cfg3.txt $O/ab_cfg2_8192.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['value'],d['roofline']['kernel_ms'],d['loss_terms'])"
