#!/bin/bash
# round 3, GPU call 31: the full -m gpu suite and the default bench on the committed final tree (after the last experiment switches / reverts)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zg
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
tail -n 4 $O/tests_gpu.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'],d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'],d['cpu_baseline']['value'])"
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -n 2 | cut -c1-200
