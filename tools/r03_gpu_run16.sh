#!/bin/bash
# round 3, GPU call 16: PeriodicEmbedding tests on hardware (parity mirrors + the reference's CUDA Dirichlet test), bench after the pack fusion
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -x -s -k "periodic" > $O/tests_periodic.log 2>&1; echo "rc=$?" >> $O/tests_periodic.log
grep -i "periodic\|passed\|failed\|rc=" $O/tests_periodic.log | tail
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 120 python bench.py --points 8192 --no-cpu-baseline --steps 200 > $O/bench_8192.json 2> $O/bench_8192.err
python - <<'PY'
import json
for f in ("bench.json", "bench_8192.json"):
    d = json.loads(open("gpurun_out/r03q/" + f).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"])
PY
