#!/bin/bash
# round 3, GPU call 10: split-operand default build — full -m gpu suite, bench (+ per-rank share), goldens at full size
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x > $O/tests_gpu.log 2>&1; echo "rc=$?" >> $O/tests_gpu.log
tail -n 8 $O/tests_gpu.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 120 python bench.py --points 8192 --no-cpu-baseline --steps 200 > $O/bench_8192.json 2> $O/bench_8192.err
timeout 300 python tools/golden_check.py > $O/golden.txt 2>&1
python - <<'PY'
import json
for f in ("bench.json", "bench_8192.json"):
    try:
        d = json.loads(open("gpurun_out/r03j/" + f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("frac_mixed_pipes"), d["loss_only_host_entry_ms"], d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -12 $O/golden.txt
