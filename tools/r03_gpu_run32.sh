#!/bin/bash
# round 3, GPU call 32: family 1 (one wave per tile) — bias + first weight fragment of a layer requested before the activation of the layer below — A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zh
mkdir -p $O
timeout 300 python tools/ab_compare.py --cfg cfg1 f1pre1 f1pre0 > $O/ab_cfg1.txt 2>&1
grep "round\|rror" $O/ab_cfg1.txt | cut -c1-160
for v in f1pre1 f1pre0; do
python - $v > $O/small_$v.txt 2>&1 <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import pinn_import
m = pinn_import.load()
m._lib.set_library(m.Library(f"neuralpde.jl_amd/csrc/abl/libpinn_{sys.argv[1]}.so"))
from neuralpde_jl_amd import workloads
for width, hidden, pts in ((16, 2, 1024), (32, 2, 1024), (32, 3, 4096), (32, 3, 262144)):
    wl = workloads.cfg2_poisson2d(points=pts, bcs_points=max(64, pts // 4), width=width, hidden=hidden)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    th = torch.tensor(rep.flat_init_params, dtype=torch.float32, device="cuda"); out = torch.zeros(eng.P + eng.K, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    for _ in range(30): eng.loss_grad_device(th.data_ptr(), out.data_ptr(), None, st.cuda_stream)
    ts = []
    for rep_ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): eng.loss_grad_device(th.data_ptr(), out.data_ptr(), None, st.cuda_stream)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100)
    print(f"{sys.argv[1]} {width:3d} x {hidden} {pts:7d} points: {np.median(ts) * 1e6:8.1f} us per evaluation (back-to-back launches)  checksum {float(out.sum()):.7g}")
PY
cat $O/small_$v.txt | grep points
done
