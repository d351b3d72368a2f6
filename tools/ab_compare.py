"""A/B timing of several builds of libpinn_hip.so on the same GPU box: the fused kernels of a BASELINE workload, alternating between
the variants neuralpde.jl_amd/csrc/abl/libpinn_<name>.so (built by `make -C neuralpde.jl_amd/csrc variant NAME=<name> VFLAGS=...`;
`head` = the product library).  Usage: python tools/ab_compare.py [--cfg cfg2|cfg3] [--points N] [--precision f64] name1 name2 ...   (two rounds each)
--precision f64: the float64 evaluation mode (pinn_loss_grad_f64, wall time of the whole call) instead of the fp32 kernels' HIP events."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
args = sys.argv[1:]
cfg, points, precision = "cfg2", None, "f32"
while args and args[0].startswith("--"):
    if args[0] == "--cfg": cfg = args[1]
    if args[0] == "--precision": precision = args[1]
    if args[0] == "--points": points = int(args[1])
    args = args[2:]
names = args or ["head"]
for rnd in range(2):
    for tag in names:
        path = None if tag == "head" else f"neuralpde.jl_amd/csrc/abl/libpinn_{tag}.so"
        m._lib.set_library(m.Library(path) if path else None)
        wl = workloads.CONFIGS[cfg](**({"points": points} if points else {}))
        rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
        eng = rep.engine
        if precision == "f64":
            eng.set_option("precision", "f64")
            th64 = np.asarray(rep.flat_init_params, dtype=np.float64)
            wall = []
            for i in range(30):
                t0 = time.perf_counter()
                losses, grad = eng.loss_grad_f64(th64)
                wall.append(time.perf_counter() - t0)
            print(f"{tag:>12s} round {rnd}: float64 evaluation ({eng.get_option('f64_path')}) median {np.median(wall[5:]) * 1e3:.3f} ms  min {np.min(wall[5:]) * 1e3:.3f} ms"
                  f"   checksum {float(np.sum(grad)):.15g}", flush=True)
            del rep, eng
            continue
        th = torch.tensor(rep.flat_init_params, dtype=torch.float32, device="cuda"); out = torch.zeros(eng.P + eng.K, dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream()
        eng.set_timing(1, -1)
        ts, wall = [], []
        for i in range(60):
            t0 = time.perf_counter()
            eng.loss_grad_device(th.data_ptr(), out.data_ptr(), None, st.cuda_stream); torch.cuda.synchronize()
            wall.append(time.perf_counter() - t0)
            ts.append([g["ms"] for g in eng.group_timings()])
        ts = np.array(ts[10:]) * 1e3
        print(f"{tag:>12s} round {rnd}: " + "  ".join(f"group{g} {np.median(ts[:, g]):.1f} us" for g in range(ts.shape[1])) +
              f"   step (events on) {np.median(wall[10:]) * 1e6:.0f} us   checksum {float(out.sum().item()):.9g}", flush=True)
        del rep, eng
