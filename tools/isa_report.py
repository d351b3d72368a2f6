#!/usr/bin/env python3
"""Static ISA report of the kernels in a `hipcc -save-temps` .s file (gfx950): registers / spills / LDS, the instruction mix, the mix per
workgroup-barrier segment and the issue pattern of the GEMM regions (operand reads vs MFMAs) — the evidence behind DESIGN.md §4.1's
statements about the merged residual kernel.

    cd /tmp/x && hipcc -O3 -std=c++17 --offload-arch=gfx950 -I<repo>/neuralpde.jl_amd/csrc/build -save-temps -c <repo>/neuralpde.jl_amd/csrc/inst2m_h64_d2.hip
    python tools/isa_report.py /tmp/x/inst2m_h64_d2-hip-amdgcn-amd-amdhsa-gfx950.s [name-substring] [max kernels]

pattern legend: r ds_read_b64, R ds_read_b128, 2 ds_read2*, T ds_read_b64_tr_b16, s LDS store, M bf16 MFMA, F fp32 MFMA, . other VALU, / scheduling fence, |B| s_barrier"""
import collections, re, sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_wave2m"
maxk = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lines = open(path).read().split("\n")
meta = {}
txt = "\n".join(lines)
for blk in txt.split("  - .agpr_count:")[1:]:
    blk = "  - .agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    meta[g("name")] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), spill=g("vgpr_spill_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l[:1] == "_" and want in l and ":" in l and not l.startswith("\t")]
seen, shown = set(), 0
for i, name in starts:
    if name in seen:
        continue
    seen.add(name)
    body = []
    for l in lines[i + 1:]:
        body.append(l)
        if "s_endpgm" in l:
            break
    ops = [m.group(1) for m in (re.match(r"\s+([a-z_0-9]+)", l) for l in body) if m]
    if sum(1 for o in ops if o.startswith("v_mfma")) < 600:      # forward-only / loss-only variants: skip, the fused kernels are the subject
        continue
    shown += 1
    if shown > maxk:
        break
    c = collections.Counter(ops)
    m = meta.get(name, {})
    print(f"== {name[:110]}...")
    print(f"   registers: vgpr {m.get('vgpr')} agpr {m.get('agpr')} spilled {m.get('spill')} scratch {m.get('scratch')} B   LDS {m.get('lds')} B   instructions {len(ops)}")
    fam = collections.Counter()
    for o, n in c.items():
        k = ("MFMA bf16" if o.startswith("v_mfma") and "bf16" in o else "MFMA fp32" if o.startswith("v_mfma") else "VALU packed" if o.startswith("v_pk") else
             "VALU transcendental" if o.startswith(("v_exp", "v_rcp", "v_log", "v_sin", "v_cos", "v_sqrt", "v_rsq")) else "VALU other" if o.startswith("v_") else
             "LDS transpose read" if "tr_b16" in o else "LDS read" if o.startswith("ds_read") else "LDS write" if o.startswith("ds_write") else "LDS other" if o.startswith("ds_") else
             "vector memory" if o.startswith(("buffer_", "global_", "scratch_", "flat_")) else "s_waitcnt" if o == "s_waitcnt" else "s_nop" if o == "s_nop" else
             "s_barrier" if o == "s_barrier" else "scalar other")
        fam[k] += n
    print("   mix: " + ", ".join(f"{k} {n}" for k, n in sorted(fam.items(), key=lambda kv: -kv[1])))
    print("   (static counts of the whole kernel: both members' tile bodies, the tape interpreter that affine residuals skip, prologue and epilogue)")
    seg, segs = collections.Counter(), []
    pat, pats = [], []
    for l in body:
        mm = re.match(r"\s+([a-z_0-9]+)", l)
        if "sched_barrier" in l:
            pat.append("/")
        if not mm:
            continue
        o = mm.group(1)
        if o == "s_barrier":
            segs.append(seg); seg = collections.Counter(); pats.append("".join(pat)); pat = []
            continue
        if o.startswith("v_mfma"):
            seg["mfma"] += 1; pat.append("M" if "bf16" in o else "F")
        elif o.startswith("v_"):
            seg["valu"] += 1; pat.append(".")
        elif "tr_b16" in o:
            seg["lds"] += 1; pat.append("T")
        elif o.startswith("ds_read_b64"):
            seg["lds"] += 1; pat.append("r")
        elif o.startswith("ds_read_b128"):
            seg["lds"] += 1; pat.append("R")
        elif o.startswith("ds_read2"):
            seg["lds"] += 1; pat.append("2")
        elif o.startswith("ds_write"):
            seg["lds"] += 1; pat.append("s")
        elif o.startswith("ds_"):
            seg["lds"] += 1
        elif o.startswith(("buffer_", "global_", "scratch_")):
            seg["vmem"] += 1
    segs.append(seg); pats.append("".join(pat))
    print("   segments between workgroup barriers (MFMA / other VALU / LDS / vector-memory instructions):")
    for j, sg in enumerate(segs):
        print(f"     {j:2d}: mfma {sg['mfma']:4d}  valu {sg['valu']:5d}  lds {sg['lds']:4d}  vmem {sg['vmem']:3d}")
    gem = [p for p in pats if p.count("M") + p.count("F") >= 40]
    print("   issue pattern of the GEMM segments (VALU runs compressed):")
    for p in gem[:12]:
        print("     " + re.sub(r"\.{4,}", lambda m_: f".[{len(m_.group(0))}]", p)[:700])
