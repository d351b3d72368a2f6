#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_profile.sh (one results database per counter group) into
  * a text table per pass and kernel (dispatch count, mean duration, mean counter value per dispatch), and
  * profiles/pmc_traffic.json — HBM-side bytes per launch of the fused residual kernels
        bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
    FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 the non-32B read requests are 128 B wide while the derived FETCH_SIZE
    expression counts them as 64 B, hence the factor 2 on the read side (/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section).
    bench.py reads this table for its `roofline.traffic` field (key = the kernel name as printed by pinn_describe).

    python tools/pmc_summarize.py <pmc output dir> <summary.txt> [profiles/pmc_traffic.json]
"""
import glob
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


def _one_key(fam, HP, NHH, D, F, PAIRS, NPAIR, PG, HI):
    nfirst = bin(F).count("1")
    lap = (HI >> 24) & 0xFF
    n3 = sum(1 for a in range(6) if ((HI >> (4 * a)) & 0xF) >= 3)
    n4 = sum(1 for a in range(6) if ((HI >> (4 * a)) & 0xF) >= 4)
    C = 1 + nfirst + NPAIR + (1 if lap else 0) + n3 + n4
    return "F%d_HP%d_NHH%d_D%d_F%x_P%x_H%x_L%x_PG%d(C=%d)" % (fam, HP, NHH, D, F, PAIRS, HI & 0xFFFFFF, lap, PG, C)


SPEC_RE = r"pk::Spec2?<(\d+), (\d+), (\d+), (\d+)u?, (\d+)(?:ull|ul|u)?, (\d+), (\d+), (\d+)u?(?:, \d+)?>"       # (r04: family 2 carries the GEMM mode as a 9th parameter)


def spec_key(kernel_name):
    """k_wave2<Spec2<HP,NHH,D,D1MASK,PAIRS,NPAIR,PG,HI>,MODE,ACT> -> the name pinn_describe prints (plan.cpp: spec_name);
    k_wave2m<Spec2<..>,Spec2<..>,ACT> (merged launch, MODE_FUSED) -> "keyA+keyB" as bench.py builds it"""
    m = re.search(r"k_wave2m<" + SPEC_RE + ", " + SPEC_RE + r", (\d+)(?:, (\d+))?>", kernel_name)      # <S0, S1, ACT[, MODE]>
    if m:
        v = [int(m.group(i)) for i in range(1, 18)]
        return _one_key(2, *v[0:8]) + "+" + _one_key(2, *v[8:16]), int(m.group(18) or 0)
    m = re.search(r"k_wave(2?)<" + SPEC_RE + r", (\d+), (\d+)>", kernel_name)
    if not m:
        return None, None
    fam = 2 if m.group(1) else 1
    v = [int(m.group(i)) for i in range(2, 12)]
    return _one_key(fam, *v[0:8]), v[8]


def main():
    src, out_txt = sys.argv[1], sys.argv[2]
    out_json = sys.argv[3] if len(sys.argv) > 3 else None
    per_kernel = defaultdict(dict)          # kernel name -> counter -> mean per dispatch
    lines = []
    for db in sorted(glob.glob(os.path.join(src, "*", "*_results.db"))):
        name = os.path.basename(os.path.dirname(db))
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, counter_name, value, duration, dispatch_id, grid_size, workgroup_size from counters_collection").fetchall()
        agg = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(dict)
        for kn, cn, v, d, disp, gs, ws in rows:
            agg[kn][cn].append(v)
            dur[kn][disp] = d
        lines.append(f"== pass: {name}")
        for kn in sorted(agg, key=lambda k: -sum(dur[k].values())):
            nd = len(dur[kn])
            lines.append(kn)
            lines.append(f"    dispatches {nd}  mean duration {sum(dur[kn].values()) / nd / 1e3:.1f} us")
            for cn, vals in sorted(agg[kn].items()):
                mean = sum(vals) / len(vals)
                lines.append(f"    {cn:<34} mean/dispatch {mean:16.1f}")
                per_kernel[kn][cn] = mean
            per_kernel[kn].setdefault("_dur_us", sum(dur[kn].values()) / nd / 1e3)
        lines.append("")
    with open(out_txt, "w") as f:
        f.write(f"# rocprofv3 --pmc summary of {src} (tools/pmc_profile.sh: one --kernel-trace --pmc pass per counter group, bench.py --steps 6 --warmup 2)\n")
        f.write("\n".join(lines) + "\n")
    if out_json:
        points = int(os.environ.get("PINN_PMC_POINTS", "65536"))
        ks = []
        for kn, cs in per_kernel.items():
            key, mode = spec_key(kn)
            if key is None or mode != 0 or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
                continue
            # points per launch: the interior kernel (C > 1) runs the `points` interior points, the boundary kernel 4 x `points`, the
            # merged launch all five terms' points
            ppl = sum(points if int(c) > 1 else 4 * points for c in re.findall(r"C=(\d+)", key))
            ks.append({"key": key, "kernel": kn, "points_per_launch": ppl,
                       "fetch_kb": cs["FETCH_SIZE"], "write_kb": cs["WRITE_SIZE"],
                       "bytes_per_launch": (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0,
                       "source": os.path.relpath(out_txt, os.path.dirname(os.path.dirname(os.path.abspath(out_json))))})
        with open(out_json, "w") as f:
            json.dump({"formula": "(2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950 wide-read correction)", "kernels": ks}, f, indent=1)
        print("wrote", out_json, len(ks), "kernels")
        # kernel_facts.json next to it: issue-side facts of the fused kernels that bench.py cannot measure in process
        #   mfma_busy            = SQ_VALU_MFMA_BUSY_CYCLES / SIMDs / (GRBM_GUI_ACTIVE / XCDs)        (1,024 SIMDs, 8 XCDs on MI355X)
        #   valu_insts_per_tile  = (SQ_INSTS_VALU - SQ_INSTS_MFMA) per tile and wave (64-row tiles, waves per tile from the kernel family)
        #   lds_array_busy       = SQ_LDS_IDX_ACTIVE / CUs / (GRBM_GUI_ACTIVE / XCDs)
        #   wait_inst_frac       = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
        facts = []
        for kn, cs in per_kernel.items():
            key, mode = spec_key(kn)
            if key is None or mode != 0 or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
                continue
            ppl = sum(points if int(c) > 1 else 4 * points for c in re.findall(r"C=(\d+)", key))
            gui = cs["GRBM_GUI_ACTIVE"] / 8.0
            e = {"key": key, "points_per_launch": ppl, "mfma_busy": cs["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / gui,
                 "source": os.path.relpath(out_txt, os.path.dirname(os.path.dirname(os.path.abspath(out_json))))}
            if "SQ_INSTS_VALU" in cs and "SQ_INSTS_MFMA" in cs:
                # rows per tile = 64 (16 points x C channels for C = 4, 64 points x 1 channel for C = 1); HP = 64: 4 waves per tile
                tiles = sum((points * int(c) if int(c) > 1 else 4 * points) // 64 for c in re.findall(r"C=(\d+)", key))
                nw = 8 if "HP128" in key else 4
                e["valu_insts_per_tile"] = (cs["SQ_INSTS_VALU"] - cs["SQ_INSTS_MFMA"]) / (tiles * nw)
                e["mfma_insts_per_tile"] = cs["SQ_INSTS_MFMA"] / (tiles * nw)
            if "SQ_LDS_IDX_ACTIVE" in cs:
                e["lds_array_busy"] = cs["SQ_LDS_IDX_ACTIVE"] / 256.0 / gui
            if "SQ_WAIT_INST_ANY" in cs and "SQ_WAVE_CYCLES" in cs:
                e["wait_inst_frac"] = cs["SQ_WAIT_INST_ANY"] / cs["SQ_WAVE_CYCLES"]
            facts.append(e)
        fj = os.path.join(os.path.dirname(os.path.abspath(out_json)), "kernel_facts.json")
        with open(fj, "w") as f:
            json.dump({"kernels": facts}, f, indent=1)
        print("wrote", fj, len(facts), "kernels")


if __name__ == "__main__":
    main()
