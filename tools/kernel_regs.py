#!/usr/bin/env python3
"""Register / spill / LDS table of every kernel in a `hipcc -save-temps` .s file (or of all `build/*.s`)."""
import re, sys
for path in sys.argv[1:]:
    txt = open(path).read()
    for blk in txt.split("  - .agpr_count:")[1:]:
        blk = "  - .agpr_count:" + blk
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        name = re.sub(r"_ZN2pk7k_wave2?INS_5Spec2?", "", name)
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>4s} spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
