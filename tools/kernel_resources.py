#!/usr/bin/env python3
"""Print VGPR/AGPR/SGPR/scratch/LDS of every kernel from a hipcc --save-temps gfx950 .s file."""
import re, sys
t = open(sys.argv[1]).read()
for blk in t.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
    short = re.sub(r"EvNS_4Pool.*", "", short)[:48]
    print(f"{short:50s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>6s} lds {g('group_segment_fixed_size'):>6s} spill {g('vgpr_spill_count')}")
