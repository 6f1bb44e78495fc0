"""Register / spill table of every kernel in a `-save-temps=obj` gfx950 assembly file (.amdhsa_kernel metadata blocks).
usage: python tools/asm_regs.py file.s [name-filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa.target|\Z)", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt and flt not in name:
        continue
    print("%-70s vgpr %4s agpr %3s sgpr %4s spill v %3s s %3s lds %7s scratch %5s" % (
        name[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
        g("group_segment_fixed_size"), g("private_segment_fixed_size")))
