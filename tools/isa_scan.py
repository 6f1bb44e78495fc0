"""Two patterns that cost this build tens of microseconds before anyone looked, found in the ISA:
  * loops that hold global loads AND wait for all of them (`s_waitcnt vmcnt(0)`) every trip: one memory round trip per trip,
    nothing in flight (the round-1 staging loops of cost_volume / prep_images_s2d);
  * scalar-operand fp32 FMAs: a wave64 v_fma_f32 / v_fmac_f32 with an SGPR source issues every 5.3 cycles on gfx950, 2.9 with
    VGPR sources (v_pk_fma_f32: 4.9 with either; profiles/r03s_filter_ab.txt).
usage: python tools/isa_scan.py file.s [name-filter]      (file.s: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only x.hip)
Prints, per kernel: loops as (instructions, loads, full waits) and the count of FMA instructions with an SGPR source."""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name:
        continue
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if mm and labels.get(mm.group(1), i) < i:
            seg = lines[labels[mm.group(1)]:i]
            nl = sum("global_load" in x or "buffer_load" in x for x in seg)
            nw = sum("vmcnt(0)" in x for x in seg)
            if nl:
                loops.append((len(seg), nl, nw))
    sfma = sum(1 for l in lines if re.search(r"\bv_(fma|fmac|mac)_f32", l) and re.search(r",\s*s\d+|,\s*s\[", l))
    pk = sum(1 for l in lines if "v_pk_fma_f32" in l)
    serial = [lp for lp in loops if lp[2] >= 1 and lp[1] <= 2 * lp[2]]
    flag = ("   <-- load/wait per trip: %s" % serial) if serial else ""
    print("%-72s loops(instr, loads, full waits) %s | v_fma with SGPR source %d, v_pk_fma %d%s" % (name[:72], loops, sfma, pk, flag))
