#!/usr/bin/env python3
"""Build check for engine.hip: in the streaming waves' code, outside the inline-assembly statements, no instruction may
touch v48 .. v167 -- the registers the asm steps own (x of the running batch, the ring, the accumulators).  The asm
statements clobber them, so hipcc keeps no value there across a statement; this verifies it does not park one there
BETWEEN two statements either (a load still in flight would overwrite it).
usage: check_engine_regs.py engine.s   (hipcc -S --cuda-device-only output)"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if "L2Z_STREAM_BEGIN" in l)
# the streaming waves' blocks lie between their marker and the gatherer's (its blocks are laid out behind them), or the
# end of the function if the gatherer came first
g = next((i for i, l in enumerate(lines) if "L2Z_GATHER_BEGIN" in l), -1)
end = g if g > start else next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
in_asm, bad = False, []
def regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out
for i in range(start, end):
    t = lines[i].strip()
    if t.startswith(";;#ASMSTART"): in_asm = True; continue
    if t.startswith(";;#ASMEND"): in_asm = False; continue
    if in_asm or not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
    r = {x for x in regs(t.split(";")[0]) if 48 <= x <= 167}
    if r: bad.append((i + 1, t))
print(f"check_engine_regs: {end - start} lines of the streaming waves checked, {len(bad)} instruction(s) outside asm touch v48..v167")
for i, t in bad[:40]: print(f"  line {i}: {t}")
sys.exit(1 if bad else 0)
