"""Instruction mix of one kernel in a hipcc -S listing, and of the basic block(s) that hold the node test.
usage: tools/isa_count.py listing.s <kernel-name-substring>"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
blocks, cur, name = [], [], "entry"
for l in src[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB") and t.split()[0].endswith(":"):
            blocks.append((name, cur)); name, cur = t.split(":")[0], []
        continue
    if re.match(r"^\.?[A-Za-z_0-9]+:", t):
        blocks.append((name, cur)); name, cur = t.split(":")[0], []
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
def kinds(ops):
    k = Counter()
    for o in ops:
        k["valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "lds" if o.startswith("ds_") else
          "vmem" if o.split("_")[0] in ("global", "scratch", "buffer", "flat") else "other"] += 1
    return dict(k)
allops = [o for _, b in blocks for o in b]
print(key, "total", len(allops), kinds(allops))
for n, b in blocks:
    cv = sum(1 for o in b if o.startswith("v_cvt_f32_ubyte"))
    if len(b) >= 40:
        print(f"  block {n}: {len(b)} instr {kinds(b)} cvt_ubyte {cv} fma {sum(1 for o in b if 'fma' in o)} "
              f"minmax {sum(1 for o in b if o.startswith(('v_max', 'v_min')))} cndmask {sum(1 for o in b if 'cndmask' in o)}")
