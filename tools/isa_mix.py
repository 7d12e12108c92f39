"""Instruction mix of one kernel in a gfx950 assembly listing (hipcc --save-temps):
    python tools/isa_mix.py file.s substring-of-the-kernel-name"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if key in l and l.startswith("_Z") and ":" in l)
end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
ins = [l.split()[0] for l in txt[start:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = Counter()
for i in ins:
    k = ("mfma" if i.startswith("v_mfma") else "valu" if i.startswith("v_") else "waitcnt" if i.startswith("s_waitcnt") else
         "s_nop" if i.startswith("s_nop") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else
         "vmem" if i.startswith(("buffer_", "global_", "flat_", "scratch_")) else i)
    c[k] += 1
print(txt[start][:80], len(ins), dict(c))
print(Counter(i for i in ins if i.startswith("v_") and not i.startswith("v_mfma")).most_common(20))
