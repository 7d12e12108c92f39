"""Static attribution of a kernel's vector instructions to source lines (line tables of a -gline-tables-only -S listing):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only --cuda-device-only -S csrc/elo_fused.hip -o /tmp/f.s
    python tools/isa_by_line.py /tmp/f.s <mangled-name-substring> [top]
Prints VALU (without MFMA) / MFMA / SALU counts per (file, line) of the kernel whose symbol contains the substring, largest
first, with the source text.  Inlined code is attributed to the innermost line (the callee's), which is what one edits."""
import collections, re, sys

listing, want = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files, cur_fn, loc = {}, None, None
agg = collections.defaultdict(lambda: [0, 0, 0])
for raw in open(listing):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', raw)
    if m:
        files[int(m.group(1))] = (m.group(3) if m.group(3).startswith("/") else m.group(2) + "/" + m.group(3)) if m.group(3) else m.group(2)
        continue
    m = re.match(r"^(_Z\S+):", raw)
    if m:
        cur_fn = m.group(1)
        continue
    if cur_fn is None or want not in cur_fn:
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", raw)
    if m:
        loc = (int(m.group(1)), int(m.group(2)))
        continue
    if ".Lfunc_end" in raw:
        cur_fn = None
        continue
    t = raw.strip()
    if not raw.startswith("\t") or t.startswith((".", ";")) or loc is None:
        continue
    op = t.split()[0]
    if "mfma" in op:
        agg[loc][1] += 1
    elif op.startswith("v_"):
        agg[loc][0] += 1
    elif op.startswith("s_"):
        agg[loc][2] += 1
src = {}
def text(f, l):
    path = files.get(f, "?")
    if path not in src:
        try:
            src[path] = open(path).read().split("\n")
        except OSError:
            src[path] = []
    lines = src[path]
    return lines[l - 1].strip()[:110] if 0 < l <= len(lines) else ""
tot = [sum(v[i] for v in agg.values()) for i in range(3)]
print("VALU %d  MFMA %d  SALU %d" % tuple(tot))
for (f, l), v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%5d %4d %4d  %s:%d  %s" % (v[0], v[1], v[2], files.get(f, "?").split("/")[-1], l, text(f, l)))
