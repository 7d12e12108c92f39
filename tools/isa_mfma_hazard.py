"""Static check of a built library for the gfx950 MFMA hazard hipcc does not know (finding 4;
tools/micro/mfma_srcc_hazard.hip): a v_mfma_f32_16x16x16_f16 that reads, as SrcC, the vDst of a v_mfma_f32_16x16x32_f16
issued fewer than 5 wait states earlier reads the accumulator as it was before that instruction's update.

    python tools/isa_mfma_hazard.py [path/to/libelo_hip.so ...]

Pulls every gfx950 code object out of the library's .hip_fatbin bundles, disassembles it (llvm-objdump) and walks every
kernel's control-flow graph with the state "wait states since each 16x16x32 MFMA wrote its vDst" (the minimum over the
paths into a block; an instruction is one wait state, `s_nop n` is n + 1, another MFMA in between clears the hazard: all
three as measured by tools/micro/mfma_srcc_hazard.hip), to a fixed point.  Prints and returns the
violating pairs.  tests/test_isa_hazard_cpu.py runs it on the shipped libraries."""
import os
import re
import struct
import subprocess
import sys
import tempfile

import json

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
NEED = 5                       # wait states measured on MI355X (tools/micro/mfma_srcc_hazard.hip); the guard holds 6
WIDE = "v_mfma_f32_16x16x32_f16"
NARROW = "v_mfma_f32_16x16x16_f16"
TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_srcc_hazard_table.json")


def load_table(path=TABLE_PATH):
    """{(first mnemonic, second mnemonic): {"exact": n, "partial": n}} -- wait states the SECOND needs behind the FIRST when it reads
    the first's vDst as SrcC (exact: SrcC is that register tuple -- the case LLVM asks 0 wait states for; partial: it only overlaps it),
    as measured over the shape cross-product by tools/mfma_hazard_matrix.sh (profiles/r05_mfma_srcc_hazard.txt); pairs that need 0
    are left out.  Without the file: the one pair round 4 measured."""
    try:
        rows = json.load(open(path))["pairs"]
    except (OSError, ValueError, KeyError):
        return {(WIDE, NARROW): {"exact": NEED, "partial": NEED + 1}}
    return {(r["first"], r["second"]): {"exact": r["exact"], "partial": r["partial"]} for r in rows}


def make_table(matrix_txt, out=TABLE_PATH):
    """Fold the PAIR lines of tools/mfma_hazard_matrix.sh's output into the checker's table (mnemonics from the generator)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro"))
    import gen_mfma_srcc_matrix as gen
    mn = {sh[0]: sh[1] for sh in gen.SHAPES}
    need = {}
    for line in open(matrix_txt):
        if line.startswith("PAIR "):
            _, a, b, rel, n = line.split()
            d = need.setdefault((mn[a], mn[b]), {"exact": 0, "partial": 0})
            k = "partial" if rel == "partial" else "exact"
            d[k] = max(d[k], int(n.split("=")[1]))
    rows = [{"first": a, "second": b, **d} for (a, b), d in sorted(need.items()) if d["exact"] or d["partial"]]
    json.dump({"source": os.path.basename(matrix_txt), "measured_on": "MI355X (gfx950), ROCm 7.2", "shapes": sorted(mn.values()), "pairs": rows},
              open(out, "w"), indent=1)
    return rows


TABLE = load_table()
FIRSTS = {a for a, _ in TABLE}
HORIZON = max([max(v.values()) for v in TABLE.values()] + [8]) + 1     # wait states after which a write is forgotten


def code_objects(path):
    """The device ELF images inside the library's clang offload bundles."""
    data = open(path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        at = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, at)
            triple = data[at + 24:at + 24 + tlen].decode()
            at += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, data[base + off:base + off + size]))
    return out


def disassemble(blob):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob)
        f.flush()
        return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], check=True, capture_output=True, text=True).stdout


def _regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return tuple(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return (int(m.group(1)),) if m else ()


def functions(text):
    """{name: [(address-or-index, mnemonic, [operands], label-or-None)]} from llvm-objdump -d or from a hipcc -S listing."""
    funcs, cur, pending = {}, None, None
    for raw in text.splitlines():
        line = raw.split("//")[0].split(";")[0].rstrip()
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line) or re.match(r"^([A-Za-z_.$][\w.$]*):\s*$", line)
        if m:
            name = m.group(1)
            if name.startswith((".L", "L")) and cur is not None and not re.match(r"^[0-9a-f]+ <", line):
                pending = name                              # a block label of a -S listing
            elif re.match(r"^[0-9a-f]+ <L\d+>:$", line) and cur is not None:
                pending = name                              # a block label of a disassembly
            else:
                cur = funcs.setdefault(name, [])
                pending = None
            continue
        s = line.strip()
        if cur is None or not s or s.startswith("."):
            continue
        parts = s.split(None, 1)
        ops = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((len(cur), parts[0], ops, pending))
        pending = None
    return funcs


def _target(ops):
    t = ops[-1] if ops else ""
    m = re.search(r"<(L\d+)>", t) or re.match(r"^([.\w$]+)$", t)
    return m.group(1) if m else None


def check_function(ins):
    """[(index of the reading MFMA, its text, wait states since the MFMA whose vDst it reads as SrcC)] for one kernel: every
    (first, second) pair of the measured table, exact and partially overlapping SrcC alike."""
    label_at = {lab: i for i, (_, _, _, lab) in enumerate(ins) if lab}
    leaders = sorted({0} | set(label_at.values()) | {i + 1 for i, (_, op, _, _) in enumerate(ins)
                                                      if op.startswith(("s_cbranch", "s_branch")) and i + 1 < len(ins)})
    block_of = {}
    for b, start in enumerate(leaders):
        end = leaders[b + 1] if b + 1 < len(leaders) else len(ins)
        for i in range(start, end):
            block_of[i] = b
    state_in = {0: {}}
    work, found = [0], {}

    def merge(b, st):
        old = state_in.get(b)
        if old is None:
            state_in[b] = dict(st)
            return True
        changed = False
        for k, v in st.items():
            if k not in old or v < old[k]:
                old[k] = v
                changed = True
        return changed

    while work:
        b = work.pop()
        start = leaders[b]
        end = leaders[b + 1] if b + 1 < len(leaders) else len(ins)
        st = dict(state_in[b])
        fall = True
        for i in range(start, end):
            _, op, ops, _ = ins[i]
            base = op[:-4] if op.endswith("_e64") else op
            if base.startswith("v_mfma") and len(ops) >= 4:
                srcc = _regs(ops[3].split()[0])
                for (dst, first), ws in st.items():
                    rule = TABLE.get((first, base))
                    if rule and set(srcc) & set(dst):
                        need = rule["exact"] if tuple(srcc) == tuple(dst) else max(rule["partial"], rule["exact"])
                        if ws < need:
                            found[i] = (i, "%s %s" % (op, ", ".join(ops)), ws)
            # measured: ONE other MFMA between the two is enough (the matrix pipe is in order: by the time the second one
            # starts, the first has written back) -- except the 2-pass 4x4x4, counted as 4; every other instruction is one wait
            # state, `s_nop n` n + 1
            step = (4 if "4x4x4" in base else HORIZON) if base.startswith("v_mfma") else int(ops[0], 0) + 1 if op == "s_nop" and ops else 1
            st = {k: v + step for k, v in st.items() if v + step <= HORIZON}
            if op.startswith("v_") and ops:                       # any other write of those registers ends the hazard
                d = set(_regs(ops[0]))
                st = {k: v for k, v in st.items() if not (d & set(k[0]))}
            if base in FIRSTS:
                st[(_regs(ops[0]), base)] = 0
            if op.startswith(("s_cbranch", "s_branch")):
                t = _target(ops)
                if t in label_at and merge(block_of[label_at[t]], st):
                    work.append(block_of[label_at[t]])
                if op.startswith("s_branch"):
                    fall = False
            if op in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and end < len(ins) and merge(block_of[end], st):
            work.append(block_of[end])
    return sorted(found.values())


def check_text(text):
    out = []
    for name, ins in functions(text).items():
        for hit in check_function(ins):
            out.append((name,) + hit)
    return out


def check_library(path):
    hits = []
    for triple, blob in code_objects(path):
        hits += check_text(disassemble(blob))
    return hits


def main(argv):
    if argv and argv[0] == "--make-table":
        rows = make_table(argv[1])
        print("%d pairs with a hazard written to %s" % (len(rows), TABLE_PATH))
        return 0
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "efficientlo-net_amd")
    libs = argv or [os.path.join(pkg, "libelo_hip.so"), os.path.join(pkg, "libelo_hip_f32.so")]
    bad = 0
    for lib in libs:
        hits = check_text(open(lib).read()) if lib.endswith(".s") else check_library(lib)
        print("%s: %d MFMAs read as SrcC the result of an MFMA of another shape inside the measured hazard window (%d pairs in the table)"
              % (lib, len(hits), len(TABLE)))
        for name, i, text, ws in hits[:40]:
            print("  %s  #%d  %d wait states: %s" % (name[:90], i, ws, text))
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
