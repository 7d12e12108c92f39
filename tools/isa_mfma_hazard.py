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

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
NEED = 5                       # wait states measured on MI355X (tools/micro/mfma_srcc_hazard.hip); the guard holds 6
WIDE = "v_mfma_f32_16x16x32_f16"
NARROW = "v_mfma_f32_16x16x16_f16"
HORIZON = 8                    # wait states after which a write is forgotten


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
    """[(index of the narrow MFMA, its text, wait states since the wide one)] for one kernel."""
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
            if op == NARROW and len(ops) >= 4:
                srcc = set(_regs(ops[3].split()[0]))
                for dst, ws in st.items():
                    if srcc & set(dst) and ws < NEED:
                        found[i] = (i, "%s %s" % (op, ", ".join(ops)), ws)
            # measured: ONE other MFMA between the two is enough (the matrix pipe is in order: by the time the narrow one
            # starts, the wide one has written back); every other instruction is one wait state, `s_nop n` n + 1
            step = NEED if op.startswith("v_mfma") else int(ops[0], 0) + 1 if op == "s_nop" and ops else 1
            st = {k: v + step for k, v in st.items() if v + step <= HORIZON}
            if op.startswith("v_") and ops:                       # any other write of those registers ends the hazard
                d = set(_regs(ops[0]))
                st = {k: v for k, v in st.items() if not (d & set(k))}
            if op == WIDE:
                st[_regs(ops[0])] = 0
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
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "efficientlo-net_amd")
    libs = argv or [os.path.join(pkg, "libelo_hip.so"), os.path.join(pkg, "libelo_hip_f32.so")]
    bad = 0
    for lib in libs:
        hits = check_text(open(lib).read()) if lib.endswith(".s") else check_library(lib)
        print("%s: %d narrow MFMAs within %d wait states of the wide MFMA whose result they accumulate onto" % (lib, len(hits), NEED))
        for name, i, text, ws in hits[:40]:
            print("  %s  #%d  %d wait states: %s" % (name[:90], i, ws, text))
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
