"""Per-kernel resource metadata of a built library: VGPRs, SGPRs, spills, scratch, LDS (the .note AMDGPU metadata of every
gfx950 code object in the .hip_fatbin bundles).   python tools/kernel_meta.py [lib.so] [name-filter ...]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_mfma_hazard import code_objects

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "/usr/bin/c++filt"


def kernels(path):
    out = []
    for _triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            text = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in text.split("  - .")[1:]:
            get = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, block) or [None, None])[1]
            name = get("name")
            if name is None or get("vgpr_count") is None:
                continue
            out.append(dict(name=name, vgpr=int(get("vgpr_count")), sgpr=int(get("sgpr_count")), vgpr_spill=int(get("vgpr_spill_count") or 0),
                            sgpr_spill=int(get("sgpr_spill_count") or 0), scratch=int(get("private_segment_fixed_size") or 0),
                            lds=int(get("group_segment_fixed_size") or 0)))
    names = subprocess.run([FILT], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.splitlines()
    for k, n in zip(out, names):
        k["demangled"] = n
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(os.path.dirname(__file__), "..", "efficientlo-net_amd", "libelo_hip.so")
    filters = [a for a in sys.argv[1:] if not a.endswith(".so")]
    print("%5s %5s %7s %7s %8s %6s  kernel" % ("vgpr", "sgpr", "v-spill", "s-spill", "scratch", "lds"))
    for k in sorted(kernels(lib), key=lambda k: k["demangled"]):
        short = re.sub(r"\(anonymous namespace\)::|elo::", "", k["demangled"])
        if filters and not any(f in short for f in filters):
            continue
        print("%5d %5d %7d %7d %8d %6d  %s" % (k["vgpr"], k["sgpr"], k["vgpr_spill"], k["sgpr_spill"], k["scratch"], k["lds"], short[:150]))
