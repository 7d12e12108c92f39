"""CPU: every ctypes mirror in efficientlo-net_amd/_lib.py has the size and the field offsets of the struct of the same
name in include/elo.h (a C program compiled against the header prints them).  A drift between the two is silent memory
corruption on the GPU; this catches it where there is no GPU."""
import ctypes
import os
import subprocess

from conftest import ROOT, load_pkg


def test_ctypes_structs_mirror_the_header(tmp_path):
    L = load_pkg("_lib")
    mirrors = {v.__name__: v for v in vars(L).values()
               if isinstance(v, type) and issubclass(v, ctypes.Structure) and v.__name__.startswith("elo_")}
    assert len(mirrors) >= 25
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "elo.h"', 'int main(void) {']
    for name, cls in sorted(mirrors.items()):
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for field, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, field, name, field))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    want = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, cls in mirrors.items():
        assert ctypes.sizeof(cls) == int(want[name]), name
        for field, _ in cls._fields_:
            assert getattr(cls, field).offset == int(want["%s.%s" % (name, field)]), (name, field)
