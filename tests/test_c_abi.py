"""The C ABI on its own terms: include/elo.h is valid C99 (CPU), and a torch-free C++ program drives
libelo_hip.so with hipMalloc'd buffers and matches the oracle bit for bit (GPU)."""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "efficientlo-net_amd")


def test_header_is_plain_c99(tmp_path):
    src = tmp_path / "use_elo.c"
    src.write_text('#include "elo.h"\nint main(void) { elo_group_args a; elo_cv1_args b; (void)a; (void)b; '
                   'return sizeof(elo_dense) > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])


@pytest.mark.gpu
def test_torch_free_cpp_driver(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libelo_oracle.so"], stdout=subprocess.DEVNULL)
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17",
                           os.path.join(ROOT, "tests", "c_abi", "abi_smoke.cpp"), "-o", exe,
                           "-L", PKG, "-lelo_hip", "-L", os.path.join(ROOT, "oracle"), "-lelo_oracle",
                           "-Wl,-rpath," + PKG, "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr
    assert out.stdout.count("bit-exact") == 2
