"""The shipped libraries hold no 16x16x16 MFMA that accumulates onto a 16x16x32 MFMA's result within the 5 wait states
gfx950 needs and hipcc does not insert (finding 4; measured by tools/micro/mfma_srcc_hazard.hip;
guard: mfma_shape_guard in csrc/elo_fused.hip).  The check is static (disassembly of the built code objects), so it
runs without a GPU -- a rebuild whose scheduling puts such a pair back fails here, not as a wrong row on some input."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_mfma_hazard", os.path.join(ROOT, "tools", "isa_mfma_hazard.py"))
H = importlib.util.module_from_spec(spec)
spec.loader.exec_module(H)

PAIR = """
0000000000001000 <kernel_a>:
	v_mfma_f32_16x16x32_f16 v[22:25], v[30:33], v[34:37], v[22:25]
	ds_read_b128 v[30:33], v73 offset:10240
	s_waitcnt lgkmcnt(1)
	v_mfma_f32_16x16x16_f16 v[38:41], v[40:41], v[62:63], v[22:25]
	s_endpgm
"""


def test_the_checker_flags_the_pair_the_bisect_ended_at():
    hits = H.check_text(PAIR)
    assert len(hits) == 1 and hits[0][0] == "kernel_a" and hits[0][3] == 2


@pytest.mark.parametrize("between,flagged", [("s_nop 3", True), ("s_nop 4", False), ("s_nop 5", False),
                                             ("v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]", False),
                                             ("v_mov_b32 v1, v2\n\tv_mov_b32 v1, v2\n\tv_mov_b32 v1, v2\n\tv_mov_b32 v1, v2", True),
                                             ("v_mov_b32 v22, 0", False)])
def test_what_counts_as_distance(between, flagged):
    """s_nop n is n + 1 wait states, another MFMA in between clears the hazard, an overwrite of the accumulator ends it."""
    text = PAIR.replace("\tds_read_b128 v[30:33], v73 offset:10240\n\ts_waitcnt lgkmcnt(1)", "\t" + between)
    assert bool(H.check_text(text)) == flagged


def test_the_hazard_is_followed_across_branches():
    text = """
0000000000001000 <kernel_b>:
	v_mfma_f32_16x16x32_f16 v[22:25], v[30:33], v[34:37], v[22:25]
	s_cbranch_vccnz 12 <L1>
	s_nop 7
0000000000001010 <L1>:
	v_mfma_f32_16x16x16_f16 v[22:25], v[40:41], v[62:63], v[22:25]
	s_endpgm
"""
    hits = H.check_text(text)
    assert len(hits) == 1 and hits[0][3] == 1


@pytest.mark.parametrize("first,second,srcc,gap,flagged", [
    # round 5: the full shape cross-product (tools/mfma_hazard_matrix.sh, profiles/r05_mfma_srcc_hazard.txt).  SrcC == the first's vDst:
    ("v_mfma_f32_16x16x32_bf16 v[22:25], v[30:33], v[34:37], v[22:25]", "v_mfma_f32_32x32x8_f16 v[40:55], v[60:61], v[62:63], {c}", "v[22:37]", "s_nop 3", True),
    ("v_mfma_f32_16x16x32_f16 v[22:25], v[30:33], v[34:37], v[22:25]", "v_mfma_f32_4x4x4_16b_f16 v[40:43], v[60:61], v[62:63], {c}", "v[22:25]", "s_nop 3", True),
    ("v_mfma_f32_16x16x32_f16 v[22:25], v[30:33], v[34:37], v[22:25]", "v_mfma_f32_4x4x4_16b_f16 v[40:43], v[60:61], v[62:63], {c}", "v[22:25]", "s_nop 4", False),
    # the gfx950 double-rate shapes behind each other and the 32x32 shapes in front of anything are interlocked when SrcC IS the vDst ...
    ("v_mfma_f32_16x16x32_f16 v[22:25], v[30:33], v[34:37], v[22:25]", "v_mfma_f32_16x16x32_bf16 v[40:43], v[30:33], v[34:37], {c}", "v[22:25]", "s_nop 0", False),
    ("v_mfma_f32_32x32x16_f16 v[0:15], v[30:33], v[34:37], v[0:15]", "v_mfma_f32_16x16x16_f16 v[40:43], v[60:61], v[62:63], {c}", "v[0:3]", "s_nop 0", False),
    # ... but not when it only OVERLAPS it: same shape, SrcC shifted by two registers needs 7 wait states
    ("v_mfma_f32_16x16x32_f16 v[22:25], v[30:33], v[34:37], v[22:25]", "v_mfma_f32_16x16x32_f16 v[40:43], v[30:33], v[34:37], {c}", "v[24:27]", "s_nop 5", True),
    ("v_mfma_f32_16x16x32_f16 v[22:25], v[30:33], v[34:37], v[22:25]", "v_mfma_f32_16x16x32_f16 v[40:43], v[30:33], v[34:37], {c}", "v[24:27]", "s_nop 6", False),
    ("v_mfma_f32_16x16x4_f32 v[22:25], v30, v31, v[22:25]", "v_mfma_f32_16x16x4_f32 v[40:43], v30, v31, {c}", "v[24:27]", "s_nop 7", True),
    ("v_mfma_f32_16x16x4_f32 v[22:25], v30, v31, v[22:25]", "v_mfma_f32_16x16x4_f32 v[40:43], v30, v31, {c}", "v[22:25]", "s_nop 0", False)])
def test_every_measured_pair_is_in_the_table(first, second, srcc, gap, flagged):
    """The checker knows every (first, second) pair the cross-product measured, for SrcC == vDst and for a partial overlap (a
    sub-range of a wider vDst is a partial overlap; a 32x32 MFMA in front is interlocked either way: measured 0)."""
    text = "0000000000001000 <kernel_c>:\n\t%s\n\t%s\n\t%s\n\ts_endpgm\n" % (first, gap, second.format(c=srcc))
    assert bool(H.check_text(text)) == flagged, H.check_text(text)
    assert len(H.TABLE) >= 27 and H.TABLE[("v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16")]["exact"] == 5


@pytest.mark.parametrize("lib", ["libelo_hip.so", "libelo_hip_f32.so"])
def test_shipped_libraries_are_free_of_the_pair(lib):
    path = os.path.join(ROOT, "efficientlo-net_amd", lib)
    if not os.path.exists(path):
        pytest.skip("library not built")
    objs = H.code_objects(path)
    assert objs and all("gfx950" in triple for triple, _ in objs)
    hits = H.check_library(path)
    assert not hits, hits[:5]


def test_inference_kernels_keep_their_accumulators_in_vgprs_and_their_dpp_maxima_in_one_instruction():
    """Two compiler choices round 4 found by counting instructions (profiles/r04_instruction_diet.txt), pinned statically:
    (1) a kernel allowed 512 registers gets the AGPR form of the matrix instructions and pays v_accvgpr_write / _read around
    every short accumulator chain -- every kernel of csrc/elo_fused.hip carries a launch bound that keeps it on the VGPR form;
    (2) an in-wave maximum step is ONE v_max_i32_dpp when update_dpp's `old` is the identity: the pooling of setconv_rr has no
    v_mov_b32_dpp left, the narrow kernels only the 16 / 32 of their row_bcast step (partial row mask: not foldable)."""
    path = os.path.join(ROOT, "efficientlo-net_amd", "libelo_hip.so")
    if not os.path.exists(path):
        pytest.skip("library not built")
    seen = 0
    for _triple, blob in H.code_objects(path):
        for name, ins in H.functions(H.disassemble(blob)).items():
            ops = [op for _, op, _, _ in ins]
            if not any(o.startswith("v_mfma") for o in ops) or "weight_grad" in name:
                continue                                     # (the training kernel's 64 accumulator registers may live anywhere)
            seen += 1
            assert not any(o.startswith("v_accvgpr") for o in ops), name
            moves = sum(o.startswith("v_mov_b32_dpp") for o in ops)
            if "setconv_rr_kernel" in name:
                assert moves == 0, (name, moves)
            if "setconv_narrow_kernel" in name:
                assert moves <= 32, (name, moves)
    assert seen > 50


def test_the_experiment_switches_live_in_a_patch_that_still_applies():
    """The product source has no "wrong results" / bisect macros (they are tools/micro/patches/elo_fused_experiments.patch, applied to
    a scratch copy by tools/micro/experiment_source.sh); the patch must keep applying to the current csrc/elo_fused.hip."""
    import re
    import shutil
    import subprocess
    src = open(os.path.join(ROOT, "efficientlo-net_amd", "csrc", "elo_fused.hip")).read()
    for macro in ("ELO_RR_WHATIF_HALF_READS", "ELO_NO_MFMA_SHAPE_GUARD", "ELO_CV1_STOP", "ELO_RR_BARRIER", "ELO_CV1_EARLY_DESCRIPTORS"):
        assert not re.search(r"#\s*if.*%s" % macro, src), macro
    assert "getenv" not in "".join(open(os.path.join(ROOT, "efficientlo-net_amd", "csrc", f)).read()
                                   for f in os.listdir(os.path.join(ROOT, "efficientlo-net_amd", "csrc")))
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) here")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "micro", "experiment_source.sh")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    patched = open(out.stdout.strip()).read()
    assert "ELO_RR_WHATIF_HALF_READS" in patched and "#ifndef ELO_NO_MFMA_SHAPE_GUARD" in patched
    shutil.rmtree(os.path.dirname(os.path.dirname(os.path.dirname(out.stdout.strip()))), ignore_errors=True)


def test_the_rejected_training_forms_live_in_a_patch_that_applies_to_its_base(tmp_path):
    """Round 6's launch-merged training reductions (replica-accumulator atomics, batch norm's backward-apply inside the weight-gradient
    kernel, weight gradients on a side stream) were built, tested on the GPU, MEASURED (profiles/r06_training_launch_merging.txt: 336
    launches fewer, the same 15 ms step) and taken out of csrc/: they are tools/micro/patches/r06_training_launch_merging.patch.  The
    weight-gradient kernel the patch extends was rewritten later in the round, so the patch names the commit it applies to (`# base:`)
    and is checked against THAT tree, read from git; the product must not carry the forms."""
    import re
    import shutil
    import subprocess
    train = open(os.path.join(ROOT, "efficientlo-net_amd", "csrc", "elo_train.hip")).read()
    assert "unsafeAtomicAdd" not in train and "bn_stats_acc_kernel" not in train
    patch = os.path.join(ROOT, "tools", "micro", "patches", "r06_training_launch_merging.patch")
    text = open(patch).read()
    base = re.search(r"^# base: ([0-9a-f]{7,40})", text, re.M).group(1)
    if shutil.which("patch") is None or shutil.which("git") is None:
        pytest.skip("no patch(1) / git here")
    if subprocess.run(["git", "cat-file", "-e", base + "^{commit}"], cwd=ROOT, capture_output=True).returncode != 0:
        pytest.skip("commit %s is not in this checkout's history" % base)
    for rel in re.findall(r"^\+\+\+ b/(\S+)", text, re.M):
        out = subprocess.run(["git", "show", "%s:%s" % (base, rel)], cwd=ROOT, capture_output=True, text=True)
        assert out.returncode == 0, rel
        (tmp_path / rel).parent.mkdir(parents=True, exist_ok=True)
        (tmp_path / rel).write_text(out.stdout)
    out = subprocess.run(["patch", "-p1", "-s", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    patched = (tmp_path / "efficientlo-net_amd" / "csrc" / "elo_train.hip").read_text()
    assert "bn_stats_acc_kernel" in patched and "struct BnFuse" in patched and "elo_dense_bn_backward" in patched



def test_the_rejected_kernel_forms_live_in_a_patch_that_still_applies(tmp_path):
    """Forms that were built, tested, MEASURED slower and therefore taken out of csrc/ in round 6: "layer 0 through the
    gather" (rowlinear_rr_kernel + the pre forms of the three chain kernels: profiles/r05_layer0_through_the_gather.txt), the LDS-staged
    narrow set-conv (setconv_tiled_kernel: profiles/r05_ab_tiled.txt), the pose head that reduces softmax_valid itself (direct), the
    slot-indexed cv_encode1 switch, forked branches, the 6 -> 8 -> 8 -> 16 layer on the matrix cores.  tools/micro/patches/
    r06_rejected_forms.patch re-creates them (kernels, C ABI, host, tests) on the product tree; the product must not carry them."""
    import shutil
    import subprocess
    fused = open(os.path.join(ROOT, "efficientlo-net_amd", "csrc", "elo_fused.hip")).read()
    for gone in ("rowlinear_rr_kernel", "setconv_tiled_kernel", "pre_c", "tiled_setconv"):
        assert gone not in fused, gone
    assert fused.count("\n") < 3700
    header = open(os.path.join(ROOT, "include", "elo.h")).read()
    for gone in ("elo_rowlinear_fused2", "encode1_slots", "tiled_setconv", "ELO_POSE_DIRECT_MAX", "centre_cols"):
        assert gone not in header, gone
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) here")
    for d in ("efficientlo-net_amd", "include", "tests"):
        shutil.copytree(os.path.join(ROOT, d), tmp_path / d, ignore=shutil.ignore_patterns("*.so", "*.o", "build", "__pycache__", "golden"))
    patch = os.path.join(ROOT, "tools", "micro", "patches", "r06_rejected_forms.patch")
    out = subprocess.run(["patch", "-p1", "-s", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    patched = (tmp_path / "efficientlo-net_amd" / "csrc" / "elo_fused.hip").read_text()
    assert "rowlinear_rr_kernel" in patched and "setconv_tiled_kernel" in patched and "bool PRE = false" in patched
