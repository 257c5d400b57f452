"""CPU-side guard over the machine code of libasr_hip.so (csrc/isa_guard.py).

MI355X erratum characterised in round 4 (profiles/r04_hazard_matrix.txt, profiles/r04_pk_opsel_erratum_ubench.txt): a packed-f32
instruction whose low result takes the low half of src0 and the HIGH half of a vector-register src1 (VOP3P op_sel = [0,1])
returns wrong values in lanes 48-63 while an MFMA instruction of another wave is in flight on the same SIMD.  The compiler picks
op_sel by itself; these tests disassemble the library that was actually built and fail if a source edit or a compiler update
brings the form in, or packed-f32 of any form into a kernel that did not opt in.
"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deeplabv3plus-augmented-superresolution_amd")


def _guard():
    spec = importlib.util.spec_from_file_location("asr_isa_guard", os.path.join(PKG, "csrc", "isa_guard.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    return g


@pytest.fixture(scope="module")
def guard():
    import __graft_entry__
    __graft_entry__.build()
    g = _guard()
    try:
        g._tool("llvm-objdump")
    except g.ToolMissing as e:                      # the ROCm image has them; a bare box does not
        pytest.skip(str(e))
    return g


def test_the_erratum_form_is_recognised_exactly(guard):
    """The sixteen forms of tools/ubench_pk_opsel_erratum.hip: the eight that returned wrong lanes beside the MFMA loop are the
    ones with op_sel = [0,1] on a vector src1, the eight that never did are not."""
    wrong = ["v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0]",
             "v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
             "v_pk_mul_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0]",
             "v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]",
             "v_pk_fma_f32 v[90:91], v[84:85], v[86:87], v[88:89] op_sel:[0,1,0] op_sel_hi:[1,0,1]",
             "v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]",
             "v_pk_mul_f32 v[90:91], v[84:85], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
             "v_pk_mul_f32 v[10:11], s[14:15], v[8:9] op_sel:[0,1]"]
    fine = ["v_pk_add_f32 v[90:91], v[84:85], v[86:87]",
            "v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]",
            "v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
            "v_pk_add_f32 v[90:91], v[84:85], s[2:3] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]",
            "v_pk_add_f32 v[90:91], v[84:85], v[86:87] op_sel:[1,1] op_sel_hi:[0,0]",
            "v_pk_mul_f32 v[90:91], v[84:85], v[86:87] op_sel:[1,0] op_sel_hi:[0,1]",
            "v_pk_fma_f32 v[90:91], v[84:85], v[86:87], v[88:89] op_sel:[0,0,1] op_sel_hi:[1,1,0]",
            "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0,1]",
            "v_add_f32_e32 v1, v2, v3", "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]"]
    assert all(guard.erratum_form(i) for i in wrong)
    assert not any(guard.erratum_form(i) for i in fine)


def test_product_library_has_no_forbidden_instruction_form(guard):
    bad = guard.violations()
    assert not bad, "\n".join(f"{rule}: {inst}  in {kern}" for kern, inst, rule in bad[:20])


def test_packed_f32_only_in_the_kernels_that_opted_in_and_never_in_the_erratum_form(guard):
    s = guard.summary()
    assert s["packed_f32"] > 1000, s                                      # the depthwise kernels do use them
    assert s["packed_f32_erratum_form"] == 0, s
    for k in s["kernels_with_packed_f32"]:
        assert any(p in k for p in guard.PK_KERNELS), k
    # the other second-lane kernels (warps, reductions) contain none at all; the SR solver's unit keeps packed-f32 and has its
    # op_sel:[0,1] instructions split by csrc/pk_postpass.py (the erratum-form count above covers it)
    second_lane = ("warp_affine", "augment_copies", "opm_", "argmax", "minmax", "threshold", "iou_counts", "class_")
    assert not any(t in k for k in s["kernels_with_packed_f32"] for t in second_lane)
    assert any("sr_forward_residual_kernel" in k for k in s["kernels_with_packed_f32"])


def test_the_postpass_splits_exactly_the_erratum_form(guard):
    """csrc/pk_postpass.py on single lines: the erratum form becomes its two unpacked halves (operand halves and negations as
    VOP3P defines them), everything else is left alone, and a case that needs a temporary register is refused."""
    spec = importlib.util.spec_from_file_location("asr_pk_postpass", os.path.join(PKG, "csrc", "pk_postpass.py"))
    pp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pp)
    assert pp.expand("\tv_pk_mul_f32 v[4:5], v[0:1], v[2:3]\n") is None
    assert pp.expand("\tv_pk_add_f32 v[4:5], v[0:1], v[2:3] op_sel:[1,0] op_sel_hi:[0,1]\n") is None
    assert pp.expand("\tv_pk_add_f32 v[74:75], v[74:75], v[46:47] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n") == [
        "\tv_add_f32_e64 v74, v74, -v47\n", "\tv_add_f32_e64 v75, v75, -v46\n"]
    assert pp.expand("\tv_pk_mul_f32 v[10:11], s[14:15], v[8:9] op_sel:[0,1]\n") == [
        "\tv_mul_f32_e64 v10, s14, v9\n", "\tv_mul_f32_e64 v11, s15, v9\n"]
    # src1 is the destination with its halves exchanged: exchanged first (v_swap_b32), then both halves in place
    assert pp.expand("\tv_pk_add_f32 v[2:3], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]\n") == [
        "\tv_swap_b32 v2, v3\n", "\tv_add_f32_e64 v2, v4, v2\n", "\tv_add_f32_e64 v3, v5, v3\n"]
    assert pp.expand("\tv_pk_add_f32 v[14:15], v[14:15], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n") == [
        "\tv_add_f32_e64 v14, v14, v15\n", "\tv_mov_b32_e32 v15, v14\n"]
    assert pp.expand("\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n") == [
        "\tv_fma_f32 v0, v2, v5, -v6\n", "\tv_fma_f32 v1, v3, v4, -v7\n"]
    with pytest.raises(pp.Unsafe):      # both halves read what the other writes: needs a temporary register
        pp.expand("\tv_pk_mul_f32 v[2:3], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n")
    new, n = pp.rewrite(["\tv_pk_mul_f32 v[4:5], v[0:1], v[2:3]\n", "\tv_pk_mul_f32 v[4:5], v[0:1], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]\n", "\ts_endpgm\n"])
    assert n == 1 and len(new) == 4 and not any(pp.is_erratum_form(l) for l in new)


def test_the_postpass_leaves_no_erratum_form_in_the_solvers_assembly(guard, tmp_path):
    """The whole unit: sr.hip compiled to device assembly with packed-f32 (what build.py's POSTPASS does) holds erratum-form
    instructions; after csrc/pk_postpass.py none, every other line untouched, and the result assembles."""
    import subprocess
    spec = importlib.util.spec_from_file_location("asr_build", os.path.join(PKG, "csrc", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    spec = importlib.util.spec_from_file_location("asr_pk_postpass", os.path.join(PKG, "csrc", "pk_postpass.py"))
    pp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pp)
    asm = str(tmp_path / "sr.s")
    flags = b.COMMON + dict(b.SOURCES)["sr.hip"]
    assert "-packed-fp32-ops" not in flags and "sr.hip" in b.POSTPASS
    subprocess.run([b._hipcc()] + flags + ["-S", "--cuda-device-only", os.path.join(PKG, "csrc", "sr.hip"), "-o", asm], check=True,
                   stderr=subprocess.DEVNULL)
    lines = open(asm).readlines()
    before = sum(pp.is_erratum_form(l) for l in lines)
    new, n = pp.rewrite(lines)
    assert before > 20 and n == before and not any(pp.is_erratum_form(l) for l in new)
    kept = [l for l in lines if not pp.is_erratum_form(l)]
    assert [l for l in new if l in set(kept)][:50] == kept[:50]                 # (the prologue is untouched)
    out = str(tmp_path / "sr.pp.s")
    open(out, "w").writelines(new)
    subprocess.run([os.path.join(pp.LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", out, "-o",
                    str(tmp_path / "sr.o")], check=True, stderr=subprocess.DEVNULL)


def test_mode_register_is_written_only_by_the_kernels_that_split_with_saturating_conversions(guard):
    writers = guard.summary()["mode_writers"]
    assert writers, "the saturating split (asr_common.h) is expected in the fused entry-flow kernels"
    for k in writers:
        assert any(m in k for m in guard.MODE_WRITERS), k


def test_the_rules_catch_a_library_that_breaks_them(guard, tmp_path):
    """The guard must not be vacuous: sr.hip compiled WITH packed-f32 and WITHOUT the post-pass (the round-3 build that went wrong
    on the GPU) holds the erratum form -- in K_fwd itself -- and is rejected."""
    import subprocess
    spec = importlib.util.spec_from_file_location("asr_build", os.path.join(PKG, "csrc", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    obj = str(tmp_path / "sr_pk.o")
    lib = str(tmp_path / "libbad.so")
    cmd = [b._hipcc()] + b.COMMON + ["-ffp-contract=off", "-c", os.path.join(PKG, "csrc", "sr.hip"), "-o", obj]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj], check=True, stderr=subprocess.DEVNULL)
    bad = guard.violations(lib)
    erratum = [(k, i) for k, i, r in bad if r.startswith("ERRATUM")]
    assert len(erratum) > 50
    assert any("sr_forward_residual_kernel" in k for k, _i in erratum)
