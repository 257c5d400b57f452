#!/usr/bin/env python3
"""Variant libraries for the localisation of the lanes-48-63 interaction (round 4).

The victim (K_fwd of the SR solver with packed-f32 instructions that take SGPR operands) and the aggressor (a kernel that
runs the saturating split of asr_common.h on another stream) are varied ONE ingredient at a time; every variant library is
the product's objects (csrc/build/*.o) with one or two translation units replaced.  tools/diag_sr_stages_under_stem.py is
then run once per library (ASR_LIB=...), 12 one-iteration solves each: "N of 12 moved".

    python tools/build_hazard_variants.py          # cross-compiles here (no GPU needed); libraries land next to libasr_hip.so
                                                   # as libasr_hz_<name>.so and travel to the GPU box with the snapshot

victim side (sr.hip; -ffp-contract=off always):
    pk       packed-f32 allowed, transform coefficients in SGPRs (the round-3 form that moved 44 of 44 times)
    pkv      packed-f32 allowed, coefficients pinned in VGPRs (-DASR_TF_IN_VGPR): K_fwd has no SGPR-operand packed op left
    (product) no packed-f32 at all
aggressor side (layers.hip = entry_stem_fused_kernel, sepconv.hip, gemm.hip):
    nosetreg    -DASR_DIAG_NO_SETREG      MODE.FP16_OVFL is never written; packed converts stay
    scalar      -DASR_DIAG_SCALAR_SPLIT   MODE.FP16_OVFL written, value-by-value clamped split (no v_cvt_pk / SDWA / pk_add from it)
    nosdwa      -mllvm -amdgpu-sdwa-peephole=0   packed split, MODE written, no SDWA forms
    sepconv_pk  -DASR_SEPCONV_PACKED_SPLIT=1      the fused separable conv on the packed split (round 3: "joins the stem")
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deeplabv3plus-augmented-superresolution_amd")
CSRC = os.path.join(PKG, "csrc")
sys.path.insert(0, CSRC)
import build as B  # noqa: E402

OBJ = os.path.join(CSRC, "build_hz")
FP = ["-ffp-contract=off"]
NOSDWA = ["-mllvm", "-amdgpu-sdwa-peephole=0"]
# object name -> (source, flags)
OBJECTS = {
    "sr_pk": ("sr.hip", FP),
    "sr_pkv": ("sr.hip", FP + ["-DASR_TF_IN_VGPR"]),
    "layers_nosetreg": ("layers.hip", ["-DASR_DIAG_NO_SETREG"]),
    "layers_scalar": ("layers.hip", ["-DASR_DIAG_SCALAR_SPLIT"]),
    "layers_nosdwa": ("layers.hip", NOSDWA),
    "layers_scalar_nosetreg": ("layers.hip", ["-DASR_DIAG_SCALAR_SPLIT", "-DASR_DIAG_NO_SETREG"]),
    "sepconv_pk": ("sepconv.hip", ["-DASR_SEPCONV_PACKED_SPLIT=1"]),
    "sepconv_pk_nosetreg": ("sepconv.hip", ["-DASR_SEPCONV_PACKED_SPLIT=1", "-DASR_DIAG_NO_SETREG"]),
    "sepconv_clamped_setreg": ("sepconv.hip", ["-DASR_SEPCONV_PACKED_SPLIT=1", "-DASR_DIAG_SCALAR_SPLIT"]),
    # second matrix: how many registers the waves allocate, i.e. who can share a SIMD's 512 with whom (nothing else changes)
    "layers_v256": ("layers.hip", ["-DASR_DIAG_STEM_TOP_VGPR=255"]),           # two stem waves fill the file: no co-resident wave
    "layers_v216": ("layers.hip", ["-DASR_DIAG_STEM_TOP_VGPR=215"]),           # 80 left: K_fwd (96 with packed-f32) does not fit
    "layers_v208": ("layers.hip", ["-DASR_DIAG_STEM_TOP_VGPR=207"]),           # 96 left: it just fits, at base 416
    "sepconv_pk_v216": ("sepconv.hip", ["-DASR_SEPCONV_PACKED_SPLIT=1", "-DASR_DIAG_SEPCONV_TOP_VGPR=215"]),
    # third matrix: the SAME benign aggressor (clamped sepconv<64>, 182 registers: K_fwd beside it at base 368 never moved) with
    # its allocation raised, so that the co-resident K_fwd wave lands at base 384 / 400 / 416 (sepconv<128> has 212: hosts nothing)
    "sepconv_v192": ("sepconv.hip", ["-DASR_DIAG_SEPCONV_TOP_VGPR=191"]),
    "sepconv_v200": ("sepconv.hip", ["-DASR_DIAG_SEPCONV_TOP_VGPR=199"]),
    "sepconv_v208": ("sepconv.hip", ["-DASR_DIAG_SEPCONV_TOP_VGPR=207"]),
    # A/B objects (tools/ab_entry_points.py): the depthwise kernels on the clamped value-by-value split of rounds 1 - 3
    "dwconv_clamped": ("dwconv.hip", ["-DASR_DW_PACKED_SPLIT=0"]),
    # ablations of the fused stem (tools/bench_stem_ablation.py): what its 19 K cycles per tile are made of
    "layers_skip1": ("layers.hip", ["-DASR_DIAG_STEM_SKIP=1"]),      # no image loads
    "layers_skip2": ("layers.hip", ["-DASR_DIAG_STEM_SKIP=2"]),      # stage 2 with one tap instead of nine
    "layers_skip4": ("layers.hip", ["-DASR_DIAG_STEM_SKIP=4"]),      # no output stores
    "layers_skip8": ("layers.hip", ["-DASR_DIAG_STEM_SKIP=8"]),      # no stage 1
    "layers_skip12": ("layers.hip", ["-DASR_DIAG_STEM_SKIP=12"]),    # stage 2 alone, no stores
    "layers_skip6": ("layers.hip", ["-DASR_DIAG_STEM_SKIP=6"]),      # stage 1 alone
    # ablations of the streaming depthwise kernel (tools/bench_step_variants.py)
    "dwconv_skip1": ("dwconv.hip", ["-DASR_DIAG_DW=1"]),      # one tap of nine
    "dwconv_skip2": ("dwconv.hip", ["-DASR_DIAG_DW=2"]),      # f32 store instead of split + lane trade
    "dwconv_skip4": ("dwconv.hip", ["-DASR_DIAG_DW=4"]),      # one load per row instead of three
    "dwconv_skip7": ("dwconv.hip", ["-DASR_DIAG_DW=7"]),      # all three: a copy with the kernel's skeleton
    # K_fwd with packed-f32 AND a stage-by-stage self-check against unpacked arithmetic (tools/diag_kfwd_selfcheck.py)
    "sr_pk_check": ("sr.hip", FP + ["-DASR_DIAG_KFWD_CHECK"]),
    "sr_pk_wait": ("sr.hip", FP + ["-DASR_DIAG_KFWD_WAIT"]),
    # ... and ONE stage of K_fwd written in unpacked asm (bisect of the victim): 1 coordinate map, 2 bilinear sample, 4 translate blend
    "sr_pk_wait_nopk1": ("sr.hip", FP + ["-DASR_DIAG_KFWD_WAIT", "-DASR_DIAG_KFWD_NOPK=1"]),
    "sr_pk_wait_nopk2": ("sr.hip", FP + ["-DASR_DIAG_KFWD_WAIT", "-DASR_DIAG_KFWD_NOPK=2"]),
    "sr_pk_wait_nopk4": ("sr.hip", FP + ["-DASR_DIAG_KFWD_WAIT", "-DASR_DIAG_KFWD_NOPK=4"]),
    "sr_pk_wait_nopk7": ("sr.hip", FP + ["-DASR_DIAG_KFWD_WAIT", "-DASR_DIAG_KFWD_NOPK=7"]),          # packed-f32, but no load of the wave in flight behind its packed ops
    "gemm_nowalk": ("gemm.hip", ["-DASR_PERSISTENT_WALK=0"]),          # A/B: every pre-split GEMM launch on the one-tile kernel (round 3's product)
    "sepconv_clamped": ("sepconv.hip", ["-DASR_SEPCONV_PACKED_SPLIT=0"]),
    "dwconv_nont": ("dwconv.hip", ["-DASR_DW_NT=0"]),                 # A/B: ordinary (cached) output stores in the depthwise kernels
    "sr_pk_v112": ("sr.hip", FP + ["-DASR_DIAG_KFWD_TOP_VGPR=111"]),
    "sr_nopk_v96": ("sr.hip", FP + B.NO_PK_F32 + ["-DASR_DIAG_KFWD_TOP_VGPR=95"]),     # the product's K_fwd, allocation raised from 64 to 96
}
# library name -> {product object replaced: variant object}
LIBS = {
    "pk": {"sr": "sr_pk"},                                                     # positive control: expect 12 of 12
    "pk_nosetreg": {"sr": "sr_pk", "layers": "layers_nosetreg"},
    "pk_scalar": {"sr": "sr_pk", "layers": "layers_scalar"},
    "pk_nosdwa": {"sr": "sr_pk", "layers": "layers_nosdwa"},
    "pk_scalar_nosetreg": {"sr": "sr_pk", "layers": "layers_scalar_nosetreg"},  # negative control for the stem: expect 0
    "pkv": {"sr": "sr_pkv"},                                                   # packed math on vector operands only
    "pk_sepconv_pk": {"sr": "sr_pk", "sepconv": "sepconv_pk"},                 # replay the sepconv launches, not the stem
    "pk_sepconv_pk_nosetreg": {"sr": "sr_pk", "sepconv": "sepconv_pk_nosetreg"},
    "pk_sepconv_clamped_setreg": {"sr": "sr_pk", "sepconv": "sepconv_clamped_setreg"},
    "pk_stem256": {"sr": "sr_pk", "layers": "layers_v256"},
    "pk_stem216": {"sr": "sr_pk", "layers": "layers_v216"},
    "pk_stem208": {"sr": "sr_pk", "layers": "layers_v208"},
    "pk_sepconv_pk_v216": {"sr": "sr_pk", "sepconv": "sepconv_pk_v216"},
    "pk112": {"sr": "sr_pk_v112"},
    "dwclamped": {"dwconv": "dwconv_clamped"},
    "dw_nont": {"dwconv": "dwconv_nont"},
    "nowalk": {"gemm": "gemm_nowalk"}, "sepconv_clamped": {"sepconv": "sepconv_clamped"},
    "pk_check": {"sr": "sr_pk_check"},
    "pk_wait": {"sr": "sr_pk_wait"},
    "pk_wait_nopk1": {"sr": "sr_pk_wait_nopk1"}, "pk_wait_nopk2": {"sr": "sr_pk_wait_nopk2"}, "pk_wait_nopk4": {"sr": "sr_pk_wait_nopk4"},
    "pk_wait_nopk7": {"sr": "sr_pk_wait_nopk7"},
    "dw_skip1": {"dwconv": "dwconv_skip1"}, "dw_skip2": {"dwconv": "dwconv_skip2"}, "dw_skip4": {"dwconv": "dwconv_skip4"},
    "dw_skip7": {"dwconv": "dwconv_skip7"},
    # fourth matrix: which PART of the stem makes it an aggressor (victim = K_fwd with packed-f32)
    "pk_stem_skip1": {"sr": "sr_pk", "layers": "layers_skip1"}, "pk_stem_skip2": {"sr": "sr_pk", "layers": "layers_skip2"},
    "pk_stem_skip4": {"sr": "sr_pk", "layers": "layers_skip4"}, "pk_stem_skip8": {"sr": "sr_pk", "layers": "layers_skip8"},
    "pk_stem_skip12": {"sr": "sr_pk", "layers": "layers_skip12"}, "pk_stem_skip6": {"sr": "sr_pk", "layers": "layers_skip6"},
    "stem_skip1": {"layers": "layers_skip1"}, "stem_skip2": {"layers": "layers_skip2"}, "stem_skip4": {"layers": "layers_skip4"},
    "stem_skip8": {"layers": "layers_skip8"}, "stem_skip12": {"layers": "layers_skip12"}, "stem_skip6": {"layers": "layers_skip6"},
    "pk_sepconv192": {"sr": "sr_pk", "sepconv": "sepconv_v192"},
    "pk_sepconv200": {"sr": "sr_pk", "sepconv": "sepconv_v200"},
    "pk_sepconv208": {"sr": "sr_pk", "sepconv": "sepconv_v208"},
    "nopk96": {"sr": "sr_nopk_v96"},
}


def main():
    B.build(verbose=False)                                                     # product objects up to date
    os.makedirs(OBJ, exist_ok=True)
    hipcc = B._hipcc()
    for name, (src, flags) in OBJECTS.items():
        o = os.path.join(OBJ, name + ".o")
        s = os.path.join(CSRC, src)
        deps = [s] + [os.path.join(CSRC, h) for h in B.HEADERS] + [os.path.abspath(__file__)]
        if B._stale(o, deps):
            extra = dict(B.SOURCES)[src]
            if src == "sr.hip":        # the sr variants state their own flags (with or without packed-f32)
                extra = [f for f in extra if f not in B.NO_PK_F32 and f not in FP]
            cmd = [hipcc] + B.COMMON + extra + flags + ["-c", s, "-o", o]
            print(" ".join(cmd), flush=True)
            B._compile(cmd)
    prod = {os.path.splitext(src)[0]: os.path.join(CSRC, "build", os.path.splitext(src)[0] + ".o") for src, _ in B.SOURCES}
    for lib, repl in LIBS.items():
        objs = [os.path.join(OBJ, repl[k] + ".o") if k in repl else o for k, o in prod.items()]
        out = os.path.join(PKG, f"libasr_hz_{lib}.so")
        if B._stale(out, objs):
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
        print(out, flush=True)


if __name__ == "__main__":
    main()
