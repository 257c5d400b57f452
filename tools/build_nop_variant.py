#!/usr/bin/env python3
"""Variant libraries whose DEVICE ASSEMBLY is edited between compiler and assembler.

The compiler puts `s_nop 0` (one wait state) between a packed-f32 instruction and a dependent consumer: gfx940+ cannot forward
the result of a packed / op_sel / SDWA-dst_sel instruction to the next instruction (LLVM: hasDstSelForwardingHazard).  Hypothesis
of round 4: beside co-resident MFMA waves ONE wait state is not enough for the last quarter of the wave (lanes 48-63).  This tool
builds sr.hip in the failing form (packed-f32, serialised loads) with `s_nop <N>` inserted after EVERY packed-f32 instruction:

    python tools/build_nop_variant.py          # -> libasr_hz_pk_wait_nop0.so (>= 1 wait state by an explicit nop everywhere)
                                               #    libasr_hz_pk_wait_nop1.so (>= 2 wait states)   libasr_hz_pk_wait_nop3.so (>= 4)
                                               #    libasr_hz_pk_wait_both_nop3.so (>= 4 before AND after)
Pipeline = what `hipcc -v -c` shows: device cc1 -> (edit) -> assembler -> lld -> clang-offload-bundler -> host cc1 with
-fcuda-include-gpubinary.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deeplabv3plus-augmented-superresolution_amd")
CSRC = os.path.join(PKG, "csrc")
sys.path.insert(0, CSRC)
import build as B  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(CSRC, "build_hz")
PK = re.compile(r"^\s+v_pk_(mul|add|fma)_f32\b")


def run(cmd):
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


_OPS = {"v_pk_add_f32": "v_add_f32", "v_pk_mul_f32": "v_mul_f32"}


def _mods(text, name, n):
    m = re.search(name + r":\[([01,]+)\]", text)
    return [int(v) for v in m.group(1).split(",")] if m else None


def expand_src1_high_select(line):
    """A two-source packed-f32 instruction whose LOW result takes the low half of src0 and the HIGH half of a vector src1
    (op_sel:[0,1]: the form that goes wrong beside MFMA, tools/ubench_pk_opsel_erratum.hip) -> its two unpacked halves.
    Returns None for every other line."""
    m = re.match(r"^\s+(v_pk_add_f32|v_pk_mul_f32) v\[(\d+):(\d+)\], ([vs])\[(\d+):(\d+)\], v\[(\d+):(\d+)\](.*)$", line)
    if not m:
        return None
    op, d0, d1, a_file = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
    a0, a1, b0, b1, rest = int(m.group(5)), int(m.group(6)), int(m.group(7)), int(m.group(8)), m.group(9)
    if a_file == "s":                                            # a scalar src0 overlaps nothing: give its halves numbers no VGPR has
        a0, a1 = -1 - a0, -1 - a1
    sel, sel_hi = _mods(rest, "op_sel", 2) or [0, 0], _mods(rest, "op_sel_hi", 2) or [1, 1]
    if sel != [0, 1]:
        return None
    neg_lo, neg_hi = _mods(rest, "neg_lo", 2) or [0, 0], _mods(rest, "neg_hi", 2) or [0, 0]
    lo_src = ((a1 if sel[0] else a0), (b1 if sel[1] else b0))
    hi_src = ((a1 if sel_hi[0] else a0), (b1 if sel_hi[1] else b0))
    def reg(r):
        return f"v{r}" if r >= 0 else f"s{-1 - r}"

    def one(dst, srcs, negs):
        return f"\t{_OPS[op]}_e64 v{dst}, {'-' if negs[0] else ''}{reg(srcs[0])}, {'-' if negs[1] else ''}{reg(srcs[1])}\n"
    if d0 in hi_src:                                            # the high half would read what the low half just wrote
        if d1 not in lo_src:
            return [one(d1, hi_src, neg_hi), one(d0, lo_src, neg_lo)]                 # the other order is safe
        if sorted(lo_src) == sorted(hi_src) and neg_lo == neg_hi == [0, 0]:           # both halves are the same commutative result
            return [one(d0, lo_src, neg_lo), f"\tv_mov_b32_e32 v{d1}, v{d0}\n"]
        if (b0, b1) == (d0, d1) and sel_hi[1] == 0 and d0 not in (a0, a1) and d1 not in (a0, a1):
            # src1 IS the destination and its halves are exchanged: exchange them first, then both halves are in place
            return [f"\tv_swap_b32 v{d0}, v{d1}\n", one(d0, (lo_src[0], d0), neg_lo), one(d1, (hi_src[0], d1), neg_hi)]
        raise RuntimeError("overlap: " + line)
    return [one(d0, lo_src, neg_lo), one(d1, hi_src, neg_hi)]


def build(nop, defines=("-DASR_DIAG_KFWD_WAIT",), tag="pk_wait", before=False, fix_opsel=False):
    os.makedirs(OBJ, exist_ok=True)
    src = os.path.join(CSRC, "sr.hip")
    flags = [f for f in B.COMMON if f not in ("-x", "hip")] + ["-ffp-contract=off"] + list(defines)
    if before:
        tag += "_both"
    stem = os.path.join(OBJ, f"sr_{tag}_nop{nop}")
    if fix_opsel:
        stem = os.path.join(OBJ, f"sr_{tag}_fixsel")
    run([B._hipcc()] + flags + ["-x", "hip", "-S", "--cuda-device-only", src, "-o", stem + ".s"])
    out, n = [], 0
    for line in open(stem + ".s"):
        if fix_opsel:
            two = expand_src1_high_select(line)
            if two:
                out += two
                n += 1
            else:
                out.append(line)
            continue
        if before and PK.match(line):
            out.append(f"\ts_nop {nop}\n")
        out.append(line)
        if PK.match(line):
            out.append(f"\ts_nop {nop}\n")
            n += 1
    with open(stem + ".nop.s", "w") as fh:
        fh.writelines(out)
    run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", stem + ".nop.s", "-o", stem + ".dev.o"])
    run([os.path.join(LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", stem + ".co", stem + ".dev.o"])
    run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + stem + ".co",
         "-output=" + stem + ".hipfb"])
    run([B._hipcc()] + flags + ["-x", "hip", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", stem + ".hipfb", "-c", src,
         "-o", stem + ".o"])
    prod = [os.path.join(CSRC, "build", os.path.splitext(s)[0] + ".o") for s, _ in B.SOURCES]
    objs = [stem + ".o" if os.path.basename(o) == "sr.o" else o for o in prod]
    lib = os.path.join(PKG, f"libasr_hz_{tag}_fixsel.so" if fix_opsel else f"libasr_hz_{tag}_nop{nop}.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(f"{lib}: " + (f"{n} packed-f32 instructions with op_sel:[0,1] rewritten as two unpacked ones (all other packed ops kept)" if fix_opsel
                        else f"s_nop {nop} after {n} packed-f32 instructions"), flush=True)


if __name__ == "__main__":
    B.build(verbose=False)
    for nop in (0, 1, 3):
        build(nop)
    build(3, before=True)          # libasr_hz_pk_wait_both_nop3.so: four wait states BEFORE and after every packed-f32 instruction
    build(0, fix_opsel=True)       # libasr_hz_pk_wait_fixsel.so: ONLY the op_sel:[0,1] instructions unpacked
