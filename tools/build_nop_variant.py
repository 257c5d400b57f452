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


import pk_postpass as PP  # noqa: E402  (csrc/pk_postpass.py: the product's rewrite of the op_sel:[0,1] instructions)


def expand_src1_high_select(line):
    return PP.expand(line)


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
