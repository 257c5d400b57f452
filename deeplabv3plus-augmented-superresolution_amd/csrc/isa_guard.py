"""Static guard over the machine code of libasr_hip.so (gfx950): instruction forms the library must not contain.

    python isa_guard.py [library]        # prints the violations, exit status 1 if there are any

Why (profiles/r04_hazard_matrix.txt, tools/ubench_pk_opsel_erratum.hip).  On MI355X a packed-f32 instruction
(v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) whose LOW result takes the low half of src0 and the HIGH half of a vector-register
src1 -- VOP3P op_sel = [0,1(,x)], whatever op_sel_hi and the negate modifiers -- returns a wrong result in lanes 48-63 while an
MFMA instruction of ANOTHER wave is in flight on the same SIMD.  Found in round 3 as SR solves going wrong next to forward
passes, characterised in round 4: a 16-form microbenchmark beside a bare MFMA loop (the eight forms with op_sel = [0,1] wrong in
~10 % of the co-resident waves, the other eight -- plain, src0 swapped, both swapped, src1 low half for both, src2 swapped, an
SGPR src1 -- never), and confirmed on the real kernel: rewriting ONLY those instructions (107 of sr.hip's 2 011 packed ops) in the
device assembly makes the solver immune, 0 of 12 where the unmodified build moves 12 of 12.  The rules:
  (1) ERRATUM  no packed-f32 instruction with op_sel[0] = 0, op_sel[1] = 1 and a vector-register src1, in ANY kernel;
  (2) POLICY   packed-f32 instructions only in the kernels that opted in (ASR_PK_F32 in asr_common.h, or a unit of build.py's
               POSTPASS; PK_KERNELS below): the compiler chooses op_sel by itself, so the set of kernels in which it may do so
               stays small and reviewed.  In sr.hip it DID emit the form: that unit goes through csrc/pk_postpass.py, which
               splits exactly those instructions in the device assembly;
  (3) MODE (s_setreg) is written only by the kernels listed in MODE_WRITERS.
csrc/build.py runs this check after every link and tests/test_isa_guard.py runs it on the CPU box, so neither a source edit nor
a compiler update can bring the form back unnoticed.

The device code of a hipcc-linked shared object sits in its .hip_fatbin section as one clang offload bundle per translation
unit; the gfx950 code objects are cut out of it here and disassembled with llvm-objdump.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_LLVM_BIN = [os.environ.get("ASR_LLVM_BIN", ""), "/opt/rocm/lib/llvm/bin", "/opt/rocm/llvm/bin"]

# kernels (name substrings) that may write the MODE register: they switch the f16 overflow clamp on for their saturating split
# (asr_common.h: asr_enable_f16_saturation)
MODE_WRITERS = ("entry_stem_fused_kernel", "conv3x3_stem_mfma_kernel", "pw_gemm_f16x3_kernel", "sepconv_fused_kernel",
                "dw_stream_full_kernel", "aspp_dw3_phase_kernel")
# kernels that carry ASR_PK_F32 (packed-f32 opted in): where it pays -- the streaming depthwise kernels, the ring GEMMs' epilogues, the
# fused entry-flow kernels, the direct conv stems
PK_KERNELS = ("dw_stream_full_kernel", "pw_gemm_f16x3_pre_ring", "sepconv_fused_kernel", "entry_stem_fused_kernel",
              "conv3x3_stem_kernel", "conv3x3_stem_mfma_kernel",
              "sr_")      # the SR solver's unit: packed-f32 on, its op_sel:[0,1] instructions split by csrc/pk_postpass.py

_PK_F32 = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")
_SGPR_SRC = re.compile(r"(?<![a-z_0-9])(s\[\d+:\d+\]|(s\d+|vcc|exec|ttmp\d+|ttmp\[\d+:\d+\]|m0)\b)")
_OP_SEL = re.compile(r"\bop_sel:\[([01,]+)\]")


def erratum_form(inst):
    """True for a packed-f32 instruction whose low result reads src0's LOW half and the HIGH half of a VECTOR-register src1
    (op_sel[0] = 0, op_sel[1] = 1): the form that returns wrong lanes 48-63 beside a co-resident MFMA (module docstring)."""
    if not _PK_F32.search(inst):
        return False
    m = _OP_SEL.search(inst)
    if not m:
        return False
    sel = [int(v) for v in m.group(1).split(",")]
    if len(sel) < 2 or sel[0] != 0 or sel[1] != 1:
        return False
    ops = inst.split(None, 1)[1].split(",")            # dst, src0, src1, ...
    return len(ops) >= 3 and ops[2].strip().startswith("v[")


class ToolMissing(RuntimeError):
    """llvm-objcopy / llvm-objdump / llvm-readelf of the ROCm LLVM are not installed where this runs."""


def _tool(name):
    for d in _LLVM_BIN:
        p = os.path.join(d, name)
        if d and os.path.exists(p):
            return p
    raise ToolMissing(f"{name} not found (set ASR_LLVM_BIN)")


def code_objects(lib_path, arch="gfx950"):
    """The device code objects (ELF images, bytes) for `arch` embedded in a hipcc-linked shared object."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([_tool("llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(tmp, "x")])
        data = open(fat, "rb").read()
    out, i = [], data.find(_MAGIC)
    while i >= 0:
        (n,) = struct.unpack_from("<Q", data, i + len(_MAGIC))
        p = i + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tlen].decode()
            p += tlen
            if arch in triple and size:
                out.append(data[i + off:i + off + size])
        i = data.find(_MAGIC, i + len(_MAGIC))
    if not out:
        raise RuntimeError(f"no {arch} code object in {lib_path}")
    return out


def disassemble(lib_path, arch="gfx950"):
    """{kernel symbol: [instruction text, ...]} over every code object of the library."""
    kernels = {}
    objdump = _tool("llvm-objdump")
    for blob in code_objects(lib_path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as fh:
            fh.write(blob)
            fh.flush()
            text = subprocess.run([objdump, "-d", "--no-show-raw-insn", f"--mcpu={arch}", fh.name], check=True,
                                  stdout=subprocess.PIPE, text=True).stdout
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = kernels.setdefault(m.group(1), [])
                continue
            line = line.strip()
            if cur is not None and line and not line.startswith(("Disassembly", "/")):
                cur.append(line.split("//")[0].strip())
    return kernels


def kernel_vgprs(lib_path, arch="gfx950"):
    """{kernel symbol: vector registers per wave (.vgpr_count of the code object metadata, before the granule of 8)}."""
    out = {}
    readelf = _tool("llvm-readelf")
    for blob in code_objects(lib_path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as fh:
            fh.write(blob)
            fh.flush()
            text = subprocess.run([readelf, "--notes", fh.name], check=True, stdout=subprocess.PIPE, text=True).stdout
        name = None
        for line in text.splitlines():
            m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "name":
                name = m.group(2)
            elif m.group(1) == "vgpr_count" and name:
                out[name] = int(m.group(2))
                name = None
    return out


def _allocated(vgprs):
    return -(-vgprs // 8) * 8


def _sources(inst):
    """Source operand text of a VOP3P instruction line (everything after the destination)."""
    ops = inst.split(None, 1)[1] if " " in inst else ""
    return ops.split(",", 1)[1] if "," in ops else ""


def violations(lib_path=None):
    """[(kernel, instruction, rule)] for the library (default: the product library next to this package)."""
    lib_path = lib_path or os.path.join(PKG, "libasr_hip.so")
    bad = []
    for kern, insts in disassemble(lib_path).items():
        opted_in = any(k in kern for k in PK_KERNELS)
        for inst in insts:
            if erratum_form(inst):
                bad.append((kern, inst, "ERRATUM: packed-f32 with op_sel = [0,1] on a vector src1 (wrong lanes 48-63 beside MFMA)"))
            elif _PK_F32.search(inst) and not opted_in:
                bad.append((kern, inst, "POLICY: packed-f32 in a kernel that did not opt in (ASR_PK_F32 / PK_KERNELS)"))
            elif inst.startswith("s_setreg") and not any(k in kern for k in MODE_WRITERS):
                bad.append((kern, inst, "MODE write outside the listed kernels"))
    return bad


def summary(lib_path=None):
    """Counts the guard's rules are about, per library."""
    lib_path = lib_path or os.path.join(PKG, "libasr_hip.so")
    pk = err = 0
    writers, holders, forms = set(), set(), {}
    for kern, insts in disassemble(lib_path).items():
        for inst in insts:
            if _PK_F32.search(inst):
                pk += 1
                err += erratum_form(inst)
                holders.add(kern)
                key = " ".join(re.findall(r"(?:op_sel|op_sel_hi|neg_lo|neg_hi):\[[01,]+\]", inst)) or "(no modifiers)"
                forms[key] = forms.get(key, 0) + 1
            if inst.startswith("s_setreg"):
                writers.add(kern)
    return {"packed_f32": pk, "packed_f32_erratum_form": err, "packed_f32_forms": forms, "kernels_with_packed_f32": sorted(holders),
            "mode_writers": sorted(writers)}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else None
    v = violations(lib)
    for kern, inst, rule in v[:40]:
        print(f"{rule}: {inst}    in {kern}")
    print(summary(lib))
    print(f"{len(v)} violation(s)")
    sys.exit(1 if v else 0)
