"""Static guard over the machine code of libasr_hip.so (gfx950): instruction forms the library must not contain.

    python isa_guard.py [library]        # prints the violations, exit status 1 if there are any

Why (profiles/r04_hazard_matrix.txt): on MI355X a wave that executes packed-f32 instructions (v_pk_mul_f32 /
v_pk_add_f32 / v_pk_fma_f32, with scalar OR vector sources) returned garbage in lanes 48-63 of some of them while it shared
a SIMD with two waves of entry_stem_fused_kernel / sepconv_fused_kernel launched on another stream -- 12 of 12 trials in
every variant in which the victim fits beside those waves (they take 200 / 208 of the SIMD's 512 vector registers each),
0 of 12 when it does not fit (aggressor raised to 216 / 256 registers) and 0 of 12 for the same victim kernel at the same
register allocation without packed-f32 instructions.  The rules:
  (a) a kernel that can share a SIMD with two 200-register waves -- vector-register allocation <= CORESIDENT_MAX_VGPR = 112 --
      contains NO packed-f32 instruction (kernel attribute ASR_NO_PK_F32 of asr_common.h, or the NO_PK_F32 flags of build.py);
  (b) no packed-f32 instruction anywhere takes a scalar-register source (round 3's narrower rule; kept: it costs nothing);
  (c) MODE (s_setreg) is written only by the kernels listed in MODE_WRITERS;
  (d) the fused entry-flow kernels (FUSED_KERNELS) allocate at least 200 registers each, which is what makes 112 the bound of (a).
csrc/build.py runs this check after every link and tests/test_isa_guard.py runs it on the CPU box, so neither a source edit nor
a compiler update can bring the forms back unnoticed.

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

# kernels (demangled-name substrings) that may write the MODE register: they switch the f16 overflow clamp on for their
# saturating split (asr_common.h: asr_enable_f16_saturation)
MODE_WRITERS = ("entry_stem_fused_kernel", "conv3x3_stem_mfma_kernel", "pw_gemm_f16x3_kernel", "sepconv_fused_kernel",
                "dw_stream_full_kernel", "aspp_dw3_phase_kernel")

# the fused entry-flow kernels keep two waves of >= 200 registers on a SIMD: what fits beside them has at most this many
FUSED_KERNELS = ("entry_stem_fused_kernel", "sepconv_fused_kernel")
FUSED_MIN_VGPR = 200                               # ASR_FUSED_MIN_VGPRS of asr_common.h
CORESIDENT_MAX_VGPR = 512 - 2 * FUSED_MIN_VGPR

_PK_F32 = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")
_SGPR_SRC = re.compile(r"(?<![a-z_0-9])(s\[\d+:\d+\]|(s\d+|vcc|exec|ttmp\d+|ttmp\[\d+:\d+\]|m0)\b)")


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
    vgprs = kernel_vgprs(lib_path)
    for kern, insts in disassemble(lib_path).items():
        if kern not in vgprs:
            bad.append((kern, "", "no register count in the code object metadata"))
            continue
        if any(k in kern for k in FUSED_KERNELS) and vgprs[kern] < FUSED_MIN_VGPR:
            bad.append((kern, "", f"fused entry-flow kernel with {vgprs[kern]} registers (< {FUSED_MIN_VGPR}): larger kernels would fit beside it"))
        small = _allocated(vgprs[kern]) <= CORESIDENT_MAX_VGPR
        for inst in insts:
            if _PK_F32.search(inst):
                if small:
                    bad.append((kern, inst, f"packed-f32 in a kernel of {vgprs[kern]} registers (fits beside the fused entry-flow kernels)"))
                elif _SGPR_SRC.search(_sources(inst)):
                    bad.append((kern, inst, "packed-f32 with a scalar-register source"))
            elif inst.startswith("s_setreg") and not any(k in kern for k in MODE_WRITERS):
                bad.append((kern, inst, "MODE write outside the listed kernels"))
    return bad


def summary(lib_path=None):
    """Counts the guard's rules are about, per library: packed-f32 ops (all / with a scalar source), MODE writers."""
    lib_path = lib_path or os.path.join(PKG, "libasr_hip.so")
    pk = pk_s = 0
    writers, holders = set(), {}
    vgprs = kernel_vgprs(lib_path)
    for kern, insts in disassemble(lib_path).items():
        for inst in insts:
            if _PK_F32.search(inst):
                pk += 1
                pk_s += bool(_SGPR_SRC.search(_sources(inst)))
                holders[kern] = vgprs.get(kern)
            if inst.startswith("s_setreg"):
                writers.add(kern)
    return {"packed_f32": pk, "packed_f32_scalar_source": pk_s, "kernels": len(vgprs),
            "kernels_with_packed_f32": len(holders),
            "fewest_registers_of_a_kernel_with_packed_f32": min((v for v in holders.values() if v is not None), default=None),
            "mode_writers": sorted(writers)}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else None
    v = violations(lib)
    for kern, inst, rule in v[:40]:
        print(f"{rule}: {inst}    in {kern}")
    print(summary(lib))
    print(f"{len(v)} violation(s)")
    sys.exit(1 if v else 0)
