#!/bin/bash
# Per-phase shader-clock breakdown of the split-f16 GEMMs' K loops (run on the GPU box).  The diagnostic library is built
# next to the product library (ASR_BUILD_VARIANT=phase -> libasr_hip_phase.so) and selected through ASR_LIB for this run.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
lib=$(ASR_BUILD_VARIANT=phase python deeplabv3plus-augmented-superresolution_amd/csrc/build.py 2> gpurun_out/gemm_phase_build.log | tail -1)
ASR_LIB=$lib python tools/bench_gemm.py > gpurun_out/gemm_phase.log 2>&1
ASR_LIB=$lib python tools/bench_presplit.py >> gpurun_out/gemm_phase.log 2>&1
grep "\[phase" gpurun_out/gemm_phase.log | tac | awk '!seen[$1 $2 $3 $4]++' | tac
