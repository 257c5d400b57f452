#!/bin/bash
# Phase profile of the pre-split LDS-DMA GEMMs (run on the GPU box).  The diagnostic library (s_memtime stamps per phase
# of the K loop, a device synchronisation per launch) is built NEXT TO the product library, never in its place:
# csrc/build.py with ASR_BUILD_VARIANT=phase -> libasr_hip_phase.so, selected for this run only through ASR_LIB.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
lib=$(ASR_BUILD_VARIANT=phase python deeplabv3plus-augmented-superresolution_amd/csrc/build.py 2> gpurun_out/gemm_phase_build.log | tail -1)
ASR_LIB=$lib python tools/bench_presplit.py > gpurun_out/gemm_phase_pre.log 2>&1
grep "\[phase-" gpurun_out/gemm_phase_pre.log | tac | awk '!seen[$1 $2 $3 $4 $5]++' | tac | cut -c1-420
