#!/bin/bash
# Per-phase shader-clock breakdown of the split-f16 GEMM's K loop (run on the GPU box):
# rebuilds the library with -DASR_GEMM_PHASE_PROFILE (never part of the product build), runs the GEMM micro-benchmark
# and leaves the "[phase]" lines in gpurun_out/gemm_phase.log.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ASR_EXTRA_HIPFLAGS=-DASR_GEMM_PHASE_PROFILE python deeplabv3plus-augmented-superresolution_amd/csrc/build.py --force > gpurun_out/gemm_phase_build.log 2>&1
python tools/bench_gemm.py > gpurun_out/gemm_phase.log 2>&1
python tools/bench_presplit.py >> gpurun_out/gemm_phase.log 2>&1
grep "\[phase" gpurun_out/gemm_phase.log | tac | awk '!seen[$1 $2 $3 $4]++' | tac
