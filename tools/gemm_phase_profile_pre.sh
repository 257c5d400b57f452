#!/bin/bash
# Phase profile of the pre-split LDS-DMA GEMM only (diagnostic build, see tools/gemm_phase_profile.sh).
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ASR_EXTRA_HIPFLAGS=-DASR_GEMM_PHASE_PROFILE python deeplabv3plus-augmented-superresolution_amd/csrc/build.py --force > gpurun_out/gemm_phase_build.log 2>&1
python tools/bench_presplit.py > gpurun_out/gemm_phase_pre.log 2>&1
grep "\[phase-pre" gpurun_out/gemm_phase_pre.log | tac | awk '!seen[$1 $2 $3 $4]++' | tac | cut -c1-330
grep "^M=" gpurun_out/gemm_phase_pre.log | sed "s/in-kernel split 128x128 .* TF\/s  pre/pre/"
