#!/bin/bash
# Third matrix of the lanes-48-63 localisation: one benign aggressor, its register allocation raised step by step.
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
: > $OUT/summary3.txt
run() {
    local name=$1 lib=$2 spec=$3
    ASR_LIB=$PWD/$PKG/libasr_hz_$lib.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [lib $lib, replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary3.txt
    return $rc
}
run sepconv64_clamped_192_regs pk_sepconv192 name:asr_sepconv_fused_f16x3:3 &&
run sepconv64_clamped_200_regs pk_sepconv200 name:asr_sepconv_fused_f16x3:3 &&
run sepconv64_clamped_208_regs pk_sepconv208 name:asr_sepconv_fused_f16x3:3
echo "matrix3 done rc=$?" | tee -a $OUT/summary3.txt
