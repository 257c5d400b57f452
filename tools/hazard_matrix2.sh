#!/bin/bash
# Second matrix of the lanes-48-63 localisation: register allocations only.  See tools/hazard_matrix.sh.
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
: > $OUT/summary2.txt
run() {
    local name=$1 lib=$2 spec=$3
    ASR_LIB=$PWD/$PKG/libasr_hz_$lib.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [lib $lib, replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary2.txt
    return $rc
}
run stem_256_regs        pk_stem256          conv:6 &&
run stem_216_regs        pk_stem216          conv:6 &&
run stem_208_regs        pk_stem208          conv:6 &&
run sepconv_packed_216   pk_sepconv_pk_v216  name:asr_sepconv_fused_f16x3:3 &&
run kfwd_packed_112_regs pk112               conv:6 &&
run kfwd_unpacked_96_regs nopk96             conv:6
echo "matrix2 done rc=$?" | tee -a $OUT/summary2.txt
