#!/bin/bash
# Fifth matrix of the lanes-48-63 localisation: the REAL victim (K_fwd with packed-f32) beside SYNTHETIC aggressors
# that do one thing each (tools/hazard_aggressors.hip), two 200-register waves per SIMD like the fused stem.
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
python3 tools/build_hazard_variants.py > $OUT/build5.log 2>&1 || { tail -5 $OUT/build5.log; exit 1; }
: > $OUT/summary5.txt
run() {
    local name=$1 spec=$2
    ASR_LIB=$PWD/$PKG/libasr_hz_pk.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary5.txt
    return $rc
}
run syn_vfma_200        synthetic:0:200:400000:2 &&
run syn_mfma16_200      synthetic:1:200:200000:2 &&
run syn_mfma32_200      synthetic:2:200:100000:2 &&
run syn_lds_200         synthetic:3:200:30000:2 &&
run syn_dpp_200         synthetic:4:200:200000:2 &&
run syn_permlane_200    synthetic:5:200:200000:2 &&
run syn_global_200      synthetic:6:200:30000:2 &&
run syn_split_200       synthetic:7:200:100000:2 &&
run syn_all_200         synthetic:8:200:20000:2 &&
run syn_all_216         synthetic:8:216:20000:2
echo "matrix5 done rc=$?" | tee -a $OUT/summary5.txt
