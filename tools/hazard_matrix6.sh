#!/bin/bash
# Sixth matrix: WHERE in the register file the victim sits.  Victim = K_fwd with packed-f32 (96 registers, v0..v89 used);
# aggressor = the bare v_mfma_f32_16x16x32_f16 loop with 128 / 160 / 200 / 208 registers per wave, two waves per SIMD:
#   128 -> victim waves at register bases 256 and 352 (physical registers up to 441)
#   160 -> bases 320 and 416 (up to 505)        200 -> base 400 (up to 489)        208 -> base 416 (up to 505)
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
python3 tools/build_hazard_variants.py > $OUT/build6.log 2>&1 || { tail -5 $OUT/build6.log; exit 1; }
: > $OUT/summary6.txt
run() {
    local name=$1 spec=$2
    ASR_LIB=$PWD/$PKG/libasr_hz_pk.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary6.txt
    return $rc
}
run mfma16_128regs synthetic:1:128:200000:2 &&
run mfma16_160regs synthetic:1:160:200000:2 &&
run mfma16_200regs synthetic:1:200:200000:2 &&
run mfma16_208regs synthetic:1:208:200000:2
echo "matrix6 done rc=$?" | tee -a $OUT/summary6.txt
