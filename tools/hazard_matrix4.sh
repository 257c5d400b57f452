#!/bin/bash
# Fourth matrix of the lanes-48-63 localisation: which part of the fused stem makes it an aggressor.
# Builds the variant libraries on the box (they are not shipped), then one diag run per library.
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
python3 tools/build_hazard_variants.py > $OUT/build4.log 2>&1 || { tail -5 $OUT/build4.log; exit 1; }
: > $OUT/summary4.txt
run() {
    local name=$1 lib=$2 spec=$3
    ASR_LIB=$PWD/$PKG/libasr_hz_$lib.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [lib $lib, replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary4.txt
    return $rc
}
run stem_no_image_loads      pk_stem_skip1   conv:6 &&
run stem_one_tap_stage2      pk_stem_skip2   conv:8 &&
run stem_no_stores           pk_stem_skip4   conv:6 &&
run stem_no_stage1           pk_stem_skip8   conv:8 &&
run stem_stage2_only_nostore pk_stem_skip12  conv:8 &&
run stem_stage1_only         pk_stem_skip6   conv:10
echo "matrix4 done rc=$?" | tee -a $OUT/summary4.txt
