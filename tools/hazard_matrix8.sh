#!/bin/bash
# Eighth matrix: bisect of the VICTIM.  K_fwd with packed-f32 and serialised loads (the form that moves most: 4 395 waves),
# with ONE of its three arithmetic stages written in unpacked asm, beside the bare MFMA loop.
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
python3 tools/build_hazard_variants.py > $OUT/build8.log 2>&1 || { tail -5 $OUT/build8.log; exit 1; }
: > $OUT/summary8.txt
run() {
    local name=$1 lib=$2 spec=$3
    ASR_LIB=$PWD/$PKG/libasr_hz_$lib.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [lib $lib, replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary8.txt
    return $rc
}
run map_unpacked        pk_wait_nopk1 synthetic:1:200:200000:2 &&
run bilinear_unpacked   pk_wait_nopk2 synthetic:1:200:200000:2 &&
run translate_unpacked  pk_wait_nopk4 synthetic:1:200:200000:2 &&
run all_three_unpacked  pk_wait_nopk7 synthetic:1:200:200000:2
echo "matrix8 done rc=$?" | tee -a $OUT/summary8.txt
