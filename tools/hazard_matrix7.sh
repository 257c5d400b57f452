#!/bin/bash
# Seventh matrix: the victim with its own loads serialised (each bilinear sample waits for its two loads before any
# arithmetic: no load of the wave is in flight behind its packed ops), beside the bare MFMA loop.
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
python3 tools/build_hazard_variants.py > $OUT/build7.log 2>&1 || { tail -5 $OUT/build7.log; exit 1; }
: > $OUT/summary7.txt
run() {
    local name=$1 lib=$2 spec=$3
    ASR_LIB=$PWD/$PKG/libasr_hz_$lib.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [lib $lib, replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary7.txt
    return $rc
}
run kfwd_pk_serialised_loads pk_wait synthetic:1:200:200000:2 &&
run kfwd_pk_selfcheck        pk_check synthetic:1:200:200000:2 &&
run kfwd_pk_control          pk      synthetic:1:200:200000:2
echo "matrix7 done rc=$?" | tee -a $OUT/summary7.txt
