#!/bin/bash
# Ninth matrix: wait states behind packed-f32 instructions.  The failing form of K_fwd (packed-f32, serialised loads:
# 4 395 waves hit) with `s_nop N` inserted after EVERY packed-f32 instruction in the device assembly (tools/build_nop_variant.py),
# beside the bare MFMA loop.  N = 0: at least one wait state by an explicit nop (what the compiler guarantees through nops OR
# independent instructions); N = 1: at least two; N = 3: at least four.
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
python3 tools/build_hazard_variants.py > $OUT/build9.log 2>&1 || { tail -5 $OUT/build9.log; exit 1; }
python3 tools/build_nop_variant.py >> $OUT/build9.log 2>&1 || { tail -5 $OUT/build9.log; exit 1; }
: > $OUT/summary9.txt
run() {
    local name=$1 lib=$2 spec=$3
    ASR_LIB=$PWD/$PKG/libasr_hz_$lib.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [lib $lib, replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary9.txt
    return $rc
}
run control_no_extra_nops  pk_wait      synthetic:1:200:200000:2 &&
run nop0_after_every_pk    pk_wait_nop0 synthetic:1:200:200000:2 &&
run nop1_after_every_pk    pk_wait_nop1 synthetic:1:200:200000:2 &&
run nop3_after_every_pk    pk_wait_nop3 synthetic:1:200:200000:2 &&
run nop3_before_and_after  pk_wait_both_nop3 synthetic:1:200:200000:2
echo "matrix9 done rc=$?" | tee -a $OUT/summary9.txt
