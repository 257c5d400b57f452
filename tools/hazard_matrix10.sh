#!/bin/bash
# Tenth matrix: confirmation on the real kernel.  K_fwd with packed-f32 and serialised loads (control: 4 400 - 9 600 waves
# hit) with ONLY the packed-f32 instructions of the form op_sel:[0,1,..] (low result from the HIGH half of a vector src1) rewritten in
# the device assembly as two unpacked instructions -- 107 of sr.hip's 2 011 packed ops; the other 1 904 stay packed
# (tools/build_nop_variant.py) -- beside the bare MFMA loop and beside the real fused stem.
set -u
T=${1:-12}
OUT=gpurun_out/hz
mkdir -p $OUT
PKG=deeplabv3plus-augmented-superresolution_amd
python3 tools/build_hazard_variants.py > $OUT/build10.log 2>&1 || { tail -5 $OUT/build10.log; exit 1; }
python3 tools/build_nop_variant.py >> $OUT/build10.log 2>&1 || { tail -5 $OUT/build10.log; exit 1; }
: > $OUT/summary10.txt
run() {
    local name=$1 lib=$2 spec=$3
    ASR_LIB=$PWD/$PKG/libasr_hz_$lib.so DIAG_REPLAY=$spec timeout -k 10 300 python3 tools/diag_sr_stages_under_stem.py $T > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name [lib $lib, replay $spec] rc=$rc: $(grep -E 'one-iteration solves moved' $OUT/$name.log | tail -1) | $(grep -E 'waves hit' $OUT/$name.log | head -1)" | tee -a $OUT/summary10.txt
    return $rc
}
run control_beside_mfma_loop   pk_wait        synthetic:1:200:200000:2 &&
run fixsel_beside_mfma_loop    pk_wait_fixsel synthetic:1:200:200000:2 &&
run fixsel_beside_mfma_128regs pk_wait_fixsel synthetic:1:128:200000:2 &&
run control_beside_stem        pk_wait        conv:6 &&
run fixsel_beside_stem         pk_wait_fixsel conv:6
echo "matrix10 done rc=$?" | tee -a $OUT/summary10.txt
