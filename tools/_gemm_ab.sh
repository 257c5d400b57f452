mkdir -p gpurun_out/r3e
B="--steps 8 --warmup 2 --no-cpu-baseline --no-f32-line --no-extra-configs"
run() { python tools/ab_forward.py --rounds 5 2>&1 | grep -v amdgpu.ids | grep -A3 "^config" ; }
ASR_BUILD_VARIANT=diag python deeplabv3plus-augmented-superresolution_amd/csrc/build.py > gpurun_out/r3e/b0.log 2>&1
export ASR_LIB=$PWD/deeplabv3plus-augmented-superresolution_amd/libasr_hip_diag.so
echo "== no setprio"; run
for alt in 1 2 4; do
  ASR_EXTRA_HIPFLAGS="-DASR_GEMM_PRIO_ALTERNATE=$alt" ASR_BUILD_VARIANT=diag python deeplabv3plus-augmented-superresolution_amd/csrc/build.py > gpurun_out/r3e/b$alt.log 2>&1
  echo "== alternate every $alt group(s)"; run
done
