for v in "" "-DSF_EXP_NOPROD" "-DSF_EXP_NOCONS" "-DSF_EXP_NOPROD -DSF_EXP_NOCONS" "-DSF_EXP_NOSTORE" "-DSF_EXP_NOPROD -DSF_EXP_NOSTORE"; do
  ASR_EXTRA_HIPFLAGS="$v" python deeplabv3plus-augmented-superresolution_amd/csrc/build.py > /dev/null 2>&1
  echo "== [$v]"; python tools/bench_sepconv.py 2>&1 | grep -v amdgpu.ids | head -2
done
