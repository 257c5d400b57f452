set -x
mkdir -p gpurun_out/r3b
B="--no-cpu-baseline --no-f32-line --no-extra-configs"
python bench.py --config 4 --steps 4 --warmup 1 $B > gpurun_out/r3b/c4_base.log 2>&1
python bench.py --config 1 --steps 10 --warmup 2 $B > gpurun_out/r3b/c1_base.log 2>&1
python bench.py --config 2 --steps 6 --warmup 2 $B > gpurun_out/r3b/c2_base.log 2>&1
ASR_EXTRA_HIPFLAGS="-DASR_DW_SMALL_MAX=64" python deeplabv3plus-augmented-superresolution_amd/csrc/build.py > gpurun_out/r3b/build64.log 2>&1
python bench.py --config 4 --steps 4 --warmup 1 $B > gpurun_out/r3b/c4_s64.log 2>&1
python bench.py --config 1 --steps 10 --warmup 2 $B > gpurun_out/r3b/c1_s64.log 2>&1
for f in c4_base c4_s64 c1_base c1_s64 c2_base; do python - $f <<'PY'
import json,sys
d=json.loads([l for l in open(f"gpurun_out/r3b/{sys.argv[1]}.log") if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["kernel_time_ms_per_step"], d["roofline_depthwise"]["frac"], d["roofline"]["frac"])
PY
done
