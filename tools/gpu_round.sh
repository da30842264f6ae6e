#!/bin/bash
# One gpurun call: GPU tests, variant A/B timings, config-5 timing.  Logs -> gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for v in $(ls tools/variants/ 2>/dev/null | sed 's/libdrm_//; s/\.so//') product; do
  for b in 65536 1048576; do
    echo "=== bench variant $v batch $b"
    lib=tools/variants/libdrm_$v.so; [ $v = product ] && lib=differentiable-robot-model_amd/csrc/libdrm_hip.so
    DRM_HIP_LIBRARY=$lib timeout 300 python bench.py --no-cpu-baseline --steps 400 --warmup 20 --batch $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], 'us/launch', d['roofline']['frac'], 'frac', d['value'], 'evals/s')"
  done
done
echo "=== bench default"; timeout 600 python bench.py 2>&1 | tail -1
} > gpurun_out/round.log 2>&1
tail -40 gpurun_out/round.log
