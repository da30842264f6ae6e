#!/bin/bash
# One gpurun call: GPU tests, variant A/B timings, io floor, config-5 timing.  Logs -> gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "=== io floor"; tools/ubench/io_floor; tools/ubench/io_floor_preload
for v in A B C; do
  for b in 65536 1048576; do
    echo "=== bench variant $v batch $b"
    DRM_HIP_LIBRARY=tools/variants/libdrm_$v.so timeout 300 python bench.py --no-cpu-baseline --steps 400 --warmup 20 --batch $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['launch_us'], 'us/launch', d['roofline']['frac'], 'frac', d['value'], 'evals/s')"
  done
done
echo "=== variant C parity subset"; DRM_HIP_LIBRARY=tools/variants/libdrm_C.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or golden or plan" 2>&1 | tail -3
echo "=== config 5"; timeout 300 python tools/bench_config5.py 2>&1 | tail -8
echo "=== bench default"; timeout 600 python bench.py 2>&1 | tail -1
} > gpurun_out/round.log 2>&1
tail -60 gpurun_out/round.log
