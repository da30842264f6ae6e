cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_multirank.py tests/test_max_sizes.py -m gpu -x -q > gpurun_out/gputest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest2.log; tail -30 gpurun_out/gputest2.log
timeout 300 python bench.py --gpus 2 --config 3 --steps 20 --warmup 5 --shared-gpu --verify-gather > gpurun_out/bench_c3_shared2.json 2> gpurun_out/bench_c3_shared2.err; tail -c 1500 gpurun_out/bench_c3_shared2.json
timeout 300 python tools/ab_rnea.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab_rnea2.txt
DRM_HIP_LIBRARY=$PWD/tools/variants/libdrm_jrot.so timeout 300 python tools/ab_rnea.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab_rnea2.txt
cat gpurun_out/ab_rnea2.txt
