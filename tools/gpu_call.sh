cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fanout or allegro" 2>&1 | tail -3
timeout 600 python tools/kernel_times.py 65536 2>&1 | grep -v amdgpu.ids | grep "allegro 4 tips" 
