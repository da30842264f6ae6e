cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/gputest12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest12.log; grep -v "^  File" gpurun_out/gputest12.log | tail -12 | cut -c1-400
timeout 900 python tools/probe_robots.py 2>&1 | grep -v amdgpu.ids > gpurun_out/probe_robots2.txt; cat gpurun_out/probe_robots2.txt
