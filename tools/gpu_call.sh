cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/gputest7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest7.log; tail -25 gpurun_out/gputest7.log | cut -c1-600
timeout 600 python tools/kernel_times.py 65536 2>&1 | grep -v amdgpu.ids | grep "allegro 4 tips\|config\|fk_jacobian\|^fk " > gpurun_out/kt_fan.txt; cat gpurun_out/kt_fan.txt
timeout 900 python tools/probe_robots.py 2>&1 | grep -v amdgpu.ids > gpurun_out/probe_robots.txt; cat gpurun_out/probe_robots.txt
