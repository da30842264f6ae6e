cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gputest3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest3.log; tail -40 gpurun_out/gputest3.log
