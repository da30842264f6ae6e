cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/gputest11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest11.log; grep -v "^  File" gpurun_out/gputest11.log | tail -25 | cut -c1-400
