cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "backward_of_an_arm" 2>&1 | grep -v Warn | tail -12
python tools/probe_robots.py 2>&1 | grep "n="
