cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rnea_backward.py tests/test_random_trees.py -m gpu -q -x 2>&1 | grep -v Warn | tail -3
python tools/probe_robots.py 2>&1 | grep "n="
