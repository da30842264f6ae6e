cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_logic.py tests/test_mass_matrix.py tests/test_random_trees.py -m gpu -q -s -x -k "carries or scratch or mass or tile" 2>&1 | grep "mass matrix\|passed\|failed\|Error\|assert" | head -30
python tools/probe_robots.py 1048576 crba 2>&1 | grep CRBA
