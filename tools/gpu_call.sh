cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_mass_matrix.py tests/test_gpu_parity.py tests/test_random_trees.py tests/test_golden_wide.py -m gpu -q -x 2>&1 | grep -v Warn | tail -3
python tools/probe_robots.py 1048576 crba 2>&1 | grep CRBA
python tools/kernel_times.py 65536 1048576 2>&1 | grep "crba        allegro"
