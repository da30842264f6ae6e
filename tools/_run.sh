cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mass_matrix.py tests/test_random_trees.py tests/test_max_sizes.py -m gpu -x -q -n 4 2>&1 | tail -3
timeout 300 python tools/probe_robots.py 1048576 crba 2>&1 | grep -v amdgpu.ids
