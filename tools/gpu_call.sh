cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "all_links or kinematic_state" 2>&1 | grep -v Warn | tail -4
python tools/probe_api.py 65536 2>&1 | grep "B=" | cut -c1-90
