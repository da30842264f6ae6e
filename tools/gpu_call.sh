cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warn | tail -3
python tools/probe_robots.py 2>&1 | grep "n=" | grep "fetch \|trifinger"
