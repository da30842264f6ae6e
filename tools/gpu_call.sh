cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warn | tail -3
python tools/kernel_times.py 65536 1048576 2>&1 | grep "rnea bwd    allegro"
python tools/probe_robots.py 2>&1 | grep "trifinger" | cut -c100-
