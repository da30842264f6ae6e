cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v Warn | grep "Error\|passed\|failed\|^FAILED" | tail
