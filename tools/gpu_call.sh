cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep "forward dynamics\|passed\|failed\|Error" | head -20
