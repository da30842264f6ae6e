cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_random_trees.py -m gpu -q -k "generated_arm" 2>&1 | grep -v Warn | grep "passed\|failed\|Error\|assert" | head -30
