cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning\|return Diff\|^$\|warnings.html\|^tests/\|^  /" | tail -8 | cut -c1-300
python __graft_entry__.py smoke 2>&1 | tail -4
