cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in iiwa7_allegro fetch panda; do
timeout 600 python -m pytest tests/test_random_trees.py -m gpu -q -x -k "persistent and $r" > gpurun_out/gputest10_$r.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputest10_$r.log; grep -v "^  File\|^$" gpurun_out/gputest10_$r.log | tail -12 | cut -c1-300
done
