#!/bin/bash
# gpurun -- bash tools/run_links_round.sh : ABI 13 (drm_walk_table_links) on the GPU — the GPU tests of the new path, the learn-dynamics
# step with and without it (torch's default and fused Adam), the kernels of one eager and one replayed step, the host profile of the
# eager steps
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; rm -rf $OUT/prof_stepdyn_links_*
cd /tmp && export TMPDIR=/tmp
python -m pytest $ROOT/tests/test_table_links.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > $OUT/test_table_links.log
( echo "== default: drm_walk_table_links (ABI 13: the table from the parameter tensors where they lie, the modules' forms inside the kernel)"
  python $ROOT/tools/bench_learn_dynamics.py 2>&1 | grep "^learnable"
  echo "== DRM_TABLE_LINKS=0: the modules' torch kernels, a cat of their outputs, drm_walk_table (before)"
  DRM_TABLE_LINKS=0 python $ROOT/tools/bench_learn_dynamics.py 2>&1 | grep "^learnable" ) > $OUT/learn_dynamics_links.txt
for mode in 1 0; do
  DRM_TABLE_LINKS=$mode rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_stepdyn_links_$mode -- python $ROOT/tools/probe_step5.py dyn graph 256 > $OUT/stepdyn_links_$mode.log 2>&1
  echo "== DRM_TABLE_LINKS=$mode (the last two of ten replays of the captured step)"; python $ROOT/tools/probe_step5.py --read $OUT/prof_stepdyn_links_$mode 2>&1
done > $OUT/step_dyn_kernels_links.txt
for mode in 1 0; do
  echo "== DRM_TABLE_LINKS=$mode: learn dynamics"; DRM_TABLE_LINKS=$mode python $ROOT/tools/profile_eager_step.py dyn 2>&1 | grep -v amdgpu.ids | head -40
  echo "== DRM_TABLE_LINKS=$mode: learn kinematics (configuration 5's literal loop)"; DRM_TABLE_LINKS=$mode python $ROOT/tools/profile_eager_step.py 2>&1 | grep -v amdgpu.ids | head -4
done > $OUT/eager_step_links.txt
tail -3 $OUT/test_table_links.log; cat $OUT/learn_dynamics_links.txt; grep "eager step\|host time\|==" $OUT/eager_step_links.txt
