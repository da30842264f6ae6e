#!/bin/bash
# Run on the GPU box (gpurun -- bash tools/profile_round.sh): every measurement of a round.  Raw outputs -> gpurun_out/,
# condensed afterwards by tools/pmc_summary.py <tag> into profiles/.
#   unprofiled lines: bench.py default / driver flags / --config 3 / two ranks sharing the GPU; kernel_times, probe_robots,
#                     config 5, the A/B of the arm dynamics kernels, the metric lab (I/O floors, rocprofv3's own overhead)
#   rocprofv3 --kernel-trace --stats: bench default (graph replay), the same eagerly launched, one run per batch size,
#                     kernel_times + probe_robots (which kernel every entry point dispatches to)
#   rocprofv3 --pmc:  FETCH_SIZE / WRITE_SIZE (separate passes, never with API tracing), SQ counters of the hot kernels
#                     (kernel_bench.py hot / configs / hand / dynamics / backward)
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-traffic --detail /dev/null"
BM="$B --no-configs"   # the metric leg alone (counter passes, per-batch stats)
# round 6: stdout of bench.py is ONE compact line (< 4 KB); the full record goes to --detail.  The default run here is the FULL one
# (--configs full: the reference beside every leg, counter traffic per leg); the driver's own form (k20) runs the fast legs
( time python $ROOT/bench.py --configs full --detail $OUT/bench_default_detail.json ) > $OUT/bench_default.json 2> $OUT/bench_default.err
( time python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_k20_detail.json ) > $OUT/bench_k20.json 2> $OUT/bench_k20.err
python $ROOT/bench.py --config 3 --detail $OUT/bench_config3_detail.json > $OUT/bench_config3.json 2> $OUT/bench_config3.err
python $ROOT/bench.py --gpus 2 --config 3 --gather all --steps 20 --warmup 5 --shared-gpu --verify-gather --no-cpu-baseline --detail $OUT/bench_config3_two_ranks_shared_gpu_detail.json > $OUT/bench_config3_two_ranks_shared_gpu.json 2> $OUT/bench_config3_two_ranks.err
python $ROOT/bench.py --gpus 2 --config 3 --gather p2p --steps 20 --warmup 5 --shared-gpu --verify-gather --no-cpu-baseline --detail $OUT/bench_config3_p2p_two_ranks_shared_gpu_detail.json > $OUT/bench_config3_p2p_two_ranks_shared_gpu.json 2> $OUT/bench_config3_p2p_two_ranks.err
python $ROOT/bench.py --gpus 2 --gather --steps 20 --warmup 5 --shared-gpu --verify-gather --no-large --no-cpu-baseline --detail $OUT/bench_metric_two_ranks_shared_gpu_detail.json > $OUT/bench_metric_two_ranks_shared_gpu.json 2> $OUT/bench_metric_two_ranks.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $B > $OUT/prof_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_k20 -- $B --gpus 1 --steps 20 --warmup 5 > $OUT/prof_stats_k20.log 2>&1
# one stats file per batch size; 65 536 also launched eagerly (graph replays report the tracer's own period, r03_rocprof_overhead.md)
for batch in 65536 4194304 16777216; do
  steps=200; [ $batch -gt 65536 ] && steps=50
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$batch -- $BM --no-large --batch $batch --steps $steps > $OUT/prof_stats_$batch.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_eager -- $BM --no-large --no-graph > $OUT/prof_stats_eager.log 2>&1
for batch in 65536 4194304; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$batch -- $BM --no-large --steps 50 --warmup 5 --batch $batch > $OUT/prof_fetch_$batch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$batch -- $BM --no-large --steps 50 --warmup 5 --batch $batch > $OUT/prof_write_$batch.log 2>&1
done
PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
DRM_SPECIALIZE=0 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq -- python $ROOT/tools/kernel_bench.py hot 65536 131072 1048576 > $OUT/prof_sq.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq_cfg -- python $ROOT/tools/kernel_bench.py configs > $OUT/prof_sq_cfg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg -- python $ROOT/tools/kernel_bench.py configs > $OUT/prof_cfg.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq3 -- python $ROOT/tools/kernel_bench.py hand 65536 > $OUT/prof_sq3.log 2>&1
for r in panda iiwa7_allegro allegro_left; do
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq4_$r -- python $ROOT/tools/kernel_bench.py dynamics $r 262144 > $OUT/prof_sq4_$r.log 2>&1
done
for r in panda_no_gripper panda allegro_left iiwa7_allegro; do
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq5_$r -- python $ROOT/tools/kernel_bench.py backward $r 262144 > $OUT/prof_sq5_$r.log 2>&1
done
# the robots' OWN kernels (constants folded in; round 6: what a plain model runs by default, DRM_SPECIALIZE=0 switches them off) — SQ
# counters, which kernel every entry point dispatches to, and the timings beside the library's
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq6_hot -- python $ROOT/tools/kernel_bench.py hot 131072 1048576 > $OUT/prof_sq6_hot.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq6_dyn -- python $ROOT/tools/kernel_bench.py dynamics panda_no_gripper 262144 > $OUT/prof_sq6_dyn.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq6_bwd -- python $ROOT/tools/kernel_bench.py backward panda_no_gripper 262144 > $OUT/prof_sq6_bwd.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_all_own -- python $ROOT/tools/kernel_times.py 65536 1048576 > $OUT/prof_all_own.log 2>&1
python $ROOT/tools/kernel_times.py 2>&1 | grep -v amdgpu.ids > $OUT/kernel_times_own.txt
DRM_SPECIALIZE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_all -- python $ROOT/tools/kernel_times.py 65536 1048576 > $OUT/prof_all.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_robots -- python $ROOT/tools/probe_robots.py > $OUT/prof_robots.log 2>&1
DRM_SPECIALIZE=0 python $ROOT/tools/kernel_times.py > $OUT/kernel_times.txt 2>&1
for r in iiwa7 panda_no_gripper; do python $ROOT/tools/ab_learnable_arm.py $r 2>&1 | grep -v amdgpu.ids; done > $OUT/ab_learnable_arm.txt
# round 6: which kernels one training step launches (configuration 5 through fk_mse_loss with and without drm_fk_mse_links; the
# learn-dynamics example), the A/B of drm_fk_mse_links against the launches it replaces, its kernels' per-wavefront time stamps
for mode in 1 0; do
  DRM_FK_MSE_LINKS=$mode rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_step5_$mode -- python $ROOT/tools/probe_step5.py > $OUT/step5_$mode.log 2>&1
  echo "== DRM_FK_MSE_LINKS=$mode"; python $ROOT/tools/probe_step5.py --read $OUT/prof_step5_$mode 2>&1 | tail -12
done > $OUT/step5_kernels.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_stepdyn -- python $ROOT/tools/probe_step5.py dyn 256 > $OUT/stepdyn.log 2>&1
python $ROOT/tools/probe_step5.py --read $OUT/prof_stepdyn > $OUT/step_dyn_kernels.txt 2>&1
# ABI 13 (drm_walk_table_links): the learn-dynamics step with and without it — step times, the kernels of a replayed step, host time
bash $ROOT/tools/run_links_round.sh > $OUT/links_round.log 2>&1
python $ROOT/tools/ab_fk_mse_links.py 2>&1 | grep -v amdgpu.ids > $OUT/ab_fk_mse_links.txt
( echo "== default: the arm's own kernel for this set of learnable blocks (shipped)"; python $ROOT/tools/bench_learn_dynamics.py 2>&1 | grep "^learnable"
  echo "== DRM_SPECIALIZE=0: the library's kernels (the round-5 path)"; DRM_SPECIALIZE=0 python $ROOT/tools/bench_learn_dynamics.py 2>&1 | grep "^learnable" ) > $OUT/learn_dynamics.txt
if [ -f $ROOT/tools/variants/libdrm_tl_fin.so ]; then
  DRM_HIP_LIBRARY=$ROOT/tools/variants/libdrm_tl_fin.so python $ROOT/tools/timeline_links.py fin 2>&1 | grep -v amdgpu.ids | sed 's/-[0-9.]*e+1[12]/      --/g' > $OUT/timeline_links.txt
  DRM_HIP_LIBRARY=$ROOT/tools/variants/libdrm_tl_arm.so python $ROOT/tools/timeline_links.py arm 2>&1 | grep -v amdgpu.ids >> $OUT/timeline_links.txt
fi
python $ROOT/tools/probe_chunks.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_chunks.txt
python $ROOT/tools/probe_nonfinite.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|return Diff" > $OUT/probe_nonfinite.txt
python $ROOT/tools/probe_robots.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_robots.txt
python $ROOT/tools/ab_rnea.py 2>&1 | grep -v amdgpu.ids > $OUT/ab_rnea.txt
for b in 1048576 65536; do python $ROOT/tools/probe_api.py $b 2>&1 | grep "B="; done > $OUT/probe_api.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_api -- python $ROOT/tools/probe_api.py 1048576 > $OUT/prof_api.log 2>&1
python $ROOT/tools/bench_config5.py 2>&1 | grep '^config5\|^  kernels' > $OUT/config5.txt
python $ROOT/tools/probe_special.py fetch 2>&1 | grep "^fetch" > $OUT/probe_special.txt
python $ROOT/tools/api_profile.py 2>&1 | grep "us per eager call" > $OUT/api_latency.txt
DRM_NO_HOSTCALL=1 python $ROOT/tools/api_profile.py 2>&1 | grep "learned model" | sed "s/^/(DRM_NO_HOSTCALL=1: the Python path) /" >> $OUT/api_latency.txt
python $ROOT/tools/ab_fan.py 2>&1 | grep "B=" > $OUT/ab_fan.txt
if [ -f $ROOT/tools/variants/libdrm_timeline.so ]; then
  DRM_HIP_LIBRARY=$ROOT/tools/variants/libdrm_timeline.so python $ROOT/tools/timeline.py 2>&1 | grep -v amdgpu.ids > $OUT/timeline.txt
fi
if [ -x $ROOT/tools/ubench/metric_lab ]; then
  $ROOT/tools/ubench/metric_lab 1048576 floors > $OUT/io_floors_2p20.txt 2>&1
  $ROOT/tools/ubench/metric_lab 131072 floors 2>&1 | grep "^device\|config 3\|inverse dynamics, 7 DoF" > $OUT/io_floors_c3_shard.txt
  $ROOT/tools/ubench/metric_lab > $OUT/metric_lab.txt 2>&1
  $ROOT/tools/ubench/metric_lab 65536 overhead > $OUT/overhead_plain.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_overhead -- $ROOT/tools/ubench/metric_lab 65536 overhead > $OUT/overhead_under_rocprofv3.txt 2>&1
fi
cd $ROOT
# keep what travels back small: per-dispatch counter rows of OUR kernels only
for f in $(find gpurun_out -name "*counter_collection.csv"); do
  head -1 $f > $f.small; grep "drm" $f >> $f.small; mv $f.small $f
done
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
du -sh gpurun_out
tail -c 1200 $OUT/bench_default.json; echo; tail -c 600 $OUT/bench_config3.json; echo; cat $OUT/kernel_times.txt | grep -v amdgpu | tail -30
