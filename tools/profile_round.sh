#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 evidence for the bench command.  Raw outputs -> gpurun_out/prof_*/,
# summarised afterwards by tools/pmc_summary.py <tag> into profiles/.
#   pass 1: --kernel-trace --stats            (per-kernel durations of the default bench run, JSON line kept)
#   pass 2: --kernel-trace --pmc FETCH_SIZE    (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together,
#   pass 3: --kernel-trace --pmc WRITE_SIZE     and counter runs must not be combined with API tracing)
#   pass 4: SQ counters for the instruction mix (arm kernels: kernel_bench.py, loop-form kernels: kernel_bench3.py)
#   pass 5: --kernel-trace --stats of tools/kernel_times.py (which kernel every entry point dispatches to)
# plus the unprofiled lines: bench.py default, bench.py --config 3, tools/kernel_times.py, the metric lab's I/O floor.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline"
python $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $ROOT/bench.py --config 3 --no-cpu-baseline > $OUT/bench_config3.json 2> $OUT/bench_config3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $B > $OUT/prof_stats.log 2>&1
# the same with the flags the round driver passes (a 20-launch timed region)
python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_k20 -- $B --gpus 1 --steps 20 --warmup 5 > $OUT/prof_stats_k20.log 2>&1
for batch in 65536 4194304; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$batch -- $B --no-large --steps 50 --warmup 5 --batch $batch > $OUT/prof_fetch_$batch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$batch -- $B --no-large --steps 50 --warmup 5 --batch $batch > $OUT/prof_write_$batch.log 2>&1
done
PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq -- python $ROOT/tools/kernel_bench.py 65536 1048576 > $OUT/prof_sq.log 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq3 -- python $ROOT/tools/kernel_bench3.py 65536 > $OUT/prof_sq3.log 2>&1
# robots with one long segment (an arm with its gripper / hand): RNEA, mass matrix, forward dynamics at 262 144 samples
for r in panda iiwa7_allegro; do
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/prof_sq4_$r -- python $ROOT/tools/kernel_bench4.py $r 262144 > $OUT/prof_sq4_$r.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_all -- python $ROOT/tools/kernel_times.py 65536 1048576 > $OUT/prof_all.log 2>&1
python $ROOT/tools/kernel_times.py > $OUT/kernel_times.txt 2>&1
python $ROOT/tools/probe_robots.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_robots.txt
python $ROOT/tools/bench_config5.py 2>&1 | grep '^config5\|^  kernels' > $OUT/config5.txt
[ -x $ROOT/tools/ubench/metric_lab ] && $ROOT/tools/ubench/metric_lab > $OUT/metric_lab.txt 2>&1
cd $ROOT
# keep what travels back small: per-dispatch counter rows of OUR kernels only
for f in $(find gpurun_out -name "*counter_collection.csv"); do
  head -1 $f > $f.small; grep "drm::" $f >> $f.small; mv $f.small $f
done
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
du -sh gpurun_out
tail -n 3 $OUT/bench_default.json $OUT/bench_config3.json | cut -c1-1500
