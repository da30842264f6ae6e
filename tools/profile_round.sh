#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 evidence for the bench command.  Raw outputs -> gpurun_out/prof_*/,
# summarised afterwards by tools/pmc_summary.py into profiles/.
#   pass 1: --kernel-trace --stats            (per-kernel durations of the default bench run)
#   pass 2: --kernel-trace --pmc FETCH_SIZE    (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together,
#   pass 3: --kernel-trace --pmc WRITE_SIZE     and counter runs must not be combined with API tracing)
#   pass 4: SQ counters for the instruction mix
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $B > $OUT/prof_stats.log 2>&1
for batch in 65536 4194304; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch_$batch -- $B --steps 50 --warmup 5 --batch $batch > $OUT/prof_fetch_$batch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write_$batch -- $B --steps 50 --warmup 5 --batch $batch > $OUT/prof_write_$batch.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/prof_sq -- python $ROOT/tools/kernel_bench.py 65536 1048576 > $OUT/prof_sq.log 2>&1
cd $ROOT
find gpurun_out -name "*.csv" | head -40
# keep what travels back small: per-dispatch counter rows of OUR kernels only
for f in $(find gpurun_out -name "*counter_collection.csv"); do
  head -1 $f > $f.small; grep "drm::" $f >> $f.small; mv $f.small $f
done
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
du -sh gpurun_out
