#!/bin/bash
# Round 3, measurement row: bench with the reference timed in-run; what rocprofv3 reports for kernels of
# known duration (empty / bytes-only / product) next to HIP events of the same process; one stats file per batch size.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
LAB=$ROOT/tools/ubench/metric_lab
$LAB 65536 overhead > $OUT/overhead_plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_overhead -- $LAB 65536 overhead > $OUT/overhead_under_rocprofv3.txt 2>&1
B="python $ROOT/bench.py --no-cpu-baseline --no-large --no-traffic"
for batch in 65536 4194304 16777216; do
  steps=200; [ $batch -gt 65536 ] && steps=50
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_$batch -- $B --batch $batch --steps $steps > $OUT/prof_stats_$batch.log 2>&1
done
cd $ROOT
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete
du -sh gpurun_out
tail -c 3000 $OUT/bench_default.json; cat $OUT/overhead_plain.txt $OUT/overhead_under_rocprofv3.txt | grep -v "^W\|^E2" | tail -20
