#!/bin/bash
# Why does the metric kernel lose 15-25 % of its HBM rate between 2^22 and 2^24 rows?  
#   1. batch sweep: powers of two against neighbouring sizes that are not (array sizes stop being aligned to each other)
#   2. rocprofv3 --pmc passes at 2^22 and 2^24: address translation (UTCL1), L2 -> fabric write / read stalls
# Run on the GPU box: bash tools/probe_2p24.sh  -> gpurun_out/p24/
ROOT=$(pwd); OUT=$ROOT/gpurun_out/p24; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-configs --no-large --steps 20 --warmup 3"
for batch in 4194304 4000000 8388608 8000000 12582912 16777216 16000000 16777152 20000000; do
  $B --batch $batch 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('batch %9d  launch %8.1f us  %6.1f ps/row  frac %.3f' % ($batch, r['launch_us'], r['launch_us']*1e6/$batch, r['frac']))"
done | tee $OUT/sweep.txt
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum" \
           "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  for batch in 4194304 16777216; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_${i}_$batch -- $B --batch $batch --steps 10 > $OUT/pmc_${i}_$batch.log 2>&1
  done
done
cd $ROOT
python - <<'PY' | tee gpurun_out/p24/counters.txt
import csv, glob, collections
for batch in (4194304, 16777216):
    acc = collections.defaultdict(list)
    for path in glob.glob('gpurun_out/p24/pmc_*_%d/**/*counter_collection.csv' % batch, recursive=True):
        for row in csv.DictReader(open(path)):
            if 'fk_jacobian_arm_kernel' in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
    print('batch', batch)
    for k in sorted(acc):
        v = acc[k]; print('   %-48s %14.0f per launch  %10.4f per row  (%d dispatches)' % (k, sum(v)/len(v), sum(v)/len(v)/batch, len(v)))
PY
find gpurun_out/p24 -name "*.csv" -delete; find gpurun_out/p24 -type d -empty -delete 2>/dev/null; true
