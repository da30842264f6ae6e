cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-large --steps 50 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print({k:r[k] for k in ('frac','traffic','traffic_measured_in_this_run','traffic_detail','traffic_over_algorithmic','traffic_source')})"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bench_line" 2>&1 | tail -3
