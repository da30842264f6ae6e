#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof_*) into profiles/rNN_*.

  rNN_bench_kernel_stats.csv   the --kernel-trace --stats table of the default bench run
  rNN_pmc_traffic.json         HBM bytes per launch of the metric kernel: FETCH_SIZE (x2: gfx950 tallies 128-B read
                               requests at 64 B, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both reported in KiB
  rNN_sq_counters.md           per-wave instruction counts / cycle split of the hot kernels (arm and loop-form)
  rNN_all_kernels_rocprof_stats.csv, rNN_bench_{default,config3,under_rocprofv3}.json, rNN_kernel_times.txt, rNN_metric_lab.txt, rNN_probe_robots.txt, rNN_config5.txt
usage: python tools/pmc_summary.py r01
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prof = os.path.join(ROOT, "profiles")


def rows(pattern):
    for path in glob.glob(os.path.join(OUT, pattern)):
        with open(path) as f:
            for r in csv.DictReader(f):
                yield r


stats = glob.glob(os.path.join(OUT, "prof_stats", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(prof, tag + "_bench_kernel_stats.csv"))
stats = glob.glob(os.path.join(OUT, "prof_stats_k20", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(prof, tag + "_bench_k20_kernel_stats.csv"))
stats = glob.glob(os.path.join(OUT, "prof_all", "*", "*_kernel_stats.csv"))
if stats:   # rows of our kernels only
    with open(stats[0]) as f, open(os.path.join(prof, tag + "_all_kernels_rocprof_stats.csv"), "w") as g:
        for i, line in enumerate(f):
            if i == 0 or "drm::" in line:
                g.write(line)


def last_json_line(path):
    if not os.path.exists(path):
        return None
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    return lines[-1] if lines else None


for src, dst in (("prof_stats.log", "_bench_under_rocprofv3.json"), ("bench_default.json", "_bench_default.json"),
                 ("prof_stats_k20.log", "_bench_k20_under_rocprofv3.json"), ("bench_k20.json", "_bench_k20.json"),
                 ("bench_config3.json", "_bench_config3.json")):
    line = last_json_line(os.path.join(OUT, src))
    if line:
        with open(os.path.join(prof, tag + dst), "w") as f:
            f.write(line + "\n")
for src, dst in (("kernel_times.txt", "_kernel_times.txt"), ("metric_lab.txt", "_metric_lab.txt"), ("probe_robots.txt", "_probe_robots.txt"),
                 ("config5.txt", "_config5.txt")):
    if os.path.exists(os.path.join(OUT, src)):
        shutil.copy(os.path.join(OUT, src), os.path.join(prof, tag + dst))

# round 3 additions: one stats file per batch size, the eagerly launched run, rocprofv3's overhead on kernels of known duration,
# the two-rank lines (ranks sharing the GPU), which kernel the robot probe dispatches to, the arm-dynamics A/B
for batch in (65536, 4194304, 16777216):
    st = glob.glob(os.path.join(OUT, "prof_stats_%d" % batch, "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(prof, "%s_bench_kernel_stats_%d.csv" % (tag, batch)))
for d, name in (("prof_stats_eager", "_bench_eager_kernel_stats.csv"), ("prof_overhead", "_overhead_kernel_stats.csv")):
    st = glob.glob(os.path.join(OUT, d, "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(prof, tag + name))
st = glob.glob(os.path.join(OUT, "prof_api", "*", "*_kernel_stats.csv"))
if st:
    with open(st[0]) as f, open(os.path.join(prof, tag + "_probe_api_rocprof_stats.csv"), "w") as g:
        for i, line in enumerate(f):
            if i == 0 or "drm::" in line:
                g.write(line)
st = glob.glob(os.path.join(OUT, "prof_robots", "*", "*_kernel_stats.csv"))
if st:
    with open(st[0]) as f, open(os.path.join(prof, tag + "_probe_robots_rocprof_stats.csv"), "w") as g:
        for i, line in enumerate(f):
            if i == 0 or "drm::" in line:
                g.write(line)
for src, dst in (("bench_config3_two_ranks_shared_gpu.json", "_bench_config3_two_ranks_shared_gpu.json"),
                 ("bench_config3_p2p_two_ranks_shared_gpu.json", "_bench_config3_p2p_two_ranks_shared_gpu.json"),
                 ("bench_metric_two_ranks_shared_gpu.json", "_bench_metric_two_ranks_shared_gpu.json"),
                 ("prof_stats_eager.log", "_bench_eager_under_rocprofv3.json")):
    line = last_json_line(os.path.join(OUT, src))
    if line:
        with open(os.path.join(prof, tag + dst), "w") as f:
            f.write(line + "\n")
st = glob.glob(os.path.join(OUT, "prof_all_own", "*", "*_kernel_stats.csv"))
if st:      # round 5: kernel_times.py under DRM_SPECIALIZE=1 — which kernels the entry points dispatch to once a robot has its own
    with open(st[0]) as f, open(os.path.join(prof, tag + "_all_kernels_own_rocprof_stats.csv"), "w") as g:
        for i, line in enumerate(f):
            if i == 0 or "drm" in line:
                g.write(line)
if os.path.exists(os.path.join(OUT, "kernel_times_own.txt")):
    shutil.copy(os.path.join(OUT, "kernel_times_own.txt"), os.path.join(prof, tag + "_kernel_times_own_kernels.txt"))
st = glob.glob(os.path.join(OUT, "prof_cfg", "*", "*_kernel_stats.csv"))
if st:      # round 4: the kernels of BASELINE configurations 2-5 (tools/kernel_bench.py configs), eagerly launched
    shutil.copy(st[0], os.path.join(prof, tag + "_configs_kernel_stats.csv"))
for src, dst in (("overhead_plain.txt", "_overhead_plain.txt"), ("ab_rnea.txt", "_ab_rnea_final.txt"), ("io_floors_2p20.txt", "_io_floors_2p20.txt"),
                 ("probe_api.txt", "_probe_api.txt"), ("probe_special.txt", "_probe_special.txt"), ("api_latency.txt", "_api_latency.txt"),
                 ("ab_fan.txt", "_ab_fan_final.txt"), ("timeline.txt", "_timeline.txt"),
                 ("ab_learnable_arm.txt", "_ab_learnable_arm.txt"), ("probe_chunks.txt", "_probe_chunks.txt"),
                 ("probe_nonfinite.txt", "_probe_nonfinite.txt"), ("step5_kernels.txt", "_step5_kernels.txt"),
                 ("step_dyn_kernels.txt", "_step_dyn_kernels.txt"), ("ab_fk_mse_links.txt", "_ab_fk_mse_links.txt"),
                 ("timeline_links.txt", "_timeline_links.txt"), ("learn_dynamics.txt", "_learn_dynamics.txt"),
                 ("learn_dynamics_links.txt", "_learn_dynamics_links.txt"), ("step_dyn_kernels_links.txt", "_step_dyn_kernels_links.txt"),
                 ("io_floors_c3_shard.txt", "_io_floors_c3_shard.txt"),
                 # round 6: bench.py's stdout is the compact line; the full records of the runs
                 ("bench_default_detail.json", "_bench_default_detail.json"), ("bench_k20_detail.json", "_bench_k20_detail.json"),
                 ("bench_config3_detail.json", "_bench_config3_detail.json"),
                 ("bench_config3_p2p_two_ranks_shared_gpu_detail.json", "_bench_config3_p2p_two_ranks_shared_gpu_detail.json"),
                 ("bench_config3_two_ranks_shared_gpu_detail.json", "_bench_config3_two_ranks_shared_gpu_detail.json"),
                 ("bench_metric_two_ranks_shared_gpu_detail.json", "_bench_metric_two_ranks_shared_gpu_detail.json")):
    if os.path.exists(os.path.join(OUT, src)):
        shutil.copy(os.path.join(OUT, src), os.path.join(prof, tag + dst))
if os.path.exists(os.path.join(OUT, "overhead_under_rocprofv3.txt")):
    with open(os.path.join(OUT, "overhead_under_rocprofv3.txt")) as f, open(os.path.join(prof, tag + "_overhead_under_rocprofv3.txt"), "w") as g:
        g.writelines(ln for ln in f if ln.startswith(("TIME", "EAGER", "device")))

traffic = {"kernel": None, "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- "
                                      "python bench.py --no-cpu-baseline --steps 50 --warmup 5 [--batch B]",
           "per_batch": {}}
for batch in (65536, 4194304):
    rec = {}
    for name, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        vals = []
        for r in rows("prof_%s_%d/*/*_counter_collection.csv" % (name, batch)):
            if r["Counter_Name"] == key and "fk_jacobian_arm_kernel<8, 7, true" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
                traffic["kernel"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if vals:
            rec[key + "_KiB_mean"] = sum(vals) / len(vals)
            rec[key + "_launches"] = len(vals)
    if len(rec) == 4:
        alg = 224 * batch
        fetch = rec["FETCH_SIZE_KiB_mean"] * 1024 * 2   # gfx950 correction
        write = rec["WRITE_SIZE_KiB_mean"] * 1024
        # round 6: a launch whose outputs exceed the Infinity Cache goes as chunks of 2^21 rows (csrc/drm_arm_kernels.hip
        # STREAM_CHUNK_TILES): the counters are per dispatch, a "launch" of the API is `chunks` of them
        chunks = batch // (1 << 21) if batch * 196 > 256 * 1024 * 1024 and batch > (1 << 21) else 1
        rec.update(fetch_bytes_corrected=fetch * chunks, write_bytes=write * chunks, traffic_bytes_per_launch=(fetch + write) * chunks,
                   dispatches_per_launch=chunks, algorithmic_bytes_per_launch=alg,
                   traffic_over_algorithmic=(fetch + write) * chunks / alg)
        traffic["per_batch"][str(batch)] = rec
with open(os.path.join(prof, tag + "_pmc_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)

acc = defaultdict(lambda: defaultdict(list))
for r in (list(rows("prof_sq/*/*_counter_collection.csv")) + list(rows("prof_sq3/*/*_counter_collection.csv")) + list(rows("prof_sq_cfg/*/*_counter_collection.csv")) +
          list(rows("prof_sq4_*/*/*_counter_collection.csv")) + list(rows("prof_sq5_*/*/*_counter_collection.csv")) +
          list(rows("prof_sq6_*/*/*_counter_collection.csv"))):     # (sq6: the robots' own kernels, DRM_SPECIALIZE=1)
    k = r["Kernel_Name"].split("(")[0].replace("void drm::", "")
    acc[(k, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"]
lines = ["# SQ counters, %s (rocprofv3 --pmc, MI355X; per wave = counter / SQ_WAVES; cycle counters are quad-cycles)" % tag,
         "(persistent kernels — rnea_records_kernel, crba_rows_kernel, forward_dynamics_aba_kernel, the backward kernels — run several 64-sample tiles per wave:"
         " tiles = ceil(B / 64), waves = the grid)", "",
         "| kernel | grid threads | waves | VALU | SALU | SMEM | LDS | WAVE_CYCLES | WAIT_ANY | ACTIVE_INST_ANY |",
         "|---|---|---|---|---|---|---|---|---|---|"]
for (k, grid), c in sorted(acc.items()):
    if "SQ_WAVES" not in c:
        continue
    waves = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])
    per = [sum(c[x]) / len(c[x]) / waves if x in c else float("nan") for x in cols]
    lines.append("| %s | %d | %d | %s |" % (k, grid, waves, " | ".join("%.0f" % v for v in per)))
with open(os.path.join(prof, tag + "_sq_counters.md"), "w") as f:
    f.write("\n".join(lines) + "\n")
print(open(os.path.join(prof, tag + "_sq_counters.md")).read())
print(json.dumps(traffic["per_batch"], indent=1)[:1200])
