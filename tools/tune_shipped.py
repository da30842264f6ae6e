#!/usr/bin/env python3
"""Run on an MI355X (gpurun -- python tools/tune_shipped.py): `model.specialize(tune=True)` for the robots whose whole-tree kernels
ship with the package (specialize.SHIPPED_TREES) — every entry point timed with the robot's own kernel and with the library's, the
faster one kept — and the tuning records (drm_special_<source key>.tuned.json) copied to gpurun_out/tuned/, from where they go into
differentiable-robot-model_amd/csrc/special_cache/ (tracked: a model then attaches exactly those entry points by default)."""
import glob
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
cache = tempfile.mkdtemp(prefix="drm_tune_")
os.environ["DRM_SPECIAL_CACHE"] = cache      # (a fresh run-time cache: only this run's records)
from gpu_probe import load  # noqa: E402
from differentiable_robot_model_amd import specialize as sp  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "tuned")
os.makedirs(out, exist_ok=True)
print("model.specialize(tune=True), 2^19 rows, us per launch")
for robot in sp.SHIPPED_TREES:
    m = load(robot)
    m.own_kernels = "off"                    # (nothing attached before the measurement)
    report = m.specialize(tune=True)
    for kernel, r in report.items():
        print("%-8s %-26s own %8.2f   library %8.2f   kept %s" % (robot, kernel, r["own_us"], r["library_us"], r["kept"]))
for f in glob.glob(os.path.join(cache, "*.tuned.json")):
    shutil.copy(f, out)
    print(os.path.basename(f), json.load(open(f))["kept"])
