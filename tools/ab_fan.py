#!/usr/bin/env python3
"""A/B timing of BASELINE configuration 4 (Allegro, four fingertips, 65 536 rows) for whatever library DRM_HIP_LIBRARY names:
sample-major (drm_fk_fanout) and link-major (compute_forward_kinematics_links) launches, hipGraph of 100, best of 7."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
from ab_rnea import graph_time

TIPS = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]
h = load("allegro_left")
idx = [h._name_to_idx_map[t] for t in TIPS]
tag = os.path.basename(os.environ.get("DRM_HIP_LIBRARY", "libdrm_hip.so"))
for B in (65536, 1 << 18):
    q = sample(h, B)[0].cuda()
    for mode, what in (("off", "library kernels"), (None, "default (own kernel)")):      # round 6: a model runs its own kernels by default
        h.own_kernels = mode
        with torch.no_grad():
            t_links = graph_time(lambda: h.compute_forward_kinematics_links(q, TIPS), launches=100, reps=7)
            t_rows = graph_time(lambda: h._fk_targets(q, idx), launches=100, reps=7)
        print("%-22s B=%8d  %-22s fk 4 tips link-major %7.2f us   sample-major %7.2f us" % (tag, B, what, t_links, t_rows), flush=True)
