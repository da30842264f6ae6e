#!/usr/bin/env python3
"""Development: per-wavefront time stamps of drm_fk_mse_links' two kernels (BASELINE configuration 5, 16 384 rows).
    tools/build_variants.sh tl_fin "-DDRM_TIMELINE -DDRM_TL_WHICH=1" tl_arm "-DDRM_TIMELINE -DDRM_TL_WHICH=2"
    DRM_HIP_LIBRARY=tools/variants/libdrm_tl_fin.so python tools/timeline_links.py fin
    DRM_HIP_LIBRARY=tools/variants/libdrm_tl_arm.so python tools/timeline_links.py arm"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample  # noqa: E402
from differentiable_robot_model_amd import backend  # noqa: E402
from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fin"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
SLOTS, WAVES = 8, 1 << 16
torch.manual_seed(0)
m, gt = load("iiwa7"), load("iiwa7")
for p in ("trans", "rot_angles"):
    m.make_link_param_learnable("iiwa_link_1", p, UnconstrainedTensor(1, 3))
ee = m._name_to_idx_map["iiwa_link_ee"]
dw = m._get_walk(("fk", (ee,)), targets=[ee])
links, base, sel = m._learnable_plan(dw)
pieces = [p.detach() for p in m._learnable_pieces(links)]
mask = m._kinematic_param_mask(dw)
q = sample(m, B)[0].cuda()
with torch.no_grad():
    want, _ = gt.compute_forward_kinematics(q, "iiwa_link_ee")
lib = backend.load_library()
fn = lib.drm_tl_read_fkb
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def read(n):
    buf = np.zeros((WAVES, SLOTS), np.uint64)
    torch.cuda.synchronize()
    assert fn(buf.ctypes.data, buf.nbytes, 1) == 0
    return buf[:n].astype(np.int64)


call = lambda: backend.fk_mse_links(dw.program, base, dw.ops_i, sel, dw.gsign, pieces, q, want, 7, mask, False)
g = torch.cuda.CUDAGraph()
call(); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(20):
        call()
g.replay(); read(8)
g.replay()
n = 8 if which == "fin" else B // 64
rec = read(n)
t0 = rec[:, 0].min()
rel = (rec[:, :7] - t0) * 0.01
np.set_printoptions(precision=2, suppress=True, linewidth=200)
if which == "fin":
    print("finish kernel, us after the first wavefront's first instruction; slots: 0 start, 1 rows requested, 2 (trig wave: d F / d rpy done), "
          "3 sums in LDS (drained), 4 past the barrier, 5 scatter done, 6 end (drained)")
    for w in range(n):
        print("wave %d" % w, rel[w])
else:
    print("chain kernel (LINKS), %d wavefronts, us after the first start; slots: 0 start, 1 parameters in LDS, 2 table built, "
          "3 chain + adjoints done, 5 partial sums published (drained)" % n)
    cols = [0, 1, 2, 4, 6, 3, 5]
    print("(slots in time order: 0 start, 1 parameters in LDS, 2 table built, 4 forward sweep done, 6 adjoint sweep at the learnable op, 3 chain done, 5 published)")
    print("median", np.median(rel[:, cols], axis=0))
    print("min   ", rel[:, cols].min(axis=0))
    print("max   ", rel[:, cols].max(axis=0))
    print("median per-wave duration of each phase:", np.median(np.diff(rec[:, cols], axis=1), axis=0) * 0.01)
