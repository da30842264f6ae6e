#!/usr/bin/env python3
"""Executed FP32 flops per evaluation of the Panda kernels, counted from the ISA hipcc writes for them (no GPU needed): every unit
is compiled to assembly with ITS Makefile flags, the robot's own kernels from the sources specialize.py generates; per kernel the
VALU instructions are weighted (v_pk_fma_f32 4, v_pk_mul / v_pk_add 2, v_fma / v_fmac / v_fmaak / v_fmamk 2, v_mul / v_add / v_sub 1,
v_sin / v_cos / v_rsq / v_rcp / v_sqrt 1; the fp64 instructions of the rare large-angle path and everything that is not arithmetic
count 0) and divided by the samples a lane carries.  "Executed", not "algorithmic": a product with a constant zero that the table-driven
kernels still multiply out counts there and is gone from the robot's own kernels.  -> profiles/r05_flops.json, which
tools/kernel_times.py reads for its vector-FP32 column.

    python tools/flop_count.py [out.json]
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "differentiable-robot-model_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
BASE = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-w", "-S", "--cuda-device-only", "-I", CSRC]
PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
NOSLP = ["-fno-slp-vectorize"]
WEIGHTS = [(r"v_pk_fma_f32", 4), (r"v_pk_(mul|add)_f32", 2), (r"v_(fma|fmac|fmaak|fmamk)_f32", 2), (r"v_(mul|add|sub|subrev)_f32", 1),
           (r"v_(sin|cos|rsq|rcp|sqrt)_f32", 1)]
# (label, unit or generated source, flags, substring of the demangled kernel name, samples per lane)
LIBRARY = [
    ("fk_jacobian library", "drm_arm_kernels.hip", PRELOAD, "fk_jacobian_arm_kernel<8, 7, true, 1, false, true>", 1),
    ("fk library", "drm_arm_kernels.hip", PRELOAD, "fk_jacobian_arm_kernel<8, 7, false, 1, false, false>", 1),
    ("rnea library (two samples per lane)", "drm_arm_dynamics.hip", PRELOAD + NOSLP, "rnea_arm2_kernel<8, 7, 7, false>", 2),
    ("rnea library (one sample per lane)", "drm_arm_dynamics.hip", PRELOAD + NOSLP, "rnea_arm_kernel<8, 7, 7, true>", 1),
    ("fk+rnea library (two samples per lane)", "drm_arm_dynamics.hip", PRELOAD + NOSLP, "fk_rnea_arm2_kernel<8, 7, 7, false>", 2),
    ("crba library", "drm_crba.hip", NOSLP, "crba_arm_kernel<8, 7, 7>", 1),
    ("fwd dyn library", "drm_forward_dynamics.hip", NOSLP, "forward_dynamics_arm_kernel<8, 7, 7>", 1),
    ("rnea bwd library", "drm_rnea_backward.hip", NOSLP, "rnea_backward_arm_kernel<8, 7, 7>", 1),
]
OWN = [
    ("rnea own", "rnea", "drm_rnea_arm_static", 2),
    ("fk+rnea own", "fk_rnea", "drm_fk_rnea_arm_static", 2),
    ("crba own", "dyn", "drm_crba_arm_static", 1),
    ("fwd dyn own", "dyn", "drm_fd_arm_static", 1),
    ("rnea bwd own (input gradients)", "dyn", "drm_rnea_backward_arm_static", 1),
]


def kernels_of(asm_path):
    """{demangled name: [instruction mnemonics]} of every kernel in an assembly file."""
    text = open(asm_path).read()
    out = {}
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)s_endpgm", text, re.S | re.M):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        out[name] = [l.split()[0] for l in m.group(2).splitlines() if re.match(r"\s+[vs]_|\s+ds_|\s+global_", l)]
    return out


def flops(mnemonics):
    total = 0
    for ins in mnemonics:
        if "f64" in ins:
            continue
        for pat, w in WEIGHTS:
            if re.fullmatch(pat + r"(_e32|_e64)?", ins):
                total += w
                break
    return total


def compile_to_asm(src, flags, tmp):
    out = os.path.join(tmp, os.path.basename(src) + ".s")
    subprocess.check_call([HIPCC] + BASE + list(flags) + ["-o", out, src], stderr=subprocess.DEVNULL)
    return out


def main():
    from helpers import load_model
    from differentiable_robot_model_amd import specialize as sp
    result = {"robot": "panda_no_gripper", "weights": {p: w for p, w in WEIGHTS}, "kernels": {}}
    with tempfile.TemporaryDirectory() as tmp:
        cache = {}
        for label, unit, flags, want, per_lane in LIBRARY:
            key = (unit, tuple(flags))
            if key not in cache:
                cache[key] = kernels_of(compile_to_asm(os.path.join(CSRC, unit), flags, tmp))
            name = next(n for n in cache[key] if want in n)
            ins = cache[key][name]
            result["kernels"][label] = {"kernel": name.split("(")[0], "flops_per_eval": flops(ins) / per_lane,
                                        "valu": sum(1 for i in ins if i.startswith("v_")), "samples_per_lane": per_lane}
        m = load_model("panda_no_gripper")
        dw = m._dynamics_walk()
        table = m._ops_f(dw).detach().numpy()
        cw = m._get_walk(("chain", 8, "folded", dw.fold_key), targets=[8], folded=True, fold_key=dw.fold_key)
        both = table.copy()
        both[7:] = m._ops_f(cw).detach().numpy()[7:]
        sources = {"rnea": sp.arm_source(table, 7, False), "fk_rnea": sp.arm_source(both, 7, True), "dyn": sp.arm_dynamics_source(table, 7)}
        built = {}
        for label, which, want, per_lane in OWN:
            if which not in built:
                path = os.path.join(tmp, which + ".hip")
                open(path, "w").write(sources[which])
                built[which] = kernels_of(compile_to_asm(path, NOSLP + sp.ARM_FLAGS, tmp))
            ins = built[which][want]
            result["kernels"][label] = {"kernel": want, "flops_per_eval": flops(ins) / per_lane,
                                        "valu": sum(1 for i in ins if i.startswith("v_")), "samples_per_lane": per_lane}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_flops.json")
    with open(out, "w") as f:
        json.dump(result, f, indent=1)
    for k, v in result["kernels"].items():
        print("%-44s %7.0f flop / eval   %5d VALU per wavefront" % (k, v["flops_per_eval"], v["valu"]))


if __name__ == "__main__":
    main()
