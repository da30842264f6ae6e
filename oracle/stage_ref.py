#!/usr/bin/env python3
"""TEST / BASELINE INFRASTRUCTURE ONLY — stages the UNMODIFIED reference into the git-ignored oracle/_ref/.

The reference (facebookresearch/differentiable-robot-model) is pure Python: there is nothing to compile.  What
"building oracle/_ref" means for it is copying, from where they lie under /root/reference, exactly the files its hot path
imports — the eight modules of ``differentiable_robot_model/`` and the URDF text files of ``diff_robot_data/`` (no
meshes) — into ``oracle/_ref/`` so that they travel to the GPU box with the gpurun snapshot (like the built .so files:
git-ignored, not gpurun-ignored) and ``bench.py``'s ``cpu_baseline`` leg can time the reference ITSELF on that box's
host cores in the same run as the GPU numbers.

Nothing of oracle/_ref/ enters git history and nothing of it is imported by the product package
(``differentiable-robot-model_amd/``); only ``oracle/ref_timing.py`` (a subprocess of bench.py's cpu_baseline leg) and
tests read it.

    python oracle/stage_ref.py            # no-op with exit 0 when /root/reference is absent (the GPU box)
"""
import filecmp
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("DRM_REFERENCE_ROOT", "/root/reference")
DEST = os.path.join(HERE, "_ref")
MODULES = ("__init__.py", "robot_model.py", "rigid_body.py", "rigid_body_params.py", "spatial_vector_algebra.py",
           "se3_so3_util.py", "urdf_utils.py", "utils.py")


def _copy(src, dst):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    if not (os.path.exists(dst) and filecmp.cmp(src, dst, shallow=False)):
        shutil.copyfile(src, dst)


def stage():
    pkg = os.path.join(REFERENCE_ROOT, "differentiable_robot_model")
    data = os.path.join(REFERENCE_ROOT, "diff_robot_data")
    if not os.path.isdir(pkg):
        print("stage_ref: %s absent, nothing staged (oracle/_ref/ %s)"
              % (REFERENCE_ROOT, "present from an earlier run" if os.path.isdir(DEST) else "absent"))
        return False
    for name in MODULES:
        _copy(os.path.join(pkg, name), os.path.join(DEST, "differentiable_robot_model", name))
    count = 0
    for dirpath, _dirs, files in os.walk(data):
        for name in files:
            if name.endswith(".urdf") or name == "__init__.py":
                src = os.path.join(dirpath, name)
                _copy(src, os.path.join(DEST, "diff_robot_data", os.path.relpath(src, data)))
                count += 1
    with open(os.path.join(DEST, "STAGED_FROM"), "w") as f:
        f.write("%s\nunmodified copies made by oracle/stage_ref.py; git-ignored; test / baseline infrastructure only\n"
                % REFERENCE_ROOT)
    print("stage_ref: %d modules + %d data files -> %s" % (len(MODULES), count, DEST))
    return True


if __name__ == "__main__":
    stage()
    sys.exit(0)
