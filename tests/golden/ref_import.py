"""Locate and import the UNMODIFIED reference library (test infrastructure only).

Only ``tests/golden/make_golden.py`` and CPU-side (``-m "not gpu"``) parity
tests use this.  ``/root/reference`` does not exist on the GPU box, so nothing
that runs there may depend on it: callers must check ``reference_available()``.
"""
import os
import sys
import warnings

REFERENCE_ROOT = os.environ.get("DRM_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shim")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "differentiable_robot_model"))


def import_reference():
    """Returns the reference's ``differentiable_robot_model.robot_model`` module."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    try:
        import urdf_parser_py.urdf  # noqa: F401  (the real package, if someone installed it)
    except ImportError:
        if _SHIM not in sys.path:
            sys.path.insert(0, _SHIM)
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    warnings.filterwarnings("ignore", category=UserWarning)
    warnings.filterwarnings("ignore", category=DeprecationWarning)
    import differentiable_robot_model.robot_model as rm  # the reference, not ours
    return rm


def reference_data_dir():
    return os.path.join(REFERENCE_ROOT, "diff_robot_data")
