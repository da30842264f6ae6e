"""differentiable-robot-model_amd — MI355X-native batched FK / Jacobian / RNEA engine.

Drop-in for the hot path of facebookresearch/differentiable-robot-model
(reference ``differentiable_robot_model/__init__.py:7-12`` exports the same names).
Import it as ``differentiable_robot_model_amd`` (the repo-root shim of that name
maps the importable name onto this directory).
"""
from .robot_model import (  # noqa: F401
    DifferentiableRobotModel,
    DifferentiableKUKAiiwa,
    DifferentiableFrankaPanda,
    DifferentiableTwoLinkRobot,
    DifferentiableTrifingerEdu,
)

__version__ = "0.1.0"
