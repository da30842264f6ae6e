"""Import shim: ``import differentiable_robot_model_amd`` -> the package in ./differentiable-robot-model_amd/.

The package directory name contains hyphens (it is the project name), which Python
cannot import directly; this module replaces itself in ``sys.modules`` with the real
package, so ``from differentiable_robot_model_amd.robot_model import ...`` works.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "differentiable-robot-model_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _module
_spec.loader.exec_module(_module)
