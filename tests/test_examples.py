"""The examples/ scripts (the reference's three learning workloads on the MI355X path) run and learn."""
import importlib.util
import os

import pytest

EX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples")


def load(name):
    import sys
    if EX not in sys.path:
        sys.path.insert(0, EX)
    spec = importlib.util.spec_from_file_location(name, os.path.join(EX, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_examples_import_without_a_gpu():
    for name in ("learn_kinematics_of_iiwa", "learn_dynamics_iiwa", "learn_forward_dynamics_iiwa"):
        assert callable(load(name).run)


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph,fused_loss", [(False, False), (True, False), (True, True), (False, True)])
def test_learn_kinematics_example(use_graph, fused_loss):
    """The reference's loop literally (eager / as a hipGraph) and the example's default (fk_mse_loss under a hipGraph)."""
    hist = load("learn_kinematics_of_iiwa").run(batch=2048, epochs=120, use_graph=use_graph, fused_loss=fused_loss, verbose=False)
    assert hist[-1] < 0.2 * hist[0]


@pytest.mark.gpu
@pytest.mark.parametrize("spd,use_graph,fused_adam", [(False, False, False), (True, True, False), (False, True, True)])
def test_learn_dynamics_example(spd, use_graph, fused_adam):
    hist = load("learn_dynamics_iiwa").run(batch=1024, epochs=150, lr=3e-2, spd=spd, use_graph=use_graph, verbose=False, fused_adam=fused_adam)
    assert hist[-1] < 0.5 * hist[0]


@pytest.mark.gpu
def test_learn_forward_dynamics_example():
    hist = load("learn_forward_dynamics_iiwa").run(batch=1024, epochs=120, verbose=False)
    assert hist[-1] < 0.3 * hist[0]


def test_examples_learn_on_the_cpu_device_too(cpu_library):
    """The reference's examples run on the CPU (its default device); so do these, through libdrm_cpu.so: the same three learning
    loops, smaller, and the losses fall."""
    hist = load("learn_kinematics_of_iiwa").run(batch=1024, epochs=60, device="cpu", verbose=False)
    assert hist[-1] < 0.2 * hist[0]
    hist = load("learn_dynamics_iiwa").run(batch=256, epochs=60, lr=3e-2, device="cpu", verbose=False)
    assert hist[-1] < 0.8 * hist[0]
    hist = load("learn_forward_dynamics_iiwa").run(batch=256, epochs=60, device="cpu", verbose=False)
    assert hist[-1] < 0.3 * hist[0]
