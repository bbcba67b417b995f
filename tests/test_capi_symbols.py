"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every
symbol include/adaflo_hip.h declares; the host mirror validates parameters like the reference."""
import ctypes
import os
import re

import pytest

import adaflo_amd
from adaflo_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "adaflo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(adaflo_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "missing export: " + n


def test_python_binding_covers_the_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    _lib.load()


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    op = adaflo_amd.NavierStokesMatrix(adaflo_amd.FlowParameters(), adaflo_amd.BrickMesh([2] * 3, [0] * 3, [1] * 3))
    ts = adaflo_amd.TimeStepping(adaflo_amd.FlowParameters())
    with pytest.raises(_lib.AdafloError):
        op.initialize(ts, True)
    with pytest.raises(_lib.AdafloError):
        op.n_dofs_u()  # ExcNotInitialized


def test_parameter_validation_mirrors_the_reference():
    with pytest.raises(NotImplementedError):
        adaflo_amd.FlowParameters(velocity_degree=1)            # parameters.cc:461-462
    with pytest.raises(ValueError):
        adaflo_amd.FlowParameters(physical_type="incompressible stationary",
                                  linearization="coupled implicit Picard")   # parameters.cc:501-504
    with pytest.raises(ValueError):
        adaflo_amd.FlowParameters(tau_grad_div=-1.0)
    p = adaflo_amd.FlowParameters(physical_type="stokes", density=3.0)
    assert p.density == 0.0                                     # parameters.cc:477-478
    assert adaflo_amd.FlowParameters(damping=2.0).stored_damping == -2.0   # parameters.cc:466-467


def test_time_stepping_bdf2_weights():
    """source/time_stepping.cc:123-200"""
    # (min step size 0: otherwise the reference's rule "min > start => max = min = start",
    # parameters.cc:593-595, pins the step size and the set_time_step below is clamped by next())
    p = adaflo_amd.FlowParameters(time_step_size_start=0.05, end_time=1.0, time_step_size_min=0.0)
    ts = adaflo_amd.TimeStepping(p)
    ts.next()
    assert (ts.weight(), ts.weight_old(), ts.weight_old_old()) == (20.0, -20.0, 0.0)
    assert (ts.factor_extrapol_old, ts.factor_extrapol_old_old) == (1.0, 0.0)
    ts.next()
    assert abs(ts.weight() - 30.0) < 1e-12 and abs(ts.weight_old() + 40.0) < 1e-12
    assert abs(ts.weight_old_old() - 10.0) < 1e-12
    assert (ts.factor_extrapol_old, ts.factor_extrapol_old_old) == (1.0, 0.0)   # not in 2nd step
    ts.next()
    assert abs(ts.factor_extrapol_old - 2.0) < 1e-12 and abs(ts.factor_extrapol_old_old + 1.0) < 1e-12
    ts.set_time_step(0.1)
    ts.next()   # the reference copies the ALREADY modified step into last_step_val (:131-134)
    c, l = 0.1, 0.1
    assert abs(ts.weight() - (2 * c + l) / (c * (c + l))) < 1e-12
    # reference defaults (parameters.cc:377-410): 0.01 / max 1 / min 0.1 => constant steps of 0.01
    q = adaflo_amd.FlowParameters()
    assert (q.time_step_size_start, q.time_step_size_max, q.time_step_size_min) == (0.01, 0.01, 0.01)
    ts = adaflo_amd.TimeStepping(adaflo_amd.FlowParameters(time_step_size_start=0.05))
    ts.next()
    ts.set_time_step(0.1)
    ts.next()
    assert abs(ts.step_size() - 0.05) < 1e-15


def test_cpp_translation_unit_compiles_and_links():
    """The shipped C++ host side include/adaflo_hip.hpp (NavierStokesMatrix, the block preconditioner and the four
    level-set operator structs with the reference's method names, templated on the vector types) and its two users under
    tests/capi_cpp/ -- drop_in.cpp (the duck-typed vmult seam of source/navier_stokes.cc:593-631 driven by a templated
    FGMRES) and operators.cpp (every method, compared with the ctypes path) -- are plain host C++17: they compile with g++
    -Wall -Werror against include/ alone and link against the shared library; they RUN in tests/test_boundary_gpu.py"""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "adaflo_amd", "lib")
    with tempfile.TemporaryDirectory() as tmp:
        for unit in ("drop_in", "operators"):
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                                   os.path.join(root, "tests", "capi_cpp", unit + ".cpp"), "-o", os.path.join(tmp, unit),
                                   "-L", libdir, "-ladaflo_hip", "-Wl,-rpath," + libdir])
