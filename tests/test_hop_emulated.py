"""The device source of the plane-per-lane Q4/Q3 kernel (adaflo_amd/csrc/ns_hop_kernel.hpp), compiled for the HOST under
the lane emulator of tests/emu/ and compared with the oracle: index logic, ownership rules, the wave-private LDS
exchanges and the cross-lane moves, without a GPU.  Not a product path (adaflo_amd never loads the emulator library);
the `-m gpu` parity tests run the same source as a gfx950 code object through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import Case, rel_l2
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "hop_emu.cpp")
DEPS = [SRC, os.path.join(HERE, "emu", "hip_emu.hpp")] + [
    os.path.join(HERE, "..", "adaflo_amd", "csrc", f) for f in ("ns_hop_kernel.hpp", "ns_hox_kernel.hpp", "basis.hpp")]
LIB = os.path.join(HERE, "emu", "_hop_emu.so")


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    lib.hop_emu_vmult.restype = C.c_int
    return lib


def face_bits(faces, ncomp):
    m = 0
    for f in faces:
        for d in range(ncomp):
            m |= 1 << (ncomp * f + d)
    return m


def run_emulated(lib, case, op=0, lx=0, iface=0, phased=0):
    prm = case.prm
    src_u, src_p = case.random_u(), case.random_p()
    lin = case.random_lin()                                  # canonical [cell][q][12]
    lin_generic = np.ascontiguousarray(lin.reshape(case.n_cells, case.nq, 12).transpose(0, 2, 1))
    stokes = prm.physical_type == 2
    lin_mode = 2 if (stokes or prm.linearization == 3) else (0 if prm.linearization == 0 else 1)
    gamma = prm.weight if prm.physical_type == 0 else 0.0
    coef = np.array([0.0 if stokes else gamma * prm.density - prm.damping, 0.0 if stokes else prm.tau1 * prm.density,
                     prm.beta, prm.tau_grad_div, prm.viscosity * prm.tau1])
    integrate_p = 0 if prm.linearization == 4 else 1
    dst_u = np.full(case.n_u, np.nan)
    dst_p = np.full(case.n_p, np.nan)
    ncell = (C.c_int * 3)(*case.ncell)
    h = (C.c_double * 3)(*[case.mesh.h[d] for d in range(3)])
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    rc = lib.hop_emu_vmult(ncell, h, op, lin_mode, integrate_p, dp(coef), face_bits(case.faces_u, 3),
                           face_bits(case.faces_p, 1), dp(lin_generic), dp(src_u), dp(src_p), dp(dst_u), dp(dst_p),
                           lx if lx else case.ncell[0], iface, phased)
    assert rc == 0
    if op == 0:
        ref_u, ref_p = orc.ns_vmult(case.mesh, 4, prm, src_u, src_p, case.con_u, case.con_p, lin=lin)
        if not integrate_p:
            dst_p = ref_p          # projection scheme: the host prepares dst_p, the kernel does not touch it
        return rel_l2(dst_u, ref_u), rel_l2(dst_p, ref_p)
    ref_u = orc.ns_velocity_vmult(case.mesh, 4, prm, src_u, case.con_u, lin=lin)
    return rel_l2(dst_u, ref_u), 0.0


TOL = 1e-12


@pytest.mark.parametrize("ncell,lx", [((2, 2, 2), 0), ((3, 2, 4), 0), ((3, 5, 3), 2), ((1, 1, 1), 0)])
def test_emulated_vmult_newton(emu, ncell, lx):
    eu, ep = run_emulated(emu, Case(ncell, k=4), lx=lx)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("lin,phys", [(1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (0, 2)])
def test_emulated_vmult_modes(emu, lin, phys):
    eu, ep = run_emulated(emu, Case((2, 3, 3), k=4, linearization=lin, physical_type=phys, tau_grad_div=0.3,
                                    damping=0.2), lx=1)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_emulated_partial_constraints_and_velocity_block(emu):
    case = Case((3, 3, 3), k=4, faces_u=[0, 3, 4], faces_p=[1, 2])
    eu, ep = run_emulated(emu, case, lx=2)
    assert eu < TOL and ep < TOL, (eu, ep)
    eu, _ = run_emulated(emu, Case((3, 3, 2), k=4, faces_u=[1, 2, 5]), op=2)
    assert eu < TOL, eu


@pytest.mark.parametrize("iface", [0b000011, 0b110100, 0b111111])
def test_emulated_phased_schedule(emu, iface):
    eu, ep = run_emulated(emu, Case((4, 5, 6), k=4), lx=2, iface=iface, phased=1)
    assert eu < TOL and ep < TOL, (eu, ep)
