"""The device source of the x-marching Q3..Q5 kernel (adaflo_amd/csrc/ns_hox_kernel.hpp), compiled for the HOST under
the lane emulator of tests/emu/ and compared with the oracle.  This is a test of index logic, ownership rules and
LDS hand-offs that runs without a GPU; it is not a product path (adaflo_amd never loads the emulator library), the
`-m gpu` parity tests run the same source as a gfx950 code object through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import Case, rel_l2
from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "hox_emu.cpp")
DEPS = [SRC, os.path.join(HERE, "emu", "hip_emu.hpp"),
        os.path.join(HERE, "..", "adaflo_amd", "csrc", "ns_hox_kernel.hpp"),
        os.path.join(HERE, "..", "adaflo_amd", "csrc", "basis.hpp")]
LIB = os.path.join(HERE, "emu", "_hox_emu.so")


@pytest.fixture(scope="module")
def emu():
    defines = os.environ.get("HOX_EMU_DEFINES", "").split()        # e.g. -DHOX_FLAGS=1 (development: another build)
    lib_path = LIB if not defines else LIB.replace(".so", "_" + "".join(c for c in "".join(defines) if c.isalnum()) + ".so")
    if not os.path.exists(lib_path) or any(os.path.getmtime(d) > os.path.getmtime(lib_path) for d in DEPS):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC"] + defines + ["-o", lib_path, SRC])
    lib = C.CDLL(lib_path)
    lib.hox_emu_vmult.restype = C.c_int
    return lib


def face_bits(faces, ncomp):
    m = 0
    for f in faces:
        for d in range(ncomp):
            m |= 1 << (ncomp * f + d)
    return m


def run_emulated(lib, case, op=0, lx=0, iface=0, phased=0, seed_src=True, varco=False, recompute=False):
    k, prm = case.k, case.prm
    src_u, src_p = case.random_u(), case.random_p()
    lin = case.random_lin()                                  # canonical [cell][q][12]
    nq = case.nq
    lin_nodal = None
    if recompute:                                            # the state the oracle's residual leaves at a nodal field
        lin_nodal = case.random_u()
        lin = np.zeros(case.n_cells * nq * 12)
        orc.ns_residual(case.mesh, k, prm, lin_nodal, case.random_p(), case.random_u(), case.random_u(), con_u=case.con_u,
                        con_p=case.con_p, lin=lin)
    lin_generic = np.ascontiguousarray(lin.reshape(case.n_cells, nq, 12).transpose(0, 2, 1))
    stokes = prm.physical_type == 2
    lin_mode = 2 if (stokes or prm.linearization == 3) else (0 if prm.linearization == 0 else 1)
    gamma = prm.weight if prm.physical_type == 0 else 0.0
    # the oracle's NSParams carries damping with the sign flipped (parameters.cc:466-467), as the engine's NSDev
    coef = np.array([0.0 if stokes else gamma * prm.density - prm.damping, 0.0 if stokes else prm.tau1 * prm.density,
                     prm.beta, prm.tau_grad_div, prm.viscosity * prm.tau1,
                     0.0 if stokes else gamma, 0.0 if stokes else prm.tau1, 0.0 if stokes else 1.0, prm.tau1])
    co = {}
    if varco:                                                # variable density / viscosity / damping per quadrature point
        rho, mu, damp = case.random_coefficients()
        co = dict(rho=rho, mu=mu, damp=damp)
    integrate_p = 0 if prm.linearization == 4 else 1
    dst_u = np.full(case.n_u, np.nan)
    dst_p = np.full(case.n_p, np.nan)
    ncell = (C.c_int * 3)(*case.ncell)
    h = (C.c_double * 3)(*[case.mesh.h[d] for d in range(3)])
    dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
    if recompute:
        lib.hox_emu_vmult_recompute.restype = C.c_int
        rc = lib.hox_emu_vmult_recompute(k, ncell, h, op, lin_mode, integrate_p, dp(coef), face_bits(case.faces_u, 3),
                                         face_bits(case.faces_p, 1), dp(lin_nodal), dp(src_u), dp(src_p), dp(dst_u),
                                         dp(dst_p), lx if lx else case.ncell[0], iface, phased)
    else:
        rc = lib.hox_emu_vmult(k, ncell, h, op, lin_mode, integrate_p, dp(coef), face_bits(case.faces_u, 3),
                               face_bits(case.faces_p, 1), dp(lin_generic), dp(src_u), dp(src_p), dp(dst_u), dp(dst_p),
                               lx if lx else case.ncell[0], iface, phased, dp(co.get("rho")), dp(co.get("mu")), dp(co.get("damp")))
    assert rc == 0
    if op == 0:
        ref_u, ref_p = orc.ns_vmult(case.mesh, k, prm, src_u, src_p, case.con_u, case.con_p, lin=lin, **co)
        if not integrate_p:
            dst_p = ref_p          # projection scheme: the host prepares dst_p, the kernel does not touch it
        return rel_l2(dst_u, ref_u), rel_l2(dst_p, ref_p)
    ref_u = orc.ns_velocity_vmult(case.mesh, k, prm, src_u, case.con_u, lin=lin, **co)
    return rel_l2(dst_u, ref_u), 0.0


TOL = 1e-12


@pytest.mark.parametrize("k,ncell,lx", [(4, (3, 2, 4), 0), (4, (3, 5, 9), 2), (3, (3, 5, 5), 0), (5, (2, 3, 3), 1)])
def test_emulated_vmult_newton(emu, k, ncell, lx):
    eu, ep = run_emulated(emu, Case(ncell, k=k), lx=lx)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("lin,phys", [(1, 0), (2, 0), (3, 0), (4, 0), (0, 1), (0, 2)])
def test_emulated_vmult_modes(emu, lin, phys):
    eu, ep = run_emulated(emu, Case((2, 3, 5), k=4, linearization=lin, physical_type=phys, tau_grad_div=0.3,
                                    damping=0.2), lx=1)
    assert eu < TOL and ep < TOL, (eu, ep)


def test_emulated_partial_constraints_and_velocity_block(emu):
    case = Case((3, 3, 5), k=4, faces_u=[0, 3, 4], faces_p=[1, 2])
    eu, ep = run_emulated(emu, case, lx=2)
    assert eu < TOL and ep < TOL, (eu, ep)
    eu, _ = run_emulated(emu, Case((3, 3, 5), k=4, faces_u=[1, 2, 5]), op=2)
    assert eu < TOL, eu


@pytest.mark.parametrize("iface", [0b000011, 0b110100, 0b111111])
def test_emulated_phased_schedule(emu, iface):
    eu, ep = run_emulated(emu, Case((6, 5, 9), k=4), lx=2, iface=iface, phased=1)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("k,ncell,lx,lin,phys", [(4, (3, 2, 5), 2, 0, 0), (4, (2, 3, 4), 0, 1, 0), (3, (3, 5, 5), 2, 0, 0),
                                                  (5, (2, 3, 2), 1, 0, 0), (4, (3, 3, 2), 0, 0, 1), (4, (2, 2, 5), 1, 0, 2)])
def test_emulated_residual_mode(emu, k, ncell, lx, lin, phys):
    """residual mode of the same kernel (template RES): cell-loop sums against the oracle's residual (which returns
    -sums), the state it stores in the streaming layout (converted back), BDF-2 history, partial constraints whose
    boundary values are read plainly, stationary and Stokes equations, Picard-type state"""
    case = Case(ncell, k=k, faces_u=[0, 3, 4, 5], faces_p=[1], linearization=lin, physical_type=phys, tau_grad_div=0.3,
                damping=0.2, density=1.3, steps=3)
    prm = case.prm
    src_u, src_p, old_u, oo_u = case.random_u(), case.random_p(), case.random_u(), case.random_u()
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, k, prm, src_u, src_p, old_u, oo_u, con_u=case.con_u, con_p=case.con_p,
                                   lin=lin_ref)
    stokes = prm.physical_type == 2
    lin_mode = 2 if stokes else (0 if prm.linearization == 0 else 1)
    gamma = prm.weight if prm.physical_type == 0 else 0.0
    coef = np.array([0.0 if stokes else gamma * prm.density - prm.damping, 0.0 if stokes else prm.tau1 * prm.density,
                     prm.beta, prm.tau_grad_div, prm.viscosity * prm.tau1])
    old_comb = prm.weight_old * old_u + prm.weight_old_old * oo_u if prm.physical_type == 0 else None
    sum_u, sum_p = np.full(case.n_u, np.nan), np.full(case.n_p, np.nan)
    lin_generic = np.zeros(case.n_cells * 12 * case.nq)
    dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
    emu.hox_emu_residual.restype = C.c_int
    rc = emu.hox_emu_residual(k, (C.c_int * 3)(*case.ncell), (C.c_double * 3)(*[case.mesh.h[d] for d in range(3)]), lin_mode,
                              dp(coef), C.c_double(prm.density if prm.physical_type == 0 else 0.0),
                              face_bits(case.faces_u, 3), face_bits(case.faces_p, 1), dp(src_u), dp(src_p), dp(old_comb),
                              dp(sum_u), dp(sum_p), dp(lin_generic), lx if lx else case.ncell[0])
    assert rc == 0
    assert rel_l2(-sum_u, ref_u) < TOL and rel_l2(-sum_p, ref_p) < TOL
    if lin_mode != 2:
        got = lin_generic.reshape(case.n_cells, 12, case.nq).transpose(0, 2, 1)
        ncomp = 12 if lin_mode == 0 else 4
        assert rel_l2(got[:, :, :ncomp], lin_ref.reshape(case.n_cells, case.nq, 12)[:, :, :ncomp]) < TOL


@pytest.mark.parametrize("k,ncell,lin,phys,op", [(4, (3, 2, 5), 0, 0, 0), (3, (3, 5, 3), 1, 0, 0), (5, (2, 3, 2), 0, 0, 0),
                                                 (4, (2, 3, 4), 0, 2, 0), (4, (3, 3, 3), 0, 0, 2), (3, (4, 3, 5), 2, 0, 0)])
def test_emulated_variable_coefficients(emu, k, ncell, lin, phys, op):
    """two-phase Jacobian (template VARCO): density, viscosity and damping per quadrature point travel as two more
    pieces of the state stream; Newton, Picard-type, Stokes, velocity block"""
    case = Case(ncell, k=k, faces_u=[0, 3, 4], faces_p=[1], linearization=lin, physical_type=phys, tau_grad_div=0.3,
                density_diff=-0.5)
    eu, ep = run_emulated(emu, case, op=op, lx=2, varco=True)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("k,ncell,lx,lin,op,phased", [(4, (3, 2, 5), 2, 0, 0, 0), (4, (2, 3, 4), 0, 1, 0, 0), (3, (3, 5, 5), 2, 0, 0, 0),
                                                      (5, (2, 3, 2), 1, 0, 0, 0), (4, (3, 3, 3), 0, 0, 2, 0), (4, (4, 3, 5), 2, 0, 0, 1)])
def test_emulated_recompute_state_mode(emu, k, ncell, lx, lin, op, phased):
    """recompute-state mode (template RCP): the kernel interpolates (u_lin, grad u_lin) from the nodal field the
    residual was evaluated at (boundary values included) instead of streaming them; reference: the oracle's vmult on
    the state the oracle's residual stored at that field.  Newton, Picard-type, velocity block, phased schedule"""
    case = Case(ncell, k=k, faces_u=[0, 3, 4, 5], faces_p=[1], linearization=lin, tau_grad_div=0.3, damping=0.2, steps=3)
    eu, ep = run_emulated(emu, case, op=op, lx=lx, recompute=True, phased=phased, iface=0b010011 if phased else 0)
    assert eu < TOL and ep < TOL, (eu, ep)


@pytest.mark.parametrize("k,ncell,lx,lin", [(4, (3, 2, 5), 2, 2), (4, (2, 3, 4), 0, 3), (3, (3, 5, 5), 2, 2), (5, (2, 3, 2), 1, 2),
                                            (3, (4, 3, 3), 0, 3)])
def test_emulated_residual_of_the_extrapolating_schemes(emu, k, ncell, lx, lin):
    """residual mode with template EXT: the semi-implicit (2) and explicit (3) treatments of convection linearise about
    extrap_old u_old + extrap_old_old u_old_old (navier_stokes_matrix.cc:644-647, 740-782); the kernel evaluates value and
    gradient of that nodal combination as one more field; sums and the stored state (u_ext, div u_ext) against the oracle"""
    case = Case(ncell, k=k, faces_u=[0, 3, 4, 5], faces_p=[1], linearization=lin, tau_grad_div=0.3, damping=0.2,
                density=1.3, steps=3)
    prm = case.prm
    src_u, src_p, old_u, oo_u = case.random_u(), case.random_p(), case.random_u(), case.random_u()
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, k, prm, src_u, src_p, old_u, oo_u, con_u=case.con_u, con_p=case.con_p,
                                   lin=lin_ref)
    lin_mode = 1 if lin == 2 else 2
    coef = np.array([prm.weight * prm.density - prm.damping, prm.tau1 * prm.density, prm.beta, prm.tau_grad_div,
                     prm.viscosity * prm.tau1])
    old_comb = prm.weight_old * old_u + prm.weight_old_old * oo_u
    ext_comb = prm.extrap_old * old_u + prm.extrap_old_old * oo_u
    sum_u, sum_p = np.full(case.n_u, np.nan), np.full(case.n_p, np.nan)
    lin_generic = np.zeros(case.n_cells * 12 * case.nq)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    emu.hox_emu_residual_extrapolated.restype = C.c_int
    rc = emu.hox_emu_residual_extrapolated(k, (C.c_int * 3)(*case.ncell), (C.c_double * 3)(*[case.mesh.h[d] for d in range(3)]),
                                           lin_mode, dp(coef), C.c_double(prm.density), face_bits(case.faces_u, 3),
                                           face_bits(case.faces_p, 1), dp(src_u), dp(src_p), dp(old_comb), dp(ext_comb),
                                           dp(sum_u), dp(sum_p), dp(lin_generic), lx if lx else case.ncell[0])
    assert rc == 0
    assert rel_l2(-sum_u, ref_u) < TOL and rel_l2(-sum_p, ref_p) < TOL, (rel_l2(-sum_u, ref_u), rel_l2(-sum_p, ref_p))
    if lin_mode == 1:
        got = lin_generic.reshape(case.n_cells, 12, case.nq).transpose(0, 2, 1)
        assert rel_l2(got[:, :, :4], lin_ref.reshape(case.n_cells, case.nq, 12)[:, :, :4]) < TOL


@pytest.mark.parametrize("k,ncell,lx,lin,phys", [(4, (3, 2, 5), 2, 0, 0), (3, (3, 5, 5), 0, 1, 0), (5, (2, 3, 2), 1, 0, 0),
                                                  (4, (3, 3, 2), 0, 0, 1)])
def test_emulated_variable_coefficient_residual(emu, k, ncell, lx, lin, phys):
    """residual mode with variable coefficients (templates RES + VARCO, round 6; navier_stokes_matrix.cc:636-642, 717-732,
    827-845): density / viscosity / damping arrive as a two-piece stream of their own; sums and the stored state against
    the oracle's two-phase residual; the coefficient pieces ride along behind the state pieces (checked bitwise inside the
    emulator: what the two-phase vmult will stream)"""
    case = Case(ncell, k=k, faces_u=[0, 3, 4, 5], faces_p=[1], linearization=lin, physical_type=phys, tau_grad_div=0.3,
                damping=0.2, density=1.3, density_diff=-0.5, steps=3)
    prm = case.prm
    src_u, src_p, old_u, oo_u = case.random_u(), case.random_p(), case.random_u(), case.random_u()
    rho, mu, damp = case.random_coefficients()
    lin_ref = np.zeros(case.n_cells * case.nq * 12)
    ref_u, ref_p = orc.ns_residual(case.mesh, k, prm, src_u, src_p, old_u, oo_u, con_u=case.con_u, con_p=case.con_p,
                                   lin=lin_ref, rho=rho, mu=mu, damp=damp)
    lin_mode = 0 if prm.linearization == 0 else 1
    gamma = prm.weight if prm.physical_type == 0 else 0.0
    coef = np.array([gamma * prm.density - prm.damping, prm.tau1 * prm.density, prm.beta, prm.tau_grad_div,
                     prm.viscosity * prm.tau1, gamma, prm.tau1, 1.0, prm.tau1])
    old_comb = prm.weight_old * old_u + prm.weight_old_old * oo_u if prm.physical_type == 0 else None
    sum_u, sum_p = np.full(case.n_u, np.nan), np.full(case.n_p, np.nan)
    lin_generic = np.zeros(case.n_cells * 12 * case.nq)
    dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
    emu.hox_emu_residual_varco.restype = C.c_int
    rc = emu.hox_emu_residual_varco(k, (C.c_int * 3)(*case.ncell), (C.c_double * 3)(*[case.mesh.h[d] for d in range(3)]), lin_mode,
                                    dp(coef), C.c_double(1.0 if prm.physical_type == 0 else 0.0), face_bits(case.faces_u, 3),
                                    face_bits(case.faces_p, 1), dp(src_u), dp(src_p), dp(old_comb), dp(rho), dp(mu), dp(damp),
                                    dp(sum_u), dp(sum_p), dp(lin_generic), lx if lx else case.ncell[0])
    assert rc == 0, rc
    assert rel_l2(-sum_u, ref_u) < TOL and rel_l2(-sum_p, ref_p) < TOL, (rel_l2(-sum_u, ref_u), rel_l2(-sum_p, ref_p))
    got = lin_generic.reshape(case.n_cells, 12, case.nq).transpose(0, 2, 1)
    ncomp = 12 if lin_mode == 0 else 4
    assert rel_l2(got[:, :, :ncomp], lin_ref.reshape(case.n_cells, case.nq, 12)[:, :, :ncomp]) < TOL
