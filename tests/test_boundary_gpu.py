"""The drop-in boundary beyond the operator calls: vectors in the APPLICATION's numbering (deal.II: its own DoF
numbering, locally owned entries first, ghosts appended) pass through a device-resident index map
(adaflo_vector_gather / adaflo_vector_scatter, include/adaflo_hip.h), and the C ABI is driven from C++ the way
source/navier_stokes.cc:593-631 drives NavierStokesMatrix (tests/capi_cpp/)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from adaflo_amd import _lib
from common import Case, rel_l2
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vector_gather_scatter_round_trip_and_vmult_through_the_map():
    case = Case((5, 4, 3), k=2)
    rng = np.random.default_rng(7)
    src_u, src_p, lin = case.random_u(), case.random_p(), case.random_lin()
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.ns_vmult(case.mesh, 2, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin,
                                weights=w, modes=modes)
    op = case.engine()
    op.set_linearization(lin)
    ctx, lib = op._require(), _lib.load()
    nu = case.n_u
    # the application's velocity vector: a random renumbering, the last fifth of the positions play the ghost
    # range (owned-then-ghost is just "one contiguous array" to the map), 17 extra slots nobody maps to
    n_ext = nu + 17
    pos = rng.permutation(n_ext)[:nu].astype(np.int64)            # position of engine DoF i in the application's array
    ext_src = np.full(n_ext, 123.0)
    ext_src[pos] = src_u
    dev = torch.device("cuda", 0)
    d_map = torch.from_numpy(pos).to(dev)
    d_ext_src = torch.from_numpy(ext_src).to(dev)
    d_eng_src = torch.empty(nu, dtype=torch.float64, device=dev)
    _lib.check(ctx, lib.adaflo_vector_gather(ctx, d_eng_src.data_ptr(), d_ext_src.data_ptr(), d_map.data_ptr(), nu))
    op.synchronize()
    assert np.array_equal(d_eng_src.cpu().numpy(), src_u)
    # vmult on the gathered vector, result scattered into the application's dst (copy, then add)
    d_eng_dst = torch.empty(nu, dtype=torch.float64, device=dev)
    d_sp = torch.from_numpy(src_p).to(dev)
    d_dp = torch.empty(case.n_p, dtype=torch.float64, device=dev)
    _lib.check(ctx, lib.adaflo_ns_vmult(ctx, d_eng_dst.data_ptr(), d_dp.data_ptr(), d_eng_src.data_ptr(), d_sp.data_ptr()))
    d_ext_dst = torch.full((n_ext,), -7.0, dtype=torch.float64, device=dev)
    _lib.check(ctx, lib.adaflo_vector_scatter(ctx, d_ext_dst.data_ptr(), d_eng_dst.data_ptr(), d_map.data_ptr(), nu, 0))
    op.synchronize()
    got = d_ext_dst.cpu().numpy()
    assert rel_l2(got[pos], ref_u) < TOL and rel_l2(d_dp.cpu().numpy(), ref_p) < TOL
    untouched = np.setdiff1d(np.arange(n_ext), pos)
    assert np.all(got[untouched] == -7.0)
    _lib.check(ctx, lib.adaflo_vector_scatter(ctx, d_ext_dst.data_ptr(), d_eng_dst.data_ptr(), d_map.data_ptr(), nu, 1))
    op.synchronize()
    assert np.array_equal(d_ext_dst.cpu().numpy()[pos], 2.0 * got[pos])
    # DoFs without a counterpart (-1): read as zero, never written
    pos2 = pos.copy()
    holes = rng.choice(nu, 40, replace=False)
    pos2[holes] = -1
    d_map2 = torch.from_numpy(pos2).to(dev)
    _lib.check(ctx, lib.adaflo_vector_gather(ctx, d_eng_src.data_ptr(), d_ext_src.data_ptr(), d_map2.data_ptr(), nu))
    d_ext_dst.fill_(5.0)
    _lib.check(ctx, lib.adaflo_vector_scatter(ctx, d_ext_dst.data_ptr(), d_eng_dst.data_ptr(), d_map2.data_ptr(), nu, 0))
    op.synchronize()
    g = d_eng_src.cpu().numpy()
    assert np.all(g[holes] == 0.0) and np.array_equal(np.delete(g, holes), np.delete(src_u, holes))
    assert np.all(d_ext_dst.cpu().numpy()[pos[holes]] == 5.0)
    assert lib.adaflo_vector_gather(ctx, d_eng_src.data_ptr(), d_ext_src.data_ptr(), None, nu) == _lib.ADAFLO_EINVAL


def test_cpp_translation_unit_drives_the_c_abi():
    """tests/capi_cpp/drop_in.cpp: a C++17 class with `void vmult(BlockVector &, const BlockVector &) const` over the
    C ABI and a small templated FGMRES with the engine's block preconditioner as right preconditioner -- the
    duck-typed seam of source/navier_stokes.cc:593-631 met from C++.  The program solves J du = rhs on a small
    Beltrami-type case and compares with adaflo_ns_solve_system (same algorithm inside the library)."""
    src = os.path.join(ROOT, "tests", "capi_cpp", "drop_in.cpp")
    exe = os.path.join(ROOT, "tests", "capi_cpp", "drop_in")
    libdir = os.path.join(ROOT, "adaflo_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-ladaflo_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "drop_in: OK" in out.stdout, out.stdout


def _write_case(path, arrays):
    """[int32 name length][name][int64 n][n doubles] per entry: what tests/capi_cpp/operators.cpp reads"""
    import struct
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
            b = name.encode()
            f.write(struct.pack("<i", len(b)) + b + struct.pack("<q", a.size) + a.tobytes())


def test_cpp_header_every_operator_against_the_ctypes_path(tmp_path):
    """include/adaflo_hip.hpp from a host-only C++17 translation unit (tests/capi_cpp/operators.cpp): NavierStokesMatrix
    with all its operator methods, residual, fix_linearization_point and get_matvec_statistics, the block
    preconditioner, and the four level-set operator structs (block vectors as separate allocations and as views into one
    array) -- compared in C++ at 1e-13 with the values THIS test produces through ctypes for the same inputs.  The ctypes
    values of vmult are held against the oracle here as well, so the chain oracle -> ctypes -> C++ is closed."""
    lib = _lib.load()
    rng = np.random.default_rng(2026)
    arrays = {}
    # ---- Navier-Stokes: 4 x 3 x 5 cells Q2/Q1, Dirichlet on five faces, BDF-2, Newton ---------------------------
    case = Case((4, 3, 5), k=2, faces_u=[0, 1, 2, 3, 4], faces_p=[], tau_grad_div=0.15, viscosity=0.3, damping=0.1)
    op = case.engine()
    ctx = op._require()
    d = op._desc
    arrays["ns_desc"] = list(d.ncell) + list(d.h) + list(d.origin) + [d.velocity_degree, d.ls_degree, d.velocity_constrained,
                                                                        d.pressure_constrained, d.ls_constrained,
                                                                        d.pressure_average_fix]
    # (what NavierStokesMatrix.update_parameters pushed: read back from the Python objects the same way)
    from adaflo_amd.parameters import LINEARIZATIONS, PHYSICAL_TYPES
    p_, ts = case.fp, case.ts
    arrays["ns_params"] = [PHYSICAL_TYPES[p_.physical_type], LINEARIZATIONS[p_.linearization], p_.beta, p_.tau_grad_div,
                           p_.density, p_.viscosity, p_.stored_damping, p_.density_diff, ts.weight(), ts.weight_old(),
                           ts.weight_old_old(), ts.tau1(), ts.factor_extrapol_old, ts.factor_extrapol_old_old]
    src_u, src_p, lin = case.random_u(), case.random_p(), case.random_lin()
    old_u, old_old_u, user_u, user_p = case.random_u(), case.random_u(), case.random_u(), case.random_p()
    arrays.update(src_u=src_u, src_p=src_p, lin=lin, old_u=old_u, old_old_u=old_old_u, user_u=user_u, user_p=user_p)
    op.set_linearization(lin)
    src, dst = op.block_vector(src_u, src_p), op.block_vector()
    op.vmult(dst, src)
    arrays["vmult_u"], arrays["vmult_p"] = dst.numpy()
    w, modes = case.weights_modes()
    ref_u, ref_p = orc.ns_vmult(case.mesh, 2, case.prm, src_u, src_p, case.con_u, case.con_p, lin=lin, weights=w, modes=modes)
    assert rel_l2(arrays["vmult_u"], ref_u) < TOL and rel_l2(arrays["vmult_p"], ref_p) < TOL
    op.fix_linearization_point()
    du = op.initialize_u_vector()
    op.velocity_vmult(du, src.block(0))
    arrays["velocity_vmult"] = du.numpy()
    base = op.initialize_p_vector(src_p)
    op.divergence_vmult_add(base, src.block(0), False)
    arrays["divergence"] = base.numpy()
    op.divergence_vmult_add(base, src.block(0), True)
    arrays["divergence_weighted"] = base.numpy()
    dp = op.initialize_p_vector()
    for name, fn in (("pressure_poisson", op.pressure_poisson_vmult), ("pressure_mass", op.pressure_mass_vmult),
                     ("pressure_convdiff", op.pressure_convdiff_vmult)):
        fn(dp, src.block(1))
        arrays[name] = dp.numpy()
    v = op.initialize_p_vector(src_p)
    op.apply_pressure_average_projection(v)
    arrays["projection"] = v.numpy()
    # (system_rhs is read-modify-written, as in the reference: the cell loop ADDS into it, :266-293)
    arrays["residual_in_u"], arrays["residual_in_p"] = case.random_u(), case.random_p()
    rhs = op.block_vector(arrays["residual_in_u"], arrays["residual_in_p"])
    import adaflo_amd
    op.residual(rhs, src, op.block_vector(user_u, user_p), adaflo_amd.BlockVector([op.initialize_u_vector(old_u)]),
                adaflo_amd.BlockVector([op.initialize_u_vector(old_old_u)]))
    arrays["residual_u"], arrays["residual_p"] = rhs.numpy()
    op.vmult(dst, src)
    arrays["vmult2_u"], arrays["vmult2_p"] = dst.numpy()
    _lib.check(ctx, lib.adaflo_ns_preconditioner_setup(ctx))
    _lib.check(ctx, lib.adaflo_ns_preconditioner_vmult(ctx, dst.block(0).ptr, dst.block(1).ptr, src.block(0).ptr, src.block(1).ptr))
    arrays["prec_u"], arrays["prec_p"] = dst.numpy()
    # ---- level set: 3 x 4 x 2 cells, s = 2 ------------------------------------------------------------------------
    from adaflo_amd import BrickMesh
    from adaflo_amd.level_set_okz import LevelSetOperators
    lmesh = BrickMesh([3, 4, 2], [0., 0., 0.], [1., 1., 0.8])
    ls = LevelSetOperators(lmesh, 2, velocity_degree=2, constrained_faces=(1,))
    lctx = ls._ctx
    arrays["ls_desc"] = list(lmesh.ncell) + list(lmesh.h) + list(lmesh.lower) + [2, 2, 0, 0, 1 << 1, 0]
    lp = [0.08, 0.1, 0.02, 1.5 / 0.02, -2.0 / 0.02, 0.5 / 0.02, 1.5]
    arrays["ls_params"] = lp
    _lib.check(lctx, lib.adaflo_ls_set_params(lctx, C.byref(_lib.LSParams(*lp))))
    nls, nq, ncl = ls.n_dofs, ls.n_q, ls.n_cells
    diag = ls.vector(rng.uniform(0.5, 1.5, nls))
    arrays["ls_diag"] = diag.numpy()
    ls.set_diagonal(diag)
    conv, nrm = rng.uniform(-1, 1, ncl * nq * 3), rng.uniform(-1, 1, ncl * nq * 3)
    arrays["ls_convection"], arrays["ls_normal_q"] = conv, nrm
    _lib.check(lctx, lib.adaflo_ls_set_evaluated_convection(lctx, conv.ctypes.data, 0))
    _lib.check(lctx, lib.adaflo_ls_set_evaluated_normal(lctx, nrm.ctypes.data, 0))
    s_ls, d_ls = ls.vector(rng.uniform(-1, 1, nls)), ls.vector()
    arrays["ls_src"] = s_ls.numpy()
    _lib.check(lctx, lib.adaflo_ls_advance_concentration_vmult(lctx, d_ls.ptr, s_ls.ptr))
    arrays["ls_advance"] = d_ls.numpy()
    _lib.check(lctx, lib.adaflo_ls_reinitialization_vmult(lctx, d_ls.ptr, s_ls.ptr, 0))
    arrays["ls_reinit"] = d_ls.numpy()
    _lib.check(lctx, lib.adaflo_ls_reinitialization_vmult(lctx, d_ls.ptr, s_ls.ptr, 1))
    arrays["ls_reinit_diffuse"] = d_ls.numpy()
    _lib.check(lctx, lib.adaflo_ls_compute_curvature_vmult(lctx, d_ls.ptr, s_ls.ptr, 1))
    arrays["ls_curvature"] = d_ls.numpy()
    s3, d3 = ls.vector(rng.uniform(-1, 1, 3 * nls), blocks=3), ls.vector(blocks=3)
    arrays["ls_normal_src"] = s3.numpy()
    _lib.check(lctx, lib.adaflo_ls_compute_normal_vmult(lctx, d3.ptr, s3.ptr))
    arrays["ls_normal"] = d3.numpy()
    path = str(tmp_path / "operators_case.bin")
    _write_case(path, arrays)
    # ---- the same through include/adaflo_hip.hpp -----------------------------------------------------------------
    src_cpp = os.path.join(ROOT, "tests", "capi_cpp", "operators.cpp")
    exe = str(tmp_path / "operators")
    libdir = os.path.join(ROOT, "adaflo_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src_cpp,
                           "-o", exe, "-L", libdir, "-ladaflo_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "operators: OK" in out.stdout and "FAILED" not in out.stdout, out.stdout
