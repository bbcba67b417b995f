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
