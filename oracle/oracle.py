"""ctypes front-end of the CPU oracle (oracle/adaflo_oracle.c, adaflo_oracle_fast.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by adaflo_amd/ (the product path).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_adaflo_oracle.so")
_SRCS = [os.path.join(_HERE, f) for f in ("adaflo_oracle.c", "adaflo_oracle_fast.c", "adaflo_oracle_batched.c")]
_DEPS = _SRCS + [os.path.join(_HERE, "adaflo_oracle_batched_body.h")]
_HOST = _SO + ".host"


def _host_signature():
    """-march=native ties the library to the CPU it was built on: model name + the SIMD flags that matter"""
    model, flags = "?", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "?":
                model = line.split(":", 1)[1].strip()
            if line.startswith("flags") and not flags:
                have = set(line.split(":", 1)[1].split())
                flags = " ".join(f for f in ("avx2", "fma", "avx512f", "avx512dq", "avx512vl") if f in have)
    except OSError:
        pass
    return model + " | " + flags


def build(force=False):
    """Compile the oracle with gcc (plain C, OpenMP for the fast variant)."""
    srcs = [s for s in _SRCS if os.path.exists(s)]
    sig = _host_signature()
    try:
        built_on = open(_HOST).read()
    except OSError:
        built_on = None
    stale = force or not os.path.exists(_SO) or built_on != sig or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in _DEPS)
    if stale:
        cmd = ["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-Wall",
               "-o", _SO] + srcs + ["-lm"]
        subprocess.check_call(cmd)
        with open(_HOST, "w") as f:
            f.write(sig)
    return _SO


class Mesh(C.Structure):
    _fields_ = [("dim", C.c_int), ("ncell", C.c_int * 3), ("h", C.c_double * 3),
                ("origin", C.c_double * 3)]

    @staticmethod
    def make(ncell, lower, upper):
        dim = len(ncell)
        m = Mesh()
        m.dim = dim
        for d in range(3):
            m.ncell[d] = ncell[d] if d < dim else 1
            m.h[d] = (upper[d] - lower[d]) / ncell[d] if d < dim else 1.0
            m.origin[d] = lower[d] if d < dim else 0.0
        return m

    @property
    def n_cells(self):
        return int(np.prod([self.ncell[d] for d in range(self.dim)]))

    def n_nodes(self, degree):
        return int(np.prod([degree * self.ncell[d] + 1 for d in range(self.dim)]))

    def nodes_per_dim(self, degree):
        return [degree * self.ncell[d] + 1 for d in range(self.dim)]


class NSParams(C.Structure):
    _fields_ = [("physical_type", C.c_int), ("linearization", C.c_int),
                ("beta", C.c_double), ("tau_grad_div", C.c_double),
                ("density", C.c_double), ("viscosity", C.c_double),
                ("damping", C.c_double), ("density_diff", C.c_double),
                ("weight", C.c_double), ("weight_old", C.c_double),
                ("weight_old_old", C.c_double), ("tau1", C.c_double),
                ("extrap_old", C.c_double), ("extrap_old_old", C.c_double)]

    @staticmethod
    def make(**kw):
        p = NSParams()
        d = dict(physical_type=0, linearization=0, beta=0.5, tau_grad_div=0.0, density=1.0,
                 viscosity=1.0, damping=0.0, density_diff=0.0, weight=1.0, weight_old=-1.0,
                 weight_old_old=0.0, tau1=1.0, extrap_old=1.0, extrap_old_old=0.0)
        d.update(kw)
        for k, v in d.items():
            setattr(p, k, v)
        return p


class LSParams(C.Structure):
    _fields_ = [("ls_degree", C.c_int), ("epsilon_used", C.c_double),
                ("minimal_edge_length", C.c_double), ("time_step", C.c_double),
                ("weight", C.c_double), ("cell_diameter", C.c_double),
                ("epsilon", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_n_nodes.restype = C.c_long
        _lib.orc_fast_set_threads(usable_cores())      # OpenMP threads = CPU quota of the container
    return _lib


def _p(a, ctype=C.c_double):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(ctype))


def _u8(a):
    return _p(a, C.c_uint8)


# ----------------------------------------------------------------------------- 1D data
def gauss_legendre(n):
    x = np.zeros(n)
    w = np.zeros(n)
    lib().orc_gauss_legendre(n, _p(x), _p(w))
    return x, w


def gauss_lobatto(n):
    x = np.zeros(n)
    lib().orc_gauss_lobatto(n, _p(x))
    return x


def shape_1d(fe_type, degree, xq):
    xq = np.ascontiguousarray(xq, dtype=np.float64)
    S = np.zeros((len(xq), degree + 1))
    D = np.zeros((len(xq), degree + 1))
    lib().orc_shape_1d(fe_type, degree, len(xq), _p(xq), _p(S), _p(D))
    return S, D


def node_coordinates(mesh, degree, fe_type=0):
    xyz = np.zeros((mesh.n_nodes(degree), mesh.dim))
    lib().orc_node_coordinates(C.byref(mesh), fe_type, degree, _p(xyz))
    return xyz


def boundary_mask(mesh, degree, ncomp=1, faces=None, comps=None):
    """uint8 mask [n_nodes*ncomp] of DoFs on the listed boundary faces
    (face id = 2*d + side, deal.II hyper_rectangle convention; None = all)."""
    npd = mesh.nodes_per_dim(degree)
    idx = np.indices(npd[::-1])[::-1]  # idx[d] has shape (nz, ny, nx)
    m = np.zeros(npd[::-1], dtype=bool)
    for d in range(mesh.dim):
        for side in range(2):
            if faces is None or (2 * d + side) in faces:
                m |= idx[d] == (0 if side == 0 else npd[d] - 1)
    m = m.reshape(-1)
    out = np.zeros((m.size, ncomp), dtype=np.uint8)
    for c in range(ncomp):
        if comps is None or c in comps:
            out[:, c] = m
    return out.reshape(-1)


# ----------------------------------------------------------------------------- NS
def n_lin(dim):
    return dim + dim * dim


def ns_nq(k, dim):
    return (k + 1) ** dim


def ns_vmult(mesh, k, prm, src_u, src_p, con_u=None, con_p=None, lin=None, rho=None, mu=None,
             damp=None, weights=None, modes=None):
    dst_u = np.zeros_like(src_u)
    dst_p = np.zeros_like(src_p)
    lib().orc_ns_vmult(C.byref(mesh), k, C.byref(prm), _p(src_u), _p(src_p), _p(dst_u), _p(dst_p),
                       _u8(con_u), _u8(con_p), _p(lin), _p(rho), _p(mu), _p(damp), _p(weights),
                       _p(modes))
    return dst_u, dst_p


def ns_residual(mesh, k, prm, src_u, src_p, old_u, oldold_u, con_u=None, con_p=None, lin=None,
                rho=None, mu=None, damp=None, rhs_u=None, rhs_p=None, user_u=None, user_p=None):
    """returns (system_rhs_u, system_rhs_p); `lin` (if given) is overwritten."""
    rhs_u = np.zeros_like(src_u) if rhs_u is None else rhs_u.copy()
    rhs_p = np.zeros_like(src_p) if rhs_p is None else rhs_p.copy()
    lib().orc_ns_residual(C.byref(mesh), k, C.byref(prm), _p(src_u), _p(src_p), _p(rhs_u),
                          _p(rhs_p), _p(user_u), _p(user_p), _u8(con_u), _u8(con_p), _p(lin),
                          _p(rho), _p(mu), _p(damp), _p(old_u), _p(oldold_u))
    return rhs_u, rhs_p


def ns_velocity_vmult(mesh, k, prm, src_u, con_u=None, lin=None, rho=None, mu=None, damp=None):
    dst_u = np.zeros_like(src_u)
    lib().orc_ns_velocity_vmult(C.byref(mesh), k, C.byref(prm), _p(src_u), _p(dst_u), _u8(con_u),
                                _p(lin), _p(rho), _p(mu), _p(damp))
    return dst_u


def ns_divergence_vmult_add(mesh, k, prm, src_u, dst_p, con_u=None, con_p=None, mu=None,
                            weight_by_viscosity=False):
    dst_p = dst_p.copy()
    lib().orc_ns_divergence_vmult_add(C.byref(mesh), k, C.byref(prm), _p(src_u), _p(dst_p),
                                      _u8(con_u), _u8(con_p), _p(mu), int(weight_by_viscosity))
    return dst_p


def ns_pressure_poisson_vmult(mesh, k, prm, src_p, con_p=None, rho=None):
    dst_p = np.zeros_like(src_p)
    lib().orc_ns_pressure_poisson_vmult(C.byref(mesh), k, C.byref(prm), _p(src_p), _p(dst_p),
                                        _u8(con_p), _p(rho))
    return dst_p


def ns_pressure_mass_vmult(mesh, k, prm, src_p, con_p=None, mu=None):
    dst_p = np.zeros_like(src_p)
    lib().orc_ns_pressure_mass_vmult(C.byref(mesh), k, C.byref(prm), _p(src_p), _p(dst_p),
                                     _u8(con_p), _p(mu))
    return dst_p


def ns_pressure_convdiff_vmult(mesh, k, prm, src_p, con_p=None, mu=None):
    dst_p = np.zeros_like(src_p)
    lib().orc_ns_pressure_convdiff_vmult(C.byref(mesh), k, C.byref(prm), _p(src_p), _p(dst_p),
                                         _u8(con_p), _p(mu))
    return dst_p


def ns_pressure_mass_weight(mesh, k, con_p=None):
    w = np.zeros(mesh.n_nodes(k - 1))
    lib().orc_ns_pressure_mass_weight(C.byref(mesh), k, _p(w), _u8(con_p))
    return w


def ns_pressure_projection(vec, weights, modes):
    v = vec.copy()
    lib().orc_ns_pressure_projection(C.c_long(v.size), _p(v), _p(weights), _p(modes))
    return v


# ----------------------------------------------------------------------------- level set
def ls_reinit_vmult(mesh, prm, src, normal_q, diffuse_only=False, con=None, diag=None):
    dst = np.zeros_like(src)
    lib().orc_ls_reinit_vmult(C.byref(mesh), C.byref(prm), int(diffuse_only), _p(src), _p(dst),
                              _u8(con), _p(normal_q), _p(diag))
    return dst


def ls_reinit_rhs(mesh, prm, solution, normal_vec, normal_q, diffuse_only=False, first_step=True,
                  con=None):
    dst = np.zeros_like(solution)
    lib().orc_ls_reinit_rhs(C.byref(mesh), C.byref(prm), int(diffuse_only), int(first_step),
                            _p(solution), _p(normal_vec), _p(dst), _u8(con), _p(normal_q))
    return dst


def ls_advect_vmult(mesh, prm, src, vel_q, con=None, diag=None, art_visc=None, symmetry=0):
    """art_visc [cell]: parameters.convection_stabilization (cell + boundary terms)"""
    dst = np.zeros_like(src)
    lib().orc_ls_advect_vmult(C.byref(mesh), C.byref(prm), _p(src), _p(dst), _u8(con), _p(vel_q),
                              _p(diag), _p(art_visc), C.c_uint(symmetry))
    return dst


def ls_advect_boundary_term(mesh, prm, vec, art_visc, sign, dst, con=None, symmetry=0):
    dst = dst.copy()
    lib().orc_ls_advect_boundary_term(C.byref(mesh), C.byref(prm), _p(vec), _p(art_visc), C.c_double(sign),
                                      C.c_uint(symmetry), _p(dst), _u8(con))
    return dst


def ls_max_velocity(mesh, ku, vel):
    f = lib().orc_ls_max_velocity
    f.restype = C.c_double
    return f(C.byref(mesh), ku, _p(vel))


def ls_normal_vmult(mesh, prm, src, con=None, diag=None):
    dst = np.zeros_like(src)
    lib().orc_ls_normal_vmult(C.byref(mesh), C.byref(prm), _p(src), _p(dst), _u8(con), _p(diag))
    return dst


def ls_normal_rhs(mesh, prm, solution, con=None):
    dst = np.zeros(mesh.dim * solution.size)
    lib().orc_ls_normal_rhs(C.byref(mesh), C.byref(prm), _p(solution), _p(dst), _u8(con))
    return dst


def ls_curvature_vmult(mesh, prm, src, apply_diffusion=True, con=None, diag=None):
    dst = np.zeros_like(src)
    lib().orc_ls_curvature_vmult(C.byref(mesh), C.byref(prm), int(apply_diffusion), _p(src),
                                 _p(dst), _u8(con), _p(diag))
    return dst


def ls_curvature_rhs(mesh, prm, normal_vec, con=None):
    nn = normal_vec.size // mesh.dim
    dst = np.zeros(nn)
    lib().orc_ls_curvature_rhs(C.byref(mesh), C.byref(prm), _p(normal_vec), _p(dst), _u8(con))
    return dst


# ----------------------------------------------------------------------------- analytic fields
def beltrami_u(xyz, t, nu=1.0):
    """tests/beltrami.cc:82-115 (ExactSolutionU), dim 2 and 3."""
    a = 0.25 * np.pi
    dim = xyz.shape[1]
    if dim == 3:
        d = 2.0 * a
        x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        f = np.exp(-nu * d * d * t)
        u = np.stack([
            -a * (np.exp(a * x) * np.sin(a * y + d * z) + np.exp(a * z) * np.cos(a * x + d * y)) * f,
            -a * (np.exp(a * y) * np.sin(a * z + d * x) + np.exp(a * x) * np.cos(a * y + d * z)) * f,
            -a * (np.exp(a * z) * np.sin(a * x + d * y) + np.exp(a * y) * np.cos(a * z + d * x)) * f,
        ], axis=1)
    else:
        x, y = xyz[:, 0], xyz[:, 1]
        f = np.exp(-2.0 * nu * a * a * t)
        u = np.stack([-a * np.cos(a * x) * np.sin(a * y) * f,
                      a * np.sin(a * x) * np.cos(a * y) * f], axis=1)
    return np.ascontiguousarray(u.reshape(-1))


def beltrami_p(xyz, t, nu=1.0):
    """tests/beltrami.cc:138-172 (ExactSolutionP)."""
    a = 0.25 * np.pi
    dim = xyz.shape[1]
    if dim == 3:
        d = 2.0 * a
        x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        v = -a * a * 0.5 * (
            np.exp(2 * a * x) + np.exp(2 * a * y) + np.exp(2 * a * z)
            + 2 * np.sin(a * x + d * y) * np.cos(a * z + d * x) * np.exp(a * (y + z))
            + 2 * np.sin(a * y + d * z) * np.cos(a * x + d * y) * np.exp(a * (z + x))
            + 2 * np.sin(a * z + d * x) * np.cos(a * y + d * z) * np.exp(a * (x + y))
        ) * np.exp(-2 * nu * d * d * t)
    else:
        x, y = xyz[:, 0], xyz[:, 1]
        v = -a * a * 0.25 * (np.cos(2 * a * x) + np.cos(2 * a * y)) * np.exp(-4.0 * nu * a * a * t)
    return np.ascontiguousarray(v)


# ----------------------------------------------------------------------------- fast CPU baseline
def fast_n_threads():
    return int(lib().orc_fast_n_threads())


def usable_cores():
    """cores this process may really use: scheduler affinity capped by the cgroup CPU quota (a container
    that sees 256 cores may be allowed 16; 128 OpenMP threads then run three times slower than 16)"""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def fast_set_threads(n):
    lib().orc_fast_set_threads(int(n))


def fast_ns_vmult(mesh, k, prm, src_u, src_p, con_u=None, con_p=None, lin=None, rho=None, mu=None,
                  damp=None, weights=None, modes=None, out=None):
    """sum-factorised OpenMP restatement (adaflo_oracle_fast.c); the timed CPU baseline."""
    dst_u, dst_p = out if out is not None else (np.empty_like(src_u), np.empty_like(src_p))
    rc = lib().orc_fast_ns_vmult(C.byref(mesh), k, C.byref(prm), _p(src_u), _p(src_p), _p(dst_u),
                                 _p(dst_p), _u8(con_u), _u8(con_p), _p(lin), _p(rho), _p(mu),
                                 _p(damp), _p(weights), _p(modes))
    assert rc == 0
    return dst_u, dst_p


def fast_ns_residual(mesh, k, prm, src_u, src_p, old_u, oldold_u, con_u=None, con_p=None, lin=None, rho=None, mu=None,
                     damp=None, user_u=None, user_p=None):
    """OpenMP restatement of NavierStokesMatrix::residual for the fully implicit schemes (adaflo_oracle_fast.c): returns
    (rhs_u, rhs_p), `lin` is overwritten with the state the vmults of the Newton step read"""
    rhs_u, rhs_p = np.empty_like(src_u), np.empty_like(src_p)
    rc = lib().orc_fast_ns_residual(C.byref(mesh), k, C.byref(prm), _p(src_u), _p(src_p), _p(rhs_u), _p(rhs_p),
                                    _p(user_u), _p(user_p), _u8(con_u), _u8(con_p), _p(lin), _p(rho), _p(mu), _p(damp),
                                    _p(old_u), _p(oldold_u))
    assert rc == 0, rc
    return rhs_u, rhs_p


class BatchedNSVmult:
    """cell-batched restatement (adaflo_oracle_batched.c): W cells per SIMD register, state in the batched layout,
    compile-time loop bounds, optional even-odd 1D kernels -- the data flow of deal.II's FEEvaluation path"""

    def __init__(self, mesh, k, con_u=None, con_p=None, lin=None):
        L = lib()
        L.orc_batched_prepare.restype = C.c_void_p
        L.orc_batched_isa.restype = C.c_char_p
        self._keep = (mesh, None if con_u is None else np.ascontiguousarray(con_u, dtype=np.uint8),
                      None if con_p is None else np.ascontiguousarray(con_p, dtype=np.uint8))
        self.mesh, self.k = mesh, k
        self._h = L.orc_batched_prepare(C.byref(mesh), k, _u8(self._keep[1]), _u8(self._keep[2]), _p(lin))
        assert self._h, "orc_batched_prepare refused (3D, 2 <= k <= 5)"
        self.isa, self.width = L.orc_batched_isa().decode(), int(L.orc_batched_width())

    def vmult(self, prm, src_u, src_p, weights=None, modes=None, even_odd=True, out=None):
        dst_u, dst_p = out if out is not None else (np.empty_like(src_u), np.empty_like(src_p))
        rc = lib().orc_batched_ns_vmult(C.c_void_p(self._h), C.byref(prm), _p(src_u), _p(src_p), _p(dst_u), _p(dst_p),
                                        _p(weights), _p(modes), int(even_odd))
        assert rc == 0, rc
        return dst_u, dst_p

    def __del__(self):
        try:
            if self._h:
                lib().orc_batched_free(C.c_void_p(self._h))
                self._h = None
        except Exception:
            pass


def ls_advect_rhs(mesh, prm, ku, solution, solution_old, solution_old_old, vel, vel_q,
                  weight_old, weight_old_old, use_old_old=True, con=None, vel_old=None, vel_old_old=None,
                  old_step_size=1.0, global_scaling=1.0, art_visc=None):
    """art_visc [cell] (written) switches parameters.convection_stabilization on"""
    dst = np.zeros_like(solution)
    lib().orc_ls_advect_rhs(C.byref(mesh), C.byref(prm), ku, int(use_old_old), C.c_double(weight_old),
                            C.c_double(weight_old_old), _p(solution), _p(solution_old),
                            _p(solution_old_old), _p(vel), _p(dst), _u8(con), _p(vel_q), _p(vel_old),
                            _p(vel_old_old), C.c_double(old_step_size), C.c_double(global_scaling), _p(art_visc))
    return dst


def make_ls_params(s, epsilon_used, minimal_edge_length, time_step, weight, cell_diameter, epsilon):
    p = LSParams()
    p.ls_degree, p.epsilon_used, p.minimal_edge_length = s, epsilon_used, minimal_edge_length
    p.time_step, p.weight, p.cell_diameter, p.epsilon = time_step, weight, cell_diameter, epsilon
    return p


# --------------------------------------------------------------------------- compute_force
class ForceParams(C.Structure):
    _fields_ = [("surface_tension", C.c_double), ("gravity", C.c_double), ("density", C.c_double),
                ("density_diff", C.c_double), ("viscosity", C.c_double), ("viscosity_diff", C.c_double),
                ("interpolate_grad_onto_pressure", C.c_int)]


def discrete_heaviside(x):
    f = lib().orc_discrete_heaviside
    f.restype = C.c_double
    return np.array([f(C.c_double(v)) for v in np.atleast_1d(x)])


def ls_compute_heaviside(mesh, s, epsilon, phi):
    out = np.zeros_like(phi)
    lib().orc_ls_compute_heaviside(C.byref(mesh), s, C.c_double(epsilon), _p(phi), _p(out))
    return out


def ls_compute_force(mesh, s, ku, heaviside, curvature, surface_tension=1.0, gravity=0.0, density=1.0,
                     density_diff=0.0, viscosity=1.0, viscosity_diff=0.0, interpolate_grad_onto_pressure=False,
                     con_u=None, dst_u=None):
    """returns (user_rhs velocity block, rho_q, mu_q); rho_q / mu_q are None for constant parameters"""
    p = ForceParams(surface_tension, gravity, density, density_diff, viscosity, viscosity_diff,
                    int(interpolate_grad_onto_pressure))
    dim = mesh.dim
    dst = np.zeros(mesh.n_nodes(ku) * dim) if dst_u is None else dst_u.copy()
    nq = (ku + 1) ** dim
    variable = density_diff != 0.0 or viscosity_diff != 0.0
    rho = np.zeros(mesh.n_cells * nq) if variable else None
    mu = np.zeros(mesh.n_cells * nq) if variable else None
    rc = lib().orc_ls_compute_force(C.byref(mesh), s, ku, C.byref(p), _p(heaviside), _p(curvature), _p(dst),
                                    _u8(con_u), _p(rho), _p(mu))
    assert rc == 0
    return dst, rho, mu
