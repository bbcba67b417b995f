"""Host-side mirror of adaflo::NavierStokesMatrix<dim>
(include/adaflo/navier_stokes_matrix.h:44-282): same method names, argument
meaning and error behaviour, forwarding to the C ABI (include/adaflo_hip.h).

deal.II's MatrixFree<dim> is replaced by a structured-brick description
(BrickMesh + boundary-face constraint sets)."""
import ctypes as C

import numpy as np

from . import _lib
from .parameters import LINEARIZATIONS, PHYSICAL_TYPES
from .vectors import BlockVector, DeviceVector


class BrickMesh:
    """ncell[d] hexahedra on [lower, upper]; faces numbered 2*d+side as deal.II's
    GridGenerator::subdivided_hyper_rectangle(colorize=true) boundary ids.

    dim = 2 (two entries per argument): the engine runs the same kernels with a FLAT third direction -- one node,
    one quadrature point of weight 1 (csrc/fe_kernels.hpp, SumFac<.., ZF>).  The mesh then reports ncell[2] = 1,
    h[2] = 1 and one node layer; vectors keep three velocity components per node, the third one constrained."""

    def __init__(self, ncell, lower, upper):
        assert len(ncell) in (1, 2, 3) and len(lower) == len(ncell) == len(upper)
        self.dim = len(ncell)
        pad = 3 - self.dim                      # flat directions (dim = 1: tests/1d_flow.cc, NavierStokesMatrix<1>)
        self.ncell = [int(n) for n in ncell] + [1] * pad
        self.lower = [float(x) for x in lower] + [0.0] * pad
        self.upper = [float(x) for x in upper] + [1.0] * pad
        self.h = [(u - l) / n for u, l, n in zip(self.upper, self.lower, self.ncell)]

    @property
    def hd(self):
        """edge lengths of the directions that exist"""
        return self.h[:self.dim]

    def nodes(self, degree):
        """nodes per direction of a degree-`degree` space (one layer in the flat direction)"""
        return [degree * n + 1 for n in self.ncell[:self.dim]] + [1] * (3 - self.dim)

    @property
    def n_cells(self):
        return int(np.prod(self.ncell))

    def n_nodes(self, degree):
        return int(np.prod(self.nodes(degree)))


def face_mask(faces, ncomp=1, comps=None):
    m = 0
    for f in faces:
        for c in range(ncomp):
            if comps is None or c in comps:
                m |= 1 << (ncomp * f + c) if ncomp > 1 else 1 << f
    return m


class NavierStokesMatrix:
    """Operator object usable by any Krylov solver that needs `vmult(dst, src)`."""

    def __init__(self, parameters, mesh, dirichlet_faces_u=range(6), constrained_faces_p=(),
                 device=0, stream=None, ls_degree=0, symmetry_faces_u=(), normal_flux_faces_u=()):
        """dirichlet_faces_u: all velocity components constrained; symmetry_faces_u: only the component normal to
        the face (FlowBaseAlgorithm::set_symmetry_boundary, source/flow_base_algorithm.cc:95-101); normal_flux_faces_u:
        only the tangential components (set_open_boundary_with_normal_flux, :140-155 with
        VectorTools::compute_normal_flux_constraints, source/navier_stokes.cc:294-297)"""
        self.parameters = parameters
        self.mesh = mesh
        self._lib = _lib.load()
        self._ctx = None
        self._indexed = hasattr(mesh, "cell_nodes")     # IndexedMesh: the adapter's tables instead of a brick
        if self._indexed:
            assert mesh.k == parameters.velocity_degree
            self._device, self._stream = device, stream
            self.time_stepping = None
            self._variant = 0
            self._has_variable_coefficients = False
            return
        self._desc = _lib.BrickDesc()
        d = self._desc
        d.dim = mesh.dim
        for i in range(3):
            d.ncell[i] = mesh.ncell[i]
            d.h[i] = mesh.h[i]
            d.origin[i] = mesh.lower[i]
        d.velocity_degree = parameters.velocity_degree
        d.ls_degree = ls_degree
        # dim = 2: only the faces of the directions that exist (the engine constrains the third component itself)
        d.velocity_constrained = face_mask([f for f in dirichlet_faces_u if f < 2 * mesh.dim], 3)
        for f in symmetry_faces_u:
            if f < 2 * mesh.dim:
                d.velocity_constrained |= face_mask([f], 3, comps=[f // 2])
        for f in normal_flux_faces_u:
            if f < 2 * mesh.dim:
                d.velocity_constrained |= face_mask([f], 3, comps=[c for c in range(mesh.dim) if c != f // 2])
        d.pressure_constrained = face_mask([f for f in constrained_faces_p if f < 2 * mesh.dim], 1)
        d.ls_constrained = 0
        d.device = device
        d.stream = None
        self._stream = stream       # None: engine-owned stream; int (0 = default stream): caller's
        self.time_stepping = None
        self._variant = 1
        self._has_variable_coefficients = False

    # -- lifetime -----------------------------------------------------------------------------
    def initialize(self, time_stepping, pressure_average_fix):
        """NavierStokesMatrix::initialize, source/navier_stokes_matrix.cc:85-168"""
        self.clear()
        ctx = _lib.CtxHandle()
        if self._indexed:
            m, k = self.mesh, self.mesh.k
            keep = [np.ascontiguousarray(m.cell_nodes[k], dtype=np.int32), np.ascontiguousarray(m.cell_nodes[k - 1], dtype=np.int32),
                    np.ascontiguousarray(m.constrained_u, dtype=np.uint8), np.ascontiguousarray(m.constrained_p, dtype=np.uint8),
                    np.ascontiguousarray(m.cell_extents, dtype=np.float64), np.ascontiguousarray(m.colour_offsets, dtype=np.int64)]
            d = _lib.IndexedDesc()
            d.device, d.stream, d.velocity_degree = self._device, None, k
            d.pressure_average_fix = int(bool(pressure_average_fix))
            d.n_cells, d.n_nodes_u, d.n_nodes_p = m.n_cells, m.n_nodes(k), m.n_nodes(k - 1)
            d.cell_nodes_u = keep[0].ctypes.data_as(C.POINTER(C.c_int))
            d.cell_nodes_p = keep[1].ctypes.data_as(C.POINTER(C.c_int))
            d.constrained_u = keep[2].ctypes.data_as(C.POINTER(C.c_ubyte))
            d.constrained_p = keep[3].ctypes.data_as(C.POINTER(C.c_ubyte))
            d.cell_extents = keep[4].ctypes.data_as(C.POINTER(C.c_double))
            d.n_colours = len(keep[5]) - 1
            d.colour_offsets = keep[5].ctypes.data_as(C.POINTER(C.c_int64))
            for name, degree in (("u", k), ("p", k - 1)):        # hanging-node constraints (RefinedMesh), optional
                ptr, master, weight = getattr(m, "hanging", {}).get(degree, (np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0)))
                if len(ptr) > 1:
                    keep += [np.ascontiguousarray(ptr, dtype=np.int64), np.ascontiguousarray(master, dtype=np.int32),
                             np.ascontiguousarray(weight, dtype=np.float64)]
                    setattr(d, "n_hanging_" + name, len(ptr) - 1)
                    setattr(d, "hanging_ptr_" + name, keep[-3].ctypes.data_as(C.POINTER(C.c_int64)))
                    setattr(d, "hanging_master_" + name, keep[-2].ctypes.data_as(C.POINTER(C.c_int)))
                    setattr(d, "hanging_weight_" + name, keep[-1].ctypes.data_as(C.POINTER(C.c_double)))
            code = self._lib.adaflo_ctx_create_indexed(C.byref(d), C.byref(ctx))
        else:
            self._desc.pressure_average_fix = int(bool(pressure_average_fix))
            code = self._lib.adaflo_ctx_create(C.byref(self._desc), C.byref(ctx))
        if code != 0:
            raise _lib.AdafloError("adaflo_ctx_create failed (%d): %s" % (
                code, self._lib.adaflo_last_error(None).decode()))
        ctx.alive = True
        self._ctx = ctx
        if self._stream is not None:
            _lib.check(ctx, self._lib.adaflo_set_stream(ctx, self._stream or None))
        self.time_stepping = time_stepping
        self.update_parameters()

    def clear(self):
        if self._ctx is not None:
            self._ctx.alive = False      # surviving DeviceVectors must not free into a dead context
            self._lib.adaflo_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.clear()
        except Exception:
            pass

    def _require(self):
        if self._ctx is None:
            raise _lib.AdafloError("ExcNotInitialized: call initialize() first")
        return self._ctx

    def update_parameters(self):
        """push FlowParameters + TimeStepping scalars (read at navier_stokes_matrix.cc:621-653)"""
        ctx = self._require()
        p, ts = self.parameters, self.time_stepping
        if ts.tau2() != 0.0:
            raise NotImplementedError("tau2 != 0 (navier_stokes_matrix.cc:629)")
        q = _lib.NSParams()
        q.physical_type = PHYSICAL_TYPES[p.physical_type]
        q.linearization = LINEARIZATIONS[p.linearization]
        q.beta = p.beta
        q.tau_grad_div = p.tau_grad_div
        q.density = p.density
        q.viscosity = p.viscosity
        q.damping = p.stored_damping
        q.density_diff = p.density_diff
        q.weight, q.weight_old, q.weight_old_old = ts.weight(), ts.weight_old(), ts.weight_old_old()
        q.tau1 = ts.tau1()
        q.extrap_old, q.extrap_old_old = ts.factor_extrapol_old, ts.factor_extrapol_old_old
        _lib.check(ctx, self._lib.adaflo_ns_set_params(ctx, C.byref(q)))

    # -- sizes / vectors ----------------------------------------------------------------------
    def n_dofs_u(self):
        return self._lib.adaflo_n_dofs_u(self._require())

    def n_dofs_p(self):
        return self._lib.adaflo_n_dofs_p(self._require())

    def n_cells(self):
        return self._lib.adaflo_n_cells(self._require())

    def n_q_points(self):
        return self._lib.adaflo_n_q_points_u(self._require())

    def pressure_degree(self):
        return self.parameters.velocity_degree - 1

    def initialize_u_vector(self, values=None):
        v = DeviceVector(self._require(), self.n_dofs_u())
        v.set(np.zeros(v.n) if values is None else values)
        return v

    def initialize_p_vector(self, values=None):
        v = DeviceVector(self._require(), self.n_dofs_p())
        v.set(np.zeros(v.n) if values is None else values)
        return v

    def block_vector(self, u=None, p=None):
        return BlockVector([self.initialize_u_vector(u), self.initialize_p_vector(p)])

    def synchronize(self):
        _lib.check(self._ctx, self._lib.adaflo_synchronize(self._require()))

    @staticmethod
    def has_kernel_variant(variant):
        """are the kernels of `variant` in this build of the library (2, 3: the superseded Q3..Q5 kernels, built only with
        ADAFLO_BUILD_VARIANTS=1)"""
        return bool(_lib.load().adaflo_has_kernel_variant(int(variant)))

    def set_kernel_variant(self, variant):
        _lib.check(self._ctx, self._lib.adaflo_set_kernel_variant(self._require(), variant))
        self._variant = variant

    # -- quadrature-point state ---------------------------------------------------------------
    def set_linearization(self, lin):
        """canonical [cell][q][dim+dim*dim] (navier_stokes_matrix.h:54-56)"""
        ctx = self._require()
        lin = np.ascontiguousarray(lin, dtype=np.float64)
        assert lin.size == self.n_cells() * self.n_q_points() * 12
        _lib.check(ctx, self._lib.adaflo_ns_set_linearization(ctx, lin.ctypes.data, 0))

    def get_linearization(self):
        ctx = self._require()
        out = np.empty(self.n_cells() * self.n_q_points() * 12)
        _lib.check(ctx, self._lib.adaflo_ns_get_linearization(ctx, out.ctypes.data, 0))
        return out

    def set_coefficients(self, rho=None, mu=None, damping=None):
        ctx = self._require()
        self._has_variable_coefficients = rho is not None
        if rho is None:
            _lib.check(ctx, self._lib.adaflo_ns_set_coefficients(ctx, None, None, None, 0))
            return
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (rho, mu, damping)]
        _lib.check(ctx, self._lib.adaflo_ns_set_coefficients(
            ctx, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, 0))

    def get_coefficients(self):
        """(rho, mu, damping) at the quadrature points, canonical [cell][q]"""
        ctx = self._require()
        n = self.n_cells() * self.n_q_points()
        out = [np.empty(n) for _ in range(3)]
        _lib.check(ctx, self._lib.adaflo_ns_get_coefficients(ctx, out[0].ctypes.data, out[1].ctypes.data,
                                                             out[2].ctypes.data, 0))
        return out

    def fix_linearization_point(self):
        _lib.check(self._ctx, self._lib.adaflo_ns_fix_linearization_point(self._require()))

    # -- the operator methods (navier_stokes_matrix.h:125-153) --------------------------------
    def vmult(self, dst, src):
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_vmult(ctx, dst.block(0).ptr, dst.block(1).ptr,
                                                  src.block(0).ptr, src.block(1).ptr))

    def vmult_phase(self, dst, src, phase, interface_faces):
        """one of the three parts of vmult used to overlap the inter-GPU exchange with interior
        cells (adaflo_ns_vmult_phase); interface_faces: bit mask of the brick faces shared with
        other GPUs.  No mean-value projection."""
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_vmult_phase(ctx, dst.block(0).ptr, dst.block(1).ptr,
                                                        src.block(0).ptr, src.block(1).ptr, phase,
                                                        interface_faces))

    def supports_phases(self):
        """phased execution exists for the sweep kernels (Q2/Q1; Q3..Q5 with constant coefficients)"""
        return bool(self._lib.adaflo_ns_supports_phases(self._require()))

    def residual(self, residual_vector, src, user_rhs, solution_old, solution_old_old):
        """solution_old / solution_old_old are constructor references in the reference
        (navier_stokes_matrix.h:57-64); passed per call here."""
        ctx = self._require()
        ur = user_rhs.block(0).ptr if user_rhs is not None else None
        pr = user_rhs.block(1).ptr if user_rhs is not None else None
        _lib.check(ctx, self._lib.adaflo_ns_residual(
            ctx, residual_vector.block(0).ptr, residual_vector.block(1).ptr, src.block(0).ptr,
            src.block(1).ptr, ur, pr,
            solution_old.block(0).ptr if solution_old is not None else None,
            solution_old_old.block(0).ptr if solution_old_old is not None else None))

    def velocity_vmult(self, dst, src):
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_velocity_vmult(ctx, dst.ptr, src.ptr))

    def velocity_block_diagonal(self, dst):
        """diagonal of the operator of velocity_vmult (what the Jacobi-preconditioned inner velocity solves
        of the block preconditioner use); 1 on constrained rows"""
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_velocity_block_diagonal(ctx, dst.ptr))

    def divergence_vmult_add(self, dst, src, weight_by_viscosity=False):
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_divergence_vmult_add(ctx, dst.ptr, src.ptr,
                                                                 int(weight_by_viscosity)))

    def pressure_poisson_vmult(self, dst, src):
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_pressure_poisson_vmult(ctx, dst.ptr, src.ptr))

    def pressure_mass_vmult(self, dst, src):
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_pressure_mass_vmult(ctx, dst.ptr, src.ptr))

    def pressure_convdiff_vmult(self, dst, src):
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_pressure_convdiff_vmult(ctx, dst.ptr, src.ptr))

    def apply_pressure_average_projection(self, vector):
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_apply_pressure_average_projection(ctx, vector.ptr))

    def get_matvec_statistics(self):
        ctx = self._require()
        n, s = C.c_uint(), C.c_double()
        _lib.check(ctx, self._lib.adaflo_ns_get_matvec_statistics(ctx, C.byref(n), C.byref(s)))
        return s.value, n.value

    def get_kernel_statistics(self):
        """(seconds, launches) of the dominant cell kernel since the last query"""
        ctx = self._require()
        n, s = C.c_uint(), C.c_double()
        _lib.check(ctx, self._lib.adaflo_get_kernel_statistics(ctx, C.byref(n), C.byref(s)))
        return s.value, n.value

    def set_timing(self, enabled):
        _lib.check(self._ctx, self._lib.adaflo_set_timing(self._require(), int(enabled)))

    def set_x_chunk(self, cells):
        """cells per workgroup along x of the Q3..Q5 x-marching kernel (0 = heuristic)"""
        _lib.check(self._ctx, self._lib.adaflo_set_hox_chunk(self._require(), int(cells)))

    def set_q2_chunk(self, layers):
        _lib.check(self._ctx, self._lib.adaflo_set_q2_chunk(self._require(), int(layers)))

    def set_lazy_state(self, lazy):
        """Q2/Q1 Newton residual: lay out the quadrature-point state only when somebody asks for it (default) or with
        every residual (rounds 1-5); include/adaflo_hip.h: adaflo_set_q2_lazy_state"""
        _lib.check(self._ctx, self._lib.adaflo_set_q2_lazy_state(self._require(), int(bool(lazy))))

    def pressure_mass_weight(self, dst):
        """dst += integral of the pressure shape functions (local_pressure_mass_weight)"""
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_pressure_mass_weight_add(ctx, dst.ptr))

    def apply_constrained_rows(self, dst, src):
        ctx = self._require()
        _lib.check(ctx, self._lib.adaflo_ns_apply_constrained_rows(
            ctx, dst.block(0).ptr, dst.block(1).ptr, src.block(0).ptr, src.block(1).ptr))

    def projection_active(self):
        """apply_pressure_average_projection is skipped for the projection scheme and the
        stationary equation (navier_stokes_matrix.cc:196-197)"""
        p = self.parameters
        return p.linearization != "projection" and p.physical_type != "incompressible stationary"

    def set_q2_state_pad(self, pad_16b):
        _lib.check(self._ctx, self._lib.adaflo_set_q2_state_pad(self._require(), int(pad_16b)))

    # -- torch plumbing used by the multi-GPU layer ------------------------------------------
    def _torch_device(self):
        import torch
        return torch.device("cuda", self._desc.device)

    def new_u_tensor(self):
        import torch
        return torch.zeros(self.n_dofs_u(), dtype=torch.float64, device=self._torch_device())

    def new_p_tensor(self):
        import torch
        return torch.zeros(self.n_dofs_p(), dtype=torch.float64, device=self._torch_device())

    def wrap(self, tensor):
        return DeviceVector.from_torch(self._require(), tensor)

    def synchronize_torch(self):
        import torch
        torch.cuda.synchronize(self._torch_device())
