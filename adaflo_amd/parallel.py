"""Multi-GPU layer: brick partition of the mesh, ghost-DoF exchange over RCCL (xGMI),
and the distributed NavierStokesMatrix.

Replaces what deal.II does inside `MatrixFree::cell_loop` for a
LinearAlgebra::distributed::Vector (SURVEY.md section 2 "collective call sites",
8e): `src.update_ghost_values()` before the cell loop and `dst.compress(add)`
after it, plus the one-double all-reduce of the pressure-mean projection
(source/navier_stokes_matrix.cc:201).

Layout: one process per GPU; the uniform mesh is cut into px x py x pz bricks
(1, 2x1x1, 2x2x1, 2x2x2 for 1/2/4/8 GPUs).  Every rank stores the DoFs of its
whole local brick, i.e. nodes on an inter-rank interface are REPLICATED on the
(up to 8) ranks that share them; the sharer with the lowest grid coordinates
owns the node (deal.II: "owned" vs "ghost").  A vector is *consistent* when all
replicas agree.  Both exchanges are ONE round of point-to-point messages to the
<= 26 (here <= 7) neighbours -- one message per xGMI link in the 2x2x2 layout,
no ring collective.
"""
import itertools

import numpy as np


def brick_grid(world):
    """process grid for `world` GPUs of one node"""
    grids = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}
    if world in grids:
        return grids[world]
    g = [1, 1, 1]  # generic: factor into near-cubic grid
    n, d = world, 0
    f = 2
    while n > 1:
        while n % f == 0:
            g[d % 3] *= f
            n //= f
            d += 1
        f += 1
    return tuple(g)


class BrickPartition:
    """rank -> local brick of `cells_per_rank` cells inside the global box."""

    def __init__(self, grid, rank, cells_per_rank, lower, upper):
        self.grid = tuple(grid)
        self.world = int(np.prod(grid))
        self.rank = rank
        self.cells = [int(c) for c in cells_per_rank]
        self.coords = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
        self.global_lower = [float(x) for x in lower]
        self.global_upper = [float(x) for x in upper]
        self.h = [(u - l) / (g * c) for u, l, g, c in zip(upper, lower, grid, self.cells)]
        self.lower = [l + co * c * h for l, co, c, h in zip(lower, self.coords, self.cells, self.h)]
        self.upper = [l + c * h for l, c, h in zip(self.lower, self.cells, self.h)]

    def rank_of(self, coords):
        return coords[0] + self.grid[0] * (coords[1] + self.grid[1] * coords[2])

    def physical_faces(self):
        """faces (2*d+side) of the local brick that lie on the domain boundary"""
        out = []
        for d in range(3):
            if self.coords[d] == 0:
                out.append(2 * d)
            if self.coords[d] == self.grid[d] - 1:
                out.append(2 * d + 1)
        return out

    def interface_mask(self):
        """bit 2*d+side for the faces of the local brick shared with another rank"""
        phys = self.physical_faces()
        return sum(1 << f for f in range(6) if f not in phys)

    def neighbours(self):
        """[(offset, rank)] of all existing neighbours (faces, edges, corners)"""
        out = []
        for o in itertools.product((-1, 0, 1), repeat=3):
            if o == (0, 0, 0):
                continue
            c = [self.coords[d] + o[d] for d in range(3)]
            if all(0 <= c[d] < self.grid[d] for d in range(3)):
                out.append((o, self.rank_of(c)))
        return out

    def n_global_dofs(self, k):
        """velocity + pressure unknowns of the global Taylor-Hood Q_k/Q_{k-1} problem"""
        nu = int(np.prod([k * g * c + 1 for g, c in zip(self.grid, self.cells)]))
        npr = int(np.prod([(k - 1) * g * c + 1 for g, c in zip(self.grid, self.cells)]))
        return 3 * nu + npr

    def nodes(self, degree):
        return [degree * c + 1 for c in self.cells]


def _region(offset, nn):
    """slices (z, y, x) of the local node box shared with the neighbour at `offset`"""
    sl = []
    for d in (2, 1, 0):
        o = offset[d]
        sl.append(slice(None) if o == 0 else (slice(0, 1) if o < 0 else slice(nn[d] - 1, nn[d])))
    return tuple(sl)


class HaloExchange:
    """ghost update / compress(add) for a set of fields living on the same brick.

    fields: list of (degree, ncomp); a field vector is a flat torch tensor of
    nn_z*nn_y*nn_x*ncomp doubles (node-major, components interleaved)."""

    def __init__(self, part, fields, group=None, native=None):
        self.part = part
        self.fields = fields
        self.group = group
        self.native = native      # engine context whose halo kernel packs / unpacks GPU tensors
        self.nbrs = part.neighbours()
        # order messages by dimensionality: faces first, corners last (authoritative owner last)
        self.nbrs.sort(key=lambda t: sum(abs(x) for x in t[0]))
        self._bufs = {}

    def _views(self, vecs):
        out = []
        for v, (deg, nc) in zip(vecs, self.fields):
            nn = self.part.nodes(deg)
            out.append(v.view(nn[2], nn[1], nn[0], nc))
        return out

    def _plan(self, mode, dev, dtype):
        """static description of one exchange round (who sends what where), built once"""
        import torch
        key = (mode, str(dev))
        if key in self._bufs:
            return self._bufs[key]
        send_list, recv_list = [], []
        for o, nb in self.nbrs:
            positive = all(x >= 0 for x in o)   # I am on the low side of every cut direction
            negative = all(x <= 0 for x in o)
            if mode == "add" or positive:
                send_list.append((o, nb))
            if mode == "add" or negative:
                recv_list.append((o, nb))
        if mode == "add":
            # partial sums are added in the order of their senders' ranks, this rank's own value at
            # its place in that order: every sharer of a DoF adds the same numbers in the same
            # sequence and the replicas stay bitwise identical
            recv_list.sort(key=lambda t: t[1])

        def sizes(o):
            return [int(np.prod([len(range(*sl.indices(n))) for sl, n in
                                 zip(_region(o, self.part.nodes(deg)), self.part.nodes(deg)[::-1])])) * nc
                    for deg, nc in self.fields]
        # buffer layout [field][neighbour] so that one kernel per field packs everything
        seg_send, seg_recv = {}, {}
        off = 0
        for f in range(len(self.fields)):
            for o, nb in send_list:
                n = sizes(o)[f]
                seg_send[(f, o)] = (off, n)
                off += n
        ns = off
        off = 0
        for f in range(len(self.fields)):
            for o, nb in recv_list:
                n = sizes(o)[f]
                seg_recv[(f, o)] = (off, n)
                off += n
        nr = off
        plan = dict(send_list=send_list, recv_list=recv_list, seg_send=seg_send, seg_recv=seg_recv,
                    sbuf=torch.empty(max(ns, 1), dtype=dtype, device=dev),
                    rbuf=torch.empty(max(nr, 1), dtype=dtype, device=dev))
        self._bufs[key] = plan
        return plan

    def start(self, vecs, mode):
        """pack the interface regions and post the point-to-point messages with all neighbours.
        With the nccl (RCCL) backend the transfers run asynchronously to kernels launched on the
        current stream afterwards; finish() makes the current stream wait for them and unpacks.
        On GPU tensors packing / unpacking is done by the engine's halo kernel (native=ctx),
        otherwise by torch slicing (CPU tests)."""
        import torch
        import torch.distributed as dist
        if self.part.world == 1 or not self.nbrs:
            return None
        plan = self._plan(mode, vecs[0].device, vecs[0].dtype)
        send_list, recv_list = plan["send_list"], plan["recv_list"]
        seg_send, seg_recv = plan["seg_send"], plan["seg_recv"]
        sbuf, rbuf = plan["sbuf"], plan["rbuf"]
        native = self.native is not None and vecs[0].is_cuda
        if native:
            for f, v in enumerate(vecs):
                if send_list:
                    base = seg_send[(f, send_list[0][0])][0]
                    self._native_transfer(v, sbuf[base:], f, [o for o, _ in send_list], 0)
        else:
            views = self._views(vecs)
            for f, w in enumerate(views):
                for o, nb in send_list:
                    a, n = seg_send[(f, o)]
                    sbuf[a:a + n] = w[_region(o, self.part.nodes(self.fields[f][0]))].reshape(-1)
        # one message per (neighbour, field).  With a gloo group (functional tests of the
        # multi-GPU path on a single GPU) the packed buffers are staged through host memory.
        stage = vecs[0].is_cuda and dist.get_backend(self.group) == "gloo"
        hs, hr = (sbuf.cpu(), torch.empty(rbuf.shape, dtype=rbuf.dtype)) if stage else (sbuf, rbuf)
        ops = []
        for o, nb in send_list:
            for f in range(len(self.fields)):
                a, n = seg_send[(f, o)]
                ops.append(dist.P2POp(dist.isend, hs[a:a + n], nb, group=self.group, tag=f))
        for o, nb in recv_list:
            for f in range(len(self.fields)):
                a, n = seg_recv[(f, o)]
                ops.append(dist.P2POp(dist.irecv, hr[a:a + n], nb, group=self.group, tag=f))
        works = dist.batch_isend_irecv(ops) if ops else []
        return dict(plan=plan, mode=mode, vecs=vecs, works=works, stage=stage, hr=hr, native=native)

    def finish(self, handle):
        """wait for the messages of start() and unpack (copy: faces first, corners last so that
        the lowest sharer wins; add: every replica ends up with the total)"""
        if handle is None:
            return
        plan, mode, vecs = handle["plan"], handle["mode"], handle["vecs"]
        recv_list, seg_recv, rbuf = plan["recv_list"], plan["seg_recv"], plan["rbuf"]
        for w in handle["works"]:
            w.wait()
        if handle["stage"]:
            rbuf.copy_(handle["hr"])
        if handle["native"]:
            for f, v in enumerate(vecs):
                if not recv_list:
                    continue
                if mode == "add":
                    base = seg_recv[(f, recv_list[0][0])][0]
                    self._native_transfer(v, rbuf[base:], f, [o for o, _ in recv_list], 2,
                                          self_pos=sum(1 for _, nb in recv_list if nb < self.part.rank))
                else:
                    for cls in (1, 2, 3):
                        group = [o for o, _ in recv_list if sum(abs(x) for x in o) == cls]
                        if group:
                            base = seg_recv[(f, group[0])][0]
                            self._native_transfer(v, rbuf[base:], f, group, 1)
        else:
            views = self._views(vecs)
            for f, w in enumerate(views):
                if mode == "add":
                    # same summation order as the engine's kernel: contributions sorted by rank
                    own, total, seen = w.clone(), w.clone(), None
                    import torch
                    started = torch.zeros(w.shape, dtype=torch.bool)
                    entries = [(nb, o) for o, nb in recv_list] + [(self.part.rank, None)]
                    for nb, o in sorted(entries, key=lambda t: t[0]):
                        if o is None:
                            touched = torch.ones(w.shape, dtype=torch.bool)
                            val = own
                        else:
                            a, n = seg_recv[(f, o)]
                            r = _region(o, self.part.nodes(self.fields[f][0]))
                            touched = torch.zeros(w.shape, dtype=torch.bool)
                            touched[r] = True
                            val = torch.zeros_like(w)
                            val[r] = rbuf[a:a + n].view(w[r].shape)
                        total = torch.where(touched & started, total + val, torch.where(touched, val, total))
                        started |= touched
                    w.copy_(total)
                    continue
                for o, nb in recv_list:
                    a, n = seg_recv[(f, o)]
                    r = _region(o, self.part.nodes(self.fields[f][0]))
                    w[r] = rbuf[a:a + n].view(w[r].shape)

    def _exchange(self, vecs, mode):
        """one blocking round: start + finish"""
        self.finish(self.start(vecs, mode))

    def _native_transfer(self, vec, buf, field, offsets, mode, self_pos=0):
        """engine halo kernel over the regions of `offsets` (contiguous in buf, in this order)"""
        import ctypes as C
        from . import _lib
        deg, nc = self.fields[field]
        nn = self.part.nodes(deg)
        regs = []
        for o in offsets:
            for d in range(3):
                lo = 0 if o[d] <= 0 else nn[d] - 1
                hi = nn[d] if o[d] == 0 else lo + 1
                regs += [lo, hi]
        arr = (C.c_int * len(regs))(*regs)
        nn_c = (C.c_int * 3)(*nn)
        _lib.check(self.native, _lib.load().adaflo_halo_transfer_ordered(
            self.native, vec.data_ptr(), buf.data_ptr(), nn_c, nc, len(offsets), arr, mode, self_pos))

    def compress_add(self, vecs):
        """dst.compress(VectorOperation::add): afterwards every replica holds the total"""
        self._exchange(vecs, "add")

    def update_ghost_values(self, vecs):
        """owner -> replicas"""
        self._exchange(vecs, "copy")

    def owned_mask(self, field, device=None):
        """1.0 where this rank owns the node (lowest sharer), else 0.0; flat per DoF"""
        import torch
        deg, nc = self.fields[field]
        nn = self.part.nodes(deg)
        m = torch.ones(nn[2], nn[1], nn[0], nc, dtype=torch.float64, device=device)
        for d, axis in ((0, 2), (1, 1), (2, 0)):
            if self.part.coords[d] > 0:            # a lower neighbour exists: it owns my low face
                idx = [slice(None)] * 4
                idx[axis] = 0
                m[tuple(idx)] = 0.0
        return m.reshape(-1)


def _all_reduce_sum(t, group):
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)
    return t


class NativeCommunicator:
    """adaflo_comm of the C ABI (csrc/comm.hip): pack, one message per neighbour, unpack and the
    phased overlap all happen inside the library, ordered by HIP events.  With an nccl group the
    library opens its own RCCL communicator (the unique id travels through torch.distributed);
    with any other group (gloo: the tests that run N ranks on one GPU) the packed buffers are
    moved by callbacks that stage them through host memory."""

    def __init__(self, ctx, part, group, pressure_average_fix):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        self._lib, self._ctx, self.group = _lib.load(), ctx, group
        self.handle = C.c_void_p()
        grid = (C.c_int * 3)(*part.grid)
        backend = dist.get_backend(group) if (part.world > 1 and group is not None) else "none"
        if backend == "nccl":
            uid = _lib.CommUniqueId()
            if part.rank == 0:
                code = self._lib.adaflo_comm_get_unique_id(C.byref(uid))
                if code != 0:
                    raise _lib.AdafloError("adaflo_comm_get_unique_id failed (%d)" % code)
            t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).cuda()
            dist.broadcast(t, 0, group=group)
            C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
            code = self._lib.adaflo_comm_create(ctx, C.byref(uid), part.rank, part.world, grid,
                                                int(pressure_average_fix), C.byref(self.handle))
        else:
            lib = self._lib

            def exchange(user, sbuf, soff, scnt, speer, ns, rbuf, roff, rcnt, rpeer, nr, stream):
                try:
                    lib.adaflo_synchronize(ctx)
                    sends = [torch.empty(scnt[q], dtype=torch.float64) for q in range(ns)]
                    recvs = [torch.empty(rcnt[q], dtype=torch.float64) for q in range(nr)]
                    for q in range(ns):
                        lib.adaflo_copy_d2h(ctx, sends[q].data_ptr(), sbuf + 8 * soff[q], 8 * scnt[q])
                    ops = [dist.P2POp(dist.isend, sends[q], speer[q], group=group) for q in range(ns)]
                    ops += [dist.P2POp(dist.irecv, recvs[q], rpeer[q], group=group) for q in range(nr)]
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                    for q in range(nr):
                        lib.adaflo_copy_h2d(ctx, rbuf + 8 * roff[q], recvs[q].data_ptr(), 8 * rcnt[q])
                    return 0
                except Exception:        # noqa: BLE001 -- must not propagate through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1

            def allreduce(user, buf, n, stream):
                try:
                    t = torch.empty(n, dtype=torch.float64)
                    lib.adaflo_copy_d2h(ctx, t.data_ptr(), buf, 8 * n)
                    dist.all_reduce(t, group=group)
                    lib.adaflo_copy_h2d(ctx, buf, t.data_ptr(), 8 * n)
                    return 0
                except Exception:        # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = (_lib.EXCHANGE_FN(exchange), _lib.ALLREDUCE_FN(allreduce))      # keep alive
            code = self._lib.adaflo_comm_create_custom(ctx, part.rank, part.world, grid, self._cb[0], self._cb[1], None,
                                                       int(pressure_average_fix), C.byref(self.handle))
        if code != 0:
            raise _lib.AdafloError("adaflo_comm_create failed (%d): %s" % (code, self._lib.adaflo_last_error(ctx).decode()))

    def _check(self, code):
        if code != 0:
            from . import _lib
            raise _lib.AdafloError("adaflo_comm error %d: %s" % (code, self._lib.adaflo_comm_last_error(self.handle).decode()))

    def vmult(self, dst, src, src_consistent):
        self._check(self._lib.adaflo_ns_vmult_distributed(self._ctx, self.handle, dst.block(0).ptr, dst.block(1).ptr,
                                                          src.block(0).ptr, src.block(1).ptr, int(src_consistent)))

    def update_ghost_values(self, vec):
        self._check(self._lib.adaflo_comm_update_ghost_values(self.handle, vec.block(0).ptr, vec.block(1).ptr))

    def compress_add(self, vec):
        self._check(self._lib.adaflo_comm_compress_add(self.handle, vec.block(0).ptr, vec.block(1).ptr))

    def force_phased_schedule(self, enabled=True):
        """measurement aid: the three-phase schedule also with one rank (bench.py --through-comm)"""
        self._check(self._lib.adaflo_comm_force_phased_schedule(self.handle, int(enabled)))

    def matvec_statistics(self):
        """get_matvec_statistics over the ranks (navier_stokes_matrix.cc:1194-1206): (count, MinMaxAvg of the
        accumulated seconds); collective, resets the counters"""
        import ctypes as C
        from . import _lib
        n, st = C.c_uint(0), _lib.MinMaxAvg()
        self._check(self._lib.adaflo_comm_matvec_statistics(self.handle, C.byref(n), C.byref(st)))
        return n.value, st

    def set_phase_timing(self, enabled=True):
        self._check(self._lib.adaflo_comm_set_phase_timing(self.handle, int(enabled)))

    def phase_statistics(self):
        """(count, {piece: seconds}) of this rank since the last call: the pieces of adaflo_ns_vmult_distributed"""
        import ctypes as C
        n, sec = C.c_uint(0), (C.c_double * 5)()
        self._check(self._lib.adaflo_comm_phase_statistics(self.handle, C.byref(n), sec))
        return n.value, dict(zip(("src_exchange", "interface_cells", "dst_exchange", "interior_cells", "tail"), list(sec)))

    def close(self):
        if self.handle:
            self._lib.adaflo_comm_destroy(self.handle)
            self.handle = None


class DistributedNavierStokesMatrix:
    """NavierStokesMatrix over a brick partition: local HIP engine + RCCL halo exchange.

    native_comm=True: everything behind the C ABI (adaflo_ns_vmult_distributed, NativeCommunicator);
    False: the exchange is driven from here through torch.distributed point-to-point operations."""

    def __init__(self, parameters, part, device=0, stream=None, group=None, local=None, native_comm=False,
                 through_comm=False):
        """`local`: the per-rank operator; default = the HIP engine on the local brick.  (The
        gloo/CPU tests inject an oracle-backed stand-in to check the exchange logic.)"""
        from .navier_stokes_matrix import BrickMesh, NavierStokesMatrix
        self.part = part
        self.group = group
        self.parameters = parameters
        k = parameters.velocity_degree
        if local is None:
            if stream is None:
                # the exchange mixes engine kernels with torch ops on the same tensors: run the
                # engine on torch's current stream so that everything is ordered
                import torch
                stream = torch.cuda.current_stream(torch.device("cuda", device)).cuda_stream
            mesh = BrickMesh(part.cells, part.lower, part.upper)
            local = NavierStokesMatrix(parameters, mesh, dirichlet_faces_u=part.physical_faces(),
                                       constrained_faces_p=(), device=device, stream=stream)
        self.local = local
        self.native_comm = native_comm and hasattr(local, "_ctx")
        # one rank, but through adaflo_ns_vmult_distributed (phased schedule): exists behind the C ABI only -- with the
        # torch-driven exchange there is no communicator object, the operator would skip the mean-value projection
        # while reporting through_comm
        if through_comm and not self.native_comm:
            raise ValueError("through_comm needs native_comm=True (adaflo_ns_vmult_distributed)")
        self.through_comm = through_comm
        self.comm = None
        self.overlap = True       # overlap the exchanges with interior cells where supported
        self.halo = HaloExchange(part, [(k, 3), (k - 1, 1)], group=group)
        self._w_owned = None
        self._inv = None

    def initialize(self, time_stepping, pressure_average_fix):
        single = self.part.world == 1 and not self.through_comm
        self.local.initialize(time_stepping, pressure_average_fix and single)
        self.halo.native = getattr(self.local, "_ctx", None)
        self.pressure_average_fix = pressure_average_fix
        if self.native_comm and not single:
            self.comm = NativeCommunicator(self.local._ctx, self.part, self.group, pressure_average_fix)
            if self.through_comm:
                self.comm.force_phased_schedule(True)
        elif pressure_average_fix and not single:
            self._setup_projection()

    def set_kernel_variant(self, v):
        self.local.set_kernel_variant(v)

    def _setup_projection(self):
        """global version of source/navier_stokes_matrix.cc:117-168 (mode 0)"""
        import torch
        import torch.distributed as dist
        w = self.local.new_p_tensor()
        self.local.pressure_mass_weight(self.local.wrap(w))
        self.local.synchronize()
        u_dummy = self.local.new_u_tensor()
        self.halo.compress_add([u_dummy, w])
        self._w_owned = w * self.halo.owned_mask(1, device=w.device)
        s = _all_reduce_sum(self._w_owned.sum().reshape(1), self.group)
        self._inv = 1.0 / s                       # device scalar, modes == 1 everywhere

    def get_matvec_statistics(self):
        """NavierStokesMatrix::get_matvec_statistics (navier_stokes_matrix.cc:1194-1206): ((min, max, avg, min_index,
        max_index) of the accumulated vmult seconds over the ranks, count); collective"""
        if self.comm is not None:
            n, st = self.comm.matvec_statistics()
            return (st.min, st.max, st.avg, st.min_index, st.max_index), n
        sec, n = self.local.get_matvec_statistics()
        if self.part.world > 1:
            import torch
            t = torch.zeros(self.part.world, dtype=torch.float64)
            t[self.part.rank] = sec
            t = _all_reduce_sum(t, self.group).cpu()
            return (float(t.min()), float(t.max()), float(t.mean()), int(t.argmin()), int(t.argmax())), n
        return (sec, sec, sec, 0, 0), n

    def make_consistent(self, vec):
        if self.comm is not None:
            self.comm.update_ghost_values(vec)
        elif self.part.world > 1:
            self.halo.update_ghost_values([b._keepalive for b in vec.blocks])

    def vmult(self, dst, src, src_consistent=False):
        """NavierStokesMatrix::vmult on the global problem; block vectors wrap torch tensors.

        With the Q2/Q1 sweep kernel the two exchanges are overlapped with interior cells, the way
        MatrixFree::cell_loop overlaps update_ghost_values / compress with its cell partitions:
          ghost update of src   ||  interior cells, first half
          cells at the interface, seam sums of the interface nodes
          compress(add) of dst  ||  interior cells, second half"""
        if self.part.world == 1 and self.comm is None:
            self.local.vmult(dst, src)
            return
        if self.comm is not None:
            self.comm.vmult(dst, src, src_consistent)
            return
        tsrc = [b._keepalive for b in src.blocks]
        tdst = [b._keepalive for b in dst.blocks]
        if self.overlap and getattr(self.local, "supports_phases", lambda: False)():
            mask = self.part.interface_mask()
            h = None if src_consistent else self.halo.start(tsrc, "copy")
            self.local.vmult_phase(dst, src, 0, mask)
            self.halo.finish(h)
            self.local.vmult_phase(dst, src, 1, mask)
            h = self.halo.start(tdst, "add")
            self.local.vmult_phase(dst, src, 2, mask)
            self.halo.finish(h)
        else:
            if not src_consistent:
                self.halo.update_ghost_values(tsrc)        # src.update_ghost_values()
            self.local.vmult(dst, src)                      # local cells (no projection)
            self.halo.compress_add(tdst)                    # dst.compress(add)
        self.local.apply_constrained_rows(dst, src)         # rows on boundary x interface
        if self.pressure_average_fix and self.local.projection_active():
            s = _all_reduce_sum((self._w_owned * tdst[1]).sum().reshape(1), self.group)
            tdst[1].sub_(s * self._inv)
