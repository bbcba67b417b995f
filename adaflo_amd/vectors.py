"""Device-resident stand-ins for deal.II's LinearAlgebra::distributed::{Vector,BlockVector}<double>:
a contiguous double[n] in HBM (SURVEY.md 8b "Vector layout")."""
import ctypes as C

import numpy as np

from . import _lib


class DeviceVector:
    """Owns (or borrows, e.g. from a torch tensor) a device double[n]."""

    def __init__(self, ctx, n, ptr=None, keepalive=None):
        self.ctx = ctx
        self.n = int(n)
        self._own = ptr is None
        self._keepalive = keepalive
        if ptr is None:
            p = C.c_void_p()
            _lib.check(ctx, _lib.load().adaflo_malloc(ctx, max(self.n, 1) * 8, C.byref(p)))
            ptr = p.value
        self.ptr = ptr

    @classmethod
    def from_numpy(cls, ctx, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        v = cls(ctx, a.size)
        v.set(a)
        return v

    @classmethod
    def from_torch(cls, ctx, t):
        assert t.is_cuda and t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 8
        return cls(ctx, t.numel(), ptr=t.data_ptr(), keepalive=t)

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        _lib.check(self.ctx, _lib.load().adaflo_copy_h2d(self.ctx, self.ptr, a.ctypes.data, self.n * 8))

    def numpy(self):
        out = np.empty(self.n)
        _lib.check(self.ctx, _lib.load().adaflo_copy_d2h(self.ctx, out.ctypes.data, self.ptr, self.n * 8))
        return out

    # -- the vector algebra the drivers around the solvers need (device-side)
    def fill(self, value):
        _lib.check(self.ctx, _lib.load().adaflo_vector_fill(self.ctx, self.ptr, float(value), self.n))

    def sadd(self, a, b, y):
        """self = a * self + b * y"""
        _lib.check(self.ctx, _lib.load().adaflo_vector_sadd(self.ctx, self.ptr, float(a), float(b), y.ptr, self.n))

    def add(self, y):
        self.sadd(1.0, 1.0, y)

    def dot(self, y):
        r = C.c_double()
        _lib.check(self.ctx, _lib.load().adaflo_vector_dot(self.ctx, self.ptr, y.ptr, self.n, C.byref(r)))
        return r.value

    def l2_norm(self):
        return float(np.sqrt(self.dot(self)))

    def free(self):
        # (the device memory of a destroyed context's vectors is released with hipFree semantics by
        # the driver at process end; calling into the library with a dangling context is not an option)
        if self._own and self.ptr and getattr(self.ctx, "alive", True):
            _lib.load().adaflo_free(self.ctx, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BlockVector:
    def __init__(self, blocks):
        self.blocks = list(blocks)

    def block(self, i):
        return self.blocks[i]

    def numpy(self):
        return [b.numpy() for b in self.blocks]
