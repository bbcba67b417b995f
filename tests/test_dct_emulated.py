"""The device source of the fast cosine transforms of the fast-diagonalisation inverses (adaflo_amd/csrc/fdm_dct_kernel.hpp),
compiled for the HOST under the lane emulator of tests/emu/ and compared with the plain cosine sums
y_k = sum_j cos(pi j k / N) x_j (the matrix product the transform replaces).  A test of the FFT's index logic and of the
LDS hand-offs that runs without a GPU; not a product path: tests/test_fdm_gpu.py runs the same source on the device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "dct_emu.cpp")
DEPS = [SRC, os.path.join(HERE, "emu", "hip_emu.hpp"), os.path.join(HERE, "..", "adaflo_amd", "csrc", "fdm_dct_kernel.hpp")]
LIB = os.path.join(HERE, "emu", "_dct_emu.so")


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    lib.dct_emu_apply.restype = C.c_int
    lib.dct_emu_apply.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_double)] * 8 + [C.c_double] * 3
    return lib


def cosines(n):
    j = np.arange(n)
    return np.cos(np.pi * np.outer(j, j) / (n - 1))


def apply(lib, axis, field, fused=False, scal=None, pitch=None):
    """pitch: the rows of both arrays are padded to that many elements (NaN in the padding: it must be neither read
    into a result nor written)"""
    nz, ny, nx = field.shape
    pitch = pitch or nx
    padded = np.full((nz, ny, pitch), np.nan)
    padded[:, :, :nx] = field
    out = np.full_like(padded, np.nan)
    dp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))
    s = scal or dict(lx=None, ly=None, lz=None, ax=None, ay=None, az=None, cm=0.0, cl=0.0, eps=0.0)
    rc = lib.dct_emu_apply(axis, int(fused), nx, ny, nz, pitch, dp(padded), dp(out), dp(s["lx"]), dp(s["ly"]), dp(s["lz"]),
                           dp(s["ax"]), dp(s["ay"]), dp(s["az"]), s["cm"], s["cl"], s["eps"])
    assert rc == 0
    assert np.all(np.isnan(out[:, :, nx:]))
    return np.ascontiguousarray(out[:, :, :nx])


@pytest.mark.parametrize("axis,shape", [(0, (3, 5, 65)), (0, (2, 3, 129)), (0, (1, 19, 257)), (0, (1, 3, 513)), (0, (1, 5, 1025)),
                                        (1, (3, 65, 7)), (1, (2, 129, 37)), (1, (1, 257, 18)),
                                        (2, (65, 5, 15)), (2, (257, 3, 7)), (2, (513, 2, 5)),
                                        # 5 2^m intervals (the reference's meshes: 5 x 10 coarse cells): radix-5 stage first, batches of
                                        # 51 / 25 / 12 / 6 lines that do not fill the workgroup
                                        (0, (2, 30, 81)), (0, (1, 27, 161)), (0, (1, 13, 321)), (0, (1, 7, 641)),
                                        (1, (2, 81, 53)), (1, (1, 161, 26)), (2, (161, 3, 9)), (2, (321, 2, 7)),
                                        # 3 2^m intervals: radix-3 stage first
                                        (0, (2, 23, 97)), (0, (1, 21, 193)), (0, (1, 11, 385)), (0, (1, 6, 769)), (1, (1, 97, 44)), (2, (193, 2, 9))])
def test_cosine_sums_along_an_axis(emu, axis, shape):
    """ragged batches (line counts that are no multiple of the batch), every supported length, all three axes"""
    rng = np.random.default_rng(7)
    field = rng.standard_normal(shape)
    n = shape[2 - axis]
    ref = np.moveaxis(np.tensordot(cosines(n), field, axes=([1], [2 - axis])), 0, 2 - axis)
    got = apply(emu, axis, field)
    assert np.abs(got - ref).max() < 1e-13 * n * np.abs(field).max() * 4
    got = apply(emu, axis, field, pitch=(shape[2] + 15) // 16 * 16)          # padded rows (the intermediate arrays)
    assert np.abs(got - ref).max() < 1e-13 * n * np.abs(field).max() * 4


@pytest.mark.parametrize("nz,ny,nx", [(65, 3, 6), (81, 2, 5), (161, 2, 3), (97, 2, 4)])
def test_fused_forward_scaling_backward(emu, nz, ny, nx):
    """z pass of the inverse: S diag(1 / (c_m + c_l (lx + ly + lz))) S^T with S = cosines . diag(sqrt(a)), null mode dropped"""
    rng = np.random.default_rng(11)
    field = rng.standard_normal((nz, ny, nx))
    scal = dict(lx=rng.random(nx), ly=rng.random(ny), lz=rng.random(nz), ax=rng.random(nx) + 0.5, ay=rng.random(ny) + 0.5,
                az=rng.random(nz) + 0.5, cm=0.0, cl=1.3, eps=1e-9)
    scal["lx"][0] = scal["ly"][0] = scal["lz"][0] = 0.0               # the null mode of a pure Neumann problem
    got = apply(emu, 2, field, fused=True, scal=scal, pitch=16)
    Cz = cosines(nz)
    modes = np.tensordot(Cz, field, axes=([1], [0]))
    den = scal["cm"] + scal["cl"] * (scal["lz"][:, None, None] + scal["ly"][None, :, None] + scal["lx"][None, None, :])
    fac = scal["az"][:, None, None] * scal["ay"][None, :, None] * scal["ax"][None, None, :]
    with np.errstate(divide="ignore"):
        modes = np.where(np.abs(den) > scal["eps"], modes * fac / den, 0.0)
    ref = np.tensordot(Cz, modes, axes=([1], [0]))
    assert np.abs(got - ref).max() < 1e-11 * np.abs(ref).max()
