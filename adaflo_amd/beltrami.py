"""Analytic Beltrami (Ethier-Steinman) flow of the reference's tests/beltrami.cc:82-172 in 3D:
initial / boundary data for examples and benchmarks (ExactSolutionU, ExactSolutionP)."""
import numpy as np


def velocity(xyz, t, nu=1.0):
    """u(x, t) for points xyz[n][3]; returns [n][3]"""
    a = 0.25 * np.pi
    d = 2.0 * a
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    f = -a * np.exp(-nu * d * d * t)
    return np.stack([f * (np.exp(a * x) * np.sin(a * y + d * z) + np.exp(a * z) * np.cos(a * x + d * y)),
                     f * (np.exp(a * y) * np.sin(a * z + d * x) + np.exp(a * x) * np.cos(a * y + d * z)),
                     f * (np.exp(a * z) * np.sin(a * x + d * y) + np.exp(a * y) * np.cos(a * z + d * x))], axis=1)


def pressure(xyz, t, nu=1.0):
    """p(x, t) for points xyz[n][3]; returns [n]"""
    a = 0.25 * np.pi
    d = 2.0 * a
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    s = (np.exp(2 * a * x) + np.exp(2 * a * y) + np.exp(2 * a * z)
         + 2 * np.sin(a * x + d * y) * np.cos(a * z + d * x) * np.exp(a * (y + z))
         + 2 * np.sin(a * y + d * z) * np.cos(a * x + d * y) * np.exp(a * (z + x))
         + 2 * np.sin(a * z + d * x) * np.cos(a * y + d * z) * np.exp(a * (x + y)))
    return -0.5 * a * a * s * np.exp(-2 * nu * d * d * t)
