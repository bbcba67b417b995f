"""Host logic of the mesh description (no GPU): dim = 2 bricks are handed to the engine with a flat third direction."""
import numpy as np

import adaflo_amd
from adaflo_amd.navier_stokes import node_coordinates


def test_two_dimensional_brick_reports_a_flat_third_direction():
    m = adaflo_amd.BrickMesh([40, 80], [0., 0.], [1., 2.])
    assert m.dim == 2 and m.ncell == [40, 80, 1] and m.h == [0.025, 0.025, 1.0] and m.hd == [0.025, 0.025]
    assert m.n_cells == 3200 and m.nodes(2) == [81, 161, 1] and m.n_nodes(2) == 13041 and m.n_nodes(1) == 3321
    x = node_coordinates(m, 2)
    assert x.shape == (13041, 3) and np.all(x[:, 2] == 0.0) and x[:, 0].max() == 1.0 and x[:, 1].max() == 2.0
    m3 = adaflo_amd.BrickMesh([2, 3, 4], [0., 0., 0.], [1., 1., 2.])
    assert m3.dim == 3 and m3.nodes(3) == [7, 10, 13] and m3.hd == m3.h


def test_one_dimensional_brick():
    m = adaflo_amd.BrickMesh([2048], [0.0], [2.5])
    assert m.dim == 1 and m.ncell == [2048, 1, 1] and m.nodes(2) == [4097, 1, 1] and m.n_nodes(1) == 2049
    assert m.hd == [2.5 / 2048] and node_coordinates(m, 2).shape == (4097, 3)
