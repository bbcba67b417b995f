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


def test_indexed_mesh_tables_of_an_l_shaped_union():
    """adaflo_amd.IndexedMesh (the tables of adaflo_ctx_create_indexed): node numbering shared between cells, colours
    that share no node, boundary flags on the re-entrant faces too, coordinates of the Gauss-Lobatto lattice"""
    import numpy as np
    import adaflo_amd
    cells = [(i, j, l) for l in range(2) for j in range(3) for i in range(4) if not (i >= 2 and j >= 1)]
    for k in (2, 3):
        m = adaflo_amd.IndexedMesh(cells, [0.25] * 4, [0.3, 0.4, 0.4], [0.5, 0.5], k)
        assert m.n_cells == 16 and m.cell_nodes[k].shape == (16, (k + 1) ** 3) and m.cell_nodes[k - 1].shape == (16, k ** 3)
        # nodes of the union = nodes of the 4 x 3 x 2 lattice minus those strictly inside the removed 2 x 2 x 2 block or on its outer faces
        full = (4 * k + 1) * (3 * k + 1) * (2 * k + 1)
        removed = (2 * k) * (2 * k) * (2 * k + 1)          # x > 2, y > 1 (the faces x = 2 and y = 1 stay)
        assert m.n_nodes(k) == full - removed
        for c in range(len(m.colour_offsets) - 1):
            t = m.cell_nodes[k][m.colour_offsets[c]:m.colour_offsets[c + 1]].reshape(-1)
            assert len(np.unique(t)) == len(t)
        X = m.node_coordinates(k)
        on_boundary = (np.isclose(X[:, 0], 0) | np.isclose(X[:, 1], 0) | np.isclose(X[:, 2], 0) | np.isclose(X[:, 2], 1.0) |
                       (np.isclose(X[:, 0], 1.0)) | (np.isclose(X[:, 1], 1.1)) |
                       (np.isclose(X[:, 0], 0.5) & (X[:, 1] >= 0.3 - 1e-12)) | (np.isclose(X[:, 1], 0.3) & (X[:, 0] >= 0.5 - 1e-12)))
        assert np.array_equal(m.constrained_u.reshape(-1, 3)[:, 0].astype(bool), on_boundary)
        # a node shared by two cells has one number: the first node layer of cell (1, 0, 0) is the last of cell (0, 0, 0)
        pos = {tuple(c): i for i, c in enumerate(m.cells.tolist())}
        a, b = m.cell_nodes[k][pos[(0, 0, 0)]].reshape(k + 1, k + 1, k + 1), m.cell_nodes[k][pos[(1, 0, 0)]].reshape(k + 1, k + 1, k + 1)
        assert np.array_equal(a[:, :, k], b[:, :, 0])
