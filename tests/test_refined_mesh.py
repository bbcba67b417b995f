"""RefinedMesh (adaflo_amd/indexed_mesh.py): the hanging-node constraints of a lattice with cells refined once are the
interpolation from the unrefined side (DoFTools::make_hanging_node_constraints, source/navier_stokes.cc:241-242);
the colouring keeps cells that share a node or a master apart.  CPU only."""
import numpy as np

import adaflo_amd

NCELL, H = (3, 2, 2), (0.5, 0.4, 0.3)
REFINED = [(1, 0, 0), (2, 1, 1)]


def test_the_constraints_interpolate_the_coarse_side():
    """rows sum to one and reproduce a polynomial of the element's degree at the hanging nodes"""
    for k in (2, 3, 4):
        m = adaflo_amd.RefinedMesh(NCELL, H, REFINED, k)
        for degree in (k, k - 1):
            ptr, master, weight = m.hanging[degree]
            X = m.node_coordinates(degree)
            f = X[:, 0] ** degree * X[:, 1] ** degree - 2. * X[:, 2] ** degree + X[:, 0] * X[:, 2]
            hn = m.hanging_nodes(degree)
            assert len(hn) > 0
            got = np.array([np.dot(weight[ptr[i]:ptr[i + 1]], f[master[ptr[i]:ptr[i + 1]]]) for i in range(len(hn))])
            assert np.abs(got - f[hn]).max() < 1e-13
            assert np.abs(np.add.reduceat(weight, ptr[:-1]) - 1.).max() < 1e-13


def test_no_two_cells_of_a_colour_share_a_node_or_a_master():
    m = adaflo_amd.RefinedMesh(NCELL, H, REFINED, 2)
    for degree in (2, 1):
        ptr, master, _ = m.hanging[degree]
        for c in range(len(m.colour_offsets) - 1):
            seen = {}
            for cell in range(m.colour_offsets[c], m.colour_offsets[c + 1]):
                for node in m.cell_nodes[degree][cell]:
                    for n in ([int(node)] if node >= 0 else master[ptr[-1 - node]:ptr[-node]].tolist()):
                        assert seen.setdefault(n, cell) == cell
    assert m.n_cells == 3 * 2 * 2 - 2 + 16
    assert m.constrained_p.sum() == len(m.hanging_nodes(1)) > 0
