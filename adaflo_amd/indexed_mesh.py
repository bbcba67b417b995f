"""Index-table description of a Cartesian hexahedral mesh that is not one brick (first alternative): what a
deal.II adapter copies out of MatrixFree / DoFHandler -- per-cell DoF tables, constrained-DoF flags, the diagonal of every
cell's Jacobian, a colouring of the cells -- assembled here from a set of active cells of a rectilinear lattice (an L-shaped
channel, a box with an obstacle).  The reference builds its MatrixFree from whatever the triangulation is
(source/navier_stokes.cc:396-502); include/adaflo_hip.h: adaflo_indexed_desc."""
import numpy as np


class IndexedMesh:
    """active cells `cells` (n x 3 integer lattice coordinates) of the lattice with cell sizes hx[i], hy[j], hz[k].
    Nodes of degree d: the Gauss-Lobatto lattice points touched by an active cell, numbered lexicographically (z slowest)
    among themselves.  Velocity: every component constrained on the whole boundary (faces whose neighbour is inactive);
    pressure: unconstrained.  Colours: the parities of the cell coordinates (two cells of equal parity share no node)."""

    dim = 3

    def __init__(self, cells, hx, hy, hz, velocity_degree, origin=(0.0, 0.0, 0.0)):
        cells = np.asarray(cells, dtype=np.int64).reshape(-1, 3)
        assert len(np.unique(cells, axis=0)) == len(cells), "a cell is listed twice"
        colour = (cells[:, 0] & 1) | ((cells[:, 1] & 1) << 1) | ((cells[:, 2] & 1) << 2)
        order = np.lexsort((cells[:, 0], cells[:, 1], cells[:, 2], colour))      # by colour, lexicographic inside
        self.cells, colour = cells[order], colour[order]
        self.k = int(velocity_degree)
        self.spacing = [np.asarray(h, dtype=np.float64) for h in (hx, hy, hz)]
        self.origin = tuple(float(x) for x in origin)
        used = np.unique(colour)
        self.colour_offsets = np.array([0] + [int(np.sum(colour <= c)) for c in used], dtype=np.int64)
        self.n_cells = len(self.cells)
        self.cell_extents = np.stack([self.spacing[d][self.cells[:, d]] for d in range(3)], axis=1).copy()
        self._active = {tuple(c) for c in self.cells.tolist()}
        self.cell_nodes, self.n_nodes_of, self.node_lattice = {}, {}, {}
        for degree in (self.k, self.k - 1):
            self._number(degree)
        self.constrained_u = np.repeat(self._boundary_nodes(self.k), 3).astype(np.uint8)
        self.constrained_p = np.zeros(self.n_nodes_of[self.k - 1], dtype=np.uint8)

    # -- numbering ----------------------------------------------------------------------------------------------------------
    def _number(self, degree):
        n1 = degree + 1
        loc = np.stack(np.meshgrid(np.arange(n1), np.arange(n1), np.arange(n1), indexing="ij"), axis=-1)  # [k][j][i] -> (k, j, i)
        kji = loc.reshape(-1, 3)                                                   # lexicographic, x fastest
        lat = self.cells[:, None, :] * degree + kji[None, :, ::-1]                 # (cell, local node) -> lattice (I, J, K)
        key = (lat[..., 2] << 42) | (lat[..., 1] << 21) | lat[..., 0]
        uniq, inv = np.unique(key.reshape(-1), return_inverse=True)               # sorted by (K, J, I)
        self.cell_nodes[degree] = inv.reshape(self.n_cells, n1 ** 3).astype(np.int32)
        self.n_nodes_of[degree] = len(uniq)
        self.node_lattice[degree] = np.stack([uniq & ((1 << 21) - 1), (uniq >> 21) & ((1 << 21) - 1), uniq >> 42], axis=1)

    def _boundary_nodes(self, degree):
        """nodes on a face whose neighbouring cell is not active"""
        n1 = degree + 1
        flags = np.zeros(self.n_nodes_of[degree], dtype=bool)
        loc = np.arange(n1 ** 3)
        li = [loc % n1, (loc // n1) % n1, loc // (n1 * n1)]
        for c, cell in enumerate(self.cells.tolist()):
            for axis in range(3):
                for side, step in ((0, -1), (degree, 1)):
                    nb = list(cell)
                    nb[axis] += step
                    if tuple(nb) not in self._active:
                        flags[self.cell_nodes[degree][c, li[axis] == side]] = True
        return flags

    # -- what the adapters and the tests read ----------------------------------------------------------------------------------
    def n_nodes(self, degree):
        return self.n_nodes_of[degree]

    def node_coordinates(self, degree):
        """(n_nodes, 3): FE_Q support points = Gauss-Lobatto points of every cell (navier_stokes.cc:95-106)"""
        from .navier_stokes import gauss_lobatto_points
        gl = np.asarray(gauss_lobatto_points(degree + 1))
        lat = self.node_lattice[degree]
        out = np.empty((len(lat), 3))
        for d in range(3):
            edges = self.origin[d] + np.concatenate([[0.0], np.cumsum(self.spacing[d])])
            cell = np.minimum(lat[:, d] // degree, len(self.spacing[d]) - 1)
            out[:, d] = edges[cell] + self.spacing[d][cell] * gl[lat[:, d] - degree * cell]
        return out
