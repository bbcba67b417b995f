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


def _lagrange_weights(points, x):
    """values at x of the Lagrange polynomials on `points`"""
    w = np.ones(len(points))
    for i, pi in enumerate(points):
        for j, pj in enumerate(points):
            if i != j:
                w[i] *= (x - pj) / (pi - pj)
    return w


class RefinedMesh:
    """A uniform lattice of `ncell` cells of size `h` in which the cells listed in `refined` are split once into 2 x 2 x 2
    children -- the mesh of the reference's Beltrami driver (tests/beltrami.cc:403-412 refines two cells) -- with the
    hanging-node constraints DoFTools::make_hanging_node_constraints produces (source/navier_stokes.cc:241-242): a node of a
    child that lies on a face or edge of an unrefined neighbour without being one of its nodes takes the neighbour's
    interpolant there.  Same attributes as IndexedMesh plus `hanging[degree] = (ptr, master, weight)`; in `cell_nodes` a
    hanging node appears as -1 - h.  In the vectors every node keeps a number (z, y, x lexicographic by position); hanging
    nodes are flagged constrained in both spaces (MatrixFree::get_constrained_dofs lists them,
    navier_stokes_matrix.cc:157-163), velocity nodes on the boundary as well.  Colours: greedy on the graph of cells
    that share a node or a master."""

    dim = 3

    def __init__(self, ncell, h, refined, velocity_degree, origin=(0.0, 0.0, 0.0)):
        from .navier_stokes import gauss_lobatto_points
        self.k = int(velocity_degree)
        self.ncell, self.h0, self.origin = tuple(int(n) for n in ncell), tuple(float(x) for x in h), tuple(float(x) for x in origin)
        refined = {tuple(int(x) for x in c) for c in refined}
        self._refined = refined
        cells = []                                                  # (lower corner, extent, level, parent lattice cell)
        for kz in range(self.ncell[2]):
            for jy in range(self.ncell[1]):
                for ix in range(self.ncell[0]):
                    lat = (ix, jy, kz)
                    low = np.array([self.origin[d] + lat[d] * self.h0[d] for d in range(3)])
                    if lat in refined:
                        for c in range(8):
                            o = np.array([c & 1, (c >> 1) & 1, c >> 2])
                            cells.append((low + 0.5 * o * np.array(self.h0), 0.5 * np.array(self.h0), 1, lat))
                    else:
                        cells.append((low, np.array(self.h0), 0, lat))
        tables, foot = {}, [set() for _ in cells]
        self.n_nodes_of, self._coords, self.hanging = {}, {}, {}
        for degree in (self.k, self.k - 1):
            gl = np.asarray(gauss_lobatto_points(degree + 1))
            n1 = degree + 1
            loc = np.arange(n1 ** 3)
            li = np.stack([loc % n1, (loc // n1) % n1, loc // (n1 * n1)], axis=1)      # x fastest
            pos = np.stack([low[None, :] + ext[None, :] * gl[li] for low, ext, _, _ in cells])          # (cell, local, 3)
            scale = 1.0 / (min(self.h0) * 1e-9)
            key = np.round((pos - np.array(self.origin)) * scale).astype(np.int64).reshape(-1, 3)
            order = np.lexsort((key[:, 0], key[:, 1], key[:, 2]))
            sk = key[order]
            new = np.ones(len(sk), dtype=bool)
            new[1:] = np.any(sk[1:] != sk[:-1], axis=1)
            ident = np.empty(len(sk), dtype=np.int64)
            ident[order] = np.cumsum(new) - 1
            nodes = ident.reshape(len(cells), n1 ** 3)
            n_nodes = int(ident.max()) + 1
            coords = np.empty((n_nodes, 3))
            coords[ident] = pos.reshape(-1, 3)
            # hanging nodes: a child's node inside the closure of an unrefined cell that is not a node of that cell
            coarse_of = {c[3]: i for i, c in enumerate(cells) if c[2] == 0}
            rows = {}
            for ci, (low, ext, level, lat) in enumerate(cells):
                if level == 0:
                    continue
                for dz in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        for dx in (-1, 0, 1):
                            nb = (lat[0] + dx, lat[1] + dy, lat[2] + dz)
                            if nb not in coarse_of:
                                continue
                            cc = coarse_of[nb]
                            clow, cext = cells[cc][0], cells[cc][1]
                            xi = (pos[ci] - clow[None, :]) / cext[None, :]
                            inside = np.all((xi > -1e-12) & (xi < 1 + 1e-12), axis=1)
                            for l in np.nonzero(inside)[0]:
                                node = int(nodes[ci, l])
                                if node in rows:
                                    continue
                                w = [_lagrange_weights(gl, min(max(xi[l, d], 0.0), 1.0)) for d in range(3)]
                                W = (w[2][:, None, None] * w[1][None, :, None] * w[0][None, None, :]).reshape(-1)
                                nz = np.nonzero(np.abs(W) > 1e-13)[0]
                                if len(nz) == 1:
                                    assert nodes[cc, nz[0]] == node                 # coincides with a node of the neighbour
                                    continue
                                rows[node] = (nodes[cc, nz].astype(np.int64), W[nz])
            hang_nodes = sorted(rows)
            hid = {n: i for i, n in enumerate(hang_nodes)}
            ptr = np.zeros(len(hang_nodes) + 1, dtype=np.int64)
            for i, n in enumerate(hang_nodes):
                assert not any(int(m) in rows for m in rows[n][0]), "a master is hanging: more than one level"
                ptr[i + 1] = ptr[i] + len(rows[n][0])
            master = np.concatenate([rows[n][0] for n in hang_nodes]).astype(np.int32) if hang_nodes else np.zeros(0, np.int32)
            weight = np.concatenate([rows[n][1] for n in hang_nodes]) if hang_nodes else np.zeros(0)
            table = nodes.copy()
            for n, i in hid.items():
                table[nodes == n] = -1 - i
            tables[degree], self.n_nodes_of[degree], self._coords[degree] = table, n_nodes, coords
            self.hanging[degree] = (ptr, master, weight)
            self._hanging_nodes = getattr(self, "_hanging_nodes", {})
            self._hanging_nodes[degree] = np.array(hang_nodes, dtype=np.int64)
            for ci in range(len(cells)):
                for n in nodes[ci]:
                    if int(n) in rows:
                        foot[ci].update((degree, int(m)) for m in rows[int(n)][0])
                    else:
                        foot[ci].add((degree, int(n)))
        # greedy colouring
        colour = np.full(len(cells), -1)
        owner = {}
        for ci in range(len(cells)):
            taken = {colour[o] for f in foot[ci] for o in owner.get(f, ())}
            c = 0
            while c in taken:
                c += 1
            colour[ci] = c
            for f in foot[ci]:
                owner.setdefault(f, []).append(ci)
        order = np.argsort(colour, kind="stable")
        self.n_cells = len(cells)
        self.cell_lower = np.stack([cells[i][0] for i in order])
        self.cell_extents = np.stack([cells[i][1] for i in order]).copy()
        self.cell_level = np.array([cells[i][2] for i in order])
        self.cell_nodes = {deg: tables[deg][order].astype(np.int32) for deg in tables}
        self.colour_offsets = np.array([0] + [int(np.sum(colour <= c)) for c in range(colour.max() + 1)], dtype=np.int64)
        lo = np.array(self.origin)
        hi = lo + np.array(self.ncell) * np.array(self.h0)
        X = self._coords[self.k]
        on_boundary = np.any((np.abs(X - lo) < 1e-12) | (np.abs(X - hi) < 1e-12), axis=1)
        on_boundary[self._hanging_nodes[self.k]] = True
        self.constrained_u = np.repeat(on_boundary, 3).astype(np.uint8)
        self.constrained_p = np.zeros(self.n_nodes_of[self.k - 1], dtype=np.uint8)
        self.constrained_p[self._hanging_nodes[self.k - 1]] = 1

    def n_nodes(self, degree):
        return self.n_nodes_of[degree]

    def node_coordinates(self, degree):
        return self._coords[degree]

    def hanging_nodes(self, degree):
        """numbers (in the vectors) of the hanging nodes, in the order of the constraint rows"""
        return self._hanging_nodes[degree]
