"""Host-side diagnostics of TwoPhaseBaseAlgorithm<2> (what the reference prints after every time step of its 2D
two-phase tests; not part of the hot path -- the fields are copied from the device once per call):

    compute_bubble_statistics      source/two_phase_base.cc:621-905
        area, perimeter, circularity, mean velocity and centre of mass of the region phi > 0; the interface is located
        by linear interpolation on a (k+3) x (k+3) trapezoidal sub-grid of every cell it crosses

All cells are processed at once with numpy; the cut cells' sub-quadrilaterals go through the reference's case
distinction as boolean masks."""
import numpy as np

from .navier_stokes import gauss_lobatto_points


def _lagrange(nodes, x):
    """values [len(x)][len(nodes)] of the Lagrange basis through `nodes` at the points x"""
    out = np.ones((len(x), len(nodes)))
    for i, xi in enumerate(nodes):
        for j, xj in enumerate(nodes):
            if i != j:
                out[:, i] *= (x - xj) / (xi - xj)
    return out


def _hat(s, x):
    """FE_Q_iso_Q1(s): piecewise linear hat functions on s sub-intervals of [0, 1]"""
    out = np.zeros((len(x), s + 1))
    t = np.asarray(x) * s
    i0 = np.minimum(t.astype(int), s - 1)
    out[np.arange(len(x)), i0] = 1.0 - (t - i0)
    out[np.arange(len(x)), i0 + 1] += t - i0
    return out


def compute_bubble_statistics(mesh, s, k, phi, velocity, sub_refinements=None):
    """phi: level set [n_nodes_ls]; velocity: [n_nodes_u][>= 2] (the engine's three-component layout is fine).
    Returns dict(area, perimeter, circularity, velocity[2], centre[2])."""
    assert mesh.dim == 2, "two_phase_base.cc:621 is the dim = 2 specialisation"
    ncx, ncy = mesh.ncell[0], mesh.ncell[1]
    hx, hy = mesh.h[0], mesh.h[1]
    x0, y0 = mesh.lower[0], mesh.lower[1]
    sub = k + 3 if sub_refinements is None else sub_refinements
    pts = np.linspace(0.0, 1.0, sub + 1)                                  # QIterated(QTrapezoid, sub)
    Sl, Sv = _hat(s, pts), _lagrange(gauss_lobatto_points(k + 1), pts)
    xg, wg = np.polynomial.legendre.leggauss(k)                           # interior_quadrature = QGauss(k)
    xg, wg = 0.5 * (xg + 1.0), 0.5 * wg
    Sg = _lagrange(gauss_lobatto_points(k + 1), xg)
    phi = np.asarray(phi).reshape(s * ncy + 1, s * ncx + 1)
    vel = np.asarray(velocity).reshape(k * ncy + 1, k * ncx + 1, -1)[:, :, :2]
    # local values of every cell: [cy][cx][j][i]
    iy = np.arange(ncy)[:, None] * s + np.arange(s + 1)[None, :]
    ix = np.arange(ncx)[:, None] * s + np.arange(s + 1)[None, :]
    loc = phi[iy[:, None, :, None], ix[None, :, None, :]]
    jy = np.arange(ncy)[:, None] * k + np.arange(k + 1)[None, :]
    jx = np.arange(ncx)[:, None] * k + np.arange(k + 1)[None, :]
    lv = vel[jy[:, None, :, None], jx[None, :, None, :]]                   # [cy][cx][j][i][c]
    flat = loc.reshape(ncy, ncx, -1)
    crosses = np.any(flat[:, :, 1:] * flat[:, :, :1] <= 0, axis=2)        # :668-690
    inside = ~crosses & (flat[:, :, 0] > 0)
    cxs = x0 + hx * np.arange(ncx)
    cys = y0 + hy * np.arange(ncy)
    # ---- cells entirely inside: Gauss(k) quadrature
    w2 = np.outer(wg, wg) * hx * hy                                       # [qy][qx]
    area = float(inside.sum() * w2.sum())
    com = np.array([np.sum(inside * (cxs[None, :] + hx * (wg @ xg))) * hx * hy,
                    np.sum(inside * (cys[:, None] + hy * (wg @ xg))) * hx * hy])
    ug = np.einsum("qj,pi,yxjic->yxqpc", Sg, Sg, lv)                      # (all cells; masked in the sum)
    vsum = np.einsum("yx,qp,yxqpc->c", inside.astype(float), w2, ug)
    perimeter = 0.0
    # ---- cut cells: sub-quadrilaterals [cell][dy][dx] with corners (0: x0y0, 1: x1y0, 2: x0y1, 3: x1y1)
    cy_i, cx_i = np.nonzero(crosses)
    if len(cy_i):
        cval = np.einsum("qj,pi,nji->nqp", Sl, Sl, loc[cy_i, cx_i])       # [n][py][px]
        uval = np.einsum("qj,pi,njic->nqpc", Sv, Sv, lv[cy_i, cx_i])
        px = cxs[cx_i][:, None, None] + hx * pts[None, None, :] + 0.0 * pts[None, :, None]
        py = cys[cy_i][:, None, None] + hy * pts[None, :, None] + 0.0 * pts[None, None, :]
        corner = lambda a, oy, ox: a[:, oy:oy + sub, ox:ox + sub]
        c = np.stack([corner(cval, 0, 0), corner(cval, 0, 1), corner(cval, 1, 0), corner(cval, 1, 1)], axis=-1) + 1e-22
        qx = np.stack([corner(px, 0, 0), corner(px, 0, 1), corner(px, 1, 0), corner(px, 1, 1)], axis=-1)
        qy = np.stack([corner(py, 0, 0), corner(py, 0, 1), corner(py, 1, 0), corner(py, 1, 1)], axis=-1)
        quad = np.stack([qx, qy], axis=-1)                                # [n][dy][dx][corner][2]
        uq = np.stack([corner(uval, 0, 0), corner(uval, 0, 1), corner(uval, 1, 0), corner(uval, 1, 1)], axis=-2)

        def crossing(a, b):
            hit = c[..., a] * c[..., b] <= 0
            with np.errstate(divide="ignore", invalid="ignore"):
                r = np.where(hit, c[..., a] / (c[..., a] - c[..., b]), -1.0)
            pos = quad[..., a, :] + (quad[..., b, :] - quad[..., a, :]) * r[..., None]
            return r, pos
        rx0, px0 = crossing(0, 1)
        rx1, px1 = crossing(2, 3)
        ry0, py0 = crossing(0, 2)
        ry1, py1 = crossing(1, 3)
        local_area = np.ones(c.shape[:-1])
        per = np.zeros(c.shape[:-1])

        def cut(mask, my_area, corner_value, a, b):
            nonlocal local_area, per
            local_area = local_area - np.where(mask, np.where(corner_value < 0, my_area, 1.0 - my_area), 0.0)
            per = per + np.where(mask, np.linalg.norm(a - b, axis=-1), 0.0)
        cut((rx0 > 0) & (ry0 > 0), 0.5 * rx0 * ry0, c[..., 0], px0, py0)
        cut((rx0 > 0) & (ry1 > 0), 0.5 * (1 - rx0) * ry1, c[..., 1], px0, py1)
        cut((rx0 > 0) & (rx1 > 0) & (ry0 < 0) & (ry1 < 0), 0.5 * (rx0 + rx1), c[..., 0], px0, px1)
        cut((rx1 > 0) & (ry0 > 0), 0.5 * rx1 * (1 - ry0), c[..., 2], px1, py0)
        cut((rx1 > 0) & (ry1 > 0), 0.5 * (1 - rx1) * (1 - ry1), c[..., 3], px1, py1)
        cut((ry0 > 0) & (ry1 > 0) & (rx0 < 0) & (rx1 < 0), 0.5 * (ry0 + ry1), c[..., 0], py0, py1)
        local_area = np.where((rx0 <= 0) & (rx1 <= 0) & (ry0 <= 0) & (ry1 <= 0) & (c[..., 0] <= 0), 0.0, local_area)
        my_area = local_area * (hx * hy / (sub * sub) / 4.0)              # JxW * weight_correction of a patch corner
        area += float(4.0 * my_area.sum())
        com += np.einsum("ndx,ndxkc->c", my_area, quad)
        vsum += np.einsum("ndx,ndxkc->c", my_area, uq)
        perimeter = float(per.sum())
    circularity = 2.0 * np.sqrt(area * np.pi) / perimeter if perimeter > 0 else 0.0
    return dict(area=area, perimeter=perimeter, circularity=circularity, velocity=vsum / area if area > 0 else vsum,
                centre=com / area if area > 0 else com)


def format_bubble_statistics(stat, global_omega_diameter):
    """the three lines the reference prints (two_phase_base.cc:866-888, precision 8)"""
    v, c, a = stat["velocity"] * stat["area"], stat["centre"] * stat["area"], stat["area"]
    fmt = lambda x: "%.8g" % x
    vel = [0.0 if abs(x) < 1e-7 * np.linalg.norm(v) else x / a for x in v]
    cen = [0.0 if abs(x) < 1e-7 * global_omega_diameter else x / a for x in c]
    return ["  Degree of circularity: " + fmt(stat["circularity"]),
            "  Mean bubble velocity: " + "".join(fmt(x) + "  " for x in vel),
            "  Position of the center of mass:  " + "".join(fmt(x) + "  " for x in cen)]
