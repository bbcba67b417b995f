// q1_sweep.hip -- structured trilinear (Q1) operator kernel with 2x2x2 Gauss points.
//
// FE_Q_iso_Q1(s) with QIterated(QGauss<1>(2), s) IS a Q1 element with 2-point Gauss quadrature
// on the s-times refined grid (source/level_set_base.cc:58-59, source/two_phase_base.cc:267-268),
// and the Q1 pressure of the Q2/Q1 Taylor-Hood pair with quad_index_p = QGauss(2) is the same
// thing on the cell grid (source/navier_stokes.cc:447-448).  One kernel therefore serves
//   level set : advance_concentration_vmult, reinitialization_vmult, compute_normal_vmult,
//               compute_curvature_vmult           (source/level_set_okz_*.cc)
//   pressure  : pressure_mass_vmult, pressure_poisson_vmult with a cell-wise constant
//               coefficient                       (source/navier_stokes_matrix.cc:1002-1071)
//
// MI355X mapping: a workgroup (256 lanes) owns a column of 16x16 sub-cells and sweeps LZ layers;
// a LANE owns one sub-cell per layer and keeps everything (8 nodal values, 8 Gauss points) in
// registers -- no LDS for the contractions.  Node planes are staged in LDS (each src entry is
// fetched once per workgroup), the 8 local results are combined per owned node (lower-left node
// of the sub-cell) through LDS with compile-time offsets, finished planes leave as row-contiguous
// stores.  Nodes shared between workgroups: the low-rim tile owns the node, the others store
// partial sums in slabs that a small second kernel adds (no atomics, bitwise reproducible).
// The kernel is register-light, so occupancy (not a DMA ring) hides the latency of the
// quadrature-point state stream, which is laid out [tile][layer][12][256 lanes][2 doubles].
#include <type_traits>

#include "basis.hpp"
#include "kernels.hpp"
#include "lds_dma.hpp"

namespace adaflo_hip
{
  namespace
  {
    constexpr int TS  = 16;           // sub-cells per tile edge
    constexpr int TNQ = TS + 1;       // nodes per tile edge
    constexpr int NTQ = TS * TS;      // threads
    constexpr int RIMQ = 4 * TS;      // rim nodes of a tile plane
    constexpr int QSTATE = 12 * NTQ * 2; // doubles of q-state per (tile, layer): 8 points x 3 comps per lane

    struct Q1Args
    {
      int      nsx, nsy, nsz, nnx, nny, nnz, tiles_x, tiles_y, LZ, n_chunks, mode;
      double   ih[3], jxw, ga, gb; // 1/h_sub, h0 h1 h2 / 8, N_0(g_0) = ga, N_1(g_0) = gb
      double   c_mass, c_lap, weight;
      // variable coefficients of the pressure operators (two-phase flow):
      //   coef_cell: per-cell sample of the viscosity, c_mass = 1 / (coef[cell*stride + mid] + shift)
      //              (local_pressure_mass, navier_stokes_matrix.cc:1057-1066)
      //   Q1_LAPLACE_Q3: `state` holds 1 / (weight * rho) per point of the 3x3x3 Gauss rule,
      //              [tile][layer][27][256 lanes] (local_pressure_poisson, :984-1000)
      const double *coef_cell;
      int           coef_stride, coef_mid;
      double        coef_shift, g3x[3], g3w[3];
      uint32_t con;
      double   con_sign;           // constrained rows: dst = (diag ? diag : con_sign) * src
      const double *diag, *src;
      double       *dst;
      const double *state;
      double       *slab, *zslab;
      long          comp_stride, slab_stride, zslab_stride; // blockIdx.y = scalar block (normal vector: 3)
      // Q1_ADVECT_NODAL: velocity [node][3] in `state`; svel[q (2 sub)][KU + 1] 1D shape values of FE_Q(KU) at the Gauss
      // points of the sub-cells; velocity nodes in x, y; edge of the velocity patch of a tile; subdivisions
      const double *svel;
      int           vnx, vny, wn, sub;
    };

    template <int TN>
    __device__ __forceinline__ int rim_index_q(const int i, const int j)
    {
      if (j == 0)
        return i;
      if (j == TN - 1)
        return TN + i;
      if (i == 0)
        return 2 * TN - 1 + j;
      return 3 * TN - 3 + j;
    }

    template <int MODE, int KU = 0>
    // (the nodal advection mode: three workgroups per CU -- 168 registers, 64 B of scratch for k = 2 -- measured 1.46 -> 1.28 ms)
    __global__ __launch_bounds__(NTQ, MODE == Q1_ADVECT_NODAL ? 3 : 1) void q1_sweep_kernel(const Q1Args A)
    {
      __shared__ double pl[2][TNQ * TNQ]; // node planes K, K+1 (ring)
      __shared__ double pub[2][3][NTQ];   // published high faces: [plane lk][(1,0),(0,1),(1,1)][lane]
      constexpr bool NODAL = MODE == Q1_REINIT_NODAL, NODALV = MODE == Q1_ADVECT_NODAL;
      extern __shared__ double W[];       // NODALV: velocity of the tile's node patch at the two Gauss planes [2][3][wn][wn]
      __shared__ double pln[NODAL ? 2 * 3 * TNQ * TNQ : 1]; // planes K, K+1 of the three components of the normal field

      const int  tid = threadIdx.x;
      const int  sx = tid % TS, sy = tid / TS;
      const long nwg = (long)A.tiles_x * A.tiles_y * A.n_chunks;
      const long wg  = xcd_remap(blockIdx.x, nwg);
      const int  bz = (int)(wg % A.n_chunks), bt = (int)(wg / A.n_chunks);
      const int  bx = bt % A.tiles_x, by = bt / A.tiles_x;
      const int  cz0 = bz * A.LZ, nl = min(A.LZ, A.nsz - cz0);
      const int  I0 = TS * bx, J0 = TS * by;
      const int  tcx = min(TS, A.nsx - I0), tcy = min(TS, A.nsy - J0);
      const bool valid = sx < tcx && sy < tcy;
      const bool lastx = valid && sx == tcx - 1, lasty = valid && sy == tcy - 1;
      const bool hasW = sx > 0, hasS = sy > 0;
      const size_t wgs = (size_t)bt * A.n_chunks + bz;
      const double *src_c = A.src + blockIdx.y * A.comp_stride;
      double       *dst_c = A.dst + blockIdx.y * A.comp_stride;
      double       *slab_c = A.slab + blockIdx.y * A.slab_stride, *zslab_c = A.zslab + blockIdx.y * A.zslab_stride;

      // per-lane flags of the up to 4 owned nodes (li,lj) in {0,1}^2, bit li + 2*lj
      unsigned own = 0, con = 0, seam = 0, zero = 0;
      for (int lj = 0; lj < 2; ++lj)
        for (int li = 0; li < 2; ++li)
          {
            const int  bit = li + 2 * lj, I = I0 + sx + li, J = J0 + sy + lj;
            const bool c   = (I == 0 && (A.con >> 0 & 1)) || (I == A.nnx - 1 && (A.con >> 1 & 1)) ||
                           (J == 0 && (A.con >> 2 & 1)) || (J == A.nny - 1 && (A.con >> 3 & 1));
            if (c)
              zero |= 1u << bit;
            if (!(valid && (li == 0 || lastx) && (lj == 0 || lasty)))
              continue;
            own |= 1u << bit;
            if (c)
              con |= 1u << bit;
            // high-rim node shared with another workgroup -> slab (the low-rim tile owns it)
            if ((sx + li == TS && I < A.nnx - 1) || (sy + lj == TS && J < A.nny - 1))
              seam |= 1u << bit;
          }
      const bool     conz_lo = A.con >> 4 & 1, conz_hi = A.con >> 5 & 1;
      const unsigned lane_g  = (unsigned)((J0 + sy) * A.nnx + I0 + sx);

      // Node planes travel global -> registers (issued ONE LAYER AHEAD, unconditionally: an absent node reads element 0 and
      // is replaced by zero when the value is committed) -> LDS.  Loaded straight into LDS behind `in ? load : 0` -- a
      // branch per element, the data needed on the spot -- the planes cost the nodal modes a quarter of their time
      // (reinitialisation operator 1.21 -> 0.93 ms without them, scripts/dev/exp_q1.sh).
      constexpr int NPF = NODAL ? 4 : 1, NPE = (TNQ * TNQ + NTQ - 1) / NTQ;
      double        pf[NPF][NPE];
      auto fetch_plane = [&](const int K) {
#pragma unroll
        for (int r = 0; r < NPE; ++r)
          {
            const int    e = tid + NTQ * r, i = e % TNQ, j = e / TNQ, I = I0 + i, J = J0 + j;
            const bool   in = e < TNQ * TNQ && I < A.nnx && J < A.nny && K < A.nnz;
            const size_t g = in ? ((size_t)K * A.nny + J) * A.nnx + I : 0;
            pf[0][r] = src_c[g];
            if (NODAL)
#pragma unroll
              for (int c = 0; c < 3; ++c)
                pf[NODAL ? 1 + c : 0][r] = A.state[c * A.comp_stride + g];
          }
      };
      auto commit_plane = [&](const int K) {
        double *p = pl[K & 1];
#pragma unroll
        for (int r = 0; r < NPE; ++r)
          {
            const int  e = tid + NTQ * r, i = e % TNQ, j = e / TNQ, I = I0 + i, J = J0 + j;
            const bool in = I < A.nnx && J < A.nny && K < A.nnz;
            if (e < TNQ * TNQ)
              {
                p[e] = in ? pf[0][r] : 0.;
                if (NODAL)
#pragma unroll
                  for (int c = 0; c < 3; ++c)
                    pln[((K & 1) * 3 + c) * (TNQ * TNQ) + e] = in ? pf[NODAL ? 1 + c : 0][r] : 0.;
              }
          }
      };
      // store of one owned node value (constrained rows, slabs, plain stores)
      auto emit = [&](const int bit, const int li, const int lj, const int K, const int lp, double v,
                      const bool zcon, const bool ztop) {
        if (!(own >> bit & 1u))
          return;
        const size_t idx = (size_t)K * A.nny * A.nnx + lane_g + (unsigned)(lj * A.nnx + li);
        if ((con >> bit & 1u) || zcon)
          dst_c[idx] = (A.diag ? A.diag[idx] : A.con_sign) * src_c[idx];
        else if (seam >> bit & 1u)
          slab_c[(wgs * (A.LZ + 1) + lp) * RIMQ + rim_index_q<TNQ>(sx + li, sy + lj)] = v;
        else if (ztop)
          zslab_c[wgs * (TNQ * TNQ) + (sy + lj) * TNQ + sx + li] = v;
        else
          dst_c[idx] = v;
      };

      // advection from the nodal velocity: this lane's cell within the velocity patch of the tile, 1D shape values of the
      // velocity space at its two Gauss points per direction (as q1_rhs_kernel)
      const int wn = A.wn;
      int       vx0 = 0, vy0 = 0, ix0 = 0, jy0 = 0;
      double    Sx[2][KU + 1], Sy[2][KU + 1];
      if (NODALV)
        {
          const int x = min(I0 + sx, A.nsx - 1), y = min(J0 + sy, A.nsy - 1);
          vx0 = KU * (I0 / A.sub);
          vy0 = KU * (J0 / A.sub);
          ix0 = KU * (x / A.sub) - vx0;
          jy0 = KU * (y / A.sub) - vy0;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i <= KU; ++i)
              {
                Sx[a][i] = A.svel[(2 * (x % A.sub) + a) * (KU + 1) + i];
                Sy[a][i] = A.svel[(2 * (y % A.sub) + a) * (KU + 1) + i];
              }
        }
      double carry[4] = {0., 0., 0., 0.}; // top-plane sums of the owned nodes, kept for the next layer
      // (the nodal advection mode is capped at 168 registers for three workgroups per CU: there the two more values in
      // flight cost more in scratch than the prefetch gains -- measured 1.29 vs 1.26 ms --, its planes are loaded on the spot)
#if defined(Q1_SYNC_PLANES)
      constexpr bool PLANES_AHEAD = false;
#else
      constexpr bool PLANES_AHEAD = !NODALV;
#endif
      fetch_plane(cz0);
      commit_plane(cz0);
      if (PLANES_AHEAD)
        fetch_plane(cz0 + 1);
      const double2 *state = reinterpret_cast<const double2 *>(A.state);

      for (int layer = 0; layer < nl; ++layer)
        {
          const int cz = cz0 + layer;
          if (!PLANES_AHEAD)
            fetch_plane(cz + 1);
          commit_plane(cz + 1); // (PLANES_AHEAD: fetched during the previous layer; the slot was last read two layers ago)
#if defined(Q1_EXP_NOLOAD)      // (diagnostic, wrong results: what the plane loads cost)
          if (layer < 0)
#endif
          if (PLANES_AHEAD)
            fetch_plane(cz + 2);
          // quadrature-point state of this sub-cell: issued before the barrier, used after it
          double2 st[12];
          double  cf[27];
          if (MODE == Q1_LAPLACE_Q3)
            {
              const double *cp = A.state + ((size_t)bt * A.nsz + cz) * (27 * NTQ) + tid;
#pragma unroll
              for (int q = 0; q < 27; ++q)
                cf[q] = cp[q * NTQ];
            }
          else if (MODE != Q1_MASS_LAPLACE && !NODAL && !NODALV)
            {
              const double2 *sp = state + ((size_t)bt * A.nsz + cz) * (12 * NTQ) + tid;
#pragma unroll
              for (int c = 0; c < 12; ++c)
                st[c] = sp[c * NTQ];
            }
          if (NODALV)
            {
              // velocity interpolated in z to the two Gauss planes of this layer, on the node patch of the tile (the last
              // barrier of the previous layer is behind every read of W)
              const int     kc = cz / A.sub, zl = cz % A.sub;
              const double *sz = A.svel + 2 * zl * (KU + 1);
              for (int e = tid; e < 3 * wn * wn; e += NTQ)
                {
                  const int comp = e % 3, ix = (e / 3) % wn, jy = e / (3 * wn);
                  // (patch entries beyond the mesh are never read: clamped, so that the loads are unconditional)
                  const int I = min(vx0 + ix, A.vnx - 1), J = min(vy0 + jy, A.vny - 1);
                  double    w0 = 0., w1 = 0.;
#pragma unroll
                    for (int k = 0; k <= KU; ++k)
                      {
                        const double v = A.state[(((size_t)(KU * kc + k) * A.vny + J) * A.vnx + I) * 3 + comp];
                        w0 += sz[k] * v;
                        w1 += sz[KU + 1 + k] * v;
                      }
                  W[(comp * wn + jy) * wn + ix]       = w0;
                  W[((3 + comp) * wn + jy) * wn + ix] = w1;
                }
            }
          __syncthreads();

          // ---- gather + read_dof_values (constrained -> 0) -----------------------------------
          double u[2][2][2];
#pragma unroll
          for (int lk = 0; lk < 2; ++lk)
#pragma unroll
            for (int lj = 0; lj < 2; ++lj)
#pragma unroll
              for (int li = 0; li < 2; ++li)
                {
                  double v = pl[(cz + lk) & 1][(sy + lj) * TNQ + sx + li];
                  const int K = cz + lk;
                  if ((zero >> (li + 2 * lj) & 1u) || (K == 0 && conz_lo) || (K == A.nnz - 1 && conz_hi))
                    v = 0.;
                  u[lk][lj][li] = v;
                }
          double r[2][2][2];
          if (MODE == Q1_LAPLACE_Q3)
            {
              // (grad q, c grad p) with the 3x3x3 Gauss rule and one coefficient per point.  The
              // gradient of a trilinear function is constant along its own direction.
              double DX[2][2], DY[2][2], DZ[2][2]; // [lk][lj], [lk][li], [lj][li]
#pragma unroll
              for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                  {
                    DX[a][b] = (u[a][b][1] - u[a][b][0]) * A.ih[0];
                    DY[a][b] = (u[a][1][b] - u[a][0][b]) * A.ih[1];
                    DZ[a][b] = (u[1][a][b] - u[0][a][b]) * A.ih[2];
                  }
              double RX[2][2] = {{0., 0.}, {0., 0.}}, RY[2][2] = {{0., 0.}, {0., 0.}}, RZ[2][2] = {{0., 0.}, {0., 0.}};
#pragma unroll
              for (int qz = 0; qz < 3; ++qz)
#pragma unroll
                for (int qy = 0; qy < 3; ++qy)
                  {
                    const double z1 = A.g3x[qz], z0 = 1. - z1, y1 = A.g3x[qy], y0 = 1. - y1;
                    // d/dx at (qy, qz); the sum over qx of the coefficient times weight
                    const double gxq = z0 * (y0 * DX[0][0] + y1 * DX[0][1]) + z1 * (y0 * DX[1][0] + y1 * DX[1][1]);
                    double       cx = 0.;
#pragma unroll
                    for (int qx = 0; qx < 3; ++qx)
                      cx += cf[qx + 3 * qy + 9 * qz] * A.g3w[qx];
                    const double tx = gxq * cx * (A.g3w[qy] * A.g3w[qz] * A.jxw * A.ih[0]);
                    RX[0][0] += z0 * y0 * tx;
                    RX[0][1] += z0 * y1 * tx;
                    RX[1][0] += z1 * y0 * tx;
                    RX[1][1] += z1 * y1 * tx;
                  }
#pragma unroll
              for (int qz = 0; qz < 3; ++qz)
#pragma unroll
                for (int qx = 0; qx < 3; ++qx)
                  {
                    const double z1 = A.g3x[qz], z0 = 1. - z1, x1 = A.g3x[qx], x0 = 1. - x1;
                    const double gyq = z0 * (x0 * DY[0][0] + x1 * DY[0][1]) + z1 * (x0 * DY[1][0] + x1 * DY[1][1]);
                    double       cy = 0.;
#pragma unroll
                    for (int qy = 0; qy < 3; ++qy)
                      cy += cf[qx + 3 * qy + 9 * qz] * A.g3w[qy];
                    const double ty = gyq * cy * (A.g3w[qx] * A.g3w[qz] * A.jxw * A.ih[1]);
                    RY[0][0] += z0 * x0 * ty;
                    RY[0][1] += z0 * x1 * ty;
                    RY[1][0] += z1 * x0 * ty;
                    RY[1][1] += z1 * x1 * ty;
                  }
#pragma unroll
              for (int qy = 0; qy < 3; ++qy)
#pragma unroll
                for (int qx = 0; qx < 3; ++qx)
                  {
                    const double y1 = A.g3x[qy], y0 = 1. - y1, x1 = A.g3x[qx], x0 = 1. - x1;
                    const double gzq = y0 * (x0 * DZ[0][0] + x1 * DZ[0][1]) + y1 * (x0 * DZ[1][0] + x1 * DZ[1][1]);
                    double       cz_ = 0.;
#pragma unroll
                    for (int qz = 0; qz < 3; ++qz)
                      cz_ += cf[qx + 3 * qy + 9 * qz] * A.g3w[qz];
                    const double tz = gzq * cz_ * (A.g3w[qx] * A.g3w[qy] * A.jxw * A.ih[2]);
                    RZ[0][0] += y0 * x0 * tz;
                    RZ[0][1] += y0 * x1 * tz;
                    RZ[1][0] += y1 * x0 * tz;
                    RZ[1][1] += y1 * x1 * tz;
                  }
#pragma unroll
              for (int lk = 0; lk < 2; ++lk)
#pragma unroll
                for (int lj = 0; lj < 2; ++lj)
#pragma unroll
                  for (int li = 0; li < 2; ++li)
                    r[lk][lj][li] = (li ? RX[lk][lj] : -RX[lk][lj]) + (lj ? RY[lk][li] : -RY[lk][li]) +
                                    (lk ? RZ[lj][li] : -RZ[lj][li]);
            }
          else
            {
              double c_mass_l = A.c_mass;
              if (MODE == Q1_MASS_LAPLACE && A.coef_cell) // per-cell sample (sub = 1: lane = cell)
                {
                  const long cell = min(I0 + sx, A.nsx - 1) + (long)A.nsx * (min(J0 + sy, A.nsy - 1) + (long)A.nsy * cz);
                  c_mass_l        = 1. / (A.coef_cell[cell * A.coef_stride + A.coef_mid] + A.coef_shift);
                }
          // ---- evaluate at the 2x2x2 Gauss points (trilinear, sum factorised) -----------------
          const double ga = A.ga, gb = A.gb;
          double X[2][2][2], DX[2][2];
#pragma unroll
          for (int lk = 0; lk < 2; ++lk)
#pragma unroll
            for (int lj = 0; lj < 2; ++lj)
              {
                X[lk][lj][0] = ga * u[lk][lj][0] + gb * u[lk][lj][1];
                X[lk][lj][1] = gb * u[lk][lj][0] + ga * u[lk][lj][1];
                DX[lk][lj]   = (u[lk][lj][1] - u[lk][lj][0]) * A.ih[0];
              }
          double XY[2][2][2], DY[2][2], DXY[2][2];
#pragma unroll
          for (int lk = 0; lk < 2; ++lk)
            {
#pragma unroll
              for (int qx = 0; qx < 2; ++qx)
                {
                  XY[lk][0][qx] = ga * X[lk][0][qx] + gb * X[lk][1][qx];
                  XY[lk][1][qx] = gb * X[lk][0][qx] + ga * X[lk][1][qx];
                  DY[lk][qx]    = (X[lk][1][qx] - X[lk][0][qx]) * A.ih[1];
                }
              DXY[lk][0] = ga * DX[lk][0] + gb * DX[lk][1];
              DXY[lk][1] = gb * DX[lk][0] + ga * DX[lk][1];
            }
          // values val[qz][qy][qx]; gradients are constant along their own direction:
          // gx[qz][qy], gy[qz][qx], gz[qy][qx]
          double val[2][2][2], gx[2][2], gy[2][2], gz[2][2];
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              {
                val[0][a][b] = ga * XY[0][a][b] + gb * XY[1][a][b];
                val[1][a][b] = gb * XY[0][a][b] + ga * XY[1][a][b];
                gz[a][b]     = (XY[1][a][b] - XY[0][a][b]) * A.ih[2];
              }
#pragma unroll
          for (int a = 0; a < 2; ++a)
            {
              gx[0][a] = ga * DXY[0][a] + gb * DXY[1][a];
              gx[1][a] = gb * DXY[0][a] + ga * DXY[1][a];
              gy[0][a] = ga * DY[0][a] + gb * DY[1][a];
              gy[1][a] = gb * DY[0][a] + ga * DY[1][a];
            }
          // ---- unit normal at the Gauss points from the nodal field (as q1_rhs_kernel computes evaluated_normal on
          // the first reinitialisation step, level_set_okz_reinitialization.cc:167-172)
          double nq[NODAL ? 24 : 1];
          if (NODAL)
            {
              double nv[3][2][2][2];
#pragma unroll
              for (int c = 0; c < 3; ++c)
                {
                  double w[2][2][2];
#pragma unroll
                  for (int lk = 0; lk < 2; ++lk)
#pragma unroll
                    for (int lj = 0; lj < 2; ++lj)
#pragma unroll
                      for (int li = 0; li < 2; ++li)
                        w[lk][lj][li] = pln[((((cz + lk) & 1) * 3 + c) * TNQ + sy + lj) * TNQ + sx + li];
                  double Xn[2][2][2], XYn[2][2][2];
#pragma unroll
                  for (int lk = 0; lk < 2; ++lk)
#pragma unroll
                    for (int lj = 0; lj < 2; ++lj)
                      {
                        Xn[lk][lj][0] = ga * w[lk][lj][0] + gb * w[lk][lj][1];
                        Xn[lk][lj][1] = gb * w[lk][lj][0] + ga * w[lk][lj][1];
                      }
#pragma unroll
                  for (int lk = 0; lk < 2; ++lk)
#pragma unroll
                    for (int qx = 0; qx < 2; ++qx)
                      {
                        XYn[lk][0][qx] = ga * Xn[lk][0][qx] + gb * Xn[lk][1][qx];
                        XYn[lk][1][qx] = gb * Xn[lk][0][qx] + ga * Xn[lk][1][qx];
                      }
#pragma unroll
                  for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                      {
                        nv[c][0][a][b] = ga * XYn[0][a][b] + gb * XYn[1][a][b];
                        nv[c][1][a][b] = gb * XYn[0][a][b] + ga * XYn[1][a][b];
                      }
                }
#pragma unroll
              for (int q = 0; q < 8; ++q)
                {
                  const int    qx = q & 1, qy = q >> 1 & 1, qz = q >> 2;
                  const double n0 = nv[0][qz][qy][qx], n1 = nv[1][qz][qy][qx], n2 = nv[2][qz][qy][qx];
                  // 1 / max(1e-4, |n|) = rsqrt(max(|n|^2, 1e-8)): hardware estimate + three Newton steps (to the last
                  // bit or two; the f64 square root and division of the literal form cost three times as many
                  // instructions and made this mode VALU-bound)
                  const double x = fmax(n0 * n0 + n1 * n1 + n2 * n2, 1e-8), hx = 0.5 * x;
                  double       sc = __builtin_amdgcn_rsq(x);
#pragma unroll
                  for (int it = 0; it < 3; ++it)
                    sc = sc * (1.5 - hx * sc * sc);
                  nq[3 * q] = n0 * sc, nq[3 * q + 1] = n1 * sc, nq[3 * q + 2] = n2 * sc;
                }
            }
          // ---- velocity at the 8 Gauss points from the z-interpolated node patch (level_set_okz_advance_concentration.cc:389)
          double vq[NODALV ? 24 : 1];
          if (NODALV)
            {
#pragma unroll
              for (int qz = 0; qz < 2; ++qz)
#pragma unroll
                for (int e = 0; e < 3; ++e)
                  {
                    const double *w = W + ((3 * qz + e) * wn + jy0) * wn + ix0;
                    double        tx[2][KU + 1];
#pragma unroll
                    for (int j = 0; j <= KU; ++j)
                      {
                        double a0 = 0., a1 = 0.;
#pragma unroll
                        for (int i = 0; i <= KU; ++i)
                          {
                            const double v = w[j * wn + i];
                            a0 += Sx[0][i] * v;
                            a1 += Sx[1][i] * v;
                          }
                        tx[0][j] = a0, tx[1][j] = a1;
                      }
#pragma unroll
                    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                      for (int qx = 0; qx < 2; ++qx)
                        {
                          double a = 0.;
#pragma unroll
                          for (int j = 0; j <= KU; ++j)
                            a += Sy[qy][j] * tx[qx][j];
                          vq[NODALV ? 3 * (qx + 2 * qy + 4 * qz) + e : 0] = a;
                        }
                  }
            }
          // ---- quadrature-point operation -------------------------------------------------------
          double tv[2][2][2], t0[2][2][2], t1[2][2][2], t2[2][2][2];
#pragma unroll
          for (int qz = 0; qz < 2; ++qz)
#pragma unroll
            for (int qy = 0; qy < 2; ++qy)
#pragma unroll
              for (int qx = 0; qx < 2; ++qx)
                {
                  const double v = val[qz][qy][qx], g0 = gx[qz][qy], g1 = gy[qz][qx], g2 = gz[qy][qx];
                  const int    q = qx + 2 * qy + 4 * qz;
                  double       a = 0., b0 = 0., b1 = 0., b2 = 0.;
                  if (MODE == Q1_MASS_LAPLACE)
                    {
                      a  = c_mass_l * v;
                      b0 = A.c_lap * g0;
                      b1 = A.c_lap * g1;
                      b2 = A.c_lap * g2;
                    }
                  else
                    {
                      // state element 3q+e of this lane: double2 index (3q+e)/2, component (3q+e)&1
                      const double s0 = NODALV ? vq[NODALV ? 3 * q : 0] : (NODAL ? nq[NODAL ? 3 * q : 0] : ((3 * q) & 1 ? st[(3 * q) / 2].y : st[(3 * q) / 2].x));
                      const double s1 = NODALV ? vq[NODALV ? 3 * q + 1 : 0] : (NODAL ? nq[NODAL ? 3 * q + 1 : 0] : ((3 * q + 1) & 1 ? st[(3 * q + 1) / 2].y : st[(3 * q + 1) / 2].x));
                      const double s2 = NODALV ? vq[NODALV ? 3 * q + 2 : 0] : (NODAL ? nq[NODAL ? 3 * q + 2 : 0] : ((3 * q + 2) & 1 ? st[(3 * q + 2) / 2].y : st[(3 * q + 2) / 2].x));
                      if (MODE == Q1_ADVECT || NODALV) // level_set_okz_advance_concentration.cc:244-249
                        a = A.weight * v + s0 * g0 + s1 * g1 + s2 * g2;
                      else // Q1_REINIT: level_set_okz_reinitialization.cc:88-95
                        {
                          const double ng = A.c_lap * (s0 * g0 + s1 * g1 + s2 * g2);
                          a  = A.c_mass * v;
                          b0 = ng * s0;
                          b1 = ng * s1;
                          b2 = ng * s2;
                        }
                    }
                  tv[qz][qy][qx] = a * A.jxw;
                  t0[qz][qy][qx] = b0 * (A.jxw * A.ih[0]);
                  t1[qz][qy][qx] = b1 * (A.jxw * A.ih[1]);
                  t2[qz][qy][qx] = b2 * (A.jxw * A.ih[2]);
                }
          // ---- integrate (transpose of the evaluation) -----------------------------------------
          // z: Z[lk][qy][qx] = sum_qz N_lk(qz) tv + dN_lk t2
          {
            double Zv[2][2][2], Zx[2][2][2], Zy[2][2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b)
                {
                  const double d2 = t2[0][a][b] + t2[1][a][b];
                  Zv[0][a][b] = ga * tv[0][a][b] + gb * tv[1][a][b] - d2;
                  Zv[1][a][b] = gb * tv[0][a][b] + ga * tv[1][a][b] + d2;
                  Zx[0][a][b] = ga * t0[0][a][b] + gb * t0[1][a][b];
                  Zx[1][a][b] = gb * t0[0][a][b] + ga * t0[1][a][b];
                  Zy[0][a][b] = ga * t1[0][a][b] + gb * t1[1][a][b];
                  Zy[1][a][b] = gb * t1[0][a][b] + ga * t1[1][a][b];
                }
            // y
            double Yv[2][2][2], Yx[2][2][2];
#pragma unroll
            for (int lk = 0; lk < 2; ++lk)
#pragma unroll
              for (int b = 0; b < 2; ++b)
                {
                  const double d1 = Zy[lk][0][b] + Zy[lk][1][b];
                  Yv[lk][0][b] = ga * Zv[lk][0][b] + gb * Zv[lk][1][b] - d1;
                  Yv[lk][1][b] = gb * Zv[lk][0][b] + ga * Zv[lk][1][b] + d1;
                  Yx[lk][0][b] = ga * Zx[lk][0][b] + gb * Zx[lk][1][b];
                  Yx[lk][1][b] = gb * Zx[lk][0][b] + ga * Zx[lk][1][b];
                }
            // x
#pragma unroll
            for (int lk = 0; lk < 2; ++lk)
#pragma unroll
              for (int lj = 0; lj < 2; ++lj)
                {
                  const double d0 = Yx[lk][lj][0] + Yx[lk][lj][1];
                  r[lk][lj][0] = ga * Yv[lk][lj][0] + gb * Yv[lk][lj][1] - d0;
                  r[lk][lj][1] = gb * Yv[lk][lj][0] + ga * Yv[lk][lj][1] + d0;
                }
          }
            }
          // ---- publish the high faces, combine per owned node ---------------------------------
#pragma unroll
          for (int lk = 0; lk < 2; ++lk)
            {
              pub[lk][0][tid] = r[lk][0][1];
              pub[lk][1][tid] = r[lk][1][0];
              pub[lk][2][tid] = r[lk][1][1];
            }
          __syncthreads();
#pragma unroll
          for (int lk = 0; lk < 2; ++lk)
            {
              const double w0 = hasW ? pub[lk][0][tid - 1] : 0., w1 = hasW ? pub[lk][2][tid - 1] : 0.;
              const double s0 = hasS ? pub[lk][1][tid - TS] : 0., s1 = hasS ? pub[lk][2][tid - TS] : 0.;
              const double sw = (hasW && hasS) ? pub[lk][2][tid - TS - 1] : 0.;
              double       nv[4];
              nv[0] = r[lk][0][0] + w0 + s0 + sw;
              nv[1] = r[lk][0][1] + s1; // (1,0): far column, last cells in x only
              nv[2] = r[lk][1][0] + w1; // (0,1)
              nv[3] = r[lk][1][1];
              if (lk == 1)
                {
#pragma unroll
                  for (int n = 0; n < 4; ++n)
                    carry[n] = nv[n];
                }
              else
                {
                  const bool zcon = cz == 0 && conz_lo;
#pragma unroll
                  for (int n = 0; n < 4; ++n)
                    emit(n, n & 1, n >> 1, cz, layer, nv[n] + carry[n], zcon, false);
                }
            }
          __syncthreads();
        }
      // ---- top plane of the chunk ---------------------------------------------------------------
      {
        const int  cze  = cz0 + nl;
        const bool ztop = cze < A.nsz, zcon = cze == A.nnz - 1 && conz_hi;
#pragma unroll
        for (int n = 0; n < 4; ++n)
          emit(n, n & 1, n >> 1, cze, nl, carry[n], zcon, ztop);
      }
    }

    // second pass: dst[low-rim node] += partial sums of the other sharers (and of the chunk below)
    __global__ __launch_bounds__(64) void q1_fixup_kernel(const Q1Args A0, const long n1, const long n2)
    {
      Q1Args A = A0;
      A.dst += blockIdx.y * A.comp_stride;
      A.slab += blockIdx.y * A.slab_stride;
      A.zslab += blockIdx.y * A.zslab_stride;
      for (long b = blockIdx.x; b < n1 + n2; b += gridDim.x)
        {
          if (b < n1)
            {
              const long bt = b / A.nnz;
              const int  K = (int)(b % A.nnz), bx = (int)(bt % A.tiles_x), by = (int)(bt / A.tiles_x);
              const int  ppc = A.LZ + 1, c_hi = min(K / A.LZ, A.n_chunks - 1), lp = K - A.LZ * c_hi;
              const bool zb = lp == 0 && c_hi > 0;
              const bool zc = (K == 0 && (A.con >> 4 & 1)) || (K == A.nnz - 1 && (A.con >> 5 & 1));
              // (straight-line: every load unconditional with a clamped address and issued before the first use, absent
              // terms deselected afterwards -- the loop over the sharers with its loads behind branches was a chain of three
              // to five dependent memory round trips per node; same order of the additions)
              for (int e = threadIdx.x; e < 2 * TNQ - 1; e += 64)
                {
                  const int  i = e < TNQ ? e : 0, j = e < TNQ ? 0 : e - TNQ + 1;
                  const int  I = TS * bx + i, J = TS * by + j;
                  const bool inside = I < A.nnx && J < A.nny;
                  const bool seam_x = i == 0 && I > 0, seam_y = j == 0 && J > 0;
                  const bool con = zc || (I == 0 && (A.con >> 0 & 1)) || (I == A.nnx - 1 && (A.con >> 1 & 1)) ||
                                   (J == 0 && (A.con >> 2 & 1)) || (J == A.nny - 1 && (A.con >> 3 & 1));
                  const bool act = inside && (seam_x || seam_y) && !((i == TS && I < A.nnx - 1) || (j == TS && J < A.nny - 1)) && !con;
                  // the sharers (dx, dy) = (1, 0), (0, 1), (1, 1); an absent one reads from this tile, entry 0
                  const bool has[3] = {seam_x, seam_y, seam_x && seam_y};
                  double     hi[3], lo[3];
#pragma unroll
                  for (int q = 0; q < 3; ++q)
                    {
                      const int  dx = q != 1 ? 1 : 0, dy = q != 0 ? 1 : 0;
                      const bool ok = act && has[q];
                      const long tb = ok ? (long)(by - dy) * A.tiles_x + bx - dx : bt;
                      const int  r  = ok ? rim_index_q<TNQ>(i + TS * dx, j + TS * dy) : 0;
                      hi[q] = A.slab[((tb * A.n_chunks + c_hi) * ppc + lp) * RIMQ + r];
                      lo[q] = A.slab[((tb * A.n_chunks + (zb ? c_hi - 1 : c_hi)) * ppc + (zb ? A.LZ : lp)) * RIMQ + r];
                    }
                  const double zs = A.zslab[(bt * A.n_chunks + (zb ? c_hi - 1 : 0)) * (TNQ * TNQ) + j * TNQ + i];
                  const size_t idx = ((size_t)K * A.nny + min(J, A.nny - 1)) * A.nnx + min(I, A.nnx - 1);
                  const double old = A.dst[idx];
                  double       sum = 0.;
#pragma unroll
                  for (int q = 0; q < 3; ++q)
                    {
                      sum += has[q] ? hi[q] : 0.;
                      sum += has[q] && zb ? lo[q] : 0.;
                    }
                  sum += zb ? zs : 0.;
                  if (act)
                    A.dst[idx] = old + sum;
                }
            }
          else
            {
              const long bb = b - n1, bt = bb / (A.n_chunks - 1);
              const int  m = (int)(bb % (A.n_chunks - 1)) + 1, K = A.LZ * m;
              const int  bx = (int)(bt % A.tiles_x), by = (int)(bt / A.tiles_x);
              for (int e = threadIdx.x; e < TNQ * TNQ; e += 64)
                {
                  const int i = e % TNQ, j = e / TNQ, I = TS * bx + i, J = TS * by + j;
                  if (I >= A.nnx || J >= A.nny)
                    continue;
                  const bool seam = (i == 0 && I > 0) || (i == TS && I < A.nnx - 1) || (j == 0 && J > 0) ||
                                    (j == TS && J < A.nny - 1);
                  const bool c = (I == 0 && (A.con >> 0 & 1)) || (I == A.nnx - 1 && (A.con >> 1 & 1)) ||
                                 (J == 0 && (A.con >> 2 & 1)) || (J == A.nny - 1 && (A.con >> 3 & 1));
                  if (seam || c)
                    continue;
                  A.dst[((size_t)K * A.nny + J) * A.nnx + I] += A.zslab[(bt * A.n_chunks + m - 1) * (TNQ * TNQ) + e];
                }
            }
        }
    }

    // generic [cell][3][q (2s)^3] -> [tile][layer][12][256][2]: lane = sub-cell, 24 doubles
    // (8 Gauss points x 3 components, index 3*q_local + e)
    __global__ __launch_bounds__(256) void q1_convert_state_kernel(double *__restrict__ out,
                                                                   const double *__restrict__ canon,
                                                                   const int s, const int ncx,
                                                                   const int ncy, const int nsx,
                                                                   const int nsy, const int nsz,
                                                                   const int tiles_x, const long total)
    {
      const int nq1 = 2 * s, nqc = nq1 * nq1 * nq1;
      for (long o = blockIdx.x * 256L + threadIdx.x; o < total; o += (long)gridDim.x * 256)
        {
          const int j = (int)(o & 1);
          long      r = o >> 1;
          const int lane = (int)(r % NTQ);
          r /= NTQ;
          const int c2 = (int)(r % 12);
          r /= 12;
          const int  z  = (int)(r % nsz);
          const long bt = r / nsz;
          const int  bx = (int)(bt % tiles_x), by = (int)(bt / tiles_x);
          const int  x = bx * TS + lane % TS, y = by * TS + lane / TS;
          const int  k = 2 * c2 + j, ql = k / 3, e = k % 3; // local Gauss point, component
          double     v = 0.;
          if (x < nsx && y < nsy)
            {
              const int  cx = x / s, cy = y / s, cz = z / s;
              const int  qx = 2 * (x % s) + (ql & 1), qy = 2 * (y % s) + (ql >> 1 & 1), qz = 2 * (z % s) + (ql >> 2);
              const long cell = cx + (long)ncx * (cy + (long)ncy * cz);
              v = canon[(cell * 3 + e) * nqc + qx + nq1 * (qy + nq1 * qz)];
            }
          out[o] = v;
        }
    }

    // sweep layout -> generic [cell][3][q]: the inverse of q1_convert_state_kernel (one thread per
    // generic entry), for adaflo_ls_get_evaluated_* and the generic operator kernels
    __global__ __launch_bounds__(256) void q1_unconvert_state_kernel(double *__restrict__ canon,
                                                                     const double *__restrict__ sweep,
                                                                     const int s, const int ncx,
                                                                     const int ncy, const int nsz,
                                                                     const int tiles_x, const long total)
    {
      const int nq1 = 2 * s, nqc = nq1 * nq1 * nq1;
      for (long o = blockIdx.x * 256L + threadIdx.x; o < total; o += (long)gridDim.x * 256)
        {
          const int  q = (int)(o % nqc), e = (int)((o / nqc) % 3);
          const long cell = o / (3L * nqc);
          const int  qx = q % nq1, qy = (q / nq1) % nq1, qz = q / (nq1 * nq1);
          const int  cx = (int)(cell % ncx), cy = (int)((cell / ncx) % ncy), cz = (int)(cell / ((long)ncx * ncy));
          const int  x = cx * s + qx / 2, y = cy * s + qy / 2, z = cz * s + qz / 2;
          const int  k = 3 * ((qx & 1) + 2 * (qy & 1) + 4 * (qz & 1)) + e;
          const long bt = (long)(y / TS) * tiles_x + x / TS;
          const int  lane = (y % TS) * TS + x % TS;
          canon[o] = sweep[(((bt * nsz + z) * 12 + k / 2) * NTQ + lane) * 2 + (k & 1)];
        }
    }

    // ---------------------------------------------------------------------------------------------
    // Right-hand sides of the reinitialisation and the advection equation on the sweep structure
    //   Q1RHS_REINIT  LevelSetOKZSolverReinitialization::local_reinitialize_rhs (reinitialization.cc:128-189)
    //                 dst += (grad w, n (1/2 (1 - phi^2) - diffusion n . grad phi))   [or -diffusion grad phi];
    //                 on the first step the normal at the Gauss points is interpolated from the nodal
    //                 normal field, normalised and WRITTEN as quadrature-point state in sweep layout
    //   Q1RHS_ADVECT  LevelSetOKZSolverAdvanceConcentration::local_advance_concentration_rhs
    //                 (advance_concentration.cc:288-397 without stabilisation)
    //                 dst += -(w, weight phi + u . grad phi + weight_old phi_old [+ weight_old_old phi_old_old]);
    //                 u = the FE_Q(KU) velocity at the level-set Gauss points, written as state
    // Same tile / lane / seam scheme as q1_sweep_kernel; like the cell loops they replace the kernels
    // read plainly, ADD into dst and skip constrained rows.  HBM traffic per sub-cell: the nodal
    // fields once, dst read + write, and the 192 B of quadrature-point state.
    enum
    {
      Q1RHS_REINIT = 0,
      Q1RHS_ADVECT = 1
    };

    struct Q1RhsArgs
    {
      Q1Args        q;
      const double *f[4];   // nodal fields staged per plane (reinit: phi, n_0..n_2; advect: phi, phi_old, phi_old_old)
      int           nf, flag, sub; // reinit: bit 0 diffuse_only, bit 1 first step; advect: use_old_old
      double        diffusion, weight, weight_old, weight_old_old;
      double       *state;  // [tile][layer][12][256][2]; advection: nullptr = evaluated_convection is not written
      int           state_only; // advection: write `state` and nothing else (no fields staged, no sums)
      const double *vel, *svel; // velocity [node][3]; svel[q (2 sub)][KU + 1]
      int           vnx, vny, wn; // velocity nodes in x, y; edge of the velocity patch of a tile
    };

    // trilinear evaluation at the 2x2x2 Gauss points: val[qz][qy][qx], gx[qz][qy], gy[qz][qx], gz[qy][qx]
    __device__ __forceinline__ void q1_evaluate(const double (&u)[2][2][2], const double ga, const double gb,
                                                const double (&ih)[3], double (&val)[2][2][2], double (&gx)[2][2],
                                                double (&gy)[2][2], double (&gz)[2][2])
    {
      double X[2][2][2], DX[2][2];
#pragma unroll
      for (int lk = 0; lk < 2; ++lk)
#pragma unroll
        for (int lj = 0; lj < 2; ++lj)
          {
            X[lk][lj][0] = ga * u[lk][lj][0] + gb * u[lk][lj][1];
            X[lk][lj][1] = gb * u[lk][lj][0] + ga * u[lk][lj][1];
            DX[lk][lj]   = (u[lk][lj][1] - u[lk][lj][0]) * ih[0];
          }
      double XY[2][2][2], DY[2][2], DXY[2][2];
#pragma unroll
      for (int lk = 0; lk < 2; ++lk)
        {
#pragma unroll
          for (int qx = 0; qx < 2; ++qx)
            {
              XY[lk][0][qx] = ga * X[lk][0][qx] + gb * X[lk][1][qx];
              XY[lk][1][qx] = gb * X[lk][0][qx] + ga * X[lk][1][qx];
              DY[lk][qx]    = (X[lk][1][qx] - X[lk][0][qx]) * ih[1];
            }
          DXY[lk][0] = ga * DX[lk][0] + gb * DX[lk][1];
          DXY[lk][1] = gb * DX[lk][0] + ga * DX[lk][1];
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          {
            val[0][a][b] = ga * XY[0][a][b] + gb * XY[1][a][b];
            val[1][a][b] = gb * XY[0][a][b] + ga * XY[1][a][b];
            gz[a][b]     = (XY[1][a][b] - XY[0][a][b]) * ih[2];
          }
#pragma unroll
      for (int a = 0; a < 2; ++a)
        {
          gx[0][a] = ga * DXY[0][a] + gb * DXY[1][a];
          gx[1][a] = gb * DXY[0][a] + ga * DXY[1][a];
          gy[0][a] = ga * DY[0][a] + gb * DY[1][a];
          gy[1][a] = gb * DY[0][a] + ga * DY[1][a];
        }
    }

    // transpose of q1_evaluate: r[lk][lj][li] = sum_q N tv + dN_x t0 + dN_y t1 + dN_z t2
    // (t0..t2 already carry the factor 1/h of their direction)
    __device__ __forceinline__ void q1_integrate(const double (&tv)[2][2][2], const double (&t0)[2][2][2],
                                                 const double (&t1)[2][2][2], const double (&t2)[2][2][2],
                                                 const double ga, const double gb, double (&r)[2][2][2])
    {
      double Zv[2][2][2], Zx[2][2][2], Zy[2][2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          {
            const double d2 = t2[0][a][b] + t2[1][a][b];
            Zv[0][a][b] = ga * tv[0][a][b] + gb * tv[1][a][b] - d2;
            Zv[1][a][b] = gb * tv[0][a][b] + ga * tv[1][a][b] + d2;
            Zx[0][a][b] = ga * t0[0][a][b] + gb * t0[1][a][b];
            Zx[1][a][b] = gb * t0[0][a][b] + ga * t0[1][a][b];
            Zy[0][a][b] = ga * t1[0][a][b] + gb * t1[1][a][b];
            Zy[1][a][b] = gb * t1[0][a][b] + ga * t1[1][a][b];
          }
      double Yv[2][2][2], Yx[2][2][2];
#pragma unroll
      for (int lk = 0; lk < 2; ++lk)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          {
            const double d1 = Zy[lk][0][b] + Zy[lk][1][b];
            Yv[lk][0][b] = ga * Zv[lk][0][b] + gb * Zv[lk][1][b] - d1;
            Yv[lk][1][b] = gb * Zv[lk][0][b] + ga * Zv[lk][1][b] + d1;
            Yx[lk][0][b] = ga * Zx[lk][0][b] + gb * Zx[lk][1][b];
            Yx[lk][1][b] = gb * Zx[lk][0][b] + ga * Zx[lk][1][b];
          }
#pragma unroll
      for (int lk = 0; lk < 2; ++lk)
#pragma unroll
        for (int lj = 0; lj < 2; ++lj)
          {
            const double d0 = Yx[lk][lj][0] + Yx[lk][lj][1];
            r[lk][lj][0] = ga * Yv[lk][lj][0] + gb * Yv[lk][lj][1] - d0;
            r[lk][lj][1] = gb * Yv[lk][lj][0] + ga * Yv[lk][lj][1] + d0;
          }
    }

    // KU: velocity degree (advection) / 1 = first reinitialisation step (normal field staged), 0 = later steps
    template <int MODE, int KU>
    __global__ __launch_bounds__(NTQ, 2) void q1_rhs_kernel(const Q1RhsArgs R)
    {
      constexpr int NF = MODE == Q1RHS_ADVECT ? 3 : (KU ? 4 : 1), PL = TNQ * TNQ;
      extern __shared__ double dyn[];
      double *pl = dyn, *pub = pl + 2 * NF * PL, *W = pub + 6 * NTQ; // [2][NF][PL], [2][3][NTQ], [2][3][wn][wn]
      const Q1Args &A = R.q;

      const int  tid = threadIdx.x;
      const int  sx = tid % TS, sy = tid / TS;
      const long nwg = (long)A.tiles_x * A.tiles_y * A.n_chunks;
      const long wg  = xcd_remap(blockIdx.x, nwg);
      const int  bz = (int)(wg % A.n_chunks), bt = (int)(wg / A.n_chunks);
      const int  bx = bt % A.tiles_x, by = bt / A.tiles_x;
      const int  cz0 = bz * A.LZ, nl = min(A.LZ, A.nsz - cz0);
      const int  I0 = TS * bx, J0 = TS * by;
      const int  tcx = min(TS, A.nsx - I0), tcy = min(TS, A.nsy - J0);
      const bool valid = sx < tcx && sy < tcy;
      const bool lastx = valid && sx == tcx - 1, lasty = valid && sy == tcy - 1;
      const bool hasW = sx > 0, hasS = sy > 0;
      const size_t wgs = (size_t)bt * A.n_chunks + bz;

      unsigned own = 0, con = 0, seam = 0;
      for (int lj = 0; lj < 2; ++lj)
        for (int li = 0; li < 2; ++li)
          {
            const int bit = li + 2 * lj, I = I0 + sx + li, J = J0 + sy + lj;
            if (!(valid && (li == 0 || lastx) && (lj == 0 || lasty)))
              continue;
            own |= 1u << bit;
            if ((I == 0 && (A.con >> 0 & 1)) || (I == A.nnx - 1 && (A.con >> 1 & 1)) || (J == 0 && (A.con >> 2 & 1)) ||
                (J == A.nny - 1 && (A.con >> 3 & 1)))
              con |= 1u << bit;
            if ((sx + li == TS && I < A.nnx - 1) || (sy + lj == TS && J < A.nny - 1))
              seam |= 1u << bit;
          }
      const bool     conz_lo = A.con >> 4 & 1, conz_hi = A.con >> 5 & 1;
      const unsigned lane_g  = (unsigned)((J0 + sy) * A.nnx + I0 + sx);

      // node planes: global -> registers (issued one layer ahead) -> LDS (after the layer's reads)
      constexpr int NLD = (PL + NTQ - 1) / NTQ;
      double        pre[NF][NLD];
      // (unconditional loads: an absent node reads element 0 and becomes zero when the plane is committed -- behind
      // `in ? load : 0` every element was a branch with a wait at its join)
      auto fetch_plane = [&](const int K) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
          if (f < R.nf)
#pragma unroll
            for (int r = 0; r < NLD; ++r)
              {
                const int    e = tid + NTQ * r, i = e % TNQ, j = e / TNQ, I = I0 + i, J = J0 + j;
                const bool   in = e < PL && I < A.nnx && J < A.nny && K < A.nnz;
                pre[f][r]       = R.f[f][in ? ((size_t)K * A.nny + J) * A.nnx + I : 0];
              }
      };
      auto commit_plane = [&](const int K) {
        const int slot = K & 1;
#pragma unroll
        for (int f = 0; f < NF; ++f)
          if (f < R.nf)
#pragma unroll
            for (int r = 0; r < NLD; ++r)
              {
                const int  e = tid + NTQ * r, i = e % TNQ, j = e / TNQ, I = I0 + i, J = J0 + j;
                const bool in = I < A.nnx && J < A.nny && K < A.nnz;
                if (e < PL)
                  pl[(slot * NF + f) * PL + e] = in ? pre[f][r] : 0.;
              }
      };
      // (the owner adds into dst: the old values are fetched at the top of the layer)
      double dold[4];
      auto emit = [&](const int bit, const int li, const int lj, const int K, const int lp, const double v,
                      const bool zcon, const bool ztop, const bool prefetched) {
        if (!(own >> bit & 1u) || (con >> bit & 1u) || zcon)
          return;
        const size_t idx = (size_t)K * A.nny * A.nnx + lane_g + (unsigned)(lj * A.nnx + li);
        if (seam >> bit & 1u)
          A.slab[(wgs * (A.LZ + 1) + lp) * RIMQ + rim_index_q<TNQ>(sx + li, sy + lj)] = v;
        else if (ztop)
          A.zslab[wgs * (TNQ * TNQ) + (sy + lj) * TNQ + sx + li] = v;
        else
          A.dst[idx] = (prefetched ? dold[bit] : A.dst[idx]) + v;
      };
      auto nodal = [&](const int cz, const int f, double (&u)[2][2][2]) {
#pragma unroll
        for (int lk = 0; lk < 2; ++lk)
#pragma unroll
          for (int lj = 0; lj < 2; ++lj)
#pragma unroll
            for (int li = 0; li < 2; ++li)
              u[lk][lj][li] = pl[(((cz + lk) & 1) * NF + f) * PL + (sy + lj) * TNQ + sx + li];
      };

      // advection: this lane's cell within the velocity patch of the tile, 1D shape values of the
      // velocity space at its two Gauss points per direction
      const int sub = R.sub, wn = R.wn;
      int       vx0 = 0, vy0 = 0, ix0 = 0, jy0 = 0;
      double    Sx[2][KU + 1], Sy[2][KU + 1];
      if (MODE == Q1RHS_ADVECT)
        {
          const int x = min(I0 + sx, A.nsx - 1), y = min(J0 + sy, A.nsy - 1);
          vx0 = KU * (I0 / sub);
          vy0 = KU * (J0 / sub);
          ix0 = KU * (x / sub) - vx0;
          jy0 = KU * (y / sub) - vy0;
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i <= KU; ++i)
              {
                Sx[a][i] = R.svel[(2 * (x % sub) + a) * (KU + 1) + i];
                Sy[a][i] = R.svel[(2 * (y % sub) + a) * (KU + 1) + i];
              }
        }
      const bool diffuse_only = MODE == Q1RHS_REINIT && (R.flag & 1), first = MODE == Q1RHS_REINIT && KU == 1;

      double carry[4] = {0., 0., 0., 0.};
      fetch_plane(cz0);
      commit_plane(cz0);
      fetch_plane(cz0 + 1);
      commit_plane(cz0 + 1);
      __syncthreads();
      for (int layer = 0; layer < nl; ++layer)
        {
          const int cz = cz0 + layer;
          if (layer + 1 < nl)
            fetch_plane(cz + 2);
#pragma unroll
          for (int n = 0; n < 4; ++n)
            {
              dold[n] = 0.;
              if (((own & ~con & ~seam) >> n & 1u) && !R.state_only)
                dold[n] = A.dst[(size_t)cz * A.nny * A.nnx + lane_g + (unsigned)((n >> 1) * A.nnx + (n & 1))];
            }
          double2  st[12];
          double2 *sp = reinterpret_cast<double2 *>(R.state) + ((size_t)bt * A.nsz + cz) * (12 * NTQ) + tid;
          if (MODE == Q1RHS_REINIT && !diffuse_only && !first)
            {
#pragma unroll
              for (int c = 0; c < 12; ++c)
                st[c] = sp[c * NTQ];
            }
          if (MODE == Q1RHS_ADVECT)
            {
              // velocity interpolated in z to the two Gauss planes of this layer, on the node patch of the tile
              const int     kc = cz / sub, zl = cz % sub;
              const double *sz = R.svel + 2 * zl * (KU + 1);
              for (int e = tid; e < 3 * wn * wn; e += NTQ)
                {
                  const int comp = e % 3, ix = (e / 3) % wn, jy = e / (3 * wn);
                  // (patch entries beyond the mesh are never read: clamped, so that the loads are unconditional)
                  const int I = min(vx0 + ix, R.vnx - 1), J = min(vy0 + jy, R.vny - 1);
                  double    w0 = 0., w1 = 0.;
#pragma unroll
                    for (int k = 0; k <= KU; ++k)
                      {
                        const double v = R.vel[(((size_t)(KU * kc + k) * R.vny + J) * R.vnx + I) * 3 + comp];
                        w0 += sz[k] * v;
                        w1 += sz[KU + 1 + k] * v;
                      }
                  W[(comp * wn + jy) * wn + ix]       = w0;
                  W[((3 + comp) * wn + jy) * wn + ix] = w1;
                }
              __syncthreads();
            }

          double u[2][2][2], val[2][2][2], gx[2][2], gy[2][2], gz[2][2];
          nodal(cz, 0, u);
          q1_evaluate(u, A.ga, A.gb, A.ih, val, gx, gy, gz);
          double el[24]; // quadrature-point state of this sub-cell, element 3 q + e
          double tv[2][2][2], t0[2][2][2], t1[2][2][2], t2[2][2][2];
          if (MODE == Q1RHS_REINIT)
            {
              if (KU == 1)
                {
                  double dx[2][2], dy[2][2], dz[2][2], nv[3][2][2][2];
#pragma unroll
                  for (int e = 0; e < 3; ++e)
                    {
                      nodal(cz, NF == 4 ? 1 + e : 0, u);
                      q1_evaluate(u, A.ga, A.gb, A.ih, nv[e], dx, dy, dz);
                    }
#pragma unroll
                  for (int q = 0; q < 8; ++q)
                    {
                      const int    qx = q & 1, qy = q >> 1 & 1, qz = q >> 2;
                      const double n0 = nv[0][qz][qy][qx], n1 = nv[1][qz][qy][qx], n2 = nv[2][qz][qy][qx];
                      const double sc = 1. / fmax(1e-4, sqrt(n0 * n0 + n1 * n1 + n2 * n2)); // :167-172
                      el[3 * q] = n0 * sc, el[3 * q + 1] = n1 * sc, el[3 * q + 2] = n2 * sc;
                    }
#pragma unroll
                  for (int c = 0; c < 12; ++c)
                    sp[c * NTQ] = make_double2(el[2 * c], el[2 * c + 1]);
                }
              else if (!diffuse_only)
                {
#pragma unroll
                  for (int c = 0; c < 12; ++c)
                    el[2 * c] = st[c].x, el[2 * c + 1] = st[c].y;
                }
#pragma unroll
              for (int q = 0; q < 8; ++q)
                {
                  const int    qx = q & 1, qy = q >> 1 & 1, qz = q >> 2;
                  const double g0 = gx[qz][qy], g1 = gy[qz][qx], g2 = gz[qy][qx];
                  double       b0, b1, b2;
                  if (!diffuse_only) // :176-178
                    {
                      const double v = val[qz][qy][qx], n0 = el[3 * q], n1 = el[3 * q + 1], n2 = el[3 * q + 2];
                      const double f = 0.5 * (1. - v * v) - (n0 * g0 + n1 * g1 + n2 * g2) * R.diffusion;
                      b0 = n0 * f, b1 = n1 * f, b2 = n2 * f;
                    }
                  else
                    b0 = -R.diffusion * g0, b1 = -R.diffusion * g1, b2 = -R.diffusion * g2;
                  tv[qz][qy][qx] = 0.;
                  t0[qz][qy][qx] = b0 * (A.jxw * A.ih[0]);
                  t1[qz][qy][qx] = b1 * (A.jxw * A.ih[1]);
                  t2[qz][qy][qx] = b2 * (A.jxw * A.ih[2]);
                }
            }
          else
            {
              // velocity at the 8 Gauss points from the z-interpolated node patch
#pragma unroll
              for (int qz = 0; qz < 2; ++qz)
#pragma unroll
                for (int e = 0; e < 3; ++e)
                  {
                    const double *w = W + ((3 * qz + e) * wn + jy0) * wn + ix0;
                    double        tx[2][KU + 1];
#pragma unroll
                    for (int j = 0; j <= KU; ++j)
                      {
                        double a0 = 0., a1 = 0.;
#pragma unroll
                        for (int i = 0; i <= KU; ++i)
                          {
                            const double v = w[j * wn + i];
                            a0 += Sx[0][i] * v;
                            a1 += Sx[1][i] * v;
                          }
                        tx[0][j] = a0, tx[1][j] = a1;
                      }
#pragma unroll
                    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
                      for (int qx = 0; qx < 2; ++qx)
                        {
                          double a = 0.;
#pragma unroll
                          for (int j = 0; j <= KU; ++j)
                            a += Sy[qy][j] * tx[qx][j];
                          el[3 * (qx + 2 * qy + 4 * qz) + e] = a;
                        }
                  }
              if (R.state)
#pragma unroll
                for (int c = 0; c < 12; ++c)
                  sp[c * NTQ] = make_double2(el[2 * c], el[2 * c + 1]); // :389 evaluated_convection
              if (R.state_only)
                {
                  __syncthreads(); // (W is overwritten at the top of the next layer)
                  continue;
                }
              double vo[2][2][2], voo[2][2][2], dx[2][2], dy[2][2], dz[2][2];
              nodal(cz, 1, u);
              q1_evaluate(u, A.ga, A.gb, A.ih, vo, dx, dy, dz);
              if (R.flag) // bdf_2 && step_no > 1  :375-378
                {
                  nodal(cz, 2, u);
                  q1_evaluate(u, A.ga, A.gb, A.ih, voo, dx, dy, dz);
                }
#pragma unroll
              for (int q = 0; q < 8; ++q)
                {
                  const int qx = q & 1, qy = q >> 1 & 1, qz = q >> 2;
                  double    old_value = R.weight_old * vo[qz][qy][qx];
                  if (R.flag)
                    old_value += R.weight_old_old * voo[qz][qy][qx];
                  const double ug = el[3 * q] * gx[qz][qy] + el[3 * q + 1] * gy[qz][qx] + el[3 * q + 2] * gz[qy][qx];
                  tv[qz][qy][qx] = -(val[qz][qy][qx] * R.weight + ug + old_value) * A.jxw;
                  t0[qz][qy][qx] = t1[qz][qy][qx] = t2[qz][qy][qx] = 0.;
                }
            }
          double r[2][2][2];
          q1_integrate(tv, t0, t1, t2, A.ga, A.gb, r);

          // ---- publish the high faces, combine per owned node (as in q1_sweep_kernel) ----------
#pragma unroll
          for (int lk = 0; lk < 2; ++lk)
            {
              pub[(lk * 3 + 0) * NTQ + tid] = r[lk][0][1];
              pub[(lk * 3 + 1) * NTQ + tid] = r[lk][1][0];
              pub[(lk * 3 + 2) * NTQ + tid] = r[lk][1][1];
            }
          __syncthreads();
          if (layer + 1 < nl) // every lane has read planes cz, cz + 1: plane cz + 2 takes the slot of cz
            commit_plane(cz + 2);
#pragma unroll
          for (int lk = 0; lk < 2; ++lk)
            {
              const double *pb = pub + lk * 3 * NTQ;
              const double  w0 = hasW ? pb[tid - 1] : 0., w1 = hasW ? pb[2 * NTQ + tid - 1] : 0.;
              const double  s0 = hasS ? pb[NTQ + tid - TS] : 0., s1 = hasS ? pb[2 * NTQ + tid - TS] : 0.;
              const double  sw = (hasW && hasS) ? pb[2 * NTQ + tid - TS - 1] : 0.;
              double        nv[4];
              nv[0] = r[lk][0][0] + w0 + s0 + sw;
              nv[1] = r[lk][0][1] + s1;
              nv[2] = r[lk][1][0] + w1;
              nv[3] = r[lk][1][1];
              if (lk == 1)
                {
#pragma unroll
                  for (int n = 0; n < 4; ++n)
                    carry[n] = nv[n];
                }
              else
                {
                  const bool zcon = cz == 0 && conz_lo;
#pragma unroll
                  for (int n = 0; n < 4; ++n)
                    emit(n, n & 1, n >> 1, cz, layer, nv[n] + carry[n], zcon, false, true);
                }
            }
          __syncthreads();
        }
      if (!(MODE == Q1RHS_ADVECT && R.state_only))
        {
          const int  cze  = cz0 + nl;
          const bool ztop = cze < A.nsz, zcon = cze == A.nnz - 1 && conz_hi;
#pragma unroll
          for (int n = 0; n < 4; ++n)
            emit(n, n & 1, n >> 1, cze, nl, carry[n], zcon, ztop, false);
        }
    }
  } // namespace

  namespace
  {
    // generic [cell][27] density -> 1 / (weight rho) in the lane layout [tile][layer][27][256]
    __global__ __launch_bounds__(256) void q1_convert_poisson_coef_kernel(double *__restrict__ out,
                                                                          const double *__restrict__ rho,
                                                                          const double weight, const int ncx,
                                                                          const int ncy, const int ncz,
                                                                          const int tiles_x, const long total)
    {
      for (long o = blockIdx.x * 256L + threadIdx.x; o < total; o += (long)gridDim.x * 256)
        {
          const int lane = (int)(o % NTQ);
          long      r    = o / NTQ;
          const int q    = (int)(r % 27);
          r /= 27;
          const int  z  = (int)(r % ncz);
          const long bt = r / ncz;
          const int  x = (int)(bt % tiles_x) * TS + lane % TS, y = (int)(bt / tiles_x) * TS + lane / TS;
          double     v = 0.;
          if (x < ncx && y < ncy)
            v = 1. / (weight * rho[(x + (long)ncx * (y + (long)ncy * z)) * 27 + q]);
          out[o] = v;
        }
    }

    // -------------------------------------------------------------------------------------------
    // Constant-coefficient  c_mass (v, u) + c_lap (grad v, grad u)  for trilinear elements on the
    // uniform refined grid is the tensor-product 27-point stencil
    //   c_mass Mx My Mz + c_lap (Kx My Mz + Mx Ky Mz + Mx My Kz),
    // M = h/6 [1 4 1], K = 1/h [-1 2 -1] (half rows at the domain boundary): the 2-point Gauss rule
    // of the cell loop integrates these products exactly.  Pure gather: no seams, no fix-up pass.
    // A thread owns one (i, j) node column of a z-chunk and marches in z with the in-plane sums
    //   A_k = Mx My u_k,  B_k = (Kx My + Mx Ky) u_k  of three planes in registers:
    //   dst_k = sum_dz (c_mass Mz + c_lap Kz)[dz] A_{k+dz} + c_lap Mz[dz] B_{k+dz}.
    // HBM traffic: src once (+2/LZ halo planes) + dst once = 16 B per node.
    struct StencilArgs
    {
      int           nnx, nny, nnz, LZ, n_chunks, blocks_per_plane;
      long          plane, comp_stride, flat;               // flat = bands of rows x nnx (see the kernel)
      double        m_off[3], m_ctr[3], k_off[3], k_ctr[3]; // h/6, h/3, -1/h, 1/h per direction
      double        c_mass, c_lap, con_sign;
      uint32_t      con;
      const double *diag, *src;
      double       *dst;
      double       *dot_partial; // optional: per block (sum of src_i dst_i, 0) for the CG driver (krylov.hip)
      int           plain;       // right-hand side form: src read plainly, dst += result, constrained rows skipped
    };

    // No LDS staging, no barrier: a lane owns FSR vertically adjacent nodes of a plane and marches in
    // z.  With  n = lo + hi (number of neighbours in the mesh), S(u) = [lo'] u_- + [hi'] u_+  (a
    // neighbour on a constrained face, or outside the mesh, does not count) the 1D rows are
    //   M u = h/6 (2 n u_0 + S(u)),   K u = 1/h (n u_0 - S(u)),
    // so that y costs 5 operations per node for both (t1 = 2 n u_0 + S, t2 = n u_0 - S), z 8 (the scales
    // h/6, 1/h of all directions are folded into six plane-uniform coefficients) and x 5:
    //   P = hx/6 [c_mass My Mz + c_lap (Ky Mz + My Kz)] u,   Q = c_lap / hx My Mz u,
    //   dst = n_x (2 P + Q) + [lo'] (P - Q)_west + [hi'] (P - Q)_east,
    // i.e. ONE value per node crosses lanes (ds_bpermute, the LDS crossbar is idle otherwise) --
    // 18 f64 operations per node instead of the 45 of the version that summed the nine in-plane
    // neighbours per plane out of an LDS tile (0.154 -> see DESIGN.md 4.4 for the measurements: that
    // version was bound by its own instruction stream, with and without its loads).  The lanes of a
    // wave are 62 consecutive positions of the FLATTENED band space [band of FSR rows][x] plus one
    // halo lane on either side: neighbours in x are neighbours in the flat index, the row ends need no
    // care (their weights are zero), and no lane idles on meshes with 2^m + 1 nodes per row.  A lane
    // reads FSR + 2 values per plane (one plane ahead), 8-byte coalesced.
    constexpr int FSW = 62; // owned lanes of a wave
#ifndef Q1_STENCIL_ROWS
#define Q1_STENCIL_ROWS 3
#endif
    constexpr int FSR = Q1_STENCIL_ROWS;
    __global__ __launch_bounds__(256) void q1_stencil_kernel(const StencilArgs A)
    {
      using lds_dma::load_now;
      const long nwg   = (long)A.blocks_per_plane * A.n_chunks;
      const long wg    = xcd_remap(blockIdx.x, nwg);
      const int  chunk = (int)(wg / A.blocks_per_plane);
      const int  lane  = threadIdx.x & 63;
      const long g_raw = ((wg % A.blocks_per_plane) * 4 + (threadIdx.x >> 6)) * FSW + lane - 1;
      const long g     = min(max(g_raw, 0L), A.flat - 1);
      const int  band = (int)(g / A.nnx), i = (int)(g - (long)band * A.nnx);
      const bool own  = lane >= 1 && lane <= FSW && g_raw < A.flat;
      const double *__restrict__ src_c = A.src + blockIdx.y * A.comp_stride;
      double *__restrict__       dst_c = A.dst + blockIdx.y * A.comp_stride;

      // n, [lo'], [hi'] of a 1D row
      auto row = [&](const int d, const int idx, const int n, double &cnt, double &wl, double &wh) {
        const bool lo = idx > 0, hi = idx < n - 1;
        wl  = lo && (A.plain || !(idx - 1 == 0 && (A.con >> (2 * d) & 1))) ? 1. : 0.;
        wh  = hi && (A.plain || !(idx + 1 == n - 1 && (A.con >> (2 * d + 1) & 1))) ? 1. : 0.;
        cnt = (lo ? 1. : 0.) + (hi ? 1. : 0.);
      };
      double nx, wxl, wxh, ny[FSR], wyl[FSR], wyh[FSR];
      bool   active[FSR], con_xy[FSR];
      unsigned off[FSR + 2]; // byte offsets of rows j-1 .. j+FSR of this lane's column in a plane (clamped to the mesh)
      row(0, i, A.nnx, nx, wxl, wxh);
#pragma unroll
      for (int r = 0; r < FSR; ++r)
        {
          const int j_raw = band * FSR + r, j = min(j_raw, A.nny - 1);
          row(1, j, A.nny, ny[r], wyl[r], wyh[r]);
          active[r] = own && j_raw < A.nny;
          con_xy[r] = (i == 0 && (A.con >> 0 & 1)) || (i == A.nnx - 1 && (A.con >> 1 & 1)) ||
                      (j == 0 && (A.con >> 2 & 1)) || (j == A.nny - 1 && (A.con >> 3 & 1));
        }
#pragma unroll
      for (int r = 0; r < FSR + 2; ++r)
        off[r] = (unsigned)(min(max(band * FSR + r - 1, 0), A.nny - 1) * A.nnx + i) * 8u;

      // plane-uniform coefficients of  P' = a_off (t1_- + t1_+) + n_z a_ctr t1_0 + b_off (t2_- + t2_+) + n_z b_ctr t2_0
      // and  Q' = q_off (t1_- + t1_+) + n_z q_ctr t1_0  (planes that do not count are zero in t1, t2)
      const double sx = A.m_off[0], sy = A.m_off[1], mz = A.m_off[2], kz = A.k_ctr[2]; // h/6, h/6, h/6, 1/h
      const double a_off = sx * sy * (A.c_mass * mz - A.c_lap * kz), a_ctr = sx * sy * (2. * A.c_mass * mz + A.c_lap * kz); // (n_z = 1)
      const double b_off = sx * A.c_lap * A.k_ctr[1] * mz, b_ctr = 2. * b_off;
      const double q_off = A.c_lap * A.k_ctr[0] * sy * mz, q_ctr = 2. * q_off;

      const int  k0 = chunk * A.LZ, k1 = min(k0 + A.LZ, A.nnz);
      const int  kb = max(k0 - 1, 0), ke = min(k1, A.nnz - 1); // planes read
      const bool conz_lo = (A.con >> 4 & 1) && !A.plain, conz_hi = (A.con >> 5 & 1) && !A.plain; // (neighbour planes)
      // y-sums of three planes and the values of four (the one being finished, the one being summed
      // and two on their way), rotating: the loop is unrolled twelve times so that nothing is copied
      // between registers (a copy of a value just loaded would wait for it), and the read two planes
      // ahead is unconditional (a load behind a branch makes the compiler assume at the join that it
      // may be the youngest one, i.e. wait for everything)
      double t1[3][FSR], t2[3][FSR], u[4][FSR + 2], src_dot_dst = 0.;
#pragma unroll
      for (int r = 0; r < FSR; ++r)
        t1[0][r] = t2[0][r] = t1[1][r] = t2[1][r] = t1[2][r] = t2[2][r] = 0.;
#pragma unroll
      for (int r = 0; r < FSR + 2; ++r)
        u[3][r] = 0.;
      auto read_plane = [&](const int k, double *v) {
        const char *s = reinterpret_cast<const char *>(src_c + (long)k * A.plane);
#pragma unroll
        for (int r = 0; r < FSR + 2; ++r)
          v[r] = *reinterpret_cast<const double *>(s + off[r]);
      };
      read_plane(kb, u[0]);
      read_plane(min(kb + 1, ke), u[1]);
      // plane q is summed in y (its values are in u[PH % 4]), plane k = q - 1 is finished
      auto step = [&](auto phase, const int q) {
        constexpr int PH = decltype(phase)::value, nw = PH % 3, c3 = (PH + 2) % 3, od = (PH + 1) % 3;
        constexpr int un = PH % 4, uc = (PH + 3) % 4;
        read_plane(min(q + 2, ke), u[(PH + 2) % 4]);
        if (q > ke || (q == 0 && conz_lo) || (q == A.nnz - 1 && conz_hi)) // (block-uniform)
          {
#pragma unroll
            for (int r = 0; r < FSR; ++r)
              {
                t1[nw][r] = 0.; // (same order of the assignments in both arms: the compiler merges their
                t2[nw][r] = 0.; //  tails, and with different orders the merged store gets a run-time index)
              }
          }
        else
          {
#pragma unroll
            for (int r = 0; r < FSR; ++r)
              {
                const double c = ny[r] * u[un][r + 1], sum = wyl[r] * u[un][r] + wyh[r] * u[un][r + 2];
                t1[nw][r] = 2. * c + sum;
                t2[nw][r] = c - sum;
              }
          }
        const int k = q - 1;
        if (k >= k0)
          {
            const bool   edge = k == 0 || k == A.nnz - 1; // one neighbour plane (n_z = 1)
            const double a_c = edge ? a_ctr : 2. * a_ctr, b_c = edge ? b_ctr : 2. * b_ctr, q_c = edge ? q_ctr : 2. * q_ctr;
            const bool   conz = (k == 0 && (A.con >> 4 & 1)) || (k == A.nnz - 1 && (A.con >> 5 & 1));
            char        *d    = reinterpret_cast<char *>(dst_c + (long)k * A.plane);
#pragma unroll
            for (int r = 0; r < FSR; ++r)
              {
                const double s1 = t1[od][r] + t1[nw][r], s2 = t2[od][r] + t2[nw][r];
                const double P = (a_off * s1 + a_c * t1[c3][r]) + (b_off * s2 + b_c * t2[c3][r]);
                const double Q = q_off * s1 + q_c * t1[c3][r];
                const double T = P - Q;
                const double Tw = __shfl_up(T, 1, 64), Te = __shfl_down(T, 1, 64);
                double       v  = nx * (2. * P + Q) + (wxl * Tw + wxh * Te);
                const double c0 = u[uc][r + 1];
                double      *dp = reinterpret_cast<double *>(d + off[r + 1]);
                if (active[r] && A.plain)
                  {
                    if (!(con_xy[r] || conz))
                      *dp += v;
                  }
                else if (active[r])
                  {
                    if (con_xy[r] || conz)
                      v = (A.diag ? load_now(A.diag + (long)k * A.plane + off[r + 1] / 8) : A.con_sign) * c0;
                    __builtin_nontemporal_store(v, dp);
                    src_dot_dst += c0 * v;
                  }
              }
          }
      };
      for (int q = kb; q <= k1; q += 12)
        {
          step(std::integral_constant<int, 0>(), q);
#define Q1_STENCIL_STEP(ph)                                                                                           \
  if (q + ph <= k1)                                                                                                   \
    step(std::integral_constant<int, ph>(), q + ph);
          Q1_STENCIL_STEP(1)
          Q1_STENCIL_STEP(2)
          Q1_STENCIL_STEP(3)
          Q1_STENCIL_STEP(4)
          Q1_STENCIL_STEP(5)
          Q1_STENCIL_STEP(6)
          Q1_STENCIL_STEP(7)
          Q1_STENCIL_STEP(8)
          Q1_STENCIL_STEP(9)
          Q1_STENCIL_STEP(10)
          Q1_STENCIL_STEP(11)
#undef Q1_STENCIL_STEP
        }
      if (A.dot_partial) // p . A p of the CG iteration for free: the centre value is in a register anyway
        {
          __shared__ double red[4];
          double            v = src_dot_dst;
          for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, 64);
          if ((threadIdx.x & 63) == 0)
            red[threadIdx.x >> 6] = v;
          __syncthreads();
          if (threadIdx.x == 0)
            {
              const long b = (long)blockIdx.y * gridDim.x + blockIdx.x;
              A.dot_partial[2 * b]     = (red[0] + red[1]) + (red[2] + red[3]);
              A.dot_partial[2 * b + 1] = 0.;
            }
        }
    }

    // -------------------------------------------------------------------------------------------
    // Right-hand sides of the normal and curvature projections as tensor-product stencils (same
    // marching scheme as q1_stencil_kernel), with C = int N_i N_j' = 1/2 [-1 0 1] (half rows at the
    // boundary):
    //   MODE 0  LevelSetOKZSolverComputeNormal::local_compute_normal_rhs (compute_normal.cc:141-153)
    //           dst_d += (w, d_d phi):       dst_0 = Cx My Mz phi, dst_1 = Mx Cy Mz phi, dst_2 = Mx My Cz phi
    //   MODE 1  LevelSetOKZSolverComputeCurvature::local_compute_curvature_rhs (:229-259)
    //           dst  += -(w, div n~),  n~ = n / |n| at the nodes (0 where |n| <= 1e-2)
    // Like the cell loops they replace, the kernels ADD into dst and skip constrained rows.
    struct StencilRhsArgs
    {
      int           nnx, nny, nnz, LZ, n_chunks, blocks_per_plane;
      long          plane, comp_stride;
      double        m_off[3], m_ctr[3];
      uint32_t      con;
      const double *src;
      double       *dst;
    };

    template <int MODE>
    __global__ __launch_bounds__(256) void q1_stencil_rhs_kernel(const StencilRhsArgs A)
    {
      const long nwg   = (long)A.blocks_per_plane * A.n_chunks;
      const long wg    = xcd_remap(blockIdx.x, nwg);
      const int  chunk = (int)(wg / A.blocks_per_plane);
      const long p_raw = (wg % A.blocks_per_plane) * 256 + threadIdx.x;
      const bool active = p_raw < A.plane;
      const long p      = active ? p_raw : A.plane - 1;
      const int  i = (int)(p % A.nnx), j = (int)(p / A.nnx);

      auto rows = [&](const int d, const int idx, const int n, double *m, double *c, long *off, const long stride) {
        const bool lo = idx > 0, hi = idx < n - 1;
        m[0] = lo ? A.m_off[d] : 0.;
        m[2] = hi ? A.m_off[d] : 0.;
        m[1] = ((lo ? 1. : 0.) + (hi ? 1. : 0.)) * A.m_ctr[d];
        c[0] = lo ? -0.5 : 0.;
        c[2] = hi ? 0.5 : 0.;
        c[1] = (lo ? 0.5 : 0.) + (hi ? -0.5 : 0.);
        off[0] = lo ? -stride : 0;
        off[1] = 0;
        off[2] = hi ? stride : 0;
      };
      double mx[3], cx[3], my[3], cy[3];
      long   ox[3], oy[3];
      rows(0, i, A.nnx, mx, cx, ox, 1);
      rows(1, j, A.nny, my, cy, oy, A.nnx);
      const bool con_xy = (i == 0 && (A.con >> 0 & 1)) || (i == A.nnx - 1 && (A.con >> 1 & 1)) ||
                          (j == 0 && (A.con >> 2 & 1)) || (j == A.nny - 1 && (A.con >> 3 & 1));

      // in-plane sums of plane k:  P = Mx My f,  Qx = Cx My f,  Qy = Mx Cy f
      //   MODE 0: f = phi for all three;  MODE 1: P of n~_2, Qx of n~_0, Qy of n~_1
      auto plane_sums = [&](const int k, double &P, double &Qx, double &Qy) {
        const double *s = A.src + (long)k * A.plane + p;
        double        a[3], bx[3], ay[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
          {
            const double *r = s + oy[dy];
            if (MODE == 0)
              {
                const double v0 = r[ox[0]], v1 = r[0], v2 = r[ox[2]];
                a[dy]  = mx[0] * v0 + mx[1] * v1 + mx[2] * v2;
                bx[dy] = cx[0] * v0 + cx[1] * v1 + cx[2] * v2;
                ay[dy] = a[dy];
              }
            else
              {
                double f0[3], f1[3], f2[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                  {
                    const long   o  = dx == 1 ? 0 : ox[dx];
                    const double n0 = r[o], n1 = r[o + A.comp_stride], n2 = r[o + 2 * A.comp_stride];
                    const double nr = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
                    const double sc = nr > 1e-2 ? 1. / nr : 0.;
                    f0[dx] = n0 * sc;
                    f1[dx] = n1 * sc;
                    f2[dx] = n2 * sc;
                  }
                a[dy]  = mx[0] * f2[0] + mx[1] * f2[1] + mx[2] * f2[2];
                bx[dy] = cx[0] * f0[0] + cx[1] * f0[1] + cx[2] * f0[2];
                ay[dy] = mx[0] * f1[0] + mx[1] * f1[1] + mx[2] * f1[2];
              }
          }
        P  = my[0] * a[0] + my[1] * a[1] + my[2] * a[2];
        Qx = my[0] * bx[0] + my[1] * bx[1] + my[2] * bx[2];
        Qy = cy[0] * ay[0] + cy[1] * ay[1] + cy[2] * ay[2];
      };

      const int  k0 = chunk * A.LZ, k1 = min(k0 + A.LZ, A.nnz);
      const bool conz_lo = A.con >> 4 & 1, conz_hi = A.con >> 5 & 1;
      double     Pm = 0., Qxm = 0., Qym = 0., P0, Qx0, Qy0, Pp = 0., Qxp = 0., Qyp = 0.;
      if (k0 > 0)
        plane_sums(k0 - 1, Pm, Qxm, Qym);
      plane_sums(k0, P0, Qx0, Qy0);
      for (int k = k0; k < k1; ++k)
        {
          const bool hi = k < A.nnz - 1, lo = k > 0;
          if (hi)
            plane_sums(k + 1, Pp, Qxp, Qyp);
          else
            Pp = Qxp = Qyp = 0.;
          const double nz  = (lo ? 1. : 0.) + (hi ? 1. : 0.);
          const double mzl = lo ? A.m_off[2] : 0., mzc = nz * A.m_ctr[2], mzh = hi ? A.m_off[2] : 0.;
          const double czl = lo ? -0.5 : 0., czc = (lo ? 0.5 : 0.) + (hi ? -0.5 : 0.), czh = hi ? 0.5 : 0.;
          const double rx = mzl * Qxm + mzc * Qx0 + mzh * Qxp;
          const double ry = mzl * Qym + mzc * Qy0 + mzh * Qyp;
          const double rz = czl * Pm + czc * P0 + czh * Pp;
          const bool   conz = (k == 0 && conz_lo) || (k == A.nnz - 1 && conz_hi);
          if (active && !(con_xy || conz))
            {
              const long idx = (long)k * A.plane + p;
              if (MODE == 0)
                {
                  A.dst[idx] += rx;
                  A.dst[idx + A.comp_stride] += ry;
                  A.dst[idx + 2 * A.comp_stride] += rz;
                }
              else
                A.dst[idx] -= rx + ry + rz;
            }
          Pm = P0, Qxm = Qx0, Qym = Qy0, P0 = Pp, Qx0 = Qxp, Qy0 = Qyp;
        }
    }
  } // namespace

  // ---------------------------------------------------------------------------------------------
  // host side
  // ---------------------------------------------------------------------------------------------
  int q1_convert_poisson_coef(adaflo_ctx *ctx, DeviceBuffer &out, const double *rho_generic, const double weight)
  {
    const int    ncx = ctx->desc.ncell[0], ncy = ctx->desc.ncell[1], ncz = ctx->desc.ncell[2];
    const int    tiles_x = (ncx + TS - 1) / TS, tiles_y = (ncy + TS - 1) / TS;
    const size_t count   = (size_t)tiles_x * tiles_y * ncz * 27 * NTQ;
    if (out.count != count)
      {
        if (out.p)
          (void)hipFree(out.p);
        out.p     = nullptr;
        out.count = 0;
        if (hipMalloc(&out.p, count * sizeof(double)) != hipSuccess)
          return ADAFLO_ENOMEM;
        out.count = count;
      }
    long nb = (long)((count + 255) / 256);
    if (nb > 256 * 64)
      nb = 256 * 64;
    hipLaunchKernelGGL(q1_convert_poisson_coef_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, out.p,
                       rho_generic, weight, ncx, ncy, ncz, tiles_x, (long)count);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  // quadrature-point state in sweep layout: [tile][layer][12][256 lanes][2]
  int q1_state_alloc(adaflo_ctx *ctx, DeviceBuffer &out)
  {
    const int    s = ctx->s, nsx = s * ctx->desc.ncell[0], nsy = s * ctx->desc.ncell[1], nsz = s * ctx->desc.ncell[2];
    const int    tiles_x = (nsx + TS - 1) / TS, tiles_y = (nsy + TS - 1) / TS;
    const size_t count   = (size_t)tiles_x * tiles_y * nsz * QSTATE;
    if (out.count != count)
      {
        if (out.p)
          (void)hipFree(out.p);
        out.p     = nullptr;
        out.count = 0;
        if (hipMalloc(&out.p, count * sizeof(double)) != hipSuccess)
          return ADAFLO_ENOMEM;
        out.count = count;
      }
    return 0;
  }

  int q1_convert_state(adaflo_ctx *ctx, DeviceBuffer &out, const double *canonical_dev)
  {
    if (const int e = q1_state_alloc(ctx, out))
      return e;
    const int    s = ctx->s, nsx = s * ctx->desc.ncell[0], nsy = s * ctx->desc.ncell[1], nsz = s * ctx->desc.ncell[2];
    const int    tiles_x = (nsx + TS - 1) / TS;
    const size_t count   = out.count;
    long nb = (long)((count + 255) / 256);
    if (nb > 256 * 64)
      nb = 256 * 64;
    hipLaunchKernelGGL(q1_convert_state_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, out.p,
                       canonical_dev, s, ctx->desc.ncell[0], ctx->desc.ncell[1], nsx, nsy, nsz, tiles_x,
                       (long)count);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  // sweep layout -> generic [cell][3][q]
  int q1_unconvert_state(adaflo_ctx *ctx, double *generic_dev, const DeviceBuffer &sweep)
  {
    const int  s = ctx->s, nsx = s * ctx->desc.ncell[0], nsz = s * ctx->desc.ncell[2];
    const int  tiles_x = (nsx + TS - 1) / TS;
    const long total   = (long)ctx->n_cells * 3 * 8 * s * s * s;
    long       nb      = (total + 255) / 256;
    if (nb > 256 * 64)
      nb = 256 * 64;
    hipLaunchKernelGGL(q1_unconvert_state_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, generic_dev, sweep.p,
                       s, ctx->desc.ncell[0], ctx->desc.ncell[1], nsz, tiles_x, total);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  // geometry of the sweep (tiles, z-chunks, Jacobian factors) and the slab buffers of the seams
  static int q1_setup(adaflo_ctx *ctx, const int sub, const int mode, const int n_blocks, Q1Args &A)
  {
    {
      const Quadrature1D g3 = gauss(3);
      for (int q = 0; q < 3; ++q)
        {
          A.g3x[q] = g3.x[q];
          A.g3w[q] = g3.w[q];
        }
    }
    A.nsx = sub * ctx->desc.ncell[0];
    A.nsy = sub * ctx->desc.ncell[1];
    A.nsz = sub * ctx->desc.ncell[2];
    A.nnx = A.nsx + 1;
    A.nny = A.nsy + 1;
    A.nnz = A.nsz + 1;
    A.tiles_x = (A.nsx + TS - 1) / TS;
    A.tiles_y = (A.nsy + TS - 1) / TS;
    {
      const long tiles = (long)A.tiles_x * A.tiles_y;
      int        lz    = 32;
      while (lz > 4 && tiles * ((A.nsz + lz - 1) / lz) < 2048)
        lz /= 2;
      if (lz > A.nsz)
        lz = A.nsz;
      A.LZ       = lz;
      A.n_chunks = (A.nsz + lz - 1) / lz;
    }
    A.mode = mode;
    double det = 1.;
    for (int d = 0; d < 3; ++d)
      {
        const double hs = ctx->desc.h[d] / sub;
        A.ih[d]         = 1. / hs;
        det *= hs;
      }
    A.jxw      = mode == Q1_LAPLACE_Q3 ? det : det / 8.; // (2-point rule: weights 1/2 per direction)
    A.gb       = 0.5 * (1. - 1. / std::sqrt(3.)); // first Gauss point of QGauss<1>(2) on [0,1]
    A.ga       = 1. - A.gb;
    const size_t n_wg = (size_t)A.tiles_x * A.tiles_y * A.n_chunks;
    A.comp_stride  = (long)A.nnx * A.nny * A.nnz;
    A.slab_stride  = (long)(n_wg * (A.LZ + 1) * RIMQ);
    A.zslab_stride = (long)(n_wg * TNQ * TNQ);
    const size_t need[2] = {(size_t)A.slab_stride * n_blocks, (size_t)A.zslab_stride * n_blocks};
    DeviceBuffer *buf[2] = {&ctx->q1_slab, &ctx->q1_zslab};
    for (int i = 0; i < 2; ++i)
      if (buf[i]->count < need[i])
        {
          if (buf[i]->p)
            (void)hipFree(buf[i]->p);
          buf[i]->p     = nullptr;
          buf[i]->count = 0;
          if (hipMalloc(&buf[i]->p, need[i] * sizeof(double)) != hipSuccess)
            return ADAFLO_ENOMEM;
          buf[i]->count = need[i];
        }
    A.slab  = ctx->q1_slab.p;
    A.zslab = ctx->q1_zslab.p;
    return 0;
  }

  // constant-coefficient c_mass M + c_lap K as 27-point stencil; plain = 1: right-hand side form
  static int launch_q1_stencil(adaflo_ctx *ctx, const int sub, const double c_mass, const double c_lap, const uint32_t con,
                               const double con_sign, const double *diag, double *dst, const double *src,
                               const int n_blocks, const int plain)
  {
    StencilArgs S{};
    S.nnx = sub * ctx->desc.ncell[0] + 1;
    S.nny = sub * ctx->desc.ncell[1] + 1;
    S.nnz = sub * ctx->desc.ncell[2] + 1;
    S.plane            = (long)S.nnx * S.nny;
    S.comp_stride      = S.plane * S.nnz;
    S.flat             = (long)((S.nny + FSR - 1) / FSR) * S.nnx;
    S.blocks_per_plane = (int)(((S.flat + FSW - 1) / FSW + 3) / 4); // four independent waves per workgroup
    // z-chunk: the longest of 12 / 8 / 6 / 4 planes that still gives two rounds of the ~768 resident
    // workgroups (measured, scripts/dev/stencil_probe.hip: 257 x 257 x 513 nodes 0.114 / 0.110 / 0.103 /
    // 0.099 / 0.102 / 0.105 ms for 4 / 6 / 8 / 12 / 16 / 32 planes; 161 x 161 x 321: 0.026 ms at 6..8)
    int lz = 4;
    for (const int c : {6, 8, 12})
      if ((long)S.blocks_per_plane * ((S.nnz + c - 1) / c) * n_blocks >= 1536)
        lz = c;
    if (const char *e = getenv("ADAFLO_STENCIL_LZ")) // (tuning knob of scripts/bench_ops.py)
      lz = std::max(1, atoi(e));
    S.LZ       = lz;
    S.n_chunks = (S.nnz + lz - 1) / lz;
    for (int d = 0; d < 3; ++d)
      {
        const double hs = ctx->desc.h[d] / sub;
        S.m_off[d] = hs / 6.;
        S.m_ctr[d] = hs / 3.;
        S.k_off[d] = -1. / hs;
        S.k_ctr[d] = 1. / hs;
      }
    S.c_mass   = c_mass;
    S.c_lap    = c_lap;
    S.con      = con;
    S.con_sign = con_sign;
    S.diag     = diag;
    S.src      = src;
    S.dst      = dst;
    S.plain    = plain;
    // a CG driver asks for src . dst through the context (krylov.hip): granted if the partials fit
    const long n_partial = (long)S.blocks_per_plane * S.n_chunks * n_blocks;
    ctx->fused_dot_count = 0;
    if (!plain && ctx->fused_dot_partial && n_partial <= ctx->fused_dot_capacity)
      {
        S.dot_partial        = ctx->fused_dot_partial;
        ctx->fused_dot_count = (int)n_partial;
      }
    hipEvent_t stop = ctx->timing ? ctx->kernel_timer.start(ctx->stream) : nullptr;
    hipLaunchKernelGGL(q1_stencil_kernel, dim3((unsigned)(S.blocks_per_plane * S.n_chunks), (unsigned)n_blocks),
                       dim3(256), 0, ctx->stream, S);
    if (stop)
      (void)hipEventRecord(stop, ctx->stream);
    ctx->kernel_timer.count++;
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  // the advection operator can evaluate the velocity itself when the patch of a tile fits (as the right-hand side kernel)
  bool q1_advect_nodal_supported(const adaflo_ctx *ctx)
  {
    const int    k = ctx->k, sub = ctx->s, wn = k * ((TS + sub - 2) / sub + 1) + 1;
    const size_t lds = sizeof(double) * 6 * (size_t)wn * wn;
    return k >= 2 && k <= 4 && !ctx->flat && lds <= 48 * 1024;
  }

  // sub = subdivisions of a cell (level set: s, pressure: 1); mode/coefficients as Q1Args
  int launch_q1_sweep(adaflo_ctx *ctx, const int sub, const int mode, const double c_mass,
                      const double c_lap, const double weight, const uint32_t con, const double con_sign,
                      const double *diag, double *dst, const double *src, const double *state,
                      const int n_blocks, const double *coef_cell, const int coef_stride, const int coef_mid,
                      const double coef_shift)
  {
    if (mode == Q1_MASS_LAPLACE && coef_cell == nullptr)
      return launch_q1_stencil(ctx, sub, c_mass, c_lap, con, con_sign, diag, dst, src, n_blocks, 0);
    Q1Args A{};
    A.coef_cell   = coef_cell;
    A.coef_stride = coef_stride;
    A.coef_mid    = coef_mid;
    A.coef_shift  = coef_shift;
    if (const int e = q1_setup(ctx, sub, mode, n_blocks, A))
      return e;
    const size_t n_wg = (size_t)A.tiles_x * A.tiles_y * A.n_chunks;
    A.c_mass   = c_mass;
    A.c_lap    = c_lap;
    A.weight   = weight;
    A.con      = con;
    A.con_sign = con_sign;
    A.diag     = diag;
    A.src      = src;
    A.dst      = dst;
    A.state    = state;
    const dim3 grid((unsigned)n_wg, (unsigned)n_blocks), block(NTQ);
    hipEvent_t stop = ctx->timing ? ctx->kernel_timer.start(ctx->stream) : nullptr;
    switch (mode)
      {
        case Q1_MASS_LAPLACE:
          hipLaunchKernelGGL((q1_sweep_kernel<Q1_MASS_LAPLACE>), grid, block, 0, ctx->stream, A);
          break;
        case Q1_LAPLACE_Q3:
          hipLaunchKernelGGL((q1_sweep_kernel<Q1_LAPLACE_Q3>), grid, block, 0, ctx->stream, A);
          break;
        case Q1_ADVECT:
          hipLaunchKernelGGL((q1_sweep_kernel<Q1_ADVECT>), grid, block, 0, ctx->stream, A);
          break;
        case Q1_REINIT_NODAL:
          hipLaunchKernelGGL((q1_sweep_kernel<Q1_REINIT_NODAL>), grid, block, 0, ctx->stream, A);
          break;
        case Q1_ADVECT_NODAL:
          {
            const int k = ctx->k;
            A.sub  = sub;
            A.svel = ctx->d_tab_ls + 2 * (2 * sub) * (sub + 1) + 2 * sub;
            A.vnx  = k * ctx->desc.ncell[0] + 1;
            A.vny  = k * ctx->desc.ncell[1] + 1;
            A.wn   = k * ((TS + sub - 2) / sub + 1) + 1;
            const size_t lds = sizeof(double) * 6 * (size_t)A.wn * A.wn;
            if (!q1_advect_nodal_supported(ctx))
              return ADAFLO_EUNSUPPORTED;
            hipError_t err = hipSuccess;
#define ADVN(KU)                                                                                              \
  {                                                                                                           \
    if (lds > 40 * 1024)                                                                                      \
      err = hipFuncSetAttribute(reinterpret_cast<const void *>(&q1_sweep_kernel<Q1_ADVECT_NODAL, KU>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
    hipLaunchKernelGGL((q1_sweep_kernel<Q1_ADVECT_NODAL, KU>), grid, block, lds, ctx->stream, A);             \
  }
            switch (k)
              {
                case 2:
                  ADVN(2);
                  break;
                case 3:
                  ADVN(3);
                  break;
                default:
                  ADVN(4);
              }
#undef ADVN
            if (err != hipSuccess)
              return ADAFLO_EHIP;
            break;
          }
        default:
          hipLaunchKernelGGL((q1_sweep_kernel<Q1_REINIT>), grid, block, 0, ctx->stream, A);
      }
    if (stop)
      (void)hipEventRecord(stop, ctx->stream);
    ctx->kernel_timer.count++;
    const long tiles = (long)A.tiles_x * A.tiles_y, n1 = tiles * A.nnz, n2 = tiles * (A.n_chunks - 1);
    long       nb    = n1 + n2;
    if (nb > 256 * 512)
      nb = 256 * 512;
    hipLaunchKernelGGL(q1_fixup_kernel, dim3((unsigned)nb, (unsigned)n_blocks), dim3(64), 0, ctx->stream, A, n1, n2);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  // mode 0: normal rhs (src = level set, dst = 3 blocks), 1: curvature rhs (src = normal, 3 blocks)
  int launch_q1_stencil_rhs(adaflo_ctx *ctx, const int mode, double *dst, const double *src)
  {
    StencilRhsArgs S{};
    const int sub = ctx->s;
    S.nnx = sub * ctx->desc.ncell[0] + 1;
    S.nny = sub * ctx->desc.ncell[1] + 1;
    S.nnz = sub * ctx->desc.ncell[2] + 1;
    S.plane            = (long)S.nnx * S.nny;
    S.comp_stride      = S.plane * S.nnz;
    S.blocks_per_plane = (int)((S.plane + 255) / 256);
    int lz = 32;
    while (lz > 4 && (long)S.blocks_per_plane * ((S.nnz + lz - 1) / lz) < 2048)
      lz /= 2;
    S.LZ       = lz;
    S.n_chunks = (S.nnz + lz - 1) / lz;
    for (int d = 0; d < 3; ++d)
      {
        const double hs = ctx->desc.h[d] / sub;
        S.m_off[d] = hs / 6.;
        S.m_ctr[d] = hs / 3.;
      }
    S.con = ctx->brick.con_ls;
    S.src = src;
    S.dst = dst;
    const dim3 grid((unsigned)(S.blocks_per_plane * S.n_chunks));
    if (mode == 0)
      hipLaunchKernelGGL(q1_stencil_rhs_kernel<0>, grid, dim3(256), 0, ctx->stream, S);
    else
      hipLaunchKernelGGL(q1_stencil_rhs_kernel<1>, grid, dim3(256), 0, ctx->stream, S);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }
  // Right-hand sides on the sweep structure.  kind 0: reinitialisation (f = phi, normal field; flag bit 0
  // diffuse_only, bit 1 first step), kind 1: advection (f = phi, phi_old, phi_old_old; flag = use_old_old).
  // `state` is the quadrature-point array in sweep layout (written on the first reinitialisation step and
  // by the advection, read otherwise).  Returns ADAFLO_EUNSUPPORTED when the velocity patch of a tile does
  // not fit into LDS (the caller then takes the generic kernels).
  int launch_q1_rhs(adaflo_ctx *ctx, const int kind, const int flag, double *dst, const double *f0, const double *f1,
                    const double *f2, const double *f3, const double *vel, double *state, const bool state_only)
  {
    Q1RhsArgs R{};
    const int sub = ctx->s, k = ctx->k;
    if (kind == 0 && (flag & 1))
      {
        // diffusion step (:185-187): dst += -(grad w, diffusion grad phi) = -diffusion K phi, a 27-point stencil
        const double hc = std::max(ctx->desc.h[0], std::max(ctx->desc.h[1], ctx->desc.h[2]));
        return launch_q1_stencil(ctx, sub, 0., -std::max(ctx->ls.epsilon_used, hc / sub), ctx->brick.con_ls, 0., nullptr, dst,
                                 f0, 1, 1);
      }
    if (const int e = q1_setup(ctx, sub, Q1_ADVECT, 1, R.q))
      return e;
    R.q.con = ctx->brick.con_ls;
    R.q.dst = dst;
    R.f[0] = f0, R.f[1] = f1, R.f[2] = f2, R.f[3] = f3;
    R.flag  = flag;
    R.sub   = sub;
    R.state = state;
    const LSDev &P     = ctx->ls;
    const double hcell = std::max(ctx->desc.h[0], std::max(ctx->desc.h[1], ctx->desc.h[2]));
    R.diffusion        = std::max(P.epsilon_used, hcell / sub);
    R.weight           = P.weight;
    R.weight_old       = P.weight_old;
    R.weight_old_old   = P.weight_old_old;
    const size_t n_wg = (size_t)R.q.tiles_x * R.q.tiles_y * R.q.n_chunks;
    const dim3   grid((unsigned)n_wg), block(NTQ);
    size_t       lds = 0;
    hipError_t   err = hipSuccess;
    if (kind == 0)
      {
        const bool first = !(flag & 1) && (flag & 2);
        R.nf = first ? 4 : 1;
        lds  = sizeof(double) * (2 * R.nf * TNQ * TNQ + 6 * NTQ);
        if (first)
          hipLaunchKernelGGL((q1_rhs_kernel<Q1RHS_REINIT, 1>), grid, block, lds, ctx->stream, R);
        else
          hipLaunchKernelGGL((q1_rhs_kernel<Q1RHS_REINIT, 0>), grid, block, lds, ctx->stream, R);
      }
    else
      {
        R.nf         = state_only ? 0 : (flag ? 3 : 2);
        R.state_only = state_only;
        R.vel        = vel;
        R.svel = ctx->d_tab_ls + 2 * (2 * sub) * (sub + 1) + 2 * sub;
        R.vnx  = k * ctx->desc.ncell[0] + 1;
        R.vny  = k * ctx->desc.ncell[1] + 1;
        R.wn   = k * ((TS + sub - 2) / sub + 1) + 1;
        lds    = sizeof(double) * (2 * 3 * TNQ * TNQ + 6 * NTQ + 6 * (size_t)R.wn * R.wn);
        if (lds > 96 * 1024)
          return ADAFLO_EUNSUPPORTED;
#define ADV(KU)                                                                                           \
  {                                                                                                       \
    if (lds > 64 * 1024)                                                                                  \
      err = hipFuncSetAttribute(reinterpret_cast<const void *>(&q1_rhs_kernel<Q1RHS_ADVECT, KU>),         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
    hipLaunchKernelGGL((q1_rhs_kernel<Q1RHS_ADVECT, KU>), grid, block, lds, ctx->stream, R);              \
  }
        switch (k)
          {
            case 2:
              ADV(2);
              break;
            case 3:
              ADV(3);
              break;
            case 4:
              ADV(4);
              break;
            default:
              return ADAFLO_EUNSUPPORTED;
          }
#undef ADV
      }
    if (err != hipSuccess || hipGetLastError() != hipSuccess)
      return ADAFLO_EHIP;
    const long tiles = (long)R.q.tiles_x * R.q.tiles_y, n1 = tiles * R.q.nnz, n2 = tiles * (R.q.n_chunks - 1);
    long       nb    = n1 + n2;
    if (nb > 256 * 512)
      nb = 256 * 512;
    if (!state_only)
      hipLaunchKernelGGL(q1_fixup_kernel, dim3((unsigned)nb, 1u), dim3(64), 0, ctx->stream, R.q, n1, n2);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }
} // namespace adaflo_hip
