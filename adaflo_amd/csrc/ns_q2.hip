// ns_q2.hip -- specialised streaming kernel for the headline case: 3D Taylor-Hood
// Q2/Q1 NavierStokesMatrix::vmult / velocity_vmult with constant rho, mu
// (reference: local_operation<1,...>, source/navier_stokes_matrix.cc:601-916).
//
// MI355X-first design (DESIGN.md section "Q2/Q1 sweep kernel"):
//  * A workgroup (256 threads = 4 waves) owns a column of 8x8 cells and sweeps
//    LZ cell layers in z.  Per layer each QUAD of lanes owns one cell: lanes
//    0..2 = velocity components, lane 3 = pressure.  The whole sum factorisation
//    of a cell component (27 values) lives in REGISTERS; components talk to each
//    other only at the quadrature points, through DPP quad permutes.  No LDS
//    traffic for the tensor contractions at all.
//  * The DoF vectors are staged through LDS one node plane at a time (coalesced
//    row loads, each src entry fetched once per workgroup); the integrated cell
//    results are combined per node in LDS and leave as coalesced row stores.
//    Only nodes on the lateral / z seams between workgroups use f64 atomics.
//  * The quadrature-point linearisation state (12 doubles per point, 87 % of the
//    compulsory HBM traffic) is streamed exactly once, 16 B per lane, in the
//    order the quads consume it (layout produced by q2_convert_state_kernel).
//  * The Q1 pressure is expanded to the Q2 nodal basis at gather time, so all
//    four lanes of a quad run the identical Q2 interpolation code.
#include "kernels.hpp"

namespace adaflo_hip
{
  namespace
  {
    constexpr int TX = 8, TY = 8;          // cells per workgroup layer
    constexpr int NT = 256;                // threads per workgroup
    constexpr int PNX = 2 * TX + 1;        // velocity nodes per tile row (17)
    constexpr int PNY = 2 * TY + 1;
    constexpr int UPLANE = PNX * PNY * 3;  // doubles per velocity node plane (867)
    constexpr int QNX = TX + 1, QNY = TY + 1;
    constexpr int PPLANE = QNX * QNY;      // doubles per pressure node plane (81)
    constexpr int NCELL = TX * TY;         // 64
    constexpr int SCR_U = 27 * NCELL * 3;  // scratch: [local node][cell][comp]
    constexpr int SCR_P = 8 * NCELL;
    constexpr int STATE_PER_LAYER = 27 * 2 * NCELL * 3 * 2; // doubles per (tile, layer)
    constexpr int PF = 3;                  // q-points of state prefetched ahead

    struct Q2Args
    {
      int    ncx, ncy, ncz, nnx, nny, nnz, npx, npy, npz;
      int    tiles_x, tiles_y, LZ, n_chunks;
      double ih[3];
      double s0, s1, s2;   // first row of the 3x3 interpolation matrix (rows: [s0 s1 s2], [0 1 0], [s2 s1 s0])
      // collocation derivative at the 3 Gauss points: rows (a0 a1 a2), (-b 0 b), (-a2 -a1 -a0)
      double a0, a1, a2, b;
      double wj[4];        // JxW by number of "middle" indices of the q-point: det * w0^(3-m) * w1^m
      double cA, cB;       // conv = cA * u + cB * res, cA = gamma*rho - damping, cB = tau1*rho
      double beta, tau_gd, tmu;
      int    integrate_p;
      uint32_t con_u, con_p;
      const double *src_u, *src_p;
      double       *dst_u, *dst_p;
      const double *state;
    };

    template <int SEL>
    __device__ __forceinline__ double quad_bcast(const double x)
    {
      constexpr int ctrl = SEL * 0x55; // quad_perm:[SEL,SEL,SEL,SEL]
      int lo = __double2loint(x), hi = __double2hiint(x);
      lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
      hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
      return __hiloint2double(hi, lo);
    }

    __device__ __forceinline__ double sel3(const int d, const double a, const double b, const double c)
    {
      return d == 0 ? a : (d == 1 ? b : c);
    }

    // one line of the 1D interpolation (Gauss-Lobatto nodes -> Gauss points, degree 2):
    // middle row is the identity because the middle node sits on the middle Gauss point
    __device__ __forceinline__ void interp3(double &x0, double &x1, double &x2, const double s0,
                                            const double s1, const double s2)
    {
      const double t0 = s0 * x0 + s1 * x1 + s2 * x2;
      const double t2 = s2 * x0 + s1 * x1 + s0 * x2;
      x0 = t0;
      x2 = t2;
    }
    // transpose of interp3
    __device__ __forceinline__ void interp3_t(double &x0, double &x1, double &x2, const double s0,
                                              const double s1, const double s2)
    {
      const double t0 = s0 * x0 + s2 * x2;
      const double t1 = s1 * (x0 + x2) + x1;
      const double t2 = s2 * x0 + s0 * x2;
      x0 = t0;
      x1 = t1;
      x2 = t2;
    }

    // derivative at point p of the quadratic through (v0,v1,v2) at the 3 Gauss points
    __device__ __forceinline__ double dline(const int p, const double v0, const double v1,
                                            const double v2, const double a0, const double a1,
                                            const double a2, const double b)
    {
      if (p == 0)
        return a0 * v0 + a1 * v1 + a2 * v2;
      if (p == 1)
        return b * (v2 - v0);
      return -(a2 * v0 + a1 * v1 + a0 * v2);
    }
    // transpose: (r0,r1,r2) += row p of the derivative matrix times t
    __device__ __forceinline__ void dline_t(const int p, const double t, double &r0, double &r1,
                                            double &r2, const double a0, const double a1,
                                            const double a2, const double b)
    {
      if (p == 0)
        {
          r0 += a0 * t;
          r1 += a1 * t;
          r2 += a2 * t;
        }
      else if (p == 1)
        {
          r0 -= b * t;
          r2 += b * t;
        }
      else
        {
          r0 -= a2 * t;
          r1 -= a1 * t;
          r2 -= a0 * t;
        }
    }

    // load one velocity node plane K of the tile into LDS, constraints resolved
    __device__ __forceinline__ void load_u_plane(const Q2Args &A, double *pl, const int K,
                                                 const int I0, const int J0)
    {
      for (int e = threadIdx.x; e < UPLANE; e += NT)
        {
          const int comp = e % 3, n = e / 3, i = n % PNX, j = n / PNX;
          const int I = I0 + i, J = J0 + j;
          double    v = 0.;
          if (I < A.nnx && J < A.nny && K < A.nnz)
            {
              v = A.src_u[((long)(K * (long)A.nny + J) * A.nnx + I) * 3 + comp];
              if (on_constrained_face(I, J, K, A.nnx, A.nny, A.nnz, A.con_u, 3, comp))
                v = 0.;
            }
          pl[e] = v;
        }
    }

    __device__ __forceinline__ void load_p_plane(const Q2Args &A, double *pl, const int K,
                                                 const int I0, const int J0)
    {
      for (int e = threadIdx.x; e < PPLANE; e += NT)
        {
          const int i = e % QNX, j = e / QNX;
          const int I = I0 + i, J = J0 + j;
          double    v = 0.;
          if (I < A.npx && J < A.npy && K < A.npz)
            {
              v = A.src_p[(long)(K * (long)A.npy + J) * A.npx + I];
              if (on_constrained_face(I, J, K, A.npx, A.npy, A.npz, A.con_p, 1, 0))
                v = 0.;
            }
          pl[e] = v;
        }
    }

    // sum of the cell-local contributions to tile node (i,j) of local plane lk
    // (cells beyond the domain inside a partial tile, tcx/tcy = valid extents, are skipped)
    __device__ __forceinline__ double node_sum_u(const double *scr, const int i, const int j,
                                                 const int lk, const int comp, const int tcx,
                                                 const int tcy)
    {
      double    s   = 0.;
      const int cx0 = i >> 1, cy0 = j >> 1;
#pragma unroll
      for (int sy = 0; sy < 2; ++sy)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
          {
            // candidate cell (cx0 - sx, cy0 - sy) sees the node at local index li = i - 2*cx
            const int cx = cx0 - sx, cy = cy0 - sy;
            const int li = i - 2 * cx, lj = j - 2 * cy;
            if (cx >= 0 && cx < tcx && cy >= 0 && cy < tcy && li <= 2 && lj <= 2)
              s += scr[((li + 3 * lj + 9 * lk) * NCELL + cy * TX + cx) * 3 + comp];
          }
      return s;
    }

    __device__ __forceinline__ double node_sum_p(const double *scr, const int i, const int j,
                                                 const int lk, const int tcx, const int tcy)
    {
      double s = 0.;
#pragma unroll
      for (int sy = 0; sy < 2; ++sy)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
          {
            const int cx = i - sx, cy = j - sy; // local node index li = sx
            if (cx >= 0 && cx < tcx && cy >= 0 && cy < tcy)
              s += scr[((sx + 2 * sy + 4 * lk)) * NCELL + cy * TX + cx];
          }
      return s;
    }

    // write one finished velocity node plane (tile-local entries e = tid + r*NT)
    template <bool FROM_CARRY>
    __device__ __forceinline__ void emit_u_plane(const Q2Args &A, const double *scr, const int lk,
                                                 double (&carry)[4], const int K, const int I0,
                                                 const int J0, const bool zseam, const int tcx,
                                                 const int tcy)
    {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        {
          const int e = threadIdx.x + r * NT;
          if (e < UPLANE)
            {
              const int comp = e % 3, n = e / 3, i = n % PNX, j = n / PNX;
              const int I = I0 + i, J = J0 + j;
              double    v;
              if (FROM_CARRY)
                v = carry[r];
              else
                {
                  v = node_sum_u(scr, i, j, lk, comp, tcx, tcy);
                  if (lk == 0)
                    v += carry[r];
                }
              if (I < A.nnx && J < A.nny)
                {
                  const long idx = ((long)(K * (long)A.nny + J) * A.nnx + I) * 3 + comp;
                  if (on_constrained_face(I, J, K, A.nnx, A.nny, A.nnz, A.con_u, 3, comp))
                    A.dst_u[idx] = A.src_u[idx]; // :247-256 (+1 on the velocity block)
                  else
                    {
                      const bool seam = zseam || (i == 0 && I > 0) || (i == PNX - 1 && I < A.nnx - 1) ||
                                        (j == 0 && J > 0) || (j == PNY - 1 && J < A.nny - 1);
                      if (seam)
                        unsafeAtomicAdd(&A.dst_u[idx], v);
                      else
                        A.dst_u[idx] = v;
                    }
                }
            }
        }
    }

    template <bool FROM_CARRY>
    __device__ __forceinline__ void emit_p_plane(const Q2Args &A, const double *scr, double &carry,
                                                 const int K, const int I0, const int J0,
                                                 const bool zseam, const int tcx, const int tcy)
    {
      const int e = threadIdx.x;
      if (e < PPLANE)
        {
          const int i = e % QNX, j = e / QNX;
          const int I = I0 + i, J = J0 + j;
          double    v = carry;
          if (!FROM_CARRY)
            v += node_sum_p(scr, i, j, 0, tcx, tcy);
          if (I < A.npx && J < A.npy)
            {
              const long idx = (long)(K * (long)A.npy + J) * A.npx + I;
              if (on_constrained_face(I, J, K, A.npx, A.npy, A.npz, A.con_p, 1, 0))
                A.dst_p[idx] = -A.src_p[idx]; // -1 on the pressure block of vmult
              else
                {
                  const bool seam = zseam || (i == 0 && I > 0) || (i == QNX - 1 && I < A.npx - 1) ||
                                    (j == 0 && J > 0) || (j == QNY - 1 && J < A.npy - 1);
                  if (seam)
                    unsafeAtomicAdd(&A.dst_p[idx], v);
                  else
                    A.dst_p[idx] = v;
                }
            }
        }
    }

    // LIN_MODE: 0 Newton (state = u_lin, grad u_lin), 1 Picard-type (state = u_lin, div u_lin),
    //           2 no convective linearisation (explicit convection, Stokes)
    template <int LIN_MODE, bool WITH_P>
    __global__ __launch_bounds__(NT, 2) void ns_q2_kernel(const Q2Args A)
    {
      __shared__ double u_pl[3 * UPLANE];
      __shared__ double p_pl[2 * PPLANE];
      __shared__ double scr_u[SCR_U];
      __shared__ double scr_p[SCR_P];

      const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
      const int d = lane & 3, cq = lane >> 2;
      const int cxl = cq & 7, cyl = 2 * wave + (cq >> 3), cell = cyl * TX + cxl;

      // workgroup -> (tile, z-chunk); chunks of one tile column are consecutive
      // in the remapped index so that an XCD's L2 sees neighbouring work
      const long nwg = (long)A.tiles_x * A.tiles_y * A.n_chunks;
      const long wg  = xcd_remap(blockIdx.x, nwg);
      const int  bz  = (int)(wg % A.n_chunks);
      const int  bt  = (int)(wg / A.n_chunks);
      const int  bx = bt % A.tiles_x, by = bt / A.tiles_x;
      const int  cz0 = bz * A.LZ;
      const int  nl  = min(A.LZ, A.ncz - cz0);
      const int  I0 = 2 * TX * bx, J0 = 2 * TY * by;   // velocity node origin of the tile
      const int  Ip0 = TX * bx, Jp0 = TY * by;        // pressure node origin
      const int  tcx = min(TX, A.ncx - TX * bx), tcy = min(TY, A.ncy - TY * by);

      const double s0 = A.s0, s1 = A.s1, s2 = A.s2;
      const bool   is_p  = d == 3;
      const double tmu_l = is_p ? 0. : A.tmu; // the pressure lane integrates no gradient terms
      const double d0 = d == 0 ? 1. : 0., d1 = d == 1 ? 1. : 0., d2 = d == 2 ? 1. : 0.;

      double carry_u[4] = {0., 0., 0., 0.}, carry_p = 0.;

      // prologue: bottom planes of the first layer
      load_u_plane(A, u_pl + ((2 * cz0) % 3) * UPLANE, 2 * cz0, I0, J0);
      load_u_plane(A, u_pl + ((2 * cz0 + 1) % 3) * UPLANE, 2 * cz0 + 1, I0, J0);
      load_u_plane(A, u_pl + ((2 * cz0 + 2) % 3) * UPLANE, 2 * cz0 + 2, I0, J0);
      if (WITH_P)
        {
          load_p_plane(A, p_pl + (cz0 % 2) * PPLANE, cz0, Ip0, Jp0);
          load_p_plane(A, p_pl + ((cz0 + 1) % 2) * PPLANE, cz0 + 1, Ip0, Jp0);
        }
      __syncthreads();

      const double2 *state = reinterpret_cast<const double2 *>(A.state);

      for (int layer = 0; layer < nl; ++layer)
        {
          const int cz = cz0 + layer;

          // ---- B: gather my 27 (8) values from the LDS node planes ---------------------
          double V[27];
          if (!is_p)
            {
#pragma unroll
              for (int c = 0; c < 3; ++c)
                {
                  const double *pl = u_pl + ((2 * cz + c) % 3) * UPLANE + (2 * cyl * PNX + 2 * cxl) * 3 + d;
#pragma unroll
                  for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                      V[a + 3 * b + 9 * c] = pl[(b * PNX + a) * 3];
                }
            }
          else
            {
              // Q1 -> Q2 nodal expansion (mid nodes = averages)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                {
                  const double *pl = p_pl + ((cz + c) % 2) * PPLANE + cyl * QNX + cxl;
#pragma unroll
                  for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                      V[2 * a + 6 * b + 18 * c] = WITH_P ? pl[b * QNX + a] : 0.;
                }
#pragma unroll
              for (int c = 0; c < 3; c += 2)
#pragma unroll
                for (int b = 0; b < 3; b += 2)
                  V[1 + 3 * b + 9 * c] = 0.5 * (V[0 + 3 * b + 9 * c] + V[2 + 3 * b + 9 * c]);
#pragma unroll
              for (int c = 0; c < 3; c += 2)
#pragma unroll
                for (int a = 0; a < 3; ++a)
                  V[a + 3 + 9 * c] = 0.5 * (V[a + 9 * c] + V[a + 6 + 9 * c]);
#pragma unroll
              for (int n = 0; n < 9; ++n)
                V[n + 9] = 0.5 * (V[n] + V[n + 18]);
            }

          // ---- C: interpolate to the Gauss points (in place) ----------------------------
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int b = 0; b < 3; ++b)
              interp3(V[3 * b + 9 * c], V[1 + 3 * b + 9 * c], V[2 + 3 * b + 9 * c], s0, s1, s2);
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int a = 0; a < 3; ++a)
              interp3(V[a + 9 * c], V[a + 3 + 9 * c], V[a + 6 + 9 * c], s0, s1, s2);
#pragma unroll
          for (int n = 0; n < 9; ++n)
            interp3(V[n], V[n + 9], V[n + 18], s0, s1, s2);

          // ---- quadrature-point loop (source/navier_stokes_matrix.cc:702-893) ------------
          double R[27];
#pragma unroll
          for (int n = 0; n < 27; ++n)
            R[n] = 0.;

          // uniform (scalar) base of this (tile, layer) block + per-lane offset, in double2 units
          const double2 *sp   = state + ((size_t)bt * A.ncz + cz) * (27 * 2 * NCELL * 3);
          const unsigned slan = cell * 3 + (is_p ? 0 : d);
          double2        sbuf[PF][2];
          if (LIN_MODE != 2)
            {
#pragma unroll
              for (int q = 0; q < PF; ++q)
                {
                  sbuf[q][0] = sp[(2 * q) * (NCELL * 3) + slan];
                  sbuf[q][1] = sp[(2 * q + 1) * (NCELL * 3) + slan];
                }
            }

#pragma unroll
          for (int q = 0; q < 27; ++q)
            {
              const int qx = q % 3, qy = (q / 3) % 3, qz = q / 9;
              double2   st0 = make_double2(0., 0.), st1 = make_double2(0., 0.);
              if (LIN_MODE != 2)
                {
                  st0 = sbuf[q % PF][0];
                  st1 = sbuf[q % PF][1];
                  if (q + PF < 27)
                    {
                      sbuf[q % PF][0] = sp[(2 * (q + PF)) * (NCELL * 3) + slan];
                      sbuf[q % PF][1] = sp[(2 * (q + PF) + 1) * (NCELL * 3) + slan];
                    }
                  asm volatile("" ::: "memory");
                }
              const double Vq = V[q];
              // reference-cell derivatives by the collocation derivative, then J^{-T}
              const double g0 = dline(qx, V[0 + 3 * qy + 9 * qz], V[1 + 3 * qy + 9 * qz], V[2 + 3 * qy + 9 * qz],
                                      A.a0, A.a1, A.a2, A.b) * A.ih[0];
              const double g1 = dline(qy, V[qx + 9 * qz], V[qx + 3 + 9 * qz], V[qx + 6 + 9 * qz],
                                      A.a0, A.a1, A.a2, A.b) * A.ih[1];
              const double g2 = dline(qz, V[qx + 3 * qy], V[qx + 3 * qy + 9], V[qx + 3 * qy + 18],
                                      A.a0, A.a1, A.a2, A.b) * A.ih[2];

              // gradient rows of the three velocity components, visible to all four lanes
              const double G00 = quad_bcast<0>(g0), G01 = quad_bcast<0>(g1), G02 = quad_bcast<0>(g2);
              const double G10 = quad_bcast<1>(g0), G11 = quad_bcast<1>(g1), G12 = quad_bcast<1>(g2);
              const double G20 = quad_bcast<2>(g0), G21 = quad_bcast<2>(g1), G22 = quad_bcast<2>(g2);
              const double div = G00 + G11 + G22; // :706

              double conv = A.cA * Vq; // :717, :827-835
              if (LIN_MODE == 0)       // Newton :802-816
                {
                  const double u0 = quad_bcast<0>(Vq), u1 = quad_bcast<1>(Vq), u2 = quad_bcast<2>(Vq);
                  const double ub0 = quad_bcast<0>(st0.x), ub1 = quad_bcast<1>(st0.x), ub2 = quad_bcast<2>(st0.x);
                  const double trl = quad_bcast<0>(st0.y) + quad_bcast<1>(st1.x) + quad_bcast<2>(st1.y);
                  double       res = A.beta * (div * st0.x + trl * Vq);
                  res += ub0 * g0 + u0 * st0.y;
                  res += ub1 * g1 + u1 * st1.x;
                  res += ub2 * g2 + u2 * st1.y;
                  conv += A.cB * res;
                }
              else if (LIN_MODE == 1) // Picard-type :817-826, state = (u_lin, div_lin)
                {
                  const double ub0 = quad_bcast<0>(st0.x), ub1 = quad_bcast<1>(st0.x), ub2 = quad_bcast<2>(st0.x);
                  double       res = (A.beta * st0.y) * Vq;
                  res += ub0 * g0;
                  res += ub1 * g1;
                  res += ub2 * g2;
                  conv += A.cB * res;
                }

              const double jxw = A.wj[(qx == 1) + (qy == 1) + (qz == 1)];
              // column d of the velocity gradient (transpose part of the symmetric gradient)
              const double c0 = sel3(d, G00, G01, G02), c1 = sel3(d, G10, G11, G12), c2 = sel3(d, G20, G21, G22);
              double       diag = A.tau_gd * div;
              if (WITH_P)
                diag -= quad_bcast<3>(Vq);
              diag *= jxw;
              // :859-892: row d of tmu (grad u + grad u^T) + (tau_gd div - p) I, times JxW J^{-1}
              const double tmj = tmu_l * jxw;
              const double tg0 = (tmj * (g0 + c0) + d0 * diag) * A.ih[0];
              const double tg1 = (tmj * (g1 + c1) + d1 * diag) * A.ih[1];
              const double tg2 = (tmj * (g2 + c2) + d2 * diag) * A.ih[2];
              // test value: momentum rows (:837) or the pressure row (q, -div u) (:853-856)
              const double tv = (is_p ? -div : conv) * jxw;

              // integrate (collocation derivative transposed), accumulate at the Gauss points
              R[q] += tv;
              dline_t(qx, tg0, R[0 + 3 * qy + 9 * qz], R[1 + 3 * qy + 9 * qz], R[2 + 3 * qy + 9 * qz], A.a0, A.a1, A.a2, A.b);
              dline_t(qy, tg1, R[qx + 9 * qz], R[qx + 3 + 9 * qz], R[qx + 6 + 9 * qz], A.a0, A.a1, A.a2, A.b);
              dline_t(qz, tg2, R[qx + 3 * qy], R[qx + 3 * qy + 9], R[qx + 3 * qy + 18], A.a0, A.a1, A.a2, A.b);
            }

          // ---- transposed interpolation back to the nodes --------------------------------
#pragma unroll
          for (int n = 0; n < 9; ++n)
            interp3_t(R[n], R[n + 9], R[n + 18], s0, s1, s2);
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int a = 0; a < 3; ++a)
              interp3_t(R[a + 9 * c], R[a + 3 + 9 * c], R[a + 6 + 9 * c], s0, s1, s2);
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int b = 0; b < 3; ++b)
              interp3_t(R[3 * b + 9 * c], R[1 + 3 * b + 9 * c], R[2 + 3 * b + 9 * c], s0, s1, s2);

          // ---- D: cell results -> LDS scratch ---------------------------------------------
          if (!is_p)
            {
#pragma unroll
              for (int l = 0; l < 27; ++l)
                scr_u[(l * NCELL + cell) * 3 + d] = R[l];
            }
          else
            {
              // Q2 -> Q1 test functions: phi^Q1_a = sum_i phi^Q1_a(x_i) phi^Q2_i, weights 1, 1/2, 0
#pragma unroll
              for (int n = 0; n < 9; ++n)
                {
                  R[n] += 0.5 * R[n + 9];
                  R[n + 18] += 0.5 * R[n + 9];
                }
#pragma unroll
              for (int c = 0; c < 3; c += 2)
#pragma unroll
                for (int a = 0; a < 3; ++a)
                  {
                    R[a + 9 * c] += 0.5 * R[a + 3 + 9 * c];
                    R[a + 6 + 9 * c] += 0.5 * R[a + 3 + 9 * c];
                  }
#pragma unroll
              for (int c = 0; c < 3; c += 2)
#pragma unroll
                for (int b = 0; b < 3; b += 2)
                  {
                    R[3 * b + 9 * c] += 0.5 * R[1 + 3 * b + 9 * c];
                    R[2 + 3 * b + 9 * c] += 0.5 * R[1 + 3 * b + 9 * c];
                  }
#pragma unroll
              for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                  for (int a = 0; a < 2; ++a)
                    scr_p[(a + 2 * b + 4 * c) * NCELL + cell] = R[2 * a + 6 * b + 18 * c];
            }
          __syncthreads();

          // ---- E: combine per node, write the two finished planes, stage next planes -----
          {
            const bool zseam0 = layer == 0 && cz0 > 0;
            emit_u_plane<false>(A, scr_u, 0, carry_u, 2 * cz, I0, J0, zseam0, tcx, tcy);
            emit_u_plane<false>(A, scr_u, 1, carry_u, 2 * cz + 1, I0, J0, false, tcx, tcy);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              {
                const int e = tid + r * NT;
                if (e < UPLANE)
                  {
                    const int comp = e % 3, n = e / 3;
                    carry_u[r]     = node_sum_u(scr_u, n % PNX, n / PNX, 2, comp, tcx, tcy);
                  }
              }
            if (WITH_P && A.integrate_p)
              {
                emit_p_plane<false>(A, scr_p, carry_p, cz, Ip0, Jp0, zseam0, tcx, tcy);
                if (tid < PPLANE)
                  carry_p = node_sum_p(scr_p, tid % QNX, tid / QNX, 1, tcx, tcy);
              }
            if (layer + 1 < nl)
              {
                load_u_plane(A, u_pl + ((2 * cz + 3) % 3) * UPLANE, 2 * cz + 3, I0, J0);
                load_u_plane(A, u_pl + ((2 * cz + 4) % 3) * UPLANE, 2 * cz + 4, I0, J0);
                if (WITH_P)
                  load_p_plane(A, p_pl + ((cz + 2) % 2) * PPLANE, cz + 2, Ip0, Jp0);
              }
          }
          __syncthreads();
        }

      // top plane of the chunk
      {
        const int  cze   = cz0 + nl;
        const bool zseam = cze < A.ncz;
        emit_u_plane<true>(A, scr_u, 0, carry_u, 2 * cze, I0, J0, zseam, tcx, tcy);
        if (WITH_P && A.integrate_p)
          emit_p_plane<true>(A, scr_p, carry_p, cze, Ip0, Jp0, zseam, tcx, tcy);
      }
    }

    // generic [cell][12][27] -> streaming layout [tile][layer][q][half][cell-in-tile*3+d][2]
    __global__ __launch_bounds__(256) void q2_convert_state_kernel(double *__restrict__ out,
                                                                   const double *__restrict__ gen,
                                                                   const int ncx, const int ncy,
                                                                   const int ncz, const int tiles_x,
                                                                   const long total, const int lin_mode)
    {
      for (long o = blockIdx.x * 256L + threadIdx.x; o < total; o += (long)gridDim.x * 256)
        {
          const int  j    = (int)(o & 1);
          long       r    = o >> 1;
          const int  cd   = (int)(r % (NCELL * 3));
          r /= NCELL * 3;
          const int half = (int)(r & 1);
          r >>= 1;
          const int q = (int)(r % 27);
          r /= 27;
          const int  cz = (int)(r % ncz);
          const long bt = r / ncz;
          const int  bx = (int)(bt % tiles_x), by = (int)(bt / tiles_x);
          const int  cl = cd / 3, d = cd % 3;
          const int  cx = bx * TX + (cl % TX), cy = by * TY + (cl / TX);
          double     v  = 0.;
          if (cx < ncx && cy < ncy)
            {
              const long cell = cx + (long)ncx * (cy + (long)ncy * cz);
              int        comp;
              if (half == 0 && j == 0)
                comp = d; // u_lin[d]
              else if (lin_mode == 1)
                comp = 3; // div_lin (second[0][0])
              else
                comp = 3 + 3 * d + (2 * half + j - 1); // grad_lin[d][e]
              v = gen[(cell * NLIN + comp) * 27 + q];
            }
          out[o] = v;
        }
    }
  } // namespace

  static int q2_lin_mode(const adaflo_ctx *ctx)
  {
    if (ctx->ns.physical_type == ADAFLO_STOKES || ctx->ns.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT)
      return 2;
    return ctx->ns.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON ? 0 : 1;
  }

  bool q2_supported(const adaflo_ctx *ctx)
  {
    return ctx->k == 2 && ctx->rho.p == nullptr && ctx->mu.p == nullptr && ctx->damp.p == nullptr;
  }

  int q2_prepare_state(adaflo_ctx *ctx)
  {
    if (ctx->lin_q2_valid || q2_lin_mode(ctx) == 2)
      return 0;
    const int    tiles_x = (ctx->desc.ncell[0] + TX - 1) / TX, tiles_y = (ctx->desc.ncell[1] + TY - 1) / TY;
    const size_t count   = (size_t)tiles_x * tiles_y * ctx->desc.ncell[2] * STATE_PER_LAYER;
    if (ctx->lin_q2.count != count)
      {
        if (ctx->lin_q2.p)
          (void)hipFree(ctx->lin_q2.p);
        ctx->lin_q2.p     = nullptr;
        ctx->lin_q2.count = 0;
        if (hipMalloc(&ctx->lin_q2.p, count * sizeof(double)) != hipSuccess)
          return ADAFLO_ENOMEM;
        ctx->lin_q2.count = count;
      }
    long nb = (long)((count + 255) / 256);
    if (nb > 256 * 32)
      nb = 256 * 32;
    hipLaunchKernelGGL(q2_convert_state_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream,
                       ctx->lin_q2.p, ctx->lin.p, ctx->desc.ncell[0], ctx->desc.ncell[1],
                       ctx->desc.ncell[2], tiles_x, (long)count, q2_lin_mode(ctx));
    if (hipGetLastError() != hipSuccess)
      return ADAFLO_EHIP;
    ctx->lin_q2_valid = true;
    ctx->lin_q2_mode  = q2_lin_mode(ctx);
    return 0;
  }

  int launch_ns_vmult_q2(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p,
                         const double *src_u, const double *src_p)
  {
    if (ctx->lin_q2_valid && ctx->lin_q2_mode != q2_lin_mode(ctx))
      ctx->lin_q2_valid = false;
    if (int e = q2_prepare_state(ctx))
      return e;
    Q2Args A{};
    A.ncx = ctx->desc.ncell[0];
    A.ncy = ctx->desc.ncell[1];
    A.ncz = ctx->desc.ncell[2];
    A.nnx = 2 * A.ncx + 1;
    A.nny = 2 * A.ncy + 1;
    A.nnz = 2 * A.ncz + 1;
    A.npx = A.ncx + 1;
    A.npy = A.ncy + 1;
    A.npz = A.ncz + 1;
    A.tiles_x = (A.ncx + TX - 1) / TX;
    A.tiles_y = (A.ncy + TY - 1) / TY;
    // z-chunk length: aim at >= 4 resident rounds of 512 workgroups, chunks >= 8 layers
    {
      const long tiles = (long)A.tiles_x * A.tiles_y;
      int        lz    = ctx->q2_lz > 0 ? ctx->q2_lz : 16;
      while (lz > 4 && tiles * ((A.ncz + lz - 1) / lz) < 1024)
        lz /= 2;
      if (lz > A.ncz)
        lz = A.ncz;
      A.LZ       = lz;
      A.n_chunks = (A.ncz + lz - 1) / lz;
    }
    for (int d = 0; d < 3; ++d)
      A.ih[d] = 1. / ctx->desc.h[d];
    {
      const double       det = ctx->desc.h[0] * ctx->desc.h[1] * ctx->desc.h[2];
      const Quadrature1D qu  = gauss(3);
      const Shape1D      su  = shape_fe_q(2, qu);
      A.s0 = su.S[0];
      A.s1 = su.S[1];
      A.s2 = su.S[2];
      const std::vector<double> dc = collocation_derivative(qu);
      A.a0 = dc[0];
      A.a1 = dc[1];
      A.a2 = dc[2];
      A.b  = dc[5];
      for (int m = 0; m < 4; ++m)
        A.wj[m] = det * std::pow(qu.w[0], 3 - m) * std::pow(qu.w[1], m);
    }
    const NSDev &P      = ctx->ns;
    const bool   stokes = P.physical_type == ADAFLO_STOKES;
    const double gamma  = P.physical_type == ADAFLO_INCOMPRESSIBLE ? P.weight : 0.;
    // conv = (gamma u + tau1 res) rho - damping u   (:717,:827-835); Stokes: no value terms (:708)
    A.cA          = stokes ? 0. : gamma * P.density - P.damping;
    A.cB          = stokes ? 0. : P.tau1 * P.density;
    A.beta        = P.beta;
    A.tau_gd      = P.tau_grad_div;
    A.tmu         = P.viscosity * P.tau1; // :841-845
    A.integrate_p = P.linearization != ADAFLO_PROJECTION;
    A.con_u       = ctx->brick.con_u;
    A.con_p       = ctx->brick.con_p;
    A.src_u       = src_u;
    A.src_p       = src_p;
    A.dst_u       = dst_u;
    A.dst_p       = dst_p;
    A.state       = (op == OP_VMULT_VELOCITY && ctx->lin_q2_prec.p) ? ctx->lin_q2_prec.p : ctx->lin_q2.p;
    const int  lin_mode = q2_lin_mode(ctx);
    const bool with_p   = op == OP_VMULT;

    // seam nodes are accumulated with atomics -> dst must start from zero (:229)
    if (hipMemsetAsync(dst_u, 0, sizeof(double) * 3 * ctx->n_nodes_u, ctx->stream) != hipSuccess)
      return ADAFLO_EHIP;
    if (with_p)
      {
        if (A.integrate_p)
          {
            if (hipMemsetAsync(dst_p, 0, sizeof(double) * ctx->n_nodes_p, ctx->stream) != hipSuccess)
              return ADAFLO_EHIP;
          }
        else if (int e = launch_prepare_dst(ctx, dst_p, src_p, ctx->n_nodes_p, 1, A.npx, A.npy, A.npz,
                                            A.con_p, -1., true))
          return e;
      }
    const long nwg  = (long)A.tiles_x * A.tiles_y * A.n_chunks;
    hipEvent_t stop = ctx->timing ? ctx->kernel_timer.start(ctx->stream) : nullptr;
    const dim3 grid((unsigned)nwg), block(NT);
    if (with_p)
      switch (lin_mode)
        {
          case 0:
            hipLaunchKernelGGL((ns_q2_kernel<0, true>), grid, block, 0, ctx->stream, A);
            break;
          case 1:
            hipLaunchKernelGGL((ns_q2_kernel<1, true>), grid, block, 0, ctx->stream, A);
            break;
          default:
            hipLaunchKernelGGL((ns_q2_kernel<2, true>), grid, block, 0, ctx->stream, A);
        }
    else
      switch (lin_mode)
        {
          case 0:
            hipLaunchKernelGGL((ns_q2_kernel<0, false>), grid, block, 0, ctx->stream, A);
            break;
          case 1:
            hipLaunchKernelGGL((ns_q2_kernel<1, false>), grid, block, 0, ctx->stream, A);
            break;
          default:
            hipLaunchKernelGGL((ns_q2_kernel<2, false>), grid, block, 0, ctx->stream, A);
        }
    if (stop)
      (void)hipEventRecord(stop, ctx->stream);
    ctx->kernel_timer.count++;
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }
} // namespace adaflo_hip
