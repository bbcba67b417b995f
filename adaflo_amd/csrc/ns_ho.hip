// ns_ho.hip -- Taylor-Hood Q_k/Q_{k-1} sweep kernel for the higher degrees (k = 3, 4, 5):
// NavierStokesMatrix::vmult / velocity_vmult with constant coefficients
// (source/navier_stokes_matrix.cc:601-916, vmult and vmult_velocity branches).
//
// Same structure as the Q2/Q1 kernel (ns_q2.hip) -- a workgroup owns a column of TCX x TCY
// cells and sweeps z, nodes shared between cells of the tile are combined through LDS, nodes
// shared between workgroups go through slabs + a fix-up kernel (no atomics, no memset) -- but
// with the work split the way sum factorisation wants it at higher degree:
//   * (k+1)^2 threads per cell; thread (i,j) owns the z-LINE of nodes / quadrature points
//     (i,j,0..k) in registers, so the z contractions, the quadrature-point physics and the
//     carry of the top node plane into the next cell layer need no LDS at all;
//   * for the x and y contractions the same (k+1)^2 threads re-distribute over the (k+1)^2
//     lines of that direction: a thread reads ONE line from LDS, applies the 1D matrix with
//     wave-uniform coefficients (scalar registers) and writes the line back;
//   * collocation: interpolate to the Gauss points once, then differentiate there with the
//     (k+1)x(k+1) collocation derivative (x: fused into the x sweep, z: registers,
//     y: 1D stencil read with the thread's own matrix row).
// FP64 MFMA was considered for the 1D contractions and rejected: on CDNA4 the f64 matrix rate
// equals the f64 vector rate (78.6 TF both), the matrices are only (k+1)x(k+1) <= 6x6 and would
// have to be padded to the 16x16x4 tile, and the kernel is bound by the HBM stream of the
// quadrature-point state (12 doubles per point) anyway.
// STATUS: parity-green (tests/test_ns_parity_gpu.py, tests/test_golden_gpu.py) but opt-in
// (adaflo_set_kernel_variant(ctx, 2)): measured 64^3 Q4/Q3 Newton 3.1 ms at two workgroups per CU
// (register spills: 65 doubles of values + gradients per thread are live in the quadrature loop)
// and 2.5 ms at one workgroup per CU, against 2.3 ms of the generic LDS kernel; PMC counters show
// 72 % of the wave cycles parked in waits (serial stage chain at 1-2 waves per SIMD), not issue.
// A second design was tried and discarded (not in the tree): four lanes per z-line (three velocity
// components + pressure, DPP quad broadcasts for the physics as in ns_q2.hip) and [component]
// [cell][line] regrouping for the LDS sweeps in 1024-thread workgroups.  A lane then holds one
// component, but still ~150 live registers at k = 4 against the 128-VGPR budget of four waves per
// SIMD: the spills went to scratch memory and the kernel ran at 14 ms (64^3 Q4/Q3 Newton).
// The linearisation state is read in the generic layout [cell][12][(k+1)^3] the residual kernel
// writes: for a fixed component the (k+1)^2 threads of a cell read consecutive doubles.
#include "basis.hpp"
#include "kernels.hpp"

namespace adaflo_hip
{
  namespace
  {
    template <int K>
    struct HOCfg;
    template <>
    struct HOCfg<3>
    {
      static constexpr int TCX = 4, TCY = 4; // 16 cells x 16 threads = 256
    };
    template <>
    struct HOCfg<4>
    {
      static constexpr int TCX = 4, TCY = 2; // 8 cells x 25 threads = 200
    };
    template <>
    struct HOCfg<5>
    {
      static constexpr int TCX = 3, TCY = 2; // 6 cells x 36 threads = 216
    };
    // keep the machine scheduler from interleaving independent components / points (it would hoist
    // every load to the top and triple the live registers)
#ifdef HO_FENCES
#define HO_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define HO_FENCE()
#endif
#ifndef HO_LB
#define HO_LB 2
#endif
    constexpr int NTH = 256;
    constexpr int NMAX = 6;

    struct HOArgs
    {
      int    ncx, ncy, ncz, nnx, nny, nnz, npx, npy, npz, tiles_x, tiles_y, LZ, n_chunks;
      double S[NMAX * NMAX];  // S[q][i]  nodal (Gauss-Lobatto) -> Gauss points, degree k
      double D[NMAX * NMAX];  // D[q][q'] collocation derivative at the Gauss points
      double Sp[NMAX * NMAX]; // Sp[q][i] pressure nodes (degree k-1) -> Gauss points
      double w[NMAX];
      double ih[3], det, cA, cB, beta, tau_gd, tmu;
      int    integrate_p;
      uint32_t con_u, con_p;
      const double *src_u, *src_p, *lin;
      double       *dst_u, *dst_p;
      double       *slab_u, *zslab_u, *slab_p, *zslab_p;
    };

    template <int TNX, int TNY>
    __device__ __forceinline__ int rim_index_ho(const int i, const int j)
    {
      if (j == 0)
        return i;
      if (j == TNY - 1)
        return TNX + i;
      if (i == 0)
        return 2 * TNX + j - 1;
      return 2 * TNX + TNY - 2 + j - 1;
    }

    template <int K, int LIN_MODE, bool WITH_P>
    __global__ __launch_bounds__(NTH, HO_LB) void ns_ho_kernel(const HOArgs A)
    {
      using C = HOCfg<K>;
      constexpr int N = K + 1, NP = K, KP = K - 1, NL = N * N, N3 = N * N * N;
      constexpr int TCX = C::TCX, TCY = C::TCY, NCELL = TCX * TCY;
      constexpr int TNX = K * TCX + 1, TNY = K * TCY + 1, TPX = KP * TCX + 1, TPY = KP * TCY + 1;
      constexpr int RIMU = 2 * TNX + 2 * (TNY - 2), RIMP = 2 * TPX + 2 * (TPY - 2);
      static_assert(NCELL * NL <= NTH, "tile does not fit the workgroup");
      extern __shared__ double lds[];
      // (the lanes of the last wave beyond NCELL*NL threads work on a dummy cell slot NCELL of the
      // LDS arrays: no divergent branches around the LDS traffic)
      constexpr int NCS = NCELL * NL == NTH ? NCELL : NCELL + 1;
      double *ARR0 = lds, *ARR1 = ARR0 + NCS * 3 * N3, *PARR = ARR1 + NCS * 3 * N3;
      double *PUB_E = ARR1, *PUB_N = ARR1 + NCS * N * 4 * N; // [cell][j or i][4][k], alias of ARR1
      static_assert(2 * NCS * N * 4 * N <= NCS * 3 * N3, "publish area must fit into ARR1");

      // The 1D matrices live in scalar registers.  Gauss and Gauss-Lobatto points are symmetric
      // about 1/2, so M[q][i] = +-M[n-1-q][m-1-i]: only the first half of each (flattened) matrix
      // is ever read, which lets all three stay resident in SGPRs instead of being spilled.
      auto Sv = [&](const int q, const int m) {
        const int f = q * N + m;
        return 2 * f < N * N ? A.S[f] : A.S[N * N - 1 - f];
      };
      auto Dv = [&](const int q, const int m) {
        const int f = q * N + m;
        return 2 * f < N * N ? A.D[f] : -A.D[N * N - 1 - f];
      };
      auto Spv = [&](const int q, const int m) {
        const int f = q * NP + m;
        return 2 * f < N * NP ? A.Sp[f] : A.Sp[N * NP - 1 - f];
      };

      const int  tid    = threadIdx.x;
      const bool active = tid < NCELL * NL;
      const int  cell = active ? tid / NL : NCELL, l = active ? tid % NL : 0;
      const int  i = l % N, j = l / N; // also: (a,b) of the owned quadrature line, line ids of the sweeps
      const int  cxl = cell % TCX, cyl = cell / TCX;

      const long nwg = (long)A.tiles_x * A.tiles_y * A.n_chunks;
      const long wg  = xcd_remap(blockIdx.x, nwg);
      const int  bz = (int)(wg % A.n_chunks), bt = (int)(wg / A.n_chunks);
      const int  bx = bt % A.tiles_x, by = bt / A.tiles_x;
      const int  cz0 = bz * A.LZ, nl = min(A.LZ, A.ncz - cz0);
      const int  tcx = min(TCX, A.ncx - bx * TCX), tcy = min(TCY, A.ncy - by * TCY);
      const bool valid = active && cxl < tcx && cyl < tcy;
      // cells outside the mesh / dummy lanes compute on cell (0,0) of the tile: their results are
      // never used (nothing is emitted, no valid cell collects from them), addresses stay legal
      const int  cx = bx * TCX + (valid ? cxl : 0), cy = by * TCY + (valid ? cyl : 0);
      const bool lastx = valid && cxl == tcx - 1, lasty = valid && cyl == tcy - 1;
      const bool hasW = cxl > 0, hasS = cyl > 0;
      const size_t wgs = (size_t)bt * A.n_chunks + bz;

      // velocity node column of this thread
      const int  I = K * cx + i, J = K * cy + j;
      const bool own_u  = valid && (i < K || lastx) && (j < K || lasty);
      const bool seam_u = (K * cxl + i == TNX - 1 && I < A.nnx - 1) || (K * cyl + j == TNY - 1 && J < A.nny - 1);
      unsigned   cmask  = 0; // in-plane constrained components
      for (int d = 0; d < 3; ++d)
        if ((I == 0 && (A.con_u >> (0 + d) & 1)) || (I == A.nnx - 1 && (A.con_u >> (3 + d) & 1)) ||
            (J == 0 && (A.con_u >> (6 + d) & 1)) || (J == A.nny - 1 && (A.con_u >> (9 + d) & 1)))
          cmask |= 1u << d;
      // pressure node column (threads with i, j < NP)
      const bool pth = valid && i < NP && j < NP;
      const int  Ip = KP * cx + min(i, KP), Jp = KP * cy + min(j, KP);
      const bool own_p  = pth && (i < KP || lastx) && (j < KP || lasty);
      const bool seam_p = (KP * cxl + i == TPX - 1 && Ip < A.npx - 1) || (KP * cyl + j == TPY - 1 && Jp < A.npy - 1);
      const bool pcon   = (Ip == 0 && (A.con_p >> 0 & 1)) || (Ip == A.npx - 1 && (A.con_p >> 1 & 1)) ||
                        (Jp == 0 && (A.con_p >> 2 & 1)) || (Jp == A.npy - 1 && (A.con_p >> 3 & 1));

      // all per-lane predicates in ONE register; the loop re-derives them from an opaque copy
      // (kept as separate booleans the compiler parks ~20 loop-invariant lane masks in SGPR pairs,
      // which pushes the 1D matrices out of the scalar register file)
      enum
      {
        F_OWN_U = 1, F_SEAM_U = 2, F_CON0 = 4, F_OWN_P = 32, F_SEAM_P = 64, F_PCON = 128, F_W = 256, F_S = 512,
        F_IK = 1024, F_JK = 2048, F_PTH = 4096, F_IKP = 8192, F_JKP = 16384
      };
      const unsigned flags = (own_u ? F_OWN_U : 0) | (seam_u ? F_SEAM_U : 0) | (cmask * F_CON0) | (own_p ? F_OWN_P : 0) |
                             (seam_p ? F_SEAM_P : 0) | (pcon ? F_PCON : 0) | ((i == 0 && hasW) ? F_W : 0) |
                             ((j == 0 && hasS) ? F_S : 0) | (i == K ? F_IK : 0) | (j == K ? F_JK : 0) |
                             ((i < NP && j < NP) ? F_PTH : 0) | (i == KP ? F_IKP : 0) | (j == KP ? F_JKP : 0);

      // this thread's row / column of the collocation derivative (y direction)
      double Drow[N], Dcol[N];
#pragma unroll
      for (int m = 0; m < N; ++m)
        {
          Drow[m] = A.D[j * N + m]; // (runtime row: full matrix in memory)
          Dcol[m] = A.D[m * N + j];
        }
      const double wab = A.det * A.w[i] * A.w[j];

      // addresses = wave-uniform base pointer (scalar registers) + 32-bit per-thread element offset
      const unsigned ubase = (unsigned)((J * A.nnx + I) * 3), pbase = (unsigned)(Jp * A.npx + Ip);
      const size_t   plane_u = (size_t)A.nny * A.nnx * 3, plane_p = (size_t)A.npy * A.npx;
      const unsigned loff = (unsigned)((cy * A.ncx + cx) * (NLIN * N3) + j * N + i); // + uniform layer / comp / point part
      auto lidx = [&](const int d, const int c, const int jj, const int ii) {
        return ((cell * 3 + d) * N + c) * NL + jj * N + ii;
      };
      auto pidx = [&](const int c, const int jj, const int ii) { return (cell * N + c) * NL + jj * N + ii; };

      // store of one owned velocity node (3 components) / pressure node
      auto emit_u = [&](const unsigned fl, const int Kz, const int lp, const double *v, const bool ztop) {
        if (!(fl & F_OWN_U))
          return;
        const double *sp = A.src_u + (size_t)Kz * plane_u;
        double       *dp = A.dst_u + (size_t)Kz * plane_u;
#pragma unroll
        for (int d = 0; d < 3; ++d)
          {
            const bool con = (fl & (F_CON0 << d)) || (Kz == 0 && (A.con_u >> (12 + d) & 1)) ||
                             (Kz == A.nnz - 1 && (A.con_u >> (15 + d) & 1));
            if (con)
              dp[ubase + d] = sp[ubase + d]; // :250-252
            else if (fl & F_SEAM_U)
              A.slab_u[((wgs * (K * A.LZ + 1) + lp) * RIMU + rim_index_ho<TNX, TNY>(K * cxl + i, K * cyl + j)) * 3 + d] = v[d];
            else if (ztop)
              A.zslab_u[(wgs * (TNX * TNY) + (K * cyl + j) * TNX + K * cxl + i) * 3 + d] = v[d];
            else
              dp[ubase + d] = v[d];
          }
      };
      auto emit_p = [&](const unsigned fl, const int Kz, const int lp, const double v, const bool ztop) {
        if (!(fl & F_OWN_P) || !WITH_P || !A.integrate_p)
          return;
        const double *sp = A.src_p + (size_t)Kz * plane_p;
        double       *dp = A.dst_p + (size_t)Kz * plane_p;
        const bool    con = (fl & F_PCON) || (Kz == 0 && (A.con_p >> 4 & 1)) || (Kz == A.npz - 1 && (A.con_p >> 5 & 1));
        if (con)
          dp[pbase] = -sp[pbase]; // :253-255
        else if (fl & F_SEAM_P)
          A.slab_p[(wgs * (KP * A.LZ + 1) + lp) * RIMP + rim_index_ho<TPX, TPY>(KP * cxl + i, KP * cyl + j)] = v;
        else if (ztop)
          A.zslab_p[wgs * (TPX * TPY) + (KP * cyl + j) * TPX + KP * cxl + i] = v;
        else
          dp[pbase] = v;
      };

      double src_top[3] = {0., 0., 0.}, srcp_top = 0.; // raw src of the top node plane of the previous layer
      double carry[3] = {0., 0., 0.}, carry_p = 0.;    // assembled top-plane sums of the previous layer
      {
        const double *pl = A.src_u + (size_t)(K * cz0) * plane_u;
#pragma unroll
        for (int d = 0; d < 3; ++d)
          src_top[d] = pl[ubase + d];
        if (WITH_P)
          srcp_top = (A.src_p + (size_t)(KP * cz0) * plane_p)[pbase];
      }

      for (int layer = 0; layer < nl; ++layer)
        {
          const int cz = cz0 + layer;
          // ---- nodal values of my z-line (read_dof_values: constrained entries read as zero) ----
          double U[3][N], P[N];
#pragma unroll
          for (int d = 0; d < 3; ++d)
            U[d][0] = src_top[d];
          P[0] = srcp_top;
#pragma unroll
          for (int k = 1; k < N; ++k)
            {
              const double *pl = A.src_u + (size_t)(K * cz + k) * plane_u;
#pragma unroll
              for (int d = 0; d < 3; ++d)
                U[d][k] = pl[ubase + d];
            }
          P[KP] = 0.;
          if (WITH_P)
            {
#pragma unroll
              for (int k = 1; k < NP; ++k)
                P[k] = (A.src_p + (size_t)(KP * cz + k) * plane_p)[pbase];
            }
          unsigned fl = flags;
          asm volatile("" : "+v"(fl));
#pragma unroll
          for (int d = 0; d < 3; ++d)
            {
              src_top[d] = U[d][K];
              const bool zlo = cz == 0 && (A.con_u >> (12 + d) & 1), zhi = cz == A.ncz - 1 && (A.con_u >> (15 + d) & 1);
              if (fl & (F_CON0 << d))
                {
#pragma unroll
                  for (int k = 0; k < N; ++k)
                    U[d][k] = 0.;
                }
              if (zlo)
                U[d][0] = 0.;
              if (zhi)
                U[d][K] = 0.;
            }
          srcp_top = P[KP];
          if (WITH_P)
            {
              if (fl & F_PCON)
                {
#pragma unroll
                  for (int k = 0; k < NP; ++k)
                    P[k] = 0.;
                }
              if (cz == 0 && (A.con_p >> 4 & 1))
                P[0] = 0.;
              if (cz == A.ncz - 1 && (A.con_p >> 5 & 1))
                P[KP] = 0.;
            }

          // ---- Z: nodal z-line -> Gauss points (registers), hand over to the y sweep ----------
          {
#pragma unroll
              for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int c = 0; c < N; ++c)
                  {
                    double s = 0.;
#pragma unroll
                    for (int k = 0; k < N; ++k)
                      s += Sv(c, k) * U[d][k];
                    ARR0[lidx(d, c, j, i)] = s;
                    HO_FENCE();
                  }
              if (WITH_P)
                {
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    {
                      double s = 0.;
#pragma unroll
                      for (int k = 0; k < NP; ++k)
                        s += Spv(c, k) * P[k];
                      PARR[pidx(c, j, i)] = s;
                    }
                }
            }
          __syncthreads();
          // ---- Y: line (x = i, level c = j) along y ------------------------------------------
          {
              const int ii = i, c = j;
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  double in[N];
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    in[m] = ARR0[lidx(d, c, m, ii)];
#pragma unroll
                  for (int b = 0; b < N; ++b)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < N; ++m)
                        s += Sv(b, m) * in[m];
                      ARR0[lidx(d, c, b, ii)] = s;
                    }
                  HO_FENCE();
                }
              if (WITH_P && ii < NP)
                {
                  double in[NP];
#pragma unroll
                  for (int m = 0; m < NP; ++m)
                    in[m] = PARR[pidx(c, m, ii)];
#pragma unroll
                  for (int b = 0; b < N; ++b)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < NP; ++m)
                        s += Spv(b, m) * in[m];
                      PARR[pidx(c, b, ii)] = s;
                    }
                }
            }
          __syncthreads();
          // ---- X: line (y = i, level c = j) along x; values and d/dx at the Gauss points -------
          {
              const int b = i, c = j;
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  double in[N], v[N];
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    in[m] = ARR0[lidx(d, c, b, m)];
#pragma unroll
                  for (int a = 0; a < N; ++a)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < N; ++m)
                        s += Sv(a, m) * in[m];
                      v[a] = s;
                    }
#pragma unroll
                  for (int a = 0; a < N; ++a)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < N; ++m)
                        s += Dv(a, m) * v[m];
                      ARR0[lidx(d, c, b, a)] = v[a];
                      ARR1[lidx(d, c, b, a)] = s;
                    }
                  HO_FENCE();
                }
              if (WITH_P)
                {
                  double in[NP];
#pragma unroll
                  for (int m = 0; m < NP; ++m)
                    in[m] = PARR[pidx(c, b, m)];
#pragma unroll
                  for (int a = 0; a < N; ++a)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < NP; ++m)
                        s += Spv(a, m) * in[m];
                      PARR[pidx(c, b, a)] = s;
                    }
                }
            }
          __syncthreads();
          // ---- G: my quadrature line (a = i, b = j, c = 0..k): values and real-space gradients --
          double val[3][N], gx[3][N], gy[3][N], gz[3][N], pv[N];
#pragma unroll
          for (int d = 0; d < 3; ++d)
            {
#pragma unroll
              for (int c = 0; c < N; ++c)
                {
                  double s = 0.;
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    s += Drow[m] * ARR0[lidx(d, c, m, i)];
                  gy[d][c]  = s * A.ih[1];
                  gx[d][c]  = ARR1[lidx(d, c, j, i)] * A.ih[0];
                  val[d][c] = ARR0[lidx(d, c, j, i)];
                  HO_FENCE();
                }
#pragma unroll
              for (int c = 0; c < N; ++c)
                {
                  double s = 0.;
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    s += Dv(c, m) * val[d][m];
                  gz[d][c] = s * A.ih[2];
                }
            }
#pragma unroll
          for (int c = 0; c < N; ++c)
            pv[c] = WITH_P ? PARR[pidx(c, j, i)] : 0.;
          __syncthreads(); // everybody has read ARR0 / ARR1 / PARR

          // ---- quadrature-point operation (navier_stokes_matrix.cc:702-893) -------------------
          const double *lin = A.lin + (size_t)cz * A.ncy * A.ncx * (NLIN * N3);
#pragma unroll
          for (int c = 0; c < N; ++c)
            {
              double lu[3] = {0., 0., 0.}, lg[3][3] = {{0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
              if (LIN_MODE != 2)
                {
#pragma unroll
                  for (int d = 0; d < 3; ++d)
                    lu[d] = (lin + d * N3 + c * NL)[loff];
                  if (LIN_MODE == 0)
                    {
#pragma unroll
                      for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                          lg[d][e] = (lin + (3 + 3 * d + e) * N3 + c * NL)[loff];
                    }
                  else
                    lg[0][0] = (lin + 3 * N3 + c * NL)[loff]; // div of the linearisation point
                }
              const double jxw = wab * A.w[c];
              const double u[3]    = {val[0][c], val[1][c], val[2][c]};
              const double g[3][3] = {{gx[0][c], gy[0][c], gz[0][c]}, {gx[1][c], gy[1][c], gz[1][c]}, {gx[2][c], gy[2][c], gz[2][c]}};
              const double div = g[0][0] + g[1][1] + g[2][2];
              double       conv[3];
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  double res = 0.;
                  if (LIN_MODE == 0) // Newton :802-816
                    {
                      res = A.beta * (div * lu[d] + (lg[0][0] + lg[1][1] + lg[2][2]) * u[d]);
#pragma unroll
                      for (int e = 0; e < 3; ++e)
                        res += lu[e] * g[d][e] + u[e] * lg[d][e];
                    }
                  else if (LIN_MODE == 1) // Picard-type :817-826
                    {
                      res = A.beta * lg[0][0] * u[d];
#pragma unroll
                      for (int e = 0; e < 3; ++e)
                        res += lu[e] * g[d][e];
                    }
                  conv[d] = A.cA * u[d] + A.cB * res; // :717,:827-835
                }
              const double diag = A.tau_gd * div - (WITH_P ? pv[c] : 0.);
              pv[c]             = -div * jxw; // :853-856
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  val[d][c] = conv[d] * jxw;
                  // :859-892 row d of tmu (grad u + grad u^T) + (tau_gd div - p) I, times JxW J^{-1}
                  gx[d][c] = (A.tmu * (g[d][0] + g[0][d]) + (d == 0 ? diag : 0.)) * (jxw * A.ih[0]);
                  gy[d][c] = (A.tmu * (g[d][1] + g[1][d]) + (d == 1 ? diag : 0.)) * (jxw * A.ih[1]);
                  gz[d][c] = (A.tmu * (g[d][2] + g[2][d]) + (d == 2 ? diag : 0.)) * (jxw * A.ih[2]);
                }
              HO_FENCE();
            }

          // ---- integrate: transposed derivative in z (registers) and y (LDS), then the sweeps ---
          {
#pragma unroll
              for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int c = 0; c < N; ++c)
                  ARR1[lidx(d, c, j, i)] = gy[d][c];
            }
          __syncthreads();
#pragma unroll
          for (int d = 0; d < 3; ++d)
            {
              double acc[N];
#pragma unroll
              for (int c = 0; c < N; ++c)
                {
                  double s = val[d][c];
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    s += Dcol[m] * ARR1[lidx(d, c, m, i)] + Dv(m, c) * gz[d][m];
                  acc[c] = s;
                  HO_FENCE();
                }
#pragma unroll
              for (int c = 0; c < N; ++c)
                val[d][c] = acc[c];
            }
          __syncthreads(); // ARR1 is read, it now takes the x-derivative test values
          {
#pragma unroll
              for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int c = 0; c < N; ++c)
                  {
                    ARR0[lidx(d, c, j, i)] = val[d][c];
                    ARR1[lidx(d, c, j, i)] = gx[d][c];
                  }
              if (WITH_P)
                {
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    PARR[pidx(c, j, i)] = pv[c];
                }
            }
          __syncthreads();
          // ---- XT: line (y = i, level c = j): + D^T (x part), then S^T along x ------------------
          {
              const int b = i, c = j;
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  double t[N], gl[N];
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    {
                      t[m]  = ARR0[lidx(d, c, b, m)];
                      gl[m] = ARR1[lidx(d, c, b, m)];
                    }
#pragma unroll
                  for (int a = 0; a < N; ++a)
                    {
                      double s = t[a];
#pragma unroll
                      for (int m = 0; m < N; ++m)
                        s += Dv(m, a) * gl[m];
                      t[a] = s;
                    }
#pragma unroll
                  for (int ii = 0; ii < N; ++ii)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < N; ++m)
                        s += Sv(m, ii) * t[m];
                      ARR0[lidx(d, c, b, ii)] = s;
                    }
                  HO_FENCE();
                }
              if (WITH_P)
                {
                  double t[N];
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    t[m] = PARR[pidx(c, b, m)];
#pragma unroll
                  for (int ii = 0; ii < NP; ++ii)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < N; ++m)
                        s += Spv(m, ii) * t[m];
                      PARR[pidx(c, b, ii)] = s;
                    }
                }
            }
          __syncthreads();
          // ---- YT: line (x = i, level c = j): S^T along y ---------------------------------------
          {
              const int ii = i, c = j;
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  double t[N];
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    t[m] = ARR0[lidx(d, c, m, ii)];
#pragma unroll
                  for (int jj = 0; jj < N; ++jj)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < N; ++m)
                        s += Sv(m, jj) * t[m];
                      ARR0[lidx(d, c, jj, ii)] = s;
                    }
                  HO_FENCE();
                }
              if (WITH_P && ii < NP)
                {
                  double t[N];
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    t[m] = PARR[pidx(c, m, ii)];
#pragma unroll
                  for (int jj = 0; jj < NP; ++jj)
                    {
                      double s = 0.;
#pragma unroll
                      for (int m = 0; m < N; ++m)
                        s += Spv(m, jj) * t[m];
                      PARR[pidx(c, jj, ii)] = s;
                    }
                }
            }
          __syncthreads();
          // ---- ZT: my nodal z-line: S^T along z (registers) -------------------------------------
          double R[3][N], Rp[N];
#pragma unroll
          for (int d = 0; d < 3; ++d)
            {
              double t[N];
#pragma unroll
              for (int c = 0; c < N; ++c)
                t[c] = ARR0[lidx(d, c, j, i)];
#pragma unroll
              for (int k = 0; k < N; ++k)
                {
                  double s = 0.;
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    s += Sv(c, k) * t[c];
                  R[d][k] = s;
                }
              HO_FENCE();
            }
#pragma unroll
          for (int k = 0; k < N; ++k)
            Rp[k] = 0.;
          fl = flags;
          asm volatile("" : "+v"(fl));
          if (WITH_P && (fl & F_PTH))
            {
              double t[N];
#pragma unroll
              for (int c = 0; c < N; ++c)
                t[c] = PARR[pidx(c, j, i)];
#pragma unroll
              for (int k = 0; k < NP; ++k)
                {
                  double s = 0.;
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    s += Spv(c, k) * t[c];
                  Rp[k] = s;
                }
            }
          // ---- combine the cells of the tile per owned node -------------------------------------
          // (ARR1 was last read in XT, two barriers ago: its space now holds the published faces)
          {
              if (fl & F_IK)
                {
#pragma unroll
                  for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int k = 0; k < N; ++k)
                      PUB_E[((cell * N + j) * 4 + d) * N + k] = R[d][k];
                }
              if (fl & F_JK)
                {
#pragma unroll
                  for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int k = 0; k < N; ++k)
                      PUB_N[((cell * N + i) * 4 + d) * N + k] = R[d][k];
                }
              if (WITH_P && (fl & F_PTH))
                {
                  if (fl & F_IKP)
                    {
#pragma unroll
                      for (int k = 0; k < NP; ++k)
                        PUB_E[((cell * N + j) * 4 + 3) * N + k] = Rp[k];
                    }
                  if (fl & F_JKP)
                    {
#pragma unroll
                      for (int k = 0; k < NP; ++k)
                        PUB_N[((cell * N + i) * 4 + 3) * N + k] = Rp[k];
                    }
                }
            }
          __syncthreads();
          // W neighbour's east face row j, S neighbour's north face column i, SW corner
          if (fl & F_W)
            {
#pragma unroll
              for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int k = 0; k < N; ++k)
                  R[d][k] += PUB_E[(((cell - 1) * N + j) * 4 + d) * N + k];
              if (WITH_P)
                {
#pragma unroll
                  for (int k = 0; k < NP; ++k)
                    Rp[k] += PUB_E[(((cell - 1) * N + j) * 4 + 3) * N + k];
                }
            }
          if (fl & F_S)
            {
#pragma unroll
              for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int k = 0; k < N; ++k)
                  R[d][k] += PUB_N[(((cell - TCX) * N + i) * 4 + d) * N + k];
              if (WITH_P)
                {
#pragma unroll
                  for (int k = 0; k < NP; ++k)
                    Rp[k] += PUB_N[(((cell - TCX) * N + i) * 4 + 3) * N + k];
                }
            }
          if ((fl & (F_W | F_S)) == (F_W | F_S))
            {
#pragma unroll
              for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int k = 0; k < N; ++k)
                  R[d][k] += PUB_E[(((cell - TCX - 1) * N + K) * 4 + d) * N + k];
              if (WITH_P)
                {
#pragma unroll
                  for (int k = 0; k < NP; ++k)
                    Rp[k] += PUB_E[(((cell - TCX - 1) * N + KP) * 4 + 3) * N + k];
                }
            }
          // ---- emit the finished planes, carry the top plane -----------------------------------
          R[0][0] += carry[0];
          R[1][0] += carry[1];
          R[2][0] += carry[2];
          Rp[0] += carry_p;
          // constrained rows only exist in tiles / layers at the domain boundary: everybody else takes
          // the branch-light path (one exec region per store target instead of one per value)
          const bool slow = __builtin_amdgcn_readfirstlane(__any((fl & (7 * F_CON0 | F_PCON)) != 0u)) ||
                            (cz == 0 && ((A.con_u >> 12 & 7u) || (A.con_p >> 4 & 1u)));
          if (slow)
            {
#pragma unroll
              for (int k = 0; k < K; ++k)
                {
                  const double v[3] = {R[0][k], R[1][k], R[2][k]};
                  emit_u(fl, K * cz + k, K * layer + k, v, false);
                }
              if (WITH_P)
                {
#pragma unroll
                  for (int k = 0; k < KP; ++k)
                    emit_p(fl, KP * cz + k, KP * layer + k, Rp[k], false);
                }
            }
          else
            {
              if (fl & F_OWN_U)
                {
                  if (fl & F_SEAM_U)
                    {
                      double *sl = A.slab_u + (wgs * (K * A.LZ + 1) + K * layer) * (RIMU * 3);
                      const unsigned r3 = rim_index_ho<TNX, TNY>(K * cxl + i, K * cyl + j) * 3;
#pragma unroll
                      for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int d = 0; d < 3; ++d)
                          sl[k * (RIMU * 3) + r3 + d] = R[d][k];
                    }
                  else
                    {
#pragma unroll
                      for (int k = 0; k < K; ++k)
                        {
                          double *dp = A.dst_u + (size_t)(K * cz + k) * plane_u;
#pragma unroll
                          for (int d = 0; d < 3; ++d)
                            dp[ubase + d] = R[d][k];
                        }
                    }
                }
              if (WITH_P && A.integrate_p && (fl & F_OWN_P))
                {
                  if (fl & F_SEAM_P)
                    {
                      double *sl = A.slab_p + (wgs * (KP * A.LZ + 1) + KP * layer) * RIMP;
                      const unsigned r = rim_index_ho<TPX, TPY>(KP * cxl + i, KP * cyl + j);
#pragma unroll
                      for (int k = 0; k < KP; ++k)
                        sl[k * RIMP + r] = Rp[k];
                    }
                  else
                    {
#pragma unroll
                      for (int k = 0; k < KP; ++k)
                        (A.dst_p + (size_t)(KP * cz + k) * plane_p)[pbase] = Rp[k];
                    }
                }
            }
#pragma unroll
          for (int d = 0; d < 3; ++d)
            carry[d] = R[d][K];
          carry_p = Rp[KP];
          // (the next layer's Z stage writes ARR0 / PARR only; the publish area is rewritten
          // after several barriers)
        }
      // ---- top plane of the chunk -------------------------------------------------------------
      {
        const int  cze  = cz0 + nl;
        const bool ztop = cze < A.ncz;
        emit_u(flags, K * cze, K * nl, carry, ztop);
        if (WITH_P)
          emit_p(flags, KP * cze, KP * nl, carry_p, ztop);
      }
    }

    // second pass: the low-rim tile of a shared node adds the partial sums of the other sharers
    template <int TNX, int TNY, int NC, int DEGZ>
    __device__ __forceinline__ void ho_fix_rim(const HOArgs &A, const long bt, const int Kz, double *dst,
                                               const double *slab, const double *zslab, const int nn_x,
                                               const int nn_y, const int nn_z, const uint32_t con)
    {
      constexpr int RIM = 2 * TNX + 2 * (TNY - 2), NE = (TNX + TNY - 1) * NC;
      const int     bx = (int)(bt % A.tiles_x), by = (int)(bt / A.tiles_x);
      const int     ppc  = DEGZ * A.LZ + 1;
      const int     c_hi = min(Kz / (DEGZ * A.LZ), A.n_chunks - 1);
      const int     lp   = Kz - DEGZ * A.LZ * c_hi;
      const bool    zb   = lp == 0 && c_hi > 0;
      for (int e = threadIdx.x; e < NE; e += 64)
        {
          const int comp = e % NC, s = e / NC;
          const int i = s < TNX ? s : 0, j = s < TNX ? 0 : s - TNX + 1;
          const int I = (TNX - 1) * bx + i, J = (TNY - 1) * by + j;
          if (I >= nn_x || J >= nn_y)
            continue;
          const bool seam_x = i == 0 && I > 0, seam_y = j == 0 && J > 0;
          if (!(seam_x || seam_y))
            continue;
          if ((i == TNX - 1 && I < nn_x - 1) || (j == TNY - 1 && J < nn_y - 1))
            continue; // owned by another tile
          if (on_constrained_face(I, J, Kz, nn_x, nn_y, nn_z, con, NC == 1 ? 1 : 3, comp))
            continue;
          double sum = 0.;
          for (int dy = 0; dy <= (seam_y ? 1 : 0); ++dy)
            for (int dx = 0; dx <= (seam_x ? 1 : 0); ++dx)
              {
                if (dx == 0 && dy == 0)
                  continue;
                const long tb = (long)(by - dy) * A.tiles_x + bx - dx;
                const int  r  = rim_index_ho<TNX, TNY>(i + (TNX - 1) * dx, j + (TNY - 1) * dy);
                sum += slab[(((tb * A.n_chunks + c_hi) * ppc + lp) * RIM + r) * NC + comp];
                if (zb)
                  sum += slab[(((tb * A.n_chunks + c_hi - 1) * ppc + DEGZ * A.LZ) * RIM + r) * NC + comp];
              }
          if (zb)
            sum += zslab[((bt * A.n_chunks + c_hi - 1) * (TNX * TNY) + j * TNX + i) * NC + comp];
          dst[((long)(Kz * (long)nn_y + J) * nn_x + I) * NC + comp] += sum;
        }
    }

    template <int TNX, int TNY, int NC, int DEGZ>
    __device__ __forceinline__ void ho_fix_zplane(const HOArgs &A, const long bt, const int m, double *dst,
                                                  const double *zslab, const int nn_x, const int nn_y,
                                                  const int nn_z, const uint32_t con)
    {
      const int bx = (int)(bt % A.tiles_x), by = (int)(bt / A.tiles_x);
      const int Kz = DEGZ * A.LZ * m;
      for (int e = threadIdx.x; e < TNX * TNY * NC; e += 64)
        {
          const int comp = e % NC, n = e / NC, i = n % TNX, j = n / TNX;
          const int I = (TNX - 1) * bx + i, J = (TNY - 1) * by + j;
          if (I >= nn_x || J >= nn_y)
            continue;
          const bool seam = (i == 0 && I > 0) || (i == TNX - 1 && I < nn_x - 1) || (j == 0 && J > 0) ||
                            (j == TNY - 1 && J < nn_y - 1);
          if (seam || on_constrained_face(I, J, Kz, nn_x, nn_y, nn_z, con, NC == 1 ? 1 : 3, comp))
            continue;
          dst[((long)(Kz * (long)nn_y + J) * nn_x + I) * NC + comp] += zslab[(bt * A.n_chunks + m - 1) * (TNX * TNY * NC) + e];
        }
    }

    template <int K>
    __global__ __launch_bounds__(64) void ns_ho_fixup_kernel(const HOArgs A, const long n1, const long n2,
                                                             const long n3, const long n4)
    {
      using C = HOCfg<K>;
      constexpr int TNX = K * C::TCX + 1, TNY = K * C::TCY + 1, TPX = (K - 1) * C::TCX + 1, TPY = (K - 1) * C::TCY + 1;
      for (long b = blockIdx.x; b < n1 + n2 + n3 + n4; b += gridDim.x)
        {
          if (b < n1)
            ho_fix_rim<TNX, TNY, 3, K>(A, b / A.nnz, (int)(b % A.nnz), A.dst_u, A.slab_u, A.zslab_u, A.nnx, A.nny,
                                       A.nnz, A.con_u);
          else if (b < n1 + n2)
            ho_fix_zplane<TNX, TNY, 3, K>(A, (b - n1) / (A.n_chunks - 1), (int)((b - n1) % (A.n_chunks - 1)) + 1,
                                          A.dst_u, A.zslab_u, A.nnx, A.nny, A.nnz, A.con_u);
          else if (b < n1 + n2 + n3)
            ho_fix_rim<TPX, TPY, 1, K - 1>(A, (b - n1 - n2) / A.npz, (int)((b - n1 - n2) % A.npz), A.dst_p, A.slab_p,
                                           A.zslab_p, A.npx, A.npy, A.npz, A.con_p);
          else
            ho_fix_zplane<TPX, TPY, 1, K - 1>(A, (b - n1 - n2 - n3) / (A.n_chunks - 1),
                                              (int)((b - n1 - n2 - n3) % (A.n_chunks - 1)) + 1, A.dst_p, A.zslab_p,
                                              A.npx, A.npy, A.npz, A.con_p);
        }
    }

    int ensure(DeviceBuffer &b, const size_t need)
    {
      if (b.count >= need)
        return 0;
      if (b.p)
        (void)hipFree(b.p);
      b.p     = nullptr;
      b.count = 0;
      if (hipMalloc(&b.p, need * sizeof(double)) != hipSuccess)
        return ADAFLO_ENOMEM;
      b.count = need;
      return 0;
    }

    template <int K>
    int launch_ho(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u,
                  const double *src_p)
    {
      using C = HOCfg<K>;
      constexpr int N = K + 1, NP = K, N3 = N * N * N, NCELL = C::TCX * C::TCY;
      constexpr int TNX = K * C::TCX + 1, TNY = K * C::TCY + 1, TPX = (K - 1) * C::TCX + 1, TPY = (K - 1) * C::TCY + 1;
      constexpr int RIMU = 2 * TNX + 2 * (TNY - 2), RIMP = 2 * TPX + 2 * (TPY - 2);
      HOArgs A{};
      A.ncx = ctx->desc.ncell[0];
      A.ncy = ctx->desc.ncell[1];
      A.ncz = ctx->desc.ncell[2];
      A.nnx = K * A.ncx + 1;
      A.nny = K * A.ncy + 1;
      A.nnz = K * A.ncz + 1;
      A.npx = (K - 1) * A.ncx + 1;
      A.npy = (K - 1) * A.ncy + 1;
      A.npz = (K - 1) * A.ncz + 1;
      A.tiles_x = (A.ncx + C::TCX - 1) / C::TCX;
      A.tiles_y = (A.ncy + C::TCY - 1) / C::TCY;
      {
        const long tiles = (long)A.tiles_x * A.tiles_y;
        int        lz    = 16;
        while (lz > 2 && tiles * ((A.ncz + lz - 1) / lz) < 1024)
          lz /= 2;
        if (lz > A.ncz)
          lz = A.ncz;
        A.LZ       = lz;
        A.n_chunks = (A.ncz + lz - 1) / lz;
      }
      {
        const Quadrature1D        qu = gauss(N);
        const Shape1D             su = shape_fe_q(K, qu), sp = shape_fe_q(K - 1, qu);
        const std::vector<double> dc = collocation_derivative(qu);
        for (int q = 0; q < N; ++q)
          {
            A.w[q] = qu.w[q];
            for (int m = 0; m < N; ++m)
              {
                A.S[q * N + m] = su.S[q * N + m];
                A.D[q * N + m] = dc[q * N + m];
              }
            for (int m = 0; m < NP; ++m)
              A.Sp[q * NP + m] = sp.S[q * NP + m];
          }
      }
      for (int e = 0; e < 3; ++e)
        A.ih[e] = 1. / ctx->desc.h[e];
      A.det = ctx->desc.h[0] * ctx->desc.h[1] * ctx->desc.h[2];
      const NSDev &P      = ctx->ns;
      const bool   stokes = P.physical_type == ADAFLO_STOKES;
      const double gamma  = P.physical_type == ADAFLO_INCOMPRESSIBLE ? P.weight : 0.;
      A.cA          = stokes ? 0. : gamma * P.density - P.damping; // :717,:827-835; Stokes: no value terms (:708)
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
      A.lin         = (op == OP_VMULT_VELOCITY && ctx->lin_prec.p) ? ctx->lin_prec.p : ctx->lin.p;
      const bool with_p   = op == OP_VMULT;
      const int  lin_mode = (stokes || P.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT) ?
                              2 :
                              (P.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON ? 0 : 1);
      const size_t n_wg = (size_t)A.tiles_x * A.tiles_y * A.n_chunks;
      if (int e = ensure(ctx->q2_slab_u, n_wg * (K * A.LZ + 1) * RIMU * 3))
        return e;
      if (int e = ensure(ctx->q2_zslab_u, n_wg * TNX * TNY * 3))
        return e;
      if (int e = ensure(ctx->q2_slab_p, n_wg * ((K - 1) * A.LZ + 1) * RIMP))
        return e;
      if (int e = ensure(ctx->q2_zslab_p, n_wg * TPX * TPY))
        return e;
      A.slab_u  = ctx->q2_slab_u.p;
      A.zslab_u = ctx->q2_zslab_u.p;
      A.slab_p  = ctx->q2_slab_p.p;
      A.zslab_p = ctx->q2_zslab_p.p;
      if (with_p && !A.integrate_p)
        if (int e = launch_prepare_dst(ctx, dst_p, src_p, ctx->n_nodes_p, 1, A.npx, A.npy, A.npz, A.con_p, -1., true))
          return e;
      const size_t lds_bytes = sizeof(double) * (size_t)((NCELL * N * N == NTH ? NCELL : NCELL + 1) * 7 * N3);
      const dim3   grid((unsigned)n_wg), block(NTH);
      hipError_t   err  = hipSuccess;
      hipEvent_t   stop = ctx->timing ? ctx->kernel_timer.start(ctx->stream) : nullptr;
#define HO_LAUNCH(LM, WP)                                                                                 \
  {                                                                                                       \
    static bool attr_set = false;                                                                         \
    if (!attr_set)                                                                                        \
      {                                                                                                   \
        err      = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_ho_kernel<K, LM, WP>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);      \
        attr_set = err == hipSuccess;                                                                     \
      }                                                                                                   \
    if (err == hipSuccess)                                                                                \
      hipLaunchKernelGGL((ns_ho_kernel<K, LM, WP>), grid, block, lds_bytes, ctx->stream, A);              \
  }
      if (with_p)
        switch (lin_mode)
          {
            case 0:
              HO_LAUNCH(0, true);
              break;
            case 1:
              HO_LAUNCH(1, true);
              break;
            default:
              HO_LAUNCH(2, true);
          }
      else
        switch (lin_mode)
          {
            case 0:
              HO_LAUNCH(0, false);
              break;
            case 1:
              HO_LAUNCH(1, false);
              break;
            default:
              HO_LAUNCH(2, false);
          }
#undef HO_LAUNCH
      if (err != hipSuccess)
        return ADAFLO_EHIP;
      if (stop)
        (void)hipEventRecord(stop, ctx->stream);
      ctx->kernel_timer.count++;
      const long tiles = (long)A.tiles_x * A.tiles_y;
      const bool fix_p = with_p && A.integrate_p;
      const long n1 = tiles * A.nnz, n2 = tiles * (A.n_chunks - 1);
      const long n3 = fix_p ? tiles * A.npz : 0, n4 = fix_p ? tiles * (A.n_chunks - 1) : 0;
      long       nb = n1 + n2 + n3 + n4;
      if (nb > 256 * 512)
        nb = 256 * 512;
      hipLaunchKernelGGL((ns_ho_fixup_kernel<K>), dim3((unsigned)nb), dim3(64), 0, ctx->stream, A, n1, n2, n3, n4);
      return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
    }
  } // namespace

  bool ho_supported(const adaflo_ctx *ctx)
  {
    return ctx->k >= 3 && ctx->k <= 5 && !ctx->rho.p && !ctx->mu.p && !ctx->damp.p;
  }

  int launch_ns_vmult_ho(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u,
                         const double *src_p)
  {
    switch (ctx->k)
      {
        case 3:
          return launch_ho<3>(ctx, op, dst_u, dst_p, src_u, src_p);
        case 4:
          return launch_ho<4>(ctx, op, dst_u, dst_p, src_u, src_p);
        case 5:
          return launch_ho<5>(ctx, op, dst_u, dst_p, src_u, src_p);
        default:
          return ADAFLO_EUNSUPPORTED;
      }
  }
} // namespace adaflo_hip
