// ns_divergence.hip -- NavierStokesMatrix::divergence_vmult_add for Q2/Q1 on the uniform brick as a
// tensor-product stencil (source/navier_stokes_matrix.cc:300-332 driver, :920-961 local_divergence).
//
//   dst_p += w (q, div u),   w = -1 or -viscosity (constant)
//
// With constant coefficients the cell integrals (3-point Gauss: exact for these products) collapse to
//   (q_I, d_x u_x) = Cx (x) My (x) Mz u_x,   C = int N^p_I (N^u_j)' ,   M = int N^p_I N^u_j
// with 5-wide 1D rows (pressure node I couples to the velocity nodes 2I-2 .. 2I+2; one half of the
// row is missing at the domain boundary).  Pure gather: no seams, no atomics, bitwise reproducible.
// A lane owns the pressure node columns (I, J), (I, J + 1) of a z-chunk (two rows per lane since round 4: seven velocity
// rows for two results instead of ten) and marches through the velocity planes
// WITHOUT staging them in LDS: it reads the velocity nodes 2I, 2I+1 (48 contiguous bytes) of the
// rows 2J-2 .. 2J+4, sums them in y per pressure row, forms its own share of the x-sums and the shares its two columns
// have in the rows of the pressure nodes I-1 (node 2I as the last node of their right cell) and I+1
// (both nodes in their left cell), which travel one lane to the left / right (ds_bpermute; the lanes of
// a wave are 62 consecutive nodes of the flattened (J, I) index plus a halo lane on either side, so the
// row ends need no care beyond the masks "has a left / right cell").  Per plane that is
//   A = Cx My u_x + Mx Cy u_y,   B = Mx My u_z,
// and every plane adds  Mz[t] A + Cz[t] B  to the two or three pressure planes it couples to; three
// accumulators rotate in registers (the loop is unrolled over six planes).  Constrained velocity
// entries (:935-939) only occur on node 2I of the first / last pressure column, on the rows 0 and nvy - 1 and on the first / last plane: they are masked by
// per-lane factors.
// History: the first version staged every plane through LDS and gathered 75 values per
// node and plane from it -- bound by its own LDS / VALU instruction stream (0.19 ms at 128^3), not by
// memory.  HBM traffic: the velocity once (+ halo planes per chunk, mostly L2 hits), dst_p read + write.
#include <type_traits>

#include "basis.hpp"
#include "kernels.hpp"

namespace adaflo_hip
{
  namespace
  {
    constexpr int DSW = 62; // owned lanes of a wave

    struct DivArgs
    {
      int           npx, npy, npz, nvx, nvy, nvz, LZ, n_chunks, blocks_per_chunk;
      long          flat;                // npx * (row groups of NR pressure rows)
      double        m[3][2][3], c[2][3]; // per direction h_d int N^p_a N^u_b;  int N^p_a (N^u_b)'
      double        weight;
      uint32_t      con_u, con_p;
      int           plain;               // projection scheme: velocity read without resolving constraints
      const double *src_u;
      double       *dst_p;
    };

    // rows of node `idx` of `n` pressure nodes: offsets -2 .. 2 in velocity nodes
    __device__ __forceinline__ void div_rows(const double (&m)[2][3], const double (&c)[2][3], const int idx, const int n,
                                             double (&M)[5], double (&C)[5])
    {
      const bool lo = idx > 0, hi = idx < n - 1;
      M[0] = lo ? m[1][0] : 0.;
      M[1] = lo ? m[1][1] : 0.;
      M[2] = (lo ? m[1][2] : 0.) + (hi ? m[0][0] : 0.);
      M[3] = hi ? m[0][1] : 0.;
      M[4] = hi ? m[0][2] : 0.;
      C[0] = lo ? c[1][0] : 0.;
      C[1] = lo ? c[1][1] : 0.;
      C[2] = (lo ? c[1][2] : 0.) + (hi ? c[0][0] : 0.);
      C[3] = hi ? c[0][1] : 0.;
      C[4] = hi ? c[0][2] : 0.;
    }

    typedef double dbl2 __attribute__((ext_vector_type(2)));
    typedef dbl2 dbl2_a8 __attribute__((aligned(8)));

    // NR pressure rows per lane (round 4: 2): the rows J0, J0 + 1 share three of their five velocity rows, so a lane
    // reads 7 rows for two results instead of 10 -- the kernel is bound by what its lanes pull through L1 / L2
    // (every velocity row used to be requested 2.5 times), not by HBM
#ifndef DIV_OCC
#define DIV_OCC 2
#endif
#ifndef DIV_NR
#define DIV_NR 2
#endif
    constexpr int NR = DIV_NR, NROWS = 2 * NR + 3;

    __global__ __launch_bounds__(256, DIV_OCC) void q2q1_divergence_kernel(const DivArgs A)
    {
      const long nwg   = (long)A.blocks_per_chunk * A.n_chunks;
      const long wg    = xcd_remap(blockIdx.x, nwg);
      const int  chunk = (int)(wg / A.blocks_per_chunk);
      const int  lane  = threadIdx.x & 63;
      const long g_raw = ((wg % A.blocks_per_chunk) * 4 + (threadIdx.x >> 6)) * DSW + lane - 1;
      const long g     = min(max(g_raw, 0L), A.flat - 1); // flattened (row pair, I)
      const int  JP = (int)(g / A.npx), I = (int)(g - (long)JP * A.npx), J0 = NR * JP;
      const bool own = lane >= 1 && lane <= DSW && g_raw < A.flat;
      const bool xlo = I > 0, xhi = I < A.npx - 1;
      bool       row_ok[NR], con_xy[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r)
        {
          const int J = J0 + r;
          row_ok[r]   = J < A.npy;
          con_xy[r]   = (I == 0 && (A.con_p >> 0 & 1)) || (I == A.npx - 1 && (A.con_p >> 1 & 1)) ||
                      (J == 0 && (A.con_p >> 2 & 1)) || (J == A.npy - 1 && (A.con_p >> 3 & 1));
        }

      // y: pressure row J0 + r uses the velocity rows 2 (J0 + r) - 2 .. + 2 = rows 2 r .. 2 r + 4 of the lane's NROWS rows
      // (clamped to the mesh, their weights are zero beyond it), with the constraint masks of the y faces per component
      // (rows 0 and nvy - 1): My for u_x, u_z, Cy for u_y
      double wy[NR][3][5];
#pragma unroll
      for (int r = 0; r < NR; ++r)
        {
          double    My[5], Cy[5];
          const int J = min(J0 + r, A.npy - 1);
          div_rows(A.m[1], A.c, J, A.npy, My, Cy);
#pragma unroll
          for (int comp = 0; comp < 3; ++comp)
#pragma unroll
            for (int q = 0; q < 5; ++q)
              {
                const int  row = 2 * J - 2 + q;
                const bool con = !A.plain && ((row == 0 && (A.con_u >> (6 + comp) & 1)) || (row == A.nvy - 1 && (A.con_u >> (9 + comp) & 1)));
                wy[r][comp][q] = (con || !row_ok[r]) ? 0. : (comp == 1 ? Cy[q] : My[q]);
              }
        }
      // x: the lane reads two velocity nodes F, S = 2I, 2I+1 -- the last pressure column 2I-1, 2I
      // instead (node 2I+1 would be the first node of the next row, or beyond the vector); weights of
      // F and S in this node's own row (o), in the row of the node to the left (l); to the right the
      // weights are those of a left cell for everybody.  fx: constraint masks of the x faces (node 2I).
      const bool   last = !xhi;
      const double (&mx)[2][3] = A.m[0];
      const double o_m0 = (xlo ? mx[1][2] : 0.) + (xhi ? mx[0][0] : 0.), o_c0 = (xlo ? A.c[1][2] : 0.) + (xhi ? A.c[0][0] : 0.);
      const double oMF = last ? 0. : o_m0, oMS = last ? o_m0 : mx[0][1], oCF = last ? 0. : o_c0, oCS = last ? o_c0 : A.c[0][1];
      const double lMF = last ? 0. : mx[0][2], lMS = last ? mx[0][2] : 0., lCF = last ? 0. : A.c[0][2], lCS = last ? A.c[0][2] : 0.;
      const double from_l = xlo ? 1. : 0., from_r = xhi ? 1. : 0.;
      double       fF[3], fS[3];
#pragma unroll
      for (int comp = 0; comp < 3; ++comp)
        {
          const bool con = !A.plain && ((I == 0 && (A.con_u >> (0 + comp) & 1)) || (I == A.npx - 1 && (A.con_u >> (3 + comp) & 1)));
          fF[comp]       = con && !last ? 0. : 1.;
          fS[comp]       = con && last ? 0. : 1.;
        }
      unsigned off[NROWS]; // byte offsets of node F in the lane's rows of a velocity plane
#pragma unroll
      for (int q = 0; q < NROWS; ++q)
        off[q] = (unsigned)(min(max(2 * J0 - 2 + q, 0), A.nvy - 1) * A.nvx + (last ? 2 * I - 1 : 2 * I)) * 24u;

      const int K0 = chunk * A.LZ, K1 = min(K0 + A.LZ, A.npz);
      const int pbase = 2 * K0 - 2, t0 = K0 > 0 ? 0 : 2, t1 = min(2 * (K1 - 1) + 2, A.nvz - 1) - pbase; // planes pbase + t
      const double (&mz)[2][3] = A.m[2];
      double acc[NR][3];
#pragma unroll
      for (int r = 0; r < NR; ++r)
        acc[r][0] = acc[r][1] = acc[r][2] = 0.;
      // dst_p is read-modify-write: its old value is requested together with the velocity plane whose
      // sums complete the pressure layer (a load right before the store would add one memory latency
      // to every second plane)
      auto writes = [&](const int K, const int r) {
        const bool conz = (K == 0 && (A.con_p >> 4 & 1)) || (K == A.npz - 1 && (A.con_p >> 5 & 1));
        return K >= K0 && K < K1 && own && row_ok[r] && !(con_xy[r] || conz);
      };
      const long pnode = (long)J0 * A.npx + I; // node (I, J0) of a pressure plane; row r: + r * npx
      // plane pbase + t, t = 6 n + PH: pressure planes K0 - 1 + t / 2 (+- 1) in the slots (t / 2) % 3 = PH / 2 (+- 1)
      auto step = [&](auto phase, const int t) {
        constexpr int PH = decltype(phase)::value, sc = (PH / 2) % 3, sm = (PH / 2 + 2) % 3, sp = (PH / 2 + 1) % 3;
        const int     p = pbase + t;
        const int     K = K0 - 1 + t / 2; // the pressure plane of an even velocity plane
        const char   *s = reinterpret_cast<const char *>(A.src_u + (long)p * A.nvy * A.nvx * 3);
        double       *d = A.dst_p + (long)(K - 1) * A.npx * A.npy + pnode;
        bool          wr[NR];
        double        old[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r)
          {
            wr[r]  = PH % 2 == 0 && writes(K - 1, r);
            old[r] = 0.;
            if (wr[r])
              old[r] = d[(long)r * A.npx];
          }
        dbl2 v[NROWS][3];
#pragma unroll
        for (int q = 0; q < NROWS; ++q)
#pragma unroll
          for (int w = 0; w < 3; ++w)
            v[q][w] = *reinterpret_cast<const dbl2_a8 *>(s + off[q] + 16 * w);
        const bool zcon = !A.plain && (p == 0 || p == A.nvz - 1); // (block-uniform) constraints on the z faces
#pragma unroll
        for (int r = 0; r < NR; ++r)
          {
            // (F.x F.y) (F.z S.x) (S.y S.z): y-sums of  My u_x,  Cy u_y,  My u_z  in the columns F and S
            double pF[3], pS[3];
#pragma unroll
            for (int comp = 0; comp < 3; ++comp)
              {
                const double *w5 = wy[r][comp];
                auto          f  = [&](const int q) { return comp == 0 ? v[2 * r + q][0].x : comp == 1 ? v[2 * r + q][0].y : v[2 * r + q][1].x; };
                auto          g2 = [&](const int q) { return comp == 0 ? v[2 * r + q][1].y : comp == 1 ? v[2 * r + q][2].x : v[2 * r + q][2].y; };
                pF[comp] = fF[comp] * (((w5[0] * f(0) + w5[1] * f(1)) + (w5[3] * f(3) + w5[4] * f(4))) + w5[2] * f(2));
                pS[comp] = fS[comp] * (((w5[0] * g2(0) + w5[1] * g2(1)) + (w5[3] * g2(3) + w5[4] * g2(4))) + w5[2] * g2(2));
              }
            if (zcon)
#pragma unroll
              for (int comp = 0; comp < 3; ++comp)
                if (A.con_u >> ((p == 0 ? 12 : 15) + comp) & 1)
                  pF[comp] = pS[comp] = 0.;
            const double a_r = (A.c[1][0] * pF[0] + A.c[1][1] * pS[0]) + (mx[1][0] * pF[1] + mx[1][1] * pS[1]);
            const double b_r = mx[1][0] * pF[2] + mx[1][1] * pS[2];
            const double a_l = (lCF * pF[0] + lCS * pS[0]) + (lMF * pF[1] + lMS * pS[1]);
            const double b_l = lMF * pF[2] + lMS * pS[2];
            const double a = ((oCF * pF[0] + oCS * pS[0]) + (oMF * pF[1] + oMS * pS[1])) +
                             (from_l * __shfl_up(a_r, 1, 64) + from_r * __shfl_down(a_l, 1, 64));
            const double b = (oMF * pF[2] + oMS * pS[2]) + (from_l * __shfl_up(b_r, 1, 64) + from_r * __shfl_down(b_l, 1, 64));
            if (PH % 2 == 0)
              {
                if (wr[r])
                  d[(long)r * A.npx] = old[r] + A.weight * (acc[r][sm] + (mz[0][2] * a + A.c[0][2] * b));
                const bool lo = K > 0, hi = K < A.npz - 1;
                acc[r][sc] += ((lo ? mz[1][2] : 0.) + (hi ? mz[0][0] : 0.)) * a + ((lo ? A.c[1][2] : 0.) + (hi ? A.c[0][0] : 0.)) * b;
                acc[r][sp] = mz[1][0] * a + A.c[1][0] * b;
              }
            else
              {
                acc[r][sc] += mz[0][1] * a + A.c[0][1] * b;
                acc[r][sp] += mz[1][1] * a + A.c[1][1] * b;
              }
          }
      };
      for (int t = 0; t <= t1; t += 6)
        {
#define DIV_STEP(ph)                                                                                                  \
  if (t + ph >= t0 && t + ph <= t1)                                                                                   \
    step(std::integral_constant<int, ph>(), t + ph);
          DIV_STEP(0)
          DIV_STEP(1)
          DIV_STEP(2)
          DIV_STEP(3)
          DIV_STEP(4)
          DIV_STEP(5)
#undef DIV_STEP
        }
      // the top pressure layer of the mesh has no planes above it
      if (K1 == A.npz)
        {
          const int kk = (K1 - 1) - (K0 - 1); // = t / 2 of its own plane
#pragma unroll
          for (int r = 0; r < NR; ++r)
            if (writes(K1 - 1, r))
              A.dst_p[(long)(K1 - 1) * A.npx * A.npy + pnode + (long)r * A.npx] +=
                A.weight * (kk % 3 == 0 ? acc[r][0] : kk % 3 == 1 ? acc[r][1] : acc[r][2]);
        }
    }
  } // namespace

  bool divergence_stencil_supported(const adaflo_ctx *ctx)
  {
    return ctx->k == 2 && !ctx->flat && !ctx->indexed;
  }

  // dst_p += weight (q, div u) on the free pressure rows; `plain`: velocity read without resolving constraints
  int launch_ns_divergence_stencil(adaflo_ctx *ctx, double *dst_p, const double *src_u, const double weight,
                                   const bool plain)
  {
    DivArgs A{};
    A.npx = ctx->desc.ncell[0] + 1, A.npy = ctx->desc.ncell[1] + 1, A.npz = ctx->desc.ncell[2] + 1;
    A.nvx = 2 * ctx->desc.ncell[0] + 1, A.nvy = 2 * ctx->desc.ncell[1] + 1, A.nvz = 2 * ctx->desc.ncell[2] + 1;
    A.flat             = (long)A.npx * ((A.npy + NR - 1) / NR); // (row pairs x columns)
    A.blocks_per_chunk = (int)(((A.flat + DSW - 1) / DSW + 3) / 4); // four independent waves per workgroup
    // z-chunk: 2 LZ + 3 velocity planes per LZ pressure layers; the halo planes are read again (mostly L2 hits).  With
    // two pressure rows per lane (218 VGPRs: two waves per SIMD, 512 workgroups fill the chip) the best chunk is the
    // shortest one that still fits ONE round of workgroups -- 128^3 cells: 0.128 / 0.119 / 0.111 / 0.132 / 0.109 /
    // 0.116 / 0.127 ms for LZ = 2 / 3 / 6 / 8 / 9 (510 workgroups) / 10 / 12 -- but not below 5 layers, where the halo
    // planes and the prologue dominate (64 x 64 x 128: 0.040 / 0.033 / 0.034 / 0.041 ms for LZ = 3 / 5 / 6 / 9);
    // meshes too large for one round take many rounds of 24-layer chunks
    int lz = 24;
    for (int c = 5; c < 24; ++c)
      if ((long)A.blocks_per_chunk * ((A.npz + c - 1) / c) <= 512)
        {
          lz = c;
          break;
        }
    if (const char *e = getenv("ADAFLO_DIV_LZ")) // (tuning knob of scripts/bench_ops.py)
      lz = std::max(1, atoi(e));
    A.LZ       = lz;
    A.n_chunks = (A.npz + A.LZ - 1) / A.LZ;
    const Quadrature1D g  = gauss(3);
    const Shape1D      su = shape_fe_q(2, g), sp = shape_fe_q(1, g);
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 3; ++b)
        {
          double m = 0., c = 0.;
          for (int q = 0; q < 3; ++q)
            {
              m += g.w[q] * sp.S[q * 2 + a] * su.S[q * 3 + b];
              c += g.w[q] * sp.S[q * 2 + a] * su.D[q * 3 + b];
            }
          for (int d = 0; d < 3; ++d)
            A.m[d][a][b] = ctx->desc.h[d] * m;
          A.c[a][b] = c;
        }
    A.weight = weight;
    A.con_u  = ctx->brick.con_u;
    A.con_p  = ctx->brick.con_p;
    A.plain  = plain;
    A.src_u  = src_u;
    A.dst_p  = dst_p;
    const long  n_wg = (long)A.blocks_per_chunk * A.n_chunks;
    hipEvent_t stop = ctx->timing ? ctx->kernel_timer.start(ctx->stream) : nullptr;
    hipLaunchKernelGGL(q2q1_divergence_kernel, dim3((unsigned)n_wg), dim3(256), 0, ctx->stream, A);
    if (stop)
      (void)hipEventRecord(stop, ctx->stream);
    ctx->kernel_timer.count++;
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }
} // namespace adaflo_hip
