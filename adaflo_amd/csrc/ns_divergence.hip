// ns_divergence.hip -- NavierStokesMatrix::divergence_vmult_add for Q2/Q1 on the uniform brick as a
// tensor-product stencil (source/navier_stokes_matrix.cc:300-332 driver, :920-961 local_divergence).
//
//   dst_p += w (q, div u),   w = -1 or -viscosity (constant)
//
// With constant coefficients the cell integrals (3-point Gauss: exact for these products) collapse to
//   (q_I, d_x u_x) = Cx (x) My (x) Mz u_x,   C = int N^p_I (N^u_j)' ,   M = int N^p_I N^u_j
// with 5-wide 1D rows (pressure node I couples to the velocity nodes 2I-2 .. 2I+2; one half of the
// row is missing at the domain boundary).  Pure gather: no seams, no atomics, bitwise reproducible.
// A thread owns one pressure node column (I, J) of a z-chunk and marches through the velocity planes,
// which are staged through LDS once per workgroup (zeroing constrained entries, :935-939) with even
// and odd x-nodes split so that the 25 x 3 reads per plane are stride-1 across the lanes.  Per plane
// it forms  A = Cx My u_x + Mx Cy u_y  and  B = Mx My u_z  and keeps the last five planes in
// registers: div(K) = sum_t Mz[t] A[2K-2+t] + Cz[t] B[2K-2+t].
// HBM traffic: the velocity once (+ halo planes per chunk and tile, mostly L2 hits), dst_p read + write.
#include "basis.hpp"
#include "kernels.hpp"

namespace adaflo_hip
{
  namespace
  {
    constexpr int DPX = 32, DPY = 8;     // pressure nodes per tile (32 x 4 tiles measured 7 % slower)
    constexpr int DVX = 2 * DPX + 3;     // velocity nodes of a tile row
    constexpr int DEW = DPX + 2;         // even x-nodes per row (then DPX + 1 odd ones)

    struct DivArgs
    {
      int           npx, npy, npz, nvx, nvy, nvz, tiles_x, tiles_y, LZ, n_chunks;
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

    __global__ __launch_bounds__(DPX *DPY) void q2q1_divergence_kernel(const DivArgs A)
    {
      constexpr int DNT = DPX * DPY, DVY = 2 * DPY + 3; // threads, velocity rows of a tile plane
      constexpr int DPLANE = 3 * DVY * DVX;             // doubles per staged plane
      constexpr int DLD = (DPLANE + DNT - 1) / DNT;     // staging loads per thread
      __shared__ double lds[2][DPLANE];
      const int  tid = threadIdx.x, tx = tid % DPX, ty = tid / DPX;
      const long nwg = (long)A.tiles_x * A.tiles_y * A.n_chunks;
      const long wg  = xcd_remap(blockIdx.x, nwg);
      const int  chunk = (int)(wg % A.n_chunks), bt = (int)(wg / A.n_chunks);
      const int  bx = bt % A.tiles_x, by = bt / A.tiles_x;
      const int  I0 = bx * DPX, J0 = by * DPY;
      const int  I = min(I0 + tx, A.npx - 1), J = min(J0 + ty, A.npy - 1);
      const bool active = I0 + tx < A.npx && J0 + ty < A.npy;
      const bool con_xy = (I == 0 && (A.con_p >> 0 & 1)) || (I == A.npx - 1 && (A.con_p >> 1 & 1)) ||
                          (J == 0 && (A.con_p >> 2 & 1)) || (J == A.npy - 1 && (A.con_p >> 3 & 1));
      double Mx[5], Cx[5], My[5], Cy[5];
      div_rows(A.m[0], A.c, I, A.npx, Mx, Cx);
      div_rows(A.m[1], A.c, J, A.npy, My, Cy);

      // staging pattern of this thread: entry e = (jl, il, comp) of the velocity tile plane
      int  l_off[DLD];
      long g_off[DLD];
      unsigned ok = 0, zero_lo = 0, zero_hi = 0; // in-plane validity; entries constrained on the z faces only
#pragma unroll
      for (int r = 0; r < DLD; ++r)
        {
          const int e = tid + DNT * r, comp = e % 3, il = (e / 3) % DVX, jl = e / (3 * DVX);
          const int Iv = 2 * I0 - 2 + il, Jv = 2 * J0 - 2 + jl;
          l_off[r] = e < DPLANE ? (comp * DVY + jl) * DVX + ((il & 1) ? DEW + il / 2 : il / 2) : -1;
          bool v   = e < DPLANE && Iv >= 0 && Iv < A.nvx && Jv >= 0 && Jv < A.nvy;
          if (v && !A.plain &&
              ((Iv == 0 && (A.con_u >> (0 + comp) & 1)) || (Iv == A.nvx - 1 && (A.con_u >> (3 + comp) & 1)) ||
               (Jv == 0 && (A.con_u >> (6 + comp) & 1)) || (Jv == A.nvy - 1 && (A.con_u >> (9 + comp) & 1))))
            v = false;
          g_off[r] = v ? ((long)Jv * A.nvx + Iv) * 3 + comp : 0;
          if (v)
            ok |= 1u << r;
          if (v && !A.plain && (A.con_u >> (12 + comp) & 1))
            zero_lo |= 1u << r;
          if (v && !A.plain && (A.con_u >> (15 + comp) & 1))
            zero_hi |= 1u << r;
        }
      // planes travel global -> registers (one plane ahead) -> LDS (two buffers, one barrier per plane; a
      // single buffer with two barriers and a two-plane look-ahead measured 30 % slower)
      auto fetch = [&](double (&pre)[DLD], const int p) {
        const double  *s = A.src_u + (long)p * A.nvy * A.nvx * 3;
        const unsigned m = ok & ~(p == 0 ? zero_lo : 0u) & ~(p == A.nvz - 1 ? zero_hi : 0u);
#pragma unroll
        for (int r = 0; r < DLD; ++r)
          pre[r] = (m >> r & 1u) ? s[g_off[r]] : 0.;
      };
      auto commit = [&](const double (&pre)[DLD], double *buf) {
#pragma unroll
        for (int r = 0; r < DLD; ++r)
          if (l_off[r] >= 0)
            buf[l_off[r]] = pre[r];
      };
      // in-plane sums of the staged plane at this thread's node
      auto plane_sums = [&](const double *buf, double &Ap, double &Bp) {
        double a = 0., b = 0.;
#pragma unroll
        for (int j = 0; j < 5; ++j)
          {
            const double *r0 = buf + (2 * ty + j) * DVX + tx, *r1 = r0 + DVY * DVX, *r2 = r1 + DVY * DVX;
            // nodes 2I-2 .. 2I+2 = E[tx], O[tx], E[tx+1], O[tx+1], E[tx+2]
            const double x0 = r0[0], x1 = r0[DEW], x2 = r0[1], x3 = r0[DEW + 1], x4 = r0[2];
            const double y0 = r1[0], y1 = r1[DEW], y2 = r1[1], y3 = r1[DEW + 1], y4 = r1[2];
            const double z0 = r2[0], z1 = r2[DEW], z2 = r2[1], z3 = r2[DEW + 1], z4 = r2[2];
            const double cx = Cx[0] * x0 + Cx[1] * x1 + Cx[2] * x2 + Cx[3] * x3 + Cx[4] * x4;
            const double my = Mx[0] * y0 + Mx[1] * y1 + Mx[2] * y2 + Mx[3] * y3 + Mx[4] * y4;
            const double mz = Mx[0] * z0 + Mx[1] * z1 + Mx[2] * z2 + Mx[3] * z3 + Mx[4] * z4;
            a += My[j] * cx + Cy[j] * my;
            b += My[j] * mz;
          }
        Ap = a;
        Bp = b;
      };

      const int K0 = chunk * A.LZ, K1 = min(K0 + A.LZ, A.npz);
      const int p0 = max(2 * K0 - 2, 0), p1 = min(2 * (K1 - 1) + 2, A.nvz - 1); // velocity planes of the chunk
      double    Aw[5] = {0., 0., 0., 0., 0.}, Bw[5] = {0., 0., 0., 0., 0.};      // planes 2K-2 .. 2K+2 of the next K
      int K = K0;
      // plane p is window slot p - (2K - 2); slots fill in order and the window moves by two planes per K
      auto process = [&](const int p) {
        double a, b;
        plane_sums(lds[p & 1], a, b);
        const int t = p - (2 * K - 2);
        if (t == 2)
          Aw[2] = a, Bw[2] = b;
        else if (t == 3)
          Aw[3] = a, Bw[3] = b;
        else if (t == 4)
          Aw[4] = a, Bw[4] = b;
        else if (t == 0)
          Aw[0] = a, Bw[0] = b;
        else
          Aw[1] = a, Bw[1] = b;
        // (the top pressure layer has no planes above it: it is due together with the layer below)
        while (K < K1 && (p - (2 * K - 2) == 4 || p == p1))
          {
            const bool lo = K > 0, hi = K < A.npz - 1;
            const double (&mz)[2][3] = A.m[2];
            const double M0 = lo ? mz[1][0] : 0., M1 = lo ? mz[1][1] : 0., M2 = (lo ? mz[1][2] : 0.) + (hi ? mz[0][0] : 0.),
                         M3 = hi ? mz[0][1] : 0., M4 = hi ? mz[0][2] : 0.;
            const double C0 = lo ? A.c[1][0] : 0., C1 = lo ? A.c[1][1] : 0., C2 = (lo ? A.c[1][2] : 0.) + (hi ? A.c[0][0] : 0.),
                         C3 = hi ? A.c[0][1] : 0., C4 = hi ? A.c[0][2] : 0.;
            const double div = M0 * Aw[0] + M1 * Aw[1] + M2 * Aw[2] + M3 * Aw[3] + M4 * Aw[4] + C0 * Bw[0] + C1 * Bw[1] +
                               C2 * Bw[2] + C3 * Bw[3] + C4 * Bw[4];
            const bool conz = (K == 0 && (A.con_p >> 4 & 1)) || (K == A.npz - 1 && (A.con_p >> 5 & 1));
            if (active && !(con_xy || conz))
              {
                const long idx = ((long)K * A.npy + J) * A.npx + I;
                A.dst_p[idx] += A.weight * div;
              }
            Aw[0] = Aw[2], Aw[1] = Aw[3], Aw[2] = Aw[4], Aw[3] = 0., Aw[4] = 0.;
            Bw[0] = Bw[2], Bw[1] = Bw[3], Bw[2] = Bw[4], Bw[3] = 0., Bw[4] = 0.;
            ++K;
          }
      };
      double pre[DLD];
      fetch(pre, p0);
      commit(pre, lds[p0 & 1]);
      __syncthreads();
      for (int p = p0; p <= p1; ++p)
        {
          if (p < p1)
            fetch(pre, p + 1);
          process(p);
          if (p < p1)
            commit(pre, lds[(p + 1) & 1]);
          __syncthreads();
        }
    }
  } // namespace

  bool divergence_stencil_supported(const adaflo_ctx *ctx)
  {
    return ctx->k == 2;
  }

  // dst_p += weight (q, div u) on the free pressure rows; `plain`: velocity read without resolving constraints
  int launch_ns_divergence_stencil(adaflo_ctx *ctx, double *dst_p, const double *src_u, const double weight,
                                   const bool plain)
  {
    DivArgs A{};
    A.npx = ctx->desc.ncell[0] + 1, A.npy = ctx->desc.ncell[1] + 1, A.npz = ctx->desc.ncell[2] + 1;
    A.nvx = 2 * ctx->desc.ncell[0] + 1, A.nvy = 2 * ctx->desc.ncell[1] + 1, A.nvz = 2 * ctx->desc.ncell[2] + 1;
    // chunks of 8 pressure layers (measured best for 64 x 64 x 128 and 128^3 cells among 4 .. 32)
    A.LZ       = 8;
    A.n_chunks = (A.npz + A.LZ - 1) / A.LZ;
    A.tiles_x  = (A.npx + DPX - 1) / DPX;
    A.tiles_y  = (A.npy + DPY - 1) / DPY;
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
    const long  n_wg = (long)A.tiles_x * A.tiles_y * A.n_chunks;
    hipEvent_t stop = ctx->timing ? ctx->kernel_timer.start(ctx->stream) : nullptr;
    hipLaunchKernelGGL(q2q1_divergence_kernel, dim3((unsigned)n_wg), dim3(DPX * DPY), 0, ctx->stream, A);
    if (stop)
      (void)hipEventRecord(stop, ctx->stream);
    ctx->kernel_timer.count++;
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }
} // namespace adaflo_hip
