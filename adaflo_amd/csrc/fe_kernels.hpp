// fe_kernels.hpp -- device building blocks of the generic (any degree) cell kernels.
//
// One workgroup evaluates one cell; every tensor lives in LDS and the 1D
// contractions of the sum factorisation are distributed over the NT threads of
// the workgroup.  This is the engine's restatement of what deal.II's
// FEEvaluation::{read_dof_values, evaluate, integrate, distribute_local_to_global}
// do at the call sites in source/navier_stokes_matrix.cc:662-671,897-906 and
// source/level_set_okz_*.cc.
//
// General (non-collocation) scheme so that it also covers FE_Q_iso_Q1:
//   values    S_z S_y S_x u,  d/dx: S_z S_y D_x u,  d/dy: S_z D_y S_x u,  d/dz: D_z S_y S_x u
// Index convention: x fastest everywhere; S,D are [q][i] row-major in LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace adaflo_hip
{
  // structured brick description handed to every kernel by value
  struct BrickDev
  {
    int      ncell[3];
    double   h[3];
    uint32_t con_u, con_p, con_ls;
    // Cell loops with a scatter-add run once per COLOUR of the cells (parities of cx, cy, cz: bit d of `colour`):
    // two cells of one colour share no node, so the scatter needs no atomics and the result is bitwise
    // reproducible.  -1: all cells in one launch (kernels that do not scatter).
    int      colour = -1;
    // INDEXED context (round 6; SURVEY 8(b).1, first alternative: what an adapter copies out of a MatrixFree on any conforming
    // Cartesian mesh): per-cell node tables in the cell's lexicographic order, constraint flags per degree of freedom, the
    // cells sorted by colour (no two cells of a colour share a node) -- a launch covers cells [cell_first, cell_first + grid) --
    // and, optionally, the extents of every cell (the diagonal of its Jacobian).  Null tables = the structured brick.
    const int           *idx_u = nullptr, *idx_p = nullptr; // [n_cells][(k+1)^3], [n_cells][k^3]
    const unsigned char *flag_u = nullptr, *flag_p = nullptr; // [n_nodes_u * 3], [n_nodes_p]
    const double        *cell_h = nullptr;                    // [n_cells][3] or null (h above)
    long                 cell_first = 0;
    // hanging nodes (one level of local refinement): a table entry < 0 is hanging node h = -1 - entry; its value is
    // sum_j w_j value(master_j), j in [ptr[h], ptr[h+1]) -- what MatrixFree applies to the cells at a refined face / edge
    const long   *hang_ptr_u = nullptr, *hang_ptr_p = nullptr;
    const int    *hang_master_u = nullptr, *hang_master_p = nullptr;
    const double *hang_weight_u = nullptr, *hang_weight_p = nullptr;
  };
  // number of cells of a colour
  inline long n_cells_of_colour(const int ncell[3], const int colour)
  {
    long n = 1;
    for (int d = 0; d < 3; ++d)
      n *= (ncell[d] - (colour >> d & 1) + 1) / 2;
    return n;
  }

  // lexicographic cell index -> XCD-aware remap of blockIdx (workgroups are
  // dealt round-robin to the 8 XCDs; give each XCD a contiguous range of cells
  // so that neighbouring cells share one L2)
  __device__ __forceinline__ long xcd_remap(const long b, const long n)
  {
    const long per = n / 8;
    if (b >= per * 8)
      return b; // tail
    return (b % 8) * per + b / 8;
  }

  // lexicographic index of the cell a workgroup works on: workgroup `blk` of the launch for brick.colour
  // (or of the one launch over all n_cells cells)
  __device__ __forceinline__ long brick_cell(const BrickDev &b, const long blk, const long n_cells)
  {
    if (b.idx_u)
      return b.cell_first + blk; // (indexed context: the table order is the launch order)
    if (b.colour < 0)
      return xcd_remap(blk, n_cells);
    const int  px = b.colour & 1, py = b.colour >> 1 & 1, pz = b.colour >> 2 & 1;
    const int  nx = (b.ncell[0] - px + 1) / 2, ny = (b.ncell[1] - py + 1) / 2, nz = (b.ncell[2] - pz + 1) / 2;
    const long r  = xcd_remap(blk, (long)nx * ny * nz);
    const int  i = (int)(r % nx), j = (int)((r / nx) % ny), k = (int)(r / ((long)nx * ny));
    return (2 * i + px) + (long)b.ncell[0] * ((2 * j + py) + (long)b.ncell[1] * (2 * k + pz));
  }

  // is node (I,J,K) of a space with nn[d] nodes per direction on a constrained
  // face?  mask bit = stride*f + comp with f = 2*d + side
  __device__ __forceinline__ bool on_constrained_face(const int I, const int J, const int K,
                                                      const int nnx, const int nny, const int nnz,
                                                      const uint32_t mask, const int stride,
                                                      const int comp)
  {
    uint32_t f = 0;
    f |= (I == 0) ? (1u << (stride * 0 + comp)) : 0u;
    f |= (I == nnx - 1) ? (1u << (stride * 1 + comp)) : 0u;
    f |= (J == 0) ? (1u << (stride * 2 + comp)) : 0u;
    f |= (J == nny - 1) ? (1u << (stride * 3 + comp)) : 0u;
    f |= (K == 0) ? (1u << (stride * 4 + comp)) : 0u;
    f |= (K == nnz - 1) ? (1u << (stride * 5 + comp)) : 0u;
    return (f & mask) != 0u;
  }

  // ZF ("flat z"): the z direction has ONE node and ONE quadrature point of weight 1 -- value = the nodal value,
  // d/dz = 0.  This is how the engine runs dim = 2 (NavierStokesMatrix<2>, navier_stokes_matrix.cc:1211; the
  // reference's level-set golden outputs are all 2D): the same kernels with a degenerate third direction, the
  // 1D tables of x / y untouched.
  // ZF == 2: the y direction is flat as well -- dim = 1 (NavierStokesMatrix<1>, navier_stokes_matrix.cc:1210;
  // tests/1d_flow*.prm).  (ZF is an int: 0 = 3D, 1 = flat z, 2 = flat y and z; `true` still means 1.)
  template <int ND, int NQ, int NT, int ZF = 0>
  struct SumFac
  {
    static constexpr bool YF = ZF == 2;
    static constexpr int NDZ = ZF ? 1 : ND, NQZ = ZF ? 1 : NQ, NDY = YF ? 1 : ND, NQY = YF ? 1 : NQ;
    static constexpr int ND3 = ND * NDY * NDZ;
    static constexpr int NQ2 = NQ * NQY;
    static constexpr int NQ3 = NQ * NQY * NQZ;
    static constexpr int T1  = NQ * NDY * NDZ;
    static constexpr int T2  = NQ * NQY * NDZ;
    static constexpr int TMP = 2 * T1 + 3 * T2;
    // entries of the z (y) matrices
    static __device__ __forceinline__ double Sz(const double *S, const int idx) { return ZF ? 1. : S[idx]; }
    static __device__ __forceinline__ double Sy(const double *S, const int idx) { return YF ? 1. : S[idx]; }
    static __device__ __forceinline__ double Dy(const double *D, const int idx) { return YF ? 0. : D[idx]; }

    // u[ND3] -> val[NQ3], gx/gy/gz[NQ3] (reference-cell derivatives)
    template <bool VAL, bool GRAD>
    static __device__ void evaluate(const double *S, const double *D, const double *u, double *val,
                                    double *gx, double *gy, double *gz, double *tmp)
    {
      double *t1 = tmp, *t1d = tmp + T1, *t2 = tmp + 2 * T1, *t2dy = t2 + T2, *t2dx = t2 + 2 * T2;
      const int tid = threadIdx.x;
      for (int o = tid; o < T1; o += NT)
        {
          const int q = o % NQ, base = (o / NQ) * ND;
          double    a = 0., b = 0.;
#pragma unroll
          for (int i = 0; i < ND; ++i)
            {
              const double v = u[base + i];
              a += S[q * ND + i] * v;
              if (GRAD)
                b += D[q * ND + i] * v;
            }
          t1[o] = a;
          if (GRAD)
            t1d[o] = b;
        }
      __syncthreads();
      for (int o = tid; o < T2; o += NT)
        {
          const int q = o % NQ, r = (o / NQ) % NQY, k = o / NQ2;
          double    a = 0., dy = 0., dx = 0.;
#pragma unroll
          for (int j = 0; j < NDY; ++j)
            {
              const double v = t1[(k * NDY + j) * NQ + q];
              a += Sy(S, r * ND + j) * v;
              if (GRAD)
                {
                  dy += Dy(D, r * ND + j) * v;
                  dx += Sy(S, r * ND + j) * t1d[(k * NDY + j) * NQ + q];
                }
            }
          t2[o] = a;
          if (GRAD)
            {
              t2dy[o] = dy;
              t2dx[o] = dx;
            }
        }
      __syncthreads();
      for (int o = tid; o < NQ3; o += NT)
        {
          const int rq = o % NQ2, s = o / NQ2;
          double    a = 0., dz = 0., dy = 0., dx = 0.;
#pragma unroll
          for (int k = 0; k < NDZ; ++k)
            {
              const double v = t2[k * NQ2 + rq];
              a += Sz(S, s * ND + k) * v;
              if (GRAD)
                {
                  if constexpr (!ZF)
                    dz += D[s * ND + k] * v;
                  dy += Sz(S, s * ND + k) * t2dy[k * NQ2 + rq];
                  dx += Sz(S, s * ND + k) * t2dx[k * NQ2 + rq];
                }
            }
          if (VAL)
            val[o] = a;
          if (GRAD)
            {
              gx[o] = dx;
              gy[o] = dy;
              gz[o] = dz;
            }
        }
      __syncthreads();
    }

    // evaluate with the LAST stage kept in registers: thread tid < NQ3 returns value and reference
    // gradient at quadrature point tid (needs NT >= NQ3); nothing but `tmp` is written
    template <bool GRAD>
    static __device__ void evaluate_to_registers(const double *S, const double *D, const double *u, double *tmp,
                                                 double &val, double &gx, double &gy, double &gz)
    {
      static_assert(NT >= NQ3, "one thread per quadrature point");
      double *t1 = tmp, *t1d = tmp + T1, *t2 = tmp + 2 * T1, *t2dy = t2 + T2, *t2dx = t2 + 2 * T2;
      const int tid = threadIdx.x;
      for (int o = tid; o < T1; o += NT)
        {
          const int q = o % NQ, base = (o / NQ) * ND;
          double    a = 0., b = 0.;
#pragma unroll
          for (int i = 0; i < ND; ++i)
            {
              const double v = u[base + i];
              a += S[q * ND + i] * v;
              if (GRAD)
                b += D[q * ND + i] * v;
            }
          t1[o] = a;
          if (GRAD)
            t1d[o] = b;
        }
      __syncthreads();
      for (int o = tid; o < T2; o += NT)
        {
          const int q = o % NQ, r = (o / NQ) % NQY, k = o / NQ2;
          double    a = 0., dy = 0., dx = 0.;
#pragma unroll
          for (int j = 0; j < NDY; ++j)
            {
              const double v = t1[(k * NDY + j) * NQ + q];
              a += Sy(S, r * ND + j) * v;
              if (GRAD)
                {
                  dy += Dy(D, r * ND + j) * v;
                  dx += Sy(S, r * ND + j) * t1d[(k * NDY + j) * NQ + q];
                }
            }
          t2[o] = a;
          if (GRAD)
            {
              t2dy[o] = dy;
              t2dx[o] = dx;
            }
        }
      __syncthreads();
      val = gx = gy = gz = 0.;
      if (tid < NQ3)
        {
          const int rq = tid % NQ2, s = tid / NQ2;
#pragma unroll
          for (int k = 0; k < NDZ; ++k)
            {
              const double v = t2[k * NQ2 + rq];
              val += Sz(S, s * ND + k) * v;
              if (GRAD)
                {
                  if constexpr (!ZF)
                    gz += D[s * ND + k] * v;
                  gy += Sz(S, s * ND + k) * t2dy[k * NQ2 + rq];
                  gx += Sz(S, s * ND + k) * t2dx[k * NQ2 + rq];
                }
            }
        }
      __syncthreads();
    }

    // NB components at once (u + b ND3 -> val + b NQ3, g + (3 b + e) NQ3; tmp: NB * TMP doubles):
    // the same three stages with NB times the work between two barriers -- the cell kernels are
    // bound by barrier + LDS latency, not by LDS bandwidth
    template <int NB>
    static __device__ void evaluate_batch(const double *S, const double *D, const double *u, double *val, double *g,
                                          double *tmp)
    {
      const int tid = threadIdx.x;
      for (int ob = tid; ob < NB * T1; ob += NT)
        {
          const int b = ob / T1, o = ob - b * T1;
          const int q = o % NQ, base = (o / NQ) * ND;
          const double *ub = u + b * ND3;
          double        a = 0., bb = 0.;
#pragma unroll
          for (int i = 0; i < ND; ++i)
            {
              const double v = ub[base + i];
              a += S[q * ND + i] * v;
              bb += D[q * ND + i] * v;
            }
          tmp[b * TMP + o]      = a;
          tmp[b * TMP + T1 + o] = bb;
        }
      __syncthreads();
      for (int ob = tid; ob < NB * T2; ob += NT)
        {
          const int b = ob / T2, o = ob - b * T2;
          const int q = o % NQ, r = (o / NQ) % NQY, k = o / NQ2;
          const double *t1 = tmp + b * TMP, *t1d = t1 + T1;
          double        a = 0., dy = 0., dx = 0.;
#pragma unroll
          for (int j = 0; j < NDY; ++j)
            {
              const double v = t1[(k * NDY + j) * NQ + q];
              a += Sy(S, r * ND + j) * v;
              dy += Dy(D, r * ND + j) * v;
              dx += Sy(S, r * ND + j) * t1d[(k * NDY + j) * NQ + q];
            }
          double *t2 = tmp + b * TMP + 2 * T1;
          t2[o]          = a;
          t2[T2 + o]     = dy;
          t2[2 * T2 + o] = dx;
        }
      __syncthreads();
      for (int ob = tid; ob < NB * NQ3; ob += NT)
        {
          const int b = ob / NQ3, o = ob - b * NQ3;
          const int rq = o % NQ2, s = o / NQ2;
          const double *t2 = tmp + b * TMP + 2 * T1, *t2dy = t2 + T2, *t2dx = t2 + 2 * T2;
          double        a = 0., dz = 0., dy = 0., dx = 0.;
#pragma unroll
          for (int k = 0; k < NDZ; ++k)
            {
              const double v = t2[k * NQ2 + rq];
              a += Sz(S, s * ND + k) * v;
              if constexpr (!ZF)
                dz += D[s * ND + k] * v;
              dy += Sz(S, s * ND + k) * t2dy[k * NQ2 + rq];
              dx += Sz(S, s * ND + k) * t2dx[k * NQ2 + rq];
            }
          val[b * NQ3 + o]           = a;
          g[(3 * b + 0) * NQ3 + o] = dx;
          g[(3 * b + 1) * NQ3 + o] = dy;
          g[(3 * b + 2) * NQ3 + o] = dz;
        }
      __syncthreads();
    }

    // transpose of evaluate_batch
    template <int NB>
    static __device__ void integrate_batch(const double *S, const double *D, const double *tv, const double *tg,
                                           double *out, double *tmp)
    {
      const int tid = threadIdx.x;
      for (int ob = tid; ob < NB * T2; ob += NT)
        {
          const int b = ob / T2, o = ob - b * T2;
          const int rq = o % NQ2, k = o / NQ2;
          const double *v = tv + b * NQ3, *gx = tg + (3 * b) * NQ3, *gy = gx + NQ3, *gz = gy + NQ3;
          double        a = 0., ay = 0., ax = 0.;
#pragma unroll
          for (int s = 0; s < NQZ; ++s)
            {
              a += Sz(S, s * ND + k) * v[s * NQ2 + rq];
              if constexpr (!ZF)
                a += D[s * ND + k] * gz[s * NQ2 + rq];
              ay += Sz(S, s * ND + k) * gy[s * NQ2 + rq];
              ax += Sz(S, s * ND + k) * gx[s * NQ2 + rq];
            }
          double *t2 = tmp + b * TMP + 2 * T1;
          t2[o]          = a;
          t2[T2 + o]     = ay;
          t2[2 * T2 + o] = ax;
        }
      __syncthreads();
      for (int ob = tid; ob < NB * T1; ob += NT)
        {
          const int b = ob / T1, o = ob - b * T1;
          const int q = o % NQ, j = (o / NQ) % NDY, k = o / (NQ * NDY);
          const double *t2 = tmp + b * TMP + 2 * T1, *t2dy = t2 + T2, *t2dx = t2 + 2 * T2;
          double        bb = 0., bx = 0.;
#pragma unroll
          for (int r = 0; r < NQY; ++r)
            {
              bb += Sy(S, r * ND + j) * t2[(k * NQY + r) * NQ + q] + Dy(D, r * ND + j) * t2dy[(k * NQY + r) * NQ + q];
              bx += Sy(S, r * ND + j) * t2dx[(k * NQY + r) * NQ + q];
            }
          tmp[b * TMP + o]      = bb;
          tmp[b * TMP + T1 + o] = bx;
        }
      __syncthreads();
      for (int ob = tid; ob < NB * ND3; ob += NT)
        {
          const int b = ob / ND3, o = ob - b * ND3;
          const int i = o % ND, kj = o / ND;
          const double *t1 = tmp + b * TMP, *t1d = t1 + T1;
          double        c = 0.;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            c += S[q * ND + i] * t1[kj * NQ + q] + D[q * ND + i] * t1d[kj * NQ + q];
          out[b * ND3 + o] = c;
        }
      __syncthreads();
    }

    // transpose of evaluate: out[ND3] = S^T tv + D_x^T tgx + D_y^T tgy + D_z^T tgz
    template <bool VAL, bool GRAD>
    static __device__ void integrate(const double *S, const double *D, const double *tv,
                                     const double *tgx, const double *tgy, const double *tgz,
                                     double *out, double *tmp)
    {
      double *t1 = tmp, *t1d = tmp + T1, *t2 = tmp + 2 * T1, *t2dy = t2 + T2, *t2dx = t2 + 2 * T2;
      const int tid = threadIdx.x;
      for (int o = tid; o < T2; o += NT)
        {
          const int rq = o % NQ2, k = o / NQ2;
          double    a = 0., ay = 0., ax = 0.;
#pragma unroll
          for (int s = 0; s < NQZ; ++s)
            {
              if (VAL)
                a += Sz(S, s * ND + k) * tv[s * NQ2 + rq];
              if (GRAD)
                {
                  if constexpr (!ZF)
                    a += D[s * ND + k] * tgz[s * NQ2 + rq];
                  ay += Sz(S, s * ND + k) * tgy[s * NQ2 + rq];
                  ax += Sz(S, s * ND + k) * tgx[s * NQ2 + rq];
                }
            }
          t2[o] = a;
          if (GRAD)
            {
              t2dy[o] = ay;
              t2dx[o] = ax;
            }
        }
      __syncthreads();
      for (int o = tid; o < T1; o += NT)
        {
          const int q = o % NQ, j = (o / NQ) % NDY, k = o / (NQ * NDY);
          double    b = 0., bx = 0.;
#pragma unroll
          for (int r = 0; r < NQY; ++r)
            {
              b += Sy(S, r * ND + j) * t2[(k * NQY + r) * NQ + q];
              if (GRAD)
                {
                  b += Dy(D, r * ND + j) * t2dy[(k * NQY + r) * NQ + q];
                  bx += Sy(S, r * ND + j) * t2dx[(k * NQY + r) * NQ + q];
                }
            }
          t1[o] = b;
          if (GRAD)
            t1d[o] = bx;
        }
      __syncthreads();
      for (int o = tid; o < ND3; o += NT)
        {
          const int i = o % ND, kj = o / ND;
          double    c = 0.;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            {
              c += S[q * ND + i] * t1[kj * NQ + q];
              if (GRAD)
                c += D[q * ND + i] * t1d[kj * NQ + q];
            }
          out[o] = c;
        }
      __syncthreads();
    }
  };

  // gather the (DEG+1)^3 x NC local values of cell (cx,cy,cz); dof = node*NC+c.
  // RESOLVE: constrained entries read as zero (read_dof_values), else plain.
  template <int DEG, int NC, int NT, bool RESOLVE, int ZF = 0>
  __device__ void gather_cell(const double *__restrict__ vec, double *loc, const int cx,
                              const int cy, const int cz, const int nnx, const int nny,
                              const int nnz, const uint32_t mask)
  {
    constexpr int ND = DEG + 1, NDY = ZF == 2 ? 1 : ND, ND3 = ND * NDY * (ZF ? 1 : ND);
    for (int o = threadIdx.x; o < ND3 * NC; o += NT)
      {
        const int  c = o % NC, l = o / NC;
        const int  i = l % ND, j = (l / ND) % NDY, k = l / (ND * NDY);
        const int  I = cx * DEG + i, J = cy * DEG + j, K = cz * DEG + k;
        const long node = I + (long)nnx * (J + (long)nny * K);
        double     v    = vec[node * NC + c];
        if (RESOLVE && on_constrained_face(I, J, K, nnx, nny, nnz, mask, NC == 1 ? 1 : 3, c))
          v = 0.;
        loc[c * ND3 + l] = v;
      }
  }

  // distribute_local_to_global: scatter-add, constrained rows skipped.  No atomics: the cell loops run colour by
  // colour (BrickDev::colour), the cells of one launch share no node.  A launcher that forgot the colour loop
  // (colour < 0 = all cells in one launch) would race silently: the kernel traps instead.
  template <int DEG, int NC, int NT, int ZF = 0>
  __device__ void scatter_cell(double *__restrict__ vec, const double *loc, const int cx,
                               const int cy, const int cz, const int nnx, const int nny,
                               const int nnz, const uint32_t mask, const int colour)
  {
    if (colour < 0)
      __builtin_trap();
    constexpr int ND = DEG + 1, NDY = ZF == 2 ? 1 : ND, ND3 = ND * NDY * (ZF ? 1 : ND);
    for (int o = threadIdx.x; o < ND3 * NC; o += NT)
      {
        const int  c = o % NC, l = o / NC;
        const int  i = l % ND, j = (l / ND) % NDY, k = l / (ND * NDY);
        const int  I = cx * DEG + i, J = cy * DEG + j, K = cz * DEG + k;
        const long node = I + (long)nnx * (J + (long)nny * K);
        if (!on_constrained_face(I, J, K, nnx, nny, nnz, mask, NC == 1 ? 1 : 3, c))
          vec[node * NC + c] += loc[c * ND3 + l];
      }
  }

  // ---- indexed context: the same two operations through the node table of the cell ---------------------------------
  // (NC = 3: velocity space, node table / flags of degree DEG = k; NC = 1: pressure space)
  template <int DEG, int NC, int NT, bool RESOLVE>
  __device__ void gather_cell_indexed(const double *__restrict__ vec, double *loc, const int *__restrict__ nodes,
                                      const unsigned char *__restrict__ flags, const long *__restrict__ hptr,
                                      const int *__restrict__ hmaster, const double *__restrict__ hweight)
  {
    constexpr int ND3 = (DEG + 1) * (DEG + 1) * (DEG + 1);
    for (int o = threadIdx.x; o < ND3 * NC; o += NT)
      {
        const int c = o % NC, l = o / NC, node = nodes[l];
        double    v = 0.;
        if (node >= 0)
          {
            const long dof = (long)node * NC + c;
            v              = vec[dof];
            if (RESOLVE && flags[dof])
              v = 0.;
          }
        else // hanging node: interpolate its masters (read_dof_values resolves the constraint)
          for (long j = hptr[-1 - node]; j < hptr[-node]; ++j)
            {
              const long dof = (long)hmaster[j] * NC + c;
              if (!(RESOLVE && flags[dof]))
                v += hweight[j] * vec[dof];
            }
        loc[c * ND3 + l] = v;
      }
  }
  template <int DEG, int NC, int NT>
  __device__ void scatter_cell_indexed(double *__restrict__ vec, const double *loc, const int *__restrict__ nodes,
                                       const unsigned char *__restrict__ flags, const long *__restrict__ hptr,
                                       const int *__restrict__ hmaster, const double *__restrict__ hweight)
  {
    constexpr int ND3 = (DEG + 1) * (DEG + 1) * (DEG + 1);
    bool          any_hanging = false;
    for (int o = threadIdx.x; o < ND3 * NC; o += NT)
      {
        const int c = o % NC, l = o / NC, node = nodes[l];
        if (node < 0)
          {
            any_hanging = true;
            continue;
          }
        const long dof = (long)node * NC + c;
        if (!flags[dof])
          vec[dof] += loc[c * ND3 + l];
      }
    // hanging entries: their sums go to the masters with the weights (distribute_local_to_global).  A master may be a
    // regular node of this very cell, and two hanging nodes share masters: one thread per component walks them in table
    // order, behind the plain adds of the whole workgroup -- no atomics, the order of the additions is fixed
    if (hptr && __syncthreads_or(any_hanging))
      {
        __threadfence_block();
        if (threadIdx.x < NC)
          {
            const int c = threadIdx.x;
            for (int l = 0; l < ND3; ++l)
              if (nodes[l] < 0)
                for (long j = hptr[-1 - nodes[l]]; j < hptr[-nodes[l]]; ++j)
                  {
                    const long dof = (long)hmaster[j] * NC + c;
                    if (!flags[dof])
                      vec[dof] += hweight[j] * loc[c * ND3 + l];
                  }
          }
      }
  }
  // brick or table, by what the context carries (`c`: the cell; `resolve`: the caller's mask != "none")
  template <int DEG, int NC, int NT, bool RESOLVE, int ZF>
  __device__ __forceinline__ void gather_any(const BrickDev &b, const double *__restrict__ vec, double *loc, const long c,
                                             const int cx, const int cy, const int cz, const int nnx, const int nny,
                                             const int nnz, const uint32_t mask)
  {
    if (ZF == 0 && b.idx_u)
      {
        constexpr int ND3 = (DEG + 1) * (DEG + 1) * (DEG + 1);
        gather_cell_indexed<DEG, NC, NT, RESOLVE>(vec, loc, (NC == 3 ? b.idx_u : b.idx_p) + c * ND3, NC == 3 ? b.flag_u : b.flag_p,
                                                  NC == 3 ? b.hang_ptr_u : b.hang_ptr_p, NC == 3 ? b.hang_master_u : b.hang_master_p,
                                                  NC == 3 ? b.hang_weight_u : b.hang_weight_p);
      }
    else
      gather_cell<DEG, NC, NT, RESOLVE, ZF>(vec, loc, cx, cy, cz, nnx, nny, nnz, mask);
  }
  template <int DEG, int NC, int NT, int ZF>
  __device__ __forceinline__ void scatter_any(const BrickDev &b, double *__restrict__ vec, const double *loc, const long c,
                                              const int cx, const int cy, const int cz, const int nnx, const int nny,
                                              const int nnz, const uint32_t mask)
  {
    if (ZF == 0 && b.idx_u)
      {
        constexpr int ND3 = (DEG + 1) * (DEG + 1) * (DEG + 1);
        scatter_cell_indexed<DEG, NC, NT>(vec, loc, (NC == 3 ? b.idx_u : b.idx_p) + c * ND3, NC == 3 ? b.flag_u : b.flag_p,
                                          NC == 3 ? b.hang_ptr_u : b.hang_ptr_p, NC == 3 ? b.hang_master_u : b.hang_master_p,
                                          NC == 3 ? b.hang_weight_u : b.hang_weight_p);
      }
    else
      scatter_cell<DEG, NC, NT, ZF>(vec, loc, cx, cy, cz, nnx, nny, nnz, mask, b.colour);
  }
  // extents of cell c
  __device__ __forceinline__ const double *cell_extents(const BrickDev &b, const long c)
  {
    return b.cell_h ? b.cell_h + 3 * c : b.h;
  }
} // namespace adaflo_hip
