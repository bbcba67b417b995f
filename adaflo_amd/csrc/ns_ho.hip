// ns_ho.hip -- Taylor-Hood Q_k/Q_{k-1} sweep kernel for the higher degrees (k = 3, 4, 5):
// NavierStokesMatrix::vmult / velocity_vmult with constant coefficients
// (source/navier_stokes_matrix.cc:601-916, vmult and vmult_velocity branches).
//
// Design (round 2).  The generic cell kernel is bound by the LDS
// instruction rate (every 1D contraction reads data AND matrix entries from LDS) and by ~30
// workgroup barriers per cell.  Here:
//   * (k+1)^2 lanes per cell, lane (i,j) owns the z-LINE of nodes / quadrature points (i,j,0..k)
//     in registers: z contractions, the quadrature-point physics and the carry of the top node
//     plane into the next cell layer need no LDS;
//   * a CELL LIVES INSIDE ONE WAVE (k = 3: 4 cells of 16 lanes, k = 4: 2 cells of 25 lanes in
//     32-lane halves, k = 5: 1 cell of 36 lanes).  The x / y contractions go through a
//     wave-private LDS region: lanes write their values, then lane (a,b) reads the x-line
//     (., b) or the y-line (a, .) it needs and contracts it with ITS row of the 1D matrix held
//     in registers.  One wave's LDS operations execute in order, so these exchanges need no
//     s_barrier at all -- only the data dependency (lgkmcnt);
//   * collocation: interpolate to the Gauss points once (z in registers, x, y through LDS), then
//     per quadrature plane c: d/dz from the register line, d/dx, d/dy from x- / y-line reads,
//     physics, transposed derivatives the same way back; values and gradients of ONE plane are
//     live at a time (12 + 12 doubles instead of 65 for the whole line);
//   * z matrices are wave-uniform (scalar registers, symmetric halves), x / y rows per lane;
//   * the workgroup (4 waves) owns a column of TCX x TCY cells and sweeps z; nodes shared
//     between cells of the tile are combined through a double-buffered LDS publish area with ONE
//     workgroup barrier per cell layer; nodes shared between workgroups go through slabs + a
//     fix-up kernel (no atomics, no memset, bitwise reproducible), as in ns_q2.hip.
// The linearisation state is read in the generic layout [cell][12][(k+1)^3] the residual kernel
// writes: for a fixed component and plane the (k+1)^2 lanes of a cell read consecutive doubles.
// FP64 MFMA for the 1D contractions: measured (scripts/dev/mfma_f64_probe.hip) and
// rejected -- v_mfma_f64_16x16x4_f64 issues at the f64 vector rate and a 5x5 matrix fills at most
// 29 % of the tile even with three cells packed on the block diagonal.
//
// Superseded by the x-marching kernel (ns_hox.hip, default since round 4) and kept for comparison: compiled into the library only
// with -DADAFLO_BUILD_VARIANTS (ADAFLO_BUILD_VARIANTS=1 python adaflo_amd/build.py); the product build has the stubs at the end of this
// file and refuses kernel variant 2.
#include "basis.hpp"
#include "kernels.hpp"

#if defined(ADAFLO_BUILD_VARIANTS)
#include <cstring>
#include <utility>
#include <vector>

namespace adaflo_hip
{
  namespace
  {
    // K -> cells per wave (CPW), waves of the workgroup arranged WX x WY over the tile
    template <int K>
    struct HOCfg;
    template <>
    struct HOCfg<3>
    {
      static constexpr int CPW = 4, WX = 1, WY = 4, PLS = 16; // 16 lanes per cell
    };
    template <>
    struct HOCfg<4>
    {
      // 25 lanes per cell in a 32-lane half.  Measured alternative (round 2): planes packed to 25 slots,
      // a compact single-buffered publish area (second barrier per layer) and __launch_bounds__(256, 3)
      // bring the workgroup under 160 KB / 3, but the 168-VGPR budget of three workgroups per CU spills
      // 186 registers: 64^3 cells 1.78 -> 6.07 ms per vmult
      static constexpr int CPW = 2, WX = 2, WY = 2, PLS = 32;
    };
    template <>
    struct HOCfg<5>
    {
      static constexpr int CPW = 1, WX = 2, WY = 2, PLS = 40; // 36 lanes per cell
    };
    template <int K>
    struct HOTile
    {
      using C                  = HOCfg<K>;
      static constexpr int TCX = C::CPW * C::WX, TCY = C::WY; // cells of the workgroup tile
    };
#ifndef HO_LB
#define HO_LB 2
#endif
#ifndef HO_PIPE
#define HO_PIPE 3
#endif
    constexpr int NTH  = 256;
    constexpr int NMAX = 6;
    constexpr int L_TAB_D = NMAX * NMAX, L_TAB_SP = 2 * NMAX * NMAX, L_WAVE = 3 * NMAX * NMAX;

    template <int K>
    constexpr int ho_lds_doubles()
    {
      using C           = HOCfg<K>;
      constexpr int N   = K + 1;
      constexpr int per = C::CPW * (4 * N + 6) * C::PLS; // values [4][N] planes + exchange [6] planes
      return L_WAVE + 4 * per + 4 * (HOTile<K>::TCX * HOTile<K>::TCY) * N * 4 * N;
    }

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
      const double *tab; // [S N*N | D N*N | Sp N*NP | w N] for the scalar loads of the z contractions
      // phased execution for the multi-GPU overlap (as in ns_q2.hip): explicit workgroup list for the main
      // kernel, node filter for the fix-up (1: only nodes on the inter-GPU interface faces `iface`,
      // 2: all other nodes, 0: everything)
      const int *wg_list;
      int        wg_offset, wg_count, fix_mode;
      uint32_t   iface;
    };

    __device__ __forceinline__ bool ho_fix_skip(const HOArgs &A, const int I, const int J, const int Kz,
                                                const int nn_x, const int nn_y, const int nn_z)
    {
      if (A.fix_mode == 0)
        return false;
      const bool on = (I == 0 && (A.iface & 1u)) || (I == nn_x - 1 && (A.iface & 2u)) ||
                      (J == 0 && (A.iface & 4u)) || (J == nn_y - 1 && (A.iface & 8u)) ||
                      (Kz == 0 && (A.iface & 16u)) || (Kz == nn_z - 1 && (A.iface & 32u));
      return A.fix_mode == 1 ? !on : on;
    }

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

    // compiler-only fence between a wave's LDS write and read phases (the hardware executes one
    // wave's LDS operations in order; this keeps the compiler from moving them across)
    __device__ __forceinline__ void wave_sync()
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // LDS reads of the contraction lines are hand-issued ds_read_b64: the compiler would merge them
    // into ds_read2_b64, which runs at half the LDS rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS
    // table).  The results are tied to an explicit counted s_waitcnt (the compiler does not track
    // the completion of asm outputs); LDS returns a wave's operations in order.
    __device__ __forceinline__ unsigned lds_byte_addr(const void *p)
    {
      return (unsigned)(size_t)p; // LDS aperture: the low 32 bits are the LDS byte address
    }
    template <int OFF>
    __device__ __forceinline__ double ds_rd(const unsigned a)
    {
      double v;
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
      return v;
    }
    // wait until at most CNT LDS operations are outstanding; the values listed become usable
    template <int CNT, int NM>
    __device__ __forceinline__ void ds_wait(double (&x)[NM])
    {
      constexpr int C = CNT > 15 ? 15 : CNT;
      static_assert(NM >= 3 && NM <= 6, "line length");
      if constexpr (NM == 3)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "n"(C));
      else if constexpr (NM == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(C));
      else if constexpr (NM == 5)
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]) : "n"(C));
      else
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5])
                     : "n"(C));
    }
    template <int CNT>
    __device__ __forceinline__ void ds_wait1(double &x)
    {
      constexpr int C = CNT > 15 ? 15 : CNT;
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(C));
    }
    // x[m] = lds[base + OFF0 + 8 m STRIDE], m = 0 .. NM-1
    template <int OFF0, int STRIDE, int NM, int... M>
    __device__ __forceinline__ void rd_line_impl(const unsigned b, double (&x)[NM], std::integer_sequence<int, M...>)
    {
      ((x[M] = ds_rd<OFF0 + 8 * M * STRIDE>(b)), ...);
    }
    template <int OFF0, int STRIDE, int NM>
    __device__ __forceinline__ void rd_line(const unsigned b, double (&x)[NM])
    {
      rd_line_impl<OFF0, STRIDE>(b, x, std::make_integer_sequence<int, NM>{});
    }
    template <int C0, int NPL, int NM, int STRIDE, int PLS>
    struct LineReader
    {
      // issue the reads of line element M of plane C (recursion unrolls the compile-time offsets)
      template <int C = 0, int M = 0>
      static __device__ __forceinline__ void issue(const unsigned base, double (&ln)[NPL][NM])
      {
        if constexpr (C < NPL)
          {
            ln[C][M] = ds_rd<8 * ((C0 + C) * PLS + M * STRIDE)>(base);
            if constexpr (M + 1 < NM)
              issue<C, M + 1>(base, ln);
            else
              issue<C + 1, 0>(base, ln);
          }
      }
      template <int C = 0>
      static __device__ __forceinline__ void reduce(const double *coef, double (&ln)[NPL][NM], double *out)
      {
        if constexpr (C < NPL)
          {
            ds_wait<(NPL - 1 - C) * NM>(ln[C]);
            double s = 0.;
#pragma unroll
            for (int m = 0; m < NM; ++m)
              s += coef[m] * ln[C][m];
            out[C] = s;
            reduce<C + 1>(coef, ln, out);
          }
      }
    };
    // out[c] = sum_m coef[m] * lds[base + 8 (c * PLS + m * STRIDE)] for the NPL planes c, with the
    // lane's coefficient vector coef[m] = lds[coef_addr + 8 m CSTRIDE] (row / column of a 1D matrix)
    template <int NPL, int NM, int STRIDE, int PLS, int CSTRIDE>
    __device__ __forceinline__ void ho_contract(const unsigned base, const unsigned coef_addr, double *out)
    {
      double coef[NM], ln[NPL][NM];
      rd_line<0, CSTRIDE>(coef_addr, coef);
      LineReader<0, NPL, NM, STRIDE, PLS>::issue(base, ln);
      ds_wait<(NPL - 1) * NM>(ln[0]);
      ds_wait<(NPL - 1) * NM>(coef);
      LineReader<0, NPL, NM, STRIDE, PLS>::reduce(coef, ln, out);
    }
#define HO_FENCE() __builtin_amdgcn_sched_barrier(0)

    template <int K, int LIN_MODE, bool WITH_P>
    __global__ __launch_bounds__(NTH, HO_LB) void ns_ho_kernel(const HOArgs A)
    {
      using C = HOCfg<K>;
      constexpr int N = K + 1, NP = K, KP = K - 1, NL = N * N, N3 = N * N * N;
      constexpr int CPW = C::CPW, LPC = 64 / CPW, PLS = C::PLS;
      constexpr int TCX = HOTile<K>::TCX, TCY = HOTile<K>::TCY, NCELL = TCX * TCY;
      constexpr int TNX = K * TCX + 1, TNY = K * TCY + 1, TPX = KP * TCX + 1, TPY = KP * TCY + 1;
      constexpr int RIMU = 2 * TNX + 2 * (TNY - 2), RIMP = 2 * TPX + 2 * (TPY - 2);
      constexpr int VALW = CPW * 4 * N * PLS, TGW = CPW * 6 * PLS, PUBSZ = NCELL * N * 4 * N;
      constexpr bool ALL_SLOTS = PLS >= LPC; // every lane has an LDS slot of its own
      static_assert(NL <= LPC && NL <= PLS, "cell does not fit its lane group");
      extern __shared__ double lds[];

      // wave-uniform 1D matrices (z direction): A.tab = [S | D | Sp | w] in global memory, read with
      // scalar loads through a per-phase opaque copy of the pointer.  Gauss and Gauss-Lobatto points
      // are symmetric about 1/2, so M[q][i] = +-M[n-1-q][m-1-i]: only the first half of S / Sp is read
      typedef const double __attribute__((address_space(4))) *ctab_t; // constant address space: scalar loads
      auto symS = [](const ctab_t t, const int q, const int m) {
        const int f = q * N + m;
        return 2 * f < N * N ? t[f] : t[N * N - 1 - f];
      };
      auto symSp = [](const ctab_t t, const int q, const int m) {
        const int f = q * NP + m;
        return 2 * f < N * NP ? t[2 * N * N + f] : t[2 * N * N + N * NP - 1 - f];
      };

      const int tid  = threadIdx.x, lane = tid & 63;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int cw = lane / LPC, l = lane % LPC;
      const bool active = l < NL;
      const int  lc = active ? l : NL - 1;       // lanes beyond the cell shadow its last lane
      const int  i = lc % N, j = lc / N;          // node column / quadrature column (a, b) of this lane
      const int  cxl = (wave % C::WX) * CPW + cw, cyl = wave / C::WX, cell = cyl * TCX + cxl;
      const bool slot_ok = ALL_SLOTS || l < PLS;

      // the per-lane rows of the 1D matrices come from a small LDS table
      for (int e = tid; e < NMAX * NMAX; e += NTH)
        {
          lds[e]            = A.S[e];
          lds[L_TAB_D + e]  = A.D[e];
          lds[L_TAB_SP + e] = A.Sp[e];
        }
      __syncthreads();

      const long nwg = A.wg_list ? (long)A.wg_count : (long)A.tiles_x * A.tiles_y * A.n_chunks;
      long       wg  = xcd_remap(blockIdx.x, nwg);
      if (A.wg_list)
        wg = A.wg_list[A.wg_offset + wg];
      const int  bz = (int)(wg % A.n_chunks), bt = (int)(wg / A.n_chunks);
      const int  bx = bt % A.tiles_x, by = bt / A.tiles_x;
      const int  cz0 = bz * A.LZ, nl = min(A.LZ, A.ncz - cz0);
      const int  tcx = min(TCX, A.ncx - bx * TCX), tcy = min(TCY, A.ncy - by * TCY);
      const bool valid = active && cxl < tcx && cyl < tcy;
      // cells outside the mesh compute on cell (0,0) of the tile: their results are never used
      // (nothing is emitted, no valid cell collects from them), addresses stay legal
      const int  cx = bx * TCX + (cxl < tcx && cyl < tcy ? cxl : 0), cy = by * TCY + (cxl < tcx && cyl < tcy ? cyl : 0);
      const bool lastx = valid && cxl == tcx - 1, lasty = valid && cyl == tcy - 1;
      const bool hasW = cxl > 0, hasS = cyl > 0;
      const size_t wgs = (size_t)bt * A.n_chunks + bz;

      // velocity node column of this lane
      const int  I = K * cx + i, J = K * cy + j;
      const bool own_u  = valid && (i < K || lastx) && (j < K || lasty);
      const bool seam_u = (K * cxl + i == TNX - 1 && I < A.nnx - 1) || (K * cyl + j == TNY - 1 && J < A.nny - 1);
      unsigned   cmask  = 0; // in-plane constrained components
      for (int d = 0; d < 3; ++d)
        if ((I == 0 && (A.con_u >> (0 + d) & 1)) || (I == A.nnx - 1 && (A.con_u >> (3 + d) & 1)) ||
            (J == 0 && (A.con_u >> (6 + d) & 1)) || (J == A.nny - 1 && (A.con_u >> (9 + d) & 1)))
          cmask |= 1u << d;
      // pressure node column (lanes with i, j < NP)
      const bool pth = valid && i < NP && j < NP;
      const int  Ip = KP * cx + min(i, KP), Jp = KP * cy + min(j, KP);
      const bool own_p  = pth && (i < KP || lastx) && (j < KP || lasty);
      const bool seam_p = (KP * cxl + i == TPX - 1 && Ip < A.npx - 1) || (KP * cyl + j == TPY - 1 && Jp < A.npy - 1);
      const bool pcon   = (Ip == 0 && (A.con_p >> 0 & 1)) || (Ip == A.npx - 1 && (A.con_p >> 1 & 1)) ||
                        (Jp == 0 && (A.con_p >> 2 & 1)) || (Jp == A.npy - 1 && (A.con_p >> 3 & 1));

      // all per-lane predicates in ONE register; the loop re-derives them from an opaque copy
      enum
      {
        F_OWN_U = 1, F_SEAM_U = 2, F_CON0 = 4, F_OWN_P = 32, F_SEAM_P = 64, F_PCON = 128, F_W = 256, F_S = 512,
        F_IK = 1024, F_JK = 2048, F_PTH = 4096, F_IKP = 8192, F_JKP = 16384
      };
      const unsigned flags = (own_u ? F_OWN_U : 0) | (seam_u ? F_SEAM_U : 0) | (cmask * F_CON0) | (own_p ? F_OWN_P : 0) |
                             (seam_p ? F_SEAM_P : 0) | (pcon ? F_PCON : 0) | ((active && i == 0 && hasW) ? F_W : 0) |
                             ((active && j == 0 && hasS) ? F_S : 0) | ((active && i == K) ? F_IK : 0) |
                             ((active && j == K) ? F_JK : 0) | ((active && i < NP && j < NP) ? F_PTH : 0) |
                             (i == KP ? F_IKP : 0) | (j == KP ? F_JKP : 0);

      const double wab = A.det * A.w[i] * A.w[j];

      // addresses = wave-uniform base pointer (scalar registers) + 32-bit per-lane element offset
      const unsigned ubase = (unsigned)((J * A.nnx + I) * 3), pbase = (unsigned)(Jp * A.npx + Ip);
      const size_t   plane_u = (size_t)A.nny * A.nnx * 3, plane_p = (size_t)A.npy * A.npx;
      const unsigned loff = (unsigned)((cy * A.ncx + cx) * (NLIN * N3) + lc); // + uniform layer / comp / plane part

      // wave-private LDS: values [4][N] planes of PLS slots, exchange [6] planes; slot of this lane = l
      double *const VAL = lds + L_WAVE + wave * (VALW + TGW) + cw * (4 * N * PLS);
      double *const TG  = lds + L_WAVE + wave * (VALW + TGW) + VALW + cw * (6 * PLS);
      double *const PUB = lds + L_WAVE + 4 * (VALW + TGW); // [buffer 2][E | N][cell][j or i][4][k]
      const int     xl = j * N, yl = i;                    // first element of my x-line / y-line in a plane

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

      constexpr int NST = LIN_MODE == 0 ? 12 : (LIN_MODE == 1 ? 4 : 0); // state doubles per point
      double        st[NST > 0 ? NST : 1];                               // state of the plane to come
      // nodal z-lines of the layer to come (planes 1..K; plane 0 is the previous layer's top plane)
      double Un[3][N], Pn[N];
      auto   load_nodes = [&](const int czn, const unsigned ubase, const unsigned pbase) {
#pragma unroll
        for (int k = 1; k < N; ++k)
          {
            const double *pl = A.src_u + (size_t)min(K * czn + k, A.nnz - 1) * plane_u;
#pragma unroll
            for (int d = 0; d < 3; ++d)
              Un[d][k] = pl[ubase + d];
          }
        if (WITH_P)
          {
#pragma unroll
            for (int k = 1; k < NP; ++k)
              Pn[k] = (A.src_p + (size_t)min(KP * czn + k, A.npz - 1) * plane_p)[pbase];
          }
      };
      auto load_state = [&](const double *ln, const unsigned loff) {
#pragma unroll
        for (int e = 0; e < NST; ++e)
          st[e] = (ln + e * N3)[loff];
      };
      load_nodes(cz0, ubase, pbase);

      const unsigned a_tab = lds_byte_addr(lds), a_val = lds_byte_addr(VAL), a_tg = lds_byte_addr(TG);
      const unsigned a_x = a_val + 8 * xl, a_y = a_val + 8 * yl, a_own = a_val + 8 * l;

      for (int layer = 0; layer < nl; ++layer)
        {
          const int cz = cz0 + layer;
          // opaque copies: keeps the per-lane rows / predicates from being hoisted out of the layer
          // loop (they would stay live across the quadrature loop)
          int ro_i = i, ro_j = j;
          asm volatile("" : "+v"(ro_i), "+v"(ro_j));
          unsigned fl = flags;
          asm volatile("" : "+v"(fl));
          // (same for the per-lane address parts: everything derived from them is recomputed where it
          // is used instead of being kept in -- and spilled from -- registers across the whole loop)
          unsigned ub_ = ubase, pb_ = pbase, lo_ = loff, ax_ = a_x, ay_ = a_y, ao_ = a_own, atg_ = a_tg, atab_ = a_tab;
          asm volatile("" : "+v"(ub_), "+v"(pb_), "+v"(lo_), "+v"(ax_), "+v"(ay_), "+v"(ao_), "+v"(atg_), "+v"(atab_));
          const unsigned a_ri = atab_ + 8 * N * ro_i, a_rj = atab_ + 8 * N * ro_j;    // row i / j of S (D: + 8 L_TAB_D)
          const unsigned a_ci = atab_ + 8 * ro_i, a_cj = atab_ + 8 * ro_j;            // column i / j
          const unsigned a_pri = atab_ + 8 * (L_TAB_SP + NP * ro_i), a_prj = atab_ + 8 * (L_TAB_SP + NP * ro_j);
          const unsigned a_pci = atab_ + 8 * (L_TAB_SP + min(ro_i, KP)), a_pcj = atab_ + 8 * (L_TAB_SP + min(ro_j, KP));

          // ---- nodal values of my z-line (read_dof_values: constrained entries read as zero) ----
          {
            double U[3][N], P[N];
#pragma unroll
            for (int d = 0; d < 3; ++d)
              {
                U[d][0] = src_top[d];
#pragma unroll
                for (int k = 1; k < N; ++k)
                  U[d][k] = Un[d][k];
              }
            P[0]  = srcp_top;
            P[KP] = 0.;
            if (WITH_P)
              {
#pragma unroll
                for (int k = 1; k < NP; ++k)
                  P[k] = Pn[k];
              }
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

            // ---- Z: nodal z-line -> Gauss points (registers), hand over to the x contraction ----
            {
              ctab_t tz = (ctab_t)A.tab; // (opaque copy: the coefficients are re-loaded per phase, not spilled)
              asm volatile("" : "+s"(tz));
              if (slot_ok)
                {
#pragma unroll
                  for (int d = 0; d < 3; ++d)
                    {
#pragma unroll
                      for (int c = 0; c < N; ++c)
                        {
                          double s = 0.;
#pragma unroll
                          for (int k = 0; k < N; ++k)
                            s += symS(tz, c, k) * U[d][k];
                          VAL[(d * N + c) * PLS + l] = s;
                        }
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
                            s += symSp(tz, c, k) * P[k];
                          VAL[(3 * N + c) * PLS + l] = s;
                        }
                    }
                }
            }
          }
          // the state of the first quadrature plane arrives during the x / y contractions
          const double *lin = A.lin + (size_t)cz * A.ncy * A.ncx * (NLIN * N3);
          load_state(lin, lo_);
          wave_sync();
          // ---- X: lane (a,b) contracts the x-line (., b) of every plane with row a of S ---------
#pragma unroll 1
          for (int d = 0; d < 3; ++d)
            {
              double T[N];
              ho_contract<N, N, 1, PLS, 1>(ax_ + d * (8 * N * PLS), a_ri, T);
              wave_sync();
              if (slot_ok)
                {
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    VAL[(d * N + c) * PLS + l] = T[c];
                }
            }
          if (WITH_P)
            {
              double T[N];
              ho_contract<N, NP, 1, PLS, 1>(ax_ + 3 * (8 * N * PLS), a_pri, T);
              wave_sync();
              if (slot_ok)
                {
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    VAL[(3 * N + c) * PLS + l] = T[c];
                }
            }
          wave_sync();
          // ---- Y: the y-line (a, .) with row b of S: values at my quadrature line ---------------
          // (a component's planes are only read by its own contraction: write back per component)
#pragma unroll 1
          for (int d = 0; d < 3; ++d)
            {
              double T[N];
              ho_contract<N, N, N, PLS, 1>(ay_ + d * (8 * N * PLS), a_rj, T);
              wave_sync();
              if (slot_ok)
                {
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    VAL[(d * N + c) * PLS + l] = T[c];
                }
            }
          if (WITH_P)
            {
              double T[N];
              ho_contract<N, NP, N, PLS, 1>(ay_ + 3 * (8 * N * PLS), a_prj, T);
              wave_sync();
              if (slot_ok)
                {
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    VAL[(3 * N + c) * PLS + l] = T[c];
                }
            }
          wave_sync();

          // ---- quadrature planes: gradients, physics (navier_stokes_matrix.cc:702-893), -------
          // ---- transposed collocation derivatives                                        -------
          double R[3][N];
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int c = 0; c < N; ++c)
              R[d][c] = 0.;
          {
            ctab_t tz = (ctab_t)A.tab;
            asm volatile("" : "+s"(tz));
#pragma unroll 1
            for (int c = 0; c < N; ++c)
              {
                double Dz[N], ec[N]; // row c of D, row c of the identity (wave-uniform)
#pragma unroll
                for (int m = 0; m < N; ++m)
                  {
                    Dz[m] = tz[N * N + c * N + m];
                    ec[m] = c == m ? 1. : 0.;
                  }
                const double jxw = wab * tz[2 * N * N + N * NP + c];
                double       g[3][3], u[3], pq = 0.;
                {
                  // rows a / b of D, then x-line, y-line and own value of plane c, software-pipelined
                  // over the components
                  const unsigned off = (unsigned)c * (PLS * 8);
                  const unsigned xb = ax_ + off, yb = ay_ + off, ob = ao_ + off;
                  // HO_PIPE line-buffer sets: 2 = the reads of component d + 1 are in flight while the
                  // gradient of component d is formed (90 -> 60 VGPRs of line buffers: the Q4/Q3 Newton kernel
                  // spilled 31 registers with three sets, and every scratch reload drains the state prefetch)
                  constexpr int NB = HO_PIPE;
                  double         Dx[N], Dy[N], lx[NB][N], ly[NB][N], lz[NB][N];
                  rd_line<8 * L_TAB_D, 1>(a_ri, Dx);
                  rd_line<8 * L_TAB_D, 1>(a_rj, Dy);
                  auto issue = [&](auto dt) {
                    constexpr int d = decltype(dt)::value, b = d % NB;
                    rd_line<8 * d * N * PLS, 1>(xb, lx[b]);
                    rd_line<8 * d * N * PLS, N>(yb, ly[b]);
                    rd_line<8 * d * N * PLS, PLS>(ao_, lz[b]); // my z-line (all planes)
                    u[d] = ds_rd<8 * d * N * PLS>(ob);
                  };
                  auto grad = [&](auto dt, auto cnt) {
                    constexpr int d = decltype(dt)::value, CNT = decltype(cnt)::value, b = d % NB;
                    ds_wait<CNT>(lx[b]);
                    ds_wait<CNT>(ly[b]);
                    ds_wait<CNT>(lz[b]);
                    ds_wait1<CNT>(u[d]);
                    if (d == 0)
                      {
                        ds_wait<CNT>(Dx);
                        ds_wait<CNT>(Dy);
                      }
                    double sx = 0., sy = 0., sz = 0.;
#pragma unroll
                    for (int m = 0; m < N; ++m)
                      {
                        sx += Dx[m] * lx[b][m];
                        sy += Dy[m] * ly[b][m];
                        sz += Dz[m] * lz[b][m];
                      }
                    g[d][0] = sx * A.ih[0];
                    g[d][1] = sy * A.ih[1];
                    g[d][2] = sz * A.ih[2];
                  };
                  using I0 = std::integral_constant<int, 0>;
                  using I1 = std::integral_constant<int, 1>;
                  using I2 = std::integral_constant<int, 2>;
                  constexpr int PER = 3 * N + 1, NPQ = WITH_P ? 1 : 0;
                  if constexpr (NB >= 3)
                    {
                      issue(I0{});
                      issue(I1{});
                      grad(I0{}, std::integral_constant<int, PER>{});
                      issue(I2{});
                      if (WITH_P)
                        pq = ds_rd<8 * 3 * N * PLS>(ob);
                      grad(I1{}, std::integral_constant<int, PER + NPQ>{});
                      grad(I2{}, std::integral_constant<int, NPQ>{});
                    }
                  else if constexpr (NB == 2)
                    {
                      issue(I0{});
                      issue(I1{});
                      grad(I0{}, std::integral_constant<int, PER>{});
                      issue(I2{}); // (into the buffers of component 0)
                      if (WITH_P)
                        pq = ds_rd<8 * 3 * N * PLS>(ob);
                      grad(I1{}, std::integral_constant<int, PER + NPQ>{});
                      grad(I2{}, std::integral_constant<int, NPQ>{});
                    }
                  else
                    {
                      issue(I0{});
                      grad(I0{}, std::integral_constant<int, 0>{});
                      issue(I1{});
                      grad(I1{}, std::integral_constant<int, 0>{});
                      issue(I2{});
                      if (WITH_P)
                        pq = ds_rd<8 * 3 * N * PLS>(ob);
                      grad(I2{}, std::integral_constant<int, NPQ>{});
                    }
                  if (WITH_P)
                    ds_wait1<0>(pq);
                }
                const double div = g[0][0] + g[1][1] + g[2][2];
                double       conv[3];
#pragma unroll
                for (int d = 0; d < 3; ++d)
                  {
                    double res = 0.;
                    if (LIN_MODE == 0) // Newton :802-816; st = (u_lin[3], grad u_lin[3][3])
                      {
                        res = A.beta * (div * st[d] + (st[3] + st[7] + st[11]) * u[d]);
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                          res += st[e] * g[d][e] + u[e] * st[3 + 3 * d + e];
                      }
                    else if (LIN_MODE == 1) // Picard-type :817-826; st = (u_lin[3], div u_lin)
                      {
                        res = A.beta * st[3] * u[d];
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                          res += st[e] * g[d][e];
                      }
                    conv[d] = (A.cA * u[d] + A.cB * res) * jxw; // :717,:827-835
                  }
                // the state registers are free: fetch the next plane's state; it arrives while this
                // plane's transposed derivatives and the next plane's gradients are computed
                HO_FENCE();
                if (c + 1 < N)
                  load_state(lin + (c + 1) * NL, lo_);
                const double diag = A.tau_gd * div - pq;
                double       tgx[3], tgy[3];
#pragma unroll
                for (int d = 0; d < 3; ++d)
                  {
                    // :859-892 row d of tmu (grad u + grad u^T) + (tau_gd div - p) I, times JxW J^{-1}
                    tgx[d]           = (A.tmu * (g[d][0] + g[0][d]) + (d == 0 ? diag : 0.)) * (jxw * A.ih[0]);
                    tgy[d]           = (A.tmu * (g[d][1] + g[1][d]) + (d == 1 ? diag : 0.)) * (jxw * A.ih[1]);
                    const double tgz = (A.tmu * (g[d][2] + g[2][d]) + (d == 2 ? diag : 0.)) * (jxw * A.ih[2]);
#pragma unroll
                    for (int m = 0; m < N; ++m)
                      R[d][m] += Dz[m] * tgz; // D^T in z: registers
                  }
                wave_sync();
                if (slot_ok)
                  {
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                      {
                        TG[(2 * d) * PLS + l]     = tgx[d];
                        TG[(2 * d + 1) * PLS + l] = tgy[d];
                      }
                    if (WITH_P)
                      VAL[(3 * N + c) * PLS + l] = -div * jxw; // :853-856 (the slot's p value is consumed)
                  }
                wave_sync();
                {
                  const unsigned xb = atg_ + 8 * xl, yb = atg_ + 8 * yl;
                  double         DTx[N], DTy[N], lx[3][N], ly[3][N];
                  rd_line<8 * L_TAB_D, N>(a_ci, DTx); // columns a / b of D
                  rd_line<8 * L_TAB_D, N>(a_cj, DTy);
                  auto issue = [&](auto dt) {
                    constexpr int d = decltype(dt)::value;
                    rd_line<8 * (2 * d) * PLS, 1>(xb, lx[d]);
                    rd_line<8 * (2 * d + 1) * PLS, N>(yb, ly[d]);
                  };
                  auto accumulate = [&](auto dt, auto cnt) {
                    constexpr int d = decltype(dt)::value, CNT = decltype(cnt)::value;
                    ds_wait<CNT>(lx[d]);
                    ds_wait<CNT>(ly[d]);
                    if (d == 0)
                      {
                        ds_wait<CNT>(DTx);
                        ds_wait<CNT>(DTy);
                      }
                    double s = conv[d];
#pragma unroll
                    for (int m = 0; m < N; ++m)
                      s += DTx[m] * lx[d][m] + DTy[m] * ly[d][m];
#pragma unroll
                    for (int m = 0; m < N; ++m)
                      R[d][m] += ec[m] * s; // (R[d][c] += s with a run-time c)
                  };
                  using I0 = std::integral_constant<int, 0>;
                  using I1 = std::integral_constant<int, 1>;
                  using I2 = std::integral_constant<int, 2>;
                  issue(I0{});
                  issue(I1{});
                  accumulate(I0{}, std::integral_constant<int, 2 * N>{});
                  issue(I2{});
                  accumulate(I1{}, std::integral_constant<int, 2 * N>{});
                  accumulate(I2{}, std::integral_constant<int, 0>{});
                }
              }
          }
          // the nodal z-lines of the next layer arrive during the transposed interpolation
          load_nodes(cz + 1, ub_, pb_);

          // ---- transposed interpolation: x and y through LDS, z in registers ---------------------
          double Rn[3][N], Rpn[N];
          {
            wave_sync();
            if (slot_ok)
              {
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                  for (int c = 0; c < N; ++c)
                    VAL[(d * N + c) * PLS + l] = R[d][c];
              }
            wave_sync();
#pragma unroll 1
            for (int d = 0; d < 3; ++d)
              {
                double T[N];
                ho_contract<N, N, 1, PLS, N>(ax_ + d * (8 * N * PLS), a_ci, T);
                wave_sync();
                if (slot_ok)
                  {
#pragma unroll
                    for (int c = 0; c < N; ++c)
                      VAL[(d * N + c) * PLS + l] = T[c];
                  }
              }
            if (WITH_P)
              {
                double T[N];
                ho_contract<N, N, 1, PLS, NP>(ax_ + 3 * (8 * N * PLS), a_pci, T);
                wave_sync();
                if (slot_ok)
                  {
#pragma unroll
                    for (int c = 0; c < N; ++c)
                      VAL[(3 * N + c) * PLS + l] = T[c];
                  }
              }
            wave_sync();
            ctab_t tz = (ctab_t)A.tab;
            asm volatile("" : "+s"(tz));
            // (rolled over the components: the results go back to the lane's own LDS slots and are
            // picked up again for the combine / emit step; everything unrolled kept ~120 more VGPRs live)
#pragma unroll 1
            for (int d = 0; d < 3; ++d)
              {
                double T[N], Z[N];
                ho_contract<N, N, N, PLS, N>(ay_ + d * (8 * N * PLS), a_cj, T);
#pragma unroll
                for (int k = 0; k < N; ++k)
                  {
                    double s = 0.;
#pragma unroll
                    for (int c = 0; c < N; ++c)
                      s += symS(tz, c, k) * T[c];
                    Z[k] = s;
                  }
                wave_sync();
                if (slot_ok)
                  {
#pragma unroll
                    for (int k = 0; k < N; ++k)
                      VAL[(d * N + k) * PLS + l] = Z[k];
                  }
              }
#pragma unroll
            for (int k = 0; k < N; ++k)
              Rpn[k] = 0.;
            if (WITH_P)
              {
                double T[N];
                ho_contract<N, N, N, PLS, NP>(ay_ + 3 * (8 * N * PLS), a_pcj, T);
                if (fl & F_PTH)
                  {
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                      {
                        double s = 0.;
#pragma unroll
                        for (int c = 0; c < N; ++c)
                          s += symSp(tz, c, k) * T[c];
                        Rpn[k] = s;
                      }
                  }
              }
          }

          wave_sync();
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int k = 0; k < N; ++k)
              Rn[d][k] = VAL[(d * N + k) * PLS + l];

          // ---- combine the cells of the tile per owned node (one workgroup barrier per layer) ----
          double *const PUB_E = PUB + (layer & 1) * (2 * PUBSZ), *const PUB_N = PUB_E + PUBSZ;
          {
            if (fl & F_IK)
              {
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                  for (int k = 0; k < N; ++k)
                    PUB_E[((cell * N + j) * 4 + d) * N + k] = Rn[d][k];
              }
            if (fl & F_JK)
              {
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                  for (int k = 0; k < N; ++k)
                    PUB_N[((cell * N + i) * 4 + d) * N + k] = Rn[d][k];
              }
            if (WITH_P && (fl & F_PTH))
              {
                if (fl & F_IKP)
                  {
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                      PUB_E[((cell * N + j) * 4 + 3) * N + k] = Rpn[k];
                  }
                if (fl & F_JKP)
                  {
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                      PUB_N[((cell * N + i) * 4 + 3) * N + k] = Rpn[k];
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
                  Rn[d][k] += PUB_E[(((cell - 1) * N + j) * 4 + d) * N + k];
              if (WITH_P)
                {
#pragma unroll
                  for (int k = 0; k < NP; ++k)
                    Rpn[k] += PUB_E[(((cell - 1) * N + j) * 4 + 3) * N + k];
                }
            }
          if (fl & F_S)
            {
#pragma unroll
              for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int k = 0; k < N; ++k)
                  Rn[d][k] += PUB_N[(((cell - TCX) * N + i) * 4 + d) * N + k];
              if (WITH_P)
                {
#pragma unroll
                  for (int k = 0; k < NP; ++k)
                    Rpn[k] += PUB_N[(((cell - TCX) * N + i) * 4 + 3) * N + k];
                }
            }
          if ((fl & (F_W | F_S)) == (F_W | F_S))
            {
#pragma unroll
              for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int k = 0; k < N; ++k)
                  Rn[d][k] += PUB_E[(((cell - TCX - 1) * N + K) * 4 + d) * N + k];
              if (WITH_P)
                {
#pragma unroll
                  for (int k = 0; k < NP; ++k)
                    Rpn[k] += PUB_E[(((cell - TCX - 1) * N + KP) * 4 + 3) * N + k];
                }
            }
          // ---- emit the finished planes, carry the top plane -----------------------------------
          Rn[0][0] += carry[0];
          Rn[1][0] += carry[1];
          Rn[2][0] += carry[2];
          Rpn[0] += carry_p;
          // every owned entry goes to dst or, on the high rim of the tile, to the slab ...
          if (fl & F_OWN_U)
            {
              const bool     seam   = (fl & F_SEAM_U) != 0u;
              const unsigned r3     = rim_index_ho<TNX, TNY>(K * cxl + i, K * cyl + j) * 3;
              double        *tp     = seam ? A.slab_u + (wgs * (K * A.LZ + 1) + K * layer) * (RIMU * 3) + r3 :
                                             A.dst_u + (size_t)(K * cz) * plane_u + ub_;
              const size_t   stride = seam ? (size_t)(RIMU * 3) : plane_u;
#pragma unroll
              for (int k = 0; k < K; ++k)
#pragma unroll
                for (int d = 0; d < 3; ++d)
                  tp[k * stride + d] = Rn[d][k];
            }
          if (WITH_P && A.integrate_p && (fl & F_OWN_P))
            {
              const bool     seam   = (fl & F_SEAM_P) != 0u;
              const unsigned r      = rim_index_ho<TPX, TPY>(KP * cxl + i, KP * cyl + j);
              double        *tp     = seam ? A.slab_p + (wgs * (KP * A.LZ + 1) + KP * layer) * RIMP + r :
                                             A.dst_p + (size_t)(KP * cz) * plane_p + pb_;
              const size_t   stride = seam ? (size_t)RIMP : plane_p;
#pragma unroll
              for (int k = 0; k < KP; ++k)
                tp[k * stride] = Rpn[k];
            }
          // ... and constrained rows (tiles / layers at the domain boundary only) are then set to
          // +-src (:250-255): a later store of the same lane to the same address, or an entry the
          // fix-up kernel skips
          if (__builtin_amdgcn_readfirstlane(__any((fl & (7 * F_CON0 | F_PCON)) != 0u)) ||
              (cz == 0 && ((A.con_u >> 12 & 7u) || (A.con_p >> 4 & 1u))))
            {
              if (fl & F_OWN_U)
                {
#pragma unroll 1
                  for (int k = 0; k < K; ++k)
                    {
                      const size_t po = (size_t)(K * cz + k) * plane_u + ub_;
#pragma unroll
                      for (int d = 0; d < 3; ++d)
                        if ((fl & (F_CON0 << d)) || (K * cz + k == 0 && (A.con_u >> (12 + d) & 1)))
                          A.dst_u[po + d] = A.src_u[po + d];
                    }
                }
              if (WITH_P && A.integrate_p && (fl & F_OWN_P))
                {
#pragma unroll 1
                  for (int k = 0; k < KP; ++k)
                    {
                      const size_t po = (size_t)(KP * cz + k) * plane_p + pb_;
                      if ((fl & F_PCON) || (KP * cz + k == 0 && (A.con_p >> 4 & 1)))
                        A.dst_p[po] = -A.src_p[po];
                    }
                }
            }
#pragma unroll
          for (int d = 0; d < 3; ++d)
            carry[d] = Rn[d][K];
          carry_p = Rpn[KP];
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
          if (ho_fix_skip(A, I, J, Kz, nn_x, nn_y, nn_z))
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
          if (seam || on_constrained_face(I, J, Kz, nn_x, nn_y, nn_z, con, NC == 1 ? 1 : 3, comp) ||
              ho_fix_skip(A, I, J, Kz, nn_x, nn_y, nn_z))
            continue;
          dst[((long)(Kz * (long)nn_y + J) * nn_x + I) * NC + comp] += zslab[(bt * A.n_chunks + m - 1) * (TNX * TNY * NC) + e];
        }
    }

    template <int K>
    __global__ __launch_bounds__(64) void ns_ho_fixup_kernel(const HOArgs A, const long n1, const long n2,
                                                             const long n3, const long n4)
    {
      using C = HOTile<K>;
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

    // phase -1: the whole operator; phases 0 / 1 / 2 as in launch_ns_vmult_q2 (ns_q2.hip): 0 = first half of the
    // workgroups that touch no node of the inter-GPU interface faces `iface`, 1 = the workgroups that do + fix-up
    // of the interface nodes, 2 = the other interior workgroups + the rest of the fix-up
    template <int K>
    int launch_ho(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u,
                  const double *src_p, const int phase, const uint32_t iface)
    {
      using C = HOTile<K>;
      constexpr int N = K + 1, NP = K;
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
        int        lz    = 8; // (64^3 Q4: 4 / 8 / 16 / 32 layers per chunk -> 1.534 / 1.533 / 1.584 / 1.578 ms)
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
      if (!ctx->ho_tab.p)
        {
          std::vector<double> tab;
          for (int q = 0; q < N * N; ++q)
            tab.push_back(A.S[(q / N) * N + q % N]);
          for (int q = 0; q < N * N; ++q)
            tab.push_back(A.D[q]);
          for (int q = 0; q < N * NP; ++q)
            tab.push_back(A.Sp[q]);
          for (int q = 0; q < N; ++q)
            tab.push_back(A.w[q]);
          if (int e = ensure(ctx->ho_tab, tab.size()))
            return e;
          if (copy_to_device_now(ctx->ho_tab.p, tab.data(), tab.size() * sizeof(double)) != hipSuccess)
            return ADAFLO_EHIP;
        }
      A.tab = ctx->ho_tab.p;
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
      if (with_p && !A.integrate_p && (phase <= 0 || phase == 5)) // (5: the set-up phase of the two-stream schedule runs on the engine stream BEFORE the auxiliary stream may pack or unpack-add dst_p; in phase 3 it raced with them -- ADVICE r05)
        if (int e = launch_prepare_dst(ctx, dst_p, src_p, ctx->n_nodes_p, 1, A.npx, A.npy, A.npz, A.con_p, -1., true))
          return e;
      long nwg = (long)n_wg;
      if (phase >= 0)
        {
          // workgroup list [interface | interior A | interior B], cached per (grid, iface)
          const long key[4] = {A.tiles_x, A.tiles_y, A.n_chunks, (long)iface};
          if (!ctx->q2_wg_list || std::memcmp(key, ctx->q2_wg_key, sizeof(key)) != 0)
            {
              std::vector<int> bnd, inner;
              for (int by = 0; by < A.tiles_y; ++by)
                for (int bx = 0; bx < A.tiles_x; ++bx)
                  for (int bz = 0; bz < A.n_chunks; ++bz)
                    {
                      const bool b = (bx == 0 && (iface & 1u)) || (bx == A.tiles_x - 1 && (iface & 2u)) ||
                                     (by == 0 && (iface & 4u)) || (by == A.tiles_y - 1 && (iface & 8u)) ||
                                     (bz == 0 && (iface & 16u)) || (bz == A.n_chunks - 1 && (iface & 32u));
                      (b ? bnd : inner).push_back((by * A.tiles_x + bx) * A.n_chunks + bz);
                    }
              ctx->q2_wg_counts[0] = (int)bnd.size();
              ctx->q2_wg_counts[1] = (int)(inner.size() / 2);
              ctx->q2_wg_counts[2] = (int)(inner.size() - inner.size() / 2);
              bnd.insert(bnd.end(), inner.begin(), inner.end());
              if (ctx->q2_wg_list)
                (void)hipFree(ctx->q2_wg_list);
              ctx->q2_wg_list = nullptr;
              if (hipMalloc(&ctx->q2_wg_list, sizeof(int) * (bnd.size() + 1)) != hipSuccess)
                return ADAFLO_ENOMEM;
              if (copy_to_device_now(ctx->q2_wg_list, bnd.data(), sizeof(int) * bnd.size()) != hipSuccess)
                return ADAFLO_EHIP;
              std::memcpy(ctx->q2_wg_key, key, sizeof(key));
            }
          const int nb = ctx->q2_wg_counts[0], na = ctx->q2_wg_counts[1], nc = ctx->q2_wg_counts[2];
          A.wg_list   = ctx->q2_wg_list;
          A.wg_offset = phase == 1 ? 0 : ((phase == 0 || phase == 3) ? nb : nb + na);
          A.wg_count  = phase == 1 ? nb : (phase == 0 ? na : (phase == 3 ? na + nc : (phase >= 4 ? 0 : nc)));
          A.fix_mode  = phase == 4 ? 2 : phase; // 1: interface nodes, 2: the others (phases 0 and 3 run no fix-up; 3 = 0 + 2
                                               // without it, 4 = the fix-up of phase 2 alone, 5 = set-up only: the two-stream schedule of comm.hip)
          A.iface     = iface;
          nwg         = A.wg_count;
        }
      const size_t lds_bytes = sizeof(double) * (size_t)ho_lds_doubles<K>();
      const dim3   grid((unsigned)(nwg > 0 ? nwg : 1)), block(NTH);
      hipError_t   err  = hipSuccess;
      hipEvent_t   stop = (ctx->timing && nwg > 0) ? ctx->kernel_timer.start(ctx->stream) : nullptr;
#define HO_LAUNCH(LM, WP)                                                                                 \
  {                                                                                                       \
    static bool attr_set = false;                                                                         \
    if (!attr_set)                                                                                        \
      {                                                                                                   \
        err      = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_ho_kernel<K, LM, WP>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);      \
        attr_set = err == hipSuccess;                                                                     \
      }                                                                                                   \
    if (err == hipSuccess && nwg > 0)                                                                     \
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
      if (phase == -1 || phase == 1)
        ctx->kernel_timer.count++;
      if (phase == 0 || phase == 3 || phase == 5)
        return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
      const long tiles = (long)A.tiles_x * A.tiles_y;
      const bool fix_p = with_p && A.integrate_p;
      const long n1 = tiles * A.nnz, n2 = tiles * (A.n_chunks - 1);
      const long n3 = fix_p ? tiles * A.npz : 0, n4 = fix_p ? tiles * (A.n_chunks - 1) : 0;
      long       nb = n1 + n2 + n3 + n4;
      if (nb > 256 * 512)
        nb = 256 * 512;
      if (!(phase == 1 && iface == 0u)) // (no interface: phase 1 has nothing to fix up)
        hipLaunchKernelGGL((ns_ho_fixup_kernel<K>), dim3((unsigned)nb), dim3(64), 0, ctx->stream, A, n1, n2, n3, n4);
      return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
    }
  } // namespace

  bool ho_supported(const adaflo_ctx *ctx)
  {
    return ctx->k >= 3 && ctx->k <= 5 && !ctx->flat && !ctx->rho.p && !ctx->mu.p && !ctx->damp.p;
  }

  int launch_ns_vmult_ho(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u,
                         const double *src_p, const int phase, const uint32_t iface)
  {
    switch (ctx->k)
      {
        case 3:
          return launch_ho<3>(ctx, op, dst_u, dst_p, src_u, src_p, phase, iface);
        case 4:
          return launch_ho<4>(ctx, op, dst_u, dst_p, src_u, src_p, phase, iface);
        case 5:
          return launch_ho<5>(ctx, op, dst_u, dst_p, src_u, src_p, phase, iface);
        default:
          return ADAFLO_EUNSUPPORTED;
      }
  }
} // namespace adaflo_hip
#else
namespace adaflo_hip
{
  bool ho_supported(const adaflo_ctx *) { return false; }
  int  launch_ns_vmult_ho(adaflo_ctx *, const int, double *, double *, const double *, const double *, const int, const uint32_t)
  {
    return ADAFLO_EUNSUPPORTED;
  }
} // namespace adaflo_hip
#endif
