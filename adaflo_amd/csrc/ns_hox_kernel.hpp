// ns_hox_kernel.hpp -- device source of the x-marching Taylor-Hood Q_k/Q_{k-1} kernel (k = 3, 4, 5):
// NavierStokesMatrix::vmult / velocity_vmult with constant coefficients
// (source/navier_stokes_matrix.cc:601-916, vmult and vmult_velocity branches of local_operation).
//
// Round-4 decomposition.  The round-2 kernel (ns_ho.hip) keeps a z-line per lane and
// lets lane (a,b) read the x- / y-line it needs for EVERY output it produces: (k+1) LDS reads per output,
// ~1000 ds_read_b64 per lane and cell layer -- the LDS pipe and the ~60 dependent exchanges bound it at a
// third of the HBM roofline.  Here a lane owns a whole LINE in the direction that is being contracted:
//   * x-layout: lane (j,k) holds the (k+1) values of its x-line in registers and applies the 1D matrix to
//     them with wave-uniform coefficients (scalar registers): (k+1)^2 FMAs for (k+1) LDS reads;
//   * between two directions the data are TRANSPOSED through a wave-private LDS buffer: every lane writes
//     its line and reads the line of the next direction -- (k+1) writes + (k+1) reads per lane and direction.
//     One wave's LDS operations execute in order, so the transpositions need no s_barrier;
//   * per component: x interpolation in registers, y, z (values at the Gauss points), collocation
//     derivative in z, then y, then x; the x-layout lane ends up with value and gradient of its (k+1)
//     quadrature points; the integration is the transposed chain.  55 + 60 LDS operations per lane and
//     component instead of ~300;
//   * the workgroup (4 waves) owns a CY x CZ cross-section of cells and MARCHES ALONG X: the x-face shared
//     by consecutive cells is carried in registers, every lane reads a contiguous run of (k+1) nodes x 3
//     components of src and writes k nodes x 3 components of dst (96 B for k = 4).  Nodes shared between the
//     cells of the cross-section are combined through an LDS publish area (one workgroup barrier per step),
//     nodes shared between workgroups go through slabs + a fix-up kernel: all y/z seams are ROWS CONTIGUOUS IN
//     X (the round-2 tiles had x-seams: 24 useful bytes per 128-B line), no atomics, bitwise reproducible;
//   * the linearisation state is streamed in a layout of its own, [cell group][cx][i][piece][cell][line][2]:
//     one 16-byte load per lane, piece and quadrature point, consecutive lanes consecutive addresses.
//
// This header contains only device code written against the primitives of hox_intrin.hpp; it includes nothing
// itself so that tests/emu can run the same source on the host lane emulator.
#pragma once

namespace adaflo_hip
{
  namespace hox
  {
    // Diagnostic builds (scripts/exp_ho.sh, results are WRONG, timing only): HOX_EXP & 1 = no LDS traffic (the
    // transpositions become register copies), & 2 = no 1D contractions, & 4 = no state stream, & 8 = state from L2,
    // & 16 = no stores, & 32 = no loads of the nodal lines, & 64 = no publish / barrier / collect, & 128 = the 96-byte
    // velocity chunks a lane stores per step start on a 32-byte boundary (no partially written 32-byte granules)
#ifndef HOX_EXP
#define HOX_EXP 0
#endif
    // HOX_STAMP: s_memtime stamps at the phase boundaries of a step, summed per wave into HXArgs::stamps
    // ([workgroup * 4 + wave][10] cycles: top-of-step wait, evaluate, pressure evaluate, quadrature loop, integrate,
    // pressure integrate, publish, barrier + collect, emit, steps) -- the launcher prints the medians (development aid)
#ifndef HOX_STAMP
#define HOX_STAMP 0
#endif
    // HOX_FLAGS: the waves of a workgroup hand their published face sums to the neighbours through flags in LDS (a
    // wave waits for its lower neighbours in y / z only) instead of meeting at a workgroup barrier every step
#ifndef HOX_FLAGS
#define HOX_FLAGS 0
#endif
    // HOX_FUSED (round 5, default): evaluate and integrate carry value AND derivative arrays through the transposes
    // (x: S u, (D S) u -> y: 3 arrays -> z: value + gradient at the quadrature points of the lane's z-LINE), two exchange
    // stages per component and phase instead of four to six; 9 instead of 6 one-dimensional contractions, the same LDS
    // volume.  The quadrature loop then runs on z-lines.  0: the round-4 form (x-lines, derivative by derivative)
#ifndef HOX_FUSED
#define HOX_FUSED 1
#endif
    // the residual mode on the fused chains as well (1, default).  With the values of u AND of the old-velocity combination
    // live they need more registers: k = 4, 5 run the residual at one workgroup per CU (512 registers, no scratch;
    // measured on one box, residual call in ms, round-4 chains / fused at two workgroups per CU / fused at one: Q3/Q2 64^3
    // 1.13 / 1.12 / 1.22, Q4/Q3 64^3 2.33 / 3.08 / 2.62, Q5/Q4 48^3 2.85 / 3.87 / 2.62).  0: the round-4 chains (x-line
    // quadrature loop) storing the state pieces at the z-line positions of the vmult: scattered 16-byte stores, Q4/Q3 2.73
#ifndef HOX_RES_FUSED
#define HOX_RES_FUSED 1
#endif
    constexpr int NTH  = 256;
    constexpr int NMAX = 6;
    constexpr int NLIN_ = 12;

    // cells per wave (CWY x CWZ) and waves of the workgroup (WY x WZ) over the y-z cross-section
    template <int K>
    struct Cfg;
    template <>
    struct Cfg<3>
    {
      static constexpr int CWY = 2, CWZ = 2, WY = 2, WZ = 2;
    };
    template <>
    struct Cfg<4>
    {
      static constexpr int CWY = 2, CWZ = 1, WY = 1, WZ = 4;
    };
    template <>
    struct Cfg<5>
    {
      static constexpr int CWY = 1, CWZ = 1, WY = 2, WZ = 2;
    };

    template <int K>
    struct Geo
    {
      using C                  = Cfg<K>;
      static constexpr int N   = K + 1, NP = K, KP = K - 1, NL = N * N, N3 = N * N * N;
      static constexpr int CWY = C::CWY, CWZ = C::CWZ, WY = C::WY, WZ = C::WZ;
      static constexpr int CPW = CWY * CWZ, PL = 64 / CPW; // cells per wave, lanes per cell slot
      static constexpr int CY = CWY * WY, CZ = CWZ * WZ, NCELL = CY * CZ;
      static constexpr int TNY = K * CY + 1, TNZ = K * CZ + 1, TPY = KP * CY + 1, TPZ = KP * CZ + 1;
      static constexpr int RIMU = TNY + TNZ - 1, RIMP = TPY + TPZ - 1;
      static constexpr int BUF  = CPW * N3;        // one transposition buffer of a wave (doubles)
      static constexpr int WAVE = 3 * BUF;         // three buffers per wave
      // publish area of one direction: velocity [cell][line N][3][K] then pressure [cell][line NP][KP] (doubles)
      static constexpr int PUBV = NCELL * N * 3 * K, PUBD = PUBV + NCELL * NP * KP;
      // y neighbours inside one wave (WY == 1) exchange without the workgroup barrier: one buffer; z: two
      static constexpr int PUBY_BUFS = WY == 1 ? 1 : 2, PUB_DOUBLES = (PUBY_BUFS + 2) * PUBD;
      // state ring of a wave (LDS-DMA, k = 4 only): two quadrature points = 2 * 6 slots; a slot holds what ONE
      // global_load_lds_dwordx4 writes, 16 bytes per lane at M0 + 16 * lane, up to the last active lane
#ifndef HOX_RING5
#define HOX_RING5 1
#endif
      static constexpr bool RING = K == 4 || (K == 5 && HOX_RING5); // (k = 3: 94 KB with the ring, one workgroup per CU)
      static constexpr int  SLOT = 16 * ((CPW - 1) * PL + NL);          // bytes
      static constexpr int  RING_BYTES = RING ? 2 * (NLIN_ / 2) * SLOT : 0; // per wave
      static constexpr int  FLAG_BYTES = 64; // pub[4], done[4] (HOX_FLAGS)
      static constexpr int  LDS_BYTES = 8 * (4 * WAVE + PUB_DOUBLES) + 4 * RING_BYTES + FLAG_BYTES;
      // HOX_DEEP: a ring of ALL N points of a cell (the state of a step is issued one whole step ahead), one workgroup
      // per CU
      static constexpr int  RING_BYTES_DEEP = RING ? N * (NLIN_ / 2) * SLOT : 0;
      static constexpr int  LDS_BYTES_DEEP  = 8 * (4 * WAVE + PUB_DOUBLES) + 4 * RING_BYTES_DEEP + FLAG_BYTES;
      static_assert(NL <= PL, "cell does not fit its lane slot");
      static_assert(WY * WZ == 4, "four waves per workgroup");
    };

    // number of doubles of state per quadrature point the kernel reads
    constexpr int nst_of(const int lin_mode)
    {
      return lin_mode == 0 ? 12 : (lin_mode == 1 ? 4 : 0);
    }

    struct HXArgs
    {
      int    ncx, ncy, ncz, nnx, nny, nnz, npx, npy, npz, tiles_y, tiles_z, LX, n_chunks, ngy, ngz;
      int    integrate_p;
      uint32_t con_u, con_p;
      const double *src_u, *src_p, *lin; // lin: streaming layout of this kernel (hox_state_offset)
      const double *lin_u;               // recompute-state mode (template RCP): the nodal linearisation point; residual of
                                         // the extrapolating schemes (EXT): extrap_old u_old + extrap_old_old u_old_old
      double       *dst_u, *dst_p;
      // residual mode (template RES): nodal combination weight_old u_old + weight_old_old u_old_old (or null), its
      // factor (the density), and the streaming state the kernel WRITES for the vmults of this Newton step
      const double *old_u;
      double        c_old;
      double       *lin_out;
      double       *lin_sink; // (residual modes: one cell's worth of state behind lin_out, where cells beyond the mesh "store")
      double       *slab_u, *xslab_u, *slab_p, *xslab_p;
      const double *tab; // Tab<K>: the 1D matrices in even / odd form and the constants of the quadrature-point operation
      // phased execution for the multi-GPU overlap (as in ns_q2.hip / ns_ho.hip)
      const int *wg_list;
      int        wg_offset, wg_count, fix_mode;
      uint32_t   iface;
      unsigned long long *stamps; // diagnostic builds only (HOX_STAMP)
    };

    // mesh-dependent integers of the launch (host)
    template <int K>
    inline void hox_geometry(HXArgs &A, const int ncell[3], const int lx)
    {
      using G = Geo<K>;
      A.ncx   = ncell[0];
      A.ncy   = ncell[1];
      A.ncz   = ncell[2];
      A.nnx   = K * A.ncx + 1;
      A.nny   = K * A.ncy + 1;
      A.nnz   = K * A.ncz + 1;
      A.npx   = (K - 1) * A.ncx + 1;
      A.npy   = (K - 1) * A.ncy + 1;
      A.npz   = (K - 1) * A.ncz + 1;
      A.tiles_y  = (A.ncy + G::CY - 1) / G::CY;
      A.tiles_z  = (A.ncz + G::CZ - 1) / G::CZ;
      A.LX       = lx < 1 ? 1 : (lx > A.ncx ? A.ncx : lx);
      A.n_chunks = (A.ncx + A.LX - 1) / A.LX;
      A.ngy      = (A.ncy + G::CWY - 1) / G::CWY;
      A.ngz      = (A.ncz + G::CWZ - 1) / G::CWZ;
    }

    // workgroup list [interface | interior A | interior B] of the phased schedule (host): a workgroup is "interface"
    // if its chunk / cross-section touches one of the brick faces `iface` (bit 2 d + side)
    inline void hox_wg_lists(const HXArgs &A, const uint32_t iface, std::vector<int> &list, int counts[3])
    {
      std::vector<int> bnd, inner;
      for (int bz = 0; bz < A.tiles_z; ++bz)
        for (int by = 0; by < A.tiles_y; ++by)
          for (int bx = 0; bx < A.n_chunks; ++bx)
            {
              const bool b = (bx == 0 && (iface & 1u)) || (bx == A.n_chunks - 1 && (iface & 2u)) || (by == 0 && (iface & 4u)) ||
                             (by == A.tiles_y - 1 && (iface & 8u)) || (bz == 0 && (iface & 16u)) ||
                             (bz == A.tiles_z - 1 && (iface & 32u));
              (b ? bnd : inner).push_back((bz * A.tiles_y + by) * A.n_chunks + bx);
            }
      counts[0] = (int)bnd.size();
      counts[1] = (int)(inner.size() / 2);
      counts[2] = (int)(inner.size() - inner.size() / 2);
      list      = bnd;
      list.insert(list.end(), inner.begin(), inner.end());
    }
    // rim line of the cross-section node grid (TY x TZ nodes): the high rim in y, then the high rim in z
    template <int TY, int TZ>
    __device__ __forceinline__ int rim_line(const int jl, const int kl)
    {
      return jl == TY - 1 ? kl : TZ + jl;
    }

    // position (in doubles) of the 16-byte piece `piece` of quadrature point (i, line l) of cell (cx, cy, cz) in
    // the streaming state: [group gz][group gy][cx][i][piece][cell of the group][line][2]
    template <int K>
    __device__ __forceinline__ size_t hox_state_offset(const int ncx, const int ngy, const int npc, const int cx,
                                                       const int cy, const int cz, const int i, const int piece,
                                                       const int l)
    {
      using G           = Geo<K>;
      const int    gy = cy / G::CWY, gz = cz / G::CWZ, scw = (cz % G::CWZ) * G::CWY + cy % G::CWY;
      const size_t grp = (size_t)gz * ngy + gy;
      return (((((grp * ncx + cx) * G::N + i) * npc + piece) * G::CPW + scw) * G::NL + l) * 2;
    }

    // ---- 1D matrices in even / odd form -------------------------------------------------------------------------
    // Gauss and Gauss-Lobatto points are symmetric about 1/2, so every 1D matrix here satisfies
    // M[q][i] = sigma M[NQ-1-q][NI-1-i] (sigma = +1: interpolation, -1: derivative).  With e_i = in[i] + in[NI-1-i],
    // o_i = in[i] - in[NI-1-i] the products are  out[q] = A + B,  out[NQ-1-q] = sigma (A - B),
    //   A = sum_{i < NI/2} E[q][i] e_i + C[q] in[mid],   B = sum_{i < NI/2} O[q][i] o_i,
    //   E = (M[q][i] + M[q][NI-1-i]) / 2,  C = M[q][mid],  O = (M[q][i] - M[q][NI-1-i]) / 2
    // -- what deal.II's FEEvaluation does ("even-odd decomposition"): 13 instead of 25 coefficients for 5 x 5, i.e.
    // 26 instead of 50 scalar registers per matrix, and 21 instead of 25 vector operations per line.
    // Compact table of one matrix: [E: RQ x HI | C: RQ (NI odd) | O: RQ x HI], RQ = ceil(NQ / 2), HI = NI / 2.
    constexpr int eo_size(const int nq, const int ni)
    {
      return ((nq + 1) / 2) * (2 * (ni / 2) + (ni & 1));
    }
    // (host) M is [NQ][NI] row-major; transpose = true stores the table of M^T
    inline void eo_table(std::vector<double> &out, const double *M, const int nq, const int ni, const bool transpose)
    {
      const int NQ = transpose ? ni : nq, NI = transpose ? nq : ni;
      auto      m  = [&](const int q, const int i) { return transpose ? M[i * ni + q] : M[q * ni + i]; };
      const int RQ = (NQ + 1) / 2, HI = NI / 2;
      for (int q = 0; q < RQ; ++q)
        for (int i = 0; i < HI; ++i)
          out.push_back(0.5 * (m(q, i) + m(q, NI - 1 - i)));
      if (NI & 1)
        for (int q = 0; q < RQ; ++q)
          out.push_back(m(q, HI));
      for (int q = 0; q < RQ; ++q)
        for (int i = 0; i < HI; ++i)
          out.push_back(0.5 * (m(q, i) - m(q, NI - 1 - i)));
    }
    // table of a launch: [S | S^T | D | D^T | Sp | Sp^T | w[N] | 1/h[3] | det | cA | cB | beta | tau_gd | tmu |
    // variable coefficients (rho, mu, damping per point): gamma, tau1 (factor of rho), 1 (factor of damping; all three 0
    // for Stokes), tau1 (factor of mu)]
    template <int K>
    struct Tab
    {
      static constexpr int N = K + 1, NP = K;
      static constexpr int S = 0, ST = S + eo_size(N, N), D = ST + eo_size(N, N), DT = D + eo_size(N, N),
                           SP = DT + eo_size(N, N), SPT = SP + eo_size(N, NP), DS = SPT + eo_size(NP, N),
                           DST = DS + eo_size(N, N), C = DST + eo_size(N, N); // DS = D S: nodes -> d/dx at the Gauss points
      static constexpr int C_W = 0, C_IH = N, C_DET = N + 3, C_CA = N + 4, C_CB = N + 5, C_BETA = N + 6, C_TGD = N + 7,
                           C_TMU = N + 8, C_GAMMA = N + 9, C_T1RHO = N + 10, C_DAMPF = N + 11, C_TAU1 = N + 12,
                           SIZE = C + N + 13;
    };
    // (host) S[q][i] nodal -> Gauss points (N x N), Dc collocation derivative (N x N), Sp pressure (N x NP)
    template <int K>
    inline std::vector<double> hox_table(const double *S, const double *Dc, const double *Sp, const double *w,
                                         const double h[3], const double cA, const double cB, const double beta,
                                         const double tau_gd, const double tmu, const double gamma = 0.,
                                         const double t1rho = 0., const double dampf = 0., const double tau1 = 0.)
    {
      constexpr int       N = K + 1, NP = K;
      std::vector<double> t;
      eo_table(t, S, N, N, false);
      eo_table(t, S, N, N, true);
      eo_table(t, Dc, N, N, false);
      eo_table(t, Dc, N, N, true);
      eo_table(t, Sp, N, NP, false);
      eo_table(t, Sp, N, NP, true);
      {
        std::vector<double> DS(N * N, 0.);
        for (int q = 0; q < N; ++q)
          for (int i = 0; i < N; ++i)
            for (int r = 0; r < N; ++r)
              DS[q * N + i] += Dc[q * N + r] * S[r * N + i];
        eo_table(t, DS.data(), N, N, false);
        eo_table(t, DS.data(), N, N, true);
      }
      for (int q = 0; q < N; ++q)
        t.push_back(w[q]);
      for (int e = 0; e < 3; ++e)
        t.push_back(1. / h[e]);
      t.push_back(h[0] * h[1] * h[2]);
      t.push_back(cA);
      t.push_back(cB);
      t.push_back(beta);
      t.push_back(tau_gd);
      t.push_back(tmu);
      t.push_back(gamma);
      t.push_back(t1rho);
      t.push_back(dampf);
      t.push_back(tau1);
      return t;
    }

    // A 1D matrix in even / odd form, its coefficients in scalar registers.  load() goes BEFORE the LDS reads of the
    // phase that uses the matrix, so that the scalar loads travel while the line is read; apply<ADD>():
    // out[q] (+)= sum_i M[q][i] in[i]
    template <int NQ, int NI, int SIGMA>
    struct EoMat
    {
      static constexpr int  RQ = (NQ + 1) / 2, HQ = NQ / 2, HI = NI / 2;
      static constexpr bool MID = (NI & 1) != 0, QMID = (NQ & 1) != 0;
      static constexpr int  OE = 0, OC = RQ * HI, OO = OC + (MID ? RQ : 0), SIZE = OO + RQ * HI;
      double                c[SIZE];

      __device__ __forceinline__ void load(const ctab_t T)
      {
#pragma unroll
        for (int i = 0; i < SIZE; ++i)
          c[i] = T[i];
      }
      template <bool ADD>
      __device__ __forceinline__ void apply(const double (&in)[NI], double (&out)[NQ]) const
      {
        if constexpr ((HOX_EXP & 2) != 0)
          {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              out[q] = ADD ? out[q] + in[q % NI] : in[q % NI] * c[0];
            return;
          }
        double e[HI], o[HI];
#pragma unroll
        for (int i = 0; i < HI; ++i)
          {
            e[i] = in[i] + in[NI - 1 - i];
            o[i] = in[i] - in[NI - 1 - i];
          }
        const double mid = in[HI]; // (used only if NI is odd)
#pragma unroll
        for (int q = 0; q < HQ; ++q)
          {
            double a = c[OE + q * HI] * e[0], b = c[OO + q * HI] * o[0];
#pragma unroll
            for (int i = 1; i < HI; ++i)
              {
                a += c[OE + q * HI + i] * e[i];
                b += c[OO + q * HI + i] * o[i];
              }
            if (MID)
              a += c[OC + q] * mid;
            const double lo = a + b, hi = SIGMA > 0 ? a - b : b - a;
            out[q]          = ADD ? out[q] + lo : lo;
            out[NQ - 1 - q] = ADD ? out[NQ - 1 - q] + hi : hi;
          }
        if (QMID)
          {
            double v;
            if (SIGMA > 0)
              {
                v = c[OE + HQ * HI] * e[0];
#pragma unroll
                for (int i = 1; i < HI; ++i)
                  v += c[OE + HQ * HI + i] * e[i];
                if (MID)
                  v += c[OC + HQ] * mid;
              }
            else
              {
                v = c[OO + HQ * HI] * o[0];
#pragma unroll
                for (int i = 1; i < HI; ++i)
                  v += c[OO + HQ * HI + i] * o[i];
              }
            out[HQ] = ADD ? out[HQ] + v : v;
          }
      }
    };

    // x[m] = lds[addr + 8 (BOFF + m STRIDE)]
    template <int BOFF, int STRIDE, int NM, int M = 0>
    __device__ __forceinline__ void rd_line(const unsigned addr, double (&x)[NM])
    {
      if constexpr (M < NM)
        {
          if constexpr ((HOX_EXP & 1) != 0)
            x[M] = 1. + M;
          else
            x[M] = ds_rd<8 * (BOFF + M * STRIDE)>(addr);
          rd_line<BOFF, STRIDE, NM, M + 1>(addr, x);
        }
    }
    template <int BOFF, int STRIDE, int NM>
    __device__ __forceinline__ void wr_line(double *const p, const double (&x)[NM])
    {
#pragma unroll
      for (int m = 0; m < NM; ++m)
        {
          if constexpr ((HOX_EXP & 1) != 0)
            sink(x[m]);
          else
            p[BOFF + m * STRIDE] = x[m];
        }
    }

    // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    template <int N, class F, int I = 0>
    __device__ __forceinline__ void static_for(F &&f)
    {
      if constexpr (I < N)
        {
          f(std::integral_constant<int, I>{});
          static_for<N, F, I + 1>(static_cast<F &&>(f));
        }
    }
    // st[2 e], st[2 e + 1] = the 16 bytes of piece Q0 + e of my lane in the state ring: all reads in flight together
    template <int Q0, int NPC, int RS, int SLOT, int E = 0>
    __device__ __forceinline__ void ring_read_issue(const unsigned ring_lane, hox_double2 (&v)[NPC])
    {
      if constexpr (E < NPC)
        {
          ds_rd128<((Q0 + E) % RS) * SLOT>(ring_lane, v[E]);
          ring_read_issue<Q0, NPC, RS, SLOT, E + 1>(ring_lane, v);
        }
    }
    template <int Q0, int NPC, int RS, int SLOT, int NSTA>
    __device__ __forceinline__ void ring_read_impl(const unsigned ring_lane, double (&st)[NSTA])
    {
      hox_double2 v[NPC];
      ring_read_issue<Q0, NPC, RS, SLOT>(ring_lane, v);
#pragma unroll
      for (int e = 0; e + 1 < NPC; e += 2)
        lds_arrived(v[e], v[e + 1]);
      if (NPC & 1)
        lds_arrived(v[NPC - 1], v[NPC - 1]);
#pragma unroll
      for (int e = 0; e < NPC; ++e)
        {
          st[2 * e]     = v[e].x;
          st[2 * e + 1] = v[e].y;
        }
    }

#ifndef HOX_LB
#define HOX_LB 2
#endif
#ifndef HOX_NODES_LATE
#define HOX_NODES_LATE 0
#endif
    // HOX_DEEP = 1 (experiment, round 5): k = 4 with a streamed state at ONE workgroup per CU (512 registers, 147 KB of
    // LDS) and a state ring that holds a whole cell: every point's pieces are issued one step (~15 k cycles) before they
    // are read, where the two-point ring issues three of five points ~500 cycles ahead of an HBM round trip of ~2 700
#ifndef HOX_DEEP
#define HOX_DEEP 0
#endif
    template <int K, int LIN_MODE, bool RES, bool VARCO, bool RCP>
    constexpr bool hox_deep()
    {
      return HOX_DEEP && Geo<K>::RING && !RES && !VARCO && !RCP && LIN_MODE != 2 && !(HOX_EXP & 4);
    }

    // RES: residual mode (source/navier_stokes_matrix.cc:266-293, 663-686, 725-800 for the schemes without
    // extrapolated old velocities): plain reads of the current solution (boundary values included), the nonlinear
    // quadrature-point operation, the values (u, grad u) or (u, div u) STORED as the streaming state of the next vmults
    // instead of read; constrained rows of the sums get 0 (the scatter skips them); the driver forms
    // rhs = user_rhs - rhs - sums.  LIN_MODE 0 / 1 / 2 = which state is written (Newton / Picard-type / none).
    // VARCO: variable density / viscosity / damping at the quadrature points (two-phase flow, :636-642, :827-845): two
    // more pieces per point in the state stream, (rho, mu) and (damping, -); register prefetch also for k = 4 (the
    // ring would not fit the LDS of two workgroups per CU)
    // RCP: recompute-state mode (round 5, as ns_q2.hip): the linearisation state (u_lin, grad u_lin) at the quadrature
    // points is the interpolation of the nodal field the last residual was evaluated at -- three more evaluate chains per
    // cell on node lines that are read like the source vector (72 B per node) instead of 96 B of state per quadrature point
    // EXT (with RES): the residual of the schemes that linearise about the extrapolated old velocity
    // (source/navier_stokes_matrix.cc:644-647, 740-782: semi-implicit, LIN_MODE 1, stores (u_ext, div u_ext) as the state;
    // explicit, LIN_MODE 2, stores nothing): u_ext = extrap_old u_old + extrap_old_old u_old_old is linear in the nodal
    // values, so the launcher combines the nodes once and the kernel evaluates value and gradient of ONE more field -- the
    // chains of the recompute-state mode; one workgroup per CU (512 registers)
    template <int K, int LIN_MODE, bool WITH_P, bool RES = false, bool VARCO = false, bool RCP = false, bool EXT = false>
#ifndef HOX_RES_LB
#define HOX_RES_LB (K == 3 ? HOX_LB : 1)
#endif
#ifndef HOX_RCP_LB
#define HOX_RCP_LB HOX_LB
#endif
    // k = 5 at one workgroup per CU (256 VGPRs + 146 AGPRs + 568 scalar-register spills into VGPR lanes).  History: the build
    // of commit b79a1e7 (round 5) stored wrong, run-to-run DIFFERENT pressure rows (and sometimes died of a memory fault) on
    // meshes with a partial z-tile; after an address rewrite (b0746b5) it was exact, round 6 shipped it -- and an unrelated
    // edit of this header brought the fault back.  Root cause (round 6, profiles/r06_k5_ext_round6.log):
    // the residual mode stored the state under `if (fl & F_CELL)` at the point of the highest register pressure, and hipcc 7.2
    // placed eight AGPR spill copies of values that are live for ALL lanes (the flag word, the lane number, ...) into the FLOW
    // block of that if / else, ahead of the s_andn2_saveexec that flips EXEC -- where they execute under the THEN mask.  The
    // lanes of cells beyond the mesh (whole waves for k = 5: one cell per wave) came back from the join with whatever the
    // AGPRs held before, i.e. with random flags and lane numbers, and stored their sums over the rows of cell (0, 0) of the
    // cross-section -- the valid cell of a partial tile -- or to wild addresses.  Proof: the SAME listing with those eight
    // copies moved behind the s_or_b64 exec that closes the region (scripts/dev/isa_patch_build.py flow-spills) is exact,
    // twelve runs of twelve; the unpatched one fails every run.  Fix at the source: no branch there -- cells beyond the
    // mesh store into a sink (HXArgs::lin_sink); scripts/dev/isa_flow_audit.py lists such copies per kernel (now: none in
    // any unit of the library).
#ifndef HOX_EXT_LB
#define HOX_EXT_LB 1
#endif
    __global__ __launch_bounds__(NTH, (EXT ? HOX_EXT_LB : (RES ? HOX_RES_LB : (RCP ? HOX_RCP_LB : (hox_deep<K, LIN_MODE, RES, VARCO, RCP>() ? 1 : HOX_LB)))))
      void ns_hox_kernel(const HXArgs A)
    {
      static_assert(!EXT || (RES && LIN_MODE != 0 && !RCP && HOX_FUSED && HOX_RES_FUSED), "extrapolating residual: semi-implicit / explicit scheme");
      static_assert(!RCP || (!RES && !VARCO && LIN_MODE != 2 && HOX_FUSED), "recompute-state mode: Newton / Picard vmult, constant coefficients");
      using G           = Geo<K>;
      constexpr int N = G::N, NP = G::NP, KP = G::KP, NL = G::NL, N3 = G::N3, NN = N * N;
      constexpr int CPW = G::CPW, PL = G::PL, CY = G::CY, CZ = G::CZ, CWY = G::CWY, CWZ = G::CWZ, WY = G::WY;
      constexpr int TNY = G::TNY, TNZ = G::TNZ, TPY = G::TPY, TPZ = G::TPZ, RIMU = G::RIMU, RIMP = G::RIMP;
      constexpr int BUF = G::BUF, PUBD = G::PUBD, PUBV = G::PUBV;
      constexpr bool FUSED = HOX_FUSED && (!RES || HOX_RES_FUSED);     // value + gradient through two exchanges, z-line loop
      constexpr int NSTL = (RES || RCP) ? 0 : nst_of(LIN_MODE);       // linearisation values READ per point
      constexpr int NST = NSTL + (VARCO ? 4 : 0), NPC = NST / 2;       // ... with the coefficients (rho, mu | damping, -)
      static_assert(!(RES && VARCO) || EXT || LIN_MODE != 2, "variable-coefficient residual: not for Stokes");
      constexpr int NSO = RES ? nst_of(LIN_MODE) : 0, NPO = NSO / 2;  // ... WRITTEN per point (residual mode)
      // RES + VARCO (round 6; two-phase flow with k >= 3, :636-642, :717-732, :827-845): the coefficients arrive as a stream
      // of their own, two pieces per point ((rho, mu), (damping, -): the layout of a Stokes-type state with coefficients),
      // and leave behind the state pieces, so that what the residual writes IS the streaming state of the
      // variable-coefficient vmults (NPO + 2 pieces per point)
      constexpr int NPOV = NPO + ((RES && VARCO) ? 2 : 0);
      static_assert(!RES || WITH_P, "the residual has both blocks");
      using TB = Tab<K>;
      double *const lds = dyn_lds();

      const int tid = threadIdx.x, hw_lane = tid & 63;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int cw = hw_lane / PL, l0 = hw_lane % PL;
      // lanes beyond the (k+1)^2 lines of a cell leave at once (k = 4: 7 of 32, k = 5: 28 of 64): the rest of the
      // kernel runs under ONE exec mask, no LDS store needs a branch (s_barrier counts waves, not lanes).
      // HOX_MIRROR = 1 (diagnostic, round 6): in the residual modes the spare lanes stay and MIRROR a line of their cell -- the
      // same loads, the same stores of the same values to the same addresses --, so that the kernel runs with a full EXEC
      // mask.  Built to test whether the k = 5 fault (HOX_EXT_LB above) comes from retired lanes: it does not.
#ifndef HOX_MIRROR
#define HOX_MIRROR 0
#endif
      constexpr bool MIRROR = HOX_MIRROR && RES && 2 * NL >= PL;
      if (!MIRROR && l0 >= NL)
        return;
      const int l = (MIRROR && l0 >= NL) ? l0 - NL : l0, lane = cw * PL + l; // (lane: the logical lane all roles derive from)
      constexpr bool active = true;
      const int      lc = l;
      const int      a = lc % N, b = lc / N;
      const int  cyl = (wave % WY) * CWY + cw % CWY, czl = (wave / WY) * CWZ + cw / CWY;

      const long nwg = A.wg_list ? (long)A.wg_count : (long)A.tiles_y * A.tiles_z * A.n_chunks;
      long       wg  = xcd_remap(blockIdx.x, nwg);
      if (A.wg_list)
        wg = A.wg_list[A.wg_offset + wg];
      const int bx = (int)(wg % A.n_chunks), bt = (int)(wg / A.n_chunks);
      const int by = bt % A.tiles_y, bz = bt / A.tiles_y;
      const int cx0 = bx * A.LX, ns = min(A.LX, A.ncx - cx0);
      const int tcy = min(CY, A.ncy - by * CY), tcz = min(CZ, A.ncz - bz * CZ);
      const bool cell_ok = cyl < tcy && czl < tcz, valid = active && cell_ok;
      // cells outside the mesh compute on cell (0,0) of the cross-section: nothing of theirs is emitted and no
      // valid cell collects from them; all addresses stay legal
      const int  cy = by * CY + (cell_ok ? cyl : 0), cz = bz * CZ + (cell_ok ? czl : 0);
      const bool lasty = cyl == tcy - 1, lastz = czl == tcz - 1;
      const bool x_seam_end = cx0 + ns < A.ncx; // the chunk ends inside the mesh

      // ---- x-layout roles of this lane: x-line (j,k) = (a,b) of its cell -------------------------------------
      const int  J = K * cy + a, Kz = K * cz + b;
      const bool own_u  = valid && (a < K || lasty) && (b < K || lastz);
      const bool seam_u = (a == K && J < A.nny - 1) || (b == K && Kz < A.nnz - 1); // (given own_u: high rim of the workgroup)
      unsigned   cmask  = 0; // components constrained on the whole line (y / z faces)
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if ((J == 0 && (A.con_u >> (6 + d) & 1)) || (J == A.nny - 1 && (A.con_u >> (9 + d) & 1)) ||
            (Kz == 0 && (A.con_u >> (12 + d) & 1)) || (Kz == A.nnz - 1 && (A.con_u >> (15 + d) & 1)))
          cmask |= 1u << d;
      const bool pth = active && a < NP && b < NP;
      const int  ap = min(a, KP), bp = min(b, KP);
      const int  Jp = KP * cy + ap, Kp = KP * cz + bp;
      const bool own_p  = valid && pth && (a < KP || lasty) && (b < KP || lastz);
      const bool seam_p = (a == KP && Jp < A.npy - 1) || (b == KP && Kp < A.npz - 1);
      const bool pcon   = (Jp == 0 && (A.con_p >> 2 & 1)) || (Jp == A.npy - 1 && (A.con_p >> 3 & 1)) ||
                        (Kp == 0 && (A.con_p >> 4 & 1)) || (Kp == A.npz - 1 && (A.con_p >> 5 & 1));
      // does this workgroup touch a constrained face at all (wave-uniform: guards the rare +-src stores)
      const bool wg_con_yz = (by == 0 && ((A.con_u >> 6 & 7u) || (A.con_p >> 2 & 1u))) ||
                             (by == A.tiles_y - 1 && ((A.con_u >> 9 & 7u) || (A.con_p >> 3 & 1u))) ||
                             (bz == 0 && ((A.con_u >> 12 & 7u) || (A.con_p >> 4 & 1u))) ||
                             (bz == A.tiles_z - 1 && ((A.con_u >> 15 & 7u) || (A.con_p >> 5 & 1u)));

      enum
      {
        F_OWN_U = 1, F_SEAM_U = 2, F_CON0 = 4, F_OWN_P = 32, F_SEAM_P = 64, F_PCON = 128, F_CY = 256, F_CZ = 512,
        F_AK = 1024, F_BK = 2048, F_PTH = 4096, F_AKP = 8192, F_BKP = 16384, F_ACT = 32768, F_CELL = 65536
      };
      const unsigned flags = (own_u ? F_OWN_U : 0) | (seam_u ? F_SEAM_U : 0) | (cmask * F_CON0) | (own_p ? F_OWN_P : 0) |
                             (seam_p ? F_SEAM_P : 0) | (pcon ? F_PCON : 0) | ((active && a == 0 && cyl > 0) ? F_CY : 0) |
                             ((active && b == 0 && czl > 0) ? F_CZ : 0) | ((active && a == K) ? F_AK : 0) |
                             ((active && b == K) ? F_BK : 0) | (pth ? F_PTH : 0) | ((pth && a == KP) ? F_AKP : 0) |
                             ((pth && b == KP) ? F_BKP : 0) | (active ? F_ACT : 0) | (valid ? F_CELL : 0);

      const double wab = A.tab[TB::C + TB::C_DET] * A.tab[TB::C + TB::C_W + a] * A.tab[TB::C + TB::C_W + b];

      // global rows: wave-uniform base pointer + 32-bit per-lane offset (doubles)
      const unsigned urow = (unsigned)(((size_t)Kz * A.nny + J) * A.nnx * 3), prow = (unsigned)(((size_t)Kp * A.npy + Jp) * A.npx);
      // state: per-lane pointer to the first piece of the first cell of my row of cell groups
      // state: wave-uniform base of my wave's row of cell groups + 32-bit lane offset.  (A cell beyond the mesh reads
      // its own slot of the zero-padded group, a group beyond the mesh the last one: legal addresses, unused values.)
      constexpr unsigned ST_PIECE = CPW * NL * 2, ST_POINT = NPC * ST_PIECE, ST_CELL = N * ST_POINT; // doubles (read)
      constexpr unsigned SO_POINT = NPOV * ST_PIECE, SO_CELL = N * SO_POINT;                          // ... (written)
      const int          gyw = min((by * CY + (wave % WY) * CWY) / CWY, A.ngy - 1), gzw = min((bz * CZ + (wave / WY) * CWZ) / CWZ, A.ngz - 1);
      const double *const stg = A.lin + (NST > 0 ? ((size_t)gzw * A.ngy + gyw) * A.ncx * ST_CELL : 0);
      double *const       sog = RES && NSO > 0 ? A.lin_out + ((size_t)gzw * A.ngy + gyw) * A.ncx * SO_CELL : nullptr;
      const unsigned      st_lane = (unsigned)(cw * NL + lc) * 2;

      // wave-private transposition buffers T0, T1, T2 and the lane's line bases in the three layouts
      double *const  WB  = lds + wave * G::WAVE + cw * N3;
      double *const  px  = WB + N * a + NN * b; // x-line (., a, b): stride 1
      double *const  py  = WB + a + NN * b;     // y-line (a, ., b): stride N
      double *const  pz  = WB + a + N * b;      // z-line (a, b, .): stride N*N
      const unsigned ax = lds_byte_addr(px), ay = lds_byte_addr(py), az = lds_byte_addr(pz);
      double *const  PUBY = lds + 4 * G::WAVE, *const PUBZ = PUBY + G::PUBY_BUFS * PUBD; // publish areas (Geo)
      // state ring of my wave
      constexpr bool RING = G::RING && NST > 0 && !VARCO && !(HOX_EXP & 4);
      constexpr bool DEEP = hox_deep<K, LIN_MODE, RES, VARCO, RCP>();
      // k = 5 spills a few registers in the integrate phase; a scratch reload is followed by vmcnt(0), which waits for every
      // copy in flight -- so the two points of the NEXT cell are issued after the integrate phase, not at the end of the
      // quadrature loop (they still have the combine, emit and evaluate phases, > 10 k cycles, to arrive)
#ifndef HOX_LATE_RING
#define HOX_LATE_RING 1
#endif
      constexpr bool LATE_RING = HOX_LATE_RING && RING && !DEEP && K == 5;
      constexpr int  RPTS = DEEP ? N : 2, RING_BYTES = DEEP ? G::RING_BYTES_DEEP : G::RING_BYTES; // points in the ring
      constexpr int  SLOT = G::SLOT, RS = RPTS * (NST / 2 > 0 ? NST / 2 : 1);
      char *const    ring = reinterpret_cast<char *>(lds + 4 * G::WAVE + G::PUB_DOUBLES) + wave * RING_BYTES;
      const unsigned ring_m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_byte_addr(ring)), ring_lane = lds_byte_addr(ring) + 16 * hw_lane;
      // hand-off flags (HOX_FLAGS): pub[w] = number of combines wave w has published, done[w] = ... has collected
      const unsigned flag0 = lds_byte_addr(reinterpret_cast<char *>(lds + 4 * G::WAVE + G::PUB_DOUBLES) + 4 * RING_BYTES);
      int            seq   = 0;
      if (HOX_FLAGS)
        {
          if (lane < 8 && wave == 0)
            lds_flag_set(flag0 + 4 * lane, 0);
          __syncthreads();
        }

      const ctab_t tab = as_ctab(A.tab);

      // nodal x-lines of the step to come
      double Un[3][N], Pn[NP], Ln[(RCP || EXT) ? 3 : 1][N];
      auto   load_nodes = [&](const int cxn) {
        if ((HOX_EXP & 32) && cxn > cx0)
          return;
        const int     cxc = min(cxn, A.ncx - 1);
        const double *pu  = A.src_u + (size_t)(K * cxc) * 3;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int d = 0; d < 3; ++d)
            Un[d][i] = (pu + urow)[i * 3 + d]; // (ONE per-lane address + immediate offsets: unsigned index sums are 18 addresses, hoisted and spilled)
        if constexpr (RCP || EXT)
          {
            const double *pl = A.lin_u + (size_t)(K * cxc) * 3;
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
              for (int d = 0; d < 3; ++d)
                Ln[d][i] = (pl + urow)[i * 3 + d];
          }
        if (WITH_P)
          {
            const double *pp = A.src_p + (size_t)(KP * cxc);
#pragma unroll
            for (int i = 0; i < NP; ++i)
              Pn[i] = (pp + prow)[i];
          }
      };
      double st[RCP ? nst_of(LIN_MODE) : (NST > 0 ? NST : 1)];
      auto   load_state = [&](const double *const base, const unsigned off) { // base wave-uniform, off per lane
#pragma unroll
        for (int e = 0; e < NPC; ++e)
          {
            st[2 * e]     = (base + off)[e * ST_PIECE]; // (one per-lane address + immediate offsets)
            st[2 * e + 1] = (base + off)[e * ST_PIECE + 1];
          }
      };

#if HOX_STAMP
      unsigned long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock_now();
#define HOX_MARK(j)                          \
  {                                          \
    const unsigned long long tn = clock_now(); \
    acc[j] += tn - tlast;                    \
    tlast = tn;                              \
  }
#else
#define HOX_MARK(j)
#endif
      // ---- state ring (k = 4): the quadrature loop eats a point's state in ~200 cycles, an HBM round trip takes
      // ~2700, and there are no registers to fetch points ahead.  LDS-DMA moves the pieces into a ring of two
      // points per wave; a point's pieces are re-issued two points ahead as soon as its slots have been read, the
      // first two points of the NEXT step while this step integrates.  Piece q = point * NPC + e lives in slot q % RS.
      auto ring_issue = [&](const double *const cell_base, auto pt_, auto half_) {
        constexpr int PT = decltype(pt_)::value, H = decltype(half_)::value;
        const double *const sb = uniform_ptr(cell_base);
#pragma unroll
        for (int e = 0; e < NPC; ++e)
          dma_b128(sb + (PT * NPC + e) * ST_PIECE, st_lane * 8, ring_m0 + (H * NPC + e) * SLOT);
      };
      if (RING)
        {
          const double *const c0 = stg + (size_t)cx0 * ST_CELL;
          static_for<RPTS>([&](auto p_) { ring_issue(c0, p_, p_); });
        }

      // ---- combine the partial sums of the cross-section per owned line, emit NV nodes -------------------------
      // R / Rp: the lane's x-line sums, nodes 0 .. NV-1 (NVP-1) are final in x.  I0 / Ip0: global x index of
      // node 0; xl0: its index inside the chunk; endplane: the nodes form the chunk's last plane
      auto combine = [&](auto nv_, auto nvp_, double (&R)[3][N], double (&Rp)[NP], const unsigned fl, int lane_o,
                         const int parity, const int I0, const int Ip0, const int xl0, const int xlp0, const bool endplane) {
        constexpr int NV = decltype(nv_)::value, NVP = decltype(nvp_)::value;
        // the lane's indices are re-derived from an opaque copy of its number: computed once before the marching loop
        // they would stay live (or be spilled) across the whole step for one use here
        opaque(lane_o);
        const int l = lane_o % PL, cw = lane_o / PL, a = l % N, b = l / N;
        const int cyl = (wave % WY) * CWY + cw % CWY, czl = (wave / WY) * CWZ + cw / CWY, cell = czl * CY + cyl;
        const int jl = K * cyl + a, kl = K * czl + b, jlp = KP * cyl + min(a, KP), klp = KP * czl + min(b, KP);
        const bool     ok_ = cyl < tcy && czl < tcz;
        const int      cy_ = by * CY + (ok_ ? cyl : 0), cz_ = bz * CZ + (ok_ ? czl : 0);
        const unsigned urow = (unsigned)(((size_t)(K * cz_ + b) * A.nny + (K * cy_ + a)) * A.nnx * 3),
                       prow = (unsigned)(((size_t)(KP * cz_ + min(b, KP)) * A.npy + (KP * cy_ + min(a, KP))) * A.npx);
        double *const PY = PUBY + (G::PUBY_BUFS == 2 ? parity * PUBD : 0), *const PZ = PUBZ + parity * PUBD;
        constexpr int WZ_ = G::WZ;
        const int     wy_ = wave % WY, wz_ = wave / WY;
        if (HOX_FLAGS)
          {
            // the buffers of this parity were last filled two combines ago: my upper neighbours must have collected them
            ++seq;
            if (seq > 2)
              {
                if (wy_ + 1 < WY)
                  lds_flag_wait(flag0 + 16 + 4 * (wave + 1), seq - 2);
                if (wz_ + 1 < WZ_)
                  lds_flag_wait(flag0 + 16 + 4 * (wave + WY), seq - 2);
                if (wy_ + 1 < WY && wz_ + 1 < WZ_)
                  lds_flag_wait(flag0 + 16 + 4 * (wave + WY + 1), seq - 2);
              }
          }
        wave_sync(); // (the single y buffer: every lane of the wave is done with the last collect)
        if (!(HOX_EXP & 64))
          {
        if (fl & F_AK)
          {
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
              for (int i = 0; i < NV; ++i)
                PY[((cell * N + b) * 3 + d) * K + i] = R[d][i];
          }
        if (fl & F_BK)
          {
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
              for (int i = 0; i < NV; ++i)
                PZ[((cell * N + a) * 3 + d) * K + i] = R[d][i];
          }
        if (WITH_P)
          {
            if (fl & F_AKP)
              {
#pragma unroll
                for (int i = 0; i < NVP; ++i)
                  PY[PUBV + (cell * NP + b) * KP + i] = Rp[i];
              }
            if (fl & F_BKP)
              {
#pragma unroll
                for (int i = 0; i < NVP; ++i)
                  PZ[PUBV + (cell * NP + a) * KP + i] = Rp[i];
              }
          }
        HOX_MARK(6)
        if (HOX_FLAGS)
          {
            lds_flag_set(flag0 + 4 * wave, seq); // (after my publish writes: the LDS serves a wave's requests in order)
            if (wy_ > 0)
              lds_flag_wait(flag0 + 4 * (wave - 1), seq);
            if (wz_ > 0)
              lds_flag_wait(flag0 + 4 * (wave - WY), seq);
            if (wy_ > 0 && wz_ > 0)
              lds_flag_wait(flag0 + 4 * (wave - WY - 1), seq);
            wave_sync();
          }
        else
          lds_barrier(); // (LDS only: the prefetches and stores of the step stay in flight)
        // lower neighbour in y: its line (K, b); in z: its line (a, K); both: the corner line (K, K) of the diagonal
        // cell, which that cell published in both directions -- taken from the z area (double-buffered)
        if (fl & F_CY)
          {
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
              for (int i = 0; i < NV; ++i)
                R[d][i] += PY[(((cell - 1) * N + b) * 3 + d) * K + i];
            if (WITH_P && (fl & F_PTH))
              {
#pragma unroll
                for (int i = 0; i < NVP; ++i)
                  Rp[i] += PY[PUBV + ((cell - 1) * NP + b) * KP + i];
              }
          }
        if (fl & F_CZ)
          {
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
              for (int i = 0; i < NV; ++i)
                R[d][i] += PZ[(((cell - CY) * N + a) * 3 + d) * K + i];
            if (WITH_P && (fl & F_PTH))
              {
#pragma unroll
                for (int i = 0; i < NVP; ++i)
                  Rp[i] += PZ[PUBV + ((cell - CY) * NP + a) * KP + i];
              }
          }
        if ((fl & (F_CY | F_CZ)) == (F_CY | F_CZ))
          {
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
              for (int i = 0; i < NV; ++i)
                R[d][i] += PZ[(((cell - CY - 1) * N + K) * 3 + d) * K + i];
            if (WITH_P && (fl & F_PTH))
              {
#pragma unroll
                for (int i = 0; i < NVP; ++i)
                  Rp[i] += PZ[PUBV + ((cell - CY - 1) * NP + KP) * KP + i];
              }
          }
          }
        if (HOX_FLAGS)
          lds_flag_set(flag0 + 16 + 4 * wave, seq); // (waits for my collect reads: lgkmcnt(0))
        HOX_MARK(7)
        // ---- emit: dst, or the slab on the high rim of the workgroup, or the x-slab at the end of a chunk ------
        const bool to_xslab = endplane && x_seam_end;
        if ((HOX_EXP & 16) && I0 >= 0) // (diagnostic: keep the sums alive, store nothing)
          {
            sink(R[0][0] + R[1][0] + R[2][0] + Rp[0]);
            return;
          }
        if (fl & F_OWN_U)
          {
            double *tp;
            if (to_xslab)
              tp = A.xslab_u + (((size_t)wg * TNZ + kl) * TNY + jl) * 3;
            else if (fl & F_SEAM_U)
              tp = A.slab_u + (((size_t)wg * RIMU + rim_line<TNY, TNZ>(jl, kl)) * (K * A.LX + 1) + xl0) * 3;
            else
              tp = A.dst_u + (size_t)I0 * 3 + urow;
            if (HOX_EXP & 128) // (diagnostic, wrong results: every chunk starts on a 32-byte boundary)
              tp = (double *)((size_t)tp & ~(size_t)31);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
              for (int d = 0; d < 3; ++d)
                tp[i * 3 + d] = R[d][i];
          }
        if (WITH_P && A.integrate_p && (fl & F_OWN_P))
          {
            double *tp;
            if (to_xslab)
              tp = A.xslab_p + (((size_t)wg * TPZ + klp) * TPY + jlp);
            else if (fl & F_SEAM_P)
              tp = A.slab_p + (((size_t)wg * RIMP + rim_line<TPY, TPZ>(jlp, klp)) * (KP * A.LX + 1) + xlp0);
            else
              tp = A.dst_p + (size_t)Ip0 + prow;
#pragma unroll
            for (int i = 0; i < NVP; ++i)
              tp[i] = Rp[i];
          }
        // constrained rows are then set to +-src (:247-256): a later store of the same lane to the same
        // address, or an entry the fix-up kernel skips (workgroups at the domain boundary only)
        const bool xcon = (I0 == 0 && ((A.con_u & 7u) || (A.con_p & 1u))) ||
                          (I0 + NV - 1 == A.nnx - 1 && ((A.con_u >> 3 & 7u) || (A.con_p >> 1 & 1u)));
        if ((wg_con_yz || xcon) && !to_xslab)
          {
            if ((fl & F_OWN_U) && !(fl & F_SEAM_U))
              {
#pragma unroll 1
                for (int i = 0; i < NV; ++i)
                  {
                    const size_t po = (size_t)(I0 + i) * 3 + urow;
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                      if ((fl & (F_CON0 << d)) || (I0 + i == 0 && (A.con_u >> d & 1)) ||
                          (I0 + i == A.nnx - 1 && (A.con_u >> (3 + d) & 1)))
                        A.dst_u[po + d] = RES ? 0. : A.src_u[po + d];
                  }
              }
            if (WITH_P && A.integrate_p && (fl & F_OWN_P) && !(fl & F_SEAM_P))
              {
#pragma unroll 1
                for (int i = 0; i < NVP; ++i)
                  {
                    const size_t po = (size_t)(Ip0 + i) + prow;
                    if ((fl & F_PCON) || (Ip0 + i == 0 && (A.con_p & 1)) || (Ip0 + i == A.npx - 1 && (A.con_p >> 1 & 1)))
                      A.dst_p[po] = RES ? 0. : -A.src_p[po];
                  }
              }
          }
      };

      double carry[3] = {0., 0., 0.}, carry_p = 0.;
      load_nodes(cx0);

#pragma unroll 1
      for (int step = 0; step < ns; ++step)
        {
          const int cx = cx0 + step;
          unsigned  fl = flags;
          opaque(fl);
          // every use re-loads its 1D matrix through an opaque copy of the table pointer (scalar loads): kept live
          // across the step, the three matrices alone would need more scalar registers than a wave has
          auto tb = [&](const int off) {
            ctab_t t = tab;
            opaque(t);
            return t + off;
          };

          // G[d][0..3][i]: value, d/dx, d/dy, d/dz (reference cell) of component d at my N quadrature points;
          // after the quadrature loop: tested value and tested gradient
          double G[3][4][N], PQ[N];
          const double *const stc = (HOX_EXP & 8) ? A.lin : stg + (size_t)cx * ST_CELL; // (& 8: every wave streams the same 24 KB: L2 hits)
          const double *const stn = (HOX_EXP & 8) ? A.lin : stg + (size_t)min(cx + 1, cx0 + ns - 1) * ST_CELL;
          if (NST > 0 && !(HOX_EXP & 4) && !RING)
            load_state(stc, st_lane);

          HOX_MARK(0)
          // ================= evaluate (FEEvaluation::evaluate, :668-671) =====================================
          auto nodal_u = [&](auto d_, double (&U)[N]) {
            constexpr int d = decltype(d_)::value;
#pragma unroll
            for (int i = 0; i < N; ++i)
              U[i] = (!RES && (fl & (F_CON0 << d))) ? 0. : Un[d][i]; // read_dof_values: constrained entries read as zero
            if (RES) // (read_dof_values_plain: the boundary values take part, :662-671)
              return;
            if (cx == 0 && (A.con_u >> d & 1))
              U[0] = 0.;
            if (cx == A.ncx - 1 && (A.con_u >> (3 + d) & 1))
              U[K] = 0.;
          };
          auto nodal_p = [&](double (&P)[NP]) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
              P[i] = (!RES && (fl & F_PCON)) ? 0. : Pn[i];
            if (RES)
              return;
            if (cx == 0 && (A.con_p & 1))
              P[0] = 0.;
            if (cx == A.ncx - 1 && (A.con_p >> 1 & 1))
              P[KP] = 0.;
          };
          // nodal x-line -> value, d/dx, d/dy, d/dz at the Gauss points of my z-line (HOX_FUSED)
          auto eval_fused = [&](double (&U)[N], double (&O)[4][N]) {
            double           l0[N], l1[N], l2[N];
            EoMat<N, N, 1>  mS;
            EoMat<N, N, -1> mX;
            mS.load(tb(TB::S));
            mX.load(tb(TB::DS));
            mS.template apply<false>(U, l0); // S_x u
            mX.template apply<false>(U, l1); // (D S)_x u
            wr_line<0, 1, N>(px, l0);
            wr_line<BUF, 1, N>(px, l1);
            wave_sync();
            mS.load(tb(TB::S));
            mX.load(tb(TB::DS));
            rd_line<0, N, N>(ay, l0);
            rd_line<BUF, N, N>(ay, l1);
            ds_wait<0>(l0);
            ds_wait<0>(l1);
            mX.template apply<false>(l0, l2); // (D S)_y S_x u
            mS.template apply<false>(l0, U);  // S_y S_x u
            mS.template apply<false>(l1, l0); // S_y (D S)_x u
            wave_sync();
            wr_line<0, N, N>(py, U);
            wr_line<BUF, N, N>(py, l0);
            wr_line<2 * BUF, N, N>(py, l2);
            wave_sync();
            mS.load(tb(TB::S));
            mX.load(tb(TB::DS));
            rd_line<0, NN, N>(az, l0);
            rd_line<BUF, NN, N>(az, l1);
            rd_line<2 * BUF, NN, N>(az, l2);
            ds_wait<2 * N>(l0);
            mS.template apply<false>(l0, O[0]); // value
            mX.template apply<false>(l0, O[3]); // d/dz
            ds_wait<N>(l1);
            mS.template apply<false>(l1, O[1]); // d/dx
            ds_wait<0>(l2);
            mS.template apply<false>(l2, O[2]); // d/dy
            wave_sync();
          };
          // one velocity component.  HOX_FUSED: x in registers (S u and (D S) u), y (three arrays), z: value and gradient
          // at the Gauss points of my z-line -- two exchanges; otherwise x, y, z (values), then d/dz, d/dy, d/dx by
          // collocation, back on x-lines -- five exchanges
          auto eval_single = [&](auto d_) {
            constexpr int d = decltype(d_)::value;
            if constexpr (FUSED)
              {
                double U[N];
                nodal_u(d_, U);
                eval_fused(U, G[d]);
                return;
              }
            double        U[N], T[N], ln[N];
            nodal_u(d_, U);
            EoMat<N, N, 1>  mS;
            EoMat<N, N, -1> mD;
            mS.load(tb(TB::S));
            mS.template apply<false>(U, T); // x: nodes -> Gauss points
            wr_line<0, 1, N>(px, T);
            wave_sync();
            mS.load(tb(TB::S));
            rd_line<0, N, N>(ay, ln);
            ds_wait<0>(ln);
            mS.template apply<false>(ln, T); // y
            wave_sync();
            wr_line<BUF, N, N>(py, T);
            wave_sync();
            mS.load(tb(TB::S));
            mD.load(tb(TB::D));
            rd_line<BUF, NN, N>(az, ln);
            ds_wait<0>(ln);
            mS.template apply<false>(ln, T); // z: values at the Gauss points of my z-line
            mD.template apply<false>(T, ln); // d/dz (collocation)
            wave_sync();
            wr_line<0, NN, N>(pz, T);
            wr_line<2 * BUF, NN, N>(pz, ln);
            wave_sync();
            mD.load(tb(TB::D));
            rd_line<0, N, N>(ay, ln);
            ds_wait<0>(ln);
            mD.template apply<false>(ln, T); // d/dy
            wave_sync();
            wr_line<BUF, N, N>(py, T);
            wave_sync();
            mD.load(tb(TB::D));
            rd_line<0, 1, N>(ax, G[d][0]);
            rd_line<BUF, 1, N>(ax, G[d][2]);
            rd_line<2 * BUF, 1, N>(ax, G[d][3]);
            ds_wait<2 * N>(G[d][0]);
            mD.template apply<false>(G[d][0], G[d][1]); // d/dx
            ds_wait<0>(G[d][2]);
            ds_wait<0>(G[d][3]);
            wave_sync();
          };
          auto eval_p = [&]() {
            double P[NP], T[N], ln[NP];
            nodal_p(P);
            EoMat<N, NP, 1> mP;
            mP.load(tb(TB::SP));
            mP.template apply<false>(P, T); // x: [N][NP x NP lines]
            wr_line<0, 1, N>(px, T); // (lanes without a pressure line store values nobody uses: no branch)
            wave_sync();
            mP.load(tb(TB::SP));
            rd_line<0, N, NP>(ay, ln); // y-line (a, ., b), b < NP (other lanes read defined-or-not values they never use)
            ds_wait<0>(ln);
            mP.template apply<false>(ln, T);
            wave_sync();
            wr_line<BUF, N, N>(py, T);
            wave_sync();
            mP.load(tb(TB::SP));
            rd_line<BUF, NN, NP>(az, ln); // z-line (a, b, .)
            ds_wait<0>(ln);
            if constexpr (FUSED) // (the quadrature loop runs on z-lines)
              {
                mP.template apply<false>(ln, PQ);
                wave_sync();
                return;
              }
            mP.template apply<false>(ln, T);
            wave_sync();
            wr_line<0, NN, N>(pz, T);
            wave_sync();
            rd_line<0, 1, N>(ax, PQ);
            ds_wait<0>(PQ);
            wave_sync();
          };
          // residual mode: values of the old-solution combination at my quadrature points (interpolation is linear:
          // weight_old u_old + weight_old_old u_old_old was formed once per node by the launcher); x, y, z, back to x-lines
          double OQ[RES ? 3 : 1][N];
          auto   eval_old = [&](auto d_) {
            constexpr int d = decltype(d_)::value;
            double        U[N], T[N], ln[N];
            const double *po = A.old_u + (size_t)(K * cx) * 3;
#pragma unroll
            for (int i = 0; i < N; ++i)
              U[i] = (po + urow)[i * 3 + d];
            EoMat<N, N, 1> mS;
            mS.load(tb(TB::S));
            mS.template apply<false>(U, T);
            wr_line<0, 1, N>(px, T);
            wave_sync();
            mS.load(tb(TB::S));
            rd_line<0, N, N>(ay, ln);
            ds_wait<0>(ln);
            mS.template apply<false>(ln, T);
            wave_sync();
            wr_line<BUF, N, N>(py, T);
            wave_sync();
            mS.load(tb(TB::S));
            rd_line<BUF, NN, N>(az, ln);
            ds_wait<0>(ln);
            if constexpr (FUSED)
              {
                mS.template apply<false>(ln, OQ[RES ? d : 0]);
                wave_sync();
                return;
              }
            mS.template apply<false>(ln, T);
            wave_sync();
            wr_line<0, NN, N>(pz, T);
            wave_sync();
            rd_line<0, 1, N>(ax, OQ[RES ? d : 0]);
            ds_wait<0>(OQ[RES ? d : 0]);
            wave_sync();
          };
          using I0_ = std::integral_constant<int, 0>;
          using I1_ = std::integral_constant<int, 1>;
          using I2_ = std::integral_constant<int, 2>;
          if constexpr (RES)
            {
              if (A.old_u) // (wave-uniform)
                {
                  eval_old(I0_{});
                  eval_old(I1_{});
                  eval_old(I2_{});
                }
              else
                {
#pragma unroll
                  for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int i = 0; i < N; ++i)
                      OQ[d][i] = 0.;
                }
            }
          // recompute-state mode: (u_lin, grad u_lin) at my quadrature points from the nodal linearisation point (plain
          // read: the boundary values take part, as in the residual that produced the streamed state)
          double GL[(RCP || EXT) ? 3 : 1][4][N];
          if constexpr (RCP || EXT)
            {
              eval_fused(Ln[0], GL[0]);
              eval_fused(Ln[1], GL[1]);
              eval_fused(Ln[2], GL[2]);
            }
          eval_single(I0_{});
          eval_single(I1_{});
          eval_single(I2_{});
          HOX_MARK(1)
          if constexpr (WITH_P)
            eval_p();
          if constexpr (!WITH_P)
            {
#pragma unroll
              for (int i = 0; i < N; ++i)
                PQ[i] = 0.;
            }

          HOX_MARK(2)
          // ================= quadrature points of my line (:702-893) ============================================
          static_for<N>([&](auto i_) {
              constexpr int i_c = decltype(i_)::value, i = i_c;
              if constexpr (RING)
                {
                  // pieces of point i: issued two points ago (or during the last step); everything younger is the
                  // NPC pieces issued one point ago
                  if (i == 0)
                    wait_vmcnt<0>();
                  else if (i >= 2 && !DEEP)
                    {
                      if (LATE_RING && i == N - 1) // (nothing was issued at point N - 2: my pieces are the youngest)
                        wait_vmcnt<0>();
                      else
                        wait_vmcnt<NPC>();
                    }
                  ring_read_impl<(i_c % RPTS) * NPC, NPC, RS, SLOT>(ring_lane, st);
                  // the slots are free again: the same half of the ring takes the next point of its parity (in the
                  // last step of the chunk that is the same cell once more, never read: no branch in this loop)
                  if constexpr (DEEP) // (the whole cell arrived a step ago; the slots take the same point of the next cell)
                    ring_issue(stn, i_, i_);
                  else if constexpr (i_c + 2 < N)
                    ring_issue(stc, std::integral_constant<int, i_c + 2>{}, std::integral_constant<int, i_c % 2>{});
                  else if constexpr (!LATE_RING)
                    ring_issue(stn, std::integral_constant<int, i_c % 2>{}, std::integral_constant<int, i_c % 2>{});
                }
              const ctab_t cst = tb(TB::C); // constants of the quadrature-point operation
              const double jxw = wab * cst[TB::C_W + i];
              double       g[3][3], u[3];
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  u[d] = G[d][0][i];
#pragma unroll
                  for (int e = 0; e < 3; ++e)
                    g[d][e] = G[d][1 + e][i] * cst[TB::C_IH + e];
                }
              const double div = g[0][0] + g[1][1] + g[2][2];
              if constexpr (RCP)
                {
#pragma unroll
                  for (int d = 0; d < 3; ++d)
                    st[d] = GL[d][0][i];
                  if constexpr (LIN_MODE == 0)
                    {
#pragma unroll
                      for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                          st[3 + 3 * d + e] = GL[d][1 + e][i] * cst[TB::C_IH + e];
                    }
                  else
                    st[3] = GL[0][1][i] * cst[TB::C_IH + 0] + GL[1][2][i] * cst[TB::C_IH + 1] + GL[2][3][i] * cst[TB::C_IH + 2];
                }
              // :717, :827-835, :841-845 with the coefficients of this point (taken before the next point's state
              // overwrites the registers)
              double cA_q, cB_q, tmu_q;
              if constexpr (VARCO)
                {
                  cA_q  = cst[TB::C_GAMMA] * st[NSTL] - cst[TB::C_DAMPF] * st[NSTL + 2];
                  cB_q  = cst[TB::C_T1RHO] * st[NSTL];
                  tmu_q = cst[TB::C_TAU1] * st[NSTL + 1];
                }
              else
                {
                  cA_q  = cst[TB::C_CA];
                  cB_q  = cst[TB::C_CB];
                  tmu_q = cst[TB::C_TMU];
                }
              double       conv[3];
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  double res = 0.;
                  if constexpr (RES && EXT) // :740-782 convection with the extrapolated velocity
                    {
                      const double ediv = GL[0][1][i] * cst[TB::C_IH + 0] + GL[1][2][i] * cst[TB::C_IH + 1] + GL[2][3][i] * cst[TB::C_IH + 2];
                      if constexpr (LIN_MODE == 2) // explicit: the extrapolated field convects itself
                        {
                          res = cst[TB::C_BETA] * ediv * GL[d][0][i];
#pragma unroll
                          for (int e = 0; e < 3; ++e)
                            res += GL[e][0][i] * (GL[d][1 + e][i] * cst[TB::C_IH + e]);
                        }
                      else // semi-implicit: it convects the solution
                        {
                          res = cst[TB::C_BETA] * ediv * u[d];
#pragma unroll
                          for (int e = 0; e < 3; ++e)
                            res += GL[e][0][i] * g[d][e];
                        }
                    }
                  else if constexpr (RES) // :783-799 the nonlinear term of the solution itself (no extrapolation)
                    {
                      if constexpr (LIN_MODE != 2)
                        {
                          res = cst[TB::C_BETA] * div * u[d];
#pragma unroll
                          for (int e = 0; e < 3; ++e)
                            res += u[e] * g[d][e];
                        }
                    }
                  else if constexpr (LIN_MODE == 0) // Newton :802-816; st = (u_lin[3], grad u_lin[3][3])
                    {
                      res = cst[TB::C_BETA] * (div * st[d] + (st[3] + st[7] + st[11]) * u[d]);
#pragma unroll
                      for (int e = 0; e < 3; ++e)
                        res += st[e] * g[d][e] + u[e] * st[3 + 3 * d + e];
                    }
                  else if constexpr (LIN_MODE == 1) // Picard-type :817-826; st = (u_lin[3], div u_lin)
                    {
                      res = cst[TB::C_BETA] * st[3] * u[d];
#pragma unroll
                      for (int e = 0; e < 3; ++e)
                        res += st[e] * g[d][e];
                    }
                  if constexpr (RES && VARCO) // (... times the density of the point, :827: A.c_old is 1 or 0 then)
                    conv[d] = (cA_q * u[d] + (A.c_old * st[NSTL]) * OQ[d][i] + cB_q * res) * jxw;
                  else if constexpr (RES) // :717-732 with the time derivative of BDF: weight u + (weight_old u_old + ...)
                    conv[d] = (cA_q * u[d] + A.c_old * OQ[d][i] + cB_q * res) * jxw;
                  else
                    conv[d] = (cA_q * u[d] + cB_q * res) * jxw; // :717,:827-835
                }
              if constexpr (RES && NSO > 0)
                {
                  // the state of the vmults to come, in their streaming layout: piece e = values (2 e, 2 e + 1) of
                  // (u, grad u) row-major (Newton) or (u, div u) (Picard-type).  Cells beyond the mesh store into a sink
                  // behind the state -- NOT "nothing" under `if (fl & F_CELL)`: this is the point of the highest register
                  // pressure of the kernel, and hipcc 7.2 put the AGPR spills of values that are live for ALL lanes (the
                  // flag word, the lane number) into the Flow block of that if / else, where they execute under the THEN
                  // mask: lanes of cells beyond the mesh -- whole waves for k = 5 -- came back from the join with stale
                  // flags and stored to wild addresses (the k = 5 extrapolating residual of rounds 5, 6)
                    {
                      // (x-line loop, lane (y, z) = (a, b), point i along x, next to a z-line vmult: the piece of point
                      // (i, a, b) belongs to line (x, y) = (i, a), point b)
                      double *const so_cell = (fl & F_CELL) ? sog + (size_t)cx * SO_CELL : A.lin_sink;
                      double *const so = (HOX_FUSED && !FUSED) ?
                                           so_cell + ((unsigned)(b * SO_POINT) + (unsigned)(cw * NL + i + N * a) * 2) :
                                           so_cell + ((unsigned)(i * SO_POINT) + st_lane);
                      if constexpr (EXT) // (u_ext, div u_ext): the state of the semi-implicit vmult
                        {
                          so[0] = GL[0][0][i], so[1] = GL[1][0][i];
                          so[ST_PIECE]     = GL[2][0][i];
                          so[ST_PIECE + 1] = GL[0][1][i] * cst[TB::C_IH + 0] + GL[1][2][i] * cst[TB::C_IH + 1] + GL[2][3][i] * cst[TB::C_IH + 2];
                        }
                      else
                        so[0] = u[0], so[1] = u[1];
                      if constexpr (EXT)
                        {
                        }
                      else if constexpr (LIN_MODE == 0)
                        {
                          so[ST_PIECE] = u[2], so[ST_PIECE + 1] = g[0][0];
                          so[2 * ST_PIECE] = g[0][1], so[2 * ST_PIECE + 1] = g[0][2];
                          so[3 * ST_PIECE] = g[1][0], so[3 * ST_PIECE + 1] = g[1][1];
                          so[4 * ST_PIECE] = g[1][2], so[4 * ST_PIECE + 1] = g[2][0];
                          so[5 * ST_PIECE] = g[2][1], so[5 * ST_PIECE + 1] = g[2][2];
                        }
                      else
                        so[ST_PIECE] = u[2], so[ST_PIECE + 1] = div;
                      if constexpr (VARCO) // the coefficients of the point ride along (before the next point's overwrite st)
                        {
                          so[NPO * ST_PIECE] = st[NSTL], so[NPO * ST_PIECE + 1] = st[NSTL + 1];
                          so[(NPO + 1) * ST_PIECE] = st[NSTL + 2], so[(NPO + 1) * ST_PIECE + 1] = st[NSTL + 3];
                        }
                    }
                }
              if (NST > 0 && i + 1 < N && !(HOX_EXP & 4) && !RING)
                {
                  // the state registers are free now: fetch the next point's state (not earlier)
                  unsigned off = st_lane + (unsigned)((i + 1) * ST_POINT);
                  pin_after(off, conv[2]);
                  load_state(stc, off);
                }
              const double diag = cst[TB::C_TGD] * div - PQ[i];
#pragma unroll
              for (int d = 0; d < 3; ++d)
                {
                  G[d][0][i] = conv[d];
#pragma unroll
                  for (int e = 0; e < 3; ++e) // :859-892 row d of tmu (grad u + grad u^T) + (tau_gd div - p) I, times JxW J^{-1}
                    G[d][1 + e][i] = (tmu_q * (g[d][e] + g[e][d]) + (d == e ? diag : 0.)) * (jxw * cst[TB::C_IH + e]);
                }
              PQ[i] = -div * jxw; // :853-856
              __builtin_amdgcn_sched_barrier(0); // one point at a time: interleaved, the five points' temporaries add up
            });
          HOX_MARK(3)
          // ================= integrate (:897-907): the transposed chain =========================================
          double R[3][N], Rp[NP];
          auto integ_single = [&](auto d_) {
            constexpr int d = decltype(d_)::value;
            double        W[N], ln[N], l2[N];
            EoMat<N, N, 1>  mS;
            EoMat<N, N, -1> mD;
            if constexpr (FUSED)
              {
                // z in registers: S^T tv + (D S)^T tz, S^T tx, S^T ty; y: S^T . + (D S)^T ., S^T .; x: S^T . + (D S)^T .
                mS.load(tb(TB::ST));
                mD.load(tb(TB::DST));
                mS.template apply<false>(G[d][0], W);
                mD.template apply<true>(G[d][3], W);
                mS.template apply<false>(G[d][1], ln);
                mS.template apply<false>(G[d][2], l2);
                wr_line<0, NN, N>(pz, W);
                wr_line<BUF, NN, N>(pz, ln);
                wr_line<2 * BUF, NN, N>(pz, l2);
                wave_sync();
                mS.load(tb(TB::ST));
                mD.load(tb(TB::DST));
                rd_line<0, N, N>(ay, W);
                rd_line<2 * BUF, N, N>(ay, l2);
                rd_line<BUF, N, N>(ay, ln);
                ds_wait<N>(W);
                ds_wait<N>(l2);
                double B1[N], B2[N];
                mS.template apply<false>(W, B1);
                mD.template apply<true>(l2, B1);
                ds_wait<0>(ln);
                mS.template apply<false>(ln, B2);
                wave_sync();
                wr_line<0, N, N>(py, B1);
                wr_line<BUF, N, N>(py, B2);
                wave_sync();
                mS.load(tb(TB::ST));
                mD.load(tb(TB::DST));
                rd_line<0, 1, N>(ax, W);
                rd_line<BUF, 1, N>(ax, ln);
                ds_wait<N>(W);
                mS.template apply<false>(W, R[d]);
                ds_wait<0>(ln);
                mD.template apply<true>(ln, R[d]);
                wave_sync();
                return;
              }
            mD.load(tb(TB::DT));
            mD.template apply<true>(G[d][1], G[d][0]); // W = tested value + D^T (x) in registers
            wr_line<0, 1, N>(px, G[d][0]);
            wr_line<BUF, 1, N>(px, G[d][2]);
            wr_line<2 * BUF, 1, N>(px, G[d][3]);
            wave_sync();
            mD.load(tb(TB::DT));
            rd_line<0, N, N>(ay, W);
            rd_line<BUF, N, N>(ay, ln);
            ds_wait<N>(W);
            ds_wait<0>(ln);
            mD.template apply<true>(ln, W); // + D^T (y)
            wave_sync();
            wr_line<0, N, N>(py, W);
            wave_sync();
            mD.load(tb(TB::DT));
            mS.load(tb(TB::ST));
            rd_line<0, NN, N>(az, W);
            rd_line<2 * BUF, NN, N>(az, ln);
            ds_wait<N>(W);
            ds_wait<0>(ln);
            mD.template apply<true>(ln, W);   // + D^T (z)
            mS.template apply<false>(W, l2); // S^T (z): Gauss points -> nodes
            wave_sync();
            wr_line<BUF, NN, N>(pz, l2);
            wave_sync();
            mS.load(tb(TB::ST));
            rd_line<BUF, N, N>(ay, ln);
            ds_wait<0>(ln);
            mS.template apply<false>(ln, l2); // S^T (y)
            wave_sync();
            wr_line<0, N, N>(py, l2);
            wave_sync();
            mS.load(tb(TB::ST));
            rd_line<0, 1, N>(ax, ln);
            ds_wait<0>(ln);
            mS.template apply<false>(ln, R[d]); // S^T (x)
            wave_sync();
          };
          auto integ_p = [&]() {
            double ln[N], T[NP];
            EoMat<NP, N, 1> mP;
            if constexpr (FUSED) // (the tested values are on z-lines already)
              {
                mP.load(tb(TB::SPT));
                mP.template apply<false>(PQ, T);
              }
            else
              {
                wr_line<0, 1, N>(px, PQ);
                wave_sync();
                mP.load(tb(TB::SPT));
                rd_line<0, NN, N>(az, ln);
                ds_wait<0>(ln);
                mP.template apply<false>(ln, T); // z: [N][N][NP]
                wave_sync();
              }
            wr_line<BUF, NN, NP>(pz, T);
            wave_sync();
            mP.load(tb(TB::SPT));
            rd_line<BUF, N, N>(ay, ln); // y-line (a, ., b), b < NP
            ds_wait<0>(ln);
            mP.template apply<false>(ln, T);
            wave_sync();
            wr_line<0, N, NP>(py, T);
            wave_sync();
            mP.load(tb(TB::SPT));
            rd_line<0, 1, N>(ax, ln); // x-line (., a, b), a, b < NP
            ds_wait<0>(ln);
            mP.template apply<false>(ln, Rp);
            wave_sync();
          };
#pragma unroll
          for (int i = 0; i < NP; ++i)
            Rp[i] = 0.;
          integ_single(I0_{});
          // the nodal lines of the next step arrive during the rest of the integration (issued here, not earlier: the
          // quadrature loop and the first component need the registers; k = 5, HOX_NODES_LATE: one component later still)
          if (!(HOX_NODES_LATE && K == 5))
            load_nodes(cx + 1);
          integ_single(I1_{});
          if (HOX_NODES_LATE && K == 5)
            load_nodes(cx + 1);
          integ_single(I2_{});
          HOX_MARK(4)
          if constexpr (WITH_P)
            integ_p();
          if constexpr (LATE_RING)
            {
              // the first two points of the next cell (both halves of the ring were free since the quadrature loop)
              ring_issue(stn, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
              ring_issue(stn, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            }

          HOX_MARK(5)
          // ================= carry in x, combine in y / z, emit the K finished nodes ============================
#pragma unroll
          for (int d = 0; d < 3; ++d)
            {
              R[d][0] += carry[d];
              carry[d] = R[d][K];
            }
          Rp[0] += carry_p;
          carry_p = Rp[KP];
          combine(std::integral_constant<int, K>{}, std::integral_constant<int, KP>{}, R, Rp, fl, lane, step & 1, K * cx,
                  KP * cx, K * step, KP * step, false);
          HOX_MARK(8)
#if HOX_STAMP
          acc[9] += 1;
#endif
        }
#if HOX_STAMP
      if (A.stamps && lane == 0)
        for (int j = 0; j < 10; ++j)
          A.stamps[((size_t)wg * 4 + wave) * 10 + j] = acc[j];
#endif
      if (RING)
        wait_vmcnt<0>(); // no copy may land in LDS after the wave has left
      // ---- the last node plane of the chunk ------------------------------------------------------------------
      {
        double R[3][N], Rp[NP];
#pragma unroll
        for (int d = 0; d < 3; ++d)
          R[d][0] = carry[d];
        Rp[0] = carry_p;
        combine(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, R, Rp, flags, lane, ns & 1, K * (cx0 + ns),
                KP * (cx0 + ns), K * ns, KP * ns, true);
      }
    }

    // ---- second pass: the owner of a shared node (low rim in every direction) adds the partial sums of the
    // other sharers in a fixed order --------------------------------------------------------------------------
    __device__ __forceinline__ bool hox_fix_skip(const HXArgs &A, const int I, const int J, const int Kz, const int nn_x,
                                                 const int nn_y, const int nn_z)
    {
      if (A.fix_mode == 0)
        return false;
      const bool on = (I == 0 && (A.iface & 1u)) || (I == nn_x - 1 && (A.iface & 2u)) || (J == 0 && (A.iface & 4u)) ||
                      (J == nn_y - 1 && (A.iface & 8u)) || (Kz == 0 && (A.iface & 16u)) ||
                      (Kz == nn_z - 1 && (A.iface & 32u));
      return A.fix_mode == 1 ? !on : on;
    }

    // one shared entry (I, J, Kz, comp) of a space with DEG nodes per cell direction and NC components
    template <int DEG, int CY, int CZ, int NC>
    __device__ __forceinline__ void hox_fix_entry(const HXArgs &A, const int I, const int J, const int Kz, const int comp,
                                                  double *dst, const double *slab, const double *xslab, const int nn_x,
                                                  const int nn_y, const int nn_z, const uint32_t con)
    {
      constexpr int TY = DEG * CY + 1, TZ = DEG * CZ + 1, RIM = TY + TZ - 1;
      const int     px = DEG * A.LX, py = DEG * CY, pz = DEG * CZ;
      const bool    sx = I > 0 && I < nn_x - 1 && I % px == 0, sy = J > 0 && J < nn_y - 1 && J % py == 0,
                 sz = Kz > 0 && Kz < nn_z - 1 && Kz % pz == 0;
      if (!(sx || sy || sz))
        return;
      if (on_constrained_face(I, J, Kz, nn_x, nn_y, nn_z, con, NC == 1 ? 1 : 3, comp) ||
          hox_fix_skip(A, I, J, Kz, nn_x, nn_y, nn_z))
        return;
      const int bx = min(I / px, A.n_chunks - 1), by = min(J / py, A.tiles_y - 1), bz = min(Kz / pz, A.tiles_z - 1);
      double    sum = 0.;
      for (int dz = 0; dz <= (sz ? 1 : 0); ++dz)
        for (int dy = 0; dy <= (sy ? 1 : 0); ++dy)
          for (int dx = 0; dx <= (sx ? 1 : 0); ++dx)
            {
              if (dx == 0 && dy == 0 && dz == 0)
                continue;
              const size_t tb = ((size_t)(bz - dz) * A.tiles_y + (by - dy)) * A.n_chunks + (bx - dx);
              const int    jl = dy ? TY - 1 : J - py * by, kl = dz ? TZ - 1 : Kz - pz * bz;
              if (dx)
                sum += xslab[((tb * TZ + kl) * TY + jl) * NC + comp];
              else
                sum += slab[((tb * RIM + rim_line<TY, TZ>(jl, kl)) * (px + 1) + (I - px * bx)) * NC + comp];
            }
      dst[(((size_t)Kz * nn_y + J) * nn_x + I) * NC + comp] += sum;
    }

    // work of the fix-up pass: per space (velocity, then pressure) one BLOCK ITEM per seam row -- rows on y seams, then
    // rows on z seams (those also on a y seam are done there) -- and one per 256 entries of the x-seam planes.  The row of
    // a block item is block-uniform (scalar divisions only); a thread walks the entries (I, comp) of the row
    inline long hox_fix_blocks(const HXArgs &A, const bool with_p)
    {
      long blocks = 0;
      for (int space = 0; space < (with_p ? 2 : 1); ++space)
        {
          const long ny = space ? A.npy : A.nny, nz = space ? A.npz : A.nnz, nc = space ? 1 : 3;
          blocks += (long)(A.tiles_y - 1) * nz + (long)(A.tiles_z - 1) * ny + ((long)(A.n_chunks - 1) * ny * nz * nc + 255) / 256;
        }
      return blocks;
    }

    // (G: the tile geometry -- Geo<K> for this kernel, hop::PGeo for the plane-per-lane kernel of ns_hop_kernel.hpp)
    template <int K, int SPACE, class G = Geo<K>>
    __device__ __forceinline__ void hox_fix_block(const HXArgs &A, long bb, bool &done)
    {
      constexpr int  DEG = SPACE == 0 ? K : K - 1, NC = SPACE == 0 ? 3 : 1;
      const int      nn_x = SPACE == 0 ? A.nnx : A.npx, nn_y = SPACE == 0 ? A.nny : A.npy, nn_z = SPACE == 0 ? A.nnz : A.npz;
      const long     n_y = (long)(A.tiles_y - 1) * nn_z, n_z = (long)(A.tiles_z - 1) * nn_y;
      const long     n_xpl = (long)(A.n_chunks - 1) * nn_y * nn_z * NC, n_xb = (n_xpl + 255) / 256;
      double        *dst = SPACE == 0 ? A.dst_u : A.dst_p;
      const double  *slab = SPACE == 0 ? A.slab_u : A.slab_p, *xslab = SPACE == 0 ? A.xslab_u : A.xslab_p;
      const uint32_t con = SPACE == 0 ? A.con_u : A.con_p;
      done               = true;
      if (bb < n_y + n_z)
        {
          int J, Kz;
          if (bb < n_y)
            {
              J  = (int)(bb / nn_z + 1) * DEG * G::CY;
              Kz = (int)(bb % nn_z);
            }
          else
            {
              Kz = (int)((bb - n_y) / nn_y + 1) * DEG * G::CZ;
              J  = (int)((bb - n_y) % nn_y);
              if (J > 0 && J < nn_y - 1 && J % (DEG * G::CY) == 0)
                return; // on a y seam: done there
            }
          for (int e = threadIdx.x; e < nn_x * NC; e += blockDim.x)
            hox_fix_entry<DEG, G::CY, G::CZ, NC>(A, e / NC, J, Kz, e % NC, dst, slab, xslab, nn_x, nn_y, nn_z, con);
          return;
        }
      bb -= n_y + n_z;
      if (bb < n_xb)
        {
          long e = bb * 256 + threadIdx.x;
          if (e < n_xpl)
            {
              const int comp = (int)(e % NC);
              e /= NC;
              const int J = (int)(e % nn_y);
              e /= nn_y;
              const int Kz = (int)(e % nn_z), I = (int)(e / nn_z + 1) * DEG * A.LX;
              if (!((J > 0 && J < nn_y - 1 && J % (DEG * G::CY) == 0) || (Kz > 0 && Kz < nn_z - 1 && Kz % (DEG * G::CZ) == 0)))
                hox_fix_entry<DEG, G::CY, G::CZ, NC>(A, I, J, Kz, comp, dst, slab, xslab, nn_x, nn_y, nn_z, con);
            }
          return;
        }
      done = false;
    }

    template <int K, class G = Geo<K>>
    __global__ __launch_bounds__(256) void ns_hox_fixup_kernel(const HXArgs A, const int with_p)
    {
      long total = 0, first_p = 0;
      {
        const long nu_ = (long)(A.tiles_y - 1) * A.nnz + (long)(A.tiles_z - 1) * A.nny + ((long)(A.n_chunks - 1) * A.nny * A.nnz * 3 + 255) / 256;
        const long np_ = (long)(A.tiles_y - 1) * A.npz + (long)(A.tiles_z - 1) * A.npy + ((long)(A.n_chunks - 1) * A.npy * A.npz + 255) / 256;
        first_p        = nu_;
        total          = nu_ + (with_p ? np_ : 0);
      }
      for (long bb = blockIdx.x; bb < total; bb += gridDim.x)
        {
          bool done;
          if (bb < first_p)
            hox_fix_block<K, 0, G>(A, bb, done);
          else
            hox_fix_block<K, 1, G>(A, bb - first_p, done);
        }
    }

    // generic state [cell][12][N^3] (q = (k N + j) N + i) -> streaming layout of ns_hox_kernel; one thread per
    // 16-byte piece of the output
    // rho / mu / damp (generic [cell][q], all three or none): two more pieces per point, (rho, mu) and (damping, 0),
    // behind the npl pieces of the linearisation state (npc = npl + 2 then)
    template <int K>
    __global__ __launch_bounds__(256) void hox_convert_state_kernel(double *out, const double *generic, const int ncx,
                                                                    const int ncy, const int ncz, const int ngy,
                                                                    const int ngz, const int npc, const int npl = -1,
                                                                    const double *rho = nullptr, const double *mu = nullptr,
                                                                    const double *damp = nullptr)
    {
      using G          = Geo<K>;
      constexpr int N = G::N, NL = G::NL, N3 = G::N3, CPW = G::CPW;
      const long    total = (long)ngz * ngy * ncx * N * npc * CPW * NL;
      for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x)
        {
          long      r = it;
          const int l = (int)(r % NL);
          r /= NL;
          const int scw = (int)(r % CPW);
          r /= CPW;
          const int piece = (int)(r % npc);
          r /= npc;
          const int i = (int)(r % N);
          r /= N;
          const int cx = (int)(r % ncx);
          r /= ncx;
          const int gy = (int)(r % ngy), gz = (int)(r / ngy);
          const int cy = gy * G::CWY + scw % G::CWY, cz = gz * G::CWZ + scw / G::CWY;
          double    v0 = 0., v1 = 0.;
          if (cy < ncy && cz < ncz)
            {
              const size_t cellg = ((size_t)cz * ncy + cy) * ncx + cx;
              // x-lines: line l = (j, k) = (l % N, l / N), point i along x; HOX_FUSED: z-lines, l = (x, y), i along z
              const int    q     = HOX_FUSED ? (i * N + l / N) * N + l % N : ((l / N) * N + l % N) * N + i;
              if (npl < 0 || piece < npl)
                {
                  v0 = generic[(cellg * NLIN_ + 2 * piece) * N3 + q];
                  v1 = generic[(cellg * NLIN_ + 2 * piece + 1) * N3 + q];
                }
              else if (piece == npl)
                {
                  v0 = rho[cellg * N3 + q];
                  v1 = mu[cellg * N3 + q];
                }
              else
                v0 = damp[cellg * N3 + q];
            }
          out[2 * it]     = v0;
          out[2 * it + 1] = v1;
        }
    }
    // ... and back (the residual mode leaves the state in the streaming layout only; the generic copy is rebuilt on
    // demand: adaflo_ns_get_linearization, generic kernels, the velocity-block diagonal)
    template <int K>
    __global__ __launch_bounds__(256) void hox_unconvert_state_kernel(double *generic, const double *in, const int ncx,
                                                                      const int ncy, const int ncz, const int ngy,
                                                                      const int ngz, const int npc, const int npl)
    {
      using G          = Geo<K>;
      constexpr int N = G::N, NL = G::NL, N3 = G::N3, CPW = G::CPW;
      const long    total = (long)ngz * ngy * ncx * N * npc * CPW * NL;
      for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x)
        {
          long      r = it;
          const int l = (int)(r % NL);
          r /= NL;
          const int scw = (int)(r % CPW);
          r /= CPW;
          const int piece = (int)(r % npc);
          r /= npc;
          const int i = (int)(r % N);
          r /= N;
          const int cx = (int)(r % ncx);
          r /= ncx;
          const int gy = (int)(r % ngy), gz = (int)(r / ngy);
          const int cy = gy * G::CWY + scw % G::CWY, cz = gz * G::CWZ + scw / G::CWY;
          if (cy < ncy && cz < ncz && piece < npl) // (pieces npl, npl + 1 of a variable-coefficient state: rho, mu, damping)
            {
              const size_t cellg = ((size_t)cz * ncy + cy) * ncx + cx;
              const int    q     = HOX_FUSED ? (i * N + l / N) * N + l % N : ((l / N) * N + l % N) * N + i;
              generic[(cellg * NLIN_ + 2 * piece) * N3 + q]     = in[2 * it];
              generic[(cellg * NLIN_ + 2 * piece + 1) * N3 + q] = in[2 * it + 1];
            }
        }
    }
  } // namespace hox
} // namespace adaflo_hip
