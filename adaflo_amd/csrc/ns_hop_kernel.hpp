// ns_hop_kernel.hpp -- device source of the PLANE-PER-LANE Taylor-Hood Q4/Q3 kernel (round 5):
// NavierStokesMatrix::vmult / velocity_vmult with constant coefficients, the p = 3 instance of local_operation
// (source/navier_stokes_matrix.cc:64-82, 601-916).
//
// Decomposition.  The x-marching kernel of round 4 (ns_hox_kernel.hpp) gives a lane one LINE
// and pays six LDS transpositions per component and phase; its waves wait on those round trips (VALU 29 %, LDS 33 %
// busy).  Here
//   * a lane owns one x-PLANE (5 x 5 values in y, z) of ONE velocity component of ONE cell: the y and z contractions
//     (interpolation to the Gauss points, collocation derivatives, and their transposes) are register-only, 25
//     independent values per lane -- nothing the wave has to wait for;
//   * a cell is one DPP row: lane = 16 cell + 5 component + x (lane 15 of a row idle), a wave = 2 x 2 cells in (y, z).
//     Only the x contraction crosses lanes; it goes through a wave-private LDS line buffer (the LDS operations of one
//     wave execute in order: no s_barrier anywhere in this kernel).  gfx950 has no cheaper way: 64-bit DPP only knows
//     row_newbcast (3 x the multiply-adds for three 5-lane groups per row), 32-bit row shifts need two moves per
//     dword and source lane of a 5-lane group;
//   * the pressure (4 x 4 x 4 nodes) rides on the lanes of component 0: values only, same instruction stream;
//   * the coupling between the components at a quadrature point (u, the transposed gradient, div u) is one 32-byte
//     record per lane in LDS, read back with per-lane offsets -- no selects;
//   * the wave MARCHES ALONG X: plane x = 4 of a cell is plane 0 of the next one, handed from lane 4 to lane 0
//     through 2.4 KB of LDS; faces between the four cells of the wave are summed in registers with
//     v_permlane16_swap / v_permlane32_swap (a cell = a row); faces between waves are seams: whole x-rows to slabs, the
//     fix-up kernel of ns_hox_kernel.hpp adds them (no atomics, bitwise reproducible).  Waves never talk to each
//     other: a workgroup is just NW independent waves;
//   * the linearisation state is streamed by LDS-DMA in a layout of its own, [tile][cx][point][piece][lane][2]: piece A
//     of component d = (u_d, d_0 u_d), piece B = (d_1 u_d, d_2 u_d); the other components' values and the trace are
//     read from the same ring slots.
//
// Written against the primitives of hox_intrin.hpp; includes nothing itself so that tests/emu can run it on the host.
#pragma once

namespace adaflo_hip
{
  namespace hop
  {
    using hox::EoMat;
    using hox::eo_size;
    using hox::eo_table;
    using hox::HXArgs;

#ifndef HOP_NW
#define HOP_NW 2
#endif
    // HOP_EXP (diagnostic builds, WRONG results): & 4 no state stream, & 16 no stores
#ifndef HOP_EXP
#define HOP_EXP 0
#endif
    // waves per SIMD the register allocation is made for (2: 256 registers per lane; 1: 512)
    // HOP_STAMP: s_memtime stamps at the phase boundaries of a step, summed per wave into HXArgs::stamps ([tile][10]
    // cycles: top of step, evaluate u, evaluate p, quadrature loop, integrate u, integrate p, carry, emit + node loads, -,
    // steps) -- the launcher prints the medians (development aid)
#ifndef HOP_STAMP
#define HOP_STAMP 0
#endif
    // HOP_SCHED: 1 = scheduling barriers at the exchanges and after every quadrature point (phases stay phases: fewer
    // registers), 0 = the compiler may interleave neighbouring lines / points
#ifndef HOP_SCHED
#define HOP_SCHED 1
#endif
#if HOP_SCHED
#define HOP_XSYNC wave_sync
#else
#define HOP_XSYNC wave_fence
#endif
#ifndef HOP_PREFETCH_NODES
#define HOP_PREFETCH_NODES 0
#endif
#ifndef HOP_LB
#define HOP_LB 2
#endif
    constexpr int K = 4, N = 5, NP = 4, KP = 3, NQ = 25;
    constexpr int NW = HOP_NW, NTH = 64 * NW;

    // tile of one wave for the seam slabs and the fix-up pass (the role Geo<K> plays for ns_hox_kernel)
    struct PGeo
    {
      static constexpr int CY = 2, CZ = 2, TNY = K * CY + 1, TNZ = K * CZ + 1, TPY = KP * CY + 1, TPZ = KP * CZ + 1;
      static constexpr int RIMU = TNY + TNZ - 1, RIMP = TPY + TPZ - 1;
    };

    // LDS of one wave (bytes)
    constexpr int RING_SLOTS = 8, SLOT_B = 1024, RING_B = RING_SLOTS * SLOT_B; // 4 points x 2 pieces in flight
    constexpr int XB_B   = N * 64 * 8;                                        // one line (5 values) of every lane
    constexpr int PQ_B   = 4 * N * NQ * 8;                                    // p at the quadrature points [cell][x][point]
    constexpr int XC_B   = 64 * 32;                                           // (u, grad u) of a point, every lane
    constexpr int CRU_B  = 12 * NQ * 8, CRP_B = 4 * NP * NP * 8;              // x-carry: plane 4 -> plane 0 of the next cell
    constexpr int DUMMY_B = 64 * 8; // where the lanes that have nothing to store put their value (no exec-masked LDS stores)
    constexpr int O_RING = 0, O_XB = O_RING + RING_B, O_PQ = O_XB + XB_B, O_XC = O_PQ + PQ_B, O_CRU = O_XC + XC_B,
                  O_CRP = O_CRU + CRU_B, O_DUMMY = O_CRP + CRP_B, WAVE_B = O_DUMMY + DUMMY_B;
    constexpr int LDS_BYTES = NW * WAVE_B;
    static_assert(WAVE_B % 16 == 0, "alignment of the per-wave regions");

    // number of 16-byte pieces of state per lane and quadrature point
    constexpr int npc_of(const int lin_mode)
    {
      return lin_mode == 0 ? 2 : (lin_mode == 1 ? 1 : 0);
    }
    // doubles of streamed state per (tile, cell of the march): 25 points x pieces x 60 lanes x 2
    constexpr size_t state_cell_doubles(const int lin_mode)
    {
      return (size_t)NQ * npc_of(lin_mode) * 120;
    }

    // table of a launch (doubles): even / odd 1D matrices for the register contractions, the full collocation
    // derivative, constants of the quadrature-point operation, then per x (= lane within its group) the rows and columns
    // of the x matrices
    struct Tab
    {
      static constexpr int S = 0, ST = S + eo_size(N, N), SP = ST + eo_size(N, N), SPT = SP + eo_size(N, NP),
                           D = SPT + eo_size(NP, N), C = D + N * N;
      static constexpr int C_W = 0, C_IH = N, C_DET = N + 3, C_CA = N + 4, C_CB = N + 5, C_BETA = N + 6, C_TGD = N + 7,
                           C_TMU = N + 8, NC = N + 9;
      static constexpr int XT = C + 16, XSTRIDE = 32; // per x: the offsets below
      static constexpr int X_S = 0,    // S[x][m]: u(q = x) = sum_m S[x][m] W_m
        X_DS  = 5,                     // (D S)[x][m] / h_x
        X_SC  = 10,                    // S[q][x]: transposed interpolation
        X_DSC = 15,                    // (D S)[q][x]
        X_P   = 20,                    // Sp[x][m], m < 4
        X_PC  = 24,                    // Sp[q][x] (0 for x = 4)
        X_WXD = 29;                    // w[x] det J
      static constexpr int SIZE = XT + N * XSTRIDE;
    };
    // (host) S[q][i] nodal -> Gauss points (N x N), Dc collocation derivative (N x N), Sp pressure (N x NP)
    inline std::vector<double> hop_table(const double *S, const double *Dc, const double *Sp, const double *w,
                                         const double h[3], const double cA, const double cB, const double beta,
                                         const double tau_gd, const double tmu)
    {
      std::vector<double> t;
      eo_table(t, S, N, N, false);
      eo_table(t, S, N, N, true);
      eo_table(t, Sp, N, NP, false);
      eo_table(t, Sp, N, NP, true);
      for (int i = 0; i < N * N; ++i)
        t.push_back(Dc[i]);
      for (int q = 0; q < N; ++q)
        t.push_back(w[q]);
      for (int e = 0; e < 3; ++e)
        t.push_back(1. / h[e]);
      const double det = h[0] * h[1] * h[2];
      t.push_back(det);
      t.push_back(cA);
      t.push_back(cB);
      t.push_back(beta);
      t.push_back(tau_gd);
      t.push_back(tmu);
      t.resize(Tab::XT, 0.);
      double DS[N][N];
      for (int q = 0; q < N; ++q)
        for (int m = 0; m < N; ++m)
          {
            DS[q][m] = 0.;
            for (int r = 0; r < N; ++r)
              DS[q][m] += Dc[q * N + r] * S[r * N + m];
          }
      for (int x = 0; x < N; ++x)
        {
          std::vector<double> r(Tab::XSTRIDE, 0.);
          for (int m = 0; m < N; ++m)
            {
              r[Tab::X_S + m]   = S[x * N + m];
              r[Tab::X_DS + m]  = DS[x][m] / h[0];
              r[Tab::X_SC + m]  = S[m * N + x];
              r[Tab::X_DSC + m] = DS[m][x];
              r[Tab::X_PC + m]  = x < NP ? Sp[m * NP + x] : 0.;
            }
          for (int m = 0; m < NP; ++m)
            r[Tab::X_P + m] = Sp[x * NP + m];
          r[Tab::X_WXD] = w[x] * det;
          t.insert(t.end(), r.begin(), r.end());
        }
      return t;
    }

    // mesh-dependent integers of the launch (host): tiles of 2 x 2 cells, chunks of lx cells
    inline void hop_geometry(HXArgs &A, const int ncell[3], const int lx)
    {
      A.ncx      = ncell[0];
      A.ncy      = ncell[1];
      A.ncz      = ncell[2];
      A.nnx      = K * A.ncx + 1;
      A.nny      = K * A.ncy + 1;
      A.nnz      = K * A.ncz + 1;
      A.npx      = KP * A.ncx + 1;
      A.npy      = KP * A.ncy + 1;
      A.npz      = KP * A.ncz + 1;
      A.tiles_y  = (A.ncy + 1) / 2;
      A.tiles_z  = (A.ncz + 1) / 2;
      A.LX       = lx < 1 ? 1 : (lx > A.ncx ? A.ncx : lx);
      A.n_chunks = (A.ncx + A.LX - 1) / A.LX;
      A.ngy      = A.tiles_y;
      A.ngz      = A.tiles_z;
    }

    // generic state [cell][12][125] (q = (zq 5 + yq) 5 + xq) -> streaming layout [tile][cx][point yq + 5 zq][piece]
    // [compact lane 15 cell + 5 d + xq][2]; cells beyond the mesh are zero.  One thread per 16-byte piece.
    __global__ __launch_bounds__(256) void hop_convert_state_kernel(double *out, const double *generic, const int ncx,
                                                                    const int ncy, const int ncz, const int ngy,
                                                                    const int ngz, const int lin_mode)
    {
      const int  npc   = npc_of(lin_mode);
      const long total = (long)ngz * ngy * ncx * NQ * npc * 60;
      for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x)
        {
          long      r  = it;
          const int cl = (int)(r % 60);
          r /= 60;
          const int piece = (int)(r % npc);
          r /= npc;
          const int pt = (int)(r % NQ);
          r /= NQ;
          const int cx = (int)(r % ncx);
          r /= ncx;
          const int gy = (int)(r % ngy), gz = (int)(r / ngy);
          const int c = cl / 15, d = (cl % 15) / 5, xq = cl % 5;
          const int cy = 2 * gy + (c & 1), cz = 2 * gz + (c >> 1);
          double    v0 = 0., v1 = 0.;
          if (cy < ncy && cz < ncz)
            {
              const size_t cellg = ((size_t)cz * ncy + cy) * ncx + cx;
              const int    q     = pt * N + xq; // pt = yq + 5 zq
              const double *g    = generic + cellg * 12 * 125 + q;
              if (lin_mode == 0)
                {
                  v0 = piece == 0 ? g[d * 125] : g[(3 + 3 * d + 1) * 125];
                  v1 = piece == 0 ? g[(3 + 3 * d) * 125] : g[(3 + 3 * d + 2) * 125];
                }
              else
                {
                  v0 = g[d * 125];
                  v1 = g[3 * 125]; // div u_lin
                }
            }
          out[2 * it]     = v0;
          out[2 * it + 1] = v1;
        }
    }

    // ---- the kernel ------------------------------------------------------------------------------------------
    // LIN_MODE 0 / 1 / 2: Newton (u, grad u) / Picard-type (u, div u) / no state (Stokes, explicit velocity)
    template <int LIN_MODE, bool WITH_P>
    __global__ __launch_bounds__(NTH, HOP_LB) void ns_hop_kernel(const HXArgs A)
    {
      using TB            = Tab;
      constexpr int NPC   = (HOP_EXP & 4) ? 0 : npc_of(LIN_MODE);
      constexpr int LA    = 3;                // points of state in flight ahead of the one being read
      constexpr int RSL   = RING_SLOTS / 2;   // points the ring holds
      static_assert(LA + 1 <= RSL && (RSL & (RSL - 1)) == 0 && NQ % RSL == 1, "ring geometry");

      const int tid = threadIdx.x, lane = tid & 63;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int c = lane >> 4, r16 = lane & 15;
      if (r16 == 15)
        return; // the idle lane of every row leaves: the rest of the kernel runs under one exec mask
      const int d = r16 / 5, x = r16 - 5 * d;

      // ---- which tile (2 x 2 cells in y, z) and chunk of the march ------------------------------------------------
      const long ntile = A.wg_list ? (long)A.wg_count : (long)A.tiles_y * A.tiles_z * A.n_chunks;
      long       tile  = xcd_remap(blockIdx.x, gridDim.x) * NW + wave;
      if (tile >= ntile)
        return;
      if (A.wg_list)
        tile = A.wg_list[A.wg_offset + tile];
      const int  bx = (int)(tile % A.n_chunks), bt = (int)(tile / A.n_chunks);
      const int  by = bt % A.tiles_y, bz = bt / A.tiles_y;
      const int  cx0 = bx * A.LX, ns = min(A.LX, A.ncx - cx0);
      const int  tcy = min(2, A.ncy - 2 * by), tcz = min(2, A.ncz - 2 * bz);
      const int  cyl = c & 1, czl = c >> 1;
      const bool cell_ok = cyl < tcy && czl < tcz;
      // cells outside the mesh compute on cell (0, 0) of the tile: nothing of theirs is emitted, addresses stay legal
      const int  cy = 2 * by + (cell_ok ? cyl : 0), cz = 2 * bz + (cell_ok ? czl : 0);
      const bool lasty = cyl == tcy - 1, lastz = czl == tcz - 1;
      const bool x_seam_end = cx0 + ns < A.ncx;

      // ---- LDS regions of my wave ---------------------------------------------------------------------------------
      char *const    wlds    = reinterpret_cast<char *>(dyn_lds()) + wave * WAVE_B;
      const unsigned ring_m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_byte_addr(wlds + O_RING));
      const char    *ring_cx = wlds + O_RING + (16 * c + x) * 16; // slot of component 0 of my (cell, x)
      double *const  xb      = reinterpret_cast<double *>(wlds + O_XB);
      double *const  xb_own  = xb + lane;                         // value v of my line: xb_own[64 v]
      const double  *xb_grp  = xb + 16 * c + 5 * d;               // ... of plane m of my component: xb_grp[64 v + m]
      const double  *xb_p    = xb + 16 * c;                       // ... of pressure plane m (lanes of component 0)
      double *const  pq_cx   = reinterpret_cast<double *>(wlds + O_PQ) + (c * N + x) * NQ;
      const double  *pq_c    = reinterpret_cast<double *>(wlds + O_PQ) + c * N * NQ;
      double *const  xc_own  = reinterpret_cast<double *>(wlds + O_XC) + lane * 4;
      const double  *xc_cx   = reinterpret_cast<double *>(wlds + O_XC) + (16 * c + x) * 4; // component e: + 20 e
      double *const  cru     = reinterpret_cast<double *>(wlds + O_CRU) + (3 * c + d) * NQ;
      double *const  crp     = reinterpret_cast<double *>(wlds + O_CRP) + c * NP * NP;
      // stores that only some lanes have to make are unconditional: the others write to a slot of their own (a branch
      // around an LDS store splits the unrolled quadrature loop into hundreds of basic blocks: thousands of spilled
      // registers in the first build).  Entry n of the target is at tgt + n * stride (stride 0: the dummy slot)
      double *const  dummy   = reinterpret_cast<double *>(wlds + O_DUMMY) + lane;
      double *const  pq_w    = d == 0 ? pq_cx : dummy;
      const int      pq_ws   = d == 0 ? 1 : 0;
      double *const  cru_w   = x == K ? cru : dummy;
      const int      cru_ws  = x == K ? 1 : 0;
      double *const  crp_w   = (d == 0 && x == KP) ? crp : dummy;
      const int      crp_ws  = (d == 0 && x == KP) ? 1 : 0;

      // ---- per-lane constants -------------------------------------------------------------------------------------
      const ctab_t   tab = as_ctab(A.tab);
      const double  *xt  = A.tab + TB::XT + x * TB::XSTRIDE;
      // global rows: wave-uniform base pointer + 32-bit per-lane offset (doubles)
      const unsigned sy = (unsigned)A.nnx * 3, sz = (unsigned)A.nny * sy;
      const unsigned ubase = (unsigned)(K * cz) * sz + (unsigned)(K * cy) * sy + (unsigned)(x * 3 + d);
      const unsigned spy = (unsigned)A.npx, spz = (unsigned)A.npy * spy;
      const int      xp  = min(x, KP); // pressure plane of this lane (lane 4: a copy of plane 3, never used)
      const unsigned pbase = (unsigned)(KP * cz) * spz + (unsigned)(KP * cy) * spy + (unsigned)xp;
      // Dirichlet rows on y / z faces: bit j + 5 k of the velocity plane, bit j + 4 k of the pressure plane
      unsigned zmask = 0, zmask_p = 0;
      {
        const bool ylo = cy == 0 && (A.con_u >> (6 + d) & 1), yhi = cy == A.ncy - 1 && (A.con_u >> (9 + d) & 1);
        const bool zlo = cz == 0 && (A.con_u >> (12 + d) & 1), zhi = cz == A.ncz - 1 && (A.con_u >> (15 + d) & 1);
        for (int k = 0; k < N; ++k)
          for (int j = 0; j < N; ++j)
            if ((j == 0 && ylo) || (j == K && yhi) || (k == 0 && zlo) || (k == K && zhi))
              zmask |= 1u << (j + N * k);
        const bool pylo = cy == 0 && (A.con_p >> 2 & 1), pyhi = cy == A.ncy - 1 && (A.con_p >> 3 & 1);
        const bool pzlo = cz == 0 && (A.con_p >> 4 & 1), pzhi = cz == A.ncz - 1 && (A.con_p >> 5 & 1);
        for (int k = 0; k < NP; ++k)
          for (int j = 0; j < NP; ++j)
            if ((j == 0 && pylo) || (j == KP && pyhi) || (k == 0 && pzlo) || (k == KP && pzhi))
              zmask_p |= 1u << (j + NP * k);
      }
      const bool any_con = A.con_u != 0u || A.con_p != 0u; // (wave-uniform: guards the rare paths)

      // state: wave-uniform base of my tile's row of cells + per-lane byte offset
      constexpr unsigned ST_PIECE = 120, ST_POINT = (NPC > 0 ? NPC : 1) * ST_PIECE, ST_CELL = NQ * ST_POINT; // doubles
      const double *const stg = NPC > 0 ? A.lin + ((size_t)bz * A.ngy + by) * A.ncx * ST_CELL : nullptr;
      const unsigned      st_lane = (unsigned)(lane - c) * 16;                                              // bytes
      // pieces of point PT of the cell at `cell_base` -> the ring slots of that point
      auto ring_issue = [&](const double *const cell_base, const int pt, const int slot_pt) {
        const double *const sb = uniform_ptr(cell_base);
#pragma unroll
        for (int e = 0; e < NPC; ++e)
          dma_b128(sb + (pt * NPC + e) * ST_PIECE, st_lane, ring_m0 + (slot_pt * 2 + e) * SLOT_B);
      };
      if (NPC > 0)
        {
          const double *const c0 = stg + (size_t)cx0 * ST_CELL;
#pragma unroll
          for (int p = 0; p < LA; ++p)
            ring_issue(c0, p, p);
        }

      auto tb = [&](const int off) {
        ctab_t t = tab;
        opaque(t);
        return t + off;
      };

      // ---- nodal planes of a step -----------------------------------------------------------------------------------
      double U[NQ], P[NP * NP];
      // (the per-lane offsets go through opaque copies wherever they are used: everything that does not change from
      // step to step is otherwise computed ahead of the marching loop -- 41 load and 41 store addresses -- and spilled)
      auto   load_nodes = [&](const int cxn) {
        const int     cxc = min(cxn, A.ncx - 1);
        const double *pu  = A.src_u + (size_t)(K * cxc) * 3;
        unsigned      ub  = ubase;
        opaque(ub);
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
          for (int j = 0; j < N; ++j)
            U[j + N * k] = pu[ub + j * sy + k * sz];
        if (WITH_P)
          {
            const double *pp = A.src_p + (size_t)(KP * cxc);
            unsigned      pb = pbase;
            opaque(pb);
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
              for (int j = 0; j < NP; ++j)
                P[j + NP * k] = pp[pb + j * spy + k * spz];
          }
      };

      // ---- sums over the faces inside the wave, emit one node plane -------------------------------------------------
      // Nu: partial sums of velocity node plane I (component d), Npn: of pressure node plane Ip (lanes of component 0).
      // `emit_u` / `emit_p`: this lane's plane is final in x.  xl / xlp: index of the plane inside the chunk
      auto emit = [&](double (&Nu)[NQ], double (&Npn)[NP * NP], const bool emit_u, const bool emit_p, const int I,
                      const int Ip, const int xl, const int xlp, const bool endplane) {
        // y faces: the upper cell of a pair (odd row) owns the shared line; then z faces (rows 2, 3 own)
#pragma unroll
        for (int k = 0; k < N; ++k)
          {
            const double v = from_row_below(Nu[K + N * k]);
            Nu[N * k] += (cyl == 1) ? v : 0.;
          }
#pragma unroll
        for (int j = 0; j < N; ++j)
          {
            const double v = from_half_below(Nu[j + N * K]);
            Nu[j] += (czl == 1) ? v : 0.;
          }
        if (WITH_P)
          {
#pragma unroll
            for (int k = 0; k < NP; ++k)
              {
                const double v = from_row_below(Npn[KP + NP * k]);
                Npn[NP * k] += (cyl == 1) ? v : 0.;
              }
#pragma unroll
            for (int j = 0; j < NP; ++j)
              {
                const double v = from_half_below(Npn[j + NP * KP]);
                Npn[j] += (czl == 1) ? v : 0.;
              }
          }
        if (HOP_EXP & 16)
          {
            sink(Nu[0] + Nu[NQ - 1] + Npn[0]);
            return;
          }
        const bool to_xslab = endplane && x_seam_end;
        unsigned   ub = ubase - (unsigned)(x * 3), pb = pbase - (unsigned)xp; // first entry of my rows, opaque (see load_nodes)
        opaque(ub);
        opaque(pb);
        int cyl_o = cyl, czl_o = czl;
        opaque(cyl_o);
        opaque(czl_o);
        const int  jl0 = K * cyl_o, kl0 = K * czl_o;
        if (emit_u && cell_ok)
          {
            double *const drow = A.dst_u + (size_t)I * 3;
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
              for (int j = 0; j < N; ++j)
                {
                  if ((j == K && !lasty) || (k == K && !lastz))
                    continue; // given to the cell above
                  const int  J = K * cy + j, Kz = K * cz + k;
                  const bool seam = (j == K && J < A.nny - 1) || (k == K && Kz < A.nnz - 1); // high rim of the tile
                  double    *tp;
                  if (to_xslab)
                    tp = A.xslab_u + (((size_t)tile * PGeo::TNZ + (kl0 + k)) * PGeo::TNY + (jl0 + j)) * 3 + d;
                  else if ((j == K || k == K) && seam)
                    tp = A.slab_u +
                         (((size_t)tile * PGeo::RIMU + hox::rim_line<PGeo::TNY, PGeo::TNZ>(jl0 + j, kl0 + k)) * (K * A.LX + 1) + xl) * 3 + d;
                  else
                    tp = drow + (ub + j * sy + k * sz);
                  *tp = Nu[j + N * k];
                }
          }
        if (WITH_P && A.integrate_p && emit_p && cell_ok)
          {
            double *const drow = A.dst_p + (size_t)Ip;
            const int     jp0 = KP * cyl_o, kp0 = KP * czl_o;
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
              for (int j = 0; j < NP; ++j)
                {
                  if ((j == KP && !lasty) || (k == KP && !lastz))
                    continue;
                  const int  J = KP * cy + j, Kz = KP * cz + k;
                  const bool seam = (j == KP && J < A.npy - 1) || (k == KP && Kz < A.npz - 1);
                  double    *tp;
                  if (to_xslab)
                    tp = A.xslab_p + (((size_t)tile * PGeo::TPZ + (kp0 + k)) * PGeo::TPY + (jp0 + j));
                  else if ((j == KP || k == KP) && seam)
                    tp = A.slab_p +
                         (((size_t)tile * PGeo::RIMP + hox::rim_line<PGeo::TPY, PGeo::TPZ>(jp0 + j, kp0 + k)) * (KP * A.LX + 1) + xlp);
                  else
                    tp = drow + (pb + j * spy + k * spz);
                  *tp = Npn[j + NP * k];
                }
          }
        // constrained rows are then set to +-src (:247-256): a later store of the same lane to the same address, or an
        // entry the fix-up kernel skips (domain boundary only)
        if (any_con && !to_xslab)
          {
            const bool xcon_u = (I == 0 && (A.con_u >> d & 1)) || (I == A.nnx - 1 && (A.con_u >> (3 + d) & 1));
            if (emit_u && cell_ok && (zmask != 0u || xcon_u))
              {
#pragma unroll 1
                for (int n = 0; n < NQ; ++n)
                  {
                    const int j = n % N, k = n / N;
                    if ((j == K && !lasty) || (k == K && !lastz))
                      continue;
                    const int  J = K * cy + j, Kz = K * cz + k;
                    const bool seam = (j == K && J < A.nny - 1) || (k == K && Kz < A.nnz - 1);
                    if (seam || !(xcon_u || (zmask >> n & 1)))
                      continue;
                    const size_t po = (size_t)I * 3 + (ub + j * sy + k * sz);
                    A.dst_u[po]     = A.src_u[po];
                  }
              }
            const bool xcon_p = (Ip == 0 && (A.con_p & 1)) || (Ip == A.npx - 1 && (A.con_p >> 1 & 1));
            if (WITH_P && A.integrate_p && emit_p && cell_ok && (zmask_p != 0u || xcon_p))
              {
#pragma unroll 1
                for (int n = 0; n < NP * NP; ++n)
                  {
                    const int j = n % NP, k = n / NP;
                    if ((j == KP && !lasty) || (k == KP && !lastz))
                      continue;
                    const int  J = KP * cy + j, Kz = KP * cz + k;
                    const bool seam = (j == KP && J < A.npy - 1) || (k == KP && Kz < A.npz - 1);
                    if (seam || !(xcon_p || (zmask_p >> n & 1)))
                      continue;
                    const size_t po = (size_t)Ip + (pb + j * spy + k * spz);
                    A.dst_p[po]     = -A.src_p[po];
                  }
              }
          }
      };

      load_nodes(cx0);
#if HOP_STAMP
      unsigned long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock_now();
#define HOP_MARK(j)                            \
  {                                            \
    const unsigned long long tn = clock_now(); \
    acc[j] += tn - tlast;                      \
    tlast = tn;                                \
  }
#else
#define HOP_MARK(j)
#endif

#pragma unroll 1
      for (int step = 0; step < ns; ++step)
        {
          const int           cx  = cx0 + step;
          const double *const stc = NPC > 0 ? stg + (size_t)cx * ST_CELL : nullptr;
          const double *const stn = NPC > 0 ? stg + (size_t)min(cx + 1, cx0 + ns - 1) * ST_CELL : nullptr;
          // ring slots are numbered along the whole march: point p of this step lives in slot (25 step + p) mod 4, and
          // 25 = 1 mod 4
          const int rot = step & (RSL - 1);

          // ================= evaluate (FEEvaluation::evaluate, :668-671) ===========================================
          // read_dof_values: constrained entries read as zero
          if (any_con)
            {
              const bool xz = (cx == 0 && x == 0 && (A.con_u >> d & 1)) || (cx == A.ncx - 1 && x == K && (A.con_u >> (3 + d) & 1));
              if (zmask != 0u || xz)
                {
#pragma unroll
                  for (int n = 0; n < NQ; ++n)
                    U[n] = (xz || (zmask >> n & 1)) ? 0. : U[n];
                }
              if (WITH_P)
                {
                  const bool xzp = (cx == 0 && xp == 0 && (A.con_p & 1)) || (cx == A.ncx - 1 && xp == KP && (A.con_p >> 1 & 1));
                  if (zmask_p != 0u || xzp)
                    {
#pragma unroll
                      for (int n = 0; n < NP * NP; ++n)
                        P[n] = (xzp || (zmask_p >> n & 1)) ? 0. : P[n];
                    }
                }
            }
          HOP_MARK(0)
          // velocity: y and z in registers (values at (x node, yq, zq))
          {
            EoMat<N, N, 1> mS;
            mS.load(tb(TB::S));
#pragma unroll
            for (int k = 0; k < N; ++k)
              {
                double in[N], out[N];
#pragma unroll
                for (int j = 0; j < N; ++j)
                  in[j] = U[j + N * k];
                mS.template apply<false>(in, out);
#pragma unroll
                for (int j = 0; j < N; ++j)
                  U[j + N * k] = out[j];
              }
#pragma unroll
            for (int j = 0; j < N; ++j)
              {
                double in[N], out[N];
#pragma unroll
                for (int k = 0; k < N; ++k)
                  in[k] = U[j + N * k];
                mS.template apply<false>(in, out);
#pragma unroll
                for (int k = 0; k < N; ++k)
                  U[j + N * k] = out[k];
              }
          }
          // x across the lanes of my group, line by line: u and d/dx at my quadrature points
          double ux[NQ];
          {
            double xs[N], xds[N];
            {
              const double *t = xt;
              opaque_ptr(t);
#pragma unroll
              for (int m = 0; m < N; ++m)
                xs[m] = t[TB::X_S + m], xds[m] = t[TB::X_DS + m];
            }
#pragma unroll
            for (int k = 0; k < N; ++k)
              {
                emu_sync(); // (the readers of the previous line are done)
#pragma unroll
                for (int j = 0; j < N; ++j)
                  xb_own[64 * j] = U[j + N * k];
                HOP_XSYNC();
#pragma unroll
                for (int j = 0; j < N; ++j)
                  {
                    double u = 0., g = 0.;
#pragma unroll
                    for (int m = 0; m < N; ++m)
                      {
                        const double w = xb_grp[64 * j + m];
                        u += xs[m] * w;
                        g += xds[m] * w;
                      }
                    U[j + N * k]  = u;
                    ux[j + N * k] = g;
                  }
              }
          }
          HOP_MARK(1)
          // pressure: y, z in registers, x across the lanes of component 0; p at the quadrature points -> LDS
          if constexpr (WITH_P)
            {
              double          PW[NQ];
              EoMat<N, NP, 1> mP;
              mP.load(tb(TB::SP));
              {
                double T1[N * NP]; // [yq][k]
#pragma unroll
                for (int k = 0; k < NP; ++k)
                  {
                    double in[NP], out[N];
#pragma unroll
                    for (int j = 0; j < NP; ++j)
                      in[j] = P[j + NP * k];
                    mP.template apply<false>(in, out);
#pragma unroll
                    for (int j = 0; j < N; ++j)
                      T1[j + N * k] = out[j];
                  }
#pragma unroll
                for (int j = 0; j < N; ++j)
                  {
                    double in[NP], out[N];
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                      in[k] = T1[j + N * k];
                    mP.template apply<false>(in, out);
#pragma unroll
                    for (int k = 0; k < N; ++k)
                      PW[j + N * k] = out[k];
                  }
              }
              double xpr[NP];
              {
                const double *t = xt;
                opaque_ptr(t);
#pragma unroll
                for (int m = 0; m < NP; ++m)
                  xpr[m] = t[TB::X_P + m];
              }
#pragma unroll
              for (int k = 0; k < N; ++k)
                {
                  emu_sync();
#pragma unroll
                  for (int j = 0; j < N; ++j)
                    xb_own[64 * j] = PW[j + N * k];
                  HOP_XSYNC();
                  double pl[N];
#pragma unroll
                  for (int j = 0; j < N; ++j)
                    {
                      double v = 0.;
#pragma unroll
                      for (int m = 0; m < NP; ++m)
                        v += xpr[m] * xb_p[64 * j + m];
                      pl[j] = v;
                    }
                  int ws = pq_ws; // (opaque: the 25 addresses are otherwise hoisted out of the marching loop and spilled)
                  opaque(ws);
#pragma unroll
                  for (int j = 0; j < N; ++j)
                    pq_w[(j + N * k) * ws] = pl[j];
                }
            }
          HOP_XSYNC();

          HOP_MARK(2)
          // ================= quadrature points of my plane (:702-893) ================================================
          double R[NQ];
#pragma unroll
          for (int n = 0; n < NQ; ++n)
            R[n] = 0.;
          {
            const double wxd = xt[TB::X_WXD];
            const ctab_t cst = tb(TB::C), dmt = tb(TB::D);
            // collocation derivative: D[i][j] = -D[4 - i][4 - j], 13 of the 25 entries are kept in scalar registers
            auto dm = [&](const int i) { return i <= 12 ? dmt[i] : -dmt[24 - i]; };
            // (e == d ? 1 : 0): the diagonal of the tested gradient without selects
            const double sel0 = d == 0 ? 1. : 0., sel1 = d == 1 ? 1. : 0., sel2 = d == 2 ? 1. : 0.;
            hox::static_for<N>([&](auto zq_) {
              constexpr int zq = decltype(zq_)::value;
              hox::static_for<N>([&](auto yq_) {
                constexpr int yq = decltype(yq_)::value, p = yq + N * zq;
                // ---- state of this point from the ring ------------------------------------------------------------
                double ub[3] = {0., 0., 0.}, L[3] = {0., 0., 0.}, tr = 0., ubd = 0.;
                int    rot_p = rot;
                opaque_s(rot_p);
                if constexpr (NPC > 0)
                  {
                    if constexpr (p >= LA)
                      wait_vmcnt<(LA - 1) * NPC>();
                    // (p < LA: issued during the last step or before the loop; the waits of the nodal loads, which
                    // are younger, have covered them)
                    // (the slot offset is re-derived from an opaque copy of `rot`: computed once per step for all 25
                    // points it would be 25 live -- spilled -- registers)
                    const int   sl = ((p + rot_p) & (RSL - 1)) * 2;
                    const char *ra = ring_cx + sl * SLOT_B;
                    ub[0] = *reinterpret_cast<const double *>(ra);
                    ub[1] = *reinterpret_cast<const double *>(ra + 80);
                    ub[2] = *reinterpret_cast<const double *>(ra + 160);
                    const double *own = reinterpret_cast<const double *>(ra + 80 * d);
                    ubd               = own[0];
                    if constexpr (LIN_MODE == 0)
                      {
                        const char *rb = ra + SLOT_B;
                        L[0]           = own[1];
                        L[1]           = *reinterpret_cast<const double *>(rb + 80 * d);
                        L[2]           = *reinterpret_cast<const double *>(rb + 80 * d + 8);
                        tr = *reinterpret_cast<const double *>(ra + 8) + *reinterpret_cast<const double *>(rb + 80) +
                             *reinterpret_cast<const double *>(rb + 160 + 8);
                      }
                    else
                      tr = own[1]; // div u_lin
                  }
                // ---- my component: value and gradient ----------------------------------------------------------------
                double gy = 0., gz = 0.;
#pragma unroll
                for (int j = 0; j < N; ++j)
                  gy += dm(yq * N + j) * U[j + N * zq];
#pragma unroll
                for (int k = 0; k < N; ++k)
                  gz += dm(zq * N + k) * U[yq + N * k];
                const double u = U[p], g0 = ux[p], g1 = gy * cst[TB::C_IH + 1], g2 = gz * cst[TB::C_IH + 2];
                // ---- the other components: one record per lane, read back with per-lane offsets ---------------------
                emu_sync();
                xc_own[0] = u, xc_own[1] = g0, xc_own[2] = g1, xc_own[3] = g2;
                wave_fence();
                const double u0 = xc_cx[0], u1 = xc_cx[20], u2 = xc_cx[40];
                const double t0 = xc_cx[1 + d], t1 = xc_cx[21 + d], t2 = xc_cx[41 + d]; // d_d u_e: column d of grad u
                const double div = xc_cx[1] + xc_cx[22] + xc_cx[43];
                const double pq  = WITH_P ? pq_cx[p] : 0.;
                if constexpr (NPC > 0)
                  {
                    // the slots of this point are free: its successor LA points ahead takes the slots of point p - 1
                    constexpr int pn = p + LA;
                    if constexpr (pn < NQ)
                      ring_issue(stc, pn, (pn + rot_p) & (RSL - 1));
                    else
                      ring_issue(stn, pn - NQ, (pn + rot_p) & (RSL - 1));
                  }
                // ---- quadrature-point operation ----------------------------------------------------------------------
                const double jxw = wxd * (cst[TB::C_W + yq] * cst[TB::C_W + zq]);
                double       res = 0.;
                if constexpr (LIN_MODE == 0) // Newton :802-816
                  res = cst[TB::C_BETA] * (div * ubd + tr * u) + ub[0] * g0 + ub[1] * g1 + ub[2] * g2 +
                        u0 * L[0] + u1 * L[1] + u2 * L[2];
                else if constexpr (LIN_MODE == 1) // Picard-type :817-826
                  res = cst[TB::C_BETA] * tr * u + ub[0] * g0 + ub[1] * g1 + ub[2] * g2;
                const double conv = (cst[TB::C_CA] * u + cst[TB::C_CB] * res) * jxw; // :717,:827-835
                const double diag = cst[TB::C_TGD] * div - pq;
                const double tmu  = cst[TB::C_TMU];
                // :859-892 row d of tmu (grad u + grad u^T) + (tau_gd div - p) I, times JxW J^{-1}
                const double tg0 = (tmu * (g0 + t0) + sel0 * diag) * (jxw * cst[TB::C_IH + 0]);
                const double tg1 = (tmu * (g1 + t1) + sel1 * diag) * (jxw * cst[TB::C_IH + 1]);
                const double tg2 = (tmu * (g2 + t2) + sel2 * diag) * (jxw * cst[TB::C_IH + 2]);
                if constexpr (WITH_P)
                  {
                    emu_sync(); // (the other components have read p)
                    // (the stride through an opaque copy: the 25 addresses are otherwise computed once, ahead of the
                    // marching loop, and spilled)
                    int ws = pq_ws;
                    opaque(ws);
                    pq_w[p * ws] = -div * jxw; // :853-856, in place of p
                  }
                // ---- integrate, y and z in registers -----------------------------------------------------------------
                R[p] += conv;
#pragma unroll
                for (int j = 0; j < N; ++j)
                  R[j + N * zq] += dm(yq * N + j) * tg1;
#pragma unroll
                for (int k = 0; k < N; ++k)
                  R[yq + N * k] += dm(zq * N + k) * tg2;
                ux[p] = tg0; // (d/dx of this point is consumed: its slot takes the tested x-gradient)
                if (HOP_SCHED)
                  __builtin_amdgcn_sched_barrier(0); // one point at a time: interleaved, the points' temporaries add up
              });
            });
          }

          HOP_MARK(3)
          // ================= integrate (:897-907): x across the lanes, then z and y in registers =====================
          {
            double xsc[N], xdsc[N];
            {
              const double *t = xt;
              opaque_ptr(t);
#pragma unroll
              for (int m = 0; m < N; ++m)
                xsc[m] = t[TB::X_SC + m], xdsc[m] = t[TB::X_DSC + m];
            }
            // line by line: sum_q S[q][x] R(q) + (D S)[q][x] tx(q), q = the lanes of my group
#pragma unroll
            for (int k = 0; k < N; ++k)
              {
                emu_sync();
#pragma unroll
                for (int j = 0; j < N; ++j)
                  xb_own[64 * j] = ux[j + N * k];
                HOP_XSYNC();
                double acc[N];
#pragma unroll
                for (int j = 0; j < N; ++j)
                  {
                    double v = 0.;
#pragma unroll
                    for (int m = 0; m < N; ++m)
                      v += xdsc[m] * xb_grp[64 * j + m];
                    acc[j] = v;
                  }
                emu_sync();
#pragma unroll
                for (int j = 0; j < N; ++j)
                  xb_own[64 * j] = R[j + N * k];
                HOP_XSYNC();
#pragma unroll
                for (int j = 0; j < N; ++j)
                  {
                    double v = acc[j];
#pragma unroll
                    for (int m = 0; m < N; ++m)
                      v += xsc[m] * xb_grp[64 * j + m];
                    R[j + N * k] = v;
                  }
              }
            EoMat<N, N, 1> mT;
            mT.load(tb(TB::ST));
#pragma unroll
            for (int j = 0; j < N; ++j)
              {
                double in[N], out[N];
#pragma unroll
                for (int k = 0; k < N; ++k)
                  in[k] = R[j + N * k];
                mT.template apply<false>(in, out);
#pragma unroll
                for (int k = 0; k < N; ++k)
                  R[j + N * k] = out[k];
              }
#pragma unroll
            for (int k = 0; k < N; ++k)
              {
                double in[N], out[N];
#pragma unroll
                for (int j = 0; j < N; ++j)
                  in[j] = R[j + N * k];
                mT.template apply<false>(in, out);
#pragma unroll
                for (int j = 0; j < N; ++j)
                  R[j + N * k] = out[j];
              }
          }
          HOP_MARK(4)
          double Rp[NP * NP];
#pragma unroll
          for (int n = 0; n < NP * NP; ++n)
            Rp[n] = 0.;
          if constexpr (WITH_P)
            {
              double xpc[N];
              {
                const double *t = xt;
                opaque_ptr(t);
#pragma unroll
                for (int m = 0; m < N; ++m)
                  xpc[m] = t[TB::X_PC + m];
              }
              HOP_XSYNC();
              double T[NQ];
#pragma unroll
              for (int n = 0; n < NQ; ++n)
                {
                  double v = 0.;
#pragma unroll
                  for (int m = 0; m < N; ++m)
                    v += xpc[m] * pq_c[m * NQ + n];
                  T[n] = v;
                }
              EoMat<NP, N, 1> mPT;
              mPT.load(tb(TB::SPT));
              double T2[N * NP]; // [yq][k]
#pragma unroll
              for (int j = 0; j < N; ++j)
                {
                  double in[N], out[NP];
#pragma unroll
                  for (int k = 0; k < N; ++k)
                    in[k] = T[j + N * k];
                  mPT.template apply<false>(in, out);
#pragma unroll
                  for (int k = 0; k < NP; ++k)
                    T2[j + N * k] = out[k];
                }
#pragma unroll
              for (int k = 0; k < NP; ++k)
                {
                  double in[N], out[NP];
#pragma unroll
                  for (int j = 0; j < N; ++j)
                    in[j] = T2[j + N * k];
                  mPT.template apply<false>(in, out);
#pragma unroll
                  for (int j = 0; j < NP; ++j)
                    Rp[j + NP * k] = out[j];
                }
            }

          HOP_MARK(5)
          // ================= carry in x, faces inside the wave, emit the finished node planes =========================
          emu_sync();
          {
            const bool take = x == 0 && step > 0;
#pragma unroll
            for (int n = 0; n < NQ; ++n)
              {
                const double v = cru[n];
                R[n] += take ? v : 0.;
              }
            if (WITH_P)
              {
#pragma unroll
                for (int n = 0; n < NP * NP; ++n)
                  {
                    const double v = crp[n];
                    Rp[n] += take ? v : 0.;
                  }
              }
          }
          emu_sync();
          {
            int ws = cru_ws, wsp = crp_ws;
            opaque(ws);
            opaque(wsp);
#pragma unroll
            for (int n = 0; n < NQ; ++n)
              cru_w[n * ws] = R[n];
            if (WITH_P)
              {
#pragma unroll
                for (int n = 0; n < NP * NP; ++n)
                  crp_w[n * wsp] = Rp[n];
              }
          }
          HOP_MARK(6)
#if HOP_PREFETCH_NODES
          // the nodal planes of the next step travel while this one is emitted
          load_nodes(cx + 1);
#endif
          emit(R, Rp, x < K, d == 0 && x < KP, K * cx + x, KP * cx + xp, K * step + x, KP * step + xp, false);
#if !HOP_PREFETCH_NODES
          load_nodes(cx + 1);
#endif
          HOP_MARK(7)
#if HOP_STAMP
          acc[9] += 1;
#endif
        }
#if HOP_STAMP
      if (A.stamps && lane == 0)
        for (int j = 0; j < 10; ++j)
          A.stamps[(size_t)tile * 10 + j] = acc[j];
#endif
      if (NPC > 0)
        wait_vmcnt<0>(); // no copy may land in LDS after the wave has left
      // ---- the last node plane of the chunk ----------------------------------------------------------------------------
      {
        double R[NQ], Rp[NP * NP];
        HOP_XSYNC();
#pragma unroll
        for (int n = 0; n < NQ; ++n)
          R[n] = cru[n];
#pragma unroll
        for (int n = 0; n < NP * NP; ++n)
          Rp[n] = WITH_P ? crp[n] : 0.;
        emit(R, Rp, x == 0, d == 0 && x == 0, K * (cx0 + ns), KP * (cx0 + ns), K * ns, KP * ns, true);
      }
    }
  } // namespace hop
} // namespace adaflo_hip
