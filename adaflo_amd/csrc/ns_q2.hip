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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

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

    struct Q2Args
    {
      int    ncx, ncy, ncz, nnx, nny, nnz, npx, npy, npz;
      int    tiles_x, tiles_y, LZ, n_chunks;
      double s0, s1, s2;   // first row of the 3x3 interpolation matrix (rows: [s0 s1 s2], [0 1 0], [s2 s1 s0])
      // collocation derivative at the 3 Gauss points: rows (a0 a1 a2), (-b 0 b), (-a2 -a1 -a0)
      double ah[3][4];     // (a0, a1, a2, b) / h[e] per direction (ISO: only ah[0] is read)
      double wj[4];        // JxW by number of "middle" indices of the q-point: det * w0^(3-m) * w1^m
      double cA, cB;       // conv = cA * u + cB * res, cA = gamma*rho - damping, cB = tau1*rho
      double beta, tau_gd, tmu;
      double gamma, tau1; // variable coefficients: cA = gamma rho_q - damping_q, cB = tau1 rho_q, tmu = tau1 mu_q
      int    integrate_p;
      long   state_stride; // double2 elements per (tile, layer) block (payload + skew padding)
      double *slab_u, *zslab_u, *slab_p, *zslab_p; // seam partial sums, see q2_seam_fixup_kernel
      const double *ext_u; // residual of the extrapolating schemes (template EXT): extrap_old u_old + extrap_old_old u_old_old
      uint32_t con_u, con_p;
      const double *src_u, *src_p;
      double       *dst_u, *dst_p;
      const double *state;
      const double *rho, *mu, *damp; // recompute mode with variable coefficients: generic arrays [cell][q]
      // residual mode (template RES): combination of the old solutions at the nodes
      // (weight_old u_old + weight_old_old u_old_old), its factor rho_old in the momentum row,
      // and the streaming state array the kernel WRITES (linearisation point = src)
      const double *old_u;
      double        c_old;
      double       *state_out;
      double        c_div; // divergence mode: -1 or -viscosity (weight_by_viscosity)
      // phased execution for the multi-GPU overlap (launch_ns_vmult_q2): explicit workgroup list
      // for the main kernel, node filter for the fix-up (1: only nodes on the inter-GPU interface
      // faces `iface`, 2: all other nodes, 0: everything)
      const int *wg_list;
      int        wg_offset, wg_count, fix_mode;
      uint32_t   iface;
    };

    __device__ __forceinline__ bool fix_skip(const Q2Args &A, const int I, const int J, const int K,
                                             const int nn_x, const int nn_y, const int nn_z)
    {
      if (A.fix_mode == 0)
        return false;
      const bool on = (I == 0 && (A.iface & 1u)) || (I == nn_x - 1 && (A.iface & 2u)) ||
                      (J == 0 && (A.iface & 4u)) || (J == nn_y - 1 && (A.iface & 8u)) ||
                      (K == 0 && (A.iface & 16u)) || (K == nn_z - 1 && (A.iface & 32u));
      return A.fix_mode == 1 ? !on : on;
    }

    // Q2_SWIZZLE (experiment, round 5): bit mask of the broadcasts that go through the LDS crossbar (ds_swizzle_b32, quad
    // mode: no LDS memory, the LDS pipeline instead of a VALU slot) instead of v_mov_b32_dpp; WHICH = group of the call site
#ifndef Q2_SWIZZLE
#define Q2_SWIZZLE 0
#endif
    // GUARD (kernels built for one workgroup per CU): the two halves go through an (empty) asm statement before the DPP moves
    // read them.  Without it hipcc 7.2 (clang 22) MISCOMPILES the 512-register build of the extrapolating residual --
    // deterministically wrong sums on every mesh.  What round 6 established (profiles/r06_q2_ext_*.log, DESIGN.md
    // "register-allocation dependent results"):
    //  * not a pipeline hazard: on the MI355X a DPP move needs ONE wait state behind an f64 / 32-bit VALU result, behind
    //    v_accvgpr_read and none behind an SALU write of EXEC (scripts/dev/hazard_probe.hip; the compiler pads two), and the
    //    wrong code object stays wrong, same values, with `s_nop 7` behind EVERY VALU instruction of the kernel (listing
    //    edited and re-assembled, scripts/dev/isa_patch_build.py);
    //  * not the hand-written asm: wait states behind every asm statement that restores EXEC change nothing;
    //  * the instruction stream is what is wrong: the same source is bitwise right as soon as the register allocation
    //    differs (-mllvm -enable-misched=0, -vgpr-regalloc=basic, -amdgpu-sdwa-peephole=0, this guard, or 256 registers),
    //    while -amdgpu-dpp-combine=0, -enable-post-misched=0, -amdgpu-waitcnt-forcezero, -amdgpu-spill-vgpr-to-agpr=0,
    //    -amdgpu-enable-rewrite-partial-reg-uses=0 leave it wrong; -verify-machineinstrs is silent;
    //  * the symptom: the unscaled, lane-local gradient entries (d_e u_d, twice that for e = d) of the layer's last
    //    quadrature point are ADDED to three finished node values of the layer (the first layer adds whatever the
    //    registers held before): a stale register read in the combine phase, guarding any subset of the broadcast sites
    //    moves the error elsewhere (scripts/dev/lb_diff_ext.sh, -DQ2_QG_SITES).
    // The guard forces copies and pins the order of the moves against the other asm statements, which is enough to keep the
    // allocator away from the bad assignment; tests/test_lb_differential_gpu.py compares every kernel built for 512 registers
    // with the 256-register build of the same source bit by bit, so a compiler update that brings the problem back fails a
    // test instead of a simulation.  Kernels at two waves per SIMD keep the direct form (two moves less per broadcast).
    template <int SEL, int WHICH = 0, bool GUARD = false>
    __device__ __forceinline__ double quad_bcast(const double x)
    {
      constexpr int ctrl = SEL * 0x55; // quad_perm:[SEL,SEL,SEL,SEL]
      int lo = __double2loint(x), hi = __double2hiint(x);
      if constexpr (GUARD)
        asm volatile("" : "+v"(lo), "+v"(hi));
      if constexpr ((Q2_SWIZZLE >> WHICH) & 1)
        {
          lo = __builtin_amdgcn_ds_swizzle(lo, 0x8000 | ctrl);
          hi = __builtin_amdgcn_ds_swizzle(hi, 0x8000 | ctrl);
        }
      else
        {
          lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
          hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
        }
      return __hiloint2double(hi, lo);
    }

    // general permutation inside the quad: lane l receives the value of lane P_l (CTRL = P_0 | P_1 << 2 | P_2 << 4 | P_3 << 6)
    template <int CTRL, bool GUARD = false>
    __device__ __forceinline__ double quad_permute(const double x)
    {
      int lo = __double2loint(x), hi = __double2hiint(x);
      if constexpr (GUARD)
        asm volatile("" : "+v"(lo), "+v"(hi));
      lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
      hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
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

    // ---------------------------------------------------------------------------------
    // LDS carve-up (dynamic shared memory, doubles)
    // ---------------------------------------------------------------------------------
#ifndef Q2_RING
#define Q2_RING 9
#endif
    constexpr int RING      = Q2_RING;        // state pieces per wave in the LDS ring (54 % RING == 0)
    constexpr int PIECE     = 96;             // doubles per piece: 48 lanes x 16 B
    constexpr int L_RING    = 0;
    // node planes in LDS: rows padded to a multiple of 16 B so that one 16-B-per-lane LDS-DMA
    // instruction copies whole rows (26 / 5 lanes per row); the row's last double sits one slot
    // later when the row length is odd (the last chunk is read shifted back by 8 B so that the
    // copy never reads beyond the row), see dma_rows_b128
    constexpr int UROW      = 52;                         // 17 nodes x 3 comps = 51 doubles (+1)
    constexpr int PROW      = 10;                         // 9 doubles (+1)
    constexpr int UPLANE_L  = PNY * UROW;                 // 884
    constexpr int PPLANE_L  = QNY * PROW;                 // 90
    constexpr int L_UPL     = L_RING + 4 * RING * PIECE;  // 3 velocity node planes
    constexpr int L_UPL_    = L_UPL;
    static_assert(3 * (PNY * 52) <= 4 * RING * PIECE, "the residual mode keeps the old-solution planes in the ring area");
    constexpr int L_PPL     = L_UPL + 3 * UPLANE_L;       // 2 pressure node planes
    // publish scratch: the unconditional reads "west / south of the first cells" (deselected afterwards) reach
    // up to 27 doubles below a slot, i.e. (slot 0 of plane 0) below the scratch itself: 4 doubles in front
    // (cell 0 reads [-3, -1]).
    // The scratch is only live between the quadrature loop and the end of a layer.  During the loop it
    // serves as an EXTENSION OF THE STATE RING (NRING_X more slots per wave, see RingSched): the kernel is
    // limited by how well the state stream overlaps with the arithmetic, i.e. by the pieces in flight.
    constexpr int NRING_X   = 9;                          // extension slots per wave (constant coefficients)
    constexpr int L_EXT     = L_PPL + 2 * PPLANE_L;       // [4 waves][NRING_X][PIECE]
    constexpr int L_SCRU    = L_EXT + 4;                  // [3 planes][5 slots][192]
    constexpr int L_SCRP    = L_SCRU + 15 * NCELL * 3;    // [2 planes][3 slots][64]
    constexpr int L_RIMT    = L_EXT + 4 * NRING_X * PIECE; // rim-thread descriptors: 192 x int4 (waves 0..2)
    static_assert(L_SCRP + 6 * NCELL <= L_RIMT, "the publish scratch lives inside the ring extension");
    constexpr int L_TOTAL   = L_RIMT + 384;
    static_assert(L_TOTAL * 8 <= 80 * 1024, "two workgroups per CU");

    // Issue schedule of the state ring, all compile time.  A layer has 54 pieces p = 2q + half per wave;
    // piece p lives in slot p % NS, NS = NR real + NX extension slots (NS divides 54: the slots do not
    // depend on the layer).  After point q has been consumed the pieces LA ahead are issued into the two
    // slots that became free (LA = NS for even NS, NS - 1 for odd NS); pieces of the NEXT layer are issued
    // that way only into real slots -- the extension is the publish scratch and is busy between the loop
    // and the end of the layer -- and the next layer's pieces in extension slots follow as one burst at the
    // top of that layer.  younger(q) = number of vector-memory operations issued after the later of the
    // pieces 2q, 2q+1 and before point q waits for them (vmcnt counts in order of issue): the burst, the
    // node-plane copies (npl), the pieces of the points in between -- and at least `nst` stores of the
    // combine phase (fewer than the real count only makes the wait stronger).
    template <int NR, int NX>
    struct RingSched
    {
      static constexpr int NS = NR + NX, LA = (NS % 2 == 0) ? NS : NS - 1;
      static_assert(54 % NS == 0, "ring geometry");
      static constexpr bool real_slot(const int p) { return p % NS < NR; }
      // does point q issue piece t = 2q + LA + j (t >= 54: piece t - 54 of the next layer)?
      static constexpr bool issues(const int t) { return t < 54 || real_slot(t - 54); }
      static constexpr int n_issued(const int q) { return (issues(2 * q + LA) ? 1 : 0) + (issues(2 * q + LA + 1) ? 1 : 0); }
      static constexpr int n_burst()
      {
        int n = 0;
        for (int p = 0; p < LA; ++p)
          n += real_slot(p) ? 0 : 1;
        return n;
      }
      // position of piece p of the current layer in the sequence
      // [pieces issued by the previous layer][nst stores][burst][npl plane copies][points 0, 1, ...]
      static constexpr int position(const int p, const int npl, const int nst)
      {
        int pos = 0;
        if (p < LA && real_slot(p))
          {
            for (int r = 0; r < p; ++r)
              pos += real_slot(r) ? 1 : 0;
            return pos;
          }
        for (int r = 0; r < LA; ++r)
          pos += real_slot(r) ? 1 : 0;
        pos += nst;
        if (p < LA)
          {
            for (int r = 0; r < p; ++r)
              pos += real_slot(r) ? 0 : 1;
            return pos;
          }
        pos += n_burst() + npl;
        for (int q = 0; 2 * q + LA < p; ++q)
          pos += (2 * q + LA + 1 < p) ? n_issued(q) : (issues(2 * q + LA) ? 1 : 0);
        return pos;
      }
      static constexpr int younger(const int q, const int npl, const int nst)
      {
        int total = 0; // operations in the sequence before point q issues its own
        for (int r = 0; r < LA; ++r)
          total += 1;  // every piece below LA: previous layer or burst
        total += nst + npl;
        for (int i = 0; i < q; ++i)
          total += n_issued(i);
        const int pa = position(2 * q, npl, nst), pb = position(2 * q + 1, npl, nst);
        return total - ((pa > pb ? pa : pb) + 1);
      }
    };
    // younger(q, ...) as a table: an index that is constant after unrolling folds to an immediate
    template <int NR, int NX, int NPL, int NST>
    struct RingWaits
    {
      struct Table
      {
        int v[27];
      };
      static constexpr Table make()
      {
        Table t{};
        for (int q = 0; q < 27; ++q)
          t.v[q] = RingSched<NR, NX>::younger(q, NPL, NST);
        return t;
      }
      static constexpr Table table = make();
    };
    // (the schedule of round 1 / 2: 9 slots, 8 pieces ahead -- 6 behind the pieces of a point)
    static_assert(RingSched<9, 0>::younger(0, 6, 0) == 12 && RingSched<9, 0>::younger(3, 6, 0) == 12 &&
                    RingSched<9, 0>::younger(4, 6, 0) == 6 && RingSched<9, 0>::younger(26, 6, 0) == 6,
                  "ring schedule");
    [[maybe_unused]] constexpr int XQ = 18;             // doubles per quad record: 4 lanes x 4 + pad (bank spread, 16-B aligned)
    constexpr int NPL_U     = 5;              // plane-DMA instructions per wave and layer (fixed count)
    constexpr int NPL_P     = 1;

    // ---------------------------------------------------------------------------------
    // Seams between workgroups.  A node on the lateral rim of a tile is shared by up to four
    // tiles, a node on the top/bottom plane of a z-chunk by two chunks.  Instead of f64 atomics
    // (measured: the scattered atomic requests cost 0.85 ms of a 2.5 ms kernel) every workgroup
    // stores its PARTIAL sums of such nodes in a private slab with plain stores and a small
    // second kernel (q2_seam_fixup_kernel) adds the partials per node in a fixed order, which
    // also makes the result bitwise reproducible.
    //   slab [wg][plane lp of the chunk][rim r][comp]   rim nodes of the tile, every plane
    //   zslab[wg][tile node][comp]                      top plane of a chunk that has a chunk above
    // ---------------------------------------------------------------------------------
    template <int TN>
    __device__ __forceinline__ int rim_index(const int i, const int j)
    {
      // j == 0: i | j == TN-1: TN + i | i == 0: 2TN-1 + j (1 <= j <= TN-2) | i == TN-1: 3TN-3 + j
      if (j == 0)
        return i;
      if (j == TN - 1)
        return TN + i;
      if (i == 0)
        return 2 * TN - 1 + j;
      return 3 * TN - 3 + j;
    }
    constexpr int RIM_U = 4 * (PNX - 1), RIM_P = 4 * (QNX - 1); // 64, 32

#if defined(Q2_NOSTORE)
#define Q2_STORE(lhs, rhs) do { if ((rhs) == 1.2345e300) lhs = (rhs); } while (0)
#define Q2_STORE_SLAB(lhs, rhs) Q2_STORE(lhs, rhs)
#elif defined(Q2_NOSLAB)
#define Q2_STORE(lhs, rhs) lhs = (rhs)
#define Q2_STORE_SLAB(lhs, rhs) do { if ((rhs) == 1.2345e300) lhs = (rhs); } while (0)
#elif defined(Q2_NODST)
#define Q2_STORE(lhs, rhs) do { if ((rhs) == 1.2345e300) lhs = (rhs); } while (0)
#define Q2_STORE_SLAB(lhs, rhs) lhs = (rhs)
#else
#define Q2_STORE(lhs, rhs) lhs = (rhs)
#if defined(Q2_SLAB_NT)
#define Q2_STORE_SLAB(lhs, rhs) __builtin_nontemporal_store((rhs), &(lhs))
#else
#define Q2_STORE_SLAB(lhs, rhs) lhs = (rhs)
#endif
#endif

    // development (round 6, scripts/dev/lb_diff_ext.sh): wait states behind every asm statement that restores EXEC
#if defined(Q2_EXEC_PAD)
#define Q2_PAD_STR2(x) #x
#define Q2_PAD_STR(x) Q2_PAD_STR2(x)
#define Q2_EXEC_PAD_ASM "\n\ts_nop " Q2_PAD_STR(Q2_EXEC_PAD)
#else
#define Q2_EXEC_PAD_ASM ""
#endif
    __device__ __forceinline__ void lds_barrier()
    {
      // LDS-only workgroup barrier: unlike __syncthreads() it does not drain vmcnt, so the
      // asynchronous global->LDS copies stay in flight across it
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    template <int N>
    __device__ __forceinline__ void wait_vmcnt()
    {
      // (asm + memory clobber: LDS reads of DMA'd data must not be hoisted above the wait)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    }

    // Asynchronous global -> LDS copy (LDS-DMA), hand-issued so that (i) no exec-mask branch
    // splits the quadrature loop into basic blocks, (ii) the address is scalar base + 32-bit
    // lane offset, (iii) hipcc does not pessimise following LDS reads with vmcnt(0).  The copies
    // are invisible to the compiler's waitcnt bookkeeping: every consumer is guarded by an
    // explicit wait_vmcnt<> below.  EXEC is all ones wherever these are called.
    __device__ __forceinline__ unsigned uniform32(const unsigned x)
    {
      return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
    }
    __device__ __forceinline__ unsigned long long uniform64(const unsigned long long x)
    {
      return ((unsigned long long)uniform32((unsigned)(x >> 32)) << 32) | uniform32((unsigned)x);
    }

    //   lds_byte : wave-uniform LDS byte address (goes to M0); lane l writes lds_byte + 16*l
    __device__ __forceinline__ void dma_b128(const void *sbase, const unsigned voff,
                                             const unsigned lds_byte, const unsigned long long mask)
    {
      // (s_nop 4 first: the scalar base may be fresh from v_readfirstlane -- uniform64 -- and a VMEM instruction must be five
      // wait states behind a VALU write of a scalar register it reads; the compiler pads its own instructions, not the
      // inside of this statement.  Without it the copy can run with the base of the copy before: a valid plane of another
      // field -- round 5, the 512-register build of the extrapolating residual)
      asm volatile("s_nop 4\n\t"
                   "s_mov_b32 m0, %0\n\t"
                   "s_mov_b64 exec, %3\n\t"
#if defined(Q2_EXP) && Q2_EXP == 5
                   "global_load_lds_dwordx4 %1, %2\n\t"
#else
                   "global_load_lds_dwordx4 %1, %2 nt\n\t"
#endif
                   "s_mov_b64 exec, -1" Q2_EXEC_PAD_ASM ::"s"(lds_byte), "v"(voff), "s"(sbase), "s"(mask)
                   : "memory");
    }
    //   lane l writes lds_byte + 4*l
    __device__ __forceinline__ void dma_b32(const void *sbase, const unsigned voff,
                                            const unsigned lds_byte, const unsigned long long mask)
    {
      asm volatile("s_nop 4\n\t"
                   "s_mov_b32 m0, %0\n\t"
                   "s_mov_b64 exec, %3\n\t"
                   "global_load_lds_dword %1, %2\n\t"
                   "s_mov_b64 exec, -1" Q2_EXEC_PAD_ASM ::"s"(lds_byte), "v"(voff), "s"(sbase), "s"(mask)
                   : "memory");
    }

    // one 16-byte global store (the address is only 8-byte aligned in general)
    typedef double double2v __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ void store_b128(double *p, const double a, const double b)
    {
#if defined(Q2_NOSTORE) || defined(Q2_NODST)
      if (a != 1.2345e300)
        return;
#endif
      double2v v;
      v.x = a;
      v.y = b;
      // nt: dst is written once and not re-read by this kernel (measured -2 %)
      // (s_nop 1: a store of more than 64 bits must be two wait states ahead of a VALU write of its data registers, and
      // the compiler does not look into the statement -- its next instruction may be a v_accvgpr_read into them)
      asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    }

    // 16-byte store to sbase + voff for the lanes in `mask` (no exec branch in the caller's code)
    __device__ __forceinline__ void store_b128_masked(const void *sbase, const unsigned voff, const double a,
                                                      const double b, const unsigned long long mask)
    {
      double2v v;
      v.x = a;
      v.y = b;
      // (s_nop 0 behind the SALU instruction: two wait states between a store of more than 64 bits and a VALU write of its
      // data registers -- the compiler does not look into the statement; found in round 5 when a 512-register build, whose
      // next instruction often is a v_accvgpr_read into those registers, stored wrong values)
      asm volatile("s_mov_b64 exec, %3\n\t"
                   "global_store_dwordx4 %0, %1, %2 nt\n\t"
                   "s_mov_b64 exec, -1\n\t"
                   "s_nop 0" Q2_EXEC_PAD_ASM ::"v"(voff), "v"(v), "s"(sbase), "s"(mask)
                   : "memory");
    }

#if !defined(Q2_DST_POL) || Q2_DST_POL == 0
#define Q2_DST_POLICY "nt" // dst is written once and not re-read by this kernel
#elif Q2_DST_POL == 1
#define Q2_DST_POLICY ""
#elif Q2_DST_POL == 2
#define Q2_DST_POLICY "sc1"
#elif Q2_DST_POL == 3
#define Q2_DST_POLICY "sc0"
#else
#define Q2_DST_POLICY "sc0 sc1"
#endif
    // Stores of phase E: sbase + voff for the lanes of `mask` that are active at the call site (these run
    // inside divergent code: EXEC is narrowed and restored, not set to all ones), 8 or 16 bytes per lane
    __device__ __forceinline__ void store_b64_masked(const void *sbase, const unsigned voff, const double a,
                                                     const unsigned long long mask)
    {
#if defined(Q2_NOSTORE) || defined(Q2_NODST)
      if (a != 1.2345e300)
        return;
#endif
      unsigned long long saved;
      asm volatile("s_and_saveexec_b64 %0, %4\n\t"
                   "global_store_dwordx2 %1, %2, %3 " Q2_DST_POLICY "\n\t"
                   "s_mov_b64 exec, %0" Q2_EXEC_PAD_ASM
                   : "=&s"(saved)
                   : "v"(voff), "v"(a), "s"(sbase), "s"(mask)
                   : "memory", "scc");
    }
    __device__ __forceinline__ void store_b128_dst(const void *sbase, const unsigned voff, const double a,
                                                   const double b, const unsigned long long mask)
    {
#if defined(Q2_NOSTORE) || defined(Q2_NODST)
      if (a != 1.2345e300)
        return;
#endif
      double2v v;
      v.x = a;
      v.y = b;
      unsigned long long saved;
      asm volatile("s_and_saveexec_b64 %0, %4\n\t"
                   "global_store_dwordx4 %1, %2, %3 " Q2_DST_POLICY "\n\t"
                   "s_mov_b64 exec, %0\n\t"
                   "s_nop 0" Q2_EXEC_PAD_ASM // (two wait states behind a store of more than 64 bits, see store_b128_masked)
                   : "=&s"(saved)
                   : "v"(voff), "v"(v), "s"(sbase), "s"(mask)
                   : "memory", "scc");
    }
    // Load of a single double inside a RARELY taken branch (constrained rows of boundary tiles), waited for
    // on the spot.  A load the compiler can see would make it put `s_waitcnt vmcnt(0)` at the join behind the
    // branch -- executed by every wave in every layer, draining the asynchronous LDS-DMA queue (this cost the
    // round-2 kernel about 0.2 ms of 1.34 ms at 128^3).
    __device__ __forceinline__ double load_now(const double *p)
    {
      double v;
      asm volatile("global_load_dwordx2 %0, %1, off\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=v"(v)
                   : "v"(p)
                   : "memory");
      return v;
    }

    __device__ __forceinline__ unsigned lds_addr(const void *p)
    {
      return (unsigned)(size_t)p; // LDS aperture: the low 32 bits are the LDS byte address
    }

    // One LDS-DMA instruction (16 B per lane) copying up to ROWS_PER = 64 / LPR consecutive rows
    // of a node plane: lanes [r*LPR, (r+1)*LPR) serve row j0 + r.  A row holds nv valid doubles
    // (nv <= 2*LPR - 1); lane c of a row copies doubles [2c, 2c+1], except that for odd nv the
    // last lane copies [nv-2, nv-1] (=> the last double lands in LDS slot nv, slot nv-1 holds a
    // duplicate).  Nothing beyond the row is ever read.  The instruction is always issued (lane
    // 0 is always active; rows beyond the domain re-read the last valid row): the vmcnt
    // bookkeeping of the kernel relies on a fixed number of VMEM operations per wave.
    //   plane : scalar pointer to the first double of the global plane
    //   row_off(j) = doubles from `plane` to the first entry of tile row j (clamped inside)
    template <int LPR, int NROWS>
    __device__ __forceinline__ void dma_rows_b128(const double *plane, const int lane, const int j0,
                                                  const int nv, const int row_stride_dbl,
                                                  const int row0_dbl, const int last_row,
                                                  double *lds_row0)
    {
      const int  r  = lane / LPR, c = lane - r * LPR;
      const int  j  = j0 + r;
      const int  nchunk = (nv + 1) >> 1;
      const bool act = r < 64 / LPR && j < NROWS && c < nchunk;
      const int  dbl = (c == nchunk - 1 && (nv & 1)) ? nv - 2 : 2 * c;
      const int  jg  = min(j, last_row); // rows beyond the domain: harmless duplicate
      const unsigned voff = 8u * (unsigned)(row0_dbl + jg * row_stride_dbl + dbl);
      const unsigned long long mask = __ballot(act) | 1ull;
      // (the plane pointer is wave-uniform; say so explicitly: the "s" operand must not end up in VGPRs)
      dma_b128(reinterpret_cast<const void *>(uniform64(reinterpret_cast<unsigned long long>(plane))),
               act ? voff : 8u * (unsigned)(row0_dbl), lds_addr(lds_row0), mask);
    }

    // the NPL_U row-pair copies one wave contributes to velocity planes K0 and K0+1
    __device__ __forceinline__ void dma_u_planes(const Q2Args &A, double *lds, const int K0,
                                                 const int I0, const int J0, const int wave,
                                                 const int lane, const double *src = nullptr,
                                                 const int lds_base = -1)
    {
      const double *vec = src ? src : A.src_u;
      const int     lb  = lds_base >= 0 ? lds_base : -1;
      const int nv = 3 * min(PNX, A.nnx - I0); // valid doubles of a tile row
#pragma unroll
      for (int t = 0; t < NPL_U; ++t)
        {
          // 2 planes x 9 row pairs (the 9th pair is a single row) dealt round-robin to the 4 waves
          int n = wave + 4 * t;
          if (n >= 18)
            n -= 4;
          const int     pl = n / 9, m = n - 9 * pl, K = K0 + pl, Kg = min(K, A.nnz - 1);
          const double *plane = vec + (size_t)Kg * A.nny * A.nnx * 3;
          dma_rows_b128<26, PNY>(plane, lane, 2 * m, nv, A.nnx * 3, (J0 * A.nnx + I0) * 3,
                                 A.nny - 1 - J0, lds + (lb >= 0 ? lb : L_UPL_) + (K % 3) * UPLANE_L + 2 * m * UROW);
        }
    }

    // one plane (prologue): 9 row pairs over the 4 waves, 3 instructions per wave
    __device__ __forceinline__ void dma_u_plane_single(const Q2Args &A, double *lds, const int K,
                                                       const int I0, const int J0, const int wave,
                                                       const int lane, const double *src = nullptr,
                                                       const int lds_base = -1)
    {
      const int     nv    = 3 * min(PNX, A.nnx - I0);
      const double *plane = (src ? src : A.src_u) + (size_t)K * A.nny * A.nnx * 3;
      const int     lb    = lds_base;
#pragma unroll
      for (int t = 0; t < 3; ++t)
        {
          int m = wave + 4 * t;
          if (m >= 9)
            m -= 4;
          dma_rows_b128<26, PNY>(plane, lane, 2 * m, nv, A.nnx * 3, (J0 * A.nnx + I0) * 3,
                                 A.nny - 1 - J0, lds + (lb >= 0 ? lb : L_UPL_) + (K % 3) * UPLANE_L + 2 * m * UROW);
        }
    }

    // a pressure plane is a single instruction (9 rows x 5 lanes); every wave issues it (the
    // copies are identical) to keep the per-wave VMEM op count fixed
    __device__ __forceinline__ void dma_p_plane(const Q2Args &A, double *lds, const int K,
                                                const int I0, const int J0, const int lane)
    {
      const int     nv    = min(QNX, A.npx - I0);
      const int     Kg    = min(K, A.npz - 1);
      const double *plane = A.src_p + (size_t)Kg * A.npy * A.npx;
      dma_rows_b128<5, QNY>(plane, lane, 0, nv, A.npx, J0 * A.npx + I0, A.npy - 1 - J0,
                            lds + L_PPL + (K % 2) * PPLANE_L);
    }

    // ISO: cubic cells, one set of derivative coefficients for all directions
    // RES: residual mode (source/navier_stokes_matrix.cc:266-293 with NavierStokesOps::residual):
    // src is the solution itself (plain read incl. boundary values), the combination of the old
    // solutions enters the momentum row, the nonlinear term is evaluated with src, and the
    // quadrature-point state (:725-800) is WRITTEN in the streaming layout instead of being read
    // DIV: divergence block only (local_divergence, :920-961): the pressure lane integrates
    // (q, c_div div u), the velocity lanes evaluate but integrate and emit nothing
    // RCP: recompute-state mode (round 5, review item 3): the Newton vmult does NOT stream the 2 592 B per cell of
    // (u_lin, grad u_lin); the nodal linearisation point (the solution vector of the last residual, kept by the host) is
    // gathered, interpolated and differentiated next to the source field -- the state is a function of the nodal field,
    // :778-816 -- and the other components' values and the trace come from quad broadcasts instead of ring reads
    // EXT (with RES; round 5): the residual of the schemes that linearise about the extrapolated old velocity
    // (navier_stokes_matrix.cc:644-647, 740-782; LIN_MODE 1 = semi-implicit, stores (u_ext, div u_ext) as the state of the
    // vmults; LIN_MODE 2 = explicit).  u_ext = extrap_old u_old + extrap_old_old u_old_old is combined at the nodes by the
    // launcher; its node planes take three more plane buffers behind L_TOTAL (the kernel runs at one workgroup per CU), it
    // is interpolated like the other fields and differentiated per point like the linearisation point of the RCP mode
    template <int LIN_MODE, bool WITH_P, bool ISO, bool VARCO, bool RES = false, bool DIV = false, bool RCP = false, bool EXT = false>
    // waves per SIMD the registers are allocated for.  The residual mode holds three fields (solution, old-solution
    // combination, sums) and spilled 160-184 B at 256 registers; with 512 (one workgroup per CU) it has no scratch and
    // is faster: 128^3 2.29 -> 2.10 ms (round 5, profiles/r05_residual_lb.log).  The operator modes stream the state and
    // want two workgroups per CU.
#ifndef Q2_RES_LB
#define Q2_RES_LB 1
#endif
#ifndef Q2_LB
#define Q2_LB 2
#endif
    // The recompute mode holds three fields of 27 values as well.  Cubic cells: 256 registers + 32 B of scratch, two
    // workgroups per CU -- 128^3 kernel 1.016 ms against 1.462 ms with 512 registers / one workgroup per CU (266 used) and
    // 1.287 ms for the streaming kernel (round 5, profiles/r05_recompute.log): the mode is bound by FP64 issue and wants the
    // second wave.  Other cells (three sets of derivative coefficients): 160 B of scratch at 256, none at 512.
#ifndef Q2_RCP_LB
#define Q2_RCP_LB (ISO ? 2 : 1)
#endif
    // (EXT: four fields of 27 values -- 512 registers, no scratch; needs the guarded quad_bcast, see there.  256 registers:
    // 184-516 B of scratch, 128^3 semi-implicit 4.36 instead of 2.7 ms)
#ifndef Q2_EXT_LB
#define Q2_EXT_LB 1
#endif
    __global__ __launch_bounds__(NT, (EXT ? Q2_EXT_LB : (RES ? Q2_RES_LB : (RCP ? Q2_RCP_LB : Q2_LB)))) void ns_q2_kernel(const Q2Args A)
    {
      constexpr bool RING_ON = LIN_MODE != 2 && !RES && !RCP; // state stream through the LDS ring (RCP with variable
                                                              // coefficients: rho, mu, damping by plain loads from the
                                                              // generic arrays, 24 B per cell and point, L2-friendly)
      static_assert(!RCP || (LIN_MODE != 2 && !RES && !DIV), "recompute mode: Newton / Picard-type vmult");
      static_assert(!EXT || (RES && LIN_MODE != 0 && WITH_P), "extrapolating residual: semi-implicit / explicit (round 6: also with variable coefficients)");
      constexpr int L_OLDP = L_RING, L_EXTP = L_TOTAL; // plane buffers of the second / third nodal field
#if defined(Q2_QG_OFF) // (development: the unguarded form in the one-workgroup-per-CU builds, scripts/dev/lb_diff_ext.sh;
                       // Q2_QG_SITES: bit mask of the call-site groups that keep the guard -- 1 quadrature loop, 2 second / third
                       // field (recompute and extrapolating modes), 4 phase E, 8 coefficients)
#ifndef Q2_QG_SITES
#define Q2_QG_SITES 0
#endif
      constexpr bool QG = false;
      constexpr bool QG1 = Q2_QG_SITES & 1, QG2 = Q2_QG_SITES & 2, QG4 = Q2_QG_SITES & 4, QG8 = Q2_QG_SITES & 8;
#else
#define Q2_QG_ALL
#endif
#if defined(Q2_QG_ALL)
      constexpr bool QG = (EXT ? Q2_EXT_LB : (RES ? Q2_RES_LB : (RCP ? Q2_RCP_LB : Q2_LB))) == 1; // quad_bcast guard
      constexpr bool QG1 = QG, QG2 = QG, QG4 = QG, QG8 = QG;
#endif
      static_assert(!DIV || (LIN_MODE == 2 && WITH_P && !RES && !VARCO), "divergence mode");
      // (RES && VARCO, round 5: the residual of two-phase flow -- the coefficients of the layer's 27 points are read from the
      // generic arrays [cell][27], lane d of a quad its array (rho, mu, damping), and handed round by DPP; the state goes
      // out in the constant-coefficient layout: the vmults of this Newton step recompute it or re-lay it out with the
      // coefficient pieces)
      extern __shared__ double lds[];
      // state ring geometry.  Constant coefficients: pieces of 48 lanes x 16 B, 9 slots.
      // Variable rho/mu/damping (two-phase flow): lanes 48..63 of every piece carry the
      // coefficients of the wave's 16 cells -- (rho, mu) in half 0, (damping, -) in half 1 --
      // so the DMA instruction count per point stays at two; 6 slots of 64 lanes x 16 B.
      constexpr int PIECE_ = VARCO ? 128 : PIECE, RING_ = VARCO ? 6 : RING;
      constexpr int PLANES_ = VARCO ? 64 : 48; // double2 per wave and piece
      // + extension slots in the publish scratch while the quadrature loop runs (RingSched)
      // (measured at 128^3, round 3: 18 instead of 8 pieces in flight per wave change nothing, 1.39 ms either
      // way -- the kernel sits at the mixed read/write ceiling of the HBM system, scripts/dev/mix_probe.hip --
      // and the extension costs a fourth barrier per layer: off by default)
#if defined(Q2_RING_EXT)
      constexpr int NX_ = RING_ON ? (VARCO ? 3 : NRING_X) : 0;
#else
      constexpr int NX_ = 0;
#endif
      using RS = RingSched<RING_, NX_>;
      constexpr int NS_ = RS::NS, LA_ = RS::LA;
      static_assert(4 * RING_ * PIECE_ <= 4 * RING * PIECE && (3 * RING_ + NX_) * PIECE_ <= 4 * NRING_X * PIECE, "ring geometry");

      const int tid = threadIdx.x, lane = tid & 63;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform by construction
      const int d = lane & 3, cq = lane >> 2;
      const int cxl = cq & 7, cyl = 2 * wave + (cq >> 3), cell = cyl * TX + cxl;

      // workgroup -> (tile, z-chunk); chunks of one tile column are consecutive
      // in the remapped index so that an XCD's L2 sees neighbouring work
      const long nwg = A.wg_list ? (long)A.wg_count : (long)A.tiles_x * A.tiles_y * A.n_chunks;
      long       wg  = xcd_remap(blockIdx.x, nwg);
      if (A.wg_list)
        wg = A.wg_list[A.wg_offset + wg];
      const int  bz  = (int)(wg % A.n_chunks);
      const int  bt  = (int)(wg / A.n_chunks);
      const int  bx = bt % A.tiles_x, by = bt / A.tiles_x;
      const int  cz0 = bz * A.LZ;
      const int  nl  = min(A.LZ, A.ncz - cz0);
      const int  I0 = 2 * TX * bx, J0 = 2 * TY * by; // velocity node origin of the tile
      const int  Ip0 = TX * bx, Jp0 = TY * by;       // pressure node origin
      const int  tcx = min(TX, A.ncx - TX * bx), tcy = min(TY, A.ncy - TY * by);
      const size_t wgs = (size_t)bt * A.n_chunks + bz; // slab index of this workgroup

      const double s0 = A.s0, s1 = A.s1, s2 = A.s2;
      const bool   is_p  = d == 3;
      const double tmu_l = is_p ? 0. : A.tmu; // the pressure lane integrates no gradient terms
      const double d0 = d == 0 ? 1. : 0., d1 = d == 1 ? 1. : 0., d2 = d == 2 ? 1. : 0.;

      const bool valid = cxl < tcx && cyl < tcy;
      const bool lastx = valid && cxl == tcx - 1; // last valid cell of its row (fix_last below)
      // ---- per-lane flags: in-plane positions that read_dof_values returns as zero (gather) ---
      // bit li + 3*lj, local (li, lj) in {0,1,2}^2 (pressure lane: {0,1}^2 in the same bit positions)
      unsigned m_zero = 0;
      if (!RES)
        {
          const int  nn_x = is_p ? A.npx : A.nnx, nn_y = is_p ? A.npy : A.nny;
          const int  deg  = is_p ? 1 : 2;
          const int  ib = (is_p ? Ip0 : I0) + deg * cxl, jb = (is_p ? Jp0 : J0) + deg * cyl;
          const uint32_t con = is_p ? A.con_p : A.con_u;
          const int  st = is_p ? 1 : 3, cc = is_p ? 0 : d;
          for (int lj = 0; lj <= 2; ++lj)
            for (int li = 0; li <= 2; ++li)
              {
                const int  I = ib + li, J = jb + lj;
                const bool cx_ = (I == 0 && (con >> (st * 0 + cc) & 1)) || (I == nn_x - 1 && (con >> (st * 1 + cc) & 1));
                const bool cy_ = (J == 0 && (con >> (st * 2 + cc) & 1)) || (J == nn_y - 1 && (con >> (st * 3 + cc) & 1));
                if (cx_ || cy_)
                  m_zero |= 1u << (li + 3 * lj);
              }
        }
      // does a regular node of this tile sit on a constrained low face of the domain (phase E1)?
      const bool con_lat_tile = (I0 == 0 && (((A.con_u)&7u) != 0u || (A.con_p & 1u) != 0u)) ||
                                (J0 == 0 && (((A.con_u >> 6) & 7u) != 0u || ((A.con_p >> 2) & 1u) != 0u));
      const bool     conz_lo = is_p ? (A.con_p >> 4 & 1) : (A.con_u >> (12 + d) & 1);
      const bool     conz_hi = is_p ? (A.con_p >> 5 & 1) : (A.con_u >> (15 + d) & 1);
      const unsigned lane_g  = is_p ? (unsigned)((Jp0 + cyl) * A.npx + Ip0 + cxl) :
                                      (unsigned)(((J0 + 2 * cyl) * A.nnx + I0 + 2 * cxl) * 3 + d);

      // the last double of an odd-length LDS plane row sits one slot later (dma_rows_b128)
      const int nv_row   = is_p ? min(QNX, A.npx - Ip0) : 3 * min(PNX, A.nnx - I0);
      const int fix_last = (lastx && (nv_row & 1) && (is_p || d == 2)) ? 1 : 0;

      double cu[4] = {0., 0., 0., 0.}; // carried top-plane sums of the regular owned nodes
      double rim_carry = 0.;           // ... and of this thread's high-rim entry (rim threads, phase E2)
      // zero the publish scratch once (masked reads of absent neighbours must see finite numbers)
      for (int e = tid; e < L_RIMT - L_EXT; e += NT)
        lds[L_EXT + e] = 0.;
      // Descriptor of the high-rim node entry this thread assembles in phase E2: thread e < 99 = velocity
      // entry (node m = e / 3, component e % 3), thread 128 + e, e < 17 = pressure entry.  Nodes: the north
      // row (i = m, j = ey) for m < TN, then the east column (i = ex, j = m - TN) below the corner, where
      // (ex, ey) = high rim of the VALID cells of the tile (= TN - 1 except in tiles clipped by the domain).
      // Its value in a plane = A + B, A and B being what one or two cells of the last column / row published:
      // .x = LDS offsets of A and B inside a plane block (16 bits each), .y = offset in a dst plane,
      // .z = offset in a slab plane | offset in a z-slab plane << 16, .w = flags: 1 entry exists, 2 / 4 A / B
      // present, 8 seam (partial sum -> slab), 16 on a constrained lateral face, 32 / 64 constrained on the
      // bottom / top face of the domain.
      if (tid < 192) // (all lanes of waves 0..2 read their descriptor in phase E2)
        {
          const bool rp  = tid >= 128;
          const int  e   = rp ? tid - 128 : tid;
          const int  deg = rp ? 1 : 2, nc = rp ? 1 : 3, TN = deg * TX + 1;
          const int  m = e / nc, comp = e - m * nc;
          const int  ex = deg * tcx, ey = deg * tcy;
          const bool north = m < TN;
          const int  ti = north ? m : ex, tj = north ? ey : m - TN;
          bool       ok = m < 2 * TN - 1 && (north ? ti <= ex : tj < ey);
          if (rp ? !(WITH_P) : DIV)
            ok = false;
          const int  c   = (north ? ti : tj) / deg;          // cell index along the rim
          const bool odd = deg == 2 && ((north ? ti : tj) & 1);
          const int  cmax = north ? tcx : tcy;               // cells along the rim
          const bool hasA = c < cmax, hasB = !odd && c >= 1;
          const int  ca = min(c, cmax - 1), cb = max(c - 1, 0);
          const int  cellA = north ? (tcy - 1) * TX + ca : ca * TX + tcx - 1;
          const int  cellB = north ? (tcy - 1) * TX + cb : cb * TX + tcx - 1;
          int        slotA;
          if (rp)
            slotA = north ? 1 : 0;
          else
            slotA = north ? (odd ? 4 : 3) : (odd ? 1 : 0);
          const int offA = rp ? slotA * NCELL + cellA : slotA * (NCELL * 3) + cellA * 3 + comp;
          const int offB = rp ? 2 * NCELL + cellB : 2 * (NCELL * 3) + cellB * 3 + comp;
          const int nn_x = rp ? A.npx : A.nnx, nn_y = rp ? A.npy : A.nny;
          const int I = (rp ? Ip0 : I0) + ti, J = (rp ? Jp0 : J0) + tj;
          const uint32_t con = rp ? A.con_p : A.con_u;
          const int  st = rp ? 1 : 3;
          const bool cx_ = (I == 0 && (con >> (st * 0 + comp) & 1)) || (I == nn_x - 1 && (con >> (st * 1 + comp) & 1));
          const bool cy_ = (J == 0 && (con >> (st * 2 + comp) & 1)) || (J == nn_y - 1 && (con >> (st * 3 + comp) & 1));
          const bool seam = (ti == TN - 1 && I < nn_x - 1) || (tj == TN - 1 && J < nn_y - 1);
          const bool czl = rp ? (A.con_p >> 4 & 1) : (A.con_u >> (12 + comp) & 1);
          const bool czh = rp ? (A.con_p >> 5 & 1) : (A.con_u >> (15 + comp) & 1);
          int4       rd;
          rd.x = offA | (offB << 16);
          rd.y = (J * nn_x + I) * nc + comp;
          rd.z = ((rp ? rim_index<QNX>(min(ti, QNX - 1), min(tj, QNY - 1)) : rim_index<PNX>(min(ti, PNX - 1), min(tj, PNY - 1))) * nc + comp) |
                 (((tj * TN + ti) * nc + comp) << 16);
          rd.w = (ok ? 1 : 0) | (hasA ? 2 : 0) | (hasB ? 4 : 0) | (seam ? 8 : 0) | ((cx_ || cy_) ? 16 : 0) |
                 (czl ? 32 : 0) | (czh ? 64 : 0);
          reinterpret_cast<int4 *>(lds + L_RIMT)[tid] = rd;
        }

      // ---- prologue: first node planes and first state pieces ---------------------------------
      const double2 *state = reinterpret_cast<const double2 *>(A.state);
      double        *ringw = lds + L_RING + wave * RING_ * PIECE_;
      const unsigned slan  = cell * 3 + (is_p ? 0 : d); // my element of a piece
      {
        dma_u_plane_single(A, lds, 2 * cz0, I0, J0, wave, lane);
        dma_u_planes(A, lds, 2 * cz0 + 1, I0, J0, wave, lane);
        if (WITH_P)
          {
            dma_p_plane(A, lds, cz0, Ip0, Jp0, lane);
            dma_p_plane(A, lds, cz0 + 1, Ip0, Jp0, lane);
          }
        if (RES || RCP) // node planes of the old-solution combination (RCP: of the linearisation point) live in the
          {             // (unused) ring area
            dma_u_plane_single(A, lds, 2 * cz0, I0, J0, wave, lane, A.old_u, L_OLDP);
            dma_u_planes(A, lds, 2 * cz0 + 1, I0, J0, wave, lane, A.old_u, L_OLDP);
          }
        if (EXT)
          {
            dma_u_plane_single(A, lds, 2 * cz0, I0, J0, wave, lane, A.ext_u, L_EXTP);
            dma_u_planes(A, lds, 2 * cz0 + 1, I0, J0, wave, lane, A.ext_u, L_EXTP);
          }
      }
      // wave-private exchange records (alias of the publish scratch, which is only live in D/E)
#if defined(Q2_LDS_EXCHANGE)
      double        *xb = lds + L_SCRU + wave * (16 * XQ);
      const int      xq = cq * XQ;
#endif
      const unsigned ring_byte = lds_addr(ringw);
      const unsigned ext_byte  = ring_byte + (L_EXT - L_RING) * 8; // same wave stride as the real ring
      const unsigned piece_voff = 16u * (unsigned)(VARCO ? lane : min(lane, 47));
      auto issue_piece = [&](const int layer_cz, const int p) {
        // piece p (= 2q + half) of cell layer layer_cz: 48 consecutive double2 of this wave
        const double2 *g = state + ((size_t)bt * A.ncz + layer_cz) * A.state_stride +
                           (size_t)p * (4 * PLANES_) + wave * PLANES_;
        const int      sl = p % NS_; // slot: real ring or extension
        dma_b128(g, piece_voff, sl < RING_ ? ring_byte + sl * (PIECE_ * 8) : ext_byte + (sl - RING_) * (PIECE_ * 8),
                 VARCO ? 0xffffffffffffffffull : 0x0000ffffffffffffull);
      };
      // the pieces of a layer that live in extension slots and were not issued by the layer before
      auto issue_burst = [&](const int layer_cz) {
#pragma unroll
        for (int p = 0; p < LA_; ++p)
          if (!RS::real_slot(p))
            issue_piece(layer_cz, p);
      };
      if (RING_ON)
        {
#pragma unroll
          for (int p = 0; p < LA_; ++p)
            if (RS::real_slot(p))
              issue_piece(cz0, p);
          if (NX_ > 0)
            {
              lds_barrier(); // the extension has been zeroed by everybody
              issue_burst(cz0);
            }
        }
      // planes (and, to keep the first layer's counted waits valid whatever was issued here, everything
      // else) must have landed before anybody gathers from them
      wait_vmcnt<0>();
      lds_barrier();
      // residual mode: the state of a point leaves as two 16-byte stores per velocity lane
      const unsigned           sout_voff = 16u * (unsigned)(wave * 48 + cq * 3 + (is_p ? 0 : d));
      // (lanes d < 3; lazy state, round 6: no buffer -- the stores are issued with an EMPTY mask and move nothing.  A sink
      // buffer "that stays in L2" was the first form: its non-temporal stores still wrote 3.7 of the 5.4 GB through to
      // memory, profiles/r06_pmc_res_lazy.txt)
      // (the wait behind the quadrature loop then may not count on them: Q2_RES_WAIT_ALL -- a run-time choice between two
      // counted waits there split the layer body into blocks and the 512-register builds went to 3-5 KB of scratch, 28 ms)
#ifndef Q2_RES_WAIT_ALL
#define Q2_RES_WAIT_ALL 1
#endif
      const unsigned long long sout_mask = A.state_out ? 0x7777777777777777ull : 0ull;
      auto interp_all = [&](double *X) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int b = 0; b < 3; ++b)
            interp3(X[3 * b + 9 * c], X[1 + 3 * b + 9 * c], X[2 + 3 * b + 9 * c], s0, s1, s2);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int a = 0; a < 3; ++a)
            interp3(X[a + 9 * c], X[a + 3 + 9 * c], X[a + 6 + 9 * c], s0, s1, s2);
#pragma unroll
        for (int n = 0; n < 9; ++n)
          interp3(X[n], X[n + 9], X[n + 18], s0, s1, s2);
      };

      // recompute mode with variable coefficients (round 5, second form): lane d of a quad loads ITS coefficient array
      // (rho, mu, damping) for the 27 points of the cell layer and the three values go round the quad by DPP -- 27 instead
      // of 81 loads per lane and layer --, issued in thirds a third ahead (the first third of a layer during the last
      // third of the layer before), so that no load is waited for right after it was issued
#ifndef Q2_RCP_COEF_PIPE
#define Q2_RCP_COEF_PIPE 1
#endif
      // (round 6: also the operator WITHOUT a linearisation state -- explicit convection of two-phase flow -- reads its
      // coefficients this way: there is no state stream for them to ride on)
      constexpr bool CGEN  = VARCO && !RES && !DIV && (RCP || LIN_MODE == 2); // coefficients from the generic arrays
      constexpr bool CPIPE = CGEN && Q2_RCP_COEF_PIPE;
      const double *const coef_arr = CPIPE ? (d == 1 ? A.mu : (d == 2 ? A.damp : A.rho)) : nullptr;
      // (first entry of my cell of layer 0 in the arrays [cell][27], and the step to the next layer)
      const unsigned coef_first = CPIPE ? (unsigned)((((size_t)(TY * by + (valid ? cyl : 0))) * A.ncx + (TX * bx + (valid ? cxl : 0))) * 27) : 0u;
      const unsigned coef_layer = (unsigned)(A.ncy * A.ncx * 27);
      double CPN[9]; // (the first third of the layer to come)
      if (CPIPE)
        {
          const unsigned c0 = coef_first + (unsigned)cz0 * coef_layer;
#pragma unroll
          for (int n = 0; n < 9; ++n)
            CPN[n] = (coef_arr + c0)[n]; // (one per-lane address + immediate offsets)
        }
      for (int layer = 0; layer < nl; ++layer)
        {
          const int cz      = cz0 + layer;
          const int cz_next = layer + 1 < nl ? cz + 1 : cz;
          // the publish scratch is free again: this layer's pieces that live in extension slots
          // (the first layer's were issued in the prologue)
          if (NX_ > 0 && layer > 0)
            issue_burst(cz);

          // recompute mode with variable coefficients: first entry of my cell in the generic arrays [cell][27] (cells
          // beyond the mesh read the tile's first cell: legal address, unused values)
          double CP0[9], CP1[9], CP2[9];
          if (CPIPE)
            {
#pragma unroll
              for (int n = 0; n < 9; ++n)
                CP0[n] = CPN[n];
            }
          const unsigned coef_cell = (CGEN || (RES && VARCO)) ?
                                       (unsigned)((((size_t)cz * A.ncy + (TY * by + (valid ? cyl : 0))) * A.ncx + (TX * bx + (valid ? cxl : 0))) * 27) :
                                       0u;
          double R[27];
          double V2[(RES || RCP) ? 27 : 1]; // second nodal field: old-solution combination (RES) / linearisation point (RCP)
          if (RES || RCP)
            {
              // RES: (w, rho (weight_old u_old + weight_old_old u_old_old)) of the momentum row (:675-686,
              // :730-733): interpolate the nodal combination, it only enters through its values
#pragma unroll
              for (int c = 0; c < 3; ++c)
                {
                  const double *pl = lds + L_OLDP + ((2 * cz + c) % 3) * UPLANE_L + 2 * cyl * UROW + 2 * cxl * 3 + (is_p ? 0 : d);
#pragma unroll
                  for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                      V2[a + 3 * b + 9 * c] = pl[b * UROW + a * 3 + (a == 2 ? fix_last : 0)];
                }
              interp_all(V2);
            }
          // variable-coefficient residual: the layer's coefficients, one array per lane of the quad (issued before the
          // plane copies of the next layer: the compiler's wait for them then leaves those copies in flight)
          double CQ[(RES && VARCO) ? 27 : 1];
          if (RES && VARCO)
            {
              const double *const arr = d == 1 ? A.mu : (d == 2 ? A.damp : A.rho);
#pragma unroll
              for (int n = 0; n < 27; ++n)
                CQ[n] = (arr + coef_cell)[n]; // (one per-lane address + immediate offsets, not 27 index sums)
            }
          double V3[EXT ? 27 : 1]; // third nodal field: the extrapolated velocity
          if (EXT)
            {
#pragma unroll
              for (int c = 0; c < 3; ++c)
                {
                  const double *pl = lds + L_EXTP + ((2 * cz + c) % 3) * UPLANE_L + 2 * cyl * UROW + 2 * cxl * 3 + (is_p ? 0 : d);
#pragma unroll
                  for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                      V3[a + 3 * b + 9 * c] = pl[b * UROW + a * 3 + (a == 2 ? fix_last : 0)];
                }
              interp_all(V3);
            }
          if (RES && !VARCO)
            {
              const double co = is_p ? 0. : A.c_old;
#pragma unroll
              for (int n = 0; n < 27; ++n)
                R[n] = (co * A.wj[(n % 3 == 1) + ((n / 3) % 3 == 1) + (n / 9 == 1)]) * V2[n];
            }
          else
            {
#pragma unroll
              for (int n = 0; n < 27; ++n)
                R[n] = 0.;
            }
          // ---- B: gather my 27 (8) values from the LDS node planes ---------------------
          double V[27];
          if (!is_p)
            {
#pragma unroll
              for (int c = 0; c < 3; ++c)
                {
                  const double *pl = lds + L_UPL + ((2 * cz + c) % 3) * UPLANE_L + 2 * cyl * UROW + 2 * cxl * 3 + d;
#pragma unroll
                  for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                      V[a + 3 * b + 9 * c] = pl[b * UROW + a * 3 + (a == 2 ? fix_last : 0)];
                }
            }
          else
            {
              // Q1 -> Q2 nodal expansion (mid nodes = averages)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                {
                  const double *pl = lds + L_PPL + ((cz + c) % 2) * PPLANE_L + cyl * PROW + cxl;
#pragma unroll
                  for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                      V[2 * a + 6 * b + 18 * c] = WITH_P ? pl[b * PROW + a + (a == 1 ? fix_last : 0)] : 0.;
                }
            }
          // read_dof_values: constrained entries read as zero (boundary tiles / layers only)
          {
            const bool zlo = !RES && conz_lo && cz == 0, zhi = !RES && conz_hi && cz == A.ncz - 1;
            if (__builtin_amdgcn_readfirstlane(__any(m_zero != 0u || zlo || zhi)))
              {
                const int deg = is_p ? 1 : 2;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                  for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                      {
                        // pressure lane: only the corner positions hold values at this point
                        const int  li = is_p ? a / 2 : a, lj = is_p ? b / 2 : b;
                        const bool z  = (m_zero >> (li + 3 * lj) & 1u) || (zlo && c == 0) || (zhi && c == 2);
                        (void)deg;
                        if (z)
                          V[a + 3 * b + 9 * c] = 0.;
                      }
              }
          }
          if (is_p)
            {
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
          // every wave has finished reading planes 2cz, 2cz+1: refill their slots for the next layer
          lds_barrier();
          // (issued for the last layer of the chunk too: fixed VMEM op count, see wait_vmcnt uses)
          dma_u_planes(A, lds, 2 * cz + 3, I0, J0, wave, lane);
          if (WITH_P)
            dma_p_plane(A, lds, cz + 2, Ip0, Jp0, lane);
          if (RES || RCP)
            dma_u_planes(A, lds, 2 * cz + 3, I0, J0, wave, lane, A.old_u, L_OLDP);
          if (EXT)
            dma_u_planes(A, lds, 2 * cz + 3, I0, J0, wave, lane, A.ext_u, L_EXTP);

          // ---- C: interpolate to the Gauss points (in place) ----------------------------
          interp_all(V);

          // ---- quadrature-point loop (source/navier_stokes_matrix.cc:702-893) ------------
          const double2 *sout_layer = reinterpret_cast<const double2 *>(A.state_out) +
                                      ((size_t)bt * A.ncz + cz) * A.state_stride;

          // (three thirds of nine points: the coefficient loads of the recompute mode are issued between them)
#pragma unroll
          for (int third = 0; third < 3; ++third)
          {
          if (CPIPE)
            {
              // the third to come: 9..17 and 18..26 of this layer, 0..8 of the next (the last layer of the chunk fetches
              // its own once more: legal addresses, unused values)
              const unsigned cn = third == 2 ? coef_first + (unsigned)cz_next * coef_layer : coef_cell + 9u * (unsigned)(third + 1);
#pragma unroll
              for (int n = 0; n < 9; ++n)
                {
                  const double v = (coef_arr + cn)[n];
                  if (third == 0)
                    CP1[n] = v;
                  else if (third == 1)
                    CP2[n] = v;
                  else
                    CPN[n] = v;
                }
            }
#pragma unroll
          for (int q = 9 * third; q < 9 * third + 9; ++q)
            {
              const int qx = q % 3, qy = (q / 3) % 3, qz = q / 9;
              double2   st0 = make_double2(0., 0.), st1 = make_double2(0., 0.);
              double    r_ub0 = 0., r_ub1 = 0., r_ub2 = 0., r_trl = 0.;
              double    r_rho = 0., r_mu = 0., r_damp = 0.;
              if (RING_ON)
                {
                  // counted wait: everything but the operations issued after the pieces of this point
                  // (RingSched::younger)
                  constexpr int npl_ = NPL_U + (WITH_P ? NPL_P : 0);
#if !defined(Q2_EXP) || (Q2_EXP != 1 && Q2_EXP != 2)
                  // (q is a constant once the loop is unrolled: the count folds to an immediate)
                  // (no stores assumed: a wave of a clipped tile may own no valid cell, and a store with an empty
                  // EXEC mask may not count)
                  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(RingWaits<RING_, NX_, npl_, 0>::table.v[q]) : "memory");
#endif
                  const int sl0 = (2 * q) % NS_, sl1 = (2 * q + 1) % NS_;
                  // (one base register: the extension uses the wave stride of the real ring)
                  const double *pc0 = ringw + (sl0 < RING_ ? sl0 * PIECE_ : (L_EXT - L_RING) + (sl0 - RING_) * PIECE_);
                  const double *pc1 = ringw + (sl1 < RING_ ? sl1 * PIECE_ : (L_EXT - L_RING) + (sl1 - RING_) * PIECE_);
                  st0 = reinterpret_cast<const double2 *>(pc0)[slan - wave * 48];
                  st1 = reinterpret_cast<const double2 *>(pc1)[slan - wave * 48];
                  // the quad's three (u_lin_e, grad_e0) and (grad_e1, grad_e2) entries
                  const double *rq0 = pc0 + 6 * cq;
                  const double *rq1 = pc1 + 6 * cq;
                  if (VARCO) // quad-uniform reads of the cell's (rho, mu) and damping
                    {
                      const double *rc0 = pc0 + 96 + 2 * cq;
                      const double *rc1 = pc1 + 96 + 2 * cq;
                      r_rho = rc0[0], r_mu = rc0[1], r_damp = rc1[0];
                    }
                  if (LIN_MODE == 0)
                    {
                      r_ub0 = rq0[0], r_ub1 = rq0[2], r_ub2 = rq0[4];
                      r_trl = rq0[1] + rq1[2] + rq1[5];
                    }
                  else
                    r_ub0 = rq0[0], r_ub1 = rq0[2], r_ub2 = rq0[4];
                  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                  // refill the two slots that became free (after the last layer of the chunk
                  // the same pieces are harmlessly fetched again: fixed VMEM op count)
#if !defined(Q2_EXP) || Q2_EXP != 2
#pragma unroll
                  for (int j = 0; j < 2; ++j)
                    {
                      const int t = 2 * q + LA_ + j;
                      if (t < 54)
                        issue_piece(cz, t);
                      else if (RS::real_slot(t - 54))
                        issue_piece(cz_next, t - 54);
                    }
#endif
                }
#if defined(Q2_EXP) && Q2_EXP == 4
              // diagnostic: stream only -- consume the state minimally, skip the arithmetic
              R[q] += st0.x + st1.y + r_trl + r_ub0 + r_ub1 + r_ub2;
              continue;
#endif
              const double Vq = V[q];
              // reference-cell derivatives by the collocation derivative, then J^{-T}
              constexpr int e1 = ISO ? 0 : 1, e2 = ISO ? 0 : 2;
              const double  g0 = dline(qx, V[0 + 3 * qy + 9 * qz], V[1 + 3 * qy + 9 * qz], V[2 + 3 * qy + 9 * qz],
                                       A.ah[0][0], A.ah[0][1], A.ah[0][2], A.ah[0][3]);
              const double  g1 = dline(qy, V[qx + 9 * qz], V[qx + 3 + 9 * qz], V[qx + 6 + 9 * qz],
                                       A.ah[e1][0], A.ah[e1][1], A.ah[e1][2], A.ah[e1][3]);
              const double  g2 = dline(qz, V[qx + 3 * qy], V[qx + 3 * qy + 9], V[qx + 3 * qy + 18],
                                       A.ah[e2][0], A.ah[e2][1], A.ah[e2][2], A.ah[e2][3]);

              if (CGEN && !CPIPE)
                {
                  const unsigned cq_ = coef_cell + (unsigned)q;
                  r_rho = A.rho[cq_], r_mu = A.mu[cq_], r_damp = A.damp[cq_];
                }
              if (RES && VARCO)
                r_rho = quad_bcast<0, 0, QG8>(CQ[q]), r_mu = quad_bcast<1, 0, QG8>(CQ[q]), r_damp = quad_bcast<2, 0, QG8>(CQ[q]);
              if (CPIPE)
                {
                  const double cq_ = q < 9 ? CP0[q % 9] : (q < 18 ? CP1[q % 9] : CP2[q % 9]);
                  r_rho = quad_bcast<0, 0, QG8>(cq_), r_mu = quad_bcast<1, 0, QG8>(cq_), r_damp = quad_bcast<2, 0, QG8>(cq_);
                }
              if (RCP)
                {
                  // the state of this point from the interpolated nodal linearisation point: my component's value and
                  // gradient row (what the ring delivers as st0, st1), the other components' values and the trace of
                  // the gradient through the quad
                  const double b0 = dline(qx, V2[0 + 3 * qy + 9 * qz], V2[1 + 3 * qy + 9 * qz], V2[2 + 3 * qy + 9 * qz],
                                          A.ah[0][0], A.ah[0][1], A.ah[0][2], A.ah[0][3]);
                  const double b1 = dline(qy, V2[qx + 9 * qz], V2[qx + 3 + 9 * qz], V2[qx + 6 + 9 * qz],
                                          A.ah[e1][0], A.ah[e1][1], A.ah[e1][2], A.ah[e1][3]);
                  const double b2 = dline(qz, V2[qx + 3 * qy], V2[qx + 3 * qy + 9], V2[qx + 3 * qy + 18],
                                          A.ah[e2][0], A.ah[e2][1], A.ah[e2][2], A.ah[e2][3]);
                  const double vb = V2[q];
                  st0   = make_double2(vb, b0);
                  st1   = make_double2(b1, b2);
                  r_ub0 = quad_bcast<0, 2, QG2>(vb), r_ub1 = quad_bcast<1, 2, QG2>(vb), r_ub2 = quad_bcast<2, 2, QG2>(vb);
                  r_trl = quad_bcast<0, 3, QG2>(b0) + quad_bcast<1, 3, QG2>(b1) + quad_bcast<2, 3, QG2>(b2);
                }
#if !defined(Q2_LDS_EXCHANGE)
              // gradient rows of the three velocity components, visible to all four lanes (DPP)
              const double G00 = quad_bcast<0, 0, QG1>(g0), G01 = quad_bcast<0, 0, QG1>(g1), G02 = quad_bcast<0, 0, QG1>(g2);
              const double G10 = quad_bcast<1, 0, QG1>(g0), G11 = quad_bcast<1, 0, QG1>(g1), G12 = quad_bcast<1, 0, QG1>(g2);
              const double G20 = quad_bcast<2, 0, QG1>(g0), G21 = quad_bcast<2, 0, QG1>(g1), G22 = quad_bcast<2, 0, QG1>(g2);
              const double c0 = sel3(d, G00, G01, G02), c1 = sel3(d, G10, G11, G12), c2 = sel3(d, G20, G21, G22);
              const double u0 = quad_bcast<0, 1, QG1>(Vq), u1 = quad_bcast<1, 1, QG1>(Vq), u2 = quad_bcast<2, 1, QG1>(Vq);
              const double pres = quad_bcast<3, 1, QG1>(Vq);
#else
              // (measured alternative, 17 % slower: exposed LDS latency per point + one more barrier)
              // Exchange inside the quad through a wave-private LDS record [lane d][g0 g1 g2 v]:
              // one wave's LDS operations execute in order, so the reads below see this
              // iteration's writes of all four lanes without any barrier.  Lane d picks column d
              // of the velocity gradient with a lane-dependent ADDRESS instead of register selects.
              {
                double2 *w2 = reinterpret_cast<double2 *>(xb + xq + 4 * d);
                w2[0] = make_double2(g0, g1);
                w2[1] = make_double2(g2, Vq);
              }
              const double c0 = xb[xq + d], c1 = xb[xq + 4 + d], c2 = xb[xq + 8 + d];
              const double G00 = xb[xq], G11 = xb[xq + 5], G22 = xb[xq + 10];
              const double u0 = xb[xq + 3], u1 = xb[xq + 7], u2 = xb[xq + 11];
              const double pres = xb[xq + 15];
#endif
              const double div = G00 + G11 + G22; // :706
              (void)u0, (void)u1, (void)u2, (void)pres;

              // :717, :827-835, :841-845 with the coefficients of this point
              const double cA_q = VARCO ? A.gamma * r_rho - r_damp : A.cA;
              const double cB_q = VARCO ? A.tau1 * r_rho : A.cB;
              const double tmu_q = VARCO ? (is_p ? 0. : A.tau1 * r_mu) : tmu_l;
              double conv = cA_q * Vq;
              if (RES && VARCO) // :727-732 times the density of the point (A.c_old: 1 with a time derivative, else 0)
                conv += (A.c_old * r_rho) * V2[q];
              if (RES && EXT)
                {
                  // :740-782 value and gradient row of my component of the extrapolated velocity, the other components'
                  // values and div u_ext through the quad
                  const double b0 = dline(qx, V3[0 + 3 * qy + 9 * qz], V3[1 + 3 * qy + 9 * qz], V3[2 + 3 * qy + 9 * qz],
                                          A.ah[0][0], A.ah[0][1], A.ah[0][2], A.ah[0][3]);
                  const double b1 = dline(qy, V3[qx + 9 * qz], V3[qx + 3 + 9 * qz], V3[qx + 6 + 9 * qz],
                                          A.ah[e1][0], A.ah[e1][1], A.ah[e1][2], A.ah[e1][3]);
                  const double b2 = dline(qz, V3[qx + 3 * qy], V3[qx + 3 * qy + 9], V3[qx + 3 * qy + 18],
                                          A.ah[e2][0], A.ah[e2][1], A.ah[e2][2], A.ah[e2][3]);
                  const double vb = V3[q];
                  const double ov0 = quad_bcast<0, 0, QG2>(vb), ov1 = quad_bcast<1, 0, QG2>(vb), ov2 = quad_bcast<2, 0, QG2>(vb);
                  const double ediv = quad_bcast<0, 0, QG2>(b0) + quad_bcast<1, 0, QG2>(b1) + quad_bcast<2, 0, QG2>(b2);
                  if (LIN_MODE == 2) // explicit: the extrapolated field convects itself
                    conv += cB_q * (A.beta * ediv * vb + ov0 * b0 + ov1 * b1 + ov2 * b2);
                  else // semi-implicit: it convects the solution; (u_ext, div u_ext) is the state of the vmults
                    {
                      conv += cB_q * (A.beta * ediv * Vq + ov0 * g0 + ov1 * g1 + ov2 * g2);
                      const double2 *sp = sout_layer + (size_t)(2 * q) * (4 * 48);
                      store_b128_masked(sp, sout_voff, vb, ediv, sout_mask);
                      store_b128_masked(sp + 4 * 48, sout_voff, ediv, ediv, sout_mask);
                    }
                }
              else if (RES)
                {
                  if (LIN_MODE != 2)
                    {
                      // :725-760 nonlinear term of the solution itself; the point's state goes out in
                      // the streaming layout: (u_d, d_0 u_d | d_1 u_d, d_2 u_d) or (u_d, div u | ...)
                      conv += cB_q * (A.beta * div * Vq + u0 * g0 + u1 * g1 + u2 * g2);
                      const double2 *sp = sout_layer + (size_t)(2 * q) * (4 * 48);
                      store_b128_masked(sp, sout_voff, Vq, LIN_MODE == 0 ? g0 : div, sout_mask);
                      store_b128_masked(sp + 4 * 48, sout_voff, LIN_MODE == 0 ? g1 : div, LIN_MODE == 0 ? g2 : div,
                                        sout_mask);
                    }
                }
              else if (LIN_MODE == 0)       // Newton :802-816
                {
                  // u_lin of all components and tr(grad u_lin): quad-uniform LDS reads of the ring
                  const double ub0 = r_ub0, ub1 = r_ub1, ub2 = r_ub2, trl = r_trl;
                  double       res = A.beta * (div * st0.x + trl * Vq);
                  res += ub0 * g0 + u0 * st0.y;
                  res += ub1 * g1 + u1 * st1.x;
                  res += ub2 * g2 + u2 * st1.y;
                  conv += cB_q * res;
                }
              else if (LIN_MODE == 1) // Picard-type :817-826, state = (u_lin, div_lin)
                {
                  // (recompute mode, round 6: div u_lin is the trace formed through the quad -- for the schemes that linearise
                  // about the extrapolated velocity the nodal field is that extrapolation)
                  const double ub0 = r_ub0, ub1 = r_ub1, ub2 = r_ub2;
                  double       res = (A.beta * (RCP ? r_trl : st0.y)) * Vq;
                  res += ub0 * g0;
                  res += ub1 * g1;
                  res += ub2 * g2;
                  conv += cB_q * res;
                }

              const double jxw = A.wj[(qx == 1) + (qy == 1) + (qz == 1)];
              if (DIV)
                {
                  R[q] += (is_p ? A.c_div * div : 0.) * jxw;
                  continue;
                }
              // c_e = column d of the velocity gradient (transpose part of the symmetric gradient)
              double diag = A.tau_gd * div;
              if (WITH_P)
                diag -= pres;
              diag *= jxw;
              // :859-892: row d of tmu (grad u + grad u^T) + (tau_gd div - p) I, times JxW J^{-1}
              const double tmj = tmu_q * jxw;
              const double tg0 = tmj * (g0 + c0) + d0 * diag;
              const double tg1 = tmj * (g1 + c1) + d1 * diag;
              const double tg2 = tmj * (g2 + c2) + d2 * diag;
              // test value: momentum rows (:837) or the pressure row (q, -div u) (:853-856)
              const double tv = (is_p ? -div : conv) * jxw;

              // integrate (collocation derivative transposed), accumulate at the Gauss points
              R[q] += tv;
              dline_t(qx, tg0, R[0 + 3 * qy + 9 * qz], R[1 + 3 * qy + 9 * qz], R[2 + 3 * qy + 9 * qz],
                      A.ah[0][0], A.ah[0][1], A.ah[0][2], A.ah[0][3]);
              dline_t(qy, tg1, R[qx + 9 * qz], R[qx + 3 + 9 * qz], R[qx + 6 + 9 * qz],
                      A.ah[e1][0], A.ah[e1][1], A.ah[e1][2], A.ah[e1][3]);
              dline_t(qz, tg2, R[qx + 3 * qy], R[qx + 3 * qy + 9], R[qx + 3 * qy + 18],
                      A.ah[e2][0], A.ah[e2][1], A.ah[e2][2], A.ah[e2][3]);
            }
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

          // my plane copies for the next layer are older than the AHEAD pieces issued last (residual
          // mode: than the 54 state stores of this layer)
          if (RING_ON)
            {
              // (the plane copies are older than the pieces of the next layer issued by the last points)
              constexpr int tail = RS::n_issued(26) + RS::n_issued(25) + RS::n_issued(24) + RS::n_issued(23) +
                                   RS::n_issued(22) + RS::n_issued(21) + RS::n_issued(20) + RS::n_issued(19) + RS::n_issued(18);
              constexpr int last9 = NX_ > 0 ? tail : LA_;
              wait_vmcnt<(last9 < LA_ ? last9 : LA_)>();
              // every wave must have consumed its pieces in the extension (= publish scratch) before anybody publishes
              if (NX_ > 0)
                lds_barrier();
            }
          else if (RES && LIN_MODE != 2 && !Q2_RES_WAIT_ALL)
            wait_vmcnt<54>(); // (rounds 4-5: the plane copies are older than the 54 state stores of this layer)
          else
            wait_vmcnt<0>();

#if defined(Q2_LDS_EXCHANGE)
          // the exchange records alias the publish scratch: every wave must have left its
          // quadrature loop before anybody publishes
          lds_barrier();
#endif
          // ---- D: publish what the east / north neighbour cells need ----------------------
          // velocity: local (2,0) (2,1) (2,2) (0,2) (1,2) of every plane -> slots 0..4
          if (DIV && !is_p)
            {
            }
          else if (!is_p)
            {
              double *sc = lds + L_SCRU + cell * 3 + d;
#pragma unroll
              for (int lk = 0; lk < 3; ++lk)
                {
                  sc[(lk * 5 + 0) * (NCELL * 3)] = R[2 + 0 + 9 * lk];
                  sc[(lk * 5 + 1) * (NCELL * 3)] = R[2 + 3 + 9 * lk];
                  sc[(lk * 5 + 2) * (NCELL * 3)] = R[2 + 6 + 9 * lk];
                  sc[(lk * 5 + 3) * (NCELL * 3)] = R[0 + 6 + 9 * lk];
                  sc[(lk * 5 + 4) * (NCELL * 3)] = R[1 + 6 + 9 * lk];
                }
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
              // pressure: local (1,0) (0,1) (1,1) of both planes -> slots 0..2
              double *sc = lds + L_SCRP + cell;
#pragma unroll
              for (int lk = 0; lk < 2; ++lk)
                {
                  sc[(lk * 3 + 0) * NCELL] = R[2 + 0 + 18 * lk];
                  sc[(lk * 3 + 1) * NCELL] = R[0 + 6 + 18 * lk];
                  sc[(lk * 3 + 2) * NCELL] = R[2 + 6 + 18 * lk];
                }
            }
          lds_barrier();
#if defined(Q2_SKIP_E)
          // diagnostic: no combine / no stores (the integrated values stay live through a fake use)
          {
            double acc = 0.;
#pragma unroll
            for (int n = 0; n < 27; ++n)
              acc += R[n];
            if (acc == 1.2345e300)
              A.dst_u[0] = acc;
            lds_barrier();
            continue;
          }
#endif

          // ---- E: combine per owned node and write the two finished planes -----------------
          // (round 3: straight-line code.  The old version cost 0.34 ms of a 1.34 ms kernel at 128^3 --
          // measured with -DQ2_SKIP_E -- for ~0.1 ms worth of stores: per-node branches, one LDS
          // round trip per neighbour value, 64-bit address arithmetic.)
          // E1: every lane finishes the REGULAR nodes of its cell, local (li, lj) in {0,1}^2 (pressure:
          //     (0,0)): own value + what the west / south / south-west cells published, all LDS reads
          //     issued together, absent neighbours masked by 0/1 factors in the FMAs;
          // E2: the nodes on the HIGH rim of the tile (partial sums for the seam slabs, or final values on
          //     the domain boundary) are assembled by dedicated "rim threads" from the same published
          //     values, one node entry per thread, descriptors in LDS (L_RIMT), carry in one register.
          // Re-derive the lane coordinates from an opaque copy of the thread id: otherwise the
          // compiler hoists all per-lane address arithmetic of this phase out of the layer loop
          // and spills it around the quadrature loop (scratch reloads = s_waitcnt vmcnt(0) = a
          // drained DMA queue every layer).
          int tid_e = threadIdx.x;
          asm volatile("" : "+v"(tid_e));
          const int  d = tid_e & 3, cell = tid_e >> 2; // cell = cyl * TX + cxl
          const int  cxl = cell & 7, cyl = cell >> 3;
          const bool is_p = d == 3;
          const bool valid = cxl < tcx && cyl < tcy;
          const bool hasW = cxl > 0, hasS = cyl > 0;
          const unsigned long long vmask_u = __ballot(valid && !is_p), vmask_p = __ballot(valid && is_p);
          // constrained rows among the regular nodes: only on the low domain faces (the high faces are rim
          // nodes) and on the bottom plane; wave-uniform test, boundary tiles only
          const bool zc_lo = cz == 0 && (((A.con_u >> 12) & 7u) != 0u || ((A.con_p >> 4) & 1u) != 0u);
          const bool slow  = con_lat_tile || zc_lo;
          if (DIV && !is_p)
            {
            }
          else if (!is_p)
            {
              const double *sc = lds + L_SCRU + cell * 3 + d;
              // (summed into R in place: nothing but R is live across the stores below)
#pragma unroll
              for (int lk = 0; lk < 3; ++lk)
                {
                  const double w0 = sc[(lk * 5 + 0) * (NCELL * 3) - 3];  // west cell's (2,0) = my (0,0)
                  const double w1 = sc[(lk * 5 + 1) * (NCELL * 3) - 3];  //             (2,1) = my (0,1)
                  const double t0 = sc[(lk * 5 + 3) * (NCELL * 3) - 24]; // south cell's (0,2) = my (0,0)
                  const double t1 = sc[(lk * 5 + 4) * (NCELL * 3) - 24]; //              (1,2) = my (1,0)
                  const double sw = sc[(lk * 5 + 2) * (NCELL * 3) - 27]; // south-west cell's (2,2) = my (0,0)
                  // (selects, not 0/1 factors: what an absent neighbour's slot holds may be anything -- the cells
                  // of a clipped tile beyond the mesh publish sums of uninitialised LDS)
                  R[0 + 9 * lk] = ((R[0 + 9 * lk] + (hasW ? w0 : 0.)) + (hasS ? t0 : 0.)) + (hasW && hasS ? sw : 0.);
                  R[1 + 9 * lk] += hasS ? t1 : 0.;
                  R[3 + 9 * lk] += hasW ? w1 : 0.;
                }
#if defined(Q2_DST_WRAP)
              const unsigned voff = (8u * (unsigned)(lane_g + d)) & (Q2_DST_WRAP - 1u);
#else
              const unsigned voff = 8u * (unsigned)(lane_g + d); // (lane_g - d) + 2 d
#endif
#pragma unroll
              for (int lk = 0; lk < 3; ++lk)
                {
                  double a0 = R[0 + 9 * lk], a1 = R[1 + 9 * lk], a3 = R[3 + 9 * lk], a4 = R[4 + 9 * lk];
                  if (lk == 2)
                    {
                      // top plane: finished only after the next layer -> carry
                      cu[0] = a0, cu[1] = a1, cu[2] = a3, cu[3] = a4;
                    }
                  else
                    {
                      if (lk == 0)
                        a0 += cu[0], a1 += cu[1], a3 += cu[2], a4 += cu[3];
                      const int     K  = 2 * cz + lk;
                      const double *sp = A.src_u + (size_t)K * A.nny * A.nnx * 3;
                      double       *dp = A.dst_u + (size_t)K * A.nny * A.nnx * 3;
                      if (slow)
                        {
                          // constrained rows carry +src (:247-256), 0 in the residual
                          const bool zc = lk == 0 && cz == 0 && ((A.con_u >> (12 + d)) & 1u);
                          const bool cw = cxl == 0 && I0 == 0 && ((A.con_u >> d) & 1u);
                          const bool cs = cyl == 0 && J0 == 0 && ((A.con_u >> (6 + d)) & 1u);
                          if (valid && (zc || cw || cs))
                            a0 = RES ? 0. : load_now(sp + lane_g);
                          if (valid && (zc || cs))
                            a1 = RES ? 0. : load_now(sp + lane_g + 3);
                          if (valid && (zc || cw))
                            a3 = RES ? 0. : load_now(sp + lane_g + (unsigned)(A.nnx * 3));
                          if (valid && zc)
                            a4 = RES ? 0. : load_now(sp + lane_g + (unsigned)(A.nnx * 3 + 3));
                        }
                      // the quad's 2 nodes x 3 components of a row are 48 contiguous bytes; regroup them
                      // inside the quad so that every lane stores 16 B (scalar row base + lane offset)
                      // lane 0 stores (x_0, x_1) of the first node, lane 1 (x_2 of the first, y_0 of the second), lane 2 (y_1, y_2):
                      // the SENDING lane picks which of its two values travels (one select), one permutation delivers it --
                      // two selects + two permutations per store instead of six broadcasts + two three-way selects
                      // (round 6: 20 -> 8 instructions per store, four stores per layer)
                      {
                        const double s1 = d == 1 ? a1 : a0, s2 = d == 1 ? a0 : a1;
                        store_b128_dst(dp, voff, quad_permute<0xD8, QG4>(s1), quad_permute<0xE1, QG4>(s2), vmask_u);
                      }
                      {
                        const double s1 = d == 1 ? a4 : a3, s2 = d == 1 ? a3 : a4;
                        store_b128_dst(dp + A.nnx * 3, voff, quad_permute<0xD8, QG4>(s1), quad_permute<0xE1, QG4>(s2), vmask_u);
                      }
                    }
                }
            }
          else if (WITH_P && A.integrate_p)
            {
              const double *sc = lds + L_SCRP + cell;
              double        w0[2], t0[2], sw[2];
#pragma unroll
              for (int lk = 0; lk < 2; ++lk)
                {
                  w0[lk] = sc[(lk * 3 + 0) * NCELL - 1]; // W (1,0)
                  t0[lk] = sc[(lk * 3 + 1) * NCELL - 8]; // S (0,1)
                  sw[lk] = sc[(lk * 3 + 2) * NCELL - 9]; // SW (1,1)
                }
              double a0 = (((R[0] + (hasW ? w0[0] : 0.)) + (hasS ? t0[0] : 0.)) + (hasW && hasS ? sw[0] : 0.)) + cu[0];
              cu[0]     = ((R[18] + (hasW ? w0[1] : 0.)) + (hasS ? t0[1] : 0.)) + (hasW && hasS ? sw[1] : 0.);
              const double *sp = A.src_p + (size_t)cz * A.npy * A.npx;
              double       *dp = A.dst_p + (size_t)cz * A.npy * A.npx;
              if (slow)
                {
                  const bool con = (cz == 0 && ((A.con_p >> 4) & 1u)) || (cxl == 0 && Ip0 == 0 && (A.con_p & 1u)) ||
                                   (cyl == 0 && Jp0 == 0 && ((A.con_p >> 2) & 1u));
                  if (valid && con)
                    a0 = RES ? 0. : -load_now(sp + lane_g); // -1 on the pressure block of vmult
                }
              store_b64_masked(dp, 8u * lane_g, a0, vmask_p);
            }

          // ---- E2: rim threads (waves 0, 1: velocity entries; wave 2: pressure entries) ----------
          if (wave < 2 && !DIV)
            {
              const int4 rd = reinterpret_cast<const int4 *>(lds + L_RIMT)[tid_e];
              const int  fl = rd.w;
              const double *pa = lds + L_SCRU + (rd.x & 0xffff), *pb = lds + L_SCRU + (rd.x >> 16);
              const double x0 = pa[0], y0 = pb[0], x1 = pa[5 * NCELL * 3], y1 = pb[5 * NCELL * 3];
              const double x2 = pa[10 * NCELL * 3], y2 = pb[10 * NCELL * 3];
              const bool   mA = fl & 2, mB = fl & 4;
              const double v0 = ((mA ? x0 : 0.) + (mB ? y0 : 0.)) + rim_carry;
              const double v1 = (mA ? x1 : 0.) + (mB ? y1 : 0.);
              rim_carry       = (mA ? x2 : 0.) + (mB ? y2 : 0.);
              if (fl & 1)
                {
#pragma unroll
                  for (int lk = 0; lk < 2; ++lk)
                    {
                      const int    K   = 2 * cz + lk;
                      const size_t idx = (size_t)K * A.nny * A.nnx * 3 + (unsigned)rd.y;
                      const double v   = lk == 0 ? v0 : v1;
                      if ((fl & 16) || (K == 0 && (fl & 32)))
                        Q2_STORE(A.dst_u[idx], RES ? 0. : load_now(A.src_u + idx));
                      else if (fl & 8)
                        Q2_STORE_SLAB(A.slab_u[(wgs * (2 * A.LZ + 1) + 2 * layer + lk) * (RIM_U * 3) + (rd.z & 0xffff)], v);
                      else
                        Q2_STORE(A.dst_u[idx], v);
                    }
                }
            }
          else if (wave == 2 && WITH_P && A.integrate_p)
            {
              const int4 rd = reinterpret_cast<const int4 *>(lds + L_RIMT)[tid_e];
              const int  fl = rd.w;
              const double *pa = lds + L_SCRP + (rd.x & 0xffff), *pb = lds + L_SCRP + (rd.x >> 16);
              const double x0 = pa[0], y0 = pb[0], x1 = pa[3 * NCELL], y1 = pb[3 * NCELL];
              const bool   mA = fl & 2, mB = fl & 4;
              const double v0 = ((mA ? x0 : 0.) + (mB ? y0 : 0.)) + rim_carry;
              rim_carry       = (mA ? x1 : 0.) + (mB ? y1 : 0.);
              if (fl & 1)
                {
                  const size_t idx = (size_t)cz * A.npy * A.npx + (unsigned)rd.y;
                  if ((fl & 16) || (cz == 0 && (fl & 32)))
                    Q2_STORE(A.dst_p[idx], RES ? 0. : -load_now(A.src_p + idx));
                  else if (fl & 8)
                    Q2_STORE_SLAB(A.slab_p[(wgs * (A.LZ + 1) + layer) * RIM_P + (rd.z & 0xffff)], v0);
                  else
                    Q2_STORE(A.dst_p[idx], v0);
                }
            }
          // scratch may be overwritten by the next layer only after everybody is done
          lds_barrier();
        }

      // ---- top plane of the chunk --------------------------------------------------------------
      {
        const int  cze   = cz0 + nl;
        const bool zseam = cze < A.ncz;
        const int  cxl = cell & 7, cyl = cell >> 3;
        const bool valid = cxl < tcx && cyl < tcy;
        if (DIV && !is_p)
          {
          }
        else if (!is_p)
          {
            const int    K     = 2 * cze;
            const bool   zcon  = K == A.nnz - 1 && ((A.con_u >> (15 + d)) & 1u);
            const bool   cw    = cxl == 0 && I0 == 0 && ((A.con_u >> d) & 1u);
            const bool   cs    = cyl == 0 && J0 == 0 && ((A.con_u >> (6 + d)) & 1u);
            const size_t pbase = (size_t)K * A.nny * A.nnx * 3;
#pragma unroll
            for (int n = 0; n < 4; ++n)
              if (valid)
                {
                  const int    li = n & 1, lj = n >> 1;
                  const size_t idx = pbase + lane_g + (unsigned)((lj * A.nnx + li) * 3);
                  if (zcon || (li == 0 && cw) || (lj == 0 && cs))
                    Q2_STORE(A.dst_u[idx], RES ? 0. : load_now(A.src_u + idx));
                  else if (zseam)
                    Q2_STORE_SLAB(A.zslab_u[(wgs * (PNX * PNY) + (2 * cyl + lj) * PNX + 2 * cxl + li) * 3 + d], cu[n]);
                  else
                    Q2_STORE(A.dst_u[idx], cu[n]);
                }
          }
        else if (WITH_P && A.integrate_p)
          {
            const int    K    = cze;
            const bool   con  = (K == A.npz - 1 && ((A.con_p >> 5) & 1u)) || (cxl == 0 && Ip0 == 0 && (A.con_p & 1u)) ||
                                (cyl == 0 && Jp0 == 0 && ((A.con_p >> 2) & 1u));
            const size_t idx  = (size_t)K * A.npy * A.npx + lane_g;
            if (valid)
              {
                if (con)
                  Q2_STORE(A.dst_p[idx], RES ? 0. : -load_now(A.src_p + idx));
                else if (zseam)
                  Q2_STORE_SLAB(A.zslab_p[wgs * PPLANE + cyl * QNX + cxl], cu[0]);
                else
                  Q2_STORE(A.dst_p[idx], cu[0]);
              }
          }
        // rim entries of the top plane
        const bool rim_u = wave < 2 && !DIV, rim_p = wave == 2 && WITH_P && A.integrate_p;
        if (rim_u || rim_p)
          {
            const int4 rd = reinterpret_cast<const int4 *>(lds + L_RIMT)[tid];
            const int  fl = rd.w;
            if (fl & 1)
              {
                if (rim_u)
                  {
                    const int    K   = 2 * cze;
                    const size_t idx = (size_t)K * A.nny * A.nnx * 3 + (unsigned)rd.y;
                    if ((fl & 16) || (K == A.nnz - 1 && (fl & 64)))
                      Q2_STORE(A.dst_u[idx], RES ? 0. : load_now(A.src_u + idx));
                    else if (fl & 8)
                      Q2_STORE_SLAB(A.slab_u[(wgs * (2 * A.LZ + 1) + 2 * nl) * (RIM_U * 3) + (rd.z & 0xffff)], rim_carry);
                    else if (zseam)
                      Q2_STORE_SLAB(A.zslab_u[wgs * (PNX * PNY * 3) + (rd.z >> 16)], rim_carry);
                    else
                      Q2_STORE(A.dst_u[idx], rim_carry);
                  }
                else
                  {
                    const int    K   = cze;
                    const size_t idx = (size_t)K * A.npy * A.npx + (unsigned)rd.y;
                    if ((fl & 16) || (K == A.npz - 1 && (fl & 64)))
                      Q2_STORE(A.dst_p[idx], RES ? 0. : -load_now(A.src_p + idx));
                    else if (fl & 8)
                      Q2_STORE_SLAB(A.slab_p[(wgs * (A.LZ + 1) + nl) * RIM_P + (rd.z & 0xffff)], rim_carry);
                    else if (zseam)
                      Q2_STORE_SLAB(A.zslab_p[wgs * PPLANE + (rd.z >> 16)], rim_carry);
                    else
                      Q2_STORE(A.dst_p[idx], rim_carry);
                  }
              }
          }
      }
    }


    // second pass: add the seam partials per node (fixed order -> reproducible).
    // One wave per (tile, plane): lane = low-rim entry, so the south row (contiguous in dst and
    // in the slabs) is accessed coalesced and all index arithmetic but the lane part is scalar.
    template <int DEG, int NC>
    __device__ __forceinline__ void fixup_rim(const Q2Args &A, const long bt, const int K, double *dst,
                                              const double *slab, const double *zslab,
                                              const int nn_x, const int nn_y, const int nn_z,
                                              const uint32_t con)
    {
      // the tile owns its LOW-rim nodes; its own partial sum is already in dst; add the partial
      // sums of the west / south / south-west tiles (their high rim) and, on a chunk boundary
      // plane, of the chunk below
      constexpr int TN = DEG * TX + 1, RIM = 4 * (TN - 1), NE = (2 * TN - 1) * NC;
      const int     bx = (int)(bt % A.tiles_x), by = (int)(bt / A.tiles_x);
      const int     ppc  = DEG * A.LZ + 1;
      const int     c_hi = min(K / (DEG * A.LZ), A.n_chunks - 1);
      const int     lp   = K - DEG * A.LZ * c_hi;
      const bool    zb   = lp == 0 && c_hi > 0; // K is the top plane of the chunk below as well
#if defined(Q2_FIXUP_LOOP) // (the form of rounds 1-3, kept for A/B timing: loads behind branches, one dependent chain per node)
      for (int e = threadIdx.x & 63; e < NE; e += 64)
        {
          const int comp = e % NC, s = e / NC;
          const int i = s < TN ? s : 0, j = s < TN ? 0 : s - TN + 1; // south row, then west column
          const int I = DEG * TX * bx + i, J = DEG * TY * by + j;
          if (I >= nn_x || J >= nn_y)
            continue;
          const bool seam_x = i == 0 && I > 0, seam_y = j == 0 && J > 0;
          if (!(seam_x || seam_y))
            continue;
          // a node that is on this tile's HIGH rim in the other direction belongs to another owner
          if ((i == TN - 1 && I < nn_x - 1) || (j == TN - 1 && J < nn_y - 1))
            continue;
          if (on_constrained_face(I, J, K, nn_x, nn_y, nn_z, con, NC == 1 ? 1 : 3, comp))
            continue; // dst = +-src already written by every sharer
          if (fix_skip(A, I, J, K, nn_x, nn_y, nn_z))
            continue;
          double sum = 0.;
          for (int dy = 0; dy <= (seam_y ? 1 : 0); ++dy)
            for (int dx = 0; dx <= (seam_x ? 1 : 0); ++dx)
              {
                if (dx == 0 && dy == 0)
                  continue;
                const long tb = (long)(by - dy) * A.tiles_x + bx - dx;
                const int  r  = rim_index<TN>(i + (TN - 1) * dx, j + (TN - 1) * dy);
                sum += slab[(((tb * A.n_chunks + c_hi) * ppc + lp) * RIM + r) * NC + comp];
                if (zb)
                  sum += slab[(((tb * A.n_chunks + c_hi - 1) * ppc + DEG * A.LZ) * RIM + r) * NC + comp];
              }
          if (zb) // this tile's own partial of the chunk below
            sum += zslab[((bt * A.n_chunks + c_hi - 1) * (TN * TN) + j * TN + i) * NC + comp];
          dst[((long)(K * (long)nn_y + J) * nn_x + I) * NC + comp] += sum;
        }
#else
      // Straight-line: every load unconditional with a clamped address and issued before the first use, absent terms
      // deselected afterwards, the additions in the order of the loop above (bitwise the same sums).
      for (int e = threadIdx.x & 63; e < NE; e += 64)
        {
          const int  comp = e % NC, s = e / NC;
          const int  i = s < TN ? s : 0, j = s < TN ? 0 : s - TN + 1; // south row, then west column
          const int  I = DEG * TX * bx + i, J = DEG * TY * by + j;
          const bool seam_x = i == 0 && I > 0, seam_y = j == 0 && J > 0;
          // (a node on this tile's HIGH rim in the other direction belongs to another owner; on a constrained face
          // dst = +-src was written by every sharer)
          const bool act = I < nn_x && J < nn_y && (seam_x || seam_y) &&
                           !((i == TN - 1 && I < nn_x - 1) || (j == TN - 1 && J < nn_y - 1)) &&
                           !on_constrained_face(I, J, K, nn_x, nn_y, nn_z, con, NC == 1 ? 1 : 3, comp) &&
                           !fix_skip(A, I, J, K, nn_x, nn_y, nn_z);
          // the sharers (dx, dy) = (1, 0), (0, 1), (1, 1); an absent one reads from this tile, entry 0
          const bool has[3] = {seam_x, seam_y, seam_x && seam_y};
          double     hi[3], lo[3];
#pragma unroll
          for (int q = 0; q < 3; ++q)
            {
              const int  dx = q != 1 ? 1 : 0, dy = q != 0 ? 1 : 0;
              const bool ok = act && has[q];
              const long tb = ok ? (long)(by - dy) * A.tiles_x + bx - dx : bt;
              const int  r  = ok ? rim_index<TN>(i + (TN - 1) * dx, j + (TN - 1) * dy) : 0;
              hi[q] = slab[(((tb * A.n_chunks + c_hi) * ppc + lp) * RIM + r) * NC + comp];
              lo[q] = slab[(((tb * A.n_chunks + (zb ? c_hi - 1 : c_hi)) * ppc + (zb ? DEG * A.LZ : lp)) * RIM + r) * NC + comp];
            }
          // this tile's own partial of the chunk below
          const double zs  = zslab[((bt * A.n_chunks + (zb ? c_hi - 1 : 0)) * (TN * TN) + j * TN + i) * NC + comp];
          const long   idx = ((long)(K * (long)nn_y + min(J, nn_y - 1)) * nn_x + min(I, nn_x - 1)) * NC + comp;
          const double old = dst[idx];
          double       sum = 0.;
#pragma unroll
          for (int q = 0; q < 3; ++q)
            {
              sum += has[q] ? hi[q] : 0.;
              sum += has[q] && zb ? lo[q] : 0.;
            }
          sum += zb ? zs : 0.;
          if (act)
            dst[idx] = old + sum;
        }
#endif
    }

    // interior (non-seam) nodes of a chunk-boundary plane: dst (bottom partial of the upper chunk)
    // += top partial of the lower chunk
    template <int DEG, int NC>
    __device__ __forceinline__ void fixup_zplane(const Q2Args &A, const long bt, const int m, double *dst,
                                                 const double *zslab, const int nn_x, const int nn_y,
                                                 const int nn_z, const uint32_t con)
    {
      constexpr int TN = DEG * TX + 1;
      const int     bx = (int)(bt % A.tiles_x), by = (int)(bt / A.tiles_x);
      const int     K  = DEG * A.LZ * m;
      for (int e = threadIdx.x & 63; e < TN * TN * NC; e += 64)
        {
          const int comp = e % NC, n = e / NC, i = n % TN, j = n / TN;
          const int I = DEG * TX * bx + i, J = DEG * TY * by + j;
          if (I >= nn_x || J >= nn_y)
            continue;
          const bool seam = (i == 0 && I > 0) || (i == TN - 1 && I < nn_x - 1) || (j == 0 && J > 0) ||
                            (j == TN - 1 && J < nn_y - 1);
          if (seam || on_constrained_face(I, J, K, nn_x, nn_y, nn_z, con, NC == 1 ? 1 : 3, comp) ||
              fix_skip(A, I, J, K, nn_x, nn_y, nn_z))
            continue;
          dst[((long)(K * (long)nn_y + J) * nn_x + I) * NC + comp] += zslab[(bt * A.n_chunks + m - 1) * (TN * TN * NC) + e];
        }
    }

    // blocks [0,n1): velocity rim (tile, K); [n1,n1+n2): velocity z-planes (tile, m);
    // then the same two ranges for the pressure
    // (one wave per work item, FIXW waves per workgroup)
    constexpr int FIXW = 4;
    __global__ __launch_bounds__(64 * FIXW) void q2_seam_fixup_kernel(const Q2Args A, const long n1, const long n2,
                                                                      const long n3, const long n4)
    {
      for (long b = (long)blockIdx.x * FIXW + (threadIdx.x >> 6); b < n1 + n2 + n3 + n4; b += (long)gridDim.x * FIXW)
        {
          if (b < n1)
            fixup_rim<2, 3>(A, b / A.nnz, (int)(b % A.nnz), A.dst_u, A.slab_u, A.zslab_u, A.nnx, A.nny, A.nnz, A.con_u);
          else if (b < n1 + n2)
            fixup_zplane<2, 3>(A, (b - n1) / (A.n_chunks - 1), (int)((b - n1) % (A.n_chunks - 1)) + 1, A.dst_u,
                               A.zslab_u, A.nnx, A.nny, A.nnz, A.con_u);
          else if (b < n1 + n2 + n3)
            fixup_rim<1, 1>(A, (b - n1 - n2) / A.npz, (int)((b - n1 - n2) % A.npz), A.dst_p, A.slab_p, A.zslab_p,
                            A.npx, A.npy, A.npz, A.con_p);
          else
            fixup_zplane<1, 1>(A, (b - n1 - n2 - n3) / (A.n_chunks - 1), (int)((b - n1 - n2 - n3) % (A.n_chunks - 1)) + 1,
                               A.dst_p, A.zslab_p, A.npx, A.npy, A.npz, A.con_p);
        }
    }

    // generic [cell][12][27] (+ [cell][27] coefficient arrays) -> streaming layout
    // [tile][layer][q][half][wave][lane][2]: lanes 0..47 of a wave = (cell-in-wave*3+d) state
    // entries; with variable coefficients 16 more lanes per wave = (rho, mu) / (damping, 0) of
    // the wave's 16 cells.
    // One workgroup per (tile, layer, half, wave = 16 cells): the 6 (+2) components a half needs
    // are read as whole 216-B runs (27 points of one cell and component are contiguous in the
    // generic layout), transposed through LDS (28 KB: several workgroups per CU) and written as
    // 27 contiguous runs of 768 (1024) B.
    constexpr int CV_STRIDE = 8 * 27 + 1; // doubles per cell in LDS: 6 state + 2 coefficient comps (+1: bank skew)
    __global__ __launch_bounds__(256) void q2_convert_state_kernel(double *__restrict__ out,
                                                                   const double *__restrict__ gen,
                                                                   const double *__restrict__ rho,
                                                                   const double *__restrict__ mu,
                                                                   const double *__restrict__ damp,
                                                                   const int ncx, const int ncy,
                                                                   const int ncz, const int tiles_x,
                                                                   const int lin_mode, const long stride2)
    {
      extern __shared__ double cv[]; // [cell 16][slot 8][27]
      const int  lanes = rho ? 64 : 48;
      const long blk   = blockIdx.x;
      const int  half = (int)(blk & 1), wave = (int)(blk >> 1 & 3);
      const int  cz   = (int)((blk >> 3) % ncz);
      const long bt   = (blk >> 3) / ncz;
      const int  bx = (int)(bt % tiles_x), by = (int)(bt / tiles_x);
      // slot s < 6: state entry (d = s / 2, j = s % 2) of this half; slots 6, 7: coefficients j = 0, 1
      for (int e = threadIdx.x; e < 16 * 8 * 27; e += 256)
        {
          const int q = e % 27, s = (e / 27) % 8, cw = e / (27 * 8), cl = wave * 16 + cw;
          const int cx = bx * TX + (cl % TX), cy = by * TY + (cl / TX);
          double    v  = 0.;
          if (cx < ncx && cy < ncy && (s < 6 || rho))
            {
              const long cell = cx + (long)ncx * (cy + (long)ncy * cz);
              if (s < 6)
                {
                  const int d = s / 2, j = s % 2;
                  int       comp;
                  if (half == 0 && j == 0)
                    comp = d; // u_lin[d]
                  else if (lin_mode == 1)
                    comp = 3; // div_lin (second[0][0])
                  else
                    comp = 3 + 3 * d + (2 * half + j - 1); // grad_lin[d][e]
                  v = gen[(cell * NLIN + comp) * 27 + q];
                }
              else if (half == 0)
                v = (s == 6 ? rho : mu)[cell * 27 + q];
              else if (s == 6)
                v = damp[cell * 27 + q];
            }
          cv[cw * CV_STRIDE + s * 27 + q] = v;
        }
      __syncthreads();
      double *o = out + ((bt * ncz + cz) * (2 * stride2)) + (long)(half * 4 + wave) * (lanes * 2);
      const int per_q = 2 * 4 * lanes * 2; // doubles per point: both halves, four waves
      for (int e = threadIdx.x; e < 27 * lanes * 2; e += 256)
        {
          const int j = e & 1, lane = (e >> 1) % lanes, q = (e >> 1) / lanes;
          const int cw = lane < 48 ? lane / 3 : lane - 48;
          const int s  = lane < 48 ? 2 * (lane % 3) + j : 6 + j;
          o[(long)q * per_q + lane * 2 + j] = cv[cw * CV_STRIDE + s * 27 + q];
        }
    }
    // inverse of q2_convert_state_kernel (constant-coefficient layout): streaming -> generic
    // [cell][12][27].  Newton: all 12 entries; Picard-type: u_lin and div_lin (entry 3).
    __global__ __launch_bounds__(256) void q2_unconvert_state_kernel(double *__restrict__ gen,
                                                                     const double *__restrict__ str,
                                                                     const int ncx, const int ncy,
                                                                     const int ncz, const int tiles_x,
                                                                     const int lin_mode, const long stride2)
    {
      extern __shared__ double cv[]; // [cell 16][slot 6][27]
      const long blk   = blockIdx.x;
      const int  half = (int)(blk & 1), wave = (int)(blk >> 1 & 3);
      const int  cz   = (int)((blk >> 3) % ncz);
      const long bt   = (blk >> 3) / ncz;
      const int  bx = (int)(bt % tiles_x), by = (int)(bt / tiles_x);
      const double *o     = str + ((bt * ncz + cz) * (2 * stride2)) + (long)(half * 4 + wave) * (48 * 2);
      const int     per_q = 2 * 4 * 48 * 2;
      for (int e = threadIdx.x; e < 27 * 48 * 2; e += 256)
        {
          const int j = e & 1, lane = (e >> 1) % 48, q = (e >> 1) / 48;
          cv[(lane / 3) * CV_STRIDE + (2 * (lane % 3) + j) * 27 + q] = o[(long)q * per_q + lane * 2 + j];
        }
      __syncthreads();
      for (int e = threadIdx.x; e < 16 * 6 * 27; e += 256)
        {
          const int q = e % 27, s = (e / 27) % 6, cw = e / (27 * 6), cl = wave * 16 + cw;
          const int cx = bx * TX + (cl % TX), cy = by * TY + (cl / TX);
          if (cx >= ncx || cy >= ncy)
            continue;
          const int d = s / 2, j = s % 2;
          int       comp;
          if (half == 0 && j == 0)
            comp = d;
          else if (lin_mode == 1)
            comp = (half == 0 && d == 0) ? 3 : -1;
          else
            comp = 3 + 3 * d + (2 * half + j - 1);
          if (comp >= 0)
            gen[((cx + (long)ncx * (cy + (long)ncy * cz)) * NLIN + comp) * 27 + q] = cv[cw * CV_STRIDE + s * 27 + q];
        }
    }
  } // namespace

  int q2_unconvert_state(adaflo_ctx *ctx, double *generic, const double *streaming, const int lin_mode)
  {
    const int    tiles_x = (ctx->desc.ncell[0] + TX - 1) / TX, tiles_y = (ctx->desc.ncell[1] + TY - 1) / TY;
    const long   stride2 = 27L * 2 * 4 * 48 + ctx->q2_state_pad;
    const size_t lds     = sizeof(double) * 16 * CV_STRIDE;
    static bool  attr_set = false;
    if (!attr_set)
      {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&q2_unconvert_state_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
          return ADAFLO_EHIP;
        attr_set = true;
      }
    const long nb = (long)tiles_x * tiles_y * ctx->desc.ncell[2] * 8;
    hipLaunchKernelGGL(q2_unconvert_state_kernel, dim3((unsigned)nb), dim3(256), lds, ctx->stream, generic, streaming,
                       ctx->desc.ncell[0], ctx->desc.ncell[1], ctx->desc.ncell[2], tiles_x, lin_mode, stride2);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  static int q2_lin_mode(const adaflo_ctx *ctx)
  {
    if (ctx->ns.physical_type == ADAFLO_STOKES || ctx->ns.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT)
      return 2;
    return ctx->ns.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON ? 0 : 1;
  }

  // variable coefficients (two-phase flow) ride along with the linearisation state, so they
  // need a linearisation that has one (Newton / Picard-type); Stokes or explicit convection
  // with variable coefficients stays on the generic kernel
  static bool q2_varco(const adaflo_ctx *ctx)
  {
    return ctx->rho.p && ctx->mu.p && ctx->damp.p;
  }

  bool q2_supported(const adaflo_ctx *ctx)
  {
    if (ctx->k != 2 || ctx->flat || ctx->indexed) // (an indexed context has no brick to sweep)
      return false;
    if (!ctx->rho.p && !ctx->mu.p && !ctx->damp.p)
      return true;
    // (no linearisation state to carry the coefficients: round 6 reads them from the generic arrays for the explicit scheme;
    // Stokes with variable coefficients -- no value terms at all, :708 -- stays on the generic kernel)
    return q2_varco(ctx) && (q2_lin_mode(ctx) != 2 || (ctx->ns.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT &&
                                                       ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE));
  }

  int q2_prepare_state(adaflo_ctx *ctx)
  {
    if (int e = q2_materialize_state(ctx))
      return e;
    if (ctx->lin_q2_valid || q2_lin_mode(ctx) == 2)
      return 0;
    // (the streaming copy is rebuilt from the generic one, which must exist and be current:
    // adaflo_ns_set_params / adaflo_ns_set_coefficients see to that before they invalidate anything)
    if (!ctx->lin.p || !ctx->lin_generic_valid)
      return ADAFLO_ENOTINIT;
    const int    tiles_x = (ctx->desc.ncell[0] + TX - 1) / TX, tiles_y = (ctx->desc.ncell[1] + TY - 1) / TY;
    const bool   varco     = q2_varco(ctx);
    const long   per_layer = 27L * 2 * 4 * (varco ? 64 : 48) * 2;
    const long   stride2 = per_layer / 2 + ctx->q2_state_pad;
    const size_t count   = (size_t)tiles_x * tiles_y * ctx->desc.ncell[2] * 2 * stride2;
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
    if (ctx->q2_state_pad > 0 && hipMemsetAsync(ctx->lin_q2.p, 0, count * sizeof(double), ctx->stream) != hipSuccess)
      return ADAFLO_EHIP;
    {
      const size_t lds = sizeof(double) * 16 * CV_STRIDE;
      static bool  attr_set = false;
      if (!attr_set)
        {
          if (hipFuncSetAttribute(reinterpret_cast<const void *>(&q2_convert_state_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ADAFLO_EHIP;
          attr_set = true;
        }
      const long nb = (long)tiles_x * tiles_y * ctx->desc.ncell[2] * 8;
      hipLaunchKernelGGL(q2_convert_state_kernel, dim3((unsigned)nb), dim3(256), lds, ctx->stream, ctx->lin_q2.p,
                         ctx->lin.p, varco ? ctx->rho.p : nullptr, ctx->mu.p, ctx->damp.p, ctx->desc.ncell[0],
                         ctx->desc.ncell[1], ctx->desc.ncell[2], tiles_x, q2_lin_mode(ctx), stride2);
    }
    if (hipGetLastError() != hipSuccess)
      return ADAFLO_EHIP;
    ctx->lin_q2_valid = true;
    ctx->lin_q2_mode  = q2_lin_mode(ctx);
    ctx->lin_q2_varco = varco;
    return 0;
  }

  // development builds (scripts/exp_q2.sh -DQ2_FAST_BUILD): only the instantiations of the headline
  // benchmark (Newton, with pressure, cubic cells, constant coefficients; vmult and residual)
  static constexpr bool q2_instantiated(const int lin_mode, const bool with_p, const bool iso, const bool varco,
                                        const bool div)
  {
#if defined(Q2_FAST_BUILD)
    return lin_mode == 0 && with_p && iso && !varco && !div;
#else
    (void)lin_mode, (void)with_p, (void)iso, (void)varco, (void)div;
    return true;
#endif
  }

  // phase -1: the whole operator.  Phases 0/1/2 (multi-GPU overlap): 0 = first half of the
  // workgroups that touch no node of the inter-GPU interface faces `iface` (bit = face 2*dim+side),
  // 1 = the workgroups that do + seam fix-up of the interface nodes (dst is then final there and
  // can be packed), 2 = the remaining interior workgroups + the rest of the fix-up.
  static int q2_launch(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u,
                       const double *src_p, const int phase, const uint32_t iface, const bool residual,
                       const double *res_old, const double res_c_old, const double *res_ext = nullptr,
                       const bool discard_state = false)
  {
    // recompute-state mode (default of kernel variant 1 since round 5; variant 4 streams): Newton vmult / velocity_vmult
    // on the nodal linearisation point the last residual left (only then is the state a function of a nodal field the
    // engine knows); variable coefficients are read from the generic arrays.  velocity_vmult: the frozen nodal copy and
    // the frozen coefficients if fix_linearization_point has been called, else the current ones -- as the streamed state
    const bool frozen    = op == OP_VMULT_VELOCITY && (ctx->lin_q2_prec.p || ctx->lin_prec.p);
    const bool recompute = ctx->q2_recompute && !residual && (op == OP_VMULT || op == OP_VMULT_VELOCITY) && q2_lin_mode(ctx) != 2 &&
                           (frozen ? (ctx->lin_nodal_prec_valid && ctx->lin_nodal_prec.p != nullptr) : lin_nodal_current(ctx));
    const bool rc_varco  = recompute && (frozen ? ctx->rho_prec.p != nullptr : q2_varco(ctx));
    // explicit scheme (no state): velocity_vmult takes the coefficients fix_linearization_point froze, if any (as the
    // generic kernel, make_ns_args)
    const bool ex_frozen = !residual && q2_lin_mode(ctx) == 2 && op == OP_VMULT_VELOCITY && ctx->rho_prec.p != nullptr;
    const bool ex_varco  = !residual && q2_lin_mode(ctx) == 2 && op != OP_DIVERGENCE && (ex_frozen || q2_varco(ctx));
    if (!residual && op != OP_DIVERGENCE && !recompute)
      {
        if (int e = q2_materialize_state(ctx)) // (a residual deferred the layout of the state this kernel streams)
          return e;
        if (ctx->lin_q2_valid && (ctx->lin_q2_mode != q2_lin_mode(ctx) || ctx->lin_q2_varco != q2_varco(ctx)))
          {
            // (a streaming copy without coefficient pieces that is THE state -- the variable-coefficient residual wrote
            // it, then the kernel variant was changed to the streaming one: the generic copy first)
            if (!ctx->lin_generic_valid && !ctx->lin_q2_varco && ctx->lin_q2_mode == q2_lin_mode(ctx))
              {
                const size_t count = (size_t)ctx->n_cells * ctx->nq_u * 12;
                if (ctx->lin.count != count)
                  {
                    if (ctx->lin.p)
                      (void)hipFree(ctx->lin.p);
                    ctx->lin.p     = nullptr;
                    ctx->lin.count = 0;
                    if (hipMalloc(&ctx->lin.p, count * sizeof(double)) != hipSuccess)
                      return ADAFLO_ENOMEM;
                    ctx->lin.count = count;
                    if (hipMemsetAsync(ctx->lin.p, 0, count * sizeof(double), ctx->stream) != hipSuccess)
                      return ADAFLO_EHIP;
                  }
                if (int e = q2_unconvert_state(ctx, ctx->lin.p, ctx->lin_q2.p, ctx->lin_q2_mode))
                  return e;
                ctx->lin_generic_valid = true;
                ctx->lin_gen++;
              }
            ctx->lin_q2_valid = false;
          }
        if (int e = q2_prepare_state(ctx))
          return e;
      }
    Q2Args A{};
    const bool divergence = op == OP_DIVERGENCE;
    A.old_u     = recompute ? (frozen ? ctx->lin_nodal_prec.p : ctx->lin_nodal.p) : res_old;
    const bool res_varco = residual && q2_varco(ctx); // (two-phase residual: coefficients from the generic arrays)
    if (rc_varco)
      {
        A.rho  = frozen ? ctx->rho_prec.p : ctx->rho.p;
        A.mu   = frozen ? ctx->mu_prec.p : ctx->mu.p;
        A.damp = frozen ? ctx->damp_prec.p : ctx->damp.p;
      }
    if (res_varco)
      A.rho = ctx->rho.p, A.mu = ctx->mu.p, A.damp = ctx->damp.p;
    if (ex_varco)
      {
        A.rho  = ex_frozen ? ctx->rho_prec.p : ctx->rho.p;
        A.mu   = ex_frozen ? ctx->mu_prec.p : ctx->mu.p;
        A.damp = ex_frozen ? ctx->damp_prec.p : ctx->damp.p;
      }
    A.c_old     = res_c_old;
    A.ext_u     = res_ext; // (residual of the semi-implicit / explicit scheme: the extrapolated velocity at the nodes)
    A.state_out = residual ? (discard_state ? nullptr : ctx->lin_q2.p) : nullptr; // (null: the kernel masks its state stores off)
    A.c_div     = res_c_old; // (divergence mode passes its weight here)
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
      // z-chunks of 32 cell layers while that leaves >= 1024 workgroups (128^3: 36.1 against 35.5 GDoF/s with 16:
      // fewer chunk prologues and chunk-boundary planes in the fix-up pass)
      // (recompute mode, round 6: four rounds of workgroups instead of two -- 128^3: 16 layers 1.139 ms against 1.156 with 32,
      // 160^3 2.150 against 2.169; 256^3 has 16 rounds with 32 already and loses 1.4 % with 16: profiles/r06_chunk_sweep.log.
      // The streaming kernel pays more per chunk prologue and keeps the 1024 of round 3.)
      const long want  = recompute ? 2048 : 1024;
      int        lz    = ctx->q2_lz > 0 ? ctx->q2_lz : 32;
      while (lz > 4 && tiles * ((A.ncz + lz - 1) / lz) < want)
        lz /= 2;
      if (lz > A.ncz)
        lz = A.ncz;
      A.LZ       = lz;
      A.n_chunks = (A.ncz + lz - 1) / lz;
    }
    {
      const double       det = ctx->desc.h[0] * ctx->desc.h[1] * ctx->desc.h[2];
      const Quadrature1D qu  = gauss(3);
      const Shape1D      su  = shape_fe_q(2, qu);
      A.s0 = su.S[0];
      A.s1 = su.S[1];
      A.s2 = su.S[2];
      const std::vector<double> dc = collocation_derivative(qu);
      for (int e = 0; e < 3; ++e)
        {
          const double ih = 1. / ctx->desc.h[e];
          A.ah[e][0]      = dc[0] * ih;
          A.ah[e][1]      = dc[1] * ih;
          A.ah[e][2]      = dc[2] * ih;
          A.ah[e][3]      = dc[5] * ih;
        }
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
    A.gamma       = gamma;
    A.tau1        = P.tau1;
    // the frozen copy of velocity_vmult carries the coefficients it was built with
    const bool use_prec = op == OP_VMULT_VELOCITY && ctx->lin_q2_prec.p;
    const bool varco    = residual ? false : (q2_lin_mode(ctx) == 2 ? ex_varco : (use_prec ? ctx->lin_q2_prec_varco : q2_varco(ctx)));
    A.state_stride = 27L * 2 * 4 * (varco ? 64 : 48) + ctx->q2_state_pad;
    A.con_u       = ctx->brick.con_u;
    A.con_p       = ctx->brick.con_p;
    A.src_u       = src_u;
    A.src_p       = src_p;
    A.dst_u       = dst_u;
    A.dst_p       = dst_p;
    A.state       = use_prec ? ctx->lin_q2_prec.p : ctx->lin_q2.p;
    const int  lin_mode = divergence ? 2 : q2_lin_mode(ctx);
    const bool with_p   = op == OP_VMULT || divergence;

    // seam partial sums go to slabs (no atomics, no zero-initialisation of dst needed: every
    // entry of dst is written exactly once by the main kernel or by the fix-up kernel)
    {
      const size_t n_wg = (size_t)A.tiles_x * A.tiles_y * A.n_chunks;
      const size_t need[4] = {n_wg * (2 * A.LZ + 1) * RIM_U * 3, n_wg * UPLANE,
                              n_wg * (A.LZ + 1) * RIM_P, n_wg * PPLANE};
      DeviceBuffer *buf[4] = {&ctx->q2_slab_u, &ctx->q2_zslab_u, &ctx->q2_slab_p, &ctx->q2_zslab_p};
      for (int i = 0; i < 4; ++i)
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
      A.slab_u  = ctx->q2_slab_u.p;
      A.zslab_u = ctx->q2_zslab_u.p;
      A.slab_p  = ctx->q2_slab_p.p;
      A.zslab_p = ctx->q2_zslab_p.p;
    }
    if (with_p && !A.integrate_p && (phase <= 0 || phase == 5)) // (5: the set-up phase of the two-stream schedule runs on the engine stream BEFORE the auxiliary stream may pack or unpack-add dst_p; in phase 3 it raced with them -- ADVICE r05)
      {
        // projection scheme: the pressure rows are not integrated (:902-907) -- vmult: 0 and -src on constrained rows; the
        // residual's cell-loop sums: 0
        if (int e = residual ? launch_fill(ctx, dst_p, 0., ctx->n_nodes_p) :
                               launch_prepare_dst(ctx, dst_p, src_p, ctx->n_nodes_p, 1, A.npx, A.npy, A.npz, A.con_p, -1., true))
          return e;
      }
    long nwg = (long)A.tiles_x * A.tiles_y * A.n_chunks;
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
    if (static const bool dbg = std::getenv("ADAFLO_DEBUG_PTR") != nullptr; dbg)
      std::fprintf(stderr, "q2_launch: state %p src_u %p dst_u %p slab_u %p zslab_u %p\n", (const void *)A.state,
                   (const void *)A.src_u, (const void *)A.dst_u, (const void *)A.slab_u, (const void *)A.zslab_u);
    hipEvent_t stop = (ctx->timing && nwg > 0) ? ctx->kernel_timer.start(ctx->stream) : nullptr;
    const bool   iso = ctx->desc.h[0] == ctx->desc.h[1] && ctx->desc.h[1] == ctx->desc.h[2];
    const dim3   grid((unsigned)(nwg > 0 ? nwg : 1)), block(NT);
    const size_t lds_bytes = sizeof(double) * (L_TOTAL + (res_ext ? 3 * UPLANE_L : 0)); // (+ the node planes of the extrapolated velocity)
    hipError_t   err       = hipSuccess;
#define Q2_LAUNCH_V(LM, WP, IS, VC, RS)                                                               \
  {                                                                                             \
  if constexpr (!q2_instantiated(LM, WP, IS, VC, false))                                        \
    err = hipErrorNotSupported;                                                                 \
  else                                                                                          \
  {                                                                                             \
    static bool attr_set = false;                                                               \
    if (!attr_set)                                                                              \
      {                                                                                         \
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_q2_kernel<LM, WP, IS, VC, RS>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
        attr_set = err == hipSuccess;                                                           \
      }                                                                                         \
    if (err == hipSuccess && nwg > 0)                                                           \
      hipLaunchKernelGGL((ns_q2_kernel<LM, WP, IS, VC, RS>), grid, block, lds_bytes, ctx->stream, A); \
  }                                                                                             \
  }
#define Q2_LAUNCH_V6(LM, WP, IS, VC, RS, DV)                                                          \
  {                                                                                             \
  if constexpr (!q2_instantiated(LM, WP, IS, VC, DV))                                           \
    err = hipErrorNotSupported;                                                                 \
  else                                                                                          \
  {                                                                                             \
    static bool attr_set = false;                                                               \
    if (!attr_set)                                                                              \
      {                                                                                         \
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_q2_kernel<LM, WP, IS, VC, RS, DV>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
        attr_set = err == hipSuccess;                                                           \
      }                                                                                         \
    if (err == hipSuccess && nwg > 0)                                                           \
      hipLaunchKernelGGL((ns_q2_kernel<LM, WP, IS, VC, RS, DV>), grid, block, lds_bytes, ctx->stream, A); \
  }                                                                                             \
  }
#define Q2_LAUNCH_V7(LM, WP, IS, VC, RS, DV, RC)                                                      \
  {                                                                                             \
    static bool attr_set = false;                                                               \
    if (!attr_set)                                                                              \
      {                                                                                         \
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_q2_kernel<LM, WP, IS, VC, RS, DV, RC>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
        attr_set = err == hipSuccess;                                                           \
      }                                                                                         \
    if (err == hipSuccess && nwg > 0)                                                           \
      hipLaunchKernelGGL((ns_q2_kernel<LM, WP, IS, VC, RS, DV, RC>), grid, block, lds_bytes, ctx->stream, A); \
  }
#define Q2_LAUNCH_I(LM, WP, IS)              \
  {                                          \
    if (varco)                               \
      Q2_LAUNCH_V(LM, WP, IS, true, false)   \
    else                                     \
      Q2_LAUNCH_V(LM, WP, IS, false, false)  \
  }
#define Q2_LAUNCH(LM, WP)      \
  {                            \
    if (iso)                   \
      Q2_LAUNCH_I(LM, WP, true) \
    else                       \
      Q2_LAUNCH_I(LM, WP, false) \
  }
    if (divergence)
      {
        if (iso)
          Q2_LAUNCH_V6(2, true, true, false, false, true)
        else
          Q2_LAUNCH_V6(2, true, false, false, false, true)
      }
    else if (residual && res_ext)
      {
#define Q2_LAUNCH_EXT_V(LM, IS, VC)                                                                                 \
  {                                                                                                                 \
    static bool attr_set = false;                                                                                   \
    if (!attr_set)                                                                                                  \
      {                                                                                                             \
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_q2_kernel<LM, true, IS, VC, true, false, false, true>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                      \
        attr_set = err == hipSuccess;                                                                               \
      }                                                                                                             \
    if (err == hipSuccess && nwg > 0)                                                                               \
      hipLaunchKernelGGL((ns_q2_kernel<LM, true, IS, VC, true, false, false, true>), grid, block, lds_bytes, ctx->stream, A); \
  }
#define Q2_LAUNCH_EXT(LM, IS)           \
  {                                     \
    if (res_varco)                      \
      Q2_LAUNCH_EXT_V(LM, IS, true)     \
    else                                \
      Q2_LAUNCH_EXT_V(LM, IS, false)    \
  }
        if (iso && lin_mode == 1)
          Q2_LAUNCH_EXT(1, true)
        else if (iso)
          Q2_LAUNCH_EXT(2, true)
        else if (lin_mode == 1)
          Q2_LAUNCH_EXT(1, false)
        else
          Q2_LAUNCH_EXT(2, false)
#undef Q2_LAUNCH_EXT
#undef Q2_LAUNCH_EXT_V
      }
    else if (res_varco)
      {
        // (variable coefficients: Newton / Picard-type, q2_residual_supported)
        if (iso && lin_mode == 0)
          Q2_LAUNCH_V(0, true, true, true, true)
        else if (iso)
          Q2_LAUNCH_V(1, true, true, true, true)
        else if (lin_mode == 0)
          Q2_LAUNCH_V(0, true, false, true, true)
        else
          Q2_LAUNCH_V(1, true, false, true, true)
      }
    else if (residual)
      {
        // (constant coefficients, src = the solution: launch_ns_residual_q2)
        if (iso)
          switch (lin_mode)
            {
              case 0:
                Q2_LAUNCH_V(0, true, true, false, true);
                break;
              case 1:
                Q2_LAUNCH_V(1, true, true, false, true);
                break;
              default:
                Q2_LAUNCH_V(2, true, true, false, true);
            }
        else
          switch (lin_mode)
            {
              case 0:
                Q2_LAUNCH_V(0, true, false, false, true);
                break;
              case 1:
                Q2_LAUNCH_V(1, true, false, false, true);
                break;
              default:
                Q2_LAUNCH_V(2, true, false, false, true);
            }
      }
    else if (recompute)
      {
#define Q2_LAUNCH_RC(WP, IS)                                 \
  {                                                          \
    if (rc_varco && lin_mode == 0)                           \
      Q2_LAUNCH_V7(0, WP, IS, true, false, false, true)      \
    else if (rc_varco)                                       \
      Q2_LAUNCH_V7(1, WP, IS, true, false, false, true)      \
    else if (lin_mode == 0)                                  \
      Q2_LAUNCH_V7(0, WP, IS, false, false, false, true)     \
    else                                                     \
      Q2_LAUNCH_V7(1, WP, IS, false, false, false, true)     \
  }
        if (with_p && iso)
          Q2_LAUNCH_RC(true, true)
        else if (with_p)
          Q2_LAUNCH_RC(true, false)
        else if (iso)
          Q2_LAUNCH_RC(false, true)
        else
          Q2_LAUNCH_RC(false, false)
#undef Q2_LAUNCH_RC
      }
    else if (with_p)
      switch (lin_mode)
        {
          case 0:
            Q2_LAUNCH(0, true);
            break;
          case 1:
            Q2_LAUNCH(1, true);
            break;
          default:
            Q2_LAUNCH(2, true);
        }
    else
      switch (lin_mode)
        {
          case 0:
            Q2_LAUNCH(0, false);
            break;
          case 1:
            Q2_LAUNCH(1, false);
            break;
          default:
            Q2_LAUNCH(2, false);
        }
#undef Q2_LAUNCH
#undef Q2_LAUNCH_I
#undef Q2_LAUNCH_V
#undef Q2_LAUNCH_V6
    if (err != hipSuccess)
      return ADAFLO_EHIP;
    if (stop)
      (void)hipEventRecord(stop, ctx->stream);
    if (phase == -1 || phase == 1)
      ctx->kernel_timer.count++;
    if (phase != 0 && phase != 3 && phase != 5 && !(phase == 1 && iface == 0u)) // (no interface: phase 1 has nothing to fix up)
    {
      const long tiles = (long)A.tiles_x * A.tiles_y;
      const bool fix_p = with_p && A.integrate_p;
      const long n1 = divergence ? 0 : tiles * A.nnz, n2 = divergence ? 0 : tiles * (A.n_chunks - 1);
      const long n3 = fix_p ? tiles * A.npz : 0, n4 = fix_p ? tiles * (A.n_chunks - 1) : 0;
      long       nb = (n1 + n2 + n3 + n4 + FIXW - 1) / FIXW;
      if (nb > 256 * 128)
        nb = 256 * 128;
      hipLaunchKernelGGL(q2_seam_fixup_kernel, dim3((unsigned)nb), dim3(64 * FIXW), 0, ctx->stream, A, n1, n2, n3, n4);
    }
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_ns_vmult_q2(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u,
                         const double *src_p, const int phase, const uint32_t iface)
  {
    return q2_launch(ctx, op, dst_u, dst_p, src_u, src_p, phase, iface, false, nullptr, 0.);
  }

  // sum_p = cell loop of local_divergence (q, weight div u) with the constraints of src_u resolved;
  // entries on constrained pressure rows are undefined (the caller skips them)
  int launch_ns_divergence_q2(adaflo_ctx *ctx, double *sum_p, const double *src_u, const double *any_p, const double weight)
  {
    return q2_launch(ctx, OP_DIVERGENCE, nullptr, sum_p, src_u, any_p, -1, 0u, false, nullptr, weight);
  }

  // residual on the sweep structure: Newton / Picard-type linearisations (the state they store is
  // that of the solution itself) and Stokes, constant coefficients
  bool q2_residual_supported(const adaflo_ctx *ctx)
  {
    if (ctx->k != 2 || ctx->flat)
      return false;
    // variable coefficients (two-phase flow, round 5): with the Newton -- round 6: or the Picard-type -- linearisation and the
    // recompute-state mode of the vmults (kernel variant 1) -- the state then leaves the kernel in the constant-coefficient
    // layout (or not at all: lazy state) and nobody reads it unless asked (get_linearization, a change of variant: re-laid
    // out then)
    // (round 6, second half: the schemes that linearise about the extrapolated velocity as well -- time-dependent equations)
    if (ctx->rho.p || ctx->mu.p || ctx->damp.p)
      return q2_varco(ctx) && ctx->q2_recompute && ctx->ns.physical_type != ADAFLO_STOKES &&
             (ctx->ns.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON || ctx->ns.linearization == ADAFLO_COUPLED_IMPLICIT_PICARD ||
              ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE);
    const NSDev &P = ctx->ns;
    if (P.physical_type == ADAFLO_STOKES || P.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON ||
        P.linearization == ADAFLO_COUPLED_IMPLICIT_PICARD)
      return true;
    // (round 5) the schemes that linearise about the extrapolated old velocity, time-dependent equations (the old
    // solutions exist); (round 6) the projection scheme: the semi-implicit residual without the pressure rows (:644-647,
    // :902-907)
    return P.physical_type == ADAFLO_INCOMPRESSIBLE &&
           (P.linearization == ADAFLO_COUPLED_VELOCITY_SEMI_IMPLICIT || P.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT ||
            P.linearization == ADAFLO_PROJECTION);
  }

  // sum_u / sum_p = cell-loop result of NavierStokesOps::residual (zero on constrained rows);
  // old_comb = weight_old u_old + weight_old_old u_old_old at the nodes (or nullptr).  Leaves the
  // quadrature-point state of `src` in the streaming layout (ctx->lin_q2).
  // (the caller has bumped ctx->lin_serial for the state it is about to produce)
  // (src_u: the nodal field whose interpolation IS the state -- the solution for Newton / Picard-type, the extrapolated old
  // velocity for the semi-implicit and the projection scheme)
  int q2_capture_nodal(adaflo_ctx *ctx, const double *src_u)
  {
    if (!ctx->q2_recompute || q2_lin_mode(ctx) == 2)
      return 0;
    const size_t nu = 3 * (size_t)ctx->n_nodes_u;
    if (ctx->lin_nodal.count != nu)
      {
        if (ctx->lin_nodal.p)
          (void)hipFree(ctx->lin_nodal.p);
        ctx->lin_nodal.p     = nullptr;
        ctx->lin_nodal.count = 0;
        if (hipMalloc(&ctx->lin_nodal.p, nu * sizeof(double)) != hipSuccess)
          return ADAFLO_ENOMEM;
        ctx->lin_nodal.count = nu;
      }
    if (hipMemcpyAsync(ctx->lin_nodal.p, src_u, nu * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
      return ADAFLO_EHIP;
    ctx->lin_nodal_serial = ctx->lin_serial;
    return 0;
  }

  // the streaming copy of the state as the residual mode writes it (constant-coefficient layout)
  static int q2_alloc_state(adaflo_ctx *ctx)
  {
    const int    tiles_x = (ctx->desc.ncell[0] + TX - 1) / TX, tiles_y = (ctx->desc.ncell[1] + TY - 1) / TY;
    const long   stride2 = 27L * 2 * 4 * 48 + ctx->q2_state_pad;
    const size_t count   = (size_t)tiles_x * tiles_y * ctx->desc.ncell[2] * 2 * stride2;
    if (ctx->lin_q2.count != count)
      {
        if (ctx->lin_q2.p)
          (void)hipFree(ctx->lin_q2.p);
        ctx->lin_q2.p     = nullptr;
        ctx->lin_q2.count = 0;
        if (hipMalloc(&ctx->lin_q2.p, count * sizeof(double)) != hipSuccess)
          return ADAFLO_ENOMEM;
        ctx->lin_q2.count = count;
        // cells beyond the mesh in partial tiles are never written: define them once
        if (hipMemsetAsync(ctx->lin_q2.p, 0, count * sizeof(double), ctx->stream) != hipSuccess)
          return ADAFLO_EHIP;
      }
    return 0;
  }
  static int q2_ensure_buffer(DeviceBuffer &b, const size_t count)
  {
    if (b.count >= count && b.p)
      return 0;
    if (b.p)
      (void)hipFree(b.p);
    b.p     = nullptr;
    b.count = 0;
    if (hipMalloc(&b.p, count * sizeof(double)) != hipSuccess)
      return ADAFLO_ENOMEM;
    b.count = count;
    return 0;
  }

  // Lazy state: the Newton residual below skips the layout of the quadrature-point state when the recompute-state vmult is
  // its consumer; whoever needs the state after all (adaflo_ns_get_linearization, the generic and the streaming kernels,
  // adaflo_ns_fix_linearization_point, a change of scheme) gets it here: the residual kernel once more on the nodal
  // linearisation point, for the state alone -- the sums go to the work vectors of the residual and are dropped.
  // Invariant: lin_q2_deferred implies lin_nodal_current (every ctx->lin_serial++ site clears the flag).
  int q2_materialize_state(adaflo_ctx *ctx)
  {
    if (!ctx->lin_q2_deferred)
      return 0;
    if (!lin_nodal_current(ctx) || q2_lin_mode(ctx) == 2)
      return ADAFLO_ENOTINIT;
    // (semi-implicit / projection scheme: the state is that of the extrapolated velocity, which is what the nodal copy holds)
    const bool ext = ctx->ns.linearization == ADAFLO_COUPLED_VELOCITY_SEMI_IMPLICIT || ctx->ns.linearization == ADAFLO_PROJECTION;
    const size_t nu = 3 * (size_t)ctx->n_nodes_u, np = (size_t)ctx->n_nodes_p;
    if (int e = q2_alloc_state(ctx))
      return e;
    if (int e = q2_ensure_buffer(ctx->res_sum_u, nu))
      return e;
    if (int e = q2_ensure_buffer(ctx->res_sum_p, np))
      return e;
    if (int e = q2_ensure_buffer(ctx->res_ext, nu)) // (a pressure to read: the state does not depend on it)
      return e;
    if (hipMemsetAsync(ctx->res_ext.p, 0, np * sizeof(double), ctx->stream) != hipSuccess)
      return ADAFLO_EHIP;
    if (int e = q2_launch(ctx, OP_VMULT, ctx->res_sum_u.p, ctx->res_sum_p.p, ctx->lin_nodal.p, ctx->res_ext.p, -1, 0u, true,
                          ctx->lin_nodal.p, 0., ext ? ctx->lin_nodal.p : nullptr, false))
      return e;
    ctx->lin_q2_deferred = false;
    ctx->lin_q2_valid    = true;
    ctx->lin_q2_mode     = q2_lin_mode(ctx);
    ctx->lin_q2_varco    = false;
    return 0;
  }

  int launch_ns_residual_q2(adaflo_ctx *ctx, double *sum_u, double *sum_p, const double *src_u,
                            const double *src_p, const double *old_comb, const double *ext_comb)
  {
    const int lin_mode = q2_lin_mode(ctx);
    // Newton + recompute-state vmult (kernel variant 1): nobody reads the laid-out state unless asked (q2_materialize_state)
    const bool defer = lin_mode != 2 && ctx->q2_recompute && ctx->q2_lazy_state;
    ctx->lin_q2_deferred = false; // (the caller has bumped lin_serial: whatever was deferred belongs to the state before)
    if (!defer && lin_mode != 2)
      {
        if (int e = q2_alloc_state(ctx))
          return e;
      }
    // (variable coefficients: the kernel multiplies by the density of the point)
    const double c_old = (old_comb && ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE) ? (q2_varco(ctx) ? 1. : ctx->ns.density) : 0.;
    if (int e = q2_launch(ctx, OP_VMULT, sum_u, sum_p, src_u, src_p, -1, 0u, true, old_comb ? old_comb : src_u, c_old, ext_comb, defer))
      return e;
    if (defer)
      ctx->lin_q2_valid = false;
    else if (lin_mode != 2)
      {
        ctx->lin_q2_valid = true;
        ctx->lin_q2_mode  = lin_mode;
        ctx->lin_q2_varco = false;
      }
    // recompute-state mode: keep the nodal linearisation point (the solution this residual was evaluated at; the
    // extrapolated old velocity for the schemes that linearise about it)
    if (lin_mode != 2)
      if (int e = q2_capture_nodal(ctx, ext_comb ? ext_comb : src_u))
        return e;
    ctx->lin_q2_deferred = defer; // (with the nodal copy in place: the invariant of q2_materialize_state)
    return 0;
  }
} // namespace adaflo_hip
