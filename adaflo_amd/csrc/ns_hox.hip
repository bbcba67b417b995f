// ns_hox.hip -- host side of the x-marching Taylor-Hood Q_k/Q_{k-1} kernel (k = 3, 4, 5); the device source
// and the description of the decomposition are in ns_hox_kernel.hpp.
// NavierStokesMatrix::vmult / velocity_vmult with constant coefficients
// (source/navier_stokes_matrix.cc:221-262, 337-382, 601-916).
#include "basis.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "hox_intrin.hpp"
#include "ns_hox_kernel.hpp"

// recompute-state mode of the x-marching kernels (template RCP of ns_hox_kernel): measured
// SLOWER than the streamed state for k = 3, 4, 5 (Q4/Q3 64^3: 1.85 against 1.32 ms), so the product build does not
// instantiate it; -DHOX_RCP_BUILD=1 reproduces the measurement (scripts/dev/ab_hox_rcp.sh), the emulator tests cover it
#ifndef HOX_RCP_BUILD
#define HOX_RCP_BUILD 0
#endif

namespace adaflo_hip
{
  namespace
  {
    using namespace hox;

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

    int lin_mode_of(const adaflo_ctx *ctx)
    {
      const NSDev &P = ctx->ns;
      if (P.physical_type == ADAFLO_STOKES || P.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT)
        return 2;
      return P.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON ? 0 : 1;
    }

    template <int K>
    size_t state_doubles(const adaflo_ctx *ctx, const int lin_mode, const bool varco = false)
    {
      using G         = Geo<K>;
      const size_t ngy = (ctx->desc.ncell[1] + G::CWY - 1) / G::CWY, ngz = (ctx->desc.ncell[2] + G::CWZ - 1) / G::CWZ;
      return ngz * ngy * (size_t)ctx->desc.ncell[0] * G::N * (nst_of(lin_mode) / 2 + (varco ? 2 : 0)) * G::CPW * G::NL * 2;
    }

    // streaming copy of the (frozen) linearisation state -- and of the variable coefficients, if any --, rebuilt when the
    // generic copies it was made from changed (lin_gen / lin_prec_gen count every change of state AND coefficients)
    template <int K>
    int prepare_state(adaflo_ctx *ctx, const bool prec, const int lin_mode, const double **out, bool *varco_out)
    {
      using G = Geo<K>;
      *out    = nullptr;
      const bool          use_prec = prec && (ctx->lin_prec.p || ctx->hox_lin_prec_primary || ctx->rho_prec.p);
      const bool          varco    = use_prec ? ctx->rho_prec.p != nullptr : ctx->rho.p != nullptr;
      *varco_out                   = varco;
      if (lin_mode == 2 && !varco)
        return 0;
      const DeviceBuffer &gen      = use_prec ? ctx->lin_prec : ctx->lin;
      DeviceBuffer       &str      = use_prec ? ctx->hox_lin_prec : ctx->hox_lin;
      unsigned long      &have     = use_prec ? ctx->hox_lin_prec_gen : ctx->hox_lin_gen;
      int                &mode     = use_prec ? ctx->hox_lin_prec_mode : ctx->hox_lin_mode;
      bool               &has_co   = use_prec ? ctx->hox_lin_prec_varco : ctx->hox_lin_varco;
      const unsigned long want     = use_prec ? ctx->lin_prec_gen : ctx->lin_gen;
      const size_t need = state_doubles<K>(ctx, lin_mode, varco);
      if (str.p && have == want && mode == lin_mode && has_co == varco && str.count >= need)
        {
          *out = str.p;
          return 0;
        }
      // (a state that exists in the streaming layout only -- written by the residual mode -- is current by
      // construction; a change of scheme brings the generic copy up to date first, adaflo_ns_set_params)
      if (lin_mode != 2 && (!gen.p || !(use_prec ? ctx->lin_prec_generic_valid : ctx->lin_generic_valid)))
        return ADAFLO_ENOTINIT;
      if (int e = ensure(str, need))
        return e;
      const int  ncx = ctx->desc.ncell[0], ncy = ctx->desc.ncell[1], ncz = ctx->desc.ncell[2];
      const int  ngy = (ncy + G::CWY - 1) / G::CWY, ngz = (ncz + G::CWZ - 1) / G::CWZ;
      const long pieces = (long)(need / 2);
      long       nb     = (pieces + 255) / 256;
      if (nb > 256 * 64)
        nb = 256 * 64;
      const int     npl = nst_of(lin_mode) / 2;
      const double *rho = use_prec ? ctx->rho_prec.p : ctx->rho.p, *mu = use_prec ? ctx->mu_prec.p : ctx->mu.p,
                   *damp = use_prec ? ctx->damp_prec.p : ctx->damp.p;
      hipLaunchKernelGGL((hox_convert_state_kernel<K>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, str.p, gen.p, ncx, ncy,
                         ncz, ngy, ngz, npl + (varco ? 2 : 0), varco ? npl : -1, rho, mu, damp);
      if (hipGetLastError() != hipSuccess)
        return ADAFLO_EHIP;
      have   = want;
      mode   = lin_mode;
      has_co = varco;
      *out   = str.p;
      return 0;
    }

    // coefficient stream of the variable-coefficient residual: (rho, mu), (damping, -) per quadrature point in the layout of
    // a state without linearisation pieces, rebuilt when the arrays changed (ctx->coef_gen)
    template <int K>
    int prepare_coefficients(adaflo_ctx *ctx, const double **out)
    {
      using G           = Geo<K>;
      const size_t need = state_doubles<K>(ctx, 2, true);
      if (ctx->hox_coef.p && ctx->hox_coef.count >= need && ctx->hox_coef_gen == ctx->coef_gen)
        {
          *out = ctx->hox_coef.p;
          return 0;
        }
      if (int e = ensure(ctx->hox_coef, need))
        return e;
      const int  ncx = ctx->desc.ncell[0], ncy = ctx->desc.ncell[1], ncz = ctx->desc.ncell[2];
      const int  ngy = (ncy + G::CWY - 1) / G::CWY, ngz = (ncz + G::CWZ - 1) / G::CWZ;
      long       nb  = ((long)(need / 2) + 255) / 256;
      if (nb > 256 * 64)
        nb = 256 * 64;
      hipLaunchKernelGGL((hox_convert_state_kernel<K>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, ctx->hox_coef.p,
                         (const double *)nullptr, ncx, ncy, ncz, ngy, ngz, 2, 0, ctx->rho.p, ctx->mu.p, ctx->damp.p);
      if (hipGetLastError() != hipSuccess)
        return ADAFLO_EHIP;
      ctx->hox_coef_gen = ctx->coef_gen;
      *out              = ctx->hox_coef.p;
      return 0;
    }

    // residual = true: dst_u / dst_p receive the cell-loop sums, old_comb is the nodal combination of the old solutions
    // (or null), the state is WRITTEN to ctx->hox_lin
    template <int K>
    int launch_hox(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u, const double *src_p,
                   const int phase, const uint32_t iface, const bool residual = false, const double *old_comb = nullptr,
                   const double *ext_comb = nullptr)
    {
      using G         = Geo<K>;
      constexpr int N = K + 1;
      HXArgs        A{};
      const int     lin_mode = lin_mode_of(ctx);
      {
        int lx = ctx->hox_lx > 0 ? ctx->hox_lx : 0;
        if (lx == 0)
          {
            // x-chunks by a small cost model (round 5): the workgroups run in rounds of (workgroups per CU) x (CUs) slots;
            // a workgroup of m chunks per row costs its chunk length + ~2.5 steps (prologue, the seam plane, the tail of the
            // pipeline); take the m with the least rounds x cost, chunks of at least 4 cells.  (Before: "at least 1024
            // workgroups" -- fine where the tiles fill the slots evenly, e.g. 512 tiles of Q4/Q3 64^3, but Q5/Q4 48^3 has 576
            // tiles: two chunks are 2.25 rounds, four are 4.5: kernel 1.18 -> 1.08 ms.)
            static const int n_cu = [] {
              int dev = 0, n = 0;
              if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
                n = 256;
              return n;
            }();
            const bool one_per_cu = residual && K > 3; // (HOX_RES_LB / HOX_EXT_LB: 512 registers)
            const long slots = (long)n_cu * (one_per_cu ? 1 : 2);
            const long tiles = (long)((ctx->desc.ncell[1] + G::CY - 1) / G::CY) * ((ctx->desc.ncell[2] + G::CZ - 1) / G::CZ);
            const int  ncx   = ctx->desc.ncell[0];
            double     best  = 1e300;
            lx               = ncx;
            for (int m = 1; m <= ncx; ++m)
              {
                const int l = (ncx + m - 1) / m;
                if (l < 4 && m > 1)
                  break;
                const long   chunks = (ncx + l - 1) / l, rounds = (tiles * chunks + slots - 1) / slots;
                const double cost   = (double)rounds * (l + 2.5);
                if (cost < best * 0.999)
                  best = cost, lx = l;
              }
          }
        hox_geometry<K>(A, ctx->desc.ncell, lx);
      }
      if ((size_t)A.nnx * A.nny * A.nnz * 3 >= ((size_t)1 << 32))
        return ADAFLO_EUNSUPPORTED; // 32-bit row offsets
      const NSDev &P      = ctx->ns;
      const bool   stokes = P.physical_type == ADAFLO_STOKES;
      const double gamma  = P.physical_type == ADAFLO_INCOMPRESSIBLE ? P.weight : 0.;
      {
        // 1D matrices (even / odd form) and the constants of the quadrature-point operation: one small table,
        // re-uploaded when a parameter changed
        const Quadrature1D        qu = gauss(N);
        const Shape1D             su = shape_fe_q(K, qu), sp = shape_fe_q(K - 1, qu);
        const std::vector<double> dc = collocation_derivative(qu);
        const double cA = stokes ? 0. : gamma * P.density - P.damping; // :717,:827-835; Stokes: no value terms (:708)
        const double cB = stokes ? 0. : P.tau1 * P.density;
        const std::vector<double> tab = hox_table<K>(su.S.data(), dc.data(), sp.S.data(), qu.w.data(), ctx->desc.h, cA, cB,
                                                     P.beta, P.tau_grad_div, P.viscosity * P.tau1 /* :841-845 */,
                                                     stokes ? 0. : gamma, stokes ? 0. : P.tau1, stokes ? 0. : 1., P.tau1);
        if (tab != ctx->hox_tab_host)
          {
            if (int e = ensure(ctx->hox_tab, tab.size()))
              return e;
            // (stream-ordered: earlier launches that read the old table are finished first)
            if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
                copy_to_device_now(ctx->hox_tab.p, tab.data(), tab.size() * sizeof(double)) != hipSuccess)
              return ADAFLO_EHIP;
            ctx->hox_tab_host = tab;
          }
      }
      A.tab         = ctx->hox_tab.p;
      A.integrate_p = P.linearization != ADAFLO_PROJECTION;
      A.con_u       = ctx->brick.con_u;
      A.con_p       = ctx->brick.con_p;
      A.src_u       = src_u;
      A.src_p       = src_p;
      A.dst_u       = dst_u;
      A.dst_p       = dst_p;
      bool varco = false, recompute = false;
      if (residual)
        {
          varco   = ctx->rho.p != nullptr; // (two-phase residual, round 6: hox_residual_supported admits Newton / Picard-type)
          A.old_u = old_comb;
          A.c_old = old_comb ? (varco ? 1. : P.density) : 0.; // (variable density: the kernel multiplies by the point's)
          A.lin_u = ext_comb; // (schemes that linearise about the extrapolated old velocity)
          if (varco)
            if (int e = prepare_coefficients<K>(ctx, &A.lin))
              return e;
          if (lin_mode != 2)
            {
              // (+ one cell: the sink of the cells beyond the mesh, HXArgs::lin_sink)
              const size_t n_state = state_doubles<K>(ctx, lin_mode, varco);
              if (int e = ensure(ctx->hox_lin, n_state + (size_t)N * (nst_of(lin_mode) / 2 + (varco ? 2 : 0)) * G::CPW * G::NL * 2))
                return e;
              A.lin_out  = ctx->hox_lin.p;
              A.lin_sink = ctx->hox_lin.p + n_state;
            }
        }
      else
        {
          // recompute-state mode (default of kernel variant 1; variant 4 streams): the state is the interpolation of the
          // nodal field the last residual was evaluated at -- constant coefficients; velocity_vmult takes the frozen
          // nodal copy if fix_linearization_point has been called, as the streamed state
          const bool frozen = op == OP_VMULT_VELOCITY && (ctx->lin_prec.p || ctx->hox_lin_prec_primary || ctx->rho_prec.p);
          const bool co     = frozen ? ctx->rho_prec.p != nullptr : ctx->rho.p != nullptr;
          recompute = HOX_FUSED && HOX_RCP_BUILD && ctx->q2_recompute && lin_mode != 2 && !co &&
                      (frozen ? (ctx->lin_nodal_prec_valid && ctx->lin_nodal_prec.p != nullptr) : lin_nodal_current(ctx));
          if (recompute)
            A.lin_u = frozen ? ctx->lin_nodal_prec.p : ctx->lin_nodal.p;
          else if (int e = prepare_state<K>(ctx, op == OP_VMULT_VELOCITY, lin_mode, &A.lin, &varco))
            return e;
        }
      const bool   with_p = op == OP_VMULT || residual;
      const size_t n_wg   = (size_t)A.tiles_y * A.tiles_z * A.n_chunks;
      if (int e = ensure(ctx->hox_slab_u, n_wg * G::RIMU * (K * A.LX + 1) * 3))
        return e;
      if (int e = ensure(ctx->hox_xslab_u, n_wg * G::TNY * G::TNZ * 3))
        return e;
      if (int e = ensure(ctx->hox_slab_p, n_wg * G::RIMP * ((K - 1) * A.LX + 1)))
        return e;
      if (int e = ensure(ctx->hox_xslab_p, n_wg * G::TPY * G::TPZ))
        return e;
#if HOX_STAMP
      {
        static unsigned long long *stamps = nullptr;
        if (!stamps)
          (void)hipMalloc(&stamps, (size_t)1 << 24);
        A.stamps = stamps;
      }
#endif
      A.slab_u  = ctx->hox_slab_u.p;
      A.xslab_u = ctx->hox_xslab_u.p;
      A.slab_p  = ctx->hox_slab_p.p;
      A.xslab_p = ctx->hox_xslab_p.p;
      if (with_p && !A.integrate_p && (phase <= 0 || phase == 5)) // (5: the set-up phase of the two-stream schedule runs on the engine stream BEFORE the auxiliary stream may pack or unpack-add dst_p; in phase 3 it raced with them -- ADVICE r05)
        if (int e = residual ? launch_fill(ctx, dst_p, 0., ctx->n_nodes_p) : // (the residual's sums of rows that are not integrated: 0)
                               launch_prepare_dst(ctx, dst_p, src_p, ctx->n_nodes_p, 1, A.npx, A.npy, A.npz, A.con_p, -1., true))
          return e;
      long nwg = (long)n_wg;
      if (phase >= 0)
        {
          // workgroup list [interface | interior A | interior B], cached per (grid, iface)
          const long key[4] = {A.tiles_y, A.tiles_z, A.n_chunks, (long)iface};
          if (!ctx->hox_wg_list || std::memcmp(key, ctx->hox_wg_key, sizeof(key)) != 0)
            {
              std::vector<int> bnd;
              hox_wg_lists(A, iface, bnd, ctx->hox_wg_counts);
              if (ctx->hox_wg_list)
                (void)hipFree(ctx->hox_wg_list);
              ctx->hox_wg_list = nullptr;
              if (hipMalloc(&ctx->hox_wg_list, sizeof(int) * (bnd.size() + 1)) != hipSuccess)
                return ADAFLO_ENOMEM;
              if (copy_to_device_now(ctx->hox_wg_list, bnd.data(), sizeof(int) * bnd.size()) != hipSuccess)
                return ADAFLO_EHIP;
              std::memcpy(ctx->hox_wg_key, key, sizeof(key));
            }
          const int nb = ctx->hox_wg_counts[0], na = ctx->hox_wg_counts[1], nc = ctx->hox_wg_counts[2];
          A.wg_list   = ctx->hox_wg_list;
          A.wg_offset = phase == 1 ? 0 : ((phase == 0 || phase == 3) ? nb : nb + na);
          A.wg_count  = phase == 1 ? nb : (phase == 0 ? na : (phase == 3 ? na + nc : (phase >= 4 ? 0 : nc)));
          A.fix_mode  = phase == 4 ? 2 : phase; // 1: interface nodes, 2: the others (phases 0 and 3 run no fix-up; 3 = 0 + 2
                                               // without it, 4 = the fix-up of phase 2 alone, 5 = set-up only: the two-stream schedule of comm.hip)
          A.iface     = iface;
          nwg         = A.wg_count;
        }
      const bool   deep      = HOX_DEEP && G::RING && !residual && !varco && !recompute && lin_mode != 2;
      const size_t lds_bytes = (size_t)(deep ? G::LDS_BYTES_DEEP : G::LDS_BYTES);
      const dim3   grid((unsigned)(nwg > 0 ? nwg : 1)), block(NTH);
      hipError_t   err  = hipSuccess;
      hipEvent_t   stop = (ctx->timing && nwg > 0) ? ctx->kernel_timer.start(ctx->stream) : nullptr;
#define HOX_LAUNCH_V(LM, WP, VC)                                                                                \
  {                                                                                                            \
    static bool attr_set = false;                                                                              \
    if (!attr_set)                                                                                             \
      {                                                                                                        \
        err      = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_hox_kernel<K, LM, WP, false, VC>),   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);           \
        attr_set = err == hipSuccess;                                                                          \
      }                                                                                                        \
    if (err == hipSuccess && nwg > 0)                                                                          \
      hipLaunchKernelGGL((ns_hox_kernel<K, LM, WP, false, VC>), grid, block, lds_bytes, ctx->stream, A);       \
  }
#define HOX_LAUNCH_RC(LM, WP)                                                                                       \
  {                                                                                                                \
    static bool attr_set = false;                                                                                  \
    if (!attr_set)                                                                                                 \
      {                                                                                                            \
        err      = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_hox_kernel<K, LM, WP, false, false, true>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);               \
        attr_set = err == hipSuccess;                                                                              \
      }                                                                                                            \
    if (err == hipSuccess && nwg > 0)                                                                              \
      hipLaunchKernelGGL((ns_hox_kernel<K, LM, WP, false, false, true>), grid, block, lds_bytes, ctx->stream, A);  \
  }
#define HOX_LAUNCH(LM, WP)                     \
  {                                            \
    if (varco)                                 \
      HOX_LAUNCH_V(LM, WP, true)               \
    else if (recompute)                        \
      {                                        \
        if constexpr (LM != 2 && HOX_FUSED && HOX_RCP_BUILD) \
          HOX_LAUNCH_RC(LM, WP)                \
      }                                        \
    else                                       \
      HOX_LAUNCH_V(LM, WP, false)              \
  }
#define HOX_LAUNCH_RES_V(LM, VC)                                                                              \
  {                                                                                                           \
    static bool attr_set = false;                                                                             \
    if (!attr_set)                                                                                            \
      {                                                                                                       \
        err      = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_hox_kernel<K, LM, true, true, VC>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);          \
        attr_set = err == hipSuccess;                                                                         \
      }                                                                                                       \
    if (err == hipSuccess && nwg > 0)                                                                         \
      hipLaunchKernelGGL((ns_hox_kernel<K, LM, true, true, VC>), grid, block, lds_bytes, ctx->stream, A);     \
  }
#define HOX_LAUNCH_RES(LM)             \
  {                                    \
    if (varco)                         \
      {                                \
        if constexpr (LM != 2)         \
          HOX_LAUNCH_RES_V(LM, true)   \
        else                           \
          err = hipErrorNotSupported;  \
      }                                \
    else                               \
      HOX_LAUNCH_RES_V(LM, false)      \
  }
#ifndef HOX_EXT_KMAX
#define HOX_EXT_KMAX 5 // (k = 5 at one workgroup per CU: HOX_EXT_LB in ns_hox_kernel.hpp; its fault of rounds 5 / 6 -- spill copies that
                       // hipcc placed under the THEN mask of `if (fl & F_CELL)` -- went with that branch; -DHOX_EXT_KMAX=4 sends the
                       // instance to the generic kernel)
#endif
#define HOX_LAUNCH_RES_EXT_V(LM, VC)                                                                                   \
  {                                                                                                                    \
    static bool attr_set = false;                                                                                      \
    if (!attr_set)                                                                                                     \
      {                                                                                                                \
        err      = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_hox_kernel<K, LM, true, true, VC, false, true>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                   \
        attr_set = err == hipSuccess;                                                                                  \
      }                                                                                                                \
    if (err == hipSuccess && nwg > 0)                                                                                  \
      hipLaunchKernelGGL((ns_hox_kernel<K, LM, true, true, VC, false, true>), grid, block, lds_bytes, ctx->stream, A); \
  }
#define HOX_LAUNCH_RES_EXT(LM)           \
  {                                      \
    if (varco)                           \
      HOX_LAUNCH_RES_EXT_V(LM, true)     \
    else                                 \
      HOX_LAUNCH_RES_EXT_V(LM, false)    \
  }
      if (residual && ext_comb)
        {
          if constexpr (K <= HOX_EXT_KMAX)
            {
              if (lin_mode == 1)
                HOX_LAUNCH_RES_EXT(1)
              else
                HOX_LAUNCH_RES_EXT(2)
            }
          else
            err = hipErrorNotSupported;
        }
      else if (residual)
        switch (lin_mode)
          {
            case 0:
              HOX_LAUNCH_RES(0);
              break;
            case 1:
              HOX_LAUNCH_RES(1);
              break;
            default:
              HOX_LAUNCH_RES(2);
          }
      else if (with_p)
        switch (lin_mode)
          {
            case 0:
              HOX_LAUNCH(0, true);
              break;
            case 1:
              HOX_LAUNCH(1, true);
              break;
            default:
              HOX_LAUNCH(2, true);
          }
      else
        switch (lin_mode)
          {
            case 0:
              HOX_LAUNCH(0, false);
              break;
            case 1:
              HOX_LAUNCH(1, false);
              break;
            default:
              HOX_LAUNCH(2, false);
          }
#undef HOX_LAUNCH
#undef HOX_LAUNCH_RC
#undef HOX_LAUNCH_V
#undef HOX_LAUNCH_RES
#undef HOX_LAUNCH_RES_V
#undef HOX_LAUNCH_RES_EXT
#undef HOX_LAUNCH_RES_EXT_V
      if (err != hipSuccess)
        return ADAFLO_EHIP;
      if (stop)
        (void)hipEventRecord(stop, ctx->stream);
#if HOX_STAMP
      {
        // development aid: medians over the waves of the per-phase cycle sums of the launch just made
        static int calls = 0;
        if (++calls == 20)
          {
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(n_wg * 4 * 10);
            (void)hipMemcpy(h.data(), A.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            const char *name[9] = {"top-of-step wait", "evaluate u", "evaluate p", "quadrature loop", "integrate u", "integrate p", "publish", "barrier + collect", "emit"};
            double      total   = 0;
            for (int j = 0; j < 9; ++j)
              {
                std::vector<double> v;
                for (size_t w = 0; w < n_wg * 4; ++w)
                  if (h[w * 10 + 9])
                    v.push_back((double)h[w * 10 + j] / (double)h[w * 10 + 9]);
                std::sort(v.begin(), v.end());
                std::fprintf(stderr, "hox stamp: %-18s median %8.0f  p10 %8.0f  p90 %8.0f cycles per step\n", name[j],
                             v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
                total += v[v.size() / 2];
              }
            std::fprintf(stderr, "hox stamp: sum of medians %.0f cycles per step\n", total);
          }
      }
#endif
      if (residual && lin_mode != 2)
        {
          if (HOX_RCP_BUILD)
            if (int e = q2_capture_nodal(ctx, ext_comb ? ext_comb : src_u)) // (the recompute-state mode of the vmults of this Newton step)
              return e;
          // the streaming copy is now THE state: the generic copy is stale until somebody asks for it
          ctx->lin_gen++;
          ctx->hox_lin_gen       = ctx->lin_gen;
          ctx->hox_lin_mode      = lin_mode;
          ctx->hox_lin_varco     = varco; // (the variable-coefficient residual lets the coefficient pieces ride along)
          ctx->hox_lin_primary   = true;
          ctx->lin_generic_valid = false;
          ctx->lin_q2_valid      = false;
        }
      if (phase == -1 || phase == 1)
        ctx->kernel_timer.count++;
      if (phase == 0 || phase == 3 || phase == 5)
        return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
      const bool fix_p = with_p && A.integrate_p;
      const long blocks = hox_fix_blocks(A, fix_p); // one per seam row, one per 256 entries of the x-seam planes
      if (blocks > 0 && !(phase == 1 && iface == 0u)) // (no interface: phase 1 has nothing to fix up)
        {
          const long nb = blocks > 256 * 256 ? 256 * 256 : blocks;
          hipLaunchKernelGGL((ns_hox_fixup_kernel<K>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, A, fix_p ? 1 : 0);
        }
      return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
    }
  } // namespace

  bool hox_supported(const adaflo_ctx *ctx)
  {
    return ctx->k >= 3 && ctx->k <= 5 && !ctx->flat && !ctx->indexed; // (constant and, since the two-phase mode, variable coefficients)
  }

  // constant coefficients; Stokes, Newton, Picard-type, and (round 5, time-dependent equations: the old solutions exist)
  // the schemes that linearise about the extrapolated old velocity (navier_stokes_matrix.cc:740-782); (round 6) the
  // projection scheme: the semi-implicit residual without the pressure rows (:644-647, :902-907)
  bool hox_residual_supported(const adaflo_ctx *ctx)
  {
    const int lin = ctx->ns.linearization;
    if (!hox_supported(ctx))
      return false;
    // variable coefficients (two-phase flow, round 6): all three arrays; Newton / Picard-type, and the schemes that linearise
    // about the extrapolated old velocity where their residual mode is built (time-dependent equations)
    if (ctx->rho.p || ctx->mu.p || ctx->damp.p)
      return ctx->rho.p && ctx->mu.p && ctx->damp.p && ctx->ns.physical_type != ADAFLO_STOKES &&
             (lin == ADAFLO_COUPLED_IMPLICIT_NEWTON || lin == ADAFLO_COUPLED_IMPLICIT_PICARD ||
              (HOX_FUSED && HOX_RES_FUSED && ctx->k <= HOX_EXT_KMAX && ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE));
    if (ctx->ns.physical_type == ADAFLO_STOKES || lin == ADAFLO_COUPLED_IMPLICIT_NEWTON || lin == ADAFLO_COUPLED_IMPLICIT_PICARD)
      return true;
    // (k = 5 needs one workgroup per CU: 3.13 ms at 48^3 against 4.96 ms at two and 4.01 ms for the generic kernel)
    return HOX_FUSED && HOX_RES_FUSED && ctx->k <= HOX_EXT_KMAX && ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE;
  }

  int launch_ns_residual_hox(adaflo_ctx *ctx, double *sum_u, double *sum_p, const double *src_u, const double *src_p,
                             const double *old_comb, const double *ext_comb)
  {
    switch (ctx->k)
      {
        case 3:
          return launch_hox<3>(ctx, OP_VMULT, sum_u, sum_p, src_u, src_p, -1, 0u, true, old_comb, ext_comb);
        case 4:
          return launch_hox<4>(ctx, OP_VMULT, sum_u, sum_p, src_u, src_p, -1, 0u, true, old_comb, ext_comb);
        case 5:
          return launch_hox<5>(ctx, OP_VMULT, sum_u, sum_p, src_u, src_p, -1, 0u, true, old_comb, ext_comb);
        default:
          return ADAFLO_EUNSUPPORTED;
      }
  }

  // streaming state of the residual mode -> generic layout [cell][12][q] (entries the scheme does not store stay as they are)
  int hox_unconvert_state(adaflo_ctx *ctx, double *generic, const bool frozen_copy)
  {
    const double *const src  = frozen_copy ? ctx->hox_lin_prec.p : ctx->hox_lin.p;
    const int           mode = frozen_copy ? ctx->hox_lin_prec_mode : ctx->hox_lin_mode;
    if (!src || mode < 0 || mode > 1)
      return ADAFLO_ENOTINIT;
    const bool varco = frozen_copy ? ctx->hox_lin_prec_varco : ctx->hox_lin_varco; // (two more pieces per point: skipped)
    const int  ncx = ctx->desc.ncell[0], ncy = ctx->desc.ncell[1], ncz = ctx->desc.ncell[2], npl = nst_of(mode) / 2,
               npc = npl + (varco ? 2 : 0);
#define HOX_UNCONVERT(K)                                                                                                     \
  {                                                                                                                          \
    using G          = Geo<K>;                                                                                               \
    const int  ngy = (ncy + G::CWY - 1) / G::CWY, ngz = (ncz + G::CWZ - 1) / G::CWZ;                                         \
    const long pieces = (long)ngz * ngy * ncx * G::N * npc * G::CPW * G::NL;                                                 \
    long       nb     = (pieces + 255) / 256;                                                                                \
    if (nb > 256 * 64)                                                                                                       \
      nb = 256 * 64;                                                                                                         \
    hipLaunchKernelGGL((hox_unconvert_state_kernel<K>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, generic, src,            \
                       ncx, ncy, ncz, ngy, ngz, npc, npl);                                                                   \
  }
    switch (ctx->k)
      {
        case 3:
          HOX_UNCONVERT(3);
          break;
        case 4:
          HOX_UNCONVERT(4);
          break;
        case 5:
          HOX_UNCONVERT(5);
          break;
        default:
          return ADAFLO_EUNSUPPORTED;
      }
#undef HOX_UNCONVERT
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_ns_vmult_hox(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u,
                          const double *src_p, const int phase, const uint32_t iface)
  {
    switch (ctx->k)
      {
        case 3:
          return launch_hox<3>(ctx, op, dst_u, dst_p, src_u, src_p, phase, iface);
        case 4:
          return launch_hox<4>(ctx, op, dst_u, dst_p, src_u, src_p, phase, iface);
        case 5:
          return launch_hox<5>(ctx, op, dst_u, dst_p, src_u, src_p, phase, iface);
        default:
          return ADAFLO_EUNSUPPORTED;
      }
  }
} // namespace adaflo_hip
