// ns_hop.hip -- host side of the plane-per-lane Taylor-Hood Q4/Q3 kernel (round 5); the device source and the
// description of the decomposition are in ns_hop_kernel.hpp.
// NavierStokesMatrix::vmult / velocity_vmult with constant coefficients, velocity degree 4
// (source/navier_stokes_matrix.cc:221-262, 337-382, 601-916).
//
// Measured slower than the x-marching kernel and kept for comparison: compiled into the library only with
// -DADAFLO_BUILD_VARIANTS (ADAFLO_BUILD_VARIANTS=1 python adaflo_amd/build.py); the product build has the stubs at the end of this file and
// refuses kernel variant 3.
#include "basis.hpp"
#include "kernels.hpp"

#if defined(ADAFLO_BUILD_VARIANTS)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "hox_intrin.hpp"
#include "ns_hox_kernel.hpp"
#include "ns_hop_kernel.hpp"

namespace adaflo_hip
{
  namespace
  {
    using namespace hop;

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

    // streaming copy of the (frozen) linearisation state, rebuilt when the generic copy it was made from changed; the
    // caller (capi.hip) has brought the generic copy up to date
    int prepare_state(adaflo_ctx *ctx, const bool prec, const int lin_mode, const hox::HXArgs &A, const double **out)
    {
      *out = nullptr;
      if (lin_mode == 2)
        return 0;
      const bool          use_prec = prec && ctx->lin_prec.p;
      const DeviceBuffer &gen      = use_prec ? ctx->lin_prec : ctx->lin;
      DeviceBuffer       &str      = use_prec ? ctx->hop_lin_prec : ctx->hop_lin;
      unsigned long      &have     = use_prec ? ctx->hop_lin_prec_gen : ctx->hop_lin_gen;
      int                &mode     = use_prec ? ctx->hop_lin_prec_mode : ctx->hop_lin_mode;
      const unsigned long want     = use_prec ? ctx->lin_prec_gen : ctx->lin_gen;
      const size_t        need     = (size_t)A.ngz * A.ngy * A.ncx * state_cell_doubles(lin_mode);
      if (str.p && have == want && mode == lin_mode && str.count >= need)
        {
          *out = str.p;
          return 0;
        }
      if (!gen.p || !(use_prec ? ctx->lin_prec_generic_valid : ctx->lin_generic_valid))
        return ADAFLO_ENOTINIT;
      if (int e = ensure(str, need))
        return e;
      long nb = ((long)(need / 2) + 255) / 256;
      if (nb > 256 * 64)
        nb = 256 * 64;
      hipLaunchKernelGGL(hop_convert_state_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, str.p, gen.p, A.ncx, A.ncy,
                         A.ncz, A.ngy, A.ngz, lin_mode);
      if (hipGetLastError() != hipSuccess)
        return ADAFLO_EHIP;
      have = want;
      mode = lin_mode;
      *out = str.p;
      return 0;
    }
  } // namespace

  // velocity degree 4, three dimensions, constant coefficients (the variable-coefficient and residual modes stay on the
  // x-marching kernel of ns_hox.hip)
  bool hop_supported(const adaflo_ctx *ctx, const int op)
  {
    if (ctx->k != 4 || ctx->flat)
      return false;
    return op == OP_VMULT_VELOCITY ? (ctx->lin_prec.p || ctx->hox_lin_prec_primary ? !ctx->rho_prec.p : !ctx->rho.p) : !ctx->rho.p;
  }

  int launch_ns_vmult_hop(adaflo_ctx *ctx, const int op, double *dst_u, double *dst_p, const double *src_u,
                          const double *src_p, const int phase, const uint32_t iface)
  {
    hox::HXArgs A{};
    const int   lin_mode = lin_mode_of(ctx);
    {
      int lx = ctx->hox_lx > 0 ? ctx->hox_lx : 0;
      if (lx == 0)
        {
          // as long as possible (fewer x seams), but enough wave tiles for the 256 CUs x 8 waves
          const long tiles = (long)((ctx->desc.ncell[1] + 1) / 2) * ((ctx->desc.ncell[2] + 1) / 2);
          lx               = ctx->desc.ncell[0];
          while (lx > 4 && tiles * ((ctx->desc.ncell[0] + lx - 1) / lx) < 4096)
            lx = (lx + 1) / 2;
        }
      hop_geometry(A, ctx->desc.ncell, lx);
    }
    if ((size_t)A.nnx * A.nny * A.nnz * 3 >= ((size_t)1 << 32))
      return ADAFLO_EUNSUPPORTED; // 32-bit row offsets
    const NSDev &P      = ctx->ns;
    const bool   stokes = P.physical_type == ADAFLO_STOKES;
    const double gamma  = P.physical_type == ADAFLO_INCOMPRESSIBLE ? P.weight : 0.;
    {
      const Quadrature1D        qu = gauss(N);
      const Shape1D             su = shape_fe_q(K, qu), sp = shape_fe_q(K - 1, qu);
      const std::vector<double> dc = collocation_derivative(qu);
      const double cA = stokes ? 0. : gamma * P.density - P.damping; // :717,:827-835; Stokes: no value terms (:708)
      const double cB = stokes ? 0. : P.tau1 * P.density;
      const std::vector<double> tab = hop_table(su.S.data(), dc.data(), sp.S.data(), qu.w.data(), ctx->desc.h, cA, cB, P.beta,
                                                P.tau_grad_div, P.viscosity * P.tau1 /* :841-845 */);
      if (tab != ctx->hop_tab_host)
        {
          if (int e = ensure(ctx->hop_tab, tab.size()))
            return e;
          // (stream-ordered: earlier launches that read the old table are finished first)
          if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
              copy_to_device_now(ctx->hop_tab.p, tab.data(), tab.size() * sizeof(double)) != hipSuccess)
            return ADAFLO_EHIP;
          ctx->hop_tab_host = tab;
        }
    }
    A.tab         = ctx->hop_tab.p;
    A.integrate_p = P.linearization != ADAFLO_PROJECTION;
    A.con_u       = ctx->brick.con_u;
    A.con_p       = ctx->brick.con_p;
    A.src_u       = src_u;
    A.src_p       = src_p;
    A.dst_u       = dst_u;
    A.dst_p       = dst_p;
    if (int e = prepare_state(ctx, op == OP_VMULT_VELOCITY, lin_mode, A, &A.lin))
      return e;
    const bool   with_p = op == OP_VMULT;
    const size_t n_t    = (size_t)A.tiles_y * A.tiles_z * A.n_chunks;
    if (int e = ensure(ctx->hox_slab_u, n_t * PGeo::RIMU * (K * A.LX + 1) * 3))
      return e;
    if (int e = ensure(ctx->hox_xslab_u, n_t * PGeo::TNY * PGeo::TNZ * 3))
      return e;
    if (int e = ensure(ctx->hox_slab_p, n_t * PGeo::RIMP * (KP * A.LX + 1)))
      return e;
    if (int e = ensure(ctx->hox_xslab_p, n_t * PGeo::TPY * PGeo::TPZ))
      return e;
#if HOP_STAMP
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
      if (int e = launch_prepare_dst(ctx, dst_p, src_p, ctx->n_nodes_p, 1, A.npx, A.npy, A.npz, A.con_p, -1., true))
        return e;
    long nt = (long)n_t;
    if (phase >= 0)
      {
        // tile list [interface | interior A | interior B], cached per (grid, iface)
        const long key[4] = {A.tiles_y, A.tiles_z, A.n_chunks, (long)iface};
        if (!ctx->hop_wg_list || std::memcmp(key, ctx->hop_wg_key, sizeof(key)) != 0)
          {
            std::vector<int> bnd;
            hox::hox_wg_lists(A, iface, bnd, ctx->hop_wg_counts);
            if (ctx->hop_wg_list)
              (void)hipFree(ctx->hop_wg_list);
            ctx->hop_wg_list = nullptr;
            if (hipMalloc(&ctx->hop_wg_list, sizeof(int) * (bnd.size() + 1)) != hipSuccess)
              return ADAFLO_ENOMEM;
            if (copy_to_device_now(ctx->hop_wg_list, bnd.data(), sizeof(int) * bnd.size()) != hipSuccess)
              return ADAFLO_EHIP;
            std::memcpy(ctx->hop_wg_key, key, sizeof(key));
          }
        const int nb = ctx->hop_wg_counts[0], na = ctx->hop_wg_counts[1], nc = ctx->hop_wg_counts[2];
        A.wg_list   = ctx->hop_wg_list;
        A.wg_offset = phase == 1 ? 0 : ((phase == 0 || phase == 3) ? nb : nb + na);
        A.wg_count  = phase == 1 ? nb : (phase == 0 ? na : (phase == 3 ? na + nc : (phase >= 4 ? 0 : nc)));
        A.fix_mode  = phase == 4 ? 2 : phase; // 1: interface nodes, 2: the others (phases 0 and 3 run no fix-up; 3 = 0 + 2
                                               // without it, 4 = the fix-up of phase 2 alone, 5 = set-up only: the two-stream schedule of comm.hip)
        A.iface     = iface;
        nt          = A.wg_count;
      }
    const size_t lds_bytes = (size_t)LDS_BYTES;
    const long   nwg       = (nt + NW - 1) / NW;
    const dim3   grid((unsigned)(nwg > 0 ? nwg : 1)), block(NTH);
    hipError_t   err  = hipSuccess;
    hipEvent_t   stop = (ctx->timing && nwg > 0) ? ctx->kernel_timer.start(ctx->stream) : nullptr;
#define HOP_LAUNCH(LM, WP)                                                                                  \
  {                                                                                                         \
    static bool attr_set = false;                                                                           \
    if (!attr_set)                                                                                          \
      {                                                                                                     \
        err      = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_hop_kernel<LM, WP>),              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);        \
        attr_set = err == hipSuccess;                                                                       \
      }                                                                                                     \
    if (err == hipSuccess && nwg > 0)                                                                       \
      hipLaunchKernelGGL((ns_hop_kernel<LM, WP>), grid, block, lds_bytes, ctx->stream, A);                  \
  }
    if (with_p)
      switch (lin_mode)
        {
          case 0:
            HOP_LAUNCH(0, true);
            break;
          case 1:
            HOP_LAUNCH(1, true);
            break;
          default:
            HOP_LAUNCH(2, true);
        }
    else
      switch (lin_mode)
        {
          case 0:
            HOP_LAUNCH(0, false);
            break;
          case 1:
            HOP_LAUNCH(1, false);
            break;
          default:
            HOP_LAUNCH(2, false);
        }
#undef HOP_LAUNCH
    if (err != hipSuccess)
      return ADAFLO_EHIP;
    if (stop)
      (void)hipEventRecord(stop, ctx->stream);
#if HOP_STAMP
    {
      // development aid: medians over the waves of the per-phase cycle sums of the launch just made
      static int calls = 0;
      if (++calls == 20)
        {
          (void)hipDeviceSynchronize();
          std::vector<unsigned long long> h(n_t * 10);
          (void)hipMemcpy(h.data(), A.stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
          const char *name[8] = {"top of step", "evaluate u", "evaluate p", "quadrature loop", "integrate u", "integrate p", "carry", "emit + node loads"};
          double      total   = 0;
          for (int j = 0; j < 8; ++j)
            {
              std::vector<double> v;
              for (size_t w = 0; w < n_t; ++w)
                if (h[w * 10 + 9])
                  v.push_back((double)h[w * 10 + j] / (double)h[w * 10 + 9]);
              std::sort(v.begin(), v.end());
              std::fprintf(stderr, "hop stamp: %-18s median %8.0f  p10 %8.0f  p90 %8.0f cycles per step\n", name[j],
                           v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
              total += v[v.size() / 2];
            }
          std::fprintf(stderr, "hop stamp: sum of medians %.0f cycles per step\n", total);
        }
    }
#endif
    if (phase == -1 || phase == 1)
      ctx->kernel_timer.count++;
    if (phase == 0 || phase == 3 || phase == 5)
      return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
    const bool fix_p  = with_p && A.integrate_p;
    const long blocks = hox::hox_fix_blocks(A, fix_p); // one per seam row, one per 256 entries of the x-seam planes
    if (blocks > 0 && !(phase == 1 && iface == 0u))    // (no interface: phase 1 has nothing to fix up)
      {
        const long nb = blocks > 256 * 256 ? 256 * 256 : blocks;
        hipLaunchKernelGGL((hox::ns_hox_fixup_kernel<K, PGeo>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, A, fix_p ? 1 : 0);
      }
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }
} // namespace adaflo_hip
#else
namespace adaflo_hip
{
  bool hop_supported(const adaflo_ctx *, const int) { return false; }
  int  launch_ns_vmult_hop(adaflo_ctx *, const int, double *, double *, const double *, const double *, const int, const uint32_t)
  {
    return ADAFLO_EUNSUPPORTED;
  }
} // namespace adaflo_hip
#endif
