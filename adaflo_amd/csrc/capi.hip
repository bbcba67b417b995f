// capi.hip -- implementation of the C ABI declared in include/adaflo_hip.h.
//
// Each adaflo_ns_* function reproduces the *wrapper* part of the corresponding
// NavierStokesMatrix<dim> method (zeroing, dispatch on degree, constrained
// rows, mean projection, timer) -- source/navier_stokes_matrix.cc:191-483 --
// around the HIP cell kernels.
#include <chrono>
#include <cstring>

#include "kernels.hpp"

using namespace adaflo_hip;

namespace
{
  thread_local std::string g_create_error;

  int fail(adaflo_ctx *ctx, const int code, const std::string &msg)
  {
    if (ctx)
      ctx->last_error = msg;
    else
      g_create_error = msg;
    return code;
  }

#define HIP_TRY(ctx, expr)                                                          \
  do                                                                                \
    {                                                                               \
      const hipError_t e_ = (expr);                                                 \
      if (e_ != hipSuccess)                                                         \
        return fail(ctx, ADAFLO_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                               \
  while (0)

#define CHECK_CTX(ctx)                     \
  if (!(ctx))                              \
    return fail(nullptr, ADAFLO_ENOTINIT, "null context")

#define TRY(ctx, expr, what)                 \
  do                                         \
    {                                        \
      const int e_ = (expr);                 \
      if (e_ != 0)                           \
        return fail(ctx, e_, what);          \
    }                                        \
  while (0)

  int alloc(adaflo_ctx *ctx, DeviceBuffer &b, const size_t count)
  {
    if (b.count == count && b.p)
      return 0;
    if (b.p)
      (void)hipFree(b.p);
    b.p     = nullptr;
    b.count = 0;
    if (count == 0)
      return 0;
    if (hipMalloc(&b.p, count * sizeof(double)) != hipSuccess)
      return fail(ctx, ADAFLO_ENOMEM, "hipMalloc failed for quadrature-point state");
    b.count = count;
    return 0;
  }

  void release(DeviceBuffer &b)
  {
    if (b.p)
      (void)hipFree(b.p);
    b.p     = nullptr;
    b.count = 0;
  }

  int upload(adaflo_ctx *ctx, double **dptr, const std::vector<double> &host)
  {
    HIP_TRY(ctx, hipMalloc(dptr, host.size() * sizeof(double)));
    HIP_TRY(ctx, copy_to_device_now(*dptr, host.data(), host.size() * sizeof(double)));
    return 0;
  }

  // time one operator application with a pair of events (get_matvec_statistics)
  struct ScopedTimer
  {
    adaflo_ctx *ctx;
    EventTimer &t;
    hipEvent_t  stop = nullptr;
    ScopedTimer(adaflo_ctx *c, EventTimer &timer)
      : ctx(c)
      , t(timer)
    {
      if (ctx->timing)
        stop = t.start(ctx->stream);
    }
    ~ScopedTimer()
    {
      if (stop)
        (void)hipEventRecord(stop, ctx->stream);
      t.count++;
    }
  };

  NSArgs make_ns_args(adaflo_ctx *ctx, const bool prec_state)
  {
    NSArgs a{};
    a.brick   = ctx->brick;
    a.ns      = ctx->ns;
    a.lin     = prec_state && ctx->lin_prec.p && ctx->lin_prec_generic_valid ? ctx->lin_prec.p : ctx->lin.p;
    // velocity_vmult swaps ALL FOUR arrays as soon as a frozen linearisation exists (navier_stokes_matrix.cc:349-356): a state
    // frozen without coefficients is applied with CONSTANT coefficients even if coefficient arrays were set afterwards
    // (found by tests/test_state_machine_gpu.py, round 6: the current arrays were used); no frozen state: the current ones
    const bool frozen = prec_state && (ctx->lin_prec.p || ctx->lin_q2_prec.p || ctx->hox_lin_prec_primary || ctx->rho_prec.p);
    a.rho     = frozen ? ctx->rho_prec.p : ctx->rho.p;
    a.mu      = frozen ? ctx->mu_prec.p : ctx->mu.p;
    a.damp    = frozen ? ctx->damp_prec.p : ctx->damp.p;
    a.tab     = ctx->d_tab_u;
    a.n_cells = ctx->n_cells;
    return a;
  }

  bool needs_lin(const adaflo_ctx *ctx)
  {
    return ctx->ns.linearization != ADAFLO_COUPLED_VELOCITY_EXPLICIT &&
           ctx->ns.physical_type != ADAFLO_STOKES;
  }

  // is a linearisation point stored (in either layout)?
  bool has_lin(const adaflo_ctx *ctx)
  {
    return (ctx->lin.p && ctx->lin_generic_valid) || (ctx->lin_q2.p && ctx->lin_q2_valid) || ctx->hox_lin_primary || ctx->lin_q2_deferred;
  }

  // the generic copy [cell][12][q] of the state, rebuilt from the streaming copy the sweep-kernel
  // residual wrote if it is stale
  int ensure_lin_generic(adaflo_ctx *ctx)
  {
    if (!ctx->lin_generic_valid && ctx->lin_q2_deferred) // (lazy state of the Q2/Q1 residual: lay it out first)
      if (int e = q2_materialize_state(ctx))
        return e;
    if (!ctx->lin_generic_valid && ctx->hox_lin_primary)
      {
        // the residual mode of the x-marching kernel left the state in its streaming layout only
        const size_t count = (size_t)ctx->n_cells * ctx->nq_u * NLIN;
        if (ctx->lin.count != count)
          {
            if (int e = alloc(ctx, ctx->lin, count))
              return e;
            if (hipMemsetAsync(ctx->lin.p, 0, count * sizeof(double), ctx->stream) != hipSuccess)
              return ADAFLO_EHIP;
          }
        if (int e = hox_unconvert_state(ctx, ctx->lin.p))
          return e;
        ctx->lin_generic_valid = true; // (same content: lin_gen stays, the streaming copy remains current)
        return 0;
      }
    if (ctx->lin_generic_valid || !(ctx->lin_q2.p && ctx->lin_q2_valid) || ctx->lin_q2_varco)
      return 0;
    const size_t count = (size_t)ctx->n_cells * ctx->nq_u * NLIN;
    if (ctx->lin.count != count)
      {
        if (int e = alloc(ctx, ctx->lin, count))
          return e;
        if (hipMemsetAsync(ctx->lin.p, 0, count * sizeof(double), ctx->stream) != hipSuccess)
          return ADAFLO_EHIP;
      }
    if (int e = q2_unconvert_state(ctx, ctx->lin.p, ctx->lin_q2.p, ctx->lin_q2_mode))
      return e;
    ctx->lin_generic_valid = true;
    ctx->lin_gen++;
    return 0;
  }
  int ensure_lin_prec_generic(adaflo_ctx *ctx)
  {
    if (!ctx->lin_prec_generic_valid && ctx->hox_lin_prec_primary)
      {
        const size_t count = (size_t)ctx->n_cells * ctx->nq_u * NLIN;
        if (ctx->lin_prec.count != count)
          {
            if (int e = alloc(ctx, ctx->lin_prec, count))
              return e;
            if (hipMemsetAsync(ctx->lin_prec.p, 0, count * sizeof(double), ctx->stream) != hipSuccess)
              return ADAFLO_EHIP;
          }
        if (int e = hox_unconvert_state(ctx, ctx->lin_prec.p, true))
          return e;
        ctx->lin_prec_generic_valid = true;
        return 0;
      }
    if (ctx->lin_prec_generic_valid || !ctx->lin_q2_prec.p || ctx->lin_q2_prec_varco)
      return 0;
    const size_t count = (size_t)ctx->n_cells * ctx->nq_u * NLIN;
    if (ctx->lin_prec.count != count)
      {
        if (int e = alloc(ctx, ctx->lin_prec, count))
          return e;
        if (hipMemsetAsync(ctx->lin_prec.p, 0, count * sizeof(double), ctx->stream) != hipSuccess)
          return ADAFLO_EHIP;
      }
    if (int e = q2_unconvert_state(ctx, ctx->lin_prec.p, ctx->lin_q2_prec.p, ctx->lin_q2_prec_mode))
      return e;
    ctx->lin_prec_generic_valid = true;
    return 0;
  }

  int nn(const adaflo_ctx *ctx, const int degree, const int d)
  {
    return ((ctx->flat && d == 2) || (ctx->flat_y && d == 1)) ? 1 : degree * ctx->desc.ncell[d] + 1;
  }
} // namespace

// entry points that need the structured brick (level set, sweep kernels, halo plans, fast diagonalisation)
#define BRICK_ONLY(ctx)                                                                                                   \
  if ((ctx)->indexed)                                                                                                     \
    return fail(ctx, ADAFLO_EUNSUPPORTED, "not available on an indexed context (adaflo_ctx_create_indexed): needs the structured brick")

extern "C" {

const char *adaflo_last_error(const adaflo_ctx *ctx)
{
  return ctx ? ctx->last_error.c_str() : g_create_error.c_str();
}

static int ctx_create_impl(const adaflo_brick_desc *desc, adaflo_ctx *ctx)
{
  // dim = 2 (NavierStokesMatrix<2>, navier_stokes_matrix.cc:1211; all level-set golden outputs of the reference are
  // 2D): the generic kernels with a FLAT third direction -- one node, one quadrature point of weight 1, h_z = 1, the
  // third velocity component constrained everywhere -- i.e. the same templates as dim = 3 (fe_kernels.hpp: SumFac<.., ZF>).
  // Velocity degree 2 or 3 (the reference's 2D tests); the specialised 3D kernels are not used.
  // dim = 1 (NavierStokesMatrix<1>, :1210; tests/1d_flow*.prm): the second direction flat as well, no level-set spaces.
  if (desc->dim < 1 || desc->dim > 3)
    return fail(nullptr, ADAFLO_EUNSUPPORTED, "dim must be 1, 2 or 3");
  if (desc->dim == 2 && (desc->ncell[2] != 1 || desc->velocity_degree > 3))
    return fail(nullptr, ADAFLO_EUNSUPPORTED, "dim = 2: ncell[2] must be 1 and the velocity degree 2 or 3");
  if (desc->dim == 1 && (desc->ncell[1] != 1 || desc->ncell[2] != 1 || desc->velocity_degree > 3 || desc->ls_degree > 0))
    return fail(nullptr, ADAFLO_EUNSUPPORTED, "dim = 1: ncell[1] = ncell[2] = 1, velocity degree 2 or 3, no level-set spaces");
  // EXPAND_OPERATIONS (source/navier_stokes_matrix.cc:64-82): degree_p = 1 .. 5, i.e. velocity degrees 2 .. 6
  if (desc->velocity_degree < 2 || desc->velocity_degree > 6)
    return fail(nullptr, ADAFLO_EUNSUPPORTED, "velocity degree must be in [2,6] (reference: ExcNotImplemented)");
  for (int d = 0; d < 3; ++d)
    if (desc->ncell[d] < 1 || !(desc->h[d] > 0.))
      return fail(nullptr, ADAFLO_EINVAL, "invalid brick extents");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, ADAFLO_EHIP, "no HIP device available (the engine has no CPU fallback)");
  if (desc->device < 0 || desc->device >= ndev)
    return fail(nullptr, ADAFLO_EINVAL, "invalid device ordinal");

  ctx->desc       = *desc;
  ctx->flat       = desc->dim <= 2;
  ctx->flat_y     = desc->dim == 1;
  if (ctx->flat_y)
    {
      ctx->desc.h[1] = 1.;
      // ... nor does the second one in 1D (every node lies on the z faces)
      ctx->desc.velocity_constrained |= (1u << (3 * 4 + 1)) | (1u << (3 * 5 + 1));
    }
  if (ctx->flat)
    {
      ctx->desc.h[2] = 1.; // the single z point has weight 1: integrals over the brick are integrals over the x-y domain
      ctx->variant   = 0;  // generic kernels only
      ctx->pc_inner  = 0;  // (the fast-diagonalisation inverses are 3D)
      // the third velocity component does not exist in 2D: constrained on both z faces = everywhere
      ctx->desc.velocity_constrained |= (1u << (3 * 4 + 2)) | (1u << (3 * 5 + 2));
    }
  HIP_TRY(nullptr, hipSetDevice(desc->device));
  if (desc->stream)
    ctx->stream = static_cast<hipStream_t>(desc->stream);
  else
    {
      HIP_TRY(nullptr, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
      ctx->own_stream = true;
    }
  ctx->k = desc->velocity_degree;
  ctx->s = desc->ls_degree;
  const int k = ctx->k;
  ctx->n_cells   = (int64_t)desc->ncell[0] * desc->ncell[1] * desc->ncell[2];
  ctx->n_nodes_u = (int64_t)nn(ctx, k, 0) * nn(ctx, k, 1) * nn(ctx, k, 2);
  ctx->n_nodes_p = (int64_t)nn(ctx, k - 1, 0) * nn(ctx, k - 1, 1) * nn(ctx, k - 1, 2);
  ctx->n_nodes_ls = ctx->s > 0 ? (int64_t)nn(ctx, ctx->s, 0) * nn(ctx, ctx->s, 1) * nn(ctx, ctx->s, 2) : 0;
  ctx->nq_u       = (k + 1) * (ctx->flat_y ? 1 : k + 1) * (ctx->flat ? 1 : k + 1);
  for (int d = 0; d < 3; ++d)
    {
      ctx->brick.ncell[d] = desc->ncell[d];
      ctx->brick.h[d]     = ctx->desc.h[d];
    }
  ctx->brick.con_u  = ctx->desc.velocity_constrained;
  ctx->brick.con_p  = desc->pressure_constrained;
  ctx->brick.con_ls = desc->ls_constrained;

  // 1D tables: quad_index_u = QGauss(k+1), quad_index_p = QGauss(k)
  {
    const Quadrature1D  qu = gauss(k + 1), qp = gauss(k);
    const Shape1D       su = shape_fe_q(k, qu), sp = shape_fe_q(k - 1, qu), spp = shape_fe_q(k - 1, qp);
    std::vector<double> tab;
    tab.insert(tab.end(), su.S.begin(), su.S.end());
    tab.insert(tab.end(), su.D.begin(), su.D.end());
    tab.insert(tab.end(), sp.S.begin(), sp.S.end());
    tab.insert(tab.end(), sp.D.begin(), sp.D.end());
    tab.insert(tab.end(), qu.w.begin(), qu.w.end());
    TRY(nullptr, upload(nullptr, &ctx->d_tab_u, tab), g_create_error);
    std::vector<double> tabp;
    tabp.insert(tabp.end(), spp.S.begin(), spp.S.end());
    tabp.insert(tabp.end(), spp.D.begin(), spp.D.end());
    tabp.insert(tabp.end(), qp.w.begin(), qp.w.end());
    TRY(nullptr, upload(nullptr, &ctx->d_tab_pp, tabp), g_create_error);
  }

  if (ctx->s > 0)
    {
      if (ctx->s > 4)
        return fail(nullptr, ADAFLO_EUNSUPPORTED, "level-set subdivisions must be in [1,4]");
      const Quadrature1D  ql = gauss2_iterated(ctx->s);
      const Shape1D       sl = shape_fe_q_iso_q1(ctx->s, ql), sv = shape_fe_q(k, ql);
      std::vector<double> tab;
      tab.insert(tab.end(), sl.S.begin(), sl.S.end());
      tab.insert(tab.end(), sl.D.begin(), sl.D.end());
      tab.insert(tab.end(), ql.w.begin(), ql.w.end());
      tab.insert(tab.end(), sv.S.begin(), sv.S.end());
      TRY(nullptr, upload(nullptr, &ctx->d_tab_ls, tab), g_create_error);
      ctx->ls = LSDev{0., 0., 1., 1., -1., 0., 1.};
    }

  // default parameters = FlowParameters defaults relevant to the kernels
  ctx->ns = NSDev{ADAFLO_INCOMPRESSIBLE, ADAFLO_COUPLED_IMPLICIT_NEWTON, 0.5, 0., 1., 1., 0., 0.,
                  1., -1., 0., 1., 1., 0.};

  // pressure constant mode 0, source/navier_stokes_matrix.cc:117-168
  if (desc->pressure_average_fix)
    {
      const long np = ctx->n_nodes_p;
      HIP_TRY(nullptr, hipMalloc(&ctx->d_p_weights, np * sizeof(double)));
      HIP_TRY(nullptr, hipMalloc(&ctx->d_p_modes, np * sizeof(double)));
      // pres_mass = cell_loop(local_pressure_mass_weight) on a zero vector
      TRY(nullptr, launch_fill(ctx, ctx->d_p_weights, 0., np), "fill failed");
      ScalarArgs sa{};
      sa.brick   = ctx->brick; // distribute_local_to_global skips constrained rows (stay 0)
      sa.ns      = ctx->ns;
      sa.dst     = ctx->d_p_weights;
      sa.tab     = ctx->d_tab_pp;
      sa.n_cells = ctx->n_cells;
      sa.mode    = SC_MASS_WEIGHT;
      sa.nq_u3   = ctx->nq_u;
      TRY(nullptr, launch_ns_scalar_generic(ctx, sa), "mass-weight kernel launch failed");
      TRY(nullptr, launch_fill(ctx, ctx->d_p_modes, 1., np), "fill failed");
      if (ctx->brick.con_p != 0u)
        TRY(nullptr,
            launch_prepare_dst(ctx, ctx->d_p_modes, ctx->d_p_modes, np, 1, nn(ctx, k - 1, 0),
                               nn(ctx, k - 1, 1), nn(ctx, k - 1, 2), ctx->brick.con_p, 0., false),
            "mask failed");
      const double mw = host_dot(ctx, ctx->d_p_modes, ctx->d_p_weights, np);
      ctx->inv_p_weight = 1. / mw;
    }
  HIP_TRY(nullptr, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_ctx_destroy(adaflo_ctx *ctx);

// Indexed context (first alternative; include/adaflo_hip.h): the adapter's tables instead of the brick.
// Built on the brick set-up of a one-cell brick (stream, 1D tables, default parameters), then the sizes and tables replaced.
static int ctx_create_indexed_impl(const adaflo_indexed_desc *d, adaflo_ctx *ctx)
{
  const int k = d->velocity_degree;
  if (k < 2 || k > 6)
    return fail(nullptr, ADAFLO_EUNSUPPORTED, "velocity degree must be in [2,6] (reference: ExcNotImplemented)");
  if (d->n_cells < 1 || d->n_nodes_u < 1 || d->n_nodes_p < 1 || !d->cell_nodes_u || !d->cell_nodes_p || !d->constrained_u ||
      !d->constrained_p || d->n_colours < 1 || !d->colour_offsets)
    return fail(nullptr, ADAFLO_EINVAL, "indexed context: missing table");
  if (d->colour_offsets[0] != 0 || d->colour_offsets[d->n_colours] != d->n_cells)
    return fail(nullptr, ADAFLO_EINVAL, "indexed context: the colour ranges must cover the cells [0, n_cells)");
  if (d->n_nodes_u > 0x7fffffffLL / 3 || d->n_cells > 0x7fffffffLL)
    return fail(nullptr, ADAFLO_EUNSUPPORTED, "indexed context: 32-bit node tables");
  const long nu3 = (long)(k + 1) * (k + 1) * (k + 1), np3 = (long)k * k * k;
  // hanging-node constraints (optional): CSR rows of (master, weight); masters are regular nodes
  const int64_t nh_u = d->n_hanging_u, nh_p = d->n_hanging_p;
  if (nh_u < 0 || nh_p < 0 || nh_u > 0x7fffffffLL || nh_p > 0x7fffffffLL ||
      (nh_u > 0 && (!d->hanging_ptr_u || !d->hanging_master_u || !d->hanging_weight_u)) ||
      (nh_p > 0 && (!d->hanging_ptr_p || !d->hanging_master_p || !d->hanging_weight_p)))
    return fail(nullptr, ADAFLO_EINVAL, "indexed context: missing hanging-node table");
  auto check_hanging = [](const int64_t nh, const int64_t *ptr, const int *master, const int64_t n_nodes) -> bool {
    if (nh == 0)
      return true;
    if (ptr[0] != 0)
      return false;
    for (int64_t h = 0; h < nh; ++h)
      {
        if (ptr[h + 1] <= ptr[h])
          return false;
        for (int64_t j = ptr[h]; j < ptr[h + 1]; ++j)
          if (master[j] < 0 || master[j] >= n_nodes)
            return false;
      }
    return true;
  };
  if (!check_hanging(nh_u, d->hanging_ptr_u, d->hanging_master_u, d->n_nodes_u) ||
      !check_hanging(nh_p, d->hanging_ptr_p, d->hanging_master_p, d->n_nodes_p))
    return fail(nullptr, ADAFLO_EINVAL, "indexed context: hanging-node rows must be non-empty, ascending and name regular nodes");
  // the tables must be usable: indices in range, and no node twice inside one colour (the scatter has no atomics; a
  // hanging entry writes to its masters)
  {
    std::vector<int> seen_u((size_t)d->n_nodes_u, -1), seen_p((size_t)d->n_nodes_p, -1);
    auto visit = [](std::vector<int> &seen, const int node, const int cell_in_colour) -> bool {
      int &s = seen[node];
      if (s >= 0 && s != cell_in_colour)
        return false;
      s = cell_in_colour;
      return true;
    };
    for (int c = 0; c < d->n_colours; ++c)
      {
        if (d->colour_offsets[c + 1] < d->colour_offsets[c])
          return fail(nullptr, ADAFLO_EINVAL, "indexed context: colour offsets must not decrease");
        for (int64_t cell = d->colour_offsets[c]; cell < d->colour_offsets[c + 1]; ++cell)
          {
            for (long l = 0; l < nu3; ++l)
              {
                const int n = d->cell_nodes_u[cell * nu3 + l];
                if (n < -nh_u || n >= d->n_nodes_u)
                  return fail(nullptr, ADAFLO_EINVAL, "indexed context: velocity node index out of range");
              }
            for (long l = 0; l < np3; ++l)
              {
                const int n = d->cell_nodes_p[cell * np3 + l];
                if (n < -nh_p || n >= d->n_nodes_p)
                  return fail(nullptr, ADAFLO_EINVAL, "indexed context: pressure node index out of range");
              }
          }
        // two cells of colour c sharing a node: mark the nodes of every cell with its number inside the colour and look
        // for a second owner
        for (int64_t cell = d->colour_offsets[c]; cell < d->colour_offsets[c + 1]; ++cell)
          {
            const int me = (int)(cell - d->colour_offsets[c]);
            for (long l = 0; l < nu3; ++l)
              {
                const int n  = d->cell_nodes_u[cell * nu3 + l];
                bool      ok = true;
                if (n >= 0)
                  ok = visit(seen_u, n, me);
                else
                  for (int64_t j = d->hanging_ptr_u[-1 - n]; j < d->hanging_ptr_u[-n] && ok; ++j)
                    ok = visit(seen_u, d->hanging_master_u[j], me);
                if (!ok)
                  return fail(nullptr, ADAFLO_EINVAL, "indexed context: two cells of one colour share a velocity node");
              }
            for (long l = 0; l < np3; ++l)
              {
                const int n  = d->cell_nodes_p[cell * np3 + l];
                bool      ok = true;
                if (n >= 0)
                  ok = visit(seen_p, n, me);
                else
                  for (int64_t j = d->hanging_ptr_p[-1 - n]; j < d->hanging_ptr_p[-n] && ok; ++j)
                    ok = visit(seen_p, d->hanging_master_p[j], me);
                if (!ok)
                  return fail(nullptr, ADAFLO_EINVAL, "indexed context: two cells of one colour share a pressure node");
              }
          }
        std::fill(seen_u.begin(), seen_u.end(), -1);
        std::fill(seen_p.begin(), seen_p.end(), -1);
      }
  }
  adaflo_brick_desc b{};
  b.dim = 3;
  for (int i = 0; i < 3; ++i)
    {
      b.ncell[i] = 1;
      b.h[i]     = d->cell_extents ? 1. : d->h[i];
    }
  b.velocity_degree      = k;
  b.device               = d->device;
  b.stream               = d->stream;
  b.pressure_average_fix = 0; // (computed below, through the tables)
  if (int e = ctx_create_impl(&b, ctx))
    return e;
  ctx->indexed   = true;
  ctx->variant   = 0; // generic kernels only
  ctx->pc_inner  = 0;
  ctx->n_cells   = d->n_cells;
  ctx->n_nodes_u = d->n_nodes_u;
  ctx->n_nodes_p = d->n_nodes_p;
  ctx->idx_colour_off.assign(d->colour_offsets, d->colour_offsets + d->n_colours + 1);
  auto up = [&](auto **dst, const auto *src, const size_t count) -> int {
    if (hipMalloc((void **)dst, count * sizeof(**dst)) != hipSuccess)
      return ADAFLO_ENOMEM;
    return copy_to_device_now(*dst, src, count * sizeof(**dst)) == hipSuccess ? 0 : ADAFLO_EHIP;
  };
  TRY(nullptr, up(&ctx->d_idx_u, d->cell_nodes_u, (size_t)d->n_cells * nu3), "indexed context: table upload failed");
  TRY(nullptr, up(&ctx->d_idx_p, d->cell_nodes_p, (size_t)d->n_cells * np3), "indexed context: table upload failed");
  TRY(nullptr, up(&ctx->d_flag_u, d->constrained_u, (size_t)d->n_nodes_u * 3), "indexed context: table upload failed");
  TRY(nullptr, up(&ctx->d_flag_p, d->constrained_p, (size_t)d->n_nodes_p), "indexed context: table upload failed");
  if (d->cell_extents)
    {
      for (int64_t i = 0; i < 3 * d->n_cells; ++i)
        if (!(d->cell_extents[i] > 0.))
          return fail(nullptr, ADAFLO_EINVAL, "indexed context: cell extents must be positive");
      TRY(nullptr, up(&ctx->d_cell_h, d->cell_extents, (size_t)d->n_cells * 3), "indexed context: table upload failed");
    }
  ctx->brick.idx_u  = ctx->d_idx_u;
  ctx->brick.idx_p  = ctx->d_idx_p;
  ctx->brick.flag_u = ctx->d_flag_u;
  ctx->brick.flag_p = ctx->d_flag_p;
  ctx->brick.cell_h = ctx->d_cell_h;
  if (nh_u > 0)
    {
      TRY(nullptr, up(&ctx->d_hang_ptr_u, (const long *)d->hanging_ptr_u, (size_t)nh_u + 1), "indexed context: table upload failed");
      TRY(nullptr, up(&ctx->d_hang_master_u, d->hanging_master_u, (size_t)d->hanging_ptr_u[nh_u]), "indexed context: table upload failed");
      TRY(nullptr, up(&ctx->d_hang_weight_u, d->hanging_weight_u, (size_t)d->hanging_ptr_u[nh_u]), "indexed context: table upload failed");
      ctx->brick.hang_ptr_u = ctx->d_hang_ptr_u, ctx->brick.hang_master_u = ctx->d_hang_master_u, ctx->brick.hang_weight_u = ctx->d_hang_weight_u;
    }
  if (nh_p > 0)
    {
      TRY(nullptr, up(&ctx->d_hang_ptr_p, (const long *)d->hanging_ptr_p, (size_t)nh_p + 1), "indexed context: table upload failed");
      TRY(nullptr, up(&ctx->d_hang_master_p, d->hanging_master_p, (size_t)d->hanging_ptr_p[nh_p]), "indexed context: table upload failed");
      TRY(nullptr, up(&ctx->d_hang_weight_p, d->hanging_weight_p, (size_t)d->hanging_ptr_p[nh_p]), "indexed context: table upload failed");
      ctx->brick.hang_ptr_p = ctx->d_hang_ptr_p, ctx->brick.hang_master_p = ctx->d_hang_master_p, ctx->brick.hang_weight_p = ctx->d_hang_weight_p;
    }
  // pressure constant mode 0, source/navier_stokes_matrix.cc:117-168 (as in ctx_create_impl, through the tables)
  if (d->pressure_average_fix)
    {
      const long np = ctx->n_nodes_p;
      HIP_TRY(nullptr, hipMalloc(&ctx->d_p_weights, np * sizeof(double)));
      HIP_TRY(nullptr, hipMalloc(&ctx->d_p_modes, np * sizeof(double)));
      TRY(nullptr, launch_fill(ctx, ctx->d_p_weights, 0., np), "fill failed");
      ScalarArgs sa{};
      sa.brick   = ctx->brick;
      sa.ns      = ctx->ns;
      sa.dst     = ctx->d_p_weights;
      sa.tab     = ctx->d_tab_pp;
      sa.n_cells = ctx->n_cells;
      sa.mode    = SC_MASS_WEIGHT;
      sa.nq_u3   = ctx->nq_u;
      TRY(nullptr, launch_ns_scalar_generic(ctx, sa), "mass-weight kernel launch failed");
      TRY(nullptr, launch_fill(ctx, ctx->d_p_modes, 1., np), "fill failed");
      TRY(nullptr, launch_prepare_dst(ctx, ctx->d_p_modes, ctx->d_p_modes, np, 1, 1, 1, 1, 0u, 0., false), "mask failed");
      const double mw   = host_dot(ctx, ctx->d_p_modes, ctx->d_p_weights, np);
      ctx->inv_p_weight = 1. / mw;
    }
  HIP_TRY(nullptr, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_ctx_create_indexed(const adaflo_indexed_desc *desc, adaflo_ctx **out)
{
  if (!desc || !out)
    return fail(nullptr, ADAFLO_EINVAL, "null argument");
  *out = nullptr;
  adaflo_ctx *ctx = new adaflo_ctx;
  const int   rc  = ctx_create_indexed_impl(desc, ctx);
  if (rc != 0)
    {
      const std::string msg = g_create_error;
      (void)hipGetLastError();
      adaflo_ctx_destroy(ctx);
      g_create_error = msg;
      return rc;
    }
  *out = ctx;
  return 0;
}

int adaflo_ctx_create(const adaflo_brick_desc *desc, adaflo_ctx **out)
{
  if (!desc || !out)
    return fail(nullptr, ADAFLO_EINVAL, "null argument");
  *out = nullptr;
  adaflo_ctx *ctx = new adaflo_ctx;
  const int   rc  = ctx_create_impl(desc, ctx);
  if (rc != 0)
    {
      // (keeps the message of the failure; everything allocated so far goes with the context)
      const std::string msg = g_create_error;
      (void)hipGetLastError();
      adaflo_ctx_destroy(ctx);
      g_create_error = msg;
      return rc;
    }
  *out = ctx;
  return 0;
}

int adaflo_ctx_destroy(adaflo_ctx *ctx)
{
  CHECK_CTX(ctx);
  if (ctx->stream || !ctx->own_stream)
    (void)hipStreamSynchronize(ctx->stream);
  for (DeviceBuffer *b : {&ctx->lin, &ctx->rho, &ctx->mu, &ctx->damp, &ctx->lin_prec, &ctx->rho_prec,
                          &ctx->mu_prec, &ctx->damp_prec, &ctx->lin_q2, &ctx->lin_q2_prec,
                          &ctx->q2_slab_u, &ctx->q2_zslab_u, &ctx->q2_slab_p, &ctx->q2_zslab_p,
                          &ctx->ls_convection, &ctx->ls_normal, &ctx->q1_convection, &ctx->q1_normal, &ctx->q1_normal_nodal, &ctx->q1_velocity_nodal,
                          &ctx->q1_slab, &ctx->q1_zslab, &ctx->pc_inv_u, &ctx->pc_inv_pm, &ctx->pc_inv_pl,
                          &ctx->pc_ones_p, &ctx->pc_tmp_u, &ctx->pc_tmp_p, &ctx->pc_tmp_p2, &ctx->pc_work, &ctx->kr_work, &ctx->kr_basis, &ctx->kr_scalars,
                          &ctx->q1_poisson_coef, &ctx->ho_tab, &ctx->res_sum_u, &ctx->res_sum_p, &ctx->res_old, &ctx->res_ext,
                          &ctx->ls_art_visc, &ctx->ls_stab_vel_sum, &ctx->ls_stab_ls_sum, &ctx->hox_lin, &ctx->hox_lin_prec, &ctx->hox_coef, &ctx->pc_tridiag,
                          &ctx->hox_slab_u, &ctx->hox_xslab_u, &ctx->hox_slab_p, &ctx->hox_xslab_p, &ctx->hox_tab,
                          &ctx->hop_lin, &ctx->hop_lin_prec, &ctx->hop_tab, &ctx->lin_nodal, &ctx->lin_nodal_prec})
    release(*b);
  for (double *p : {ctx->d_tab_u, ctx->d_tab_pp, ctx->d_p_weights, ctx->d_p_modes, ctx->d_scratch,
                    ctx->d_tab_ls, ctx->d_ls_diag, ctx->d_tab_force, ctx->d_tab_maxvel})
    if (p)
      (void)hipFree(p);
  for (void *p : {(void *)ctx->d_idx_u, (void *)ctx->d_idx_p, (void *)ctx->d_flag_u, (void *)ctx->d_flag_p, (void *)ctx->d_cell_h,
                  (void *)ctx->d_hang_ptr_u, (void *)ctx->d_hang_ptr_p, (void *)ctx->d_hang_master_u, (void *)ctx->d_hang_master_p,
                  (void *)ctx->d_hang_weight_u, (void *)ctx->d_hang_weight_p})
    if (p)
      (void)hipFree(p);
  fdm_destroy(ctx);
  if (ctx->q2_wg_list)
    (void)hipFree(ctx->q2_wg_list);
  if (ctx->hox_wg_list)
    (void)hipFree(ctx->hox_wg_list);
  if (ctx->hop_wg_list)
    (void)hipFree(ctx->hop_wg_list);
  if (ctx->h_result)
    (void)hipHostFree(ctx->h_result);
  if (ctx->gs_host)
    (void)hipHostFree(ctx->gs_host);
  if (ctx->gs_dev)
    (void)hipFree(ctx->gs_dev);
  ctx->matvec_timer.destroy();
  ctx->kernel_timer.destroy();
  if (ctx->own_stream && ctx->stream)
    (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

int adaflo_synchronize(adaflo_ctx *ctx)
{
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_set_stream(adaflo_ctx *ctx, void *stream)
{
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream)
    (void)hipStreamDestroy(ctx->stream);
  ctx->own_stream = false;
  ctx->stream     = static_cast<hipStream_t>(stream); // NULL = the legacy default stream
  return 0;
}

void *adaflo_stream(adaflo_ctx *ctx)
{
  return ctx ? ctx->stream : nullptr;
}

int64_t adaflo_n_cells(const adaflo_ctx *ctx) { return ctx ? ctx->n_cells : 0; }
int64_t adaflo_n_dofs_u(const adaflo_ctx *ctx) { return ctx ? 3 * ctx->n_nodes_u : 0; }
int64_t adaflo_n_dofs_p(const adaflo_ctx *ctx) { return ctx ? ctx->n_nodes_p : 0; }
int64_t adaflo_n_dofs_ls(const adaflo_ctx *ctx) { return ctx ? ctx->n_nodes_ls : 0; }
int     adaflo_n_q_points_u(const adaflo_ctx *ctx) { return ctx ? ctx->nq_u : 0; }
int     adaflo_n_q_points_ls(const adaflo_ctx *ctx) { return ctx ? (ctx->flat ? 4 * ctx->s * ctx->s : 8 * ctx->s * ctx->s * ctx->s) : 0; }

int adaflo_malloc(adaflo_ctx *ctx, size_t bytes, void **dptr)
{
  CHECK_CTX(ctx);
  if (hipMalloc(dptr, bytes) != hipSuccess)
    return fail(ctx, ADAFLO_ENOMEM, "hipMalloc failed");
  return 0;
}

int adaflo_free(adaflo_ctx *ctx, void *dptr)
{
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipFree(dptr));
  return 0;
}

int adaflo_copy_h2d(adaflo_ctx *ctx, void *dst, const void *src, size_t bytes)
{
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_copy_d2h(adaflo_ctx *ctx, void *dst, const void *src, size_t bytes)
{
  CHECK_CTX(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_has_kernel_variant(int variant)
{
#if defined(ADAFLO_BUILD_VARIANTS)
  return variant >= 0 && variant <= 4;
#else
  return variant == 0 || variant == 1 || variant == 4;
#endif
}

int adaflo_set_kernel_variant(adaflo_ctx *ctx, int variant)
{
  CHECK_CTX(ctx);
  if (variant < 0 || variant > 4)
    return fail(ctx, ADAFLO_EINVAL, "unknown kernel variant");
  if (ctx->flat && variant != 0)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "dim = 2 runs on the generic kernels (variant 0) only");
  if (ctx->indexed && variant != 0)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "an indexed context runs on the generic kernels (variant 0) only");
  // (2 and 3 select a kernel of their own only for the degrees these kernels exist for; elsewhere they mean what 1 means --
  // for Q2/Q1, 2 also selects the divergence mode of the sweep kernel in divergence_vmult_add)
  if (!adaflo_has_kernel_variant(variant) && ((variant == 2 && ctx->k >= 3 && ctx->k <= 5) || (variant == 3 && ctx->k == 4)))
    return fail(ctx, ADAFLO_EUNSUPPORTED,
                "kernel variants 2 and 3 (the superseded Q3..Q5 kernels) are not in this build of the library: "
                "ADAFLO_BUILD_VARIANTS=1 python adaflo_amd/build.py --force");
  // 3 = 1 with the plane-per-lane kernel for Q4/Q3 constant-coefficient vmult / velocity_vmult (ns_hop.hip)
  // 4 = 1 WITHOUT the recompute-state mode of the Q2/Q1 Newton vmult (ns_q2.hip, RCP): the streaming kernel of rounds 1-4
  ctx->hop          = variant == 3;
  ctx->q2_recompute = variant != 4;
  ctx->variant      = variant >= 3 ? 1 : variant;
  return 0;
}

int adaflo_ns_set_params(adaflo_ctx *ctx, const adaflo_ns_params *p)
{
  CHECK_CTX(ctx);
  if (!p)
    return fail(ctx, ADAFLO_EINVAL, "null params");
  if (p->physical_type < 0 || p->physical_type > 2 || p->linearization < 0 || p->linearization > 4)
    return fail(ctx, ADAFLO_EINVAL, "invalid physical type / linearization");
  if (p->physical_type == ADAFLO_INCOMPRESSIBLE_STATIONARY &&
      p->linearization != ADAFLO_COUPLED_IMPLICIT_NEWTON)
    return fail(ctx, ADAFLO_EINVAL,
                "stationary equation requires coupled implicit Newton (parameters.cc:501-504)");
  // The sweep-kernel residual leaves the linearisation state only in the streaming layout of the scheme it
  // ran with; another scheme re-creates its streaming copy from the generic one: bring that up to date
  // while the layout of the stored copy is still known.
  if (ctx->ns_params_set && (p->physical_type != ctx->ns.physical_type || p->linearization != ctx->ns.linearization))
    {
      TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
      // ... and the streaming copies of the old scheme are dropped: the next vmult re-creates its copy in the new
      // layout; the frozen copy of velocity_vmult falls back to its generic form until the next
      // adaflo_ns_fix_linearization_point (its kernel would be instantiated for the NEW scheme)
      TRY(ctx, ensure_lin_prec_generic(ctx), "state re-layout failed");
      ctx->lin_q2_valid    = false;
      ctx->lin_q2_deferred = false;      // (laid out by ensure_lin_generic above, if it was deferred)
      ctx->hox_lin_primary      = false; // (the generic copies are current now; the x-marching kernel re-creates its own)
      ctx->hox_lin_prec_primary = false;
      release(ctx->lin_q2_prec);
      ctx->lin_nodal_prec_valid = false;
      ctx->lin_serial++;
    }
  ctx->ns = NSDev{p->physical_type, p->linearization, p->beta, p->tau_grad_div, p->density,
                  p->viscosity, p->damping, p->density_diff, p->weight, p->weight_old,
                  p->weight_old_old, p->tau1, p->extrap_old, p->extrap_old_old};
  ctx->ns_params_set = true;
  return 0;
}

int adaflo_ns_set_linearization(adaflo_ctx *ctx, const double *lin, int src_on_device)
{
  CHECK_CTX(ctx);
  const size_t count = (size_t)ctx->n_cells * ctx->nq_u * NLIN;
  TRY(ctx, alloc(ctx, ctx->lin, count), ctx->last_error);
  double *staging = nullptr;
  const double *src = lin;
  if (!src_on_device)
    {
      HIP_TRY(ctx, hipMalloc(&staging, count * sizeof(double)));
      HIP_TRY(ctx, hipMemcpyAsync(staging, lin, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      src = staging;
    }
  TRY(ctx, launch_transpose_state(ctx, ctx->lin.p, src, ctx->n_cells, ctx->nq_u, NLIN, true),
      "state re-layout failed");
  ctx->lin_q2_valid      = false;
  ctx->lin_q2_deferred   = false;
  ctx->lin_generic_valid = true;
  ctx->lin_gen++;
  ctx->lin_serial++; // (a state that did not come from a nodal field the engine knows: nothing to recompute from)
  ctx->hox_lin_primary = false;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (staging)
    (void)hipFree(staging);
  return 0;
}

int adaflo_ns_get_linearization(adaflo_ctx *ctx, double *lin, int dst_on_device)
{
  CHECK_CTX(ctx);
  if (!has_lin(ctx))
    return fail(ctx, ADAFLO_ENOTINIT, "no linearization data stored");
  TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
  const size_t count = ctx->lin.count;
  double      *dst   = lin;
  double      *staging = nullptr;
  if (!dst_on_device)
    {
      HIP_TRY(ctx, hipMalloc(&staging, count * sizeof(double)));
      dst = staging;
    }
  TRY(ctx, launch_transpose_state(ctx, dst, ctx->lin.p, ctx->n_cells, ctx->nq_u, NLIN, false),
      "state re-layout failed");
  if (staging)
    {
      HIP_TRY(ctx, hipMemcpyAsync(lin, staging, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (staging)
    (void)hipFree(staging);
  return 0;
}

int adaflo_ns_set_coefficients(adaflo_ctx *ctx, const double *rho, const double *mu,
                               const double *damping, int src_on_device)
{
  CHECK_CTX(ctx);
  const size_t count = (size_t)ctx->n_cells * ctx->nq_u;
  // a state that exists only as the Q2/Q1 streaming copy WITHOUT coefficient pieces (left by the sweep-kernel residual, also
  // the variable-coefficient one) is independent of the coefficients: it stays what it is (every time step of a two-phase
  // run sets new coefficients before its first residual -- re-laying out the state of the step before would be 3 ms at
  // 128^3 for nothing); whoever needs another layout later converts then (ensure_lin_generic, q2_launch)
  const bool keep_q2 = (ctx->lin_q2.p && ctx->lin_q2_valid && !ctx->lin_q2_varco && !ctx->lin_generic_valid && !ctx->hox_lin_primary) ||
                       ctx->lin_q2_deferred; // (a deferred state is a function of the nodal field alone)
  if (!keep_q2)
    {
      TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
      ctx->lin_q2_valid    = false; // the streaming copies of the sweep kernels carry the coefficients
      ctx->hox_lin_primary = false; // (ensure_lin_generic above brought the generic copy up to date)
    }
  ctx->lin_gen++;
  ctx->coef_gen++;
  ctx->q1_poisson_src = nullptr;
  if (!rho && !mu && !damping)
    {
      release(ctx->rho);
      release(ctx->mu);
      release(ctx->damp);
      return 0;
    }
  if (!rho || !mu || !damping)
    return fail(ctx, ADAFLO_EINVAL, "density, viscosity and damping arrays must be given together "
                                    "(navier_stokes_matrix.cc:100-108)");
  // canonical [cell][q] == generic [cell][1][q]: plain copy
  const hipMemcpyKind kind = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  TRY(ctx, alloc(ctx, ctx->rho, count), ctx->last_error);
  TRY(ctx, alloc(ctx, ctx->mu, count), ctx->last_error);
  TRY(ctx, alloc(ctx, ctx->damp, count), ctx->last_error);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->rho.p, rho, count * sizeof(double), kind, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->mu.p, mu, count * sizeof(double), kind, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->damp.p, damping, count * sizeof(double), kind, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_ns_get_coefficients(adaflo_ctx *ctx, double *rho, double *mu, double *damping, int dst_on_device)
{
  CHECK_CTX(ctx);
  if (!ctx->rho.p || !ctx->mu.p || !ctx->damp.p)
    return fail(ctx, ADAFLO_ENOTINIT, "variable coefficients not set");
  const size_t        count = (size_t)ctx->n_cells * ctx->nq_u;
  const hipMemcpyKind kind  = dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  double             *dst[3] = {rho, mu, damping};
  const double       *src[3] = {ctx->rho.p, ctx->mu.p, ctx->damp.p};
  for (int i = 0; i < 3; ++i)
    if (dst[i])
      HIP_TRY(ctx, hipMemcpyAsync(dst[i], src[i], count * sizeof(double), kind, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_ns_fix_linearization_point(adaflo_ctx *ctx)
{
  CHECK_CTX(ctx);
  TRY(ctx, q2_materialize_state(ctx), "state layout failed"); // (lazy state of the Q2/Q1 residual: the frozen copies are made from the laid-out one)
  // with the sweep kernels only the streaming copy is frozen; the generic frozen copy is rebuilt
  // from it on demand (ensure_lin_prec_generic)
  const bool streaming_only = ctx->variant >= 1 && q2_supported(ctx) && needs_lin(ctx) && ctx->lin_q2.p &&
                              ctx->lin_q2_valid && !ctx->lin_q2_varco && !ctx->rho.p;
  if (ctx->variant == 1 && hox_supported(ctx) && needs_lin(ctx) && ctx->hox_lin_primary && !ctx->lin_generic_valid)
    {
      // the state exists in the streaming layout of the x-marching kernel only (its residual mode wrote it): freeze
      // that copy; the generic frozen copy is rebuilt from it on demand (ensure_lin_prec_generic)
      TRY(ctx, alloc(ctx, ctx->hox_lin_prec, ctx->hox_lin.count), ctx->last_error);
      HIP_TRY(ctx, hipMemcpyAsync(ctx->hox_lin_prec.p, ctx->hox_lin.p, ctx->hox_lin.count * sizeof(double),
                                  hipMemcpyDeviceToDevice, ctx->stream));
      ctx->lin_prec_gen++;
      ctx->hox_lin_prec_gen       = ctx->lin_prec_gen;
      ctx->hox_lin_prec_mode      = ctx->hox_lin_mode;
      ctx->hox_lin_prec_varco     = ctx->hox_lin_varco; // coefficient pieces only if the variable-coefficient residual wrote
                                                        // the copy (a flag left by an earlier variable-coefficient vmult
                                                        // would send prepare_state to the stale generic copy)
      ctx->hox_lin_prec_primary   = true;
      ctx->lin_prec_generic_valid = false;
      if (ctx->hox_lin_varco && ctx->rho.p)
        {
          // the frozen coefficients (generic arrays: what the generic kernels and the Q1 pressure operators read)
          const DeviceBuffer *csrc[3] = {&ctx->rho, &ctx->mu, &ctx->damp};
          DeviceBuffer       *cdst[3] = {&ctx->rho_prec, &ctx->mu_prec, &ctx->damp_prec};
          for (int i = 0; i < 3; ++i)
            {
              TRY(ctx, alloc(ctx, *cdst[i], csrc[i]->count), ctx->last_error);
              HIP_TRY(ctx, hipMemcpyAsync(cdst[i]->p, csrc[i]->p, csrc[i]->count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
      else
        release(ctx->rho_prec), release(ctx->mu_prec), release(ctx->damp_prec);
      release(ctx->lin_q2_prec);
      ctx->lin_nodal_prec_valid = false;
      ctx->q1_poisson_src = nullptr;
      return 0;
    }
  ctx->hox_lin_prec_primary    = false;
  const DeviceBuffer *src[4] = {&ctx->lin, &ctx->rho, &ctx->mu, &ctx->damp};
  DeviceBuffer       *dst[4] = {&ctx->lin_prec, &ctx->rho_prec, &ctx->mu_prec, &ctx->damp_prec};
  if (!streaming_only)
    TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
  ctx->lin_prec_generic_valid = !streaming_only;
  ctx->lin_prec_gen++;
  for (int i = streaming_only ? 1 : 0; i < 4; ++i)
    {
      TRY(ctx, alloc(ctx, *dst[i], src[i]->count), ctx->last_error);
      if (src[i]->count)
        HIP_TRY(ctx, hipMemcpyAsync(dst[i]->p, src[i]->p, src[i]->count * sizeof(double),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    }
  ctx->q1_poisson_src = nullptr; // the frozen density copy may have changed content
  // keep a frozen copy in the streaming layout of the Q2/Q1 kernel as well
  if (q2_supported(ctx) && has_lin(ctx) && needs_lin(ctx))
    {
      if (ctx->lin_q2_valid && ctx->lin_q2_varco != (ctx->rho.p != nullptr))
        ctx->lin_q2_valid = false; // (written without coefficient pieces by the variable-coefficient residual; the generic copy is current by now)
      TRY(ctx, q2_prepare_state(ctx), "state conversion failed");
      TRY(ctx, alloc(ctx, ctx->lin_q2_prec, ctx->lin_q2.count), ctx->last_error);
      HIP_TRY(ctx, hipMemcpyAsync(ctx->lin_q2_prec.p, ctx->lin_q2.p, ctx->lin_q2.count * sizeof(double),
                                  hipMemcpyDeviceToDevice, ctx->stream));
      ctx->lin_q2_prec_varco = ctx->lin_q2_varco;
      ctx->lin_q2_prec_mode  = ctx->lin_q2_mode;
      // ... and the nodal linearisation point of the recompute-state mode, if the state came from a residual
      ctx->lin_nodal_prec_valid = false;
      if (lin_nodal_current(ctx))
        {
          TRY(ctx, alloc(ctx, ctx->lin_nodal_prec, ctx->lin_nodal.count), ctx->last_error);
          HIP_TRY(ctx, hipMemcpyAsync(ctx->lin_nodal_prec.p, ctx->lin_nodal.p, ctx->lin_nodal.count * sizeof(double),
                                      hipMemcpyDeviceToDevice, ctx->stream));
          ctx->lin_nodal_prec_valid = true;
        }
    }
  else
    {
      release(ctx->lin_q2_prec);
      ctx->lin_nodal_prec_valid = false;
    }
  return 0;
}

int adaflo_ns_apply_pressure_average_projection(adaflo_ctx *ctx, double *vec_p)
{
  CHECK_CTX(ctx);
  // :196-198
  if (ctx->ns.linearization == ADAFLO_PROJECTION ||
      ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE_STATIONARY || !ctx->d_p_weights)
    return 0;
  TRY(ctx,
      launch_mean_projection(ctx, vec_p, ctx->d_p_weights, ctx->d_p_modes, ctx->n_nodes_p,
                             ctx->inv_p_weight),
      "projection launch failed");
  return 0;
}

int adaflo_ns_vmult(adaflo_ctx *ctx, double *dst_u, double *dst_p, const double *src_u,
                    const double *src_p)
{
  CHECK_CTX(ctx);
  if (!dst_u || !dst_p || !src_u || !src_p)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (needs_lin(ctx) && !has_lin(ctx))
    return fail(ctx, ADAFLO_ENOTINIT, "linearization data not set (call residual or set_linearization)");
  ScopedTimer timer(ctx, ctx->matvec_timer);
  const int   k = ctx->k;
  if (ctx->variant >= 1 && q2_supported(ctx))
    {
      TRY(ctx, launch_ns_vmult_q2(ctx, OP_VMULT, dst_u, dst_p, src_u, src_p), "Q2 kernel launch failed");
    }
  else if (ctx->variant == 1 && ctx->hop && hop_supported(ctx, OP_VMULT)) // ns_hop.hip: plane-per-lane kernel (round 5)
    {
      TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
      TRY(ctx, launch_ns_vmult_hop(ctx, OP_VMULT, dst_u, dst_p, src_u, src_p), "plane-per-lane kernel launch failed");
    }
  else if (ctx->variant == 1 && hox_supported(ctx)) // ns_hox.hip: x-marching kernel (round 4)
    {
      TRY(ctx, launch_ns_vmult_hox(ctx, OP_VMULT, dst_u, dst_p, src_u, src_p), "x-marching kernel launch failed");
    }
  else if (ctx->variant == 2 && ho_supported(ctx)) // ns_ho.hip: z-sweep kernel of round 2, kept for comparison
    {
      TRY(ctx, launch_ns_vmult_ho(ctx, OP_VMULT, dst_u, dst_p, src_u, src_p), "sweep kernel launch failed");
    }
  else
    {
      // dst = 0 (:229) fused with the constrained rows (:247-256)
      TRY(ctx,
          launch_prepare_dst(ctx, dst_u, src_u, ctx->n_nodes_u, 3, nn(ctx, k, 0), nn(ctx, k, 1),
                             nn(ctx, k, 2), ctx->brick.con_u, 1., true),
          "prepare failed");
      TRY(ctx,
          launch_prepare_dst(ctx, dst_p, src_p, ctx->n_nodes_p, 1, nn(ctx, k - 1, 0), nn(ctx, k - 1, 1),
                             nn(ctx, k - 1, 2), ctx->brick.con_p, -1., true),
          "prepare failed");
      TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
      NSArgs a = make_ns_args(ctx, false);
      a.src_u  = src_u;
      a.src_p  = src_p;
      a.dst_u  = dst_u;
      a.dst_p  = dst_p;
      TRY(ctx, launch_ns_cell_generic(ctx, OP_VMULT, a), "cell kernel launch failed");
    }
  return adaflo_ns_apply_pressure_average_projection(ctx, dst_p); // :258
}

int adaflo_ns_supports_phases(adaflo_ctx *ctx)
{
  if (!ctx)
    return 0;
  if (ctx->variant >= 1 && q2_supported(ctx))
    return 1;
  return (ctx->variant == 1 && hox_supported(ctx)) || (ctx->variant == 2 && ho_supported(ctx)) ? 1 : 0;
}

int adaflo_ns_vmult_phase(adaflo_ctx *ctx, double *dst_u, double *dst_p, const double *src_u,
                          const double *src_p, int phase, unsigned interface_faces)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (!dst_u || !dst_p || !src_u || !src_p)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (phase < 0 || phase > 5)
    return fail(ctx, ADAFLO_EINVAL, "phase must be 0 ... 5");
  if (!adaflo_ns_supports_phases(ctx))
    return fail(ctx, ADAFLO_EUNSUPPORTED, "phased vmult needs one of the sweep kernels (Q2/Q1, or Q3..Q5 with constant coefficients)");
  if (needs_lin(ctx) && !has_lin(ctx))
    return fail(ctx, ADAFLO_ENOTINIT, "linearization data not set (call residual or set_linearization)");
  if (ctx->variant >= 1 && q2_supported(ctx))
    {
      TRY(ctx, launch_ns_vmult_q2(ctx, OP_VMULT, dst_u, dst_p, src_u, src_p, phase, interface_faces),
          "Q2 kernel launch failed");
    }
  else if (ctx->variant == 1 && ctx->hop && hop_supported(ctx, OP_VMULT))
    {
      TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
      TRY(ctx, launch_ns_vmult_hop(ctx, OP_VMULT, dst_u, dst_p, src_u, src_p, phase, interface_faces),
          "plane-per-lane kernel launch failed");
    }
  else if (ctx->variant == 1)
    {
      TRY(ctx, launch_ns_vmult_hox(ctx, OP_VMULT, dst_u, dst_p, src_u, src_p, phase, interface_faces),
          "x-marching kernel launch failed");
    }
  else
    {
      TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
      TRY(ctx, launch_ns_vmult_ho(ctx, OP_VMULT, dst_u, dst_p, src_u, src_p, phase, interface_faces),
          "high-order kernel launch failed");
    }
  if (phase == 1)
    ctx->matvec_timer.count++;
  return 0;
}

int adaflo_ns_residual(adaflo_ctx *ctx, double *rhs_u, double *rhs_p, const double *src_u,
                       const double *src_p, const double *user_u, const double *user_p,
                       const double *old_u, const double *old_old_u)
{
  CHECK_CTX(ctx);
  if (!rhs_u || !rhs_p || !src_u || !src_p)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE && (!old_u || !old_old_u))
    return fail(ctx, ADAFLO_EINVAL, "solution_old / solution_old_old required");
  ctx->lin_serial++; // (whatever path: the state is that of src from here on)
  ctx->lin_q2_deferred = false;
  if (ctx->variant >= 1 && q2_residual_supported(ctx))
    {
      // sweep kernel in residual mode: cell-loop sums into work vectors, then
      // rhs = user - rhs - sum (the reference's cell loop adds into system_rhs, :279-292)
      const long nu = 3 * ctx->n_nodes_u, np = ctx->n_nodes_p;
      TRY(ctx, alloc(ctx, ctx->res_sum_u, nu), ctx->last_error);
      TRY(ctx, alloc(ctx, ctx->res_sum_p, np), ctx->last_error);
      const double *old_comb = nullptr;
      if (ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE)
        {
          TRY(ctx, alloc(ctx, ctx->res_old, nu), ctx->last_error);
          TRY(ctx, launch_lincomb(ctx, ctx->res_old.p, ctx->ns.weight_old, old_u, ctx->ns.weight_old_old, old_old_u, nu),
              "old-solution combination failed");
          old_comb = ctx->res_old.p;
        }
      const double *ext_comb = nullptr;
      if (ctx->ns.linearization == ADAFLO_COUPLED_VELOCITY_SEMI_IMPLICIT || ctx->ns.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT ||
          ctx->ns.linearization == ADAFLO_PROJECTION)
        {
          // :644-647, 740-782: the extrapolated velocity extrap_old u_old + extrap_old_old u_old_old, combined at the nodes
          TRY(ctx, alloc(ctx, ctx->res_ext, nu), ctx->last_error);
          TRY(ctx, launch_lincomb(ctx, ctx->res_ext.p, ctx->ns.extrap_old, old_u, ctx->ns.extrap_old_old, old_old_u, nu),
              "extrapolation failed");
          ext_comb = ctx->res_ext.p;
        }
      TRY(ctx, launch_ns_residual_q2(ctx, ctx->res_sum_u.p, ctx->res_sum_p.p, src_u, src_p, old_comb, ext_comb),
          "Q2 residual kernel launch failed");
      if (needs_lin(ctx))
        ctx->lin_generic_valid = false;
      TRY(ctx, launch_residual_finish(ctx, rhs_u, ctx->res_sum_u.p, user_u, nu), "residual update failed");
      TRY(ctx, launch_residual_finish(ctx, rhs_p, ctx->res_sum_p.p, user_p, np), "residual update failed");
      return 0;
    }
  if (ctx->variant == 1 && hox_residual_supported(ctx))
    {
      // x-marching kernel in residual mode (k = 3, 4, 5): as above, the state goes out in its streaming layout
      const long nu = 3 * ctx->n_nodes_u, np = ctx->n_nodes_p;
      TRY(ctx, alloc(ctx, ctx->res_sum_u, nu), ctx->last_error);
      TRY(ctx, alloc(ctx, ctx->res_sum_p, np), ctx->last_error);
      const double *old_comb = nullptr;
      if (ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE)
        {
          TRY(ctx, alloc(ctx, ctx->res_old, nu), ctx->last_error);
          TRY(ctx, launch_lincomb(ctx, ctx->res_old.p, ctx->ns.weight_old, old_u, ctx->ns.weight_old_old, old_old_u, nu),
              "old-solution combination failed");
          old_comb = ctx->res_old.p;
        }
      const double *ext_comb = nullptr;
      if (ctx->ns.linearization == ADAFLO_COUPLED_VELOCITY_SEMI_IMPLICIT || ctx->ns.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT ||
          ctx->ns.linearization == ADAFLO_PROJECTION)
        {
          // :644-647, 740-782: the extrapolated velocity extrap_old u_old + extrap_old_old u_old_old, combined at the nodes
          TRY(ctx, alloc(ctx, ctx->res_ext, nu), ctx->last_error);
          TRY(ctx, launch_lincomb(ctx, ctx->res_ext.p, ctx->ns.extrap_old, old_u, ctx->ns.extrap_old_old, old_old_u, nu),
              "extrapolation failed");
          ext_comb = ctx->res_ext.p;
        }
      TRY(ctx, launch_ns_residual_hox(ctx, ctx->res_sum_u.p, ctx->res_sum_p.p, src_u, src_p, old_comb, ext_comb),
          "x-marching residual kernel launch failed");
      TRY(ctx, launch_residual_finish(ctx, rhs_u, ctx->res_sum_u.p, user_u, nu), "residual update failed");
      TRY(ctx, launch_residual_finish(ctx, rhs_p, ctx->res_sum_p.p, user_p, np), "residual update failed");
      return 0;
    }
  if (needs_lin(ctx))
    {
      TRY(ctx, alloc(ctx, ctx->lin, (size_t)ctx->n_cells * ctx->nq_u * NLIN), ctx->last_error);
      ctx->lin_generic_valid = true;
      ctx->lin_gen++;
      ctx->hox_lin_primary = false;
    }
  NSArgs a   = make_ns_args(ctx, false);
  a.src_u    = src_u;
  a.src_p    = src_p;
  a.dst_u    = rhs_u;
  a.dst_p    = rhs_p;
  a.old_u    = old_u;
  a.oldold_u = old_old_u;
  TRY(ctx, launch_ns_cell_generic(ctx, OP_RESIDUAL, a), "cell kernel launch failed");
  ctx->lin_q2_valid = false;
  // (two-phase flow: the Q2/Q1 vmult recomputes the state from this copy -- where the state is that of the solution itself)
  if (ctx->variant >= 1 && q2_supported(ctx) &&
      (ctx->ns.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON || ctx->ns.linearization == ADAFLO_COUPLED_IMPLICIT_PICARD))
    TRY(ctx, q2_capture_nodal(ctx, src_u), "nodal copy failed");
  // system_rhs.sadd(-1., 1., user_rhs)  :292
  TRY(ctx, launch_sadd(ctx, rhs_u, -1., user_u, 3 * ctx->n_nodes_u), "sadd failed");
  TRY(ctx, launch_sadd(ctx, rhs_p, -1., user_p, ctx->n_nodes_p), "sadd failed");
  return 0;
}

int adaflo_ns_velocity_vmult(adaflo_ctx *ctx, double *dst_u, const double *src_u)
{
  CHECK_CTX(ctx);
  if (!dst_u || !src_u)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (needs_lin(ctx) && !has_lin(ctx))
    return fail(ctx, ADAFLO_ENOTINIT, "linearization data not set");
  const int k = ctx->k;
  // (frozen coefficient copies without a frozen streaming copy: generic kernel; likewise a frozen state that exists in
  // the generic layout only -- after a change of scheme, adaflo_ns_set_params)
  const bool frozen_generic_only = ctx->lin_prec.p && ctx->lin_prec_generic_valid && !ctx->lin_q2_prec.p;
  // (explicit scheme: no state -- the sweep kernel reads the frozen coefficients from the generic arrays)
  const bool explicit_no_state = !needs_lin(ctx) && ctx->ns.physical_type != ADAFLO_STOKES;
  if (ctx->variant >= 1 && q2_supported(ctx) && !frozen_generic_only && (!ctx->rho_prec.p || ctx->lin_q2_prec.p || explicit_no_state))
    {
      TRY(ctx, launch_ns_vmult_q2(ctx, OP_VMULT_VELOCITY, dst_u, nullptr, src_u, nullptr),
          "Q2 kernel launch failed");
      return 0;
    }
  if (ctx->variant == 1 && ctx->hop && hop_supported(ctx, OP_VMULT_VELOCITY))
    {
      TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
      TRY(ctx, ensure_lin_prec_generic(ctx), "state re-layout failed");
      TRY(ctx, launch_ns_vmult_hop(ctx, OP_VMULT_VELOCITY, dst_u, nullptr, src_u, nullptr),
          "plane-per-lane kernel launch failed");
      return 0;
    }
  if (ctx->variant == 1 && hox_supported(ctx)) // (its frozen streaming copy carries the frozen coefficients)
    {
      TRY(ctx, launch_ns_vmult_hox(ctx, OP_VMULT_VELOCITY, dst_u, nullptr, src_u, nullptr),
          "x-marching kernel launch failed");
      return 0;
    }
  if (ctx->variant == 2 && ho_supported(ctx) && !ctx->rho_prec.p)
    {
      TRY(ctx, launch_ns_vmult_ho(ctx, OP_VMULT_VELOCITY, dst_u, nullptr, src_u, nullptr),
          "sweep kernel launch failed");
      return 0;
    }
  TRY(ctx,
      launch_prepare_dst(ctx, dst_u, src_u, ctx->n_nodes_u, 3, nn(ctx, k, 0), nn(ctx, k, 1),
                         nn(ctx, k, 2), ctx->brick.con_u, 1., true),
      "prepare failed");
  // :349-356: operate on the state frozen by fix_linearization_point, if any
  TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
  TRY(ctx, ensure_lin_prec_generic(ctx), "state re-layout failed");
  NSArgs a = make_ns_args(ctx, true);
  a.src_u  = src_u;
  a.dst_u  = dst_u;
  TRY(ctx, launch_ns_cell_generic(ctx, OP_VMULT_VELOCITY, a), "cell kernel launch failed");
  return 0;
}

int adaflo_ns_velocity_block_diagonal(adaflo_ctx *ctx, double *diagonal_u)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (!diagonal_u)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (needs_lin(ctx) && !has_lin(ctx))
    return fail(ctx, ADAFLO_ENOTINIT, "linearization data not set");
  // the same (possibly frozen) state velocity_vmult operates on, in the generic layout
  TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
  TRY(ctx, ensure_lin_prec_generic(ctx), "state re-layout failed");
  const int k = ctx->k;
  TRY(ctx, launch_fill(ctx, diagonal_u, 1., 3 * ctx->n_nodes_u) ? ADAFLO_EHIP : 0, "fill failed");
  // zero on the free rows, 1 (= the operator applied to a unit vector) on the constrained ones
  TRY(ctx,
      launch_prepare_dst(ctx, diagonal_u, diagonal_u, ctx->n_nodes_u, 3, nn(ctx, k, 0), nn(ctx, k, 1), nn(ctx, k, 2),
                         ctx->brick.con_u, 1., true),
      "prepare failed");
  const NSArgs a = make_ns_args(ctx, true);
  TRY(ctx, launch_ns_velocity_diagonal(ctx, a, diagonal_u), "diagonal kernel launch failed");
  return 0;
}

static int scalar_op(adaflo_ctx *ctx, double *dst, const double *src, const int mode,
                     const double *coef, const bool quad_u, const bool zero_dst)
{
  const int k = ctx->k;
  if (zero_dst)
    TRY(ctx,
        launch_prepare_dst(ctx, dst, src, ctx->n_nodes_p, 1, nn(ctx, k - 1, 0), nn(ctx, k - 1, 1),
                           nn(ctx, k - 1, 2), ctx->brick.con_p, 1., true),
        "prepare failed");
  ScalarArgs sa{};
  sa.brick   = ctx->brick;
  sa.ns      = ctx->ns;
  sa.src     = src;
  sa.dst     = dst;
  sa.coef_q  = coef;
  sa.tab     = quad_u ? ctx->d_tab_u : ctx->d_tab_pp;
  sa.n_cells = ctx->n_cells;
  sa.mode    = mode;
  sa.nq_u3   = ctx->nq_u;
  TRY(ctx, launch_ns_scalar_generic(ctx, sa), "scalar kernel launch failed");
  return 0;
}

int adaflo_ns_divergence_vmult_add(adaflo_ctx *ctx, double *dst_p, const double *src_u,
                                   int weight_by_viscosity)
{
  CHECK_CTX(ctx);
  if (!dst_p || !src_u)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  // Q2/Q1, constant viscosity: tensor-product stencil (csrc/ns_divergence.hip); constraints of src
  // resolved except for the projection scheme (:935-939)
  if (ctx->variant == 1 && divergence_stencil_supported(ctx) && !(weight_by_viscosity && ctx->mu.p))
    {
      TRY(ctx,
          launch_ns_divergence_stencil(ctx, dst_p, src_u, weight_by_viscosity ? -ctx->ns.viscosity : -1.,
                                       ctx->ns.linearization == ADAFLO_PROJECTION),
          "divergence kernel launch failed");
      return 0;
    }
  // variant 2: divergence mode of the sweep kernel, then dst += sums on the free rows
  if (ctx->variant >= 2 && ctx->k == 2 && !(weight_by_viscosity && ctx->mu.p) &&
      ctx->ns.linearization != ADAFLO_PROJECTION)
    {
      TRY(ctx, alloc(ctx, ctx->res_sum_p, ctx->n_nodes_p), ctx->last_error);
      TRY(ctx,
          launch_ns_divergence_q2(ctx, ctx->res_sum_p.p, src_u, dst_p, weight_by_viscosity ? -ctx->ns.viscosity : -1.),
          "Q2 divergence kernel launch failed");
      TRY(ctx,
          launch_add_unconstrained(ctx, dst_p, ctx->res_sum_p.p, ctx->n_nodes_p, 1, nn(ctx, 1, 0), nn(ctx, 1, 1),
                                   nn(ctx, 1, 2), ctx->brick.con_p),
          "add failed");
      return 0;
    }
  return scalar_op(ctx, dst_p, src_u, weight_by_viscosity ? SC_DIVERGENCE_VISC : SC_DIVERGENCE,
                   ctx->mu.p, true, false);
}

int adaflo_ns_pressure_poisson_vmult(adaflo_ctx *ctx, double *dst_p, const double *src_p)
{
  CHECK_CTX(ctx);
  if (!dst_p || !src_p)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  // :395-396 preconditioner copy of the densities if fixed
  const double *rho = ctx->rho_prec.p ? ctx->rho_prec.p : ctx->rho.p;
  const bool    var = rho && ctx->ns.linearization != ADAFLO_PROJECTION; // :976-978
  const bool    full = var && ctx->ns.physical_type != ADAFLO_INCOMPRESSIBLE_STATIONARY;
  if (ctx->variant >= 1 && ctx->k == 2 && !var)
    {
      // constant coefficient on Q1: structured sweep kernel (:1002-1031)
      const NSDev &P = ctx->ns;
      const double c = P.physical_type == ADAFLO_INCOMPRESSIBLE_STATIONARY ?
                         1. :
                         1. / (P.weight * std::min(P.density, P.density + P.density_diff));
      TRY(ctx,
          launch_q1_sweep(ctx, 1, Q1_MASS_LAPLACE, 0., c, 0., ctx->brick.con_p, 1., nullptr, dst_p, src_p, nullptr),
          "pressure kernel launch failed");
      return 0;
    }
  if (ctx->variant >= 1 && ctx->k == 2 && full)
    {
      // variable density at the 27 points of quad_index_u (:984-1000): structured sweep kernel on the
      // coefficient 1 / (weight rho) re-laid out once per density field / time-step weight
      if (ctx->q1_poisson_src != rho || ctx->q1_poisson_weight != ctx->ns.weight || !ctx->q1_poisson_coef.p)
        {
          TRY(ctx, q1_convert_poisson_coef(ctx, ctx->q1_poisson_coef, rho, ctx->ns.weight), "coefficient re-layout failed");
          ctx->q1_poisson_src    = rho;
          ctx->q1_poisson_weight = ctx->ns.weight;
        }
      TRY(ctx,
          launch_q1_sweep(ctx, 1, Q1_LAPLACE_Q3, 0., 0., 0., ctx->brick.con_p, 1., nullptr, dst_p, src_p,
                          ctx->q1_poisson_coef.p),
          "pressure kernel launch failed");
      return 0;
    }
  return scalar_op(ctx, dst_p, src_p, full ? SC_POISSON_VARIABLE : SC_POISSON_CELL, var ? rho : nullptr,
                   full, true);
}

int adaflo_ns_pressure_mass_weight_add(adaflo_ctx *ctx, double *dst_p)
{
  CHECK_CTX(ctx);
  if (!dst_p)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  return scalar_op(ctx, dst_p, nullptr, SC_MASS_WEIGHT, nullptr, false, false);
}

int adaflo_ns_apply_constrained_rows(adaflo_ctx *ctx, double *dst_u, double *dst_p,
                                     const double *src_u, const double *src_p)
{
  CHECK_CTX(ctx);
  const int k = ctx->k;
  if (dst_u && ctx->brick.con_u)
    TRY(ctx,
        launch_constrained_faces(ctx, dst_u, src_u, 3, nn(ctx, k, 0), nn(ctx, k, 1), nn(ctx, k, 2), ctx->brick.con_u, 1.),
        "constrained rows failed");
  if (dst_p && ctx->brick.con_p)
    TRY(ctx,
        launch_constrained_faces(ctx, dst_p, src_p, 1, nn(ctx, k - 1, 0), nn(ctx, k - 1, 1), nn(ctx, k - 1, 2),
                                 ctx->brick.con_p, -1.),
        "constrained rows failed");
  return 0;
}

int adaflo_ns_pressure_mass_vmult(adaflo_ctx *ctx, double *dst_p, const double *src_p)
{
  CHECK_CTX(ctx);
  if (!dst_p || !src_p)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  const double *mu = ctx->mu_prec.p ? ctx->mu_prec.p : ctx->mu.p;
  if (ctx->variant >= 1 && ctx->k == 2 && !mu)
    {
      // constant coefficient on Q1: structured sweep kernel (:1036-1071)
      const NSDev &P = ctx->ns;
      const double c = (P.linearization == ADAFLO_PROJECTION || P.physical_type == ADAFLO_INCOMPRESSIBLE_STATIONARY) ?
                         1. :
                         1. / (P.viscosity + P.tau_grad_div);
      TRY(ctx,
          launch_q1_sweep(ctx, 1, Q1_MASS_LAPLACE, c, 0., 0., ctx->brick.con_p, 1., nullptr, dst_p, src_p, nullptr),
          "pressure kernel launch failed");
      return 0;
    }
  if (ctx->variant >= 1 && ctx->k == 2 && mu &&
      !(ctx->ns.linearization == ADAFLO_PROJECTION || ctx->ns.physical_type == ADAFLO_INCOMPRESSIBLE_STATIONARY))
    {
      // per-cell viscosity sample (:1057-1066): c = 1 / (mu(mid point) + tau_grad_div)
      TRY(ctx,
          launch_q1_sweep(ctx, 1, Q1_MASS_LAPLACE, 1., 0., 0., ctx->brick.con_p, 1., nullptr, dst_p, src_p, nullptr, 1,
                          mu, ctx->nq_u, ctx->nq_u / 2, ctx->ns.tau_grad_div),
          "pressure kernel launch failed");
      return 0;
    }
  return scalar_op(ctx, dst_p, src_p, SC_MASS, mu, false, true);
}

int adaflo_ns_pressure_convdiff_vmult(adaflo_ctx *ctx, double *dst_p, const double *src_p)
{
  CHECK_CTX(ctx);
  if (!dst_p || !src_p)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (!has_lin(ctx)) // Assert(linearized_velocities.size() > 0) :1110
    return fail(ctx, ADAFLO_ENOTINIT, "linearization data not set");
  return scalar_op(ctx, dst_p, src_p, SC_CONVDIFF, ctx->mu.p, true, true);
}

static int read_timer(adaflo_ctx *ctx, EventTimer &t, unsigned *count, double *seconds)
{
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  t.fold();
  if (count)
    *count = t.count;
  if (seconds)
    *seconds = t.seconds;
  t.count   = 0;
  t.seconds = 0.;
  return 0;
}

int adaflo_ns_get_matvec_statistics(adaflo_ctx *ctx, unsigned *count, double *seconds)
{
  CHECK_CTX(ctx);
  return read_timer(ctx, ctx->matvec_timer, count, seconds);
}

int adaflo_get_kernel_statistics(adaflo_ctx *ctx, unsigned *count, double *seconds)
{
  CHECK_CTX(ctx);
  return read_timer(ctx, ctx->kernel_timer, count, seconds);
}

int adaflo_set_timing(adaflo_ctx *ctx, int enabled)
{
  CHECK_CTX(ctx);
  ctx->timing = enabled != 0;
  return 0;
}

int adaflo_fdm_apply(adaflo_ctx *ctx, int field, double *dst, const double *src, double c_mass, double c_lap)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (ctx->flat)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "the fast-diagonalisation inverses are dim = 3 only");
  if (!dst || !src || field < 0 || field > 2)
    return fail(ctx, ADAFLO_EINVAL, "invalid arguments");
  if (field == 2 && ctx->s <= 0)
    return fail(ctx, ADAFLO_ENOTINIT, "context without a level-set space");
  TRY(ctx, fdm_apply(ctx, field, dst, src, c_mass, c_lap), "fast-diagonalisation solve failed");
  return 0;
}

int adaflo_fdm_apply_sum(adaflo_ctx *ctx, int field, double *dst, const double *src, double c_mass, double c_lap,
                         double c_mass2, double c_lap2)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (ctx->flat)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "the fast-diagonalisation inverses are dim = 3 only");
  if (!dst || !src || field < 0 || field > 2 || (c_mass2 == 0. && c_lap2 == 0.))
    return fail(ctx, ADAFLO_EINVAL, "invalid arguments");
  if (field == 2 && ctx->s <= 0)
    return fail(ctx, ADAFLO_ENOTINIT, "context without a level-set space");
  const uint32_t mask = field == 0 ? ctx->brick.con_u : (field == 1 ? ctx->brick.con_p : ctx->brick.con_ls);
  if (mask != 0u)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "constrained rows: apply the two inverses separately");
  TRY(ctx, fdm_apply(ctx, field, dst, src, c_mass, c_lap, c_mass2, c_lap2), "fast-diagonalisation solve failed");
  return 0;
}

int adaflo_ns_set_iterations_before_inner_solvers(adaflo_ctx *ctx, int iterations)
{
  CHECK_CTX(ctx);
  if (iterations < 0)
    return fail(ctx, ADAFLO_EINVAL, "negative iteration count");
  ctx->pc_its_before_inner = iterations;
  return 0;
}

int adaflo_ns_preconditioner_set_cheap_velocity_iterations(adaflo_ctx *ctx, int iterations)
{
  CHECK_CTX(ctx);
  if (iterations < 0)
    return fail(ctx, ADAFLO_EINVAL, "negative iteration count");
  ctx->pc_simple_velocity_its = iterations;
  return 0;
}

int adaflo_ns_preconditioner_set_inner(adaflo_ctx *ctx, int mode)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (mode < 0 || mode > 1)
    return fail(ctx, ADAFLO_EINVAL, "unknown inner-solve mode");
  if (ctx->flat && mode == 1)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "the fast-diagonalisation inverses are dim = 3 only");
  if (mode != ctx->pc_inner)
    ctx->pc_ready = false; // the other flavour of inner solves needs its own set-up data
  ctx->pc_inner = mode;
  return 0;
}

int adaflo_set_q2_state_pad(adaflo_ctx *ctx, int pad_16B)
{
  CHECK_CTX(ctx);
  if (pad_16B < 0)
    return fail(ctx, ADAFLO_EINVAL, "negative padding");
  TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
  ctx->q2_state_pad = pad_16B;
  ctx->lin_q2_valid = false;
  return 0;
}

int adaflo_set_q2_lazy_state(adaflo_ctx *ctx, int lazy)
{
  CHECK_CTX(ctx);
  ctx->q2_lazy_state = lazy != 0; // (a state that is deferred right now stays deferred until somebody asks for it)
  return 0;
}

int adaflo_set_q2_chunk(adaflo_ctx *ctx, int layers)
{
  CHECK_CTX(ctx);
  if (layers < 0)
    return fail(ctx, ADAFLO_EINVAL, "negative chunk length");
  ctx->q2_lz = layers;
  return 0;
}

int adaflo_set_hox_chunk(adaflo_ctx *ctx, int cells)
{
  CHECK_CTX(ctx);
  if (cells < 0)
    return fail(ctx, ADAFLO_EINVAL, "negative chunk length");
  ctx->hox_lx = cells;
  return 0;
}


/* ------------------------------------------------------------------------- */
/* inter-GPU exchange helpers                                                 */
/* ------------------------------------------------------------------------- */
static int halo_transfer_impl(adaflo_ctx *ctx, double *vec, double *buf, const int *nn, int ncomp,
                         int n_regions, const int *regions, int mode, int self_pos)
{
  CHECK_CTX(ctx);
  if (!vec || !buf || !nn || !regions || n_regions < 0 || n_regions > 26 || mode < 0 || mode > 2)
    return fail(ctx, ADAFLO_EINVAL, "invalid halo transfer arguments");
  HaloPlan plan{};
  plan.n_regions = n_regions;
  plan.ncomp     = ncomp;
  for (int d = 0; d < 3; ++d)
    plan.nn[d] = nn[d];
  plan.offset[0] = 0;
  plan.self_pos  = self_pos;
  for (int r = 0; r < n_regions; ++r)
    {
      long n = ncomp;
      for (int d = 0; d < 3; ++d)
        {
          plan.lo[r][d] = regions[6 * r + 2 * d];
          plan.hi[r][d] = regions[6 * r + 2 * d + 1];
          if (plan.lo[r][d] < 0 || plan.hi[r][d] > nn[d] || plan.hi[r][d] <= plan.lo[r][d])
            return fail(ctx, ADAFLO_EINVAL, "halo region out of range");
          n *= plan.hi[r][d] - plan.lo[r][d];
        }
      plan.start[r]      = plan.offset[r];
      plan.offset[r + 1] = plan.offset[r] + n;
    }
  TRY(ctx, launch_halo(ctx, vec, buf, plan, mode), "halo kernel launch failed");
  return 0;
}

int adaflo_halo_transfer(adaflo_ctx *ctx, double *vec, double *buf, const int *nn, int ncomp, int n_regions,
                         const int *regions, int mode)
{
  if (ctx && ctx->indexed)
    return ADAFLO_EUNSUPPORTED; // (indexed context: needs the structured brick)
  return halo_transfer_impl(ctx, vec, buf, nn, ncomp, n_regions, regions, mode, 0);
}

int adaflo_halo_transfer_ordered(adaflo_ctx *ctx, double *vec, double *buf, const int *nn, int ncomp, int n_regions,
                                 const int *regions, int mode, int self_pos)
{
  if (ctx && ctx->indexed)
    return ADAFLO_EUNSUPPORTED; // (indexed context: needs the structured brick)
  return halo_transfer_impl(ctx, vec, buf, nn, ncomp, n_regions, regions, mode, self_pos);
}

/* ------------------------------------------------------------------------- */
/* level-set operators                                                        */
/* ------------------------------------------------------------------------- */
namespace
{
  int ls_ready(adaflo_ctx *ctx)
  {
    if (ctx->s <= 0 || !ctx->d_tab_ls)
      return fail(ctx, ADAFLO_ENOTINIT, "context was created without level-set spaces (ls_degree = 0)");
    return 0;
  }

  size_t ls_q_count(const adaflo_ctx *ctx)
  {
    return (size_t)ctx->n_cells * 3 * adaflo_n_q_points_ls(ctx);
  }

  int set_q_array(adaflo_ctx *ctx, DeviceBuffer &buf, const double *canonical, const int on_device)
  {
    const size_t count = ls_q_count(ctx);
    TRY(ctx, alloc(ctx, buf, count), ctx->last_error);
    double       *staging = nullptr;
    const double *src     = canonical;
    if (!on_device)
      {
        HIP_TRY(ctx, hipMalloc(&staging, count * sizeof(double)));
        HIP_TRY(ctx, hipMemcpyAsync(staging, canonical, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        src = staging;
      }
    TRY(ctx, launch_transpose_state(ctx, buf.p, src, ctx->n_cells, adaflo_n_q_points_ls(ctx), 3, true),
        "state re-layout failed");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (staging)
      (void)hipFree(staging);
    return 0;
  }

  int get_q_array(adaflo_ctx *ctx, const DeviceBuffer &buf, double *canonical, const int on_device)
  {
    if (!buf.p)
      return fail(ctx, ADAFLO_ENOTINIT, "quadrature-point array not set");
    const size_t count   = buf.count;
    double      *staging = nullptr, *dst = canonical;
    if (!on_device)
      {
        HIP_TRY(ctx, hipMalloc(&staging, count * sizeof(double)));
        dst = staging;
      }
    TRY(ctx, launch_transpose_state(ctx, dst, buf.p, ctx->n_cells, adaflo_n_q_points_ls(ctx), 3, false),
        "state re-layout failed");
    if (staging)
      HIP_TRY(ctx, hipMemcpyAsync(canonical, staging, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (staging)
      (void)hipFree(staging);
    return 0;
  }
} // namespace

// the generic [cell][3][q] copy of evaluated_convection (which = 0) / evaluated_normal (1), re-created
// from the sweep layout when the sweep right-hand side was the last writer
static int ls_generic_state(adaflo_ctx *ctx, const int which)
{
  DeviceBuffer &gen = which ? ctx->ls_normal : ctx->ls_convection, &swp = which ? ctx->q1_normal : ctx->q1_convection;
  bool         &gen_valid = which ? ctx->ls_normal_generic_valid : ctx->ls_convection_generic_valid;
  if (gen_valid && gen.p)
    return 0;
  if (which == 0 && !ctx->q1_convection_valid && ctx->q1_convection_nodal_valid)
    {
      // the sweep right-hand side kept the nodal velocity only: write evaluated_convection from it now
      TRY(ctx, q1_state_alloc(ctx, ctx->q1_convection), "out of device memory");
      TRY(ctx, launch_q1_rhs(ctx, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, ctx->q1_velocity_nodal.p, ctx->q1_convection.p, true),
          "level-set kernel launch failed");
      ctx->q1_convection_valid = true;
    }
  const bool    swp_valid = which ? ctx->q1_normal_valid : ctx->q1_convection_valid;
  if (!swp_valid || !swp.p)
    return fail(ctx, ADAFLO_ENOTINIT, "quadrature-point array not set");
  TRY(ctx, alloc(ctx, gen, ls_q_count(ctx)), ctx->last_error);
  TRY(ctx, q1_unconvert_state(ctx, gen.p, swp), "state re-layout failed");
  gen_valid = true;
  return 0;
}

int adaflo_ls_set_params(adaflo_ctx *ctx, const adaflo_ls_params *p)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (!p)
    return fail(ctx, ADAFLO_EINVAL, "null params");
  ctx->ls = LSDev{p->epsilon_used, p->minimal_edge_length, p->time_step, p->weight, p->weight_old,
                  p->weight_old_old, p->epsilon};
  return 0;
}

int adaflo_ls_set_diagonal(adaflo_ctx *ctx, const double *diag)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!ctx->d_ls_diag)
    HIP_TRY(ctx, hipMalloc(&ctx->d_ls_diag, ctx->n_nodes_ls * sizeof(double)));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ls_diag, diag, ctx->n_nodes_ls * sizeof(double),
                              hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

int adaflo_ls_set_evaluated_convection(adaflo_ctx *ctx, const double *u_q, int src_on_device)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  ctx->q1_convection_valid         = false;
  ctx->ls_convection_generic_valid = true;
  ctx->q1_convection_nodal_valid   = false;
  return set_q_array(ctx, ctx->ls_convection, u_q, src_on_device);
}

int adaflo_ls_get_evaluated_convection(adaflo_ctx *ctx, double *u_q, int dst_on_device)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_generic_state(ctx, 0))
    return e;
  return get_q_array(ctx, ctx->ls_convection, u_q, dst_on_device);
}

int adaflo_ls_set_evaluated_normal(adaflo_ctx *ctx, const double *n_q, int src_on_device)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  ctx->q1_normal_valid         = false;
  ctx->q1_normal_nodal_valid   = false;
  ctx->ls_normal_generic_valid = true;
  return set_q_array(ctx, ctx->ls_normal, n_q, src_on_device);
}

int adaflo_ls_get_evaluated_normal(adaflo_ctx *ctx, double *n_q, int dst_on_device)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_generic_state(ctx, 1))
    return e;
  return get_q_array(ctx, ctx->ls_normal, n_q, dst_on_device);
}

static int ls_vmult(adaflo_ctx *ctx, double *dst, const double *src, const int mode, const int flag,
                    double *qstate, const int nblocks)
{
  if (int e = ls_ready(ctx))
    return e;
  if (!dst || !src)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  const bool stabilised = mode == 0 && ctx->ls_stab; // convection stabilisation: generic kernels
  if (ctx->variant >= 1 && !stabilised)
    {
      // structured Q1 sweep kernel: FE_Q_iso_Q1(s) = trilinear elements on the refined grid
      if (ctx->brick.con_ls && !ctx->d_ls_diag)
        return fail(ctx, ADAFLO_ENOTINIT, "constrained rows need adaflo_ls_set_diagonal");
      const LSDev &P     = ctx->ls;
      const double hcell = ctx->flat ? std::max(ctx->desc.h[0], ctx->desc.h[1]) : std::max(ctx->desc.h[0], std::max(ctx->desc.h[1], ctx->desc.h[2]));
      // level_set_okz_reinitialization.cc:65-67,:82-85; compute_normal.cc:107-110;
      // compute_curvature.cc:112-118
      const double dtau_inv  = std::max(0.95 / (1. / (ctx->flat ? 4. : 9.) * P.minimal_edge_length / ctx->s), 1. / (5. * P.time_step));
      const double diffusion = std::max(P.epsilon_used, hcell / ctx->s);
      const double b         = std::max(P.epsilon_used / P.epsilon, hcell / ctx->s);
      int          q1mode = Q1_MASS_LAPLACE;
      double       c_mass = 1., c_lap = 0.;
      const double *state = nullptr;
      switch (mode)
        {
          case 0:
            q1mode = Q1_ADVECT;
            if (ctx->q1_convection_nodal_valid) // evaluated_convection came from a nodal velocity: evaluate it per Gauss point
              {
                q1mode = Q1_ADVECT_NODAL;
                state  = ctx->q1_velocity_nodal.p;
                break;
              }
            if (!ctx->q1_convection_valid)
              TRY(ctx, q1_convert_state(ctx, ctx->q1_convection, ctx->ls_convection.p), "state re-layout failed");
            ctx->q1_convection_valid = true;
            state                    = ctx->q1_convection.p;
            break;
          case 1:
            q1mode = Q1_REINIT;
            c_mass = dtau_inv;
            c_lap  = diffusion;
            if (ctx->q1_normal_nodal_valid) // evaluated_normal came from a nodal field: recompute it per Gauss point
              {
                q1mode = Q1_REINIT_NODAL;
                state  = ctx->q1_normal_nodal.p;
                break;
              }
            if (!ctx->q1_normal_valid)
              TRY(ctx, q1_convert_state(ctx, ctx->q1_normal, ctx->ls_normal.p), "state re-layout failed");
            ctx->q1_normal_valid = true;
            state                = ctx->q1_normal.p;
            break;
          case 2:
            c_mass = dtau_inv;
            c_lap  = diffusion;
            break;
          case 3:
            c_lap = 4. * b * b;
            break;
          default:
            c_lap = flag ? b * b : 0.;
        }
      TRY(ctx,
          launch_q1_sweep(ctx, ctx->s, q1mode, c_mass, c_lap, P.weight, ctx->brick.con_ls, 1., ctx->d_ls_diag,
                          dst, src, state, nblocks),
          "level-set kernel launch failed");
      return 0;
    }
  if (mode == 0 || mode == 1) // the generic kernels read the [cell][3][q] copy
    {
      if (int e = ls_generic_state(ctx, mode))
        return e;
      qstate = mode ? ctx->ls_normal.p : ctx->ls_convection.p;
    }
  HIP_TRY(ctx, hipMemsetAsync(dst, 0, sizeof(double) * nblocks * ctx->n_nodes_ls, ctx->stream)); // dst = 0.
  if (stabilised)
    {
      // :248-249 cell term with the artificial viscosities of the last rhs, :419-472 boundary term
      if (!ctx->ls_art_visc.p)
        return fail(ctx, ADAFLO_ENOTINIT, "artificial viscosities not set (run the stabilised rhs first)");
      LSStab st{};
      st.art_visc = ctx->ls_art_visc.p;
      st.bsign    = -1.;
      st.symmetry = ctx->ls_symmetry;
      TRY(ctx, launch_ls(ctx, 0, mode, flag, dst, src, nullptr, nullptr, nullptr, qstate, nblocks, &st),
          "level-set kernel launch failed");
      TRY(ctx, launch_ls(ctx, 3, 0, 0, dst, src, nullptr, nullptr, nullptr, nullptr, 1, &st),
          "level-set boundary kernel launch failed");
    }
  else
  TRY(ctx, launch_ls(ctx, 0, mode, flag, dst, src, nullptr, nullptr, nullptr, qstate, nblocks),
      "level-set kernel launch failed");
  TRY(ctx, launch_ls_constrained_rows(ctx, dst, src, nblocks), "constrained rows need adaflo_ls_set_diagonal");
  return 0;
}

int adaflo_ls_advance_concentration_vmult(adaflo_ctx *ctx, double *dst, const double *src)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (!ctx->ls_convection.p && !ctx->q1_convection_valid && !ctx->q1_convection_nodal_valid)
    return fail(ctx, ADAFLO_ENOTINIT, "evaluated_convection not set (run the rhs kernel first)");
  return ls_vmult(ctx, dst, src, 0 /*LS_ADVECT*/, 0, ctx->ls_convection.p, 1);
}

int adaflo_ls_advance_concentration_rhs(adaflo_ctx *ctx, double *dst, const double *solution,
                                        const double *solution_old, const double *solution_old_old,
                                        const double *vel_solution, int use_old_old)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!dst || !solution || !solution_old || !solution_old_old || !vel_solution)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (ctx->variant >= 1)
    {
      // sweep structure (csrc/q1_sweep.hip).  evaluated_convection is a function of the nodal velocity: the kernel does
      // not write it (192 B per sub-cell) when the operator can evaluate the velocity itself -- a copy of the velocity
      // vector is kept instead; otherwise it is written in sweep layout
      const bool nodal = q1_advect_nodal_supported(ctx) && getenv("ADAFLO_LS_STREAM_CONVECTION") == nullptr;
      if (nodal)
        {
          const size_t count = 3 * (size_t)ctx->n_nodes_u;
          if (ctx->q1_velocity_nodal.count != count)
            TRY(ctx, alloc(ctx, ctx->q1_velocity_nodal, count), ctx->last_error);
          HIP_TRY(ctx, hipMemcpyAsync(ctx->q1_velocity_nodal.p, vel_solution, count * sizeof(double), hipMemcpyDeviceToDevice,
                                      ctx->stream));
        }
      else
        TRY(ctx, q1_state_alloc(ctx, ctx->q1_convection), "out of device memory");
      const int e = launch_q1_rhs(ctx, 1, use_old_old ? 1 : 0, dst, solution, solution_old, solution_old_old, nullptr,
                                  vel_solution, nodal ? nullptr : ctx->q1_convection.p);
      if (e == 0)
        {
          ctx->q1_convection_valid         = !nodal;
          ctx->q1_convection_nodal_valid   = nodal;
          ctx->ls_convection_generic_valid = false;
          return 0;
        }
      if (e != ADAFLO_EUNSUPPORTED)
        return fail(ctx, e, "level-set kernel launch failed");
    }
  TRY(ctx, alloc(ctx, ctx->ls_convection, ls_q_count(ctx)), ctx->last_error);
  ctx->q1_convection_valid         = false;
  ctx->q1_convection_nodal_valid   = false;
  ctx->ls_convection_generic_valid = true;
  TRY(ctx,
      launch_ls(ctx, 2, 0, use_old_old, dst, solution, solution_old, solution_old_old, vel_solution,
                ctx->ls_convection.p, 1),
      "level-set kernel launch failed");
  return 0;
}

/* ---- convection stabilisation (parameters.convection_stabilization) -------------------- */
int adaflo_ls_set_convection_stabilization(adaflo_ctx *ctx, int enabled, double global_omega_diameter,
                                           unsigned symmetry_faces)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (enabled && !(global_omega_diameter > 0.))
    return fail(ctx, ADAFLO_EINVAL, "global_omega_diameter must be positive");
  ctx->ls_stab           = enabled != 0;
  ctx->ls_omega_diameter = global_omega_diameter;
  ctx->ls_symmetry       = symmetry_faces;
  if (enabled && !ctx->ls_art_visc.p)
    {
      TRY(ctx, alloc(ctx, ctx->ls_art_visc, (size_t)ctx->n_cells), ctx->last_error);
      TRY(ctx, launch_fill(ctx, ctx->ls_art_visc.p, 0., ctx->n_cells), "fill failed");
    }
  return 0;
}

int adaflo_ls_set_artificial_viscosities(adaflo_ctx *ctx, const double *nu, int src_on_device)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (!nu)
    return fail(ctx, ADAFLO_EINVAL, "null array");
  TRY(ctx, alloc(ctx, ctx->ls_art_visc, (size_t)ctx->n_cells), ctx->last_error);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ls_art_visc.p, nu, ctx->n_cells * sizeof(double),
                              src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_ls_get_artificial_viscosities(adaflo_ctx *ctx, double *nu, int dst_on_device)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (!nu || !ctx->ls_art_visc.p)
    return fail(ctx, ADAFLO_ENOTINIT, "artificial viscosities not set");
  HIP_TRY(ctx, hipMemcpyAsync(nu, ctx->ls_art_visc.p, ctx->n_cells * sizeof(double),
                              dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int adaflo_ls_max_velocity(adaflo_ctx *ctx, const double *vel_solution, double *max_velocity)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (!vel_solution || !max_velocity)
    return fail(ctx, ADAFLO_EINVAL, "null argument");
  const int k = ctx->k;
  if (!ctx->d_tab_maxvel)
    {
      Quadrature1D trap; // QIterated(QTrapezoid<1>(), k + 1): points j / (k + 1)
      for (int j = 0; j <= k + 1; ++j)
        {
          trap.x.push_back(double(j) / (k + 1));
          trap.w.push_back(0.);
        }
      TRY(ctx, upload(ctx, &ctx->d_tab_maxvel, shape_fe_q(k, trap).S), ctx->last_error);
    }
  unsigned long long *res = nullptr;
  HIP_TRY(ctx, hipMalloc(&res, sizeof(unsigned long long)));
  HIP_TRY(ctx, hipMemsetAsync(res, 0, sizeof(unsigned long long), ctx->stream));
  const int e = launch_ls_max_velocity(ctx, vel_solution, ctx->d_tab_maxvel, res);
  unsigned long long bits = 0;
  if (!e)
    (void)hipMemcpyAsync(&bits, res, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(res);
  if (e)
    return fail(ctx, e, "maximal-velocity kernel launch failed");
  std::memcpy(max_velocity, &bits, sizeof(double));
  return 0;
}

int adaflo_ls_stabilization_boundary_term(adaflo_ctx *ctx, double *dst, const double *vec, double sign)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!dst || !vec || !ctx->ls_art_visc.p)
    return fail(ctx, ADAFLO_ENOTINIT, "artificial viscosities not set");
  LSStab st{};
  st.art_visc = ctx->ls_art_visc.p;
  st.bsign    = sign;
  st.symmetry = ctx->ls_symmetry;
  TRY(ctx, launch_ls(ctx, 3, 0, 0, dst, vec, nullptr, nullptr, nullptr, nullptr, 1, &st),
      "level-set boundary kernel launch failed");
  return 0;
}

int adaflo_ls_advance_concentration_rhs_stabilized(adaflo_ctx *ctx, double *dst, const double *solution,
                                                   const double *solution_old, const double *solution_old_old,
                                                   const double *vel_solution, const double *vel_solution_old,
                                                   const double *vel_solution_old_old, int use_old_old,
                                                   double old_step_size, double global_max_velocity)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!ctx->ls_stab)
    return fail(ctx, ADAFLO_ENOTINIT, "call adaflo_ls_set_convection_stabilization first");
  if (!dst || !solution || !solution_old || !solution_old_old || !vel_solution || !vel_solution_old ||
      !vel_solution_old_old)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  TRY(ctx, alloc(ctx, ctx->ls_convection, ls_q_count(ctx)), ctx->last_error);
  TRY(ctx, alloc(ctx, ctx->ls_art_visc, (size_t)ctx->n_cells), ctx->last_error);
  TRY(ctx, alloc(ctx, ctx->ls_stab_vel_sum, (size_t)(3 * ctx->n_nodes_u)), ctx->last_error);
  TRY(ctx, alloc(ctx, ctx->ls_stab_ls_sum, (size_t)ctx->n_nodes_ls), ctx->last_error);
  ctx->q1_convection_valid         = false;
  ctx->q1_convection_nodal_valid   = false;
  ctx->ls_convection_generic_valid = true;
  // interpolation is linear: the sums of the two old states are formed at the nodes
  TRY(ctx, launch_lincomb(ctx, ctx->ls_stab_vel_sum.p, 1., vel_solution_old, 1., vel_solution_old_old, 3 * ctx->n_nodes_u),
      "sum failed");
  TRY(ctx, launch_lincomb(ctx, ctx->ls_stab_ls_sum.p, 1., solution_old, 1., solution_old_old, ctx->n_nodes_ls),
      "sum failed");
  LSStab st{};
  st.art_visc       = ctx->ls_art_visc.p;
  st.vel_sum        = ctx->ls_stab_vel_sum.p;
  st.ls_sum         = ctx->ls_stab_ls_sum.p;
  st.old_step_inv   = 1. / old_step_size;
  st.global_scaling = global_max_velocity * 2. * ctx->ls_omega_diameter; // :361
  st.bsign          = 1.;
  st.symmetry       = ctx->ls_symmetry;
  TRY(ctx,
      launch_ls(ctx, 2, 0, use_old_old, dst, solution, solution_old, solution_old_old, vel_solution,
                ctx->ls_convection.p, 1, &st),
      "level-set kernel launch failed");
  // :569-617 boundary part of the stabilisation term on the right-hand side
  TRY(ctx, launch_ls(ctx, 3, 0, 0, dst, solution, nullptr, nullptr, nullptr, nullptr, 1, &st),
      "level-set boundary kernel launch failed");
  return 0;
}

int adaflo_ls_reinitialization_vmult(adaflo_ctx *ctx, double *dst, const double *src, int diffuse_only)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (!diffuse_only && !ctx->ls_normal.p && !ctx->q1_normal_valid)
    return fail(ctx, ADAFLO_ENOTINIT, "evaluated_normal not set (run the rhs kernel with first_reinit_step)");
  return ls_vmult(ctx, dst, src, diffuse_only ? 2 : 1, 0, ctx->ls_normal.p, 1);
}

int adaflo_ls_reinitialization_rhs(adaflo_ctx *ctx, double *dst, const double *solution,
                                   const double *normal_vector_field, int diffuse_only,
                                   int first_reinit_step)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!dst || !solution || (!diffuse_only && first_reinit_step && !normal_vector_field))
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (!diffuse_only && !first_reinit_step && !ctx->ls_normal.p && !ctx->q1_normal_valid)
    return fail(ctx, ADAFLO_ENOTINIT, "evaluated_normal not set");
  const int flag = (diffuse_only ? 1 : 0) | (first_reinit_step ? 2 : 0);
  if (ctx->variant >= 1)
    {
      // sweep structure; the normal at the quadrature points lives in sweep layout (csrc/q1_sweep.hip)
      if (!diffuse_only)
        {
          if (first_reinit_step)
            TRY(ctx, q1_state_alloc(ctx, ctx->q1_normal), "out of device memory");
          else if (!ctx->q1_normal_valid)
            {
              TRY(ctx, q1_convert_state(ctx, ctx->q1_normal, ctx->ls_normal.p), "state re-layout failed");
              ctx->q1_normal_valid = true;
            }
        }
      const double *nv = normal_vector_field;
      TRY(ctx,
          launch_q1_rhs(ctx, 0, flag, dst, solution, nv, nv ? nv + ctx->n_nodes_ls : nullptr,
                        nv ? nv + 2 * ctx->n_nodes_ls : nullptr, nullptr, ctx->q1_normal.p),
          "level-set kernel launch failed");
      if (!diffuse_only && first_reinit_step)
        {
          ctx->q1_normal_valid         = true;
          ctx->ls_normal_generic_valid = false;
          ctx->q1_normal_nodal_valid   = false;
          static const bool nodal = getenv("ADAFLO_LS_STREAM_NORMAL") == nullptr;
          if (nodal)
            {
              const size_t count = 3 * (size_t)ctx->n_nodes_ls;
              if (ctx->q1_normal_nodal.count != count)
                TRY(ctx, alloc(ctx, ctx->q1_normal_nodal, count), ctx->last_error);
              HIP_TRY(ctx, hipMemcpyAsync(ctx->q1_normal_nodal.p, nv, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
              ctx->q1_normal_nodal_valid = true;
            }
        }
      return 0;
    }
  if (!diffuse_only)
    {
      if (first_reinit_step)
        {
          TRY(ctx, alloc(ctx, ctx->ls_normal, ls_q_count(ctx)), ctx->last_error);
          ctx->q1_normal_valid         = false;
          ctx->q1_normal_nodal_valid   = false;
          ctx->ls_normal_generic_valid = true;
        }
      else if (int e = ls_generic_state(ctx, 1))
        return e;
    }
  TRY(ctx,
      launch_ls(ctx, 1, 0 /*RHS_REINIT*/, flag, dst, solution, normal_vector_field, nullptr, nullptr, ctx->ls_normal.p, 1),
      "level-set kernel launch failed");
  return 0;
}

int adaflo_ls_compute_normal_vmult(adaflo_ctx *ctx, double *dst, const double *src)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  return ls_vmult(ctx, dst, src, 3 /*LS_NORMAL*/, 1, nullptr, 3);
}

int adaflo_ls_projection_vmult(adaflo_ctx *ctx, double *dst, const double *src)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  return ls_vmult(ctx, dst, src, 3 /*LS_NORMAL*/, 1, nullptr, 1); // one scalar block of the normal operator
}

int adaflo_ls_projection_solve(adaflo_ctx *ctx, double *dst, const double *rhs, int n_blocks)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!dst || !rhs || n_blocks < 1)
    return fail(ctx, ADAFLO_EINVAL, "invalid arguments");
  if (ctx->brick.con_ls != 0u) // (constrained rows of the operator carry the user's diagonal, not the identity)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "exact projection solve needs an unconstrained level-set space");
  if (ctx->flat)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "the fast-diagonalisation inverses are dim = 3 only");
  // the damping of compute_normal.cc:107-110 / level_set_okz.cc:262-312: 4 max(eps_used / eps, h / s)^2
  const LSDev &P     = ctx->ls;
  const double hcell = std::max(ctx->desc.h[0], std::max(ctx->desc.h[1], ctx->desc.h[2]));
  const double b     = std::max(P.epsilon_used / P.epsilon, hcell / ctx->s);
  for (int blk = 0; blk < n_blocks; ++blk)
    TRY(ctx, fdm_apply(ctx, 2, dst + (size_t)blk * ctx->n_nodes_ls, rhs + (size_t)blk * ctx->n_nodes_ls, 1., 4. * b * b),
        "fast-diagonalisation solve failed");
  return 0;
}


int adaflo_ls_compute_normal_rhs(adaflo_ctx *ctx, double *dst, const double *level_set_solution)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!dst || !level_set_solution)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (ctx->variant >= 1)
    {
      TRY(ctx, launch_q1_stencil_rhs(ctx, 0, dst, level_set_solution), "level-set kernel launch failed");
      return 0;
    }
  TRY(ctx, launch_ls(ctx, 1, 1 /*RHS_NORMAL*/, 0, dst, level_set_solution, nullptr, nullptr, nullptr, nullptr, 1),
      "level-set kernel launch failed");
  return 0;
}

int adaflo_ls_compute_curvature_vmult(adaflo_ctx *ctx, double *dst, const double *src, int apply_diffusion)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  return ls_vmult(ctx, dst, src, 4 /*LS_CURVATURE*/, apply_diffusion, nullptr, 1);
}

int adaflo_ls_mass_matrix_diagonal(adaflo_ctx *ctx, double *diagonal)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!diagonal)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  TRY(ctx, launch_ls_mass_diagonal(ctx, diagonal), "kernel launch failed");
  return 0;
}

int adaflo_ls_compute_heaviside(adaflo_ctx *ctx, double *heaviside, const double *level_set, double epsilon)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!heaviside || !level_set)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  TRY(ctx, launch_ls_heaviside(ctx, heaviside, level_set, epsilon), "heaviside kernels failed");
  return 0;
}

int adaflo_ls_curvature_correction(adaflo_ctx *ctx, double *curvature, const double *level_set)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!curvature || !level_set)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  TRY(ctx, launch_ls_curvature_correction(ctx, curvature, level_set), "curvature correction kernel failed");
  return 0;
}

int adaflo_ls_compute_force(adaflo_ctx *ctx, double *user_rhs_u, const double *heaviside,
                            const double *curvature, const adaflo_force_params *p)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!user_rhs_u || !heaviside || !curvature || !p)
    return fail(ctx, ADAFLO_EINVAL, "null argument");
  if (ctx->k > 5)
    return fail(ctx, ADAFLO_EUNSUPPORTED, "compute_force: velocity degree <= 5 (level_set_okz_template_instantations.h)");
  if (!ctx->d_tab_force)
    {
      const std::vector<double> tab = force_tables(ctx->s, ctx->k);
      TRY(ctx, upload(ctx, &ctx->d_tab_force, tab), ctx->last_error);
    }
  const bool variable = p->density_diff != 0. || p->viscosity_diff != 0.; // :333-334
  if (variable)
    {
      const size_t count = (size_t)ctx->n_cells * ctx->nq_u;
      const bool   had_damping = ctx->damp.p != nullptr;
      // (as adaflo_ns_set_coefficients: a state that exists only as the Q2/Q1 streaming copy WITHOUT coefficient pieces does
      // not depend on the coefficients and stays what it is -- this runs once per time step of a two-phase run, and the
      // re-layout it used to force was ~3 ms at 128^3 for nothing)
      const bool keep_q2 = (ctx->lin_q2.p && ctx->lin_q2_valid && !ctx->lin_q2_varco && !ctx->lin_generic_valid && !ctx->hox_lin_primary) ||
                       ctx->lin_q2_deferred; // (a deferred state is a function of the nodal field alone)
      if (!keep_q2)
        TRY(ctx, ensure_lin_generic(ctx), "state re-layout failed");
      TRY(ctx, alloc(ctx, ctx->rho, count), ctx->last_error);
      TRY(ctx, alloc(ctx, ctx->mu, count), ctx->last_error);
      TRY(ctx, alloc(ctx, ctx->damp, count), ctx->last_error);
      if (!had_damping) // variable_damping_coefficients default to parameters.damping
        TRY(ctx, launch_fill(ctx, ctx->damp.p, ctx->ns.damping, (long)count), "fill failed");
      if (!keep_q2)
        {
          ctx->lin_q2_valid    = false; // the streaming copies of the sweep kernels carry the coefficients
          ctx->hox_lin_primary = false;
        }
      ctx->lin_gen++;
      ctx->coef_gen++;
      ctx->q1_poisson_src = nullptr;
    }
  TRY(ctx,
      launch_ls_force(ctx, user_rhs_u, heaviside, curvature, ctx->d_tab_force, variable ? ctx->rho.p : nullptr,
                      variable ? ctx->mu.p : nullptr, p->surface_tension, p->gravity, p->density, p->density_diff,
                      p->viscosity, p->viscosity_diff, p->interpolate_grad_onto_pressure),
      "force kernel launch failed");
  return 0;
}

int adaflo_ls_compute_curvature_rhs(adaflo_ctx *ctx, double *dst, const double *normal_vector_field)
{
  CHECK_CTX(ctx);
  BRICK_ONLY(ctx);
  if (int e = ls_ready(ctx))
    return e;
  if (!dst || !normal_vector_field)
    return fail(ctx, ADAFLO_EINVAL, "null vector");
  if (ctx->variant >= 1)
    {
      TRY(ctx, launch_q1_stencil_rhs(ctx, 1, dst, normal_vector_field), "level-set kernel launch failed");
      return 0;
    }
  TRY(ctx, launch_ls(ctx, 1, 2 /*RHS_CURVATURE*/, 0, dst, normal_vector_field, nullptr, nullptr, nullptr, nullptr, 1),
      "level-set kernel launch failed");
  return 0;
}

} // extern "C"
