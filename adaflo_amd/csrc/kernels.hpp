// kernels.hpp -- internal launch interface between the C ABI (capi.hip) and the
// kernel translation units.  All functions enqueue on ctx->stream and return a
// hipError_t-like int (0 = ok).
#pragma once
#include "ctx.hpp"

#include <vector>

namespace adaflo_hip
{
  // Host -> device copy of set-up data (1D tables, plans) that kernels on the context's stream read next.
  // A synchronous hipMemcpy from pageable memory may return as soon as the data is staged, and only work on
  // the NULL stream is ordered after it -- the contexts' streams are created non-blocking, so a kernel
  // launched right afterwards could read the buffer before the DMA has landed (seen as a 1-in-20 wrong
  // pressure mass weight vector right after adaflo_ctx_create).  Hence: copy, then wait for the device.
  inline hipError_t copy_to_device_now(void *dst, const void *src, const size_t bytes)
  {
    const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    return e != hipSuccess ? e : hipDeviceSynchronize();
  }
  // the colours a coloured cell loop runs through: the eight parities of the brick, or the colour ranges of an indexed context
  inline int n_colours(const adaflo_ctx *ctx) { return ctx->indexed ? (int)ctx->idx_colour_off.size() - 1 : 8; }
  inline long cells_of_colour(const adaflo_ctx *ctx, const BrickDev &b, const int colour)
  {
    return ctx->indexed ? ctx->idx_colour_off[colour + 1] - ctx->idx_colour_off[colour] : n_cells_of_colour(b.ncell, colour);
  }
  inline void set_colour(const adaflo_ctx *ctx, BrickDev &b, const int colour)
  {
    b.colour = colour;
    if (ctx->indexed)
      b.cell_first = ctx->idx_colour_off[colour];
  }
  // x-marching Q_k/Q_{k-1} kernel for k = 3, 4, 5 (ns_hox.hip), constant coefficients: the default since round 4
  bool hox_supported(const adaflo_ctx *ctx);
  bool hox_residual_supported(const adaflo_ctx *ctx);
  int  launch_ns_residual_hox(adaflo_ctx *ctx, double *sum_u, double *sum_p, const double *src_u, const double *src_p,
                              const double *old_comb, const double *ext_comb = nullptr);
  int  hox_unconvert_state(adaflo_ctx *ctx, double *generic, bool frozen_copy = false);
  int  launch_ns_vmult_hox(adaflo_ctx *ctx, int op, double *dst_u, double *dst_p, const double *src_u,
                           const double *src_p, int phase = -1, uint32_t iface = 0);
  // plane-per-lane Q4/Q3 kernel (ns_hop.hip), constant coefficients, vmult and velocity_vmult
  bool hop_supported(const adaflo_ctx *ctx, int op);
  int  launch_ns_vmult_hop(adaflo_ctx *ctx, int op, double *dst_u, double *dst_p, const double *src_u,
                           const double *src_p, int phase = -1, uint32_t iface = 0);
} // namespace adaflo_hip

namespace adaflo_hip
{
  enum NSOp
  {
    OP_VMULT          = 0, // NavierStokesOps::vmult        include/adaflo/navier_stokes_matrix.h:36-41
    OP_RESIDUAL       = 1,
    OP_VMULT_VELOCITY = 2,
    OP_DIVERGENCE     = 3 // Q2/Q1 sweep kernel only: the divergence block
  };

  constexpr int NLIN = 12; // dim + dim*dim doubles of linearisation state per q-point

  struct NSArgs
  {
    BrickDev      brick;
    NSDev         ns;
    const double *src_u, *src_p, *old_u, *oldold_u;
    double       *dst_u, *dst_p;
    double       *lin;
    const double *rho, *mu, *damp;
    const double *tab;
    long          n_cells;
  };

  // generic cell kernels (ns_generic.hip)
  int launch_ns_cell_generic(adaflo_ctx *ctx, int op, const NSArgs &args);
  // diagonal of the velocity block, added into diag (zero on entry)
  int launch_ns_velocity_diagonal(adaflo_ctx *ctx, const NSArgs &args, double *diag);

  struct ScalarArgs
  {
    BrickDev      brick;
    NSDev         ns;
    const double *src;
    double       *dst;
    const double *coef_q; // per-q coefficient array in generic layout [cell][q] or nullptr
    const double *tab;
    long          n_cells;
    int           mode;
    int           nq_u3; // number of q-points of quad_index_u per cell ("mid-cell" sampling)
  };
  enum ScalarMode
  {
    SC_DIVERGENCE = 0,       // local_divergence              :920-961
    SC_DIVERGENCE_VISC,      //   weight_by_viscosity
    SC_POISSON_VARIABLE,     // local_pressure_poisson, variable rho at quad_u :984-1000
    SC_POISSON_CELL,         // local_pressure_poisson, per-cell coefficient at quad_p :1002-1031
    SC_MASS,                 // local_pressure_mass           :1036-1071
    SC_MASS_WEIGHT,          // local_pressure_mass_weight    :1075-1095
    SC_CONVDIFF              // local_pressure_convdiff       :1099-1140
  };
  int launch_ns_scalar_generic(adaflo_ctx *ctx, const ScalarArgs &args);

  // interface regions of a brick vector for the inter-GPU exchange (faces, edges, corners)
  struct HaloPlan
  {
    int  n_regions, ncomp;
    int  nn[3];
    int  lo[26][3], hi[26][3]; // half-open node index ranges
    long offset[27];           // prefix sums of the region sizes (doubles): the work items of the kernel
    long start[26];            // position of region r in the packed buffer (== offset[r] when back to back)
    int  self_pos;             // add mode: the vector's own value is summed before region self_pos
  };
  int launch_halo(adaflo_ctx *ctx, double *vec, double *buf, const HaloPlan &plan, int mode);
  // two fields in one launch; mode 3: copy, the last region that contains a node wins (regions ordered by class)
  int launch_halo_pair(adaflo_ctx *ctx, double *vec0, double *vec1, double *buf, const HaloPlan &plan0,
                       const HaloPlan &plan1, int mode);

  // vector helpers (vector_ops.hip)
  // dst = constrained ? sign*src : 0  (fuses `dst = 0` with the constrained-row
  // fix-up of source/navier_stokes_matrix.cc:229,247-256)
  int launch_prepare_dst(adaflo_ctx *ctx, double *dst, const double *src, long n_nodes, int ncomp,
                         int nnx, int nny, int nnz, uint32_t mask, double sign, bool zero_rest);
  // dst = sign * src on the constrained faces only (the rows of :247-256), nothing else touched
  int launch_constrained_faces(adaflo_ctx *ctx, double *dst, const double *src, int ncomp, int nnx, int nny, int nnz,
                               uint32_t mask, double sign);
  // v -= (w.v) * inv * modes   (apply_pressure_average_projection :191-205)
  int launch_mean_projection(adaflo_ctx *ctx, double *v, const double *w, const double *modes,
                             long n, double inv);
  int launch_sadd(adaflo_ctx *ctx, double *x, double a, const double *y, long n); // x = a*x + y
  // engine numbering <-> an application's numbering through a device-resident index map
  int launch_map_gather(adaflo_ctx *ctx, double *eng, const double *ext, const long long *map, long n);
  int launch_map_scatter(adaflo_ctx *ctx, double *ext, const double *eng, const long long *map, long n, int add);
  int launch_lincomb(adaflo_ctx *ctx, double *z, double a, const double *x, double b, const double *y, long n); // z = a x + b y
  int launch_residual_finish(adaflo_ctx *ctx, double *rhs, const double *sum, const double *user, long n); // rhs = user - rhs - sum
  int launch_fill(adaflo_ctx *ctx, double *x, double v, long n);
  // canonical [cell][q][comp] <-> generic [cell][comp][q]
  int launch_transpose_state(adaflo_ctx *ctx, double *dst, const double *src, long n_cells, int nq,
                             int ncomp, bool to_generic);
  double host_dot(adaflo_ctx *ctx, const double *a, const double *b, long n);
  // device-resident scalars of the distributed mean-value projection (comm.hip)
  int launch_dot_to(adaflo_ctx *ctx, const double *a, const double *b, long n, double *out); // *out = a . b
  int launch_gs_step(adaflo_ctx *ctx, double *w, const double *coefficient, const double *v, const double *next, long n,
                     double *out);
  int launch_sum_to(adaflo_ctx *ctx, const double *a, long n, double *out);                  // *out = sum a
  int launch_reciprocal(adaflo_ctx *ctx, double *out, const double *in);                     // *out = 1 / *in
  int launch_subtract_scaled(adaflo_ctx *ctx, double *v, const double *s, const double *t, long n); // v -= *s * *t

  // level-set operators (ls_kernels.hip); kind: 0 operator application, 1 rhs, 2 advection rhs
  // optional convection stabilisation of the advection operator (ls_kernels.hip)
  struct LSStab
  {
    double       *art_visc;
    const double *vel_sum, *ls_sum;
    double        old_step_inv, global_scaling, bsign;
    unsigned      symmetry;
  };
  int launch_ls(adaflo_ctx *ctx, int kind, int mode, int flag, double *dst, const double *src,
                const double *src2, const double *src3, const double *vel, double *qstate,
                int ncomp_blocks, const LSStab *stab = nullptr);
  int launch_ls_max_velocity(adaflo_ctx *ctx, const double *vel, const double *tab, unsigned long long *result);
  int launch_ls_constrained_rows(adaflo_ctx *ctx, double *dst, const double *src, int nblocks);

  // structured Q1 sweep kernel (q1_sweep.hip): level-set operators on the s-times refined grid
  // (sub = s) and the Q1 pressure mass / Poisson operators of the Q2/Q1 pair (sub = 1)
  enum Q1Mode
  {
    Q1_MASS_LAPLACE = 0, // (w, c_mass v) + (grad w, c_lap grad v)
    Q1_ADVECT       = 1, // (w, weight v + u . grad v)
    Q1_REINIT       = 2, // (w, c_mass v) + (grad w, c_lap (n . grad v) n)
    Q1_LAPLACE_Q3   = 3, // (grad w, c(x_q) grad v) with the 3x3x3 Gauss rule, c per point (pressure Poisson)
    Q1_REINIT_NODAL = 4, // Q1_REINIT with the unit normal at the Gauss points recomputed from the NODAL normal field
                         // (`state` = three nodal fields) instead of streamed: 24 B instead of 192 B per sub-cell
    Q1_ADVECT_NODAL = 5  // Q1_ADVECT with the FE_Q(k) velocity evaluated at the Gauss points from the NODAL velocity
                         // (`state` = velocity vector [node][3]) as the advection right-hand side does, instead of streamed
  };
  int q1_convert_state(adaflo_ctx *ctx, DeviceBuffer &out, const double *generic_dev);
  int launch_q1_stencil_rhs(adaflo_ctx *ctx, int mode, double *dst, const double *src);
  // Q2/Q1 divergence as tensor-product stencil (ns_divergence.hip)
  bool divergence_stencil_supported(const adaflo_ctx *ctx);
  int  launch_ns_divergence_stencil(adaflo_ctx *ctx, double *dst_p, const double *src_u, double weight, bool plain);
  int q1_state_alloc(adaflo_ctx *ctx, DeviceBuffer &out);
  int q1_unconvert_state(adaflo_ctx *ctx, double *generic_dev, const DeviceBuffer &sweep);
  // (state == nullptr: the advection does not write evaluated_convection; state_only: nothing but that array)
  int launch_q1_rhs(adaflo_ctx *ctx, int kind, int flag, double *dst, const double *f0, const double *f1,
                    const double *f2, const double *f3, const double *vel, double *state, bool state_only = false);
  int launch_q1_sweep(adaflo_ctx *ctx, int sub, int mode, double c_mass, double c_lap, double weight,
                      uint32_t con, double con_sign, const double *diag, double *dst, const double *src,
                      const double *state, int n_blocks = 1, const double *coef_cell = nullptr, int coef_stride = 0,
                      int coef_mid = 0, double coef_shift = 0.);
  int q1_convert_poisson_coef(adaflo_ctx *ctx, DeviceBuffer &out, const double *rho_generic, double weight);
  bool q1_advect_nodal_supported(const adaflo_ctx *ctx);

  // specialised 3D Q2/Q1 sweep kernel (ns_q2.hip)
  bool q2_supported(const adaflo_ctx *ctx);
  int  q2_prepare_state(adaflo_ctx *ctx);
  int  launch_ns_vmult_q2(adaflo_ctx *ctx, int op, double *dst_u, double *dst_p,
                          const double *src_u, const double *src_p, int phase = -1, uint32_t iface = 0);
  // recompute-state mode: nodal copy of the solution a residual has just been evaluated at (no-op unless the mode applies)
  int  q2_capture_nodal(adaflo_ctx *ctx, const double *src_u);
  inline bool lin_nodal_current(const adaflo_ctx *ctx)
  {
    return ctx->lin_nodal.p && ctx->lin_nodal_serial == ctx->lin_serial;
  }
  // residual mode of the sweep kernel (writes the quadrature-point state in the streaming layout)
  bool q2_residual_supported(const adaflo_ctx *ctx);
  int  launch_ns_residual_q2(adaflo_ctx *ctx, double *sum_u, double *sum_p, const double *src_u,
                             const double *src_p, const double *old_comb, const double *ext_comb = nullptr);
  int  q2_unconvert_state(adaflo_ctx *ctx, double *generic, const double *streaming, int lin_mode);
  // lazy state: lays out the state a residual deferred (ctx->lin_q2_deferred), no-op otherwise
  int  q2_materialize_state(adaflo_ctx *ctx);
  int  launch_ns_divergence_q2(adaflo_ctx *ctx, double *sum_p, const double *src_u, const double *any_p, double weight);
  // dst[i] += src[i] for the entries that are not on a constrained face
  int launch_add_unconstrained(adaflo_ctx *ctx, double *dst, const double *src, long n_nodes, int ncomp, int nnx,
                               int nny, int nnz, uint32_t mask);

  // compute_heaviside / local_compute_force (ls_force.hip)
  int                 launch_ls_heaviside(adaflo_ctx *ctx, double *heaviside, const double *phi, double epsilon);
  std::vector<double> force_tables(int s, int k);
  int                 launch_ls_mass_diagonal(adaflo_ctx *ctx, double *diag);
  int                 launch_ls_curvature_correction(adaflo_ctx *ctx, double *curvature, const double *phi);
  int launch_ls_force(adaflo_ctx *ctx, double *dst_u, const double *heaviside, const double *curvature,
                      const double *tab, double *rho, double *mu, double surface_tension, double gravity,
                      double density, double density_diff, double viscosity, double viscosity_diff,
                      int on_pressure);

  // fast-diagonalisation inverse of c_mass M + c_lap K on the velocity (field 0) / pressure (field 1)
  // space of the brick (fdm.hip); dst = src on constrained rows
  int  fdm_setup(adaflo_ctx *ctx);
  int  fdm_apply(adaflo_ctx *ctx, int field, double *dst, const double *src, double c_mass, double c_lap,
                 double c_mass2 = 0., double c_lap2 = 0.);
  void fdm_destroy(adaflo_ctx *ctx);

  // Q_k/Q_{k-1} sweep kernel for k = 3, 4, 5 (ns_ho.hip), constant coefficients
  bool ho_supported(const adaflo_ctx *ctx);
  int  launch_ns_vmult_ho(adaflo_ctx *ctx, int op, double *dst_u, double *dst_p, const double *src_u,
                          const double *src_p, int phase = -1, uint32_t iface = 0);
} // namespace adaflo_hip
