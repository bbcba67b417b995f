/* adaflo_hip.h -- C ABI of the MI355X-native matrix-free operator engine.
 *
 * This is the drop-in boundary for adaflo's operator hot path.  The reference
 * (kronbichler/adaflo) has no FFI layer; its seam is the C++ duck-typed
 * operator interface required by deal.II's Krylov solvers
 * (`void vmult(VectorType &dst, const VectorType &src) const`).  Every entry
 * point below names the reference method it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the deal.II-side adapter that forwards
 * `NavierStokesMatrix<dim>::vmult` & friends to these functions.
 *
 * Conventions
 *  - All functions return 0 on success, a negative ADAFLO_E* code on failure;
 *    adaflo_last_error() returns a human-readable message.  No exceptions cross
 *    the boundary (the reference throws deal.II exceptions instead).
 *  - One context = one GPU + one HIP stream; like the reference operators
 *    (mutable timers / array swaps, source/navier_stokes_matrix.cc:227,349-375)
 *    a context is NOT re-entrant.
 *  - All `double *` vector arguments are DEVICE pointers (hipMalloc / a
 *    torch.cuda tensor's data_ptr) holding one deal.II
 *    LinearAlgebra::distributed::Vector<double> block: a contiguous
 *    double[n_dofs].  adaflo_malloc/adaflo_copy_* are provided so a host-only
 *    caller (ctypes, the deal.II adapter) can stage vectors.
 *  - Mesh: structured axis-aligned brick of ncell[0] x ncell[1] x ncell[2]
 *    hexahedra of size h[d].  DoF numbering (documented, "not part of the
 *    reference contract"):
 *        nodes lexicographic over the brick (x fastest),
 *        FE_Q(k) nodes per direction: k*ncell[d]+1 (Gauss-Lobatto support points),
 *        velocity dof = node*dim + component (components interleaved per
 *        node, as in deal.II's FESystem vectors), pressure dof = node,
 *        level-set dof = node of FE_Q_iso_Q1(s): s*ncell[d]+1 per direction.
 *    Cells are lexicographic c = cx + ncx*(cy + ncy*cz); quadrature points in
 *    a cell lexicographic q = qx + n*(qy + n*qz).
 *  - Quadrature-point operator state (the reference's public begin_*
 *    accessors, include/adaflo/navier_stokes_matrix.h:162-180) crosses the
 *    boundary in CANONICAL layout [cell][q][component]; the engine converts to
 *    its internal streaming layout.
 */
#ifndef ADAFLO_HIP_H
#define ADAFLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct adaflo_ctx adaflo_ctx;

enum
{
  ADAFLO_OK            = 0,
  ADAFLO_EINVAL        = -1, /* bad argument / unsupported degree (reference: ExcNotImplemented) */
  ADAFLO_ENOTINIT      = -2, /* reference: ExcNotInitialized */
  ADAFLO_EHIP          = -3, /* HIP runtime error */
  ADAFLO_ENOMEM        = -4,
  ADAFLO_EUNSUPPORTED  = -5
};

/* FlowParameters::PhysicalType, include/adaflo/parameters.h:57-62 */
enum { ADAFLO_INCOMPRESSIBLE = 0, ADAFLO_INCOMPRESSIBLE_STATIONARY = 1, ADAFLO_STOKES = 2 };
/* FlowParameters::Linearization, include/adaflo/parameters.h:77-84 */
enum
{
  ADAFLO_COUPLED_IMPLICIT_NEWTON        = 0,
  ADAFLO_COUPLED_IMPLICIT_PICARD        = 1,
  ADAFLO_COUPLED_VELOCITY_SEMI_IMPLICIT = 2,
  ADAFLO_COUPLED_VELOCITY_EXPLICIT      = 3,
  ADAFLO_PROJECTION                     = 4
};

/* Structured-brick descriptor: replaces what NavierStokesMatrix::initialize()
 * pulls out of MatrixFree<dim> (dof indices, constrained dofs, mapping info),
 * source/navier_stokes_matrix.cc:85-168 and source/navier_stokes.cc:396-502. */
typedef struct
{
  int      dim;             /* 3 */
  int      ncell[3];
  double   h[3];
  double   origin[3];
  int      velocity_degree; /* k >= 2; pressure degree k-1 (Taylor-Hood) */
  int      ls_degree;       /* s: FE_Q_iso_Q1(s) subdivisions; 0 = no level-set spaces */
  /* homogeneous Dirichlet constraints by boundary face f = 2*d + side
   * (deal.II hyper_rectangle boundary ids): bit (3*f + comp) for velocity,
   * bit f for pressure / level set.                                          */
  uint32_t velocity_constrained;
  uint32_t pressure_constrained;
  uint32_t ls_constrained;
  int      pressure_average_fix; /* initialize(..., pressure_average_fix), :117-168 */
  int      device;               /* HIP device ordinal */
  void    *stream;               /* hipStream_t to run on, NULL = create one */
} adaflo_brick_desc;

/* The scalars local_operation reads, source/navier_stokes_matrix.cc:621-653,
 * plus TimeStepping::{weight,weight_old,weight_old_old,tau1,extrapolate}
 * (source/time_stepping.cc:123-200).                                         */
typedef struct
{
  int    physical_type;
  int    linearization;
  double beta;           /* beta_convective_term_momentum_balance */
  double tau_grad_div;
  double density;
  double viscosity;
  double damping;        /* sign as stored by FlowParameters (flipped), parameters.cc:466-467 */
  double density_diff;
  double weight, weight_old, weight_old_old, tau1;
  double extrap_old, extrap_old_old;
} adaflo_ns_params;

/* ---- indexed context (first alternative) -------------------
 * What an adapter copies out of deal.II's MatrixFree / DoFHandler for ANY Cartesian hexahedral mesh, locally refined ones
 * included: per-cell DoF tables, constrained-DoF flags, hanging-node constraints, the diagonal of every cell's Jacobian and
 * a colouring of the cells -- replaces MatrixFree<dim>::reinit for meshes that are not one brick (an L-shaped channel, a
 * box with an obstacle, the two refined cells of tests/beltrami.cc:403-412; source/navier_stokes.cc:396-502 builds its
 * MatrixFree from whatever the triangulation is).  All arrays are
 * HOST memory and are copied.  Numbering: velocity dof = node * 3 + component with the adapter's own node numbers
 * (0 .. n_nodes_u - 1), pressure dof = node; inside a cell the nodes are listed lexicographically (x fastest) in the
 * cell's local Gauss-Lobatto lattice, as FE_Q's hierarchical-to-lexicographic numbering delivers them.
 * The context runs the GENERIC kernels of the Navier-Stokes block: adaflo_ns_vmult, _residual, _velocity_vmult,
 * _divergence_vmult_add, _pressure_{poisson,mass,convdiff}_vmult, the pressure-mean projection, the quadrature-point stores
 * ([cell] = the order of the tables).  The sweep kernels, the level-set operators, the device Krylov drivers and the
 * communicator need the brick and return ADAFLO_EUNSUPPORTED. */
typedef struct
{
  int     device;
  void   *stream;              /* hipStream_t or NULL: engine-owned stream */
  int     velocity_degree;     /* k of Q_k / Q_{k-1}, 2 ... 6 */
  int     pressure_average_fix;
  int64_t n_cells, n_nodes_u, n_nodes_p;
  const int           *cell_nodes_u;  /* [n_cells][(k+1)^3] */
  const int           *cell_nodes_p;  /* [n_cells][k^3] */
  const unsigned char *constrained_u; /* [n_nodes_u * 3]: 1 = constrained row (Dirichlet, symmetry, ...) */
  const unsigned char *constrained_p; /* [n_nodes_p] */
  const double        *cell_extents;  /* [n_cells][3] edge lengths hx, hy, hz of every cell, or NULL: all cells `h` */
  double               h[3];
  int                  n_colours;     /* the cells are sorted by colour: colour c = cells [colour_offsets[c], colour_offsets[c+1]) */
  const int64_t       *colour_offsets; /* [n_colours + 1]; no two cells of one colour share a node (masters of hanging nodes count) */
  /* Hanging nodes (AffineConstraints lines with entries, DoFTools::make_hanging_node_constraints, source/navier_stokes.cc:241-242): a cell
   * table entry e < 0 names hanging node h = -1 - e, whose value is sum_j weight[j] * value(master[j]), j in
   * [ptr[h], ptr[h+1]) -- FEEvaluation::read_dof_values interpolates, distribute_local_to_global sends the cell's sums to
   * the masters with the same weights.  Masters are regular nodes (chains resolved by the adapter, as AffineConstraints::
   * close() does).  In the vectors a hanging node keeps its own number and is flagged in constrained_u / _p, so its row
   * is the identity like every constrained row (navier_stokes_matrix.cc:247-256).  All zero / NULL: no hanging nodes. */
  int64_t              n_hanging_u, n_hanging_p;
  const int64_t       *hanging_ptr_u, *hanging_ptr_p;       /* [n_hanging + 1] */
  const int           *hanging_master_u, *hanging_master_p; /* [ptr[n_hanging]] node numbers */
  const double        *hanging_weight_u, *hanging_weight_p; /* [ptr[n_hanging]] */
} adaflo_indexed_desc;
int         adaflo_ctx_create_indexed(const adaflo_indexed_desc *desc, adaflo_ctx **out);

/* ---- context ------------------------------------------------------------ */
int         adaflo_ctx_create(const adaflo_brick_desc *desc, adaflo_ctx **out);
int         adaflo_ctx_destroy(adaflo_ctx *ctx);              /* NavierStokesMatrix::clear */
const char *adaflo_last_error(const adaflo_ctx *ctx);         /* ctx may be NULL: last create error */
int         adaflo_synchronize(adaflo_ctx *ctx);
void       *adaflo_stream(adaflo_ctx *ctx);
/* run all further work of the context on the caller's stream; NULL selects the legacy default
 * stream (which is what torch uses as its current stream unless told otherwise)           */
int         adaflo_set_stream(adaflo_ctx *ctx, void *stream);

int64_t adaflo_n_cells(const adaflo_ctx *ctx);
int64_t adaflo_n_dofs_u(const adaflo_ctx *ctx);               /* NavierStokesMatrix::n_dofs_u */
int64_t adaflo_n_dofs_p(const adaflo_ctx *ctx);               /* NavierStokesMatrix::n_dofs_p */
int64_t adaflo_n_dofs_ls(const adaflo_ctx *ctx);
int     adaflo_n_q_points_u(const adaflo_ctx *ctx);           /* (k+1)^dim, quad_index_u */
int     adaflo_n_q_points_ls(const adaflo_ctx *ctx);          /* (2s)^dim */

/* ---- device memory helpers (for host-only callers) ---------------------- */
int adaflo_malloc(adaflo_ctx *ctx, size_t bytes, void **dptr);
int adaflo_free(adaflo_ctx *ctx, void *dptr);
int adaflo_copy_h2d(adaflo_ctx *ctx, void *dst, const void *src, size_t bytes);
int adaflo_copy_d2h(adaflo_ctx *ctx, void *dst, const void *src, size_t bytes);

/* ---- Navier-Stokes operator state --------------------------------------- */
int adaflo_ns_set_params(adaflo_ctx *ctx, const adaflo_ns_params *p);
/* linearized_velocities (navier_stokes_matrix.h:54-56,178): canonical
 * [cell][q][dim + dim*dim] = (u_lin[d], grad_lin[d][e]); host or device source. */
int adaflo_ns_set_linearization(adaflo_ctx *ctx, const double *lin, int src_on_device);
int adaflo_ns_get_linearization(adaflo_ctx *ctx, double *lin, int dst_on_device);
/* begin_densities / begin_viscosities / begin_damping_coeff: canonical [cell][q];
 * all NULL = constant coefficients (use_variable_coefficients() == false).      */
int adaflo_ns_set_coefficients(adaflo_ctx *ctx, const double *rho, const double *mu,
                               const double *damping, int src_on_device);
/* read access to the same stores (begin_densities etc.); any pointer may be NULL */
int adaflo_ns_get_coefficients(adaflo_ctx *ctx, double *rho, double *mu, double *damping, int dst_on_device);
int adaflo_ns_fix_linearization_point(adaflo_ctx *ctx);       /* :1144-1152 */

/* The same operator in three parts, for overlapping the inter-GPU ghost exchange with the
 * evaluation of interior cells (the reference overlaps inside MatrixFree::cell_loop's
 * update_ghost_values_start/finish and compress_start/finish around
 * source/navier_stokes_matrix.cc:232-245).  interface_faces: bit 2*dim+side set for the faces of
 * the brick that are shared with another GPU.
 *   phase 0: cells that touch no interface node (first half)  -- needs no ghost values of src
 *   phase 1: cells that touch the interface; afterwards dst is final on the interface nodes
 *            (ready for compress(add)); requires the ghost values of src
 *   phase 2: the remaining interior cells and seam sums
 *   phases 3, 4, 5 (the two-stream schedule of adaflo_ns_vmult_distributed): 3 = ALL interior cells (0 and 2 in one
 *            launch) without the seam sums, 4 = the seam sums of phase 2 alone (after phases 1 and 3), 5 = set-up only
 *            (tables and streaming copies of the state brought up to date, nothing launched on the cells)
 * After phase 2 dst equals the result of adaflo_ns_vmult without the mean-value projection.
 * Only with the sweep kernels -- Q2/Q1 (constant or variable coefficients) and Q3/Q2 .. Q5/Q4 (constant
 * coefficients) --, ADAFLO_EUNSUPPORTED otherwise; adaflo_ns_supports_phases tells (1 / 0) for the
 * current degree, coefficients and kernel variant. */
int adaflo_ns_vmult_phase(adaflo_ctx *ctx, double *dst_u, double *dst_p, const double *src_u,
                          const double *src_p, int phase, unsigned interface_faces);
int adaflo_ns_supports_phases(adaflo_ctx *ctx);

/* ---- Navier-Stokes operators (source/navier_stokes_matrix.cc) ----------- */
/* vmult :221-262 */
int adaflo_ns_vmult(adaflo_ctx *ctx, double *dst_u, double *dst_p, const double *src_u,
                    const double *src_p);
/* residual :266-293; system_rhs is read-modify-written exactly like the
 * reference (cell loop adds into it, then rhs = -rhs + user_rhs); user_* may be
 * NULL (= 0); solution_old / solution_old_old velocity blocks are the
 * references the reference's constructor takes (navier_stokes_matrix.h:57-64). */
int adaflo_ns_residual(adaflo_ctx *ctx, double *rhs_u, double *rhs_p, const double *src_u,
                       const double *src_p, const double *user_u, const double *user_p,
                       const double *old_u, const double *old_old_u);
/* velocity_vmult :337-382 */
int adaflo_ns_velocity_vmult(adaflo_ctx *ctx, double *dst_u, const double *src_u);
/* Diagonal of the operator of velocity_vmult (1 on constrained rows, like A e_i there), computed cell by cell
 * from the quadrature-point operation itself.  It stands for the diagonal of the velocity block of the
 * preconditioner matrix the reference assembles (navier_stokes_preconditioner.cc:135-300, used by its
 * ILU / AMG); here it is the Jacobi diagonal of the inner velocity solves for variable coefficients. */
int adaflo_ns_velocity_block_diagonal(adaflo_ctx *ctx, double *diagonal_u);
/* divergence_vmult_add :300-332 (dst NOT zeroed) */
int adaflo_ns_divergence_vmult_add(adaflo_ctx *ctx, double *dst_p, const double *src_u,
                                   int weight_by_viscosity);
/* pressure_poisson_vmult :386-417 */
int adaflo_ns_pressure_poisson_vmult(adaflo_ctx *ctx, double *dst_p, const double *src_p);
/* pressure_mass_vmult :421-455 */
int adaflo_ns_pressure_mass_vmult(adaflo_ctx *ctx, double *dst_p, const double *src_p);
/* pressure_convdiff_vmult :459-483 */
int adaflo_ns_pressure_convdiff_vmult(adaflo_ctx *ctx, double *dst_p, const double *src_p);
/* cell loop of local_pressure_mass_weight :1075-1095 (dst += int phi_i); building
 * block of initialize() :123-131, exposed for the distributed set-up            */
int adaflo_ns_pressure_mass_weight_add(adaflo_ctx *ctx, double *dst_p);
/* the constrained-row fix-up of vmult :247-256 alone (dst_u[c] = src_u[c],
 * dst_p[c] = -src_p[c]); needed again after an inter-GPU compress(add)          */
int adaflo_ns_apply_constrained_rows(adaflo_ctx *ctx, double *dst_u, double *dst_p,
                                     const double *src_u, const double *src_p);
/* apply_pressure_average_projection :191-205 */
int adaflo_ns_apply_pressure_average_projection(adaflo_ctx *ctx, double *vec_p);
/* get_matvec_statistics :1194-1206: number of vmult calls and accumulated wall
 * seconds since the last query (resets the counters like the reference).      */
int adaflo_ns_get_matvec_statistics(adaflo_ctx *ctx, unsigned *count, double *seconds);

/* ---- multi-GPU: packing of interface DoFs ---------------------------------- */
/* What deal.II's Utilities::MPI::Partitioner does with its import/export index lists
 * inside update_ghost_values() / compress(add): copy the n_regions node boxes
 * regions[r] = {i0,i1, j0,j1, k0,k1} (half-open) of the brick vector `vec` (nn[3] nodes,
 * ncomp interleaved components) to / from the packed message buffer `buf` (regions back to
 * back).  mode 0: buf <- vec, 1: vec <- buf, 2: vec += buf (overlapping regions are summed in
 * region order without atomics).  Building block of adaflo_comm_* below; also callable on its
 * own by a host that sends the messages itself (adaflo_amd/parallel.py over torch.distributed). */
int adaflo_halo_transfer(adaflo_ctx *ctx, double *vec, double *buf, const int *nn, int ncomp,
                         int n_regions, const int *regions, int mode);
/* the same with the position of this rank among the senders of the regions (which must arrive
 * sorted by the rank of their sender): in add mode the vector's own value enters the sum at that
 * position, so that all sharers of a DoF add the same numbers in the same order and their
 * replicas stay bitwise identical */
int adaflo_halo_transfer_ordered(adaflo_ctx *ctx, double *vec, double *buf, const int *nn, int ncomp, int n_regions,
                                 const int *regions, int mode, int self_pos);

/* ---- multi-GPU: the exchange itself --------------------------------------------------------
 * What MatrixFree::cell_loop does across MPI ranks for a LinearAlgebra::distributed::Vector
 * (source/navier_stokes_matrix.cc:232-245): import the ghost values of src, run the local cells,
 * add the ghost contributions of dst to their owners -- plus the MPI sum of the pressure-mean
 * projection (:201).  One process per GPU; the mesh is cut into grid[0] x grid[1] x grid[2]
 * bricks, rank = cx + grid[0] (cy + grid[1] cz); every rank creates its engine context on ITS
 * brick (Dirichlet faces only where the brick touches the domain boundary,
 * pressure_average_fix = 0: the communicator applies the global projection).  Nodes on an
 * inter-rank interface are replicated on the sharers; the sharer with the lowest grid
 * coordinates owns them.  Per exchange one message per neighbour (<= 26; 7 on 2 x 2 x 2),
 * posted as ONE ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a second HIP stream.
 *
 * adaflo_comm_create: RCCL transport.  Rank 0 obtains the id with adaflo_comm_get_unique_id and
 * broadcasts the 128 bytes out of band (MPI_Bcast in a deal.II application); collective over
 * the `world` ranks.  adaflo_comm_create_custom: the caller moves the packed buffers itself --
 *   exchange(user, sendbuf, send_offset[], send_count[], send_peer[], n_send,
 *                  recvbuf, recv_offset[], recv_count[], recv_peer[], n_recv, stream)
 * sends send_count[q] doubles at sendbuf + send_offset[q] to rank send_peer[q] and receives
 * likewise (device pointers; everything enqueued on `stream` before the call must be complete
 * before the data is read, and the received data must be in place when work enqueued on
 * `stream` afterwards runs); allreduce(user, buf, n, stream) sums n device doubles over all
 * ranks in place under the same ordering rules.  Both return 0 on success.                   */
typedef struct adaflo_comm adaflo_comm;
typedef struct
{
  char internal[128]; /* = ncclUniqueId */
} adaflo_comm_unique_id;
typedef int (*adaflo_exchange_fn)(void *user, double *sendbuf, const int64_t *send_offset, const int64_t *send_count,
                                  const int *send_peer, int n_send, double *recvbuf, const int64_t *recv_offset,
                                  const int64_t *recv_count, const int *recv_peer, int n_recv, void *stream);
typedef int (*adaflo_allreduce_fn)(void *user, double *buf, int n, void *stream);

int         adaflo_comm_get_unique_id(adaflo_comm_unique_id *id);
int         adaflo_comm_create(adaflo_ctx *ctx, const adaflo_comm_unique_id *id, int rank, int world, const int *grid,
                               int pressure_average_fix, adaflo_comm **out);
int         adaflo_comm_create_custom(adaflo_ctx *ctx, int rank, int world, const int *grid, adaflo_exchange_fn exchange,
                                      adaflo_allreduce_fn allreduce, void *user, int pressure_average_fix,
                                      adaflo_comm **out);
int         adaflo_comm_destroy(adaflo_comm *comm);
const char *adaflo_comm_last_error(const adaflo_comm *comm);
/* bit 2 d + side for the faces of the local brick that are shared with another rank */
unsigned    adaflo_comm_interface_faces(const adaflo_comm *comm);
/* src.update_ghost_values(): owners -> replicas; dst.compress(VectorOperation::add): afterwards
 * every replica holds the total (summed in a fixed order: replicas are bitwise identical) */
int         adaflo_comm_update_ghost_values(adaflo_comm *comm, double *vec_u, double *vec_p);
int         adaflo_comm_compress_add(adaflo_comm *comm, double *vec_u, double *vec_p);
/* NavierStokesMatrix::vmult on the global problem.  With the Q2/Q1 sweep kernel the two
 * exchanges run concurrently with the interior cells (three phases of adaflo_ns_vmult_phase,
 * HIP events between the engine stream and the communication stream, no host
 * synchronisation); src_ghosts_valid != 0 skips the ghost update of src (Krylov vectors that
 * came out of a previous distributed operation are consistent already).  src is not const:
 * its replicas are refreshed.                                                               */
int         adaflo_ns_vmult_distributed(adaflo_ctx *ctx, adaflo_comm *comm, double *dst_u, double *dst_p,
                                        double *src_u, double *src_p, int src_ghosts_valid);
/* get_matvec_statistics on the global problem (navier_stokes_matrix.cc:1194-1206: Utilities::MPI::min_max_avg of the
 * accumulated vmult time over the communicator of the solution vector + the number of applications; both counters are
 * reset).  The time of a rank is device time between two HIP events on its engine stream around each
 * adaflo_ns_vmult_distributed (exchanges included: the stream waits for them inside).  Collective over the ranks. */
typedef struct adaflo_min_max_avg
{
  double sum, min, max, avg; /* dealii::Utilities::MPI::MinMaxAvg */
  int    min_index, max_index;
} adaflo_min_max_avg;
int         adaflo_comm_matvec_statistics(adaflo_comm *comm, unsigned *count, adaflo_min_max_avg *stats);
/* Diagnostics of one rank (not collective): device seconds of the pieces of adaflo_ns_vmult_distributed accumulated since the
 * last call -- [0] ghost update of src (pack, messages, unpack), [1] cells at the inter-GPU interface + their seam sums,
 * [2] compress(add) of dst, all three on the auxiliary stream BESIDE [3] the interior cells, [4] the tail (remaining seam sums,
 * constrained rows, mean-value projection) -- and the number of operator applications.  Off by default (ten more event
 * records per application).  The reference times the operator as a whole only, navier_stokes_matrix.cc:1194-1206. */
int         adaflo_comm_set_phase_timing(adaflo_comm *comm, int enabled);
int         adaflo_comm_phase_statistics(adaflo_comm *comm, unsigned *count, double seconds[5]);
/* measurement aid (bench.py --through-comm): take the three-phase schedule -- packs, events, second stream,
 * three kernel launches, unpacks, constrained rows -- also with world = 1, where adaflo_ns_vmult_distributed
 * otherwise forwards to adaflo_ns_vmult.  Shows the fixed cost of the distributed path on one GPU.          */
int         adaflo_comm_force_phased_schedule(adaflo_comm *comm, int enabled);


/* ---- level-set operators (LevelSetOKZSolver*, source/level_set_okz_*.cc) ---- */
/* FE_Q_iso_Q1(ls_degree) on the same brick; block vectors with dim blocks (normal
 * vector field) are passed as ONE pointer to dim consecutive double[n_dofs_ls].
 * Quadrature-point arrays cross the boundary in canonical layout [cell][q][dim].  */
typedef struct
{
  double epsilon_used;        /* two_phase_base.cc:282-291 */
  double minimal_edge_length; /* util.h:97-119 */
  double time_step;           /* time_stepping.step_size() */
  double weight, weight_old, weight_old_old; /* TimeStepping weights (advection) */
  double epsilon;             /* parameters.epsilon (normal / curvature damping) */
} adaflo_ls_params;
int adaflo_ls_set_params(adaflo_ctx *ctx, const adaflo_ls_params *p);
/* preconditioner.get_vector(): diagonal used for constrained rows (device pointer),
 * e.g. level_set_okz_reinitialization.cc:227-230 */
int adaflo_ls_set_diagonal(adaflo_ctx *ctx, const double *diag);
/* evaluated_convection / evaluated_normal: the quadrature-point arrays the reference's rhs loops fill
 * (level_set_okz_advance_concentration.cc:389, level_set_okz_reinitialization.cc:167-172).  On the sweep kernels the
 * engine keeps the NODAL velocity / normal field of the last right-hand side instead and the operators evaluate it at
 * the Gauss points themselves; get_* materialises the array on demand, set_* makes the operators stream the given one. */
int adaflo_ls_set_evaluated_convection(adaflo_ctx *ctx, const double *u_q, int src_on_device);
int adaflo_ls_get_evaluated_convection(adaflo_ctx *ctx, double *u_q, int dst_on_device);
int adaflo_ls_set_evaluated_normal(adaflo_ctx *ctx, const double *n_q, int src_on_device);
int adaflo_ls_get_evaluated_normal(adaflo_ctx *ctx, double *n_q, int dst_on_device);
/* advance_concentration_vmult  level_set_okz_advance_concentration.cc:401-480 */
int adaflo_ls_advance_concentration_vmult(adaflo_ctx *ctx, double *dst, const double *src);
/* local_advance_concentration_rhs :288-397 (cell loop: adds into dst, stores the velocity at
 * the quadrature points); use_old_old = (scheme == bdf_2 && step_no > 1), :375-378 */
int adaflo_ls_advance_concentration_rhs(adaflo_ctx *ctx, double *dst, const double *solution,
                                        const double *solution_old, const double *solution_old_old,
                                        const double *vel_solution, int use_old_old);
/* `convection stabilization` of the advection operator (parameters.convection_stabilization,
 * default off).  Once enabled, adaflo_ls_advance_concentration_vmult adds the cell term
 * (grad w, nu_cell grad v) (level_set_okz_advance_concentration.cc:248-249) and the boundary term
 * -(w, n . nu_cell grad v) over the boundary faces that are not symmetry faces (:419-472; bit
 * 2 d + side of symmetry_faces), with the artificial viscosities the last stabilised rhs left
 * (public array of the reference: set / get below).  global_omega_diameter =
 * diameter_on_coarse_grid (util.h:70-100; the space diagonal of a box).                       */
int adaflo_ls_set_convection_stabilization(adaflo_ctx *ctx, int enabled, double global_omega_diameter,
                                           unsigned symmetry_faces);
int adaflo_ls_set_artificial_viscosities(adaflo_ctx *ctx, const double *nu_cell, int src_on_device);
int adaflo_ls_get_artificial_viscosities(adaflo_ctx *ctx, double *nu_cell, int dst_on_device);
/* get_maximal_velocity (:39-68): largest |u| on the iterated trapezoid points of every cell */
int adaflo_ls_max_velocity(adaflo_ctx *ctx, const double *vel_solution, double *max_velocity);
/* local_advance_concentration_rhs with the stabilisation (:344-369 artificial viscosity per cell from
 * the two old states: 0.03 max|u_old + u_old_old| h min(1, max residual / (global_max_velocity 2
 * global_omega_diameter)); :387-388 cell term) followed by the boundary part the reference's driver
 * adds (:569-617).  dst is NOT zeroed.                                                         */
int adaflo_ls_advance_concentration_rhs_stabilized(adaflo_ctx *ctx, double *dst, const double *solution,
                                                   const double *solution_old, const double *solution_old_old,
                                                   const double *vel_solution, const double *vel_solution_old,
                                                   const double *vel_solution_old_old, int use_old_old,
                                                   double old_step_size, double global_max_velocity);
/* dst += sign * boundary part of the stabilisation term of `vec` (building block of the two above) */
int adaflo_ls_stabilization_boundary_term(adaflo_ctx *ctx, double *dst, const double *vec, double sign);
/* reinitialization_vmult  level_set_okz_reinitialization.cc:193-231 */
int adaflo_ls_reinitialization_vmult(adaflo_ctx *ctx, double *dst, const double *src, int diffuse_only);
/* local_reinitialize_rhs :128-189 (adds into dst; writes evaluated_normal on the first step) */
int adaflo_ls_reinitialization_rhs(adaflo_ctx *ctx, double *dst, const double *solution,
                                   const double *normal_vector_field, int diffuse_only,
                                   int first_reinit_step);
/* compute_normal_vmult / local_compute_normal_rhs  level_set_okz_compute_normal.cc:160-183, :123-156 */
int adaflo_ls_compute_normal_vmult(adaflo_ctx *ctx, double *dst, const double *src);
int adaflo_ls_compute_normal_rhs(adaflo_ctx *ctx, double *dst, const double *level_set_solution);
/* The scalar "projection matrix" the production solves of the normal AND of the curvature use
 * (LevelSetOKZSolver::local_projection_matrix, source/level_set_okz.cc:262-312: mass + 4 max(eps_used /
 * eps, h / ls)^2 Laplace, assembled by the reference and applied per block; compute_curvature.cc:355
 * solves with it, the call with ComputeCurvatureMatrix above it is commented out).  Matrix-free here:
 * one scalar block of compute_normal_vmult. */
int adaflo_ls_projection_vmult(adaflo_ctx *ctx, double *dst, const double *src);
/* compute_curvature_vmult / local_compute_curvature_rhs  level_set_okz_compute_curvature.cc:263-304, :212-259 */
int adaflo_ls_compute_curvature_vmult(adaflo_ctx *ctx, double *dst, const double *src, int apply_diffusion);
int adaflo_ls_compute_curvature_rhs(adaflo_ctx *ctx, double *dst, const double *normal_vector_field);

/* LevelSetOKZSolver::compute_heaviside (source/level_set_okz.cc:479-540) and local_compute_force
 * (:317-413) -- SURVEY 8f rank 2, the producer of the Navier-Stokes operator's variable density /
 * viscosity arrays and of the surface-tension + gravity right-hand side.
 *   heaviside[node] = discrete_heaviside(2 epsilon / s * atanh-distance(level_set[node])) in the cells
 *     around the interface, 0 / 1 away from it (epsilon = parameters.epsilon, relative width)
 *   compute_force ADDS (v, surface_tension * kappa * grad H - gravity * rho * e_z) into the
 *     velocity block user_rhs_u (the reference zeroes navier_stokes.user_rhs before the cell loop)
 *     and, if density_diff or viscosity_diff is non-zero, overwrites the engine's density /
 *     viscosity stores (what adaflo_ns_set_coefficients would set) with
 *     density + density_diff * H(x_q), viscosity + viscosity_diff * H(x_q). */
typedef struct adaflo_force_params
{
  double surface_tension, gravity, density, density_diff, viscosity, viscosity_diff;
  int    interpolate_grad_onto_pressure; /* "grad pressure compatible" */
} adaflo_force_params;
int adaflo_ls_compute_heaviside(adaflo_ctx *ctx, double *heaviside, const double *level_set, double epsilon);
int adaflo_ls_compute_force(adaflo_ctx *ctx, double *user_rhs_u, const double *heaviside,
                            const double *curvature, const adaflo_force_params *params);
/* "curvature correction = 1" (source/level_set_okz_compute_curvature.cc:360-376): where
 * curvature[node] > 1e-4, replace it by 1 / (1 / curvature + distance / (dim - 1)), distance =
 * epsilon_used * log((1 + phi) / (1 - phi)) if 1 - phi^2 > 1e-2, else 0 (epsilon_used from
 * adaflo_ls_set_parameters). */
int adaflo_ls_curvature_correction(adaflo_ctx *ctx, double *curvature, const double *level_set);

/* --------------------------------------------------------------------------------------------
 * Krylov drivers with device-resident vectors (callers of the operators above)
 *   SolverCG / SolverBicgstab with ReductionControl and DiagonalPreconditioner as used at
 *   source/level_set_okz_reinitialization.cc:325-345      CG, ReductionControl(2000, 1e-50, 1e-6)
 *   source/level_set_okz_advance_concentration.cc:623-644 BiCGStab, ReductionControl(30, ., 1e-8)
 *   source/level_set_okz_compute_normal.cc:252-267        CG on the 3-block normal system
 *   source/level_set_okz_compute_curvature.cc:345-355     CG, rel. 1e-8
 *   source/navier_stokes_preconditioner.cc:743-773        CG on the pressure mass matrix
 * -------------------------------------------------------------------------------------------- */
typedef enum adaflo_operator
{
  ADAFLO_OP_LS_ADVANCE_CONCENTRATION   = 0, /* AdvanceConcentrationMatrix   advance_concentration.cc:484-499 */
  ADAFLO_OP_LS_REINITIALIZATION        = 1, /* ReinitializationMatrix(diffuse = false) reinitialization.cc:235-252 */
  ADAFLO_OP_LS_REINITIALIZATION_DIFFUSE = 2,
  ADAFLO_OP_LS_NORMAL                  = 3, /* ComputeNormalMatrix (3 scalar blocks) compute_normal.cc:187-203 */
  ADAFLO_OP_LS_CURVATURE               = 4, /* ComputeCurvatureMatrix        compute_curvature.cc:308-323 */
  ADAFLO_OP_NS_PRESSURE_MASS           = 5,
  ADAFLO_OP_NS_PRESSURE_POISSON        = 6,
  ADAFLO_OP_NS_VELOCITY                = 7, /* velocity_vmult (A block) */
  ADAFLO_OP_LS_PROJECTION              = 8  /* scalar projection matrix, adaflo_ls_projection_vmult */
} adaflo_operator;

typedef enum adaflo_solver
{
  ADAFLO_SOLVER_CG       = 0,
  ADAFLO_SOLVER_BICGSTAB = 1
} adaflo_solver;

/* ReductionControl(max_iterations, abs_tol, rel_tol): success when the l2 norm of the residual is
 * below abs_tol or below rel_tol times its initial value */
typedef struct adaflo_solver_control
{
  int    max_iterations;
  double abs_tol, rel_tol;
} adaflo_solver_control;

typedef struct adaflo_solver_result
{
  int    iterations; /* SolverControl::last_step() */
  int    converged;  /* 0: max_iterations reached or breakdown (deal.II throws NoConvergence) */
  double initial_residual, final_residual;
} adaflo_solver_result;

/* ---- deal.II-numbered vectors at the boundary ------------------------------------------------
 * The engine numbers the DoFs of its brick node-lexicographically (header comment of adaflo_brick_desc).
 * A LinearAlgebra::distributed::Vector<double> of the application stores them in deal.II's numbering,
 * locally owned entries first, ghost entries appended (source/navier_stokes.cc:79-83,
 * include/adaflo/block_matrix_extension.h:48-49 use begin() as one contiguous double[owned + ghost]).
 * With a DEVICE-resident index map  map[i] = position of engine DoF i in that array (-1: no counterpart,
 * e.g. a DoF deal.II eliminated) an adapter moves between the two numberings without a host permutation:
 *   adaflo_vector_gather   engine[i] = map[i] >= 0 ? dealii[map[i]] : 0            i < n
 *   adaflo_vector_scatter  dealii[map[i]] = (add ? dealii[map[i]] : 0) + engine[i]  for map[i] >= 0
 * One coalesced pass each (the engine side is contiguous).  The map must not name a deal.II entry twice
 * (a brick holds every node once); scatter with add = 1 into the owned + ghost array followed by the
 * application's compress(add) is the cell_loop convention (navier_stokes_matrix.cc:232-245), add = 0 copies
 * final values (after adaflo_ns_vmult_distributed, which has done the exchange itself).               */
int adaflo_vector_gather(adaflo_ctx *ctx, double *engine_vec, const double *dealii_vec, const int64_t *index_map,
                         int64_t n);
int adaflo_vector_scatter(adaflo_ctx *ctx, double *dealii_vec, const double *engine_vec, const int64_t *index_map,
                          int64_t n, int add);

/* vector algebra of the drivers on device vectors: x = value; x = a x + b y; (x, y) */
int adaflo_vector_fill(adaflo_ctx *ctx, double *x, double value, int64_t n);
int adaflo_vector_sadd(adaflo_ctx *ctx, double *x, double a, double b, const double *y, int64_t n);
int adaflo_vector_dot(adaflo_ctx *ctx, const double *x, const double *y, int64_t n, double *result);

/* initialize_mass_matrix_diagonal (include/adaflo/level_set_okz_preconditioner.h:35-76): diagonal of
 * the level-set mass matrix, sum_q phi_i(x_q)^2 JxW, the vector behind the DiagonalPreconditioner of
 * all level-set solves */
int adaflo_ls_mass_matrix_diagonal(adaflo_ctx *ctx, double *diagonal);

/* DiagonalPreconditioner::reinit (source/diagonal_preconditioner.cc:27-47):
 * inv[i] = |d[i]| > 1e-10 max|d| ? 1/d[i] : 1 ; device pointers */
int adaflo_invert_diagonal(adaflo_ctx *ctx, double *inverse_diagonal, const double *diagonal, int64_t n);

/* x <- approximate solution of op(x) = b, starting from the x passed in; all pointers are device
 * pointers of the operator's vector size (3 blocks for ADAFLO_OP_LS_NORMAL).  inverse_diagonal:
 * one block long (applied to every block) or NULL for no preconditioning. */
int adaflo_solve(adaflo_ctx *ctx, int op, int method, double *x, const double *b,
                 const double *inverse_diagonal, const adaflo_solver_control *control,
                 adaflo_solver_result *result);

/* Matrix-free replacement of the assembled matrices + ILU / AMG behind the inner solves
 * (navier_stokes_preconditioner.cc:636-666,:712-733; level_set_okz_compute_normal.cc:
 * 252-267): exact inverse of  c_mass M + c_lap K  (mass and Laplace matrices of the velocity space,
 * field 0, per component; of the pressure space, field 1; of the level-set space FE_Q_iso_Q1(s), field 2,
 * one scalar block: the projection matrix M + 4 max(eps_used / eps, h / s)^2 K of the normal and
 * curvature solves, level_set_okz.cc:262-312) on the brick by fast diagonalisation -- six dense 1D
 * transforms per scalar field.  dst = src on constrained rows; pseudo-inverse (null mode dropped) if
 * the operator is singular.  dst == src allowed.                                                  */
int adaflo_fdm_apply(adaflo_ctx *ctx, int field, double *dst, const double *src, double c_mass, double c_lap);
/* dst = [(c_mass M + c_lap K)^-1 + (c_mass2 M + c_lap2 K)^-1] src in ONE application (both inverses are diagonal in the
 * same modes): the pressure mass + pressure Poisson inverses of the Schur complement approximation
 * (navier_stokes_preconditioner.cc:712-733) for constant coefficients.  Unconstrained fields only. */
int adaflo_fdm_apply_sum(adaflo_ctx *ctx, int field, double *dst, const double *src, double c_mass, double c_lap,
                         double c_mass2, double c_lap2);
/* dst = (projection matrix)^-1 rhs per scalar block, exactly: what the reference's normal and curvature
 * solves approximate with CG + ILU on the assembled matrix (level_set_okz_compute_normal.cc:252-267,
 * level_set_okz_compute_curvature.cc:345-355).  n_blocks = 3 for the normal vector field.  Needs an
 * unconstrained level-set space (ADAFLO_EUNSUPPORTED otherwise).                                    */
int adaflo_ls_projection_solve(adaflo_ctx *ctx, double *dst, const double *rhs, int n_blocks);
/* inner solves of adaflo_ns_preconditioner_vmult: 0 = Krylov solves with the pointwise Jacobi
 * preconditioner, 1 (default, constant coefficients) = fast diagonalisation: the velocity block's
 * BiCGStab is right-preconditioned with the inverse of its mass + vector-Laplace part, the pressure
 * mass and Poisson solves are exact                                                              */
int adaflo_ns_preconditioner_set_inner(adaflo_ctx *ctx, int mode);
/* FlowParameters::iterations_before_inner_solvers ("lin its before inner solvers", default 50;
 * navier_stokes.cc:571-617): adaflo_ns_solve_system first runs the cheap solver whose preconditioner
 * applies the approximate inverses once (do_inner_solves = false, navier_stokes_preconditioner.cc:605-635,
 * :719-720) for that many iterations and only then, from the iterate reached, the solver with inner
 * Krylov solves.  The cheap stage is taken with the fast-diagonalisation inverses (mode 1, constant
 * coefficients); with Jacobi diagonals the solver with inner solves runs from the start.  0 = inner
 * solves at once.                                                                                 */
int adaflo_ns_set_iterations_before_inner_solvers(adaflo_ctx *ctx, int iterations);
/* Variable coefficients (Jacobi diagonals, mode 0 or two-phase flow): the cheap stage applies, as approximate
 * inverse of the velocity block, a BiCGStab solve cut off after `iterations` iterations (0 = no cheap stage:
 * inner solves to their tolerance from the start, the behaviour of rounds 1 and 2).                        */
int adaflo_ns_preconditioner_set_cheap_velocity_iterations(adaflo_ctx *ctx, int iterations);
/* number of velocity-block solves and their BiCGStab iterations since the last query */
int adaflo_ns_preconditioner_statistics(adaflo_ctx *ctx, int64_t *velocity_solves, int64_t *velocity_iterations);
/* NavierStokes::solve_system (source/navier_stokes.cc:561-653): FGMRES(restart) on adaflo_ns_vmult,
 * right-preconditioned by NavierStokesPreconditioner::vmult with inner solves
 * (source/navier_stokes_preconditioner.cc:595-737: velocity block BiCGStab, divergence, pressure
 * mass CG + pressure Poisson CG).  The reference's ILU / AMG inner preconditioners (Trilinos) are
 * replaced by Jacobi preconditioners built from the probed operator diagonals.
 *   adaflo_ns_preconditioner_setup  build_preconditioner (:747-779): fix_linearization_point + diagonals
 *   control: SolverControl(max_iterations, abs_tol) -- rel_tol is ignored; restart = 50 in the reference
 *   update_* are overwritten (solution_update = 0 first, :567). */
int adaflo_ns_preconditioner_setup(adaflo_ctx *ctx);
int adaflo_ns_preconditioner_vmult(adaflo_ctx *ctx, double *dst_u, double *dst_p, const double *src_u,
                                   const double *src_p);
int adaflo_ns_solve_system(adaflo_ctx *ctx, double *update_u, double *update_p, const double *rhs_u,
                           const double *rhs_p, const adaflo_solver_control *control, int restart,
                           adaflo_solver_result *result);

/* dominant cell-kernel statistics (device time between HIP events recorded on
 * the context's stream around the cell kernel only); used by bench.py for the
 * roofline figure.  Resets like adaflo_ns_get_matvec_statistics.                */
int adaflo_get_kernel_statistics(adaflo_ctx *ctx, unsigned *count, double *seconds);
/* switch the event timers off/on (default on) */
int adaflo_set_timing(adaflo_ctx *ctx, int enabled);
/* tuning: number of cell layers one workgroup of the Q2/Q1 kernel sweeps (0 = heuristic) */
int adaflo_set_q2_chunk(adaflo_ctx *ctx, int layers);
/* tuning: number of cells one workgroup of the Q3..Q5 x-marching kernel marches through (0 = heuristic) */
int adaflo_set_hox_chunk(adaflo_ctx *ctx, int cells);
/* tuning: skew padding (units of 16 B) between the per-(tile,layer) blocks of the streamed state */
int adaflo_set_q2_state_pad(adaflo_ctx *ctx, int pad_16B);

/* select the implementation of the operator applications: 0 = generic per-cell kernels (any
 * degree), 1 = auto (default: Q2/Q1 sweep kernel, Q3..Q5 x-marching kernel ns_hox.hip, structured Q1
 * sweep kernel for the level set and the Q1 pressure operators, generic elsewhere), 2 = as 1 but the
 * round-2 z-sweep kernel (ns_ho.hip) for Q3..Q5, kept for comparison; 3 = as 1 with the plane-per-lane kernel
 * (ns_hop.hip) for the constant-coefficient Q4/Q3 vmult / velocity_vmult (round 5, slower than 1: kept for comparison);
 * 4 = as 1 but the Q2/Q1 Newton vmult always STREAMS the quadrature-point state (rounds 1-4).  Since round 5 variant 1
 * recomputes (u_lin, grad u_lin) from the nodal solution the last adaflo_ns_residual of this context was computed at --
 * the state is that interpolation, navier_stokes_matrix.cc:778-816 (round 6: also the (u, div u) state of the Picard-type,
 * semi-implicit and projection schemes, the nodal field being the extrapolated old velocity where the scheme linearises about
 * it); variable density / viscosity / damping are read from the arrays of adaflo_ns_set_coefficients next to it -- and streams
 * only a state that was set through adaflo_ns_set_linearization (no nodal field behind it).
 * All variants are bitwise reproducible. */
int adaflo_set_kernel_variant(adaflo_ctx *ctx, int variant);
/* Lazy quadrature-point state of the Q2/Q1 Newton residual (default 1, round 6).  The recompute-state vmult of variant 1
 * never reads the state adaflo_ns_residual lays out at the quadrature points (96 B per point: 5.4 GB at 128^3, 43.5 GB at
 * 256^3), so the residual does not write it (its state stores are issued with an empty mask); the engine lays it out on demand -- adaflo_ns_get_linearization, the generic
 * and the streaming kernels, adaflo_ns_fix_linearization_point, a change of scheme -- by one more pass of the residual kernel
 * over the nodal field it kept.  Results are bitwise the same either way (NavierStokesMatrix::residual fills
 * linearized_velocities in the cell loop, navier_stokes_matrix.cc:778-799: here that is deferred, not dropped).  0 = write
 * the state with every residual (rounds 1-5). */
int adaflo_set_q2_lazy_state(adaflo_ctx *ctx, int lazy);
/* 1 if the library contains the kernels of `variant`.  Variants 2 and 3 -- the superseded Q3..Q5 kernels ns_ho.hip (rounds 2-3)
 * and ns_hop.hip (round 5, measured slower) -- are built only with ADAFLO_BUILD_VARIANTS=1 (adaflo_amd/build.py);
 * adaflo_set_kernel_variant returns ADAFLO_EUNSUPPORTED for them otherwise. */
int adaflo_has_kernel_variant(int variant);

#ifdef __cplusplus
}
#endif
#endif
