// ctx.hpp -- the engine context behind the C ABI (include/adaflo_hip.h).
//
// Plays the role of NavierStokesMatrix<dim>'s private state
// (include/adaflo/navier_stokes_matrix.h:253-282): pointers to the mesh
// description (here: a structured brick instead of MatrixFree<dim>), the
// operator parameters, and the quadrature-point arrays it owns.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/adaflo_hip.h"
#include "basis.hpp"
#include "fe_kernels.hpp"

namespace adaflo_hip
{
  // device mirror of adaflo_ns_params with the derived flags local_operation
  // computes at source/navier_stokes_matrix.cc:621-653
  struct NSDev
  {
    int    physical_type, linearization;
    double beta, tau_grad_div, density, viscosity, damping, density_diff;
    double weight, weight_old, weight_old_old, tau1, extrap_old, extrap_old_old;
  };

  // scalars of the level-set operators (LevelSetOKZSolver*Parameter structs and the
  // TimeStepping weights they read)
  struct LSDev
  {
    double epsilon_used, minimal_edge_length, time_step, weight, weight_old, weight_old_old, epsilon;
  };

  struct DeviceBuffer
  {
    double *p     = nullptr;
    size_t  count = 0;
  };

  // accumulates the device time between pairs of events recorded on a stream
  struct EventTimer
  {
    std::vector<hipEvent_t> pool;
    size_t                  used    = 0;
    unsigned                count   = 0;
    double                  seconds = 0.;

    void fold()
    {
      for (size_t i = 0; i + 1 < used; i += 2)
        {
          float ms = 0.f;
          if (hipEventElapsedTime(&ms, pool[i], pool[i + 1]) == hipSuccess)
            seconds += 1e-3 * ms;
        }
      used = 0;
    }
    // returns the stop event to record after the work (nullptr on failure)
    hipEvent_t start(hipStream_t stream)
    {
      if (used + 2 > pool.size())
        {
          if (pool.size() >= 8192)
            {
              // (the pairs may have been recorded on several streams -- the two-stream schedule of comm.hip shares this
              // timer --: wait for the stop events themselves, a synchronisation of the calling stream alone left the
              // other stream's pairs "not ready" and fold() dropped them)
              for (size_t i = 1; i < used; i += 2)
                (void)hipEventSynchronize(pool[i]);
              fold();
            }
          else
            for (int i = 0; i < 2; ++i)
              {
                hipEvent_t e;
                // (timing only, read after a stream synchronisation: no system-scope fence -- a default event writes
                // back and invalidates the caches at every record, which costs the NEXT kernel)
                if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess)
                  return nullptr;
                pool.push_back(e);
              }
        }
      hipEvent_t a = pool[used++], b = pool[used++];
      (void)hipEventRecord(a, stream);
      return b;
    }
    void destroy()
    {
      for (hipEvent_t e : pool)
        (void)hipEventDestroy(e);
      pool.clear();
    }
  };
} // namespace adaflo_hip

struct adaflo_ctx
{
  adaflo_brick_desc desc{};
  hipStream_t       stream     = nullptr;
  bool              own_stream = false;
  int               variant    = 1;

  bool    flat = false; // dim <= 2: flat third direction (capi.hip: ctx_create_impl)
  bool    flat_y = false; // dim = 1: the second direction is flat as well
  int     k = 0, s = 0;
  int64_t n_cells = 0, n_nodes_u = 0, n_nodes_p = 0, n_nodes_ls = 0;
  int     nq_u = 0; // (k+1)^3

  adaflo_hip::BrickDev brick{};
  // indexed context (adaflo_ctx_create_indexed): device copies of the adapter's tables (brick.idx_* / flag_* / cell_h point
  // into them) and the colour ranges; only the generic kernels of the Navier-Stokes block run on it
  bool              indexed = false;
  std::vector<long> idx_colour_off;
  int              *d_idx_u = nullptr, *d_idx_p = nullptr;
  unsigned char    *d_flag_u = nullptr, *d_flag_p = nullptr;
  double           *d_cell_h = nullptr;
  long             *d_hang_ptr_u = nullptr, *d_hang_ptr_p = nullptr; // hanging-node constraints (CSR), optional
  int              *d_hang_master_u = nullptr, *d_hang_master_p = nullptr;
  double           *d_hang_weight_u = nullptr, *d_hang_weight_p = nullptr;
  adaflo_hip::NSDev    ns{};
  bool                 ns_params_set = false;

  // 1D tables on the device, one packed buffer:
  //   [S_u D_u S_p D_p w] for quad_index_u (n = k+1 points)
  //   [S_pp D_pp w_pp]    for quad_index_p (n = k points), pressure only
  double *d_tab_u = nullptr, *d_tab_pp = nullptr;

  // quadrature-point state, generic layout [cell][comp][q]
  adaflo_hip::DeviceBuffer lin, rho, mu, damp;
  adaflo_hip::DeviceBuffer lin_prec, rho_prec, mu_prec, damp_prec; // fix_linearization_point

  // specialised (Q2/Q1 sweep kernel) copy of the linearisation, see ns_q2.hip
  adaflo_hip::DeviceBuffer lin_q2, lin_q2_prec;
  adaflo_hip::DeviceBuffer q2_slab_u, q2_zslab_u, q2_slab_p, q2_zslab_p; // seam partial sums
  adaflo_hip::DeviceBuffer ho_tab;                                        // 1D tables of the Q3..Q5 sweep kernel
  // x-marching Q3..Q5 kernel (ns_hox.hip): streaming copies of the (frozen) linearisation state and the generation
  // of the generic copy they were converted from, seam slabs, 1D tables, workgroup list of the phased schedule
  adaflo_hip::DeviceBuffer hox_lin, hox_lin_prec, hox_slab_u, hox_xslab_u, hox_slab_p, hox_xslab_p, hox_tab;
  // coefficient stream of the variable-coefficient residual of the x-marching kernel (round 6) and the counter of changes of
  // rho / mu / damping it is current for
  adaflo_hip::DeviceBuffer hox_coef;
  unsigned long            coef_gen = 1, hox_coef_gen = 0;
  unsigned long            lin_gen = 1, lin_prec_gen = 1, hox_lin_gen = 0, hox_lin_prec_gen = 0;
  int                      hox_lin_mode = -1, hox_lin_prec_mode = -1;
  bool                     hox_lin_varco = false, hox_lin_prec_varco = false; // the streaming copy carries rho / mu / damping
  bool                     hox_lin_primary = false; // hox_lin (written by the residual mode) is THE state, `lin` is stale
  bool                     hox_lin_prec_primary = false; // ... and its frozen copy exists in the streaming layout only
  std::vector<double>      hox_tab_host; // what hox_tab holds
  int                      hox_lx = 0; // x-chunk length override (0 = heuristic)
  int                     *hox_wg_list = nullptr;
  long                     hox_wg_key[4] = {0, 0, 0, 0};
  int                      hox_wg_counts[3] = {0, 0, 0};
  // plane-per-lane Q4/Q3 kernel (ns_hop.hip): its streaming copies of the state (keyed on lin_gen / lin_prec_gen like the
  // others), its table and tile list; `hop`: use it where it applies (kernel variant 3 = variant 1 + this)
  adaflo_hip::DeviceBuffer hop_lin, hop_lin_prec, hop_tab;
  unsigned long            hop_lin_gen = 0, hop_lin_prec_gen = 0;
  int                      hop_lin_mode = -1, hop_lin_prec_mode = -1;
  std::vector<double>      hop_tab_host;
  int                     *hop_wg_list = nullptr;
  long                     hop_wg_key[4] = {0, 0, 0, 0};
  int                      hop_wg_counts[3] = {0, 0, 0};
  bool                     hop = false;
  int                     *q2_wg_list = nullptr;     // [interface | interior A | interior B] workgroups
  long                     q2_wg_key[4] = {0, 0, 0, 0};
  int                      q2_wg_counts[3] = {0, 0, 0};
  bool                     lin_q2_valid = false, lin_q2_varco = false, lin_q2_prec_varco = false;
  // the sweep-kernel residual writes the state in the streaming layout only; the generic copies
  // (lin, lin_prec) are then stale until somebody asks for them (ensure_lin_generic, capi.hip)
  bool                     lin_generic_valid = true, lin_prec_generic_valid = true;
  int                      lin_q2_prec_mode = -1;
  adaflo_hip::DeviceBuffer res_sum_u, res_sum_p, res_old, res_ext; // work vectors of the sweep-kernel residual (res_ext: extrapolated velocity)
  int                      lin_q2_mode  = -1;
  int                      q2_lz        = 0; // z-chunk length override (0 = heuristic)
  // recompute-state mode of the Q2/Q1 kernel (on by default, kernel variant 4 switches it off): nodal copy of the
  // solution the last residual was evaluated at; valid only together with lin_q2_valid
  // lin_serial counts every change of the linearisation state (set_linearization, any residual, a change of scheme); the
  // nodal copy is current while lin_nodal_serial equals it -- whatever layouts the state has been converted to since
  bool                     q2_recompute = true, lin_nodal_prec_valid = false;
  // lazy state (round 6): a residual whose consumer is the recompute-state vmult does not WRITE the quadrature-point
  // state (5.4 GB at 128^3): lin_q2_deferred = "the state is the interpolation of lin_nodal (current by invariant) and has
  // not been laid out"; q2_materialize_state runs the residual kernel once more for the state alone when somebody needs it
  // (get_linearization, generic / streaming kernels, fix_linearization_point, a change of scheme).  Meanwhile the kernel
  // issues its state stores with an empty EXEC mask (Q2Args::state_out == nullptr)
  bool                     q2_lazy_state = true, lin_q2_deferred = false;
  unsigned long            lin_serial = 1, lin_nodal_serial = 0;
  adaflo_hip::DeviceBuffer lin_nodal, lin_nodal_prec; // (..._prec: frozen by fix_linearization_point, velocity_vmult)
  int                      q2_state_pad = 0; // skew padding (double2) per (tile, layer) state block

  // level-set operators
  adaflo_hip::LSDev        ls{};
  double                  *d_tab_ls = nullptr;  // [S D w] of FE_Q_iso_Q1(s) at QIterated(QGauss(2),s), then S of FE_Q(k)
  double                  *d_ls_diag = nullptr; // preconditioner.get_vector() for constrained rows
  double                  *d_tab_force = nullptr; // 1D tables of local_compute_force (ls_force.hip)
  adaflo_hip::DeviceBuffer ls_convection, ls_normal; // evaluated_convection / evaluated_normal [cell][3][q]
  // `convection stabilization` of the advection operator (advance_concentration.cc:344-369)
  bool                     ls_stab = false;
  double                   ls_omega_diameter = 0.;
  unsigned                 ls_symmetry = 0;
  adaflo_hip::DeviceBuffer ls_art_visc, ls_stab_vel_sum, ls_stab_ls_sum; // artificial_viscosities [cell], nodal sums
  double                  *d_tab_maxvel = nullptr;                      // FE_Q(k) at the iterated trapezoid points
  // structured Q1 sweep kernel (q1_sweep.hip): streaming copies of the two arrays, seam partial sums
  adaflo_hip::DeviceBuffer q1_convection, q1_normal, q1_slab, q1_zslab;
  bool                     q1_convection_valid = false, q1_normal_valid = false;
  // copy of the nodal normal field of the last first-step reinitialisation rhs: the reinitialisation vmult recomputes
  // the unit normal at the Gauss points from it instead of streaming the 192 B per sub-cell (q1_sweep.hip)
  adaflo_hip::DeviceBuffer q1_normal_nodal;
  bool                     q1_normal_nodal_valid = false;
  // the advection right-hand side on the sweep structure keeps the nodal velocity it was given instead of writing
  // evaluated_convection; the advection operator evaluates the velocity from it (Q1_ADVECT_NODAL), the quadrature-point
  // array is materialised only when somebody asks for it
  adaflo_hip::DeviceBuffer q1_velocity_nodal;
  bool                     q1_convection_nodal_valid = false;
  // the sweep right-hand sides write the quadrature-point arrays in sweep layout only; the generic
  // [cell][3][q] copies are re-created on demand (adaflo_ls_get_evaluated_*, generic kernels)
  bool                     ls_convection_generic_valid = false, ls_normal_generic_valid = false;
  adaflo_hip::DeviceBuffer q1_poisson_coef;   // 1 / (weight rho) per point, lane layout (two-phase pressure Poisson)
  const double            *q1_poisson_src = nullptr; // which density array it was built from ...
  double                   q1_poisson_weight = 0.;   // ... and with which time-step weight

  // block preconditioner of the coupled system (krylov.hip): inverse diagonals of the velocity
  // block, the pressure mass and the pressure Poisson operator, work vectors
  adaflo_hip::DeviceBuffer pc_inv_u, pc_inv_pm, pc_inv_pl, pc_ones_p, pc_tmp_u, pc_tmp_p, pc_tmp_p2, pc_work, kr_work, kr_basis, kr_scalars, pc_tridiag;
  bool                     pc_ready = false;
  bool                     pc_built_fdm = false; // what adaflo_ns_preconditioner_setup built: fast-diagonalisation inverses or Jacobi diagonals
  void                    *fdm = nullptr;      // fast-diagonalisation data of the inner solves (fdm.hip)
  long                     pc_velocity_iterations = 0, pc_velocity_solves = 0; // statistics of the velocity-block solves
  int                      pc_inner = 1;       // 0: Jacobi-preconditioned inner Krylov solves, 1: fast diagonalisation
  int                      pc_its_before_inner = 50; // parameters.iterations_before_inner_solvers (0: inner solves at once)
  bool                     pc_simple = false;  // current stage of adaflo_ns_solve_system: do_inner_solves == false
  int                      pc_simple_velocity_its = 0; // Jacobi mode: BiCGStab iterations of the velocity block in the cheap stage (0: no cheap stage)
  bool                     pc_poisson_fdm = true;      // Jacobi mode: pressure Poisson CG preconditioned by the constant-coefficient inverse

  // pressure constant mode (mode 0) data, source/navier_stokes_matrix.cc:117-168
  double *d_p_weights = nullptr, *d_p_modes = nullptr;
  double  inv_p_weight = 0.;
  double *d_scratch    = nullptr; // reduction scratch (partials + result)
  double *h_result = nullptr, *h_result_dev = nullptr; // pinned host copy of reduction results + its device address
  double *gs_dev = nullptr, *gs_host = nullptr;        // Gram-Schmidt coefficients of one FGMRES iteration (device, pinned host)
  // operator kernels that can leave the partial sums of src . dst on the way (the stencil kernels):
  // request (pointer + capacity in pairs) set by the CG driver, answer (pairs written, 0 = not done)
  double *fused_dot_partial = nullptr;
  long    fused_dot_capacity = 0;
  int     fused_dot_count    = 0;
  size_t  scratch_count = 0;

  // matvec statistics (get_matvec_statistics) and dominant-kernel statistics
  bool                     timing = true;
  adaflo_hip::EventTimer   matvec_timer, kernel_timer;

  std::string last_error;
};
