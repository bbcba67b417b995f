// adaflo_hip.hpp -- header-only C++ host side of the drop-in boundary: the operator classes of the reference with their
// own method names and argument order, templated on the application's vector types, over the C ABI of adaflo_hip.h.
//
//   adaflo::hip::NavierStokesMatrix<VectorType, BlockVectorType>   include/adaflo/navier_stokes_matrix.h:125-196
//       vmult, residual, divergence_vmult_add, velocity_vmult, pressure_poisson_vmult, pressure_mass_vmult,
//       pressure_convdiff_vmult, apply_pressure_average_projection, fix_linearization_point, get_matvec_statistics
//   adaflo::hip::NavierStokesPreconditioner<BlockVectorType>        source/navier_stokes_preconditioner.cc:595-737 (vmult)
//   adaflo::hip::AdvanceConcentrationMatrix<VectorType>             source/level_set_okz_advance_concentration.cc:484-499
//   adaflo::hip::ReinitializationMatrix<VectorType>                 source/level_set_okz_reinitialization.cc:235-252
//   adaflo::hip::ComputeNormalMatrix<BlockVectorType>               source/level_set_okz_compute_normal.cc:187-203
//   adaflo::hip::ComputeCurvatureMatrix<VectorType>                 source/level_set_okz_compute_curvature.cc:308-323
//
// Every class has the `void vmult(dst, src) const` shape deal.II's Krylov solvers take (SolverFGMRES / SolverCG /
// SolverBicgstab::solve(matrix, x, b, preconditioner), source/navier_stokes.cc:593-631), so the existing solver calls
// of an application compile unchanged with these classes in place of the reference's.
//
// Vector types.  A VectorType is anything whose device payload is reachable through
//     double *get_values();  const double *get_values() const;
// -- LinearAlgebra::distributed::Vector<double, MemorySpace::Default> has exactly this member (one contiguous array,
// owned entries first, ghosts appended: include/adaflo/block_matrix_extension.h:48-49) --, a BlockVectorType anything
// with `block(i)` returning such a vector (LinearAlgebra::distributed::BlockVector).  Other containers: overload
// adaflo::hip::device_values() for them.  The arrays are in the ENGINE's numbering (node-lexicographic over the brick,
// header comment of adaflo_brick_desc); an application in deal.II's numbering goes through adaflo_vector_gather /
// adaflo_vector_scatter with a device-resident index map first.
//
// Plain C++17, no HIP header: host-only translation units compile it with g++ and link libadaflo_hip.so.
#ifndef ADAFLO_HIP_HPP
#define ADAFLO_HIP_HPP

#include "adaflo_hip.h"

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>

namespace adaflo
{
  namespace hip
  {
    // deal.II's operators throw (ExcNotInitialized, ExcNotImplemented, ...); so do these
    class Error : public std::runtime_error
    {
    public:
      Error(const int code, const std::string &what)
        : std::runtime_error(what)
        , code(code)
      {}
      const int code;
    };

    inline void check(adaflo_ctx *ctx, const int code, const char *what)
    {
      if (code != ADAFLO_OK)
        throw Error(code, std::string(what) + ": " + adaflo_last_error(ctx));
    }

    template <typename VectorType>
    inline double *device_values(VectorType &v)
    {
      return v.get_values();
    }
    template <typename VectorType>
    inline const double *device_values(const VectorType &v)
    {
      return v.get_values();
    }

    // NavierStokesMatrix<dim> (dim = 3: the brick descriptor of the context carries the mesh, the degrees and the
    // constraints that the reference pulls out of MatrixFree<dim> in initialize(), navier_stokes_matrix.cc:85-168).
    // The context is NOT owned: create it with adaflo_ctx_create, push FlowParameters / TimeStepping scalars with
    // adaflo_ns_set_params whenever the time step changes (navier_stokes_matrix.cc:621-653 reads them per call).
    template <typename VectorType, typename BlockVectorType>
    class NavierStokesMatrix
    {
    public:
      // solution_old / solution_old_old: the references the reference's constructor takes
      // (navier_stokes_matrix.h:57-64); only residual() reads them.  comm: the engine's communicator when the mesh is
      // partitioned over several GPUs (vmult then is adaflo_ns_vmult_distributed), nullptr on one GPU.
      NavierStokesMatrix(adaflo_ctx *ctx, const BlockVectorType &solution_old, const BlockVectorType &solution_old_old,
                         adaflo_comm *comm = nullptr)
        : ctx(ctx)
        , comm(comm)
        , solution_old(&solution_old)
        , solution_old_old(&solution_old_old)
      {}
      // an operator that is never asked for a residual (preconditioner set-up, tests)
      explicit NavierStokesMatrix(adaflo_ctx *ctx, adaflo_comm *comm = nullptr)
        : ctx(ctx)
        , comm(comm)
        , solution_old(nullptr)
        , solution_old_old(nullptr)
      {}

      // :221-262
      void vmult(BlockVectorType &dst, const BlockVectorType &src) const
      {
        if (comm)
          {
            // (the replicas of the interface DoFs of src are refreshed: update_ghost_values, :232)
            BlockVectorType &s = const_cast<BlockVectorType &>(src);
            check(ctx,
                  adaflo_ns_vmult_distributed(ctx, comm, device_values(dst.block(0)), device_values(dst.block(1)),
                                              device_values(s.block(0)), device_values(s.block(1)), 0),
                  "NavierStokesMatrix::vmult");
          }
        else
          check(ctx,
                adaflo_ns_vmult(ctx, device_values(dst.block(0)), device_values(dst.block(1)), device_values(src.block(0)),
                                device_values(src.block(1))),
                "NavierStokesMatrix::vmult");
      }

      // :266-293; user_rhs = the right-hand side the application assembled (body force, boundary values)
      void residual(BlockVectorType &residual_vector, const BlockVectorType &src, const BlockVectorType &user_rhs) const
      {
        if (!solution_old || !solution_old_old)
          throw Error(ADAFLO_ENOTINIT, "NavierStokesMatrix::residual: constructed without the old solutions");
        check(ctx,
              adaflo_ns_residual(ctx, device_values(residual_vector.block(0)), device_values(residual_vector.block(1)),
                                 device_values(src.block(0)), device_values(src.block(1)), device_values(user_rhs.block(0)),
                                 device_values(user_rhs.block(1)), device_values(solution_old->block(0)),
                                 device_values(solution_old_old->block(0))),
              "NavierStokesMatrix::residual");
      }

      // :300-332
      void divergence_vmult_add(VectorType &dst, const VectorType &src, const bool weight_by_viscosity = false) const
      {
        check(ctx, adaflo_ns_divergence_vmult_add(ctx, device_values(dst), device_values(src), weight_by_viscosity ? 1 : 0),
              "NavierStokesMatrix::divergence_vmult_add");
      }
      // :337-382
      void velocity_vmult(VectorType &dst, const VectorType &src) const
      {
        check(ctx, adaflo_ns_velocity_vmult(ctx, device_values(dst), device_values(src)), "NavierStokesMatrix::velocity_vmult");
      }
      // :386-417
      void pressure_poisson_vmult(VectorType &dst, const VectorType &src) const
      {
        check(ctx, adaflo_ns_pressure_poisson_vmult(ctx, device_values(dst), device_values(src)),
              "NavierStokesMatrix::pressure_poisson_vmult");
      }
      // :421-455
      void pressure_mass_vmult(VectorType &dst, const VectorType &src) const
      {
        check(ctx, adaflo_ns_pressure_mass_vmult(ctx, device_values(dst), device_values(src)),
              "NavierStokesMatrix::pressure_mass_vmult");
      }
      // :459-483
      void pressure_convdiff_vmult(VectorType &dst, const VectorType &src) const
      {
        check(ctx, adaflo_ns_pressure_convdiff_vmult(ctx, device_values(dst), device_values(src)),
              "NavierStokesMatrix::pressure_convdiff_vmult");
      }
      // :191-205
      void apply_pressure_average_projection(VectorType &vector) const
      {
        check(ctx, adaflo_ns_apply_pressure_average_projection(ctx, device_values(vector)),
              "NavierStokesMatrix::apply_pressure_average_projection");
      }
      // :1144-1152
      void fix_linearization_point() const
      {
        check(ctx, adaflo_ns_fix_linearization_point(ctx), "NavierStokesMatrix::fix_linearization_point");
      }
      // :1194-1206: (min / max / avg of the accumulated vmult seconds over the ranks, number of applications); the
      // counters are reset.  Collective when a communicator is attached.
      std::pair<adaflo_min_max_avg, unsigned int> get_matvec_statistics() const
      {
        std::pair<adaflo_min_max_avg, unsigned int> r{};
        if (comm)
          {
            if (const int code = adaflo_comm_matvec_statistics(comm, &r.second, &r.first))
              throw Error(code, std::string("NavierStokesMatrix::get_matvec_statistics: ") + adaflo_comm_last_error(comm));
          }
        else
          {
            double seconds = 0.;
            check(ctx, adaflo_ns_get_matvec_statistics(ctx, &r.second, &seconds), "NavierStokesMatrix::get_matvec_statistics");
            r.first.sum = r.first.min = r.first.max = r.first.avg = seconds;
            r.first.min_index = r.first.max_index = 0;
          }
        return r;
      }

      std::int64_t n_dofs_u() const { return adaflo_n_dofs_u(ctx); }
      std::int64_t n_dofs_p() const { return adaflo_n_dofs_p(ctx); }
      adaflo_ctx  *context() const { return ctx; }

    private:
      adaflo_ctx            *ctx;
      adaflo_comm           *comm;
      const BlockVectorType *solution_old, *solution_old_old;
    };

    // NavierStokesPreconditioner::vmult with inner solves (navier_stokes_preconditioner.cc:595-737); build it with
    // adaflo_ns_preconditioner_setup (= build_preconditioner, :747-779) after every change of the linearisation point
    template <typename BlockVectorType>
    class NavierStokesPreconditioner
    {
    public:
      explicit NavierStokesPreconditioner(adaflo_ctx *ctx)
        : ctx(ctx)
      {}
      void vmult(BlockVectorType &dst, const BlockVectorType &src) const
      {
        check(ctx,
              adaflo_ns_preconditioner_vmult(ctx, device_values(dst.block(0)), device_values(dst.block(1)),
                                             device_values(src.block(0)), device_values(src.block(1))),
              "NavierStokesPreconditioner::vmult");
      }

    private:
      adaflo_ctx *ctx;
    };

    // ---- the four level-set operators: the wrapper structs the reference hands to SolverCG / SolverBicgstab ----------
    // (parameters: adaflo_ls_set_params; diagonal of the constrained rows: adaflo_ls_set_diagonal; the quadrature-point
    // data come from the right-hand-side calls of the same context, as in the reference)
    template <typename VectorType>
    struct AdvanceConcentrationMatrix
    {
      explicit AdvanceConcentrationMatrix(adaflo_ctx *problem)
        : problem(problem)
      {}
      void vmult(VectorType &dst, const VectorType &src) const
      {
        check(problem, adaflo_ls_advance_concentration_vmult(problem, device_values(dst), device_values(src)),
              "AdvanceConcentrationMatrix::vmult");
      }
      adaflo_ctx *problem;
    };

    template <typename VectorType>
    struct ReinitializationMatrix
    {
      ReinitializationMatrix(adaflo_ctx *problem, const bool diffuse_only)
        : problem(problem)
        , diffuse_only(diffuse_only)
      {}
      void vmult(VectorType &dst, const VectorType &src) const
      {
        check(problem, adaflo_ls_reinitialization_vmult(problem, device_values(dst), device_values(src), diffuse_only ? 1 : 0),
              "ReinitializationMatrix::vmult");
      }
      adaflo_ctx *problem;
      const bool  diffuse_only;
    };

    // dim = 3 blocks.  The C ABI takes the block vector as ONE array of 3 x n_dofs_ls doubles; blocks that already sit
    // back to back in memory are passed through, others are staged through a scratch array of the wrapper
    template <typename BlockVectorType>
    struct ComputeNormalMatrix
    {
      explicit ComputeNormalMatrix(adaflo_ctx *problem)
        : problem(problem)
        , n(adaflo_n_dofs_ls(problem))
      {}
      ComputeNormalMatrix(const ComputeNormalMatrix &) = delete;
      ComputeNormalMatrix &operator=(const ComputeNormalMatrix &) = delete;
      ~ComputeNormalMatrix()
      {
        if (scratch)
          (void)adaflo_free(problem, scratch);
      }
      void vmult(BlockVectorType &dst, const BlockVectorType &src) const
      {
        const double *s = device_values(src.block(0));
        double       *d = device_values(dst.block(0));
        const bool    contiguous = device_values(src.block(1)) == s + n && device_values(src.block(2)) == s + 2 * n &&
                                device_values(dst.block(1)) == d + n && device_values(dst.block(2)) == d + 2 * n;
        if (contiguous)
          {
            check(problem, adaflo_ls_compute_normal_vmult(problem, d, s), "ComputeNormalMatrix::vmult");
            return;
          }
        if (!scratch)
          check(problem, adaflo_malloc(problem, sizeof(double) * 6 * (std::size_t)n, reinterpret_cast<void **>(&scratch)),
                "ComputeNormalMatrix: scratch allocation");
        for (int b = 0; b < 3; ++b) // scratch[0 .. 3 n) = src
          check(problem, adaflo_vector_sadd(problem, scratch + b * n, 0., 1., device_values(src.block(b)), n), "copy");
        check(problem, adaflo_ls_compute_normal_vmult(problem, scratch + 3 * n, scratch), "ComputeNormalMatrix::vmult");
        for (int b = 0; b < 3; ++b)
          check(problem, adaflo_vector_sadd(problem, device_values(dst.block(b)), 0., 1., scratch + (3 + b) * n, n), "copy");
      }
      adaflo_ctx        *problem;
      const std::int64_t n;
      mutable double    *scratch = nullptr;
    };

    template <typename VectorType>
    struct ComputeCurvatureMatrix
    {
      explicit ComputeCurvatureMatrix(adaflo_ctx *problem)
        : problem(problem)
      {}
      void vmult(VectorType &dst, const VectorType &src) const
      {
        // (apply_diffusion = true, as the reference's struct)
        check(problem, adaflo_ls_compute_curvature_vmult(problem, device_values(dst), device_values(src), 1),
              "ComputeCurvatureMatrix::vmult");
      }
      adaflo_ctx *problem;
    };
  } // namespace hip
} // namespace adaflo

#endif
