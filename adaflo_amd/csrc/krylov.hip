// krylov.hip -- device-resident Krylov drivers on top of the operator C ABI:
// the callers of the operators on the path keep their vectors in HBM between operator
// applications, so the operator speed-up is not lost to PCIe.
//
//   preconditioned CG          SolverCG + DiagonalPreconditioner
//                              source/level_set_okz_reinitialization.cc:325-345 (rel. 1e-6),
//                              source/level_set_okz_compute_normal.cc:252-267,
//                              source/level_set_okz_compute_curvature.cc:345-355,
//                              source/navier_stokes_preconditioner.cc:743-773 (pressure mass)
//   preconditioned BiCGStab    SolverBicgstab (exact_residual = false)
//                              source/level_set_okz_advance_concentration.cc:623-644
//   DiagonalPreconditioner     source/diagonal_preconditioner.cc:27-124
//   ReductionControl           converged when ||r|| <= abs_tol or ||r|| <= rel_tol ||r_0||
//
// deal.II itself (the home of SolverCG / SolverBicgstab / ReductionControl) is not vendored in the
// reference tree; the algorithms below are the published ones (Hestenes-Stiefel PCG; van der
// Vorst's right-preconditioned BiCGStab with the convergence check after the first half step, as
// deal.II's SolverBicgstab does).  oracle/krylov_oracle.py restates the same recurrences in numpy.
// Scalars of the recurrences are reduced on the device (two-stage deterministic dot product) and
// read back once per use: the stopping test needs them on the host every iteration anyway.
#include "kernels.hpp"

#include <cmath>
#include <functional>
#include <string>
#include <vector>

namespace adaflo_hip
{
  namespace
  {
    constexpr int KT = 256;

    unsigned kgrid(const long n)
    {
      long b = (n + KT * 4 - 1) / (KT * 4);
      return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
    }

    // y = a*x + b*y
    __global__ __launch_bounds__(KT) void axpby_kernel(double *__restrict__ y, const double a,
                                                       const double *__restrict__ x, const double b, const long n)
    {
      for (long i = blockIdx.x * (long)KT + threadIdx.x; i < n; i += (long)gridDim.x * KT)
        y[i] = a * x[i] + b * y[i];
    }
    // z = x + a*(y + b*w)         (BiCGStab: p = r + beta (p - omega v))
    __global__ __launch_bounds__(KT) void xpaybw_kernel(double *__restrict__ z, const double *__restrict__ x,
                                                        const double a, const double *__restrict__ y, const double b,
                                                        const double *__restrict__ w, const long n)
    {
      for (long i = blockIdx.x * (long)KT + threadIdx.x; i < n; i += (long)gridDim.x * KT)
        z[i] = x[i] + a * (y[i] + b * w[i]);
    }
    // dst[blk][i] = src[blk][i] * inv_diag[i]   (DiagonalPreconditioner::vmult, :82-124; identity if null)
    __global__ __launch_bounds__(KT) void precond_kernel(double *__restrict__ dst, const double *__restrict__ src,
                                                         const double *__restrict__ inv_diag, const long n_block,
                                                         const long n)
    {
      for (long i = blockIdx.x * (long)KT + threadIdx.x; i < n; i += (long)gridDim.x * KT)
        dst[i] = inv_diag ? src[i] * inv_diag[i % n_block] : src[i];
    }
    // partial[b] = max_i |x_i| over the block's share
    __global__ __launch_bounds__(KT) void absmax_kernel(const double *__restrict__ x, const long n,
                                                        double *__restrict__ partial)
    {
      __shared__ double red[KT / 64];
      double            m = 0.;
      for (long i = blockIdx.x * (long)KT + threadIdx.x; i < n; i += (long)gridDim.x * KT)
        m = fmax(m, fabs(x[i]));
      for (int off = 32; off > 0; off >>= 1)
        m = fmax(m, __shfl_down(m, off, 64));
      if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = m;
      __syncthreads();
      if (threadIdx.x == 0)
        {
          for (int w = 1; w < KT / 64; ++w)
            m = fmax(m, red[w]);
          partial[blockIdx.x] = m;
        }
    }
    // DiagonalPreconditioner::reinit :38-45
    __global__ __launch_bounds__(KT) void invert_diag_kernel(double *__restrict__ inv, const double *__restrict__ diag,
                                                             const double threshold, const long n)
    {
      for (long i = blockIdx.x * (long)KT + threadIdx.x; i < n; i += (long)gridDim.x * KT)
        inv[i] = fabs(diag[i]) > threshold ? 1. / diag[i] : 1.;
    }

    struct Workspace
    {
      double *p = nullptr;
      ~Workspace()
      {
        if (p)
          (void)hipFree(p);
      }
    };

    using Operator = std::function<int(double *, const double *)>;

    struct Krylov
    {
      adaflo_ctx *ctx;
      long        n, n_block;
      const double *inv_diag;
      Operator      A;

      double dot(const double *a, const double *b)
      {
        return host_dot(ctx, a, b, n);
      }
      void axpby(double *y, const double a, const double *x, const double b)
      {
        hipLaunchKernelGGL(axpby_kernel, dim3(kgrid(n)), dim3(KT), 0, ctx->stream, y, a, x, b, n);
      }
      void precondition(double *dst, const double *src)
      {
        hipLaunchKernelGGL(precond_kernel, dim3(kgrid(n)), dim3(KT), 0, ctx->stream, dst, src, inv_diag, n_block, n);
      }
    };

    bool converged(const double res, const double res0, const adaflo_solver_control &c)
    {
      return res <= c.abs_tol || res <= c.rel_tol * res0; // ReductionControl
    }

    // returns 0, fills result; the iteration count follows SolverControl::last_step()
    int solve_cg(Krylov &K, double *x, const double *b, const adaflo_solver_control &c, adaflo_solver_result &out,
                 double *work)
    {
      const long n = K.n;
      double *r = work, *z = work + n, *p = work + 2 * n, *Ap = work + 3 * n;
      if (int e = K.A(Ap, x))
        return e;
      (void)hipMemcpyAsync(r, b, n * sizeof(double), hipMemcpyDeviceToDevice, K.ctx->stream);
      K.axpby(r, -1., Ap, 1.); // r = b - A x
      double res = std::sqrt(K.dot(r, r));
      out.initial_residual = res;
      out.iterations       = 0;
      if (converged(res, res, c))
        {
          out.final_residual = res;
          out.converged      = 1;
          return 0;
        }
      K.precondition(z, r);
      (void)hipMemcpyAsync(p, z, n * sizeof(double), hipMemcpyDeviceToDevice, K.ctx->stream);
      double rz = K.dot(r, z);
      for (int it = 1; it <= c.max_iterations; ++it)
        {
          if (int e = K.A(Ap, p))
            return e;
          const double alpha = rz / K.dot(p, Ap);
          K.axpby(x, alpha, p, 1.);
          K.axpby(r, -alpha, Ap, 1.);
          res            = std::sqrt(K.dot(r, r));
          out.iterations = it;
          if (converged(res, out.initial_residual, c))
            {
              out.final_residual = res;
              out.converged      = 1;
              return 0;
            }
          K.precondition(z, r);
          const double rz_new = K.dot(r, z);
          K.axpby(p, 1., z, rz_new / rz); // p = z + beta p
          rz = rz_new;
        }
      out.final_residual = res;
      out.converged      = 0;
      return 0;
    }

    int solve_bicgstab(Krylov &K, double *x, const double *b, const adaflo_solver_control &c,
                       adaflo_solver_result &out, double *work)
    {
      const long n = K.n;
      double *r = work, *rbar = work + n, *p = work + 2 * n, *v = work + 3 * n, *y = work + 4 * n, *z = work + 5 * n,
             *t = work + 6 * n;
      if (int e = K.A(v, x))
        return e;
      (void)hipMemcpyAsync(r, b, n * sizeof(double), hipMemcpyDeviceToDevice, K.ctx->stream);
      K.axpby(r, -1., v, 1.);
      (void)hipMemcpyAsync(rbar, r, n * sizeof(double), hipMemcpyDeviceToDevice, K.ctx->stream);
      double res = std::sqrt(K.dot(r, r));
      out.initial_residual = res;
      out.iterations       = 0;
      out.converged        = 0;
      if (converged(res, res, c))
        {
          out.final_residual = res;
          out.converged      = 1;
          return 0;
        }
      double rho = 1., alpha = 1., omega = 1.;
      for (int it = 1; it <= c.max_iterations; ++it)
        {
          const double rho_new = K.dot(rbar, r);
          if (rho_new == 0. || omega == 0.)
            break; // breakdown (deal.II restarts; the callers fall back to GMRES)
          if (it == 1)
            (void)hipMemcpyAsync(p, r, n * sizeof(double), hipMemcpyDeviceToDevice, K.ctx->stream);
          else
            {
              const double beta = (rho_new / rho) * (alpha / omega);
              hipLaunchKernelGGL(xpaybw_kernel, dim3(kgrid(n)), dim3(KT), 0, K.ctx->stream, p, r, beta, p, -omega, v, n);
            }
          rho = rho_new;
          K.precondition(y, p);
          if (int e = K.A(v, y))
            return e;
          alpha = rho / K.dot(rbar, v);
          K.axpby(r, -alpha, v, 1.); // s
          res            = std::sqrt(K.dot(r, r));
          out.iterations = it;
          if (converged(res, out.initial_residual, c))
            {
              K.axpby(x, alpha, y, 1.);
              out.converged = 1;
              break;
            }
          K.precondition(z, r);
          if (int e = K.A(t, z))
            return e;
          omega = K.dot(t, r) / K.dot(t, t);
          K.axpby(x, alpha, y, 1.);
          K.axpby(x, omega, z, 1.);
          K.axpby(r, -omega, t, 1.);
          res = std::sqrt(K.dot(r, r));
          if (converged(res, out.initial_residual, c))
            {
              out.converged = 1;
              break;
            }
        }
      out.final_residual = res;
      return 0;
    }

    int kfail(adaflo_ctx *ctx, const int code, const std::string &msg)
    {
      if (ctx)
        ctx->last_error = msg;
      return code;
    }
  } // namespace
} // namespace adaflo_hip

using namespace adaflo_hip;

extern "C" {

int adaflo_invert_diagonal(adaflo_ctx *ctx, double *inverse_diagonal, const double *diagonal, int64_t n)
{
  if (!ctx)
    return ADAFLO_ENOTINIT;
  if (!inverse_diagonal || !diagonal || n <= 0)
    return kfail(ctx, ADAFLO_EINVAL, "null vector");
  Workspace w;
  const unsigned nb = kgrid(n);
  if (hipMalloc(&w.p, nb * sizeof(double)) != hipSuccess)
    return kfail(ctx, ADAFLO_ENOMEM, "out of device memory");
  hipLaunchKernelGGL(absmax_kernel, dim3(nb), dim3(KT), 0, ctx->stream, diagonal, (long)n, w.p);
  std::vector<double> part(nb);
  if (hipMemcpyAsync(part.data(), w.p, nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return kfail(ctx, ADAFLO_EHIP, "reduction failed");
  double linfty = 0.;
  for (const double v : part)
    linfty = std::fmax(linfty, v);
  hipLaunchKernelGGL(invert_diag_kernel, dim3(kgrid(n)), dim3(KT), 0, ctx->stream, inverse_diagonal, diagonal,
                     1e-10 * linfty, (long)n);
  return hipGetLastError() == hipSuccess ? 0 : kfail(ctx, ADAFLO_EHIP, "kernel launch failed");
}

int adaflo_solve(adaflo_ctx *ctx, int op, int method, double *x, const double *b, const double *inverse_diagonal,
                 const adaflo_solver_control *control, adaflo_solver_result *result)
{
  if (!ctx)
    return ADAFLO_ENOTINIT;
  if (!x || !b || !control || !result)
    return kfail(ctx, ADAFLO_EINVAL, "null argument");
  Krylov K{};
  K.ctx      = ctx;
  K.inv_diag = inverse_diagonal;
  int blocks = 1;
  switch (op)
    {
      case ADAFLO_OP_LS_ADVANCE_CONCENTRATION:
        K.n_block = ctx->n_nodes_ls;
        K.A       = [ctx](double *d, const double *s) { return adaflo_ls_advance_concentration_vmult(ctx, d, s); };
        break;
      case ADAFLO_OP_LS_REINITIALIZATION:
      case ADAFLO_OP_LS_REINITIALIZATION_DIFFUSE:
        {
          const int diffuse = op == ADAFLO_OP_LS_REINITIALIZATION_DIFFUSE;
          K.n_block         = ctx->n_nodes_ls;
          K.A = [ctx, diffuse](double *d, const double *s) { return adaflo_ls_reinitialization_vmult(ctx, d, s, diffuse); };
        }
        break;
      case ADAFLO_OP_LS_NORMAL:
        K.n_block = ctx->n_nodes_ls;
        blocks    = 3;
        K.A       = [ctx](double *d, const double *s) { return adaflo_ls_compute_normal_vmult(ctx, d, s); };
        break;
      case ADAFLO_OP_LS_CURVATURE:
        K.n_block = ctx->n_nodes_ls;
        K.A       = [ctx](double *d, const double *s) { return adaflo_ls_compute_curvature_vmult(ctx, d, s, 1); };
        break;
      case ADAFLO_OP_NS_PRESSURE_MASS:
        K.n_block = ctx->n_nodes_p;
        K.A       = [ctx](double *d, const double *s) { return adaflo_ns_pressure_mass_vmult(ctx, d, s); };
        break;
      case ADAFLO_OP_NS_PRESSURE_POISSON:
        K.n_block = ctx->n_nodes_p;
        K.A       = [ctx](double *d, const double *s) { return adaflo_ns_pressure_poisson_vmult(ctx, d, s); };
        break;
      case ADAFLO_OP_NS_VELOCITY:
        K.n_block = 3 * ctx->n_nodes_u;
        K.A       = [ctx](double *d, const double *s) { return adaflo_ns_velocity_vmult(ctx, d, s); };
        break;
      default:
        return kfail(ctx, ADAFLO_EINVAL, "unknown operator");
    }
  if (K.n_block <= 0)
    return kfail(ctx, ADAFLO_ENOTINIT, "operator space not initialised (level-set degree 0?)");
  K.n = K.n_block * blocks;
  const int nvec = method == ADAFLO_SOLVER_CG ? 4 : 7;
  Workspace w;
  if (hipMalloc(&w.p, (size_t)nvec * K.n * sizeof(double)) != hipSuccess)
    return kfail(ctx, ADAFLO_ENOMEM, "out of device memory for the Krylov vectors");
  int rc;
  if (method == ADAFLO_SOLVER_CG)
    rc = solve_cg(K, x, b, *control, *result, w.p);
  else if (method == ADAFLO_SOLVER_BICGSTAB)
    rc = solve_bicgstab(K, x, b, *control, *result, w.p);
  else
    return kfail(ctx, ADAFLO_EINVAL, "unknown solver");
  if (rc != 0)
    return rc; // the operator recorded its message
  if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess)
    return kfail(ctx, ADAFLO_EHIP, "Krylov kernels failed");
  return 0;
}

} // extern "C"
