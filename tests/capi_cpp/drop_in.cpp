// drop_in.cpp -- the drop-in seam of the reference met from C++ (host-only translation unit, g++ -std=c++17,
// linked against libadaflo_hip.so; no HIP header needed: device memory goes through adaflo_malloc / adaflo_copy_*).
//
// The reference hands NavierStokesMatrix to deal.II's templated Krylov solvers, which only need
//     void vmult(VectorType &dst, const VectorType &src) const
// (source/navier_stokes.cc:593-631: SolverFGMRES<BlockVector>::solve(navier_stokes_matrix, solution_update,
// system_rhs, preconditioner)).  This file does the same with a small templated FGMRES of its own:
//   * BlockVector: two device arrays (velocity | pressure) with the few vector operations a Krylov solver needs,
//     forwarded to adaflo_vector_*; `block(i).get_values()` is what include/adaflo_hip.hpp asks of a vector type;
//   * adaflo::hip::NavierStokesMatrix::vmult          -> adaflo_ns_vmult            (the shipped header, not a test class)
//   * adaflo::hip::NavierStokesPreconditioner::vmult  -> adaflo_ns_preconditioner_vmult (with inner solves)
//   * solve_fgmres<Matrix, Vector, Preconditioner>: right-preconditioned FGMRES(50), modified Gram-Schmidt.
// It linearises a smooth field on a 6 x 5 x 4 Q2/Q1 brick (adaflo_ns_residual produces the state), solves
// J du = J x_true, and checks the solution against adaflo_ns_solve_system -- the same algorithm inside the library.
#include "adaflo_hip.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{
  void check(adaflo_ctx *ctx, const int code, const char *what)
  {
    if (code != ADAFLO_OK)
      throw std::runtime_error(std::string(what) + ": " + adaflo_last_error(ctx));
  }

  class DeviceArray
  {
  public:
    DeviceArray(adaflo_ctx *ctx, const int64_t n)
      : ctx(ctx)
      , n(n)
    {
      check(ctx, adaflo_malloc(ctx, sizeof(double) * (size_t)(n > 0 ? n : 1), reinterpret_cast<void **>(&p)), "adaflo_malloc");
      check(ctx, adaflo_vector_fill(ctx, p, 0., n), "fill");
    }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    ~DeviceArray() { (void)adaflo_free(ctx, p); }
    void from_host(const std::vector<double> &h) { check(ctx, adaflo_copy_h2d(ctx, p, h.data(), sizeof(double) * h.size()), "h2d"); }
    std::vector<double> to_host() const
    {
      std::vector<double> h((size_t)n);
      check(ctx, adaflo_copy_d2h(ctx, h.data(), p, sizeof(double) * h.size()), "d2h");
      return h;
    }
    double       *get_values() { return p; }
    const double *get_values() const { return p; }
    adaflo_ctx *ctx;
    int64_t     n;
    double     *p = nullptr;
  };

  // what LinearAlgebra::distributed::BlockVector<double> offers to SolverFGMRES, on two device blocks
  class BlockVector
  {
  public:
    BlockVector(adaflo_ctx *ctx, const int64_t nu, const int64_t np)
      : u(ctx, nu)
      , p(ctx, np)
    {}
    void reinit_like(const BlockVector &) {}
    BlockVector &operator=(const double s)
    {
      check(u.ctx, adaflo_vector_fill(u.ctx, u.p, s, u.n), "fill");
      check(u.ctx, adaflo_vector_fill(u.ctx, p.p, s, p.n), "fill");
      return *this;
    }
    // *this = a * *this + b * y
    void sadd(const double a, const double b, const BlockVector &y)
    {
      check(u.ctx, adaflo_vector_sadd(u.ctx, u.p, a, b, y.u.p, u.n), "sadd");
      check(u.ctx, adaflo_vector_sadd(u.ctx, p.p, a, b, y.p.p, p.n), "sadd");
    }
    void   equ(const double b, const BlockVector &y) { sadd(0., b, y); }
    void   add(const double b, const BlockVector &y) { sadd(1., b, y); }
    double operator*(const BlockVector &y) const
    {
      double a = 0., b = 0.;
      check(u.ctx, adaflo_vector_dot(u.ctx, u.p, y.u.p, u.n, &a), "dot");
      check(u.ctx, adaflo_vector_dot(u.ctx, p.p, y.p.p, p.n, &b), "dot");
      return a + b;
    }
    double      l2_norm() const { return std::sqrt((*this) * (*this)); }
    DeviceArray       &block(const unsigned i) { return i == 0 ? u : p; }
    const DeviceArray &block(const unsigned i) const { return i == 0 ? u : p; }
    DeviceArray        u, p;
  };

  using NavierStokesMatrixHIP         = adaflo::hip::NavierStokesMatrix<DeviceArray, BlockVector>;
  using NavierStokesPreconditionerHIP = adaflo::hip::NavierStokesPreconditioner<BlockVector>;

  // right-preconditioned flexible GMRES(restart); x is the start value and the result; returns the iterations
  template <class Matrix, class Vector, class Preconditioner, class MakeVector>
  int solve_fgmres(const Matrix &A, Vector &x, const Vector &b, const Preconditioner &M, const double abs_tol,
                   const int max_iterations, const int restart, MakeVector make, double *final_residual)
  {
    std::vector<Vector *> V, Z;
    for (int j = 0; j <= restart; ++j)
      V.push_back(make());
    for (int j = 0; j < restart; ++j)
      Z.push_back(make());
    Vector *w = make();
    std::vector<double> H((size_t)(restart + 1) * restart), g(restart + 1), cs(restart), sn(restart), y(restart);
    int    iterations = 0;
    double res        = 0.;
    for (;;)
      {
        A.vmult(*w, x);                      // r = b - A x
        V[0]->equ(1., b);
        V[0]->add(-1., *w);
        const double beta = V[0]->l2_norm();
        res               = beta;
        if (beta <= abs_tol || iterations >= max_iterations)
          break;
        V[0]->sadd(1. / beta, 0., *V[0]);
        std::fill(g.begin(), g.end(), 0.);
        g[0]   = beta;
        int kk = 0;
        for (int j = 0; j < restart && iterations < max_iterations; ++j)
          {
            M.vmult(*Z[j], *V[j]);
            A.vmult(*w, *Z[j]);
            for (int i = 0; i <= j; ++i)
              {
                const double h = (*w) * (*V[i]);
                H[(size_t)i * restart + j] = h;
                w->add(-h, *V[i]);
              }
            const double hn = w->l2_norm();
            H[(size_t)(j + 1) * restart + j] = hn;
            V[j + 1]->equ(hn > 0. ? 1. / hn : 1., *w);
            for (int i = 0; i < j; ++i)
              {
                const double a = H[(size_t)i * restart + j], c = H[(size_t)(i + 1) * restart + j];
                H[(size_t)i * restart + j]       = cs[i] * a + sn[i] * c;
                H[(size_t)(i + 1) * restart + j] = -sn[i] * a + cs[i] * c;
              }
            const double a = H[(size_t)j * restart + j], c = H[(size_t)(j + 1) * restart + j], r = std::hypot(a, c);
            cs[j] = r > 0. ? a / r : 1.;
            sn[j] = r > 0. ? c / r : 0.;
            H[(size_t)j * restart + j]       = r;
            H[(size_t)(j + 1) * restart + j] = 0.;
            g[j + 1]                         = -sn[j] * g[j];
            g[j]                             = cs[j] * g[j];
            ++iterations;
            kk  = j + 1;
            res = std::fabs(g[j + 1]);
            if (res <= abs_tol)
              break;
          }
        for (int i = kk - 1; i >= 0; --i) // back substitution, x += Z y
          {
            double s = g[i];
            for (int l = i + 1; l < kk; ++l)
              s -= H[(size_t)i * restart + l] * y[l];
            y[i] = s / H[(size_t)i * restart + i];
          }
        for (int i = 0; i < kk; ++i)
          x.add(y[i], *Z[i]);
        if (res <= abs_tol)
          break;
      }
    for (Vector *v : V)
      delete v;
    for (Vector *v : Z)
      delete v;
    delete w;
    if (final_residual)
      *final_residual = res;
    return iterations;
  }
} // namespace

int main()
{
  adaflo_ctx *ctx = nullptr;
  try
    {
      const int    nc[3] = {6, 5, 4}, k = 2;
      const double h[3]  = {1. / 3., 0.4, 0.5};
      adaflo_brick_desc d{};
      d.dim = 3;
      for (int e = 0; e < 3; ++e)
        {
          d.ncell[e]  = nc[e];
          d.h[e]      = h[e];
          d.origin[e] = -1.;
        }
      d.velocity_degree      = k;
      d.velocity_constrained = (1u << 18) - 1u; // Dirichlet on all six faces, all components
      d.pressure_average_fix = 1;
      if (adaflo_ctx_create(&d, &ctx) != ADAFLO_OK)
        throw std::runtime_error(std::string("adaflo_ctx_create: ") + adaflo_last_error(nullptr));
      // BDF-2 with constant step 0.05 (source/time_stepping.cc:160-168), Newton linearisation, nu = 0.1
      adaflo_ns_params prm{};
      prm.physical_type = ADAFLO_INCOMPRESSIBLE;
      prm.linearization = ADAFLO_COUPLED_IMPLICIT_NEWTON;
      prm.beta          = 0.5;
      prm.density       = 1.;
      prm.viscosity     = 0.1;
      prm.weight        = 1.5 / 0.05;
      prm.weight_old    = -2. / 0.05;
      prm.weight_old_old = 0.5 / 0.05;
      prm.tau1          = 1.;
      prm.extrap_old    = 2.;
      prm.extrap_old_old = -1.;
      check(ctx, adaflo_ns_set_params(ctx, &prm), "adaflo_ns_set_params");
      const int64_t nu = adaflo_n_dofs_u(ctx), np = adaflo_n_dofs_p(ctx);
      const int     nn[3] = {k * nc[0] + 1, k * nc[1] + 1, k * nc[2] + 1};
      if (nu != 3LL * nn[0] * nn[1] * nn[2])
        throw std::runtime_error("unexpected number of velocity DoFs");
      // a smooth solenoidal-ish field at the nodes (FE_Q(2) support points are equidistant), vanishing nowhere special
      std::vector<double> u0((size_t)nu), rhs_u((size_t)nu), rhs_p((size_t)np);
      for (int kz = 0; kz < nn[2]; ++kz)
        for (int jy = 0; jy < nn[1]; ++jy)
          for (int ix = 0; ix < nn[0]; ++ix)
            {
              const double x = -1. + 0.5 * h[0] * ix, y = -1. + 0.5 * h[1] * jy, z = -1. + 0.5 * h[2] * kz;
              const size_t n = ((size_t)kz * nn[1] + jy) * nn[0] + ix;
              u0[3 * n + 0] = std::sin(1.3 * y) * std::cos(0.7 * z);
              u0[3 * n + 1] = std::sin(0.9 * z) * std::cos(1.1 * x);
              u0[3 * n + 2] = std::sin(1.7 * x) * std::cos(0.5 * y);
            }
      unsigned long long s = 88172645463325252ULL; // xorshift: deterministic right-hand side
      auto rnd = [&s]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (double)(s >> 11) / 9007199254740992. - 0.5;
      };
      for (double &v : rhs_u)
        v = rnd();
      for (double &v : rhs_p)
        v = rnd();
      BlockVector sol(ctx, nu, np), old(ctx, nu, np), rhs(ctx, nu, np), tmp(ctx, nu, np);
      sol.u.from_host(u0);
      for (double &v : u0)
        v *= 0.97;
      old.u.from_host(u0);
      // the residual is the only producer of the linearisation state (navier_stokes_matrix.cc:778-798)
      check(ctx, adaflo_ns_residual(ctx, tmp.u.p, tmp.p.p, sol.u.p, sol.p.p, nullptr, nullptr, old.u.p, old.u.p),
            "adaflo_ns_residual");
      // a consistent right-hand side: rhs = J x_true (the pressure is only determined up to its mean value)
      {
        BlockVector xt(ctx, nu, np);
        xt.u.from_host(rhs_u);
        xt.p.from_host(rhs_p);
        check(ctx, adaflo_ns_vmult(ctx, rhs.u.p, rhs.p.p, xt.u.p, xt.p.p), "adaflo_ns_vmult");
      }
      check(ctx, adaflo_ns_preconditioner_setup(ctx), "adaflo_ns_preconditioner_setup");
      check(ctx, adaflo_ns_set_iterations_before_inner_solvers(ctx, 0), "set_iterations_before_inner_solvers");

      const NavierStokesMatrixHIP         matrix(ctx);
      const NavierStokesPreconditionerHIP preconditioner(ctx);
      const double tol = 1e-8 * rhs.l2_norm();
      BlockVector  x(ctx, nu, np);
      double       res_cpp = 0.;
      const int    its_cpp = solve_fgmres(matrix, x, rhs, preconditioner, tol, 200, 50,
                                       [&]() { return new BlockVector(ctx, nu, np); }, &res_cpp);
      // the library's own solver on the same system
      BlockVector           xl(ctx, nu, np);
      adaflo_solver_control control{200, tol, 0.};
      adaflo_solver_result  result{};
      check(ctx, adaflo_ns_solve_system(ctx, xl.u.p, xl.p.p, rhs.u.p, rhs.p.p, &control, 50, &result), "adaflo_ns_solve_system");
      // true residual of the C++ solve
      matrix.vmult(tmp, x);
      tmp.sadd(-1., 1., rhs);
      const double true_res = tmp.l2_norm();
      tmp.equ(1., x);
      tmp.add(-1., xl);
      const double diff = tmp.l2_norm() / xl.l2_norm();
      std::printf("drop_in: %lld + %lld DoFs, C++ FGMRES %d iterations (residual %.3e, true %.3e, tolerance %.3e), "
                  "adaflo_ns_solve_system %d iterations, relative difference of the solutions %.3e\n",
                  (long long)nu, (long long)np, its_cpp, res_cpp, true_res, tol, result.iterations, diff);
      const bool ok = its_cpp > 0 && its_cpp < 200 && result.converged && true_res <= 10. * tol &&
                      std::abs(its_cpp - result.iterations) <= 1 && diff < 1e-6;
      std::printf(ok ? "drop_in: OK\n" : "drop_in: FAILED\n");
      adaflo_ctx_destroy(ctx);
      return ok ? 0 : 1;
    }
  catch (const std::exception &e)
    {
      std::fprintf(stderr, "drop_in: %s\n", e.what());
      if (ctx)
        adaflo_ctx_destroy(ctx);
      return 2;
    }
}
