// operators.cpp -- every class and method of include/adaflo_hip.hpp called from a host-only C++17 translation unit
// (g++ -Wall -Werror, linked against libadaflo_hip.so) and compared, entry by entry, with the values the ctypes path of
// the test suite produced for the same inputs (tests/test_boundary_gpu.py writes them into the case file given as
// argv[1]; the ctypes path is what the parity tests hold against the oracle).  Tolerance 1e-13 relative l2: the two
// paths run the same device code, the deterministic kernels make them bitwise equal in practice.
//
// The vector classes here are stand-ins for LinearAlgebra::distributed::Vector / BlockVector: all the header asks
// of them is `get_values()` and `block(i)`.
#include "adaflo_hip.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{
  using Case = std::map<std::string, std::vector<double>>;

  Case read_case(const char *path)
  {
    Case  c;
    FILE *f = std::fopen(path, "rb");
    if (!f)
      throw std::runtime_error(std::string("cannot open ") + path);
    for (;;)
      {
        std::int32_t len = 0;
        if (std::fread(&len, sizeof(len), 1, f) != 1)
          break;
        std::string name((size_t)len, ' ');
        std::int64_t n = 0;
        if (std::fread(&name[0], 1, (size_t)len, f) != (size_t)len || std::fread(&n, sizeof(n), 1, f) != 1)
          throw std::runtime_error("truncated case file");
        std::vector<double> v((size_t)n);
        if (n > 0 && std::fread(v.data(), sizeof(double), (size_t)n, f) != (size_t)n)
          throw std::runtime_error("truncated case file");
        c[name] = std::move(v);
      }
    std::fclose(f);
    return c;
  }

  const std::vector<double> &get(const Case &c, const std::string &name)
  {
    const auto it = c.find(name);
    if (it == c.end())
      throw std::runtime_error("case file has no entry " + name);
    return it->second;
  }

  // LinearAlgebra::distributed::Vector<double, MemorySpace::Default>, as far as the header is concerned
  class Vector
  {
  public:
    Vector(adaflo_ctx *ctx, const std::int64_t n)
      : ctx(ctx)
      , n(n)
    {
      adaflo::hip::check(ctx, adaflo_malloc(ctx, sizeof(double) * (size_t)(n > 0 ? n : 1), reinterpret_cast<void **>(&p)), "adaflo_malloc");
      adaflo::hip::check(ctx, adaflo_vector_fill(ctx, p, 0., n), "fill");
    }
    Vector(adaflo_ctx *ctx, const std::vector<double> &h)
      : Vector(ctx, (std::int64_t)h.size())
    {
      adaflo::hip::check(ctx, adaflo_copy_h2d(ctx, p, h.data(), sizeof(double) * h.size()), "h2d");
    }
    Vector(const Vector &) = delete;
    Vector &operator=(const Vector &) = delete;
    ~Vector() { (void)adaflo_free(ctx, p); }
    double             *get_values() { return p; }
    const double       *get_values() const { return p; }
    std::vector<double> to_host() const
    {
      std::vector<double> h((size_t)n);
      adaflo::hip::check(ctx, adaflo_copy_d2h(ctx, h.data(), p, sizeof(double) * h.size()), "d2h");
      return h;
    }
    adaflo_ctx  *ctx;
    std::int64_t n;
    double      *p = nullptr;
  };

  // LinearAlgebra::distributed::BlockVector<double>: blocks are separate allocations
  class BlockVector
  {
  public:
    void          push_back(Vector *v) { blocks.emplace_back(v); }
    Vector       &block(const unsigned i) { return *blocks[i]; }
    const Vector &block(const unsigned i) const { return *blocks[i]; }

  private:
    std::vector<std::unique_ptr<Vector>> blocks;
  };

  int    n_checked = 0, n_failed = 0;
  double worst = 0.;
  void   compare(const char *what, const std::vector<double> &got, const std::vector<double> &ref)
  {
    double d = 0., r = 0.;
    if (got.size() != ref.size())
      d = r = 1.;
    else
      for (size_t i = 0; i < got.size(); ++i)
        {
          d += (got[i] - ref[i]) * (got[i] - ref[i]);
          r += ref[i] * ref[i];
        }
    const double e = std::sqrt(d) / std::max(std::sqrt(r), 1e-300);
    ++n_checked;
    worst = std::max(worst, e);
    const bool ok = e < 1e-13 && std::isfinite(e);
    if (!ok)
      ++n_failed;
    std::printf("operators: %-44s rel. l2 difference %.2e %s\n", what, e, ok ? "" : "FAILED");
  }

  adaflo_ctx *make_ctx(const std::vector<double> &d)
  {
    adaflo_brick_desc b{};
    b.dim = 3;
    for (int e = 0; e < 3; ++e)
      {
        b.ncell[e]  = (int)d[e];
        b.h[e]      = d[3 + e];
        b.origin[e] = d[6 + e];
      }
    b.velocity_degree      = (int)d[9];
    b.ls_degree            = (int)d[10];
    b.velocity_constrained = (std::uint32_t)d[11];
    b.pressure_constrained = (std::uint32_t)d[12];
    b.ls_constrained       = (std::uint32_t)d[13];
    b.pressure_average_fix = (int)d[14];
    adaflo_ctx *ctx        = nullptr;
    if (adaflo_ctx_create(&b, &ctx) != ADAFLO_OK)
      throw std::runtime_error(std::string("adaflo_ctx_create: ") + adaflo_last_error(nullptr));
    return ctx;
  }
} // namespace

int main(int argc, char **argv)
{
  if (argc < 2)
    {
      std::fprintf(stderr, "usage: operators <case file>\n");
      return 2;
    }
  adaflo_ctx *ctx = nullptr, *ls = nullptr;
  try
    {
      const Case c = read_case(argv[1]);
      using namespace adaflo::hip;
      // ------------------------------------------------------------------ Navier-Stokes operator
      ctx = make_ctx(get(c, "ns_desc"));
      {
        const std::vector<double> &q = get(c, "ns_params");
        adaflo_ns_params           prm{};
        prm.physical_type  = (int)q[0];
        prm.linearization  = (int)q[1];
        prm.beta           = q[2];
        prm.tau_grad_div   = q[3];
        prm.density        = q[4];
        prm.viscosity      = q[5];
        prm.damping        = q[6];
        prm.density_diff   = q[7];
        prm.weight         = q[8];
        prm.weight_old     = q[9];
        prm.weight_old_old = q[10];
        prm.tau1           = q[11];
        prm.extrap_old     = q[12];
        prm.extrap_old_old = q[13];
        check(ctx, adaflo_ns_set_params(ctx, &prm), "adaflo_ns_set_params");
      }
      check(ctx, adaflo_ns_set_linearization(ctx, get(c, "lin").data(), 0), "adaflo_ns_set_linearization");
      const std::int64_t nu = adaflo_n_dofs_u(ctx), np = adaflo_n_dofs_p(ctx);
      auto make_block = [&](const char *u, const char *p) {
        BlockVector b;
        b.push_back(u ? new Vector(ctx, get(c, u)) : new Vector(ctx, nu));
        b.push_back(p ? new Vector(ctx, get(c, p)) : new Vector(ctx, np));
        return b;
      };
      BlockVector src = make_block("src_u", "src_p"), old = make_block("old_u", nullptr), old_old = make_block("old_old_u", nullptr),
                  user = make_block("user_u", "user_p"), dst = make_block(nullptr, nullptr);
      const NavierStokesMatrix<Vector, BlockVector> matrix(ctx, old, old_old);
      if (matrix.n_dofs_u() != nu || matrix.n_dofs_p() != np)
        throw std::runtime_error("n_dofs");
      (void)matrix.get_matvec_statistics(); // reset
      matrix.vmult(dst, src);
      compare("vmult (velocity)", dst.block(0).to_host(), get(c, "vmult_u"));
      compare("vmult (pressure)", dst.block(1).to_host(), get(c, "vmult_p"));
      matrix.fix_linearization_point();
      matrix.velocity_vmult(dst.block(0), src.block(0));
      compare("fix_linearization_point + velocity_vmult", dst.block(0).to_host(), get(c, "velocity_vmult"));
      {
        Vector base(ctx, get(c, "src_p"));
        matrix.divergence_vmult_add(base, src.block(0));
        compare("divergence_vmult_add", base.to_host(), get(c, "divergence"));
        matrix.divergence_vmult_add(base, src.block(0), true);
        compare("divergence_vmult_add (weight_by_viscosity)", base.to_host(), get(c, "divergence_weighted"));
      }
      matrix.pressure_poisson_vmult(dst.block(1), src.block(1));
      compare("pressure_poisson_vmult", dst.block(1).to_host(), get(c, "pressure_poisson"));
      matrix.pressure_mass_vmult(dst.block(1), src.block(1));
      compare("pressure_mass_vmult", dst.block(1).to_host(), get(c, "pressure_mass"));
      matrix.pressure_convdiff_vmult(dst.block(1), src.block(1));
      compare("pressure_convdiff_vmult", dst.block(1).to_host(), get(c, "pressure_convdiff"));
      {
        Vector v(ctx, get(c, "src_p"));
        matrix.apply_pressure_average_projection(v);
        compare("apply_pressure_average_projection", v.to_host(), get(c, "projection"));
      }
      {
        // system_rhs is read-modify-written as in the reference (the cell loop adds into it, then rhs = user_rhs - rhs)
        BlockVector rhs = make_block("residual_in_u", "residual_in_p");
        matrix.residual(rhs, src, user);
        compare("residual (velocity)", rhs.block(0).to_host(), get(c, "residual_u"));
        compare("residual (pressure)", rhs.block(1).to_host(), get(c, "residual_p"));
      }
      matrix.vmult(dst, src); // on the state the residual has just written
      compare("vmult after residual (velocity)", dst.block(0).to_host(), get(c, "vmult2_u"));
      compare("vmult after residual (pressure)", dst.block(1).to_host(), get(c, "vmult2_p"));
      {
        const auto st = matrix.get_matvec_statistics();
        const bool ok = st.second == 2 && st.first.avg > 0. && st.first.min == st.first.max;
        std::printf("operators: get_matvec_statistics: %u applications, %.3e s %s\n", st.second, st.first.avg, ok ? "" : "FAILED");
        ++n_checked;
        n_failed += ok ? 0 : 1;
      }
      {
        // the block preconditioner through its wrapper: same call as the C ABI made for the reference values
        check(ctx, adaflo_ns_preconditioner_setup(ctx), "adaflo_ns_preconditioner_setup");
        const NavierStokesPreconditioner<BlockVector> preconditioner(ctx);
        preconditioner.vmult(dst, src);
        compare("NavierStokesPreconditioner::vmult (velocity)", dst.block(0).to_host(), get(c, "prec_u"));
        compare("NavierStokesPreconditioner::vmult (pressure)", dst.block(1).to_host(), get(c, "prec_p"));
      }
      {
        bool threw = false;
        try
          {
            const NavierStokesMatrix<Vector, BlockVector> no_history(ctx);
            no_history.residual(dst, src, user);
          }
        catch (const Error &e)
          {
            threw = e.code == ADAFLO_ENOTINIT;
          }
        std::printf("operators: residual without the old solutions throws %s\n", threw ? "" : "FAILED");
        ++n_checked;
        n_failed += threw ? 0 : 1;
      }
      // ------------------------------------------------------------------ level-set operators
      ls = make_ctx(get(c, "ls_desc"));
      {
        const std::vector<double> &q = get(c, "ls_params");
        adaflo_ls_params           p{};
        p.epsilon_used        = q[0];
        p.minimal_edge_length = q[1];
        p.time_step           = q[2];
        p.weight              = q[3];
        p.weight_old          = q[4];
        p.weight_old_old      = q[5];
        p.epsilon             = q[6];
        check(ls, adaflo_ls_set_params(ls, &p), "adaflo_ls_set_params");
      }
      Vector diag(ls, get(c, "ls_diag"));
      check(ls, adaflo_ls_set_diagonal(ls, diag.get_values()), "adaflo_ls_set_diagonal");
      check(ls, adaflo_ls_set_evaluated_convection(ls, get(c, "ls_convection").data(), 0), "set_evaluated_convection");
      check(ls, adaflo_ls_set_evaluated_normal(ls, get(c, "ls_normal_q").data(), 0), "set_evaluated_normal");
      const std::int64_t nls = adaflo_n_dofs_ls(ls);
      Vector             ls_src(ls, get(c, "ls_src")), ls_dst(ls, nls);
      const AdvanceConcentrationMatrix<Vector> advance(ls);
      advance.vmult(ls_dst, ls_src);
      compare("AdvanceConcentrationMatrix::vmult", ls_dst.to_host(), get(c, "ls_advance"));
      const ReinitializationMatrix<Vector> reinit(ls, false), diffuse(ls, true);
      reinit.vmult(ls_dst, ls_src);
      compare("ReinitializationMatrix::vmult", ls_dst.to_host(), get(c, "ls_reinit"));
      diffuse.vmult(ls_dst, ls_src);
      compare("ReinitializationMatrix::vmult (diffuse_only)", ls_dst.to_host(), get(c, "ls_reinit_diffuse"));
      const ComputeCurvatureMatrix<Vector> curvature(ls);
      curvature.vmult(ls_dst, ls_src);
      compare("ComputeCurvatureMatrix::vmult", ls_dst.to_host(), get(c, "ls_curvature"));
      {
        // three separately allocated blocks (the staging path of the wrapper) ...
        const std::vector<double> &n3 = get(c, "ls_normal_src");
        BlockVector                ns, nd;
        for (int b = 0; b < 3; ++b)
          {
            ns.push_back(new Vector(ls, std::vector<double>(n3.begin() + b * nls, n3.begin() + (b + 1) * nls)));
            nd.push_back(new Vector(ls, nls));
          }
        const ComputeNormalMatrix<BlockVector> normal(ls);
        normal.vmult(nd, ns);
        std::vector<double> got;
        for (int b = 0; b < 3; ++b)
          {
            const std::vector<double> h = nd.block(b).to_host();
            got.insert(got.end(), h.begin(), h.end());
          }
        compare("ComputeNormalMatrix::vmult (separate blocks)", got, get(c, "ls_normal"));
        // ... and blocks that are views into one array (passed through)
        struct View
        {
          double       *p;
          double       *get_values() { return p; }
          const double *get_values() const { return p; }
        };
        struct ViewBlocks
        {
          View        b[3];
          View       &block(const unsigned i) { return b[i]; }
          const View &block(const unsigned i) const { return b[i]; }
        };
        Vector     flat_src(ls, n3), flat_dst(ls, 3 * nls);
        ViewBlocks vs{{{flat_src.p}, {flat_src.p + nls}, {flat_src.p + 2 * nls}}}, vd{{{flat_dst.p}, {flat_dst.p + nls}, {flat_dst.p + 2 * nls}}};
        const ComputeNormalMatrix<ViewBlocks> normal_flat(ls);
        normal_flat.vmult(vd, vs);
        compare("ComputeNormalMatrix::vmult (contiguous blocks)", flat_dst.to_host(), get(c, "ls_normal"));
      }
      std::printf("operators: %d checks, %d failed, largest difference %.2e\n", n_checked, n_failed, worst);
      std::printf(n_failed == 0 ? "operators: OK\n" : "operators: FAILED\n");
      adaflo_ctx_destroy(ls);
      adaflo_ctx_destroy(ctx);
      return n_failed == 0 ? 0 : 1;
    }
  catch (const std::exception &e)
    {
      std::fprintf(stderr, "operators: %s\n", e.what());
      if (ls)
        adaflo_ctx_destroy(ls);
      if (ctx)
        adaflo_ctx_destroy(ctx);
      return 2;
    }
}
