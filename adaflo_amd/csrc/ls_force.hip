// ls_force.hip -- LevelSetOKZSolver::compute_heaviside and local_compute_force
// (source/level_set_okz.cc:479-540, :317-413; include/adaflo/level_set_base.h:122-144):
// the step between the level-set operators and the Navier-Stokes operator.  It produces
//   * the variable density / viscosity arrays of NavierStokesMatrix at its quadrature points
//     (begin_densities / begin_viscosities, written straight into the engine's stores), and
//   * the surface-tension + gravity right-hand side (v, sigma kappa grad H - g rho e_z).
// Same kernel family as ls_kernels.hip: one workgroup per cell, tensors in LDS, sum factorisation
// with the level-set space FE_Q_iso_Q1(s) evaluated at the (k+1)^3 Gauss points of the velocity
// quadrature, optional interpolation of H onto the pressure space before the gradient
// ("grad pressure compatible", interpolation_concentration_pressure, level_set_base.cc:106-122).
#include "basis.hpp"
#include "kernels.hpp"

#include <vector>

namespace adaflo_hip
{
  namespace
  {
    // include/adaflo/level_set_base.h:122-144
    __device__ double discrete_heaviside(const double xin)
    {
      const bool   pos = xin > 0;
      const double x   = pos ? -xin : xin;
      double       r;
      if (x < -2.)
        r = 0.;
      else if (x < -1.)
        r = 1. / 8. * (5. * x + x * x) + 1. / 32. * (-3. - 2. * x) * sqrt(-7. - 12. * x - 4. * x * x) -
            1. / 16 * asin(sqrt(2.) * (x + 3. / 2.)) + 23. / 32. - M_PI / 64.;
      else
        r = 1. / 8. * (3. * x + x * x) - 1. / 32. * (-1. - 2. * x) * sqrt(1. - 4. * x - 4. * x * x) +
            1. / 16 * asin(sqrt(2.) * (x + 1. / 2.)) + 15. / 32. - M_PI / 64.;
      return pos ? 1. - r : r;
    }

    // one thread per cell: is some |phi| of the cell below tanh(2)?  (:493-503)
    __global__ __launch_bounds__(256) void heaviside_flag_kernel(unsigned char *__restrict__ flag,
                                                                 const double *__restrict__ phi, const int s,
                                                                 const int ncx, const int ncy, const long n_cells,
                                                                 const int flat)
    {
      const double cutoff = tanh(2.);
      const int    nx = s * ncx + 1, ny = s * ncy + 1, sz = flat ? 0 : s; // (dim = 2: one node layer)
      for (long c = blockIdx.x * 256L + threadIdx.x; c < n_cells; c += (long)gridDim.x * 256)
        {
          const int cx = (int)(c % ncx), cy = (int)((c / ncx) % ncy), cz = (int)(c / ((long)ncx * ncy));
          bool      consider = false;
          for (int k = 0; k <= sz && !consider; ++k)
            for (int j = 0; j <= s && !consider; ++j)
              for (int i = 0; i <= s; ++i)
                if (fabs(phi[(cx * s + i) + (long)nx * ((cy * s + j) + (long)ny * (cz * s + k))]) < cutoff)
                  {
                    consider = true;
                    break;
                  }
          flag[c] = consider;
        }
    }

    // one thread per node (:504-533); nodes shared between a considered and a non-considered
    // cell take the value of the considered one (the reference lets the last cell of its loop win;
    // both agree whenever 6 epsilon / s >= 2)
    __global__ __launch_bounds__(256) void heaviside_node_kernel(double *__restrict__ heaviside,
                                                                 const double *__restrict__ phi,
                                                                 const unsigned char *__restrict__ flag, const int s,
                                                                 const int ncx, const int ncy, const int ncz,
                                                                 const double epsilon, const long n_nodes)
    {
      const double cutoff = tanh(2.);
      const int    nx = s * ncx + 1, ny = s * ncy + 1;
      for (long g = blockIdx.x * 256L + threadIdx.x; g < n_nodes; g += (long)gridDim.x * 256)
        {
          const int I = (int)(g % nx), J = (int)((g / nx) % ny), K = (int)(g / ((long)nx * ny));
          // cells containing the node: index range [floor((I-1)/s) .. I/s] clipped
          bool considered = false;
          // value written by a non-considered cell: sign of ITS first dof (:526-532); the lowest
          // such cell in lexicographic order is the last to be overwritten by nobody else here
          double away = 0.;
          bool   have_away = false;
          for (int cz = max((K - 1) / s - (K == 0 ? 0 : 0), 0); cz <= min(K / s, ncz - 1); ++cz)
            for (int cy = max((J - 1) / s, 0); cy <= min(J / s, ncy - 1); ++cy)
              for (int cx = max((I - 1) / s, 0); cx <= min(I / s, ncx - 1); ++cx)
                {
                  if (I < cx * s || I > cx * s + s || J < cy * s || J > cy * s + s || K < cz * s || K > cz * s + s) // (dim = 2: K = cz = 0)
                    continue;
                  const long c = cx + (long)ncx * (cy + (long)ncy * cz);
                  if (flag[c])
                    considered = true;
                  else
                    {
                      away      = phi[(cx * s) + (long)nx * ((cy * s) + (long)ny * (cz * s))] < 0 ? 0. : 1.;
                      have_away = true;
                    }
                }
          if (considered)
            {
              const double cv = phi[g];
              double       distance;
              if (cv < -cutoff)
                distance = -3;
              else if (cv > cutoff)
                distance = 3;
              else
                distance = log((1 + cv) / (1 - cv));
              distance *= epsilon * 2. / s;
              heaviside[g] = discrete_heaviside(distance);
            }
          else if (have_away)
            heaviside[g] = away;
        }
    }

    // initialize_mass_matrix_diagonal: FE_Q_iso_Q1(s) with 2-point Gauss per sub-interval integrates
    // phi_i^2 exactly: (h_sub / 3) per adjacent sub-interval and direction
    __global__ __launch_bounds__(256) void ls_mass_diagonal_kernel(double *__restrict__ diag, const int nx,
                                                                   const int ny, const int nz, const double unit,
                                                                   const long n_nodes)
    {
      for (long g = blockIdx.x * 256L + threadIdx.x; g < n_nodes; g += (long)gridDim.x * 256)
        {
          const int I = (int)(g % nx), J = (int)((g / nx) % ny), K = (int)(g / ((long)nx * ny));
          const int cnt = ((I > 0) + (I < nx - 1)) * ((J > 0) + (J < ny - 1)) * (nz == 1 ? 1 : (K > 0) + (K < nz - 1));
          diag[g]       = unit * cnt;
        }
    }

    struct ForceArgs
    {
      BrickDev      brick;
      const double *heaviside, *curvature, *tab;
      double       *dst_u, *rho, *mu;
      double        surface_tension, gravity, density, density_diff, viscosity, viscosity_diff;
      int           on_pressure;
      long          n_cells;
    };

    // ZF: flat third direction (dim = 2), see SumFac in fe_kernels.hpp
    template <int S, int KU, int NT, bool ZF = false>
    struct ForceCfg
    {
      static constexpr int NDL = S + 1, NQ = KU + 1, NDP = KU, NDV = KU + 1;
      static constexpr int NDL3 = NDL * NDL * (ZF ? 1 : NDL), NQ3 = NQ * NQ * (ZF ? 1 : NQ), NDP3 = NDP * NDP * (ZF ? 1 : NDP),
                           NDV3 = NDV * NDV * (ZF ? 1 : NDV);
      // tables: S_l D_l [NQ x NDL] | I1 [NDP x NDL] | S_p D_p [NQ x NDP] | S_v [NQ x NDV] | w [NQ]
      static constexpr int TAB = 2 * NQ * NDL + NDP * NDL + 2 * NQ * NDP + NQ * NDV + NQ, TABP = (TAB + 1) & ~1;
      using SFL = SumFac<NDL, NQ, NT, ZF>;  // level-set space -> Gauss points
      using SFI = SumFac<NDL, NDP, NT, ZF>; // level-set space -> pressure support points
      using SFP = SumFac<NDP, NQ, NT, ZF>;  // pressure space -> Gauss points
      using SFV = SumFac<NDV, NQ, NT, ZF>;  // velocity test functions
      static constexpr int TMP = (SFL::TMP > SFI::TMP ? SFL::TMP : SFI::TMP) > (SFP::TMP > SFV::TMP ? SFP::TMP : SFV::TMP) ?
                                   (SFL::TMP > SFI::TMP ? SFL::TMP : SFI::TMP) :
                                   (SFP::TMP > SFV::TMP ? SFP::TMP : SFV::TMP);
      static constexpr size_t LDS = TABP + 2 * NDL3 + NDP3 + 5 * NQ3 + 3 * NDV3 + TMP;
    };

    template <int S, int KU, int NT, bool ZF = false>
    __global__ __launch_bounds__(NT) void ls_force_kernel(const ForceArgs a)
    {
      using C = ForceCfg<S, KU, NT, ZF>;
      constexpr int NQ = C::NQ, NQ3 = C::NQ3, NDL = C::NDL, NDP = C::NDP, NDV = C::NDV;
      extern __shared__ double lds[];
      double *Sl = lds, *Dl = Sl + NQ * NDL, *I1 = Dl + NQ * NDL, *Sp = I1 + NDP * NDL, *Dp = Sp + NQ * NDP,
             *Sv = Dp + NQ * NDP, *wq = Sv + NQ * NDV;
      double *hl = lds + C::TABP, *cl = hl + C::NDL3, *pl = cl + C::NDL3, *hv = pl + C::NDP3, *hg = hv + NQ3,
             *cv = hg + 3 * NQ3, *vl = cv + NQ3, *tmp = vl + 3 * C::NDV3;
      const int tid = threadIdx.x;
      for (int o = tid; o < C::TAB; o += NT)
        lds[o] = a.tab[o];
      const long c   = brick_cell(a.brick, blockIdx.x, a.n_cells);
      const int  ncx = a.brick.ncell[0], ncy = a.brick.ncell[1], ncz = a.brick.ncell[2];
      const int  cx = c % ncx, cy = (c / ncx) % ncy, cz = c / ((long)ncx * ncy);
      const int  nx = S * ncx + 1, ny = S * ncy + 1, nz = ZF ? 1 : S * ncz + 1;
      const int  vx = KU * ncx + 1, vy = KU * ncy + 1, vz = ZF ? 1 : KU * ncz + 1;
      const double ih[3] = {1. / a.brick.h[0], 1. / a.brick.h[1], 1. / a.brick.h[2]};
      const double det   = a.brick.h[0] * a.brick.h[1] * a.brick.h[2];

      gather_cell<S, 1, NT, false, ZF>(a.heaviside, hl, cx, cy, cz, nx, ny, nz, 0u); // read_dof_values_plain :348
      gather_cell<S, 1, NT, false, ZF>(a.curvature, cl, cx, cy, cz, nx, ny, nz, 0u); // :384
      __syncthreads();
      // H and (unless taken from the pressure space) its gradient, curvature values
      C::SFL::template evaluate<true, true>(Sl, Dl, hl, hv, hg, hg + NQ3, hg + 2 * NQ3, tmp);
      C::SFL::template evaluate<true, false>(Sl, Dl, cl, cv, nullptr, nullptr, nullptr, tmp);
      if (a.on_pressure) // :367-380
        {
          C::SFI::template evaluate<true, false>(I1, I1, hl, pl, nullptr, nullptr, nullptr, tmp);
          C::SFP::template evaluate<false, true>(Sp, Dp, pl, nullptr, hg, hg + NQ3, hg + 2 * NQ3, tmp);
        }
      const bool variable = a.rho != nullptr;
      for (int q = tid; q < NQ3; q += NT)
        {
          const int    qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
          const double jxw = det * wq[qx] * wq[qy] * (ZF ? 1. : wq[qz]);
          double       rho = a.density;
          if (variable) // :352-365
            {
              rho                        = a.density + a.density_diff * hv[q];
              a.rho[(size_t)c * NQ3 + q] = rho;
              a.mu[(size_t)c * NQ3 + q]  = a.viscosity + a.viscosity_diff * hv[q];
            }
          const double sk = a.surface_tension * cv[q]; // :391-397
          // gravity acts on component dim - 1 (:400-404)
          hg[q]           = sk * hg[q] * ih[0] * jxw;
          hg[NQ3 + q]     = (sk * hg[NQ3 + q] * ih[1] - (ZF ? a.gravity * rho : 0.)) * jxw;
          hg[2 * NQ3 + q] = ZF ? 0. : (sk * hg[2 * NQ3 + q] * ih[2] - a.gravity * rho) * jxw;
        }
      __syncthreads();
      for (int d = 0; d < 3; ++d)
        C::SFV::template integrate<true, false>(Sv, Sv, hg + d * NQ3, nullptr, nullptr, nullptr, vl + d * C::NDV3, tmp);
      scatter_cell<KU, 3, NT, ZF>(a.dst_u, vl, cx, cy, cz, vx, vy, vz, a.brick.con_u, a.brick.colour);
    }

    template <int S, int KU, bool ZF = false>
    int launch_force_sk(adaflo_ctx *ctx, const ForceArgs &args)
    {
      constexpr int NT = 128;
      using C          = ForceCfg<S, KU, NT, ZF>;
      const size_t lds = sizeof(double) * C::LDS;
      hipError_t   err = hipSuccess;
      if (lds > 64 * 1024)
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ls_force_kernel<S, KU, NT, ZF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      ForceArgs a = args;
      for (int colour = 0; colour < 8 && err == hipSuccess; ++colour) // (no atomics in the scatter, fe_kernels.hpp)
        if (const long nc = n_cells_of_colour(a.brick.ncell, colour))
          {
            a.brick.colour = colour;
            hipLaunchKernelGGL((ls_force_kernel<S, KU, NT, ZF>), dim3((unsigned)nc), dim3(NT), lds, ctx->stream, a);
          }
      if (err == hipSuccess)
        err = hipGetLastError();
      return err == hipSuccess ? 0 : ADAFLO_EHIP;
    }

    template <int S>
    int launch_force_s(adaflo_ctx *ctx, const ForceArgs &a)
    {
      if (ctx->flat)
        return ctx->k == 2 ? launch_force_sk<S, 2, true>(ctx, a) : (ctx->k == 3 ? launch_force_sk<S, 3, true>(ctx, a) : ADAFLO_EUNSUPPORTED);
      switch (ctx->k)
        {
          case 2:
            return launch_force_sk<S, 2>(ctx, a);
          case 3:
            return launch_force_sk<S, 3>(ctx, a);
          case 4:
            return launch_force_sk<S, 4>(ctx, a);
          case 5: // (level_set_okz_template_instantations.h:32-80: degree_u 2..5 with every ls_degree 1..4)
            return launch_force_sk<S, 5>(ctx, a);
          default:
            return ADAFLO_EUNSUPPORTED;
        }
    }
  } // namespace

  int launch_ls_heaviside(adaflo_ctx *ctx, double *heaviside, const double *phi, const double epsilon)
  {
    const int  s = ctx->s, ncx = ctx->desc.ncell[0], ncy = ctx->desc.ncell[1], ncz = ctx->desc.ncell[2];
    const long n_cells = ctx->n_cells, n_nodes = ctx->n_nodes_ls;
    unsigned char *flag = nullptr;
    if (hipMalloc(&flag, (size_t)n_cells) != hipSuccess)
      return ADAFLO_ENOMEM;
    long nb = (n_cells + 255) / 256;
    hipLaunchKernelGGL(heaviside_flag_kernel, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, ctx->stream, flag,
                       phi, s, ncx, ncy, n_cells, ctx->flat ? 1 : 0);
    nb = (n_nodes + 255) / 256;
    hipLaunchKernelGGL(heaviside_node_kernel, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, ctx->stream,
                       heaviside, phi, flag, s, ncx, ncy, ncz, epsilon, n_nodes);
    const hipError_t e1 = hipStreamSynchronize(ctx->stream), e2 = hipGetLastError();
    (void)hipFree(flag);
    return (e1 == hipSuccess && e2 == hipSuccess) ? 0 : ADAFLO_EHIP;
  }

  // level_set_okz_compute_curvature.cc:360-376: extend the curvature along the normal direction to
  // the value at the interface, 1 / (1 / kappa + distance / (dim - 1))
  __global__ void curvature_correction_kernel(double *__restrict__ kappa, const double *__restrict__ phi,
                                              const double epsilon_used, const long n, const double dim_minus_1)
  {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
      {
        const double kv = kappa[i];
        if (kv > 1e-4)
          {
            const double c        = phi[i];
            const double distance = (1 - c * c) > 1e-2 ? epsilon_used * log((1. + c) / (1. - c)) : 0.;
            kappa[i]              = 1. / (1. / kv + distance / dim_minus_1);
          }
      }
  }

  int launch_ls_curvature_correction(adaflo_ctx *ctx, double *curvature, const double *phi)
  {
    const long n  = ctx->n_nodes_ls;
    const long nb = (n + 255) / 256;
    hipLaunchKernelGGL(curvature_correction_kernel, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, ctx->stream,
                       curvature, phi, ctx->ls.epsilon_used, n, ctx->flat ? 1. : 2.);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_ls_mass_diagonal(adaflo_ctx *ctx, double *diag)
  {
    const int  s = ctx->s, nx = s * ctx->desc.ncell[0] + 1, ny = s * ctx->desc.ncell[1] + 1,
              nz = ctx->flat ? 1 : s * ctx->desc.ncell[2] + 1;
    const long n = ctx->n_nodes_ls;
    const double unit = (ctx->desc.h[0] / s / 3.) * (ctx->desc.h[1] / s / 3.) * (ctx->flat ? 1. : ctx->desc.h[2] / s / 3.);
    long nb = (n + 255) / 256;
    hipLaunchKernelGGL(ls_mass_diagonal_kernel, dim3((unsigned)(nb > 65536 ? 65536 : nb)), dim3(256), 0, ctx->stream, diag,
                       nx, ny, nz, unit, n);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  // tables of the force kernel for (s, k); host vector in the layout of ForceCfg
  std::vector<double> force_tables(const int s, const int k)
  {
    const Quadrature1D  qu = gauss(k + 1);
    const Shape1D       sl = shape_fe_q_iso_q1(s, qu), sp = shape_fe_q(k - 1, qu), sv = shape_fe_q(k, qu);
    Quadrature1D        pts; // support points of the pressure element as "quadrature"
    pts.x = k - 1 == 0 ? std::vector<double>{0.5} : gauss_lobatto_points(k - 1);
    pts.w.assign(pts.x.size(), 0.);
    const Shape1D       in = shape_fe_q_iso_q1(s, pts);
    std::vector<double> t;
    t.insert(t.end(), sl.S.begin(), sl.S.end());
    t.insert(t.end(), sl.D.begin(), sl.D.end());
    t.insert(t.end(), in.S.begin(), in.S.end());
    t.insert(t.end(), sp.S.begin(), sp.S.end());
    t.insert(t.end(), sp.D.begin(), sp.D.end());
    t.insert(t.end(), sv.S.begin(), sv.S.end());
    t.insert(t.end(), qu.w.begin(), qu.w.end());
    return t;
  }

  int launch_ls_force(adaflo_ctx *ctx, double *dst_u, const double *heaviside, const double *curvature,
                      const double *tab, double *rho, double *mu, const double surface_tension, const double gravity,
                      const double density, const double density_diff, const double viscosity,
                      const double viscosity_diff, const int on_pressure)
  {
    ForceArgs a{};
    a.brick           = ctx->brick;
    a.heaviside       = heaviside;
    a.curvature       = curvature;
    a.tab             = tab;
    a.dst_u           = dst_u;
    a.rho             = rho;
    a.mu              = mu;
    a.surface_tension = surface_tension;
    a.gravity         = gravity;
    a.density         = density;
    a.density_diff    = density_diff;
    a.viscosity       = viscosity;
    a.viscosity_diff  = viscosity_diff;
    a.on_pressure     = on_pressure;
    a.n_cells         = ctx->n_cells;
    switch (ctx->s)
      {
        case 1:
          return launch_force_s<1>(ctx, a);
        case 2:
          return launch_force_s<2>(ctx, a);
        case 3:
          return launch_force_s<3>(ctx, a);
        case 4:
          return launch_force_s<4>(ctx, a);
        default:
          return ADAFLO_EUNSUPPORTED;
      }
  }
} // namespace adaflo_hip
