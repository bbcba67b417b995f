// ns_generic.hip -- generic-degree Navier-Stokes cell kernels (one workgroup per cell).
//
// Restates adaflo::NavierStokesMatrix<dim>::local_operation
// (source/navier_stokes_matrix.cc:601-916) and the scalar sub-block kernels
// local_divergence / local_pressure_* (:920-1140) for a structured brick, any
// Taylor-Hood degree k in {2..5}.  The specialised Q2/Q1 streaming kernel lives
// in ns_q2.hip; this file is the reference-shaped fallback for every other
// degree and for the operators that are not on the headline path.
#include "kernels.hpp"

namespace adaflo_hip
{
  // ZF: flat third direction (dim = 2), see SumFac in fe_kernels.hpp
  template <int K, int ZF = 0>
  struct NSLayout
  {
    static constexpr int NDU = K + 1, NDP = K, NQ = K + 1;
    static constexpr int NDU3 = NDU * (ZF == 2 ? 1 : NDU) * (ZF ? 1 : NDU), NDP3 = NDP * (ZF == 2 ? 1 : NDP) * (ZF ? 1 : NDP),
                         NQ3 = NQ * (ZF == 2 ? 1 : NQ) * (ZF ? 1 : NQ);
    static constexpr int TAB = 2 * NQ * NDU + 2 * NQ * NDP + NQ;
    static constexpr int TABP = (TAB + 1) & ~1;
    // Q2: the three velocity components go through the sum factorisation together (a third of the
    // barriers; 128^3 vmult 5.5 -> 4.6 ms).  For k >= 3 the three-fold scratch space costs more
    // occupancy than the barriers cost time (measured 64^3 Q4: 2.5 -> 3.4 ms), so one component at a time.
    static constexpr int NBATCH = K == 2 ? 3 : 1;
    // k >= 3: the kernel is bound by the number of cells a CU can hold in LDS (Q4: 22 KB per cell).
    // One thread per quadrature point keeps values and gradients of all components (and of the
    // old solutions in the residual) in REGISTERS between evaluate and integrate; only the 1D
    // sweeps go through LDS, one component at a time: Q4 14 KB per cell.
    // PMC profile of the Q4 kernel after this change: the LDS pipeline is busy 82 % of the time and
    // half of the LDS reads are 1D matrix entries.  Keeping the rows / columns a thread needs in
    // registers (one pass per stage) halves the LDS instructions but costs 120 VGPRs = half the
    // occupancy: measured Q3 1.04 -> 1.21 ms, Q4 2.38 -> 2.43 ms, Q5 2.26 -> 2.98 ms; not kept.
    static constexpr bool REGQ = K >= 3;
  };

  template <int K, int NT, int ZF = 0>
  constexpr size_t ns_lds_doubles(const bool residual)
  {
    using L = NSLayout<K, ZF>;
    if (L::REGQ || residual)
      return L::TABP + 3 * L::NDU3 + L::NDP3 + 4 * L::NQ3 + SumFac<L::NDU, L::NQ, NT, ZF>::TMP;
    size_t n = L::TABP + 3 * L::NDU3 + L::NDP3 + 3 * L::NQ3 + 9 * L::NQ3 + L::NQ3 +
               L::NBATCH * SumFac<L::NDU, L::NQ, NT, ZF>::TMP;
    if (residual)
      n += 2 * (3 * L::NQ3 + 9 * L::NQ3);
    return n;
  }

  template <int K, int OP, int NT, int ZF = 0>
  __global__ __launch_bounds__(NT) void ns_cell_kernel(const NSArgs a)
  {
    using L   = NSLayout<K, ZF>;
    using SFU = SumFac<L::NDU, L::NQ, NT, ZF>;
    using SFP = SumFac<L::NDP, L::NQ, NT, ZF>;
    constexpr int NQ = L::NQ, NQ3 = L::NQ3, NDU3 = L::NDU3, NDP3 = L::NDP3;
    constexpr bool RES = OP == OP_RESIDUAL;

    extern __shared__ double lds[];
    double *S_u = lds, *D_u = S_u + NQ * L::NDU, *S_p = D_u + NQ * L::NDU, *D_p = S_p + NQ * L::NDP,
           *wq = D_p + NQ * L::NDP;
    constexpr bool REGQ = L::REGQ || RES; // (Q2 residual: three solutions at the q-points -> registers as well)
    // REGQ: [ul | pl | qb (4 NQ3: one component's tested value + gradient) | tmp]; vu / gu / vp unused
    double *ul = lds + L::TABP, *pl = ul + 3 * NDU3, *vu = pl + NDP3, *gu = vu + (REGQ ? 1 : 3) * NQ3,
           *vp = gu + (REGQ ? 3 : 9) * NQ3, *tmp = REGQ ? vp : vp + NQ3;
    double *qb = vu;
    double  rv[3] = {0., 0., 0.}, rg[9], rp = 0.;                  // REGQ: this thread's quadrature point
    double  rvo[3] = {0., 0., 0.}, rgo[9], rvoo[3] = {0., 0., 0.}, rgoo[9];
    for (int i = 0; i < 9; ++i)
      rg[i] = rgo[i] = rgoo[i] = 0.;
    double *vo = tmp + L::NBATCH * SFU::TMP, *go = vo + 3 * NQ3, *voo = go + 9 * NQ3, *goo = voo + 3 * NQ3;
    auto evaluate_u_reg = [&](const double *u, double *val, double *grad) {
      if constexpr (REGQ)
        for (int d = 0; d < 3; ++d)
          SFU::template evaluate_to_registers<true>(S_u, D_u, u + d * NDU3, tmp, val[d], grad[3 * d], grad[3 * d + 1],
                                                    grad[3 * d + 2]);
    };
    auto evaluate_u = [&](const double *u, double *val, double *grad) {
      if constexpr (L::NBATCH == 3)
        SFU::template evaluate_batch<3>(S_u, D_u, u, val, grad, tmp);
      else
        for (int d = 0; d < 3; ++d)
          SFU::template evaluate<true, true>(S_u, D_u, u + d * NDU3, val + d * NQ3, grad + (3 * d + 0) * NQ3,
                                             grad + (3 * d + 1) * NQ3, grad + (3 * d + 2) * NQ3, tmp);
    };

    const int tid = threadIdx.x;
    for (int o = tid; o < L::TAB; o += NT)
      lds[o] = a.tab[o];

    const long c   = brick_cell(a.brick, blockIdx.x, a.n_cells);
    const int  ncx = a.brick.ncell[0], ncy = a.brick.ncell[1], ncz = a.brick.ncell[2];
    const int  cx = c % ncx, cy = (c / ncx) % ncy, cz = c / ((long)ncx * ncy);
    const int  nux = K * ncx + 1, nuy = ZF == 2 ? 1 : K * ncy + 1, nuz = ZF ? 1 : K * ncz + 1;
    const int  npx = (K - 1) * ncx + 1, npy = ZF == 2 ? 1 : (K - 1) * ncy + 1, npz = ZF ? 1 : (K - 1) * ncz + 1;

    const NSDev &P = a.ns;

    // :662-667 velocity: plain read for the residual, constraints resolved otherwise
    gather_any<K, 3, NT, !RES, ZF>(a.brick, a.src_u, ul, c, cx, cy, cz, nux, nuy, nuz, a.brick.con_u);
    if (OP != OP_VMULT_VELOCITY)
      gather_any<K - 1, 1, NT, !RES, ZF>(a.brick, a.src_p, pl, c, cx, cy, cz, npx, npy, npz, a.brick.con_p);
    __syncthreads();

    // :668-671
    if constexpr (REGQ)
      evaluate_u_reg(ul, rv, rg);
    else
      evaluate_u(ul, vu, gu);
    // :688-697
    if (OP != OP_VMULT_VELOCITY)
      {
        if constexpr (REGQ)
          {
            double d0, d1, d2;
            SFP::template evaluate_to_registers<false>(S_p, D_p, pl, tmp, rp, d0, d1, d2);
          }
        else
          SFP::template evaluate<true, false>(S_p, D_p, pl, vp, nullptr, nullptr, nullptr, tmp);
      }

    // :673-686 old solutions for the residual
    if (RES && P.physical_type == ADAFLO_INCOMPRESSIBLE)
      {
        gather_any<K, 3, NT, false, ZF>(a.brick, a.old_u, ul, c, cx, cy, cz, nux, nuy, nuz, 0u);
        __syncthreads();
        if constexpr (REGQ)
          evaluate_u_reg(ul, rvo, rgo);
        else
          evaluate_u(ul, vo, go);
        gather_any<K, 3, NT, false, ZF>(a.brick, a.oldold_u, ul, c, cx, cy, cz, nux, nuy, nuz, 0u);
        __syncthreads();
        if constexpr (REGQ)
          evaluate_u_reg(ul, rvoo, rgoo);
        else
          evaluate_u(ul, voo, goo);
      }

    // :621-653
    const double w0   = P.physical_type == ADAFLO_INCOMPRESSIBLE ? P.weight : 0.;
    const double tau1 = P.tau1, beta = P.beta;
    const bool   stokes      = P.physical_type == ADAFLO_STOKES;
    const bool   need_extrap = P.linearization == ADAFLO_PROJECTION ||
                             P.linearization == ADAFLO_COUPLED_VELOCITY_SEMI_IMPLICIT ||
                             P.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT;
    const double *const hc = cell_extents(a.brick, c);
    const double ih[3] = {1. / hc[0], 1. / hc[1], 1. / hc[2]};
    const double det   = hc[0] * hc[1] * hc[2];
    double      *lin   = a.lin ? a.lin + (size_t)c * NLIN * NQ3 : nullptr;

    // :702-893 quadrature-point loop
    for (int q = tid; q < NQ3; q += NT)
      {
        const int    qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
        const double jxw = det * wq[qx] * (ZF == 2 ? 1. : wq[qy]) * (ZF ? 1. : wq[qz]);
        double       g[3][3], val[3], conv[3] = {0., 0., 0.};
        for (int d = 0; d < 3; ++d)
          {
            val[d] = REGQ ? rv[d] : vu[d * NQ3 + q];
            for (int e = 0; e < 3; ++e)
              g[d][e] = (REGQ ? rg[3 * d + e] : gu[(3 * d + e) * NQ3 + q]) * ih[e];
          }
        const double div = g[0][0] + g[1][1] + g[2][2];
        if (!stokes)
          {
            const double rho = a.rho ? a.rho[(size_t)c * NQ3 + q] : P.density;
            for (int d = 0; d < 3; ++d)
              conv[d] = val[d] * w0;
            if (RES)
              {
                if (P.physical_type != ADAFLO_INCOMPRESSIBLE_STATIONARY)
                  for (int d = 0; d < 3; ++d)
                    conv[d] += (REGQ ? rvo[d] : vo[d * NQ3 + q]) * P.weight_old +
                               (REGQ ? rvoo[d] : voo[d * NQ3 + q]) * P.weight_old_old;
                if (need_extrap)
                  {
                    double og[3][3], ov[3];
                    for (int d = 0; d < 3; ++d)
                      {
                        for (int e = 0; e < 3; ++e)
                          og[d][e] = ((REGQ ? rgo[3 * d + e] : go[(3 * d + e) * NQ3 + q]) * P.extrap_old +
                                      (REGQ ? rgoo[3 * d + e] : goo[(3 * d + e) * NQ3 + q]) * P.extrap_old_old) * ih[e];
                        ov[d] = (REGQ ? rvo[d] : vo[d * NQ3 + q]) * P.extrap_old +
                                (REGQ ? rvoo[d] : voo[d * NQ3 + q]) * P.extrap_old_old;
                      }
                    const double ediv = og[0][0] + og[1][1] + og[2][2];
                    if (P.linearization == ADAFLO_COUPLED_VELOCITY_EXPLICIT)
                      for (int d = 0; d < 3; ++d)
                        {
                          double res = beta * ediv * ov[d];
                          for (int e = 0; e < 3; ++e)
                            res += ov[e] * og[d][e];
                          conv[d] += tau1 * res;
                        }
                    else
                      {
                        for (int d = 0; d < 3; ++d)
                          {
                            double res = beta * ediv * val[d];
                            for (int e = 0; e < 3; ++e)
                              res += ov[e] * g[d][e];
                            conv[d] += tau1 * res;
                            lin[d * NQ3 + q] = ov[d];
                          }
                        lin[3 * NQ3 + q] = ediv;
                      }
                  }
                else
                  {
                    for (int d = 0; d < 3; ++d)
                      {
                        double res = beta * div * val[d];
                        for (int e = 0; e < 3; ++e)
                          res += val[e] * g[d][e];
                        conv[d] += tau1 * res;
                        lin[d * NQ3 + q] = val[d];
                      }
                    if (P.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON)
                      for (int d = 0; d < 3; ++d)
                        for (int e = 0; e < 3; ++e)
                          lin[(3 + 3 * d + e) * NQ3 + q] = g[d][e];
                    else
                      lin[3 * NQ3 + q] = div;
                  }
              }
            else if (P.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON)
              {
                double lu[3], lg[3][3];
                for (int d = 0; d < 3; ++d)
                  {
                    lu[d] = lin[d * NQ3 + q];
                    for (int e = 0; e < 3; ++e)
                      lg[d][e] = lin[(3 + 3 * d + e) * NQ3 + q];
                  }
                const double f1 = beta * div, f2 = beta * (lg[0][0] + lg[1][1] + lg[2][2]);
                for (int d = 0; d < 3; ++d)
                  {
                    double res = f1 * lu[d] + f2 * val[d];
                    for (int e = 0; e < 3; ++e)
                      res += lu[e] * g[d][e] + val[e] * lg[d][e];
                    conv[d] += tau1 * res;
                  }
              }
            else if (P.linearization != ADAFLO_COUPLED_VELOCITY_EXPLICIT)
              {
                double lu[3];
                for (int d = 0; d < 3; ++d)
                  lu[d] = lin[d * NQ3 + q];
                const double ldiv = lin[3 * NQ3 + q];
                for (int d = 0; d < 3; ++d)
                  {
                    double res = beta * ldiv * val[d];
                    for (int e = 0; e < 3; ++e)
                      res += lu[e] * g[d][e];
                    conv[d] += tau1 * res;
                  }
              }
            const double damping = a.damp ? a.damp[(size_t)c * NQ3 + q] : P.damping;
            for (int d = 0; d < 3; ++d)
              conv[d] = conv[d] * rho - damping * val[d];
          }
        const double tmu = (a.mu ? a.mu[(size_t)c * NQ3 + q] : P.viscosity) * tau1;
        double       pres = 0.;
        if (OP != OP_VMULT_VELOCITY)
          {
            pres = REGQ ? rp : vp[q];
            if (REGQ)
              rp = -div * jxw;
            else
              vp[q] = -div * jxw;
          }
        for (int d = 0; d < 3; ++d)
          for (int e = d + 1; e < 3; ++e)
            {
              const double sym = tmu * (g[d][e] + g[e][d]);
              g[d][e] = g[e][d] = sym;
            }
        for (int d = 0; d < 3; ++d)
          {
            g[d][d] = 2. * tmu * g[d][d] + P.tau_grad_div * div;
            if (OP != OP_VMULT_VELOCITY)
              g[d][d] -= pres;
          }
        for (int d = 0; d < 3; ++d)
          {
            if (REGQ)
              {
                rv[d] = conv[d] * jxw;
                for (int e = 0; e < 3; ++e)
                  rg[3 * d + e] = g[d][e] * (jxw * ih[e]);
              }
            else
              {
                vu[d * NQ3 + q] = conv[d] * jxw; // zero for Stokes: same result as skipping values
                for (int e = 0; e < 3; ++e)
                  gu[(3 * d + e) * NQ3 + q] = g[d][e] * (jxw * ih[e]);
              }
          }
      }
    __syncthreads();

    // :897-907
    if constexpr (REGQ)
      for (int d = 0; d < 3; ++d)
        {
          if (tid < NQ3)
            {
              qb[tid] = rv[d];
              for (int e = 0; e < 3; ++e)
                qb[(1 + e) * NQ3 + tid] = rg[3 * d + e];
            }
          __syncthreads();
          SFU::template integrate<true, true>(S_u, D_u, qb, qb + NQ3, qb + 2 * NQ3, qb + 3 * NQ3, ul + d * NDU3, tmp);
        }
    else if constexpr (L::NBATCH == 3)
      SFU::template integrate_batch<3>(S_u, D_u, vu, gu, ul, tmp);
    else
      for (int d = 0; d < 3; ++d)
        SFU::template integrate<true, true>(S_u, D_u, vu + d * NQ3, gu + (3 * d + 0) * NQ3, gu + (3 * d + 1) * NQ3,
                                            gu + (3 * d + 2) * NQ3, ul + d * NDU3, tmp);
    scatter_any<K, 3, NT, ZF>(a.brick, a.dst_u, ul, c, cx, cy, cz, nux, nuy, nuz, a.brick.con_u);
    if (OP != OP_VMULT_VELOCITY && P.linearization != ADAFLO_PROJECTION)
      {
        if constexpr (REGQ)
          {
            if (tid < NQ3)
              qb[tid] = rp;
            __syncthreads();
            SFP::template integrate<true, false>(S_p, D_p, qb, nullptr, nullptr, nullptr, pl, tmp);
          }
        else
          SFP::template integrate<true, false>(S_p, D_p, vp, nullptr, nullptr, nullptr, pl, tmp);
        scatter_any<K - 1, 1, NT, ZF>(a.brick, a.dst_p, pl, c, cx, cy, cz, npx, npy, npz, a.brick.con_p);
      }
  }

  template <int K, int NT, int ZF>
  static int launch_k(adaflo_ctx *ctx, const int op, const NSArgs &args)
  {
    const dim3   block(NT);
    const size_t lds = sizeof(double) * ns_lds_doubles<K, NT, ZF>(op == OP_RESIDUAL);
    hipError_t   err = hipSuccess;
    NSArgs       a   = args;
    // one launch per colour of the cells (no atomics in the scatter, fe_kernels.hpp)
#define LAUNCH(OPV)                                                                              \
  {                                                                                              \
    if (lds > 64 * 1024)                                                                         \
      err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_cell_kernel<K, OPV, NT, ZF>),     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    for (int colour = 0; colour < n_colours(ctx) && err == hipSuccess; ++colour)                 \
      if (const long nc = cells_of_colour(ctx, a.brick, colour))                                 \
        {                                                                                        \
          set_colour(ctx, a.brick, colour);                                                      \
          hipLaunchKernelGGL((ns_cell_kernel<K, OPV, NT, ZF>), dim3((unsigned)nc), block, lds, ctx->stream, a); \
        }                                                                                        \
  }
    switch (op)
      {
        case OP_VMULT:
          LAUNCH(OP_VMULT);
          break;
        case OP_RESIDUAL:
          LAUNCH(OP_RESIDUAL);
          break;
        case OP_VMULT_VELOCITY:
          LAUNCH(OP_VMULT_VELOCITY);
          break;
        default:
          return ADAFLO_EINVAL;
      }
#undef LAUNCH
    if (err == hipSuccess)
      err = hipGetLastError();
    return err == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_ns_cell_generic(adaflo_ctx *ctx, const int op, const NSArgs &args)
  {
    hipEvent_t stop = ctx->timing ? ctx->kernel_timer.start(ctx->stream) : nullptr;
    int        rc;
    // EXPAND_OPERATIONS, source/navier_stokes_matrix.cc:64-82 (degree_p = k-1)
    if (ctx->flat_y) // dim = 1 (tests/1d_flow*.prm: Q2/Q1)
      switch (ctx->k)
        {
          case 2:
            rc = launch_k<2, 64, 2>(ctx, op, args);
            break;
          case 3:
            rc = launch_k<3, 64, 2>(ctx, op, args);
            break;
          default:
            rc = ADAFLO_EUNSUPPORTED;
        }
    else if (ctx->flat) // dim = 2 (degrees of the reference's 2D tests: Q2/Q1 and Q3/Q2)
      switch (ctx->k)
        {
          case 2:
            rc = launch_k<2, 64, 1>(ctx, op, args);
            break;
          case 3:
            rc = launch_k<3, 64, 1>(ctx, op, args);
            break;
          default:
            rc = ADAFLO_EUNSUPPORTED;
        }
    else
      switch (ctx->k)
        {
          case 2:
            rc = launch_k<2, 64, 0>(ctx, op, args);
            break;
          case 3:
            rc = launch_k<3, 64, 0>(ctx, op, args);
            break;
          case 4:
            rc = launch_k<4, 128, 0>(ctx, op, args);
            break;
          case 5:
            rc = launch_k<5, 256, 0>(ctx, op, args);
            break;
          case 6: // degree_p = 5, the last instance of EXPAND_OPERATIONS (:80-81): 343 quadrature points per cell
            rc = launch_k<6, 384, 0>(ctx, op, args);
            break;
          default:
            rc = ADAFLO_EUNSUPPORTED;
        }
    if (stop)
      (void)hipEventRecord(stop, ctx->stream);
    ctx->kernel_timer.count++;
    return rc;
  }

  // ------------------------------------------------------------------------
  // diagonal of the velocity block
  // ------------------------------------------------------------------------
  // diag[(node, c)] = a(phi_i e_c, phi_i e_c) of the operator of OP_VMULT_VELOCITY (the Jacobi diagonal of
  // the inner velocity solves; the reference takes it from the assembled preconditioner matrix,
  // source/navier_stokes_preconditioner.cc:135-300).  The quadrature-point operation of ns_cell_kernel
  // applied to the trial function u = phi_i e_c -- values delta_dc v, gradients delta_dc grad_e -- and
  // tested with the same function gives, per point,
  //   jxw [ conv_c v + tmu |grad|^2 + (tmu + tau_grad_div) grad_c^2 ],
  //   conv_c = rho (w0 v + tau1 res_c) - damping v,
  //   res_c  = beta grad_c lu_c + beta tr(lg) v + lu . grad + v lg_cc      (Newton)
  //          = beta ldiv v + lu . grad                                     (convection frozen)
  // One thread per node of the cell, the cell's coefficients staged in LDS; 27^2 x ~40 flops per Q2 cell
  // instead of the 81 operator applications of the coloured probing (krylov.hip).
  template <int K, int NT, int ZF = 0>
  __global__ __launch_bounds__(NT) void ns_velocity_diagonal_kernel(const NSArgs a, double *__restrict__ diag)
  {
    using L = NSLayout<K, ZF>;
    constexpr int NQ = L::NQ, NQ3 = L::NQ3, ND = L::NDU, ND3 = L::NDU3, NCF = NLIN + 3;
    extern __shared__ double lds[];
    const double *S_u = lds, *D_u = S_u + NQ * ND, *wq = D_u + NQ * ND + 2 * NQ * L::NDP;
    double       *cf  = lds + L::TABP; // [NCF][NQ3]: lin state (12), rho, damping, tau1 mu
    const int     tid = threadIdx.x;
    for (int o = tid; o < L::TAB; o += NT)
      lds[o] = a.tab[o];
    const long   c   = brick_cell(a.brick, blockIdx.x, a.n_cells);
    const int    ncx = a.brick.ncell[0], ncy = a.brick.ncell[1], ncz = a.brick.ncell[2];
    const int    cx = c % ncx, cy = (c / ncx) % ncy, cz = c / ((long)ncx * ncy);
    const int    nux = K * ncx + 1, nuy = ZF == 2 ? 1 : K * ncy + 1, nuz = ZF ? 1 : K * ncz + 1;
    const NSDev &P = a.ns;
    const bool   stokes = P.physical_type == ADAFLO_STOKES;
    const bool   newton = P.linearization == ADAFLO_COUPLED_IMPLICIT_NEWTON;
    const bool   frozen = !newton && P.linearization != ADAFLO_COUPLED_VELOCITY_EXPLICIT;
    for (int o = tid; o < NCF * NQ3; o += NT)
      {
        const int f = o / NQ3, q = o - f * NQ3;
        double    v;
        if (f < NLIN)
          v = (a.lin && !stokes && (newton || frozen)) ? a.lin[(size_t)c * NLIN * NQ3 + o] : 0.;
        else if (f == NLIN)
          v = a.rho ? a.rho[(size_t)c * NQ3 + q] : P.density;
        else if (f == NLIN + 1)
          v = a.damp ? a.damp[(size_t)c * NQ3 + q] : P.damping;
        else
          v = (a.mu ? a.mu[(size_t)c * NQ3 + q] : P.viscosity) * P.tau1;
        cf[o] = v;
      }
    __syncthreads();
    const double w0 = P.physical_type == ADAFLO_INCOMPRESSIBLE ? P.weight : 0.;
    const double *const hc = cell_extents(a.brick, c);
    const double ih[3] = {1. / hc[0], 1. / hc[1], 1. / hc[2]};
    const double det   = hc[0] * hc[1] * hc[2];
    for (int l = tid; l < ND3; l += NT)
      {
        constexpr int NDY = ZF == 2 ? 1 : ND, NQY = ZF == 2 ? 1 : NQ;
        const int     i = l % ND, j = (l / ND) % NDY, k = l / (ND * NDY);
        double        d[3] = {0., 0., 0.};
        for (int qz = 0; qz < (ZF ? 1 : NQ); ++qz)
          for (int qy = 0; qy < NQY; ++qy)
            {
              const double sz_ = ZF ? 1. : S_u[qz * ND + k], dz_ = ZF ? 0. : D_u[qz * ND + k];
              const double sy_ = ZF == 2 ? 1. : S_u[qy * ND + j], dy_ = ZF == 2 ? 0. : D_u[qy * ND + j];
              const double syz = sy_ * sz_;
              const double gy_ = dy_ * sz_ * ih[1], gz_ = sy_ * dz_ * ih[2];
              const double wyz = det * (ZF == 2 ? 1. : wq[qy]) * (ZF ? 1. : wq[qz]);
#pragma unroll
              for (int qx = 0; qx < NQ; ++qx)
                {
                  const int    q = qx + NQ * (qy + NQY * qz);
                  const double sx = S_u[qx * ND + i], v = sx * syz;
                  const double g[3] = {D_u[qx * ND + i] * syz * ih[0], sx * gy_, sx * gz_};
                  const double jxw = wyz * wq[qx], tmu = cf[(NLIN + 2) * NQ3 + q];
                  const double visc = tmu * (g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                  double       common = 0., lu[3] = {0., 0., 0.};
                  if (!stokes)
                    {
                      double res = 0.;
                      if (newton || frozen)
                        {
#pragma unroll
                          for (int e = 0; e < 3; ++e)
                            lu[e] = cf[e * NQ3 + q];
                          res = lu[0] * g[0] + lu[1] * g[1] + lu[2] * g[2];
                          if (newton)
                            res += P.beta * (cf[3 * NQ3 + q] + cf[7 * NQ3 + q] + cf[11 * NQ3 + q]) * v;
                          else
                            res += P.beta * cf[3 * NQ3 + q] * v;
                        }
                      common = (cf[NLIN * NQ3 + q] * (w0 * v + P.tau1 * res) - cf[(NLIN + 1) * NQ3 + q] * v) * v;
                    }
#pragma unroll
                  for (int e = 0; e < 3; ++e)
                    {
                      double t = common + visc + (tmu + P.tau_grad_div) * g[e] * g[e];
                      if (!stokes && newton) // beta grad_c lu_c + v lg_cc
                        t += cf[NLIN * NQ3 + q] * P.tau1 * (P.beta * g[e] * lu[e] + v * cf[(3 + 4 * e) * NQ3 + q]) * v;
                      d[e] += jxw * t;
                    }
                }
            }
        const int  I = cx * K + i, J = cy * K + j, Kz = cz * K + k;
        const long node = I + (long)nux * (J + (long)nuy * Kz);
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (!on_constrained_face(I, J, Kz, nux, nuy, nuz, a.brick.con_u, 3, e))
            diag[node * 3 + e] += d[e];
      }
  }

  template <int K, int NT, int ZF = 0>
  static int launch_diag(adaflo_ctx *ctx, const NSArgs &args, double *diag)
  {
    using L          = NSLayout<K, ZF>;
    const size_t lds = sizeof(double) * (L::TABP + (NLIN + 3) * L::NQ3);
    NSArgs       a   = args;
    for (int colour = 0; colour < 8; ++colour) // (cells of one colour share no node: plain adds)
      if (const long nc = n_cells_of_colour(a.brick.ncell, colour))
        {
          a.brick.colour = colour;
          hipLaunchKernelGGL((ns_velocity_diagonal_kernel<K, NT, ZF>), dim3((unsigned)nc), dim3(NT), lds, ctx->stream, a, diag);
        }
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  // diag must be zero on entry (constrained rows stay zero)
  int launch_ns_velocity_diagonal(adaflo_ctx *ctx, const NSArgs &args, double *diag)
  {
    if (ctx->flat_y)
      return ctx->k == 2 ? launch_diag<2, 64, 2>(ctx, args, diag) :
                           (ctx->k == 3 ? launch_diag<3, 64, 2>(ctx, args, diag) : ADAFLO_EUNSUPPORTED);
    if (ctx->flat)
      return ctx->k == 2 ? launch_diag<2, 64, 1>(ctx, args, diag) :
                           (ctx->k == 3 ? launch_diag<3, 64, 1>(ctx, args, diag) : ADAFLO_EUNSUPPORTED);
    switch (ctx->k)
      {
        case 2:
          return launch_diag<2, 64>(ctx, args, diag);
        case 3:
          return launch_diag<3, 64>(ctx, args, diag);
        case 4:
          return launch_diag<4, 128>(ctx, args, diag);
        case 5:
          return launch_diag<5, 256>(ctx, args, diag);
        case 6:
          return launch_diag<6, 384>(ctx, args, diag);
        default:
          return ADAFLO_EUNSUPPORTED;
      }
  }

  // ------------------------------------------------------------------------
  // scalar sub-block kernels
  // ------------------------------------------------------------------------
  template <int K, bool QU, int ZF = 0>
  struct ScLayout
  {
    static constexpr int NDU = K + 1, NDP = K, NQ = QU ? K + 1 : K;
    static constexpr int NDU3 = NDU * (ZF == 2 ? 1 : NDU) * (ZF ? 1 : NDU), NDP3 = NDP * (ZF == 2 ? 1 : NDP) * (ZF ? 1 : NDP),
                         NQ3 = NQ * (ZF == 2 ? 1 : NQ) * (ZF ? 1 : NQ);
  };

  template <int K, bool QU, int NT, int ZF = 0>
  constexpr size_t sc_lds_doubles()
  {
    using L = ScLayout<K, QU, ZF>;
    return NSLayout<K>::TABP + 3 * L::NDU3 + L::NDP3 + 9 * L::NQ3 + 4 * L::NQ3 +
           SumFac<L::NDU, L::NQ, NT, ZF>::TMP;
  }

  // QU: quadrature = quad_index_u (k+1 points, table layout of NSLayout) else
  // quad_index_p (k points, table [S_pp D_pp w_pp])
  template <int K, bool QU, int NT, int ZF = 0>
  __global__ __launch_bounds__(NT) void ns_scalar_kernel(const ScalarArgs a)
  {
    using L   = ScLayout<K, QU, ZF>;
    using SFU = SumFac<L::NDU, L::NQ, NT, ZF>;
    using SFP = SumFac<L::NDP, L::NQ, NT, ZF>;
    constexpr int NQ = L::NQ, NQ3 = L::NQ3, NDU3 = L::NDU3, NDP3 = L::NDP3;

    extern __shared__ double lds[];
    double *S_u = lds, *D_u = S_u + NQ * L::NDU;
    double *S_p = QU ? D_u + NQ * L::NDU : lds, *D_p = S_p + NQ * L::NDP, *wq = D_p + NQ * L::NDP;
    double *ul = lds + NSLayout<K>::TABP, *pl = ul + 3 * NDU3, *gu = pl + NDP3, *vp = gu + 9 * NQ3,
           *gp = vp + NQ3, *tmp = gp + 3 * NQ3;

    const int tid  = threadIdx.x;
    const int ntab = QU ? NSLayout<K>::TAB : (2 * NQ * L::NDP + NQ);
    for (int o = tid; o < ntab; o += NT)
      lds[o] = a.tab[o];

    const long c   = brick_cell(a.brick, blockIdx.x, a.n_cells);
    const int  ncx = a.brick.ncell[0], ncy = a.brick.ncell[1], ncz = a.brick.ncell[2];
    const int  cx = c % ncx, cy = (c / ncx) % ncy, cz = c / ((long)ncx * ncy);
    const int  nux = K * ncx + 1, nuy = ZF == 2 ? 1 : K * ncy + 1, nuz = ZF ? 1 : K * ncz + 1;
    const int  npx = (K - 1) * ncx + 1, npy = ZF == 2 ? 1 : (K - 1) * ncy + 1, npz = ZF ? 1 : (K - 1) * ncz + 1;
    const NSDev &P = a.ns;
    const double *const hc = cell_extents(a.brick, c);
    const double ih[3] = {1. / hc[0], 1. / hc[1], 1. / hc[2]};
    const double det   = hc[0] * hc[1] * hc[2];
    const int    mode  = a.mode;

    if (mode == SC_DIVERGENCE || mode == SC_DIVERGENCE_VISC)
      {
        // :935-939: plain read for the projection scheme
        if (P.linearization == ADAFLO_PROJECTION)
          gather_any<K, 3, NT, false, ZF>(a.brick, a.src, ul, c, cx, cy, cz, nux, nuy, nuz, 0u);
        else
          gather_any<K, 3, NT, true, ZF>(a.brick, a.src, ul, c, cx, cy, cz, nux, nuy, nuz, a.brick.con_u);
        __syncthreads();
        if (QU)
          for (int d = 0; d < 3; ++d)
            SFU::template evaluate<false, true>(S_u, D_u, ul + d * NDU3, nullptr,
                                                gu + (3 * d + 0) * NQ3, gu + (3 * d + 1) * NQ3,
                                                gu + (3 * d + 2) * NQ3, tmp);
        for (int q = tid; q < NQ3; q += NT)
          {
            const int    qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
            const double jxw = det * wq[qx] * (ZF == 2 ? 1. : wq[qy]) * (ZF ? 1. : wq[qz]);
            const double div = gu[0 * NQ3 + q] * ih[0] + gu[4 * NQ3 + q] * ih[1] + gu[8 * NQ3 + q] * ih[2];
            const double w   = mode == SC_DIVERGENCE_VISC ?
                                 (a.coef_q ? -a.coef_q[(size_t)c * NQ3 + q] : -P.viscosity) :
                                 -1.;
            vp[q] = w * div * jxw;
          }
        __syncthreads();
        SFP::template integrate<true, false>(S_p, D_p, vp, nullptr, nullptr, nullptr, pl, tmp);
        scatter_any<K - 1, 1, NT, ZF>(a.brick, a.dst, pl, c, cx, cy, cz, npx, npy, npz, a.brick.con_p);
        return;
      }

    // pressure -> pressure operators
    const bool need_val  = mode == SC_MASS;
    const bool need_grad = mode == SC_POISSON_VARIABLE || mode == SC_POISSON_CELL || mode == SC_CONVDIFF;
    if (mode == SC_MASS_WEIGHT)
      __syncthreads(); // the 1D tables staged above are read across waves below (k = 5: 125 points, 4 waves)
    else
      {
        gather_any<K - 1, 1, NT, true, ZF>(a.brick, a.src, pl, c, cx, cy, cz, npx, npy, npz, a.brick.con_p);
        __syncthreads();
        if (need_grad)
          SFP::template evaluate<false, true>(S_p, D_p, pl, nullptr, gp, gp + NQ3, gp + 2 * NQ3, tmp);
        else
          SFP::template evaluate<true, false>(S_p, D_p, pl, vp, nullptr, nullptr, nullptr, tmp);
      }
    // per-cell coefficient ("mid-cell" sampling at q-index n_q/2 of quad_index_u,
    // source/navier_stokes_matrix.cc:1016,1057,1124)
    double cell_coef = 1.;
    const size_t mid = (size_t)c * a.nq_u3 + a.nq_u3 / 2;
    if (mode == SC_POISSON_CELL)
      {
        const double rho = a.coef_q ? a.coef_q[mid] : fmin(P.density, P.density + P.density_diff);
        cell_coef = P.physical_type == ADAFLO_INCOMPRESSIBLE_STATIONARY ? 1. : 1. / (P.weight * rho);
      }
    else if (mode == SC_MASS)
      {
        const double mu = a.coef_q ? a.coef_q[mid] : P.viscosity;
        cell_coef = (P.linearization == ADAFLO_PROJECTION ||
                     P.physical_type == ADAFLO_INCOMPRESSIBLE_STATIONARY) ?
                      1. :
                      1. / (mu + P.tau_grad_div);
      }
    else if (mode == SC_CONVDIFF)
      cell_coef = a.coef_q ? a.coef_q[mid] : P.viscosity;

    for (int q = tid; q < NQ3; q += NT)
      {
        const int    qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
        const double jxw = det * wq[qx] * (ZF == 2 ? 1. : wq[qy]) * (ZF ? 1. : wq[qz]);
        if (mode == SC_MASS_WEIGHT)
          vp[q] = jxw;
        else if (need_val)
          vp[q] *= cell_coef * jxw;
        else
          {
            double cf = cell_coef;
            if (mode == SC_POISSON_VARIABLE)
              cf = 1. / (P.weight * a.coef_q[(size_t)c * NQ3 + q]);
            for (int e = 0; e < 3; ++e)
              gp[e * NQ3 + q] *= cf * jxw * ih[e] * ih[e];
          }
      }
    __syncthreads();
    if (need_grad)
      SFP::template integrate<false, true>(S_p, D_p, nullptr, gp, gp + NQ3, gp + 2 * NQ3, pl, tmp);
    else
      SFP::template integrate<true, false>(S_p, D_p, vp, nullptr, nullptr, nullptr, pl, tmp);
    scatter_any<K - 1, 1, NT, ZF>(a.brick, a.dst, pl, c, cx, cy, cz, npx, npy, npz, a.brick.con_p);
  }

  template <int K, int NT, int ZF = 0>
  static int launch_sc(adaflo_ctx *ctx, const ScalarArgs &args)
  {
    const dim3 block(NT);
    ScalarArgs a = args;
    const bool qu = args.mode == SC_DIVERGENCE || args.mode == SC_DIVERGENCE_VISC ||
                    args.mode == SC_POISSON_VARIABLE || args.mode == SC_CONVDIFF;
    hipError_t err = hipSuccess;
    if (qu)
      {
        const size_t lds = sizeof(double) * sc_lds_doubles<K, true, NT, ZF>();
        if (lds > 64 * 1024)
          err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ns_scalar_kernel<K, true, NT, ZF>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int colour = 0; colour < n_colours(ctx) && err == hipSuccess; ++colour) // (no atomics in the scatter, fe_kernels.hpp)
          if (const long nc = cells_of_colour(ctx, a.brick, colour))
            {
              set_colour(ctx, a.brick, colour);
              hipLaunchKernelGGL((ns_scalar_kernel<K, true, NT, ZF>), dim3((unsigned)nc), block, lds, ctx->stream, a);
            }
      }
    else
      {
        const size_t lds = sizeof(double) * sc_lds_doubles<K, false, NT, ZF>();
        for (int colour = 0; colour < n_colours(ctx); ++colour)
          if (const long nc = cells_of_colour(ctx, a.brick, colour))
            {
              set_colour(ctx, a.brick, colour);
              hipLaunchKernelGGL((ns_scalar_kernel<K, false, NT, ZF>), dim3((unsigned)nc), block, lds, ctx->stream, a);
            }
      }
    if (err == hipSuccess)
      err = hipGetLastError();
    return err == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_ns_scalar_generic(adaflo_ctx *ctx, const ScalarArgs &args)
  {
    if (ctx->flat_y)
      return ctx->k == 2 ? launch_sc<2, 64, 2>(ctx, args) : (ctx->k == 3 ? launch_sc<3, 64, 2>(ctx, args) : ADAFLO_EUNSUPPORTED);
    if (ctx->flat)
      return ctx->k == 2 ? launch_sc<2, 64, 1>(ctx, args) : (ctx->k == 3 ? launch_sc<3, 64, 1>(ctx, args) : ADAFLO_EUNSUPPORTED);
    switch (ctx->k)
      {
        case 2:
          return launch_sc<2, 64>(ctx, args);
        case 3:
          return launch_sc<3, 64>(ctx, args);
        case 4:
          return launch_sc<4, 128>(ctx, args);
        case 5:
          return launch_sc<5, 256>(ctx, args);
        case 6:
          return launch_sc<6, 384>(ctx, args);
        default:
          return ADAFLO_EUNSUPPORTED;
      }
  }
} // namespace adaflo_hip
