// ls_kernels.hip -- level-set (Olsson-Kreiss-Zahedi) operator cell kernels.
//
// Restates the local_* kernels of adaflo's four level-set operators for a
// structured brick (FE_Q_iso_Q1(s), quadrature QIterated(QGauss<1>(2), s),
// source/two_phase_base.cc:267-268, source/level_set_base.cc:58-59):
//   advection       source/level_set_okz_advance_concentration.cc:217-258 (vmult), :288-397 (rhs)
//   reinitialization source/level_set_okz_reinitialization.cc:53-106 (vmult), :128-189 (rhs)
//   normal           source/level_set_okz_compute_normal.cc:82-119 (vmult), :123-156 (rhs)
//   curvature        source/level_set_okz_compute_curvature.cc:86-133 (vmult), :212-259 (rhs)
// One workgroup per cell, tensors in LDS (SumFac of fe_kernels.hpp).  The optional
// `convection stabilization` of the advection operator (advance_concentration.cc:248-249 cell
// term, :344-369 artificial viscosity, :387-388 rhs term, :419-472 / :569-617 boundary term,
// :39-68 maximal velocity) is here as well.
#include "kernels.hpp"

namespace adaflo_hip
{
  enum LSMode
  {
    LS_ADVECT = 0,
    LS_REINIT,
    LS_REINIT_DIFFUSE,
    LS_NORMAL,
    LS_CURVATURE
  };

  struct LSArgs
  {
    BrickDev      brick;
    LSDev         ls;
    const double *src, *src2, *src3, *vel; // src2/src3: old solutions (advect rhs)
    double       *dst;
    double       *qstate;                  // [cell][3][q] evaluated_convection / evaluated_normal
    const double *diag;
    const double *tab;                     // [S D w] of the LS space, then [S_vel] for advect rhs
    long          n_cells, n_nodes;
    int           mode, flag;
    // convection stabilisation: artificial_viscosities [cell] (read by the operator, written by
    // the advection rhs), nodal sums u_old + u_old_old / phi_old + phi_old_old, 1 / old step size,
    // global_max_velocity * 2 * global_omega_diameter, symmetry faces, sign of the boundary term
    double       *art_visc;
    const double *vel_sum, *ls_sum;
    double        old_step_inv, global_scaling, bsign;
    unsigned      symmetry;
  };

  // ZF: flat third direction (dim = 2), see SumFac in fe_kernels.hpp
  template <int S, int NT, bool ZF = false>
  constexpr size_t ls_lds_doubles(const int extra)
  {
    constexpr int ND = S + 1, NQ = 2 * S;
    return ((2 * NQ * ND + NQ + 1) & ~1) + ND * ND * (ZF ? 1 : ND) + 4 * NQ * NQ * (ZF ? 1 : NQ) + SumFac<ND, NQ, NT, ZF>::TMP + extra;
  }

  template <int S, bool ZF = false>
  struct LSCell
  {
    static constexpr int ND = S + 1, NQ = 2 * S, ND3 = ND * ND * (ZF ? 1 : ND), NQ3 = NQ * NQ * (ZF ? 1 : NQ);
    static constexpr int TAB = 2 * NQ * ND + NQ, TABP = (TAB + 1) & ~1;
  };

  // ------------------------------------------------------------------ operator applications
  template <int S, int NT, bool ZF = false>
  __global__ __launch_bounds__(NT) void ls_vmult_kernel(const LSArgs a)
  {
    using C  = LSCell<S, ZF>;
    using SF = SumFac<C::ND, C::NQ, NT, ZF>;
    constexpr int NQ = C::NQ, NQ3 = C::NQ3;
    extern __shared__ double lds[];
    double *Sm = lds, *Dm = Sm + NQ * C::ND, *wq = Dm + NQ * C::ND;
    double *ul = lds + C::TABP, *val = ul + C::ND3, *grad = val + NQ3, *tmp = grad + 3 * NQ3;

    const int tid = threadIdx.x;
    for (int o = tid; o < C::TAB; o += NT)
      lds[o] = a.tab[o];
    const long c   = brick_cell(a.brick, blockIdx.x, a.n_cells);
    const int  ncx = a.brick.ncell[0], ncy = a.brick.ncell[1], ncz = a.brick.ncell[2];
    const int  cx = c % ncx, cy = (c / ncx) % ncy, cz = c / ((long)ncx * ncy);
    const int  nx = S * ncx + 1, ny = S * ncy + 1, nz = ZF ? 1 : S * ncz + 1;
    const int  comp = blockIdx.y; // LS_NORMAL: one scalar block per component
    const double *src = a.src + (size_t)comp * a.n_nodes;
    double       *dst = a.dst + (size_t)comp * a.n_nodes;

    gather_cell<S, 1, NT, true, ZF>(src, ul, cx, cy, cz, nx, ny, nz, a.brick.con_ls);
    __syncthreads();
    SF::template evaluate<true, true>(Sm, Dm, ul, val, grad, grad + NQ3, grad + 2 * NQ3, tmp);

    const LSDev &P     = a.ls;
    const double ih[3] = {1. / a.brick.h[0], 1. / a.brick.h[1], 1. / a.brick.h[2]};
    const double det   = a.brick.h[0] * a.brick.h[1] * a.brick.h[2];
    const double hcell = (ZF ? fmax(a.brick.h[0], a.brick.h[1]) : fmax(a.brick.h[0], fmax(a.brick.h[1], a.brick.h[2]))); // util.h:47-120
    // level_set_okz_reinitialization.cc:65-67,:82-85
    const double dtau_inv  = fmax(0.95 / (1. / (ZF ? 4. : 9.) * P.minimal_edge_length / S), 1. / (5. * P.time_step)); // (1 / dim^2)
    const double diffusion = fmax(P.epsilon_used, hcell / (double)S);
    const double b         = fmax(P.epsilon_used / P.epsilon, hcell / (double)S);
    // compute_normal.cc:107-110 (damping_scale_factor = 4), compute_curvature.cc:112-118
    const double damping = a.mode == LS_NORMAL ? 4. * b * b : (a.flag ? b * b : 0.);
    const double *qs = a.qstate ? a.qstate + (size_t)c * 3 * NQ3 : nullptr;

    for (int q = tid; q < NQ3; q += NT)
      {
        const int    qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
        const double jxw = det * wq[qx] * wq[qy] * (ZF ? 1. : wq[qz]);
        const double v   = val[q];
        double       g[3];
        for (int e = 0; e < 3; ++e)
          g[e] = grad[e * NQ3 + q] * ih[e];
        double tv = 0., tg[3] = {0., 0., 0.};
        switch (a.mode)
          {
            case LS_ADVECT: // :244-249
              tv = v * P.weight + qs[q] * g[0] + qs[NQ3 + q] * g[1] + qs[2 * NQ3 + q] * g[2];
              if (a.art_visc)
                for (int e = 0; e < 3; ++e)
                  tg[e] = a.art_visc[c] * g[e];
              break;
            case LS_REINIT: // :88-95
              {
                const double n0 = qs[q], n1 = qs[NQ3 + q], n2 = qs[2 * NQ3 + q];
                const double ng = diffusion * (n0 * g[0] + n1 * g[1] + n2 * g[2]);
                tv    = dtau_inv * v;
                tg[0] = ng * n0;
                tg[1] = ng * n1;
                tg[2] = ng * n2;
              }
              break;
            case LS_REINIT_DIFFUSE: // :96-100
              tv = dtau_inv * v;
              for (int e = 0; e < 3; ++e)
                tg[e] = diffusion * g[e];
              break;
            default: // LS_NORMAL / LS_CURVATURE: (w, n) + (grad w, damping grad n)
              tv = v;
              for (int e = 0; e < 3; ++e)
                tg[e] = damping * g[e];
          }
        val[q] = tv * jxw;
        for (int e = 0; e < 3; ++e)
          grad[e * NQ3 + q] = tg[e] * (jxw * ih[e]);
      }
    __syncthreads();
    SF::template integrate<true, true>(Sm, Dm, val, grad, grad + NQ3, grad + 2 * NQ3, ul, tmp);
    scatter_cell<S, 1, NT, ZF>(dst, ul, cx, cy, cz, nx, ny, nz, a.brick.con_ls, a.brick.colour);
  }

  // ------------------------------------------------------------------ right-hand sides
  enum LSRhs
  {
    RHS_REINIT = 0,
    RHS_NORMAL,
    RHS_CURVATURE
  };

  template <int S, int NT, bool ZF = false>
  __global__ __launch_bounds__(NT) void ls_rhs_kernel(const LSArgs a)
  {
    using C  = LSCell<S, ZF>;
    using SF = SumFac<C::ND, C::NQ, NT, ZF>;
    constexpr int NQ = C::NQ, NQ3 = C::NQ3, ND3 = C::ND3;
    extern __shared__ double lds[];
    double *Sm = lds, *Dm = Sm + NQ * C::ND, *wq = Dm + NQ * C::ND;
    double *ul = lds + C::TABP, *val = ul + ND3, *grad = val + NQ3, *tmp = grad + 3 * NQ3;
    double *nl = tmp + SF::TMP; // [3][ND3] normal dofs (curvature) / [3][NQ3] normal values (reinit)

    const int tid = threadIdx.x;
    for (int o = tid; o < C::TAB; o += NT)
      lds[o] = a.tab[o];
    const long c   = brick_cell(a.brick, blockIdx.x, a.n_cells);
    const int  ncx = a.brick.ncell[0], ncy = a.brick.ncell[1], ncz = a.brick.ncell[2];
    const int  cx = c % ncx, cy = (c / ncx) % ncy, cz = c / ((long)ncx * ncy);
    const int  nx = S * ncx + 1, ny = S * ncy + 1, nz = ZF ? 1 : S * ncz + 1;
    const LSDev &P     = a.ls;
    const double ih[3] = {1. / a.brick.h[0], 1. / a.brick.h[1], 1. / a.brick.h[2]};
    const double det   = a.brick.h[0] * a.brick.h[1] * a.brick.h[2];
    const double hcell = (ZF ? fmax(a.brick.h[0], a.brick.h[1]) : fmax(a.brick.h[0], fmax(a.brick.h[1], a.brick.h[2])));
    const double diffusion = fmax(P.epsilon_used, hcell / (double)S);

    if (a.mode == RHS_CURVATURE)
      {
        // :229-259: normalise the normal at the DoFs, rhs = (w, -div(n/|n|))
        int nonzero = 0;
        for (int d = 0; d < 3; ++d)
          gather_cell<S, 1, NT, false, ZF>(a.src + (size_t)d * a.n_nodes, nl + d * ND3, cx, cy, cz, nx, ny, nz, 0u);
        __syncthreads();
        for (int i = tid; i < ND3; i += NT)
          {
            const double n0 = nl[i], n1 = nl[ND3 + i], n2 = nl[2 * ND3 + i];
            const double nr = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
            if (nr > 1e-2)
              {
                nonzero       = 1;
                nl[i]         = n0 / nr;
                nl[ND3 + i]   = n1 / nr;
                nl[2 * ND3 + i] = n2 / nr;
              }
            else
              nl[i] = nl[ND3 + i] = nl[2 * ND3 + i] = 0.;
          }
        if (!__syncthreads_or(nonzero)) // :250
          return;
        for (int q = tid; q < NQ3; q += NT)
          val[q] = 0.;
        for (int d = 0; d < 3; ++d)
          {
            SF::template evaluate<false, true>(Sm, Dm, nl + d * ND3, nullptr, grad, grad + NQ3, grad + 2 * NQ3, tmp);
            for (int q = tid; q < NQ3; q += NT)
              val[q] -= grad[d * NQ3 + q] * ih[d];
            __syncthreads();
          }
        for (int q = tid; q < NQ3; q += NT)
          {
            const int qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
            val[q] *= det * wq[qx] * wq[qy] * (ZF ? 1. : wq[qz]);
          }
        __syncthreads();
        SF::template integrate<true, false>(Sm, Dm, val, nullptr, nullptr, nullptr, ul, tmp);
        scatter_cell<S, 1, NT, ZF>(a.dst, ul, cx, cy, cz, nx, ny, nz, a.brick.con_ls, a.brick.colour);
        return;
      }

    // level-set value and gradient (plain read)
    gather_cell<S, 1, NT, false, ZF>(a.src, ul, cx, cy, cz, nx, ny, nz, 0u);
    __syncthreads();
    SF::template evaluate<true, true>(Sm, Dm, ul, val, grad, grad + NQ3, grad + 2 * NQ3, tmp);

    if (a.mode == RHS_NORMAL)
      {
        // :141-153: (w, grad phi), one block per component
        for (int d = 0; d < 3; ++d)
          {
            for (int q = tid; q < NQ3; q += NT)
              {
                const int qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
                grad[d * NQ3 + q] *= ih[d] * det * wq[qx] * wq[qy] * (ZF ? 1. : wq[qz]);
              }
            __syncthreads();
            SF::template integrate<true, false>(Sm, Dm, grad + d * NQ3, nullptr, nullptr, nullptr, ul, tmp);
            scatter_cell<S, 1, NT, ZF>(a.dst + (size_t)d * a.n_nodes, ul, cx, cy, cz, nx, ny, nz, a.brick.con_ls, a.brick.colour);
            __syncthreads();
          }
        return;
      }

    // RHS_REINIT :128-189; a.flag bit 0 = diffuse_only, bit 1 = first_reinit_step
    const bool diffuse_only = a.flag & 1, first = a.flag & 2;
    double    *nq = a.qstate + (size_t)c * 3 * NQ3;
    if (!diffuse_only && first)
      {
        for (int d = 0; d < 3; ++d)
          {
            gather_cell<S, 1, NT, false, ZF>(a.src2 + (size_t)d * a.n_nodes, ul, cx, cy, cz, nx, ny, nz, 0u);
            __syncthreads();
            SF::template evaluate<true, false>(Sm, Dm, ul, nl + d * NQ3, nullptr, nullptr, nullptr, tmp);
          }
      }
    for (int q = tid; q < NQ3; q += NT)
      {
        const int    qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
        const double jxw = det * wq[qx] * wq[qy] * (ZF ? 1. : wq[qz]);
        double       g[3];
        for (int e = 0; e < 3; ++e)
          g[e] = grad[e * NQ3 + q] * ih[e];
        double tg[3];
        if (!diffuse_only)
          {
            double n[3];
            if (first) // :167-172
              {
                n[0] = nl[q];
                n[1] = nl[NQ3 + q];
                n[2] = nl[2 * NQ3 + q];
                const double sc = fmax(1e-4, sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]));
                for (int e = 0; e < 3; ++e)
                  {
                    n[e] /= sc;
                    nq[e * NQ3 + q] = n[e];
                  }
              }
            else
              for (int e = 0; e < 3; ++e)
                n[e] = nq[e * NQ3 + q];
            const double v = val[q];
            const double f = 0.5 * (1. - v * v) - (n[0] * g[0] + n[1] * g[1] + n[2] * g[2]) * diffusion; // :176-178
            for (int e = 0; e < 3; ++e)
              tg[e] = n[e] * f;
          }
        else
          for (int e = 0; e < 3; ++e)
            tg[e] = -diffusion * g[e];
        for (int e = 0; e < 3; ++e)
          grad[e * NQ3 + q] = tg[e] * (jxw * ih[e]);
      }
    __syncthreads();
    SF::template integrate<false, true>(Sm, Dm, nullptr, grad, grad + NQ3, grad + 2 * NQ3, ul, tmp);
    scatter_cell<S, 1, NT, ZF>(a.dst, ul, cx, cy, cz, nx, ny, nz, a.brick.con_ls, a.brick.colour);
  }

  // advection right-hand side :288-397 (velocity of degree KU evaluated at the LS quadrature)
  template <int S, int KU, int NT, bool ZF = false>
  __global__ __launch_bounds__(NT) void ls_advect_rhs_kernel(const LSArgs a)
  {
    using C   = LSCell<S, ZF>;
    using SF  = SumFac<C::ND, C::NQ, NT, ZF>;
    using SFV = SumFac<KU + 1, C::NQ, NT, ZF>;
    constexpr int NQ = C::NQ, NQ3 = C::NQ3, ND3 = C::ND3, NDV3 = (KU + 1) * (KU + 1) * (ZF ? 1 : KU + 1);
    constexpr int TMPX = SF::TMP > SFV::TMP ? SF::TMP : SFV::TMP;
    extern __shared__ double lds[];
    double *Sm = lds, *Dm = Sm + NQ * C::ND, *wq = Dm + NQ * C::ND;
    double *ul = lds + C::TABP, *val = ul + ND3, *grad = val + NQ3, *tmp = grad + 3 * NQ3;
    double *Sv = tmp + TMPX, *vl = Sv + NQ * (KU + 1), *vq = vl + 3 * NDV3, *vo = vq + 3 * NQ3, *voo = vo + NQ3;
    double *red = voo + NQ3; // [2][NT] reduction scratch of the stabilisation

    const int tid = threadIdx.x;
    for (int o = tid; o < C::TAB; o += NT)
      lds[o] = a.tab[o];
    for (int o = tid; o < NQ * (KU + 1); o += NT)
      Sv[o] = a.tab[C::TAB + o];
    const long c   = brick_cell(a.brick, blockIdx.x, a.n_cells);
    const int  ncx = a.brick.ncell[0], ncy = a.brick.ncell[1], ncz = a.brick.ncell[2];
    const int  cx = c % ncx, cy = (c / ncx) % ncy, cz = c / ((long)ncx * ncy);
    const int  nx = S * ncx + 1, ny = S * ncy + 1, nz = ZF ? 1 : S * ncz + 1;
    const int  vx = KU * ncx + 1, vy = KU * ncy + 1, vz = ZF ? 1 : KU * ncz + 1;
    const LSDev &P     = a.ls;
    const double ih[3] = {1. / a.brick.h[0], 1. / a.brick.h[1], 1. / a.brick.h[2]};
    const double det   = a.brick.h[0] * a.brick.h[1] * a.brick.h[2];

    // old level-set values
    gather_cell<S, 1, NT, false, ZF>(a.src2, ul, cx, cy, cz, nx, ny, nz, 0u);
    __syncthreads();
    SF::template evaluate<true, false>(Sm, Dm, ul, vo, nullptr, nullptr, nullptr, tmp);
    gather_cell<S, 1, NT, false, ZF>(a.src3, ul, cx, cy, cz, nx, ny, nz, 0u);
    __syncthreads();
    SF::template evaluate<true, false>(Sm, Dm, ul, voo, nullptr, nullptr, nullptr, tmp);
    double nu_cell = 0.;
    if (a.art_visc)
      {
        // :344-369: residual of the concentration equation with the two old states; interpolation
        // is linear, so u_old + u_old_old and grad(phi_old + phi_old_old) come from nodal sums
        gather_cell<KU, 3, NT, false, ZF>(a.vel_sum, vl, cx, cy, cz, vx, vy, vz, 0u);
        gather_cell<S, 1, NT, false, ZF>(a.ls_sum, ul, cx, cy, cz, nx, ny, nz, 0u);
        __syncthreads();
        for (int d = 0; d < 3; ++d)
          SFV::template evaluate<true, false>(Sv, Sv, vl + d * NDV3, vq + d * NQ3, nullptr, nullptr, nullptr, tmp);
        SF::template evaluate<false, true>(Sm, Dm, ul, nullptr, grad, grad + NQ3, grad + 2 * NQ3, tmp);
        double max_res = 0., max_vel = 0.;
        for (int q = tid; q < NQ3; q += NT)
          {
            double ugr = 0., u2 = 0.;
            for (int e = 0; e < 3; ++e)
              {
                const double u = vq[e * NQ3 + q];
                ugr += u * grad[e * NQ3 + q] * ih[e];
                u2 += u * u;
              }
            max_res = fmax(max_res, fabs((vo[q] - voo[q]) * a.old_step_inv + 0.25 * ugr));
            max_vel = fmax(max_vel, sqrt(u2));
          }
        __syncthreads();
        red[tid]      = max_res;
        red[NT + tid] = max_vel;
        __syncthreads();
        for (int o = 0; o < NT; ++o) // (every thread reduces: the result is needed by all of them)
          {
            max_res = fmax(max_res, red[o]);
            max_vel = fmax(max_vel, red[NT + o]);
          }
        const double hcell = (ZF ? fmax(a.brick.h[0], a.brick.h[1]) : fmax(a.brick.h[0], fmax(a.brick.h[1], a.brick.h[2])));
        nu_cell            = 0.03 * max_vel * hcell * fmin(1., max_res / a.global_scaling);
        if (tid == 0)
          a.art_visc[c] = nu_cell;
        __syncthreads();
      }
    // velocity values at the LS quadrature points
    gather_cell<KU, 3, NT, false, ZF>(a.vel, vl, cx, cy, cz, vx, vy, vz, 0u);
    __syncthreads();
    for (int d = 0; d < 3; ++d)
      SFV::template evaluate<true, false>(Sv, Sv, vl + d * NDV3, vq + d * NQ3, nullptr, nullptr, nullptr, tmp);
    // current level set
    gather_cell<S, 1, NT, false, ZF>(a.src, ul, cx, cy, cz, nx, ny, nz, 0u);
    __syncthreads();
    SF::template evaluate<true, true>(Sm, Dm, ul, val, grad, grad + NQ3, grad + 2 * NQ3, tmp);

    double *uq = a.qstate + (size_t)c * 3 * NQ3;
    for (int q = tid; q < NQ3; q += NT)
      {
        const int    qx = q % NQ, qy = (q / NQ) % NQ, qz = q / (NQ * NQ);
        const double jxw = det * wq[qx] * wq[qy] * (ZF ? 1. : wq[qz]);
        double old_value = P.weight_old * vo[q];
        if (a.flag) // bdf_2 && step_no > 1  :375-378
          old_value += P.weight_old_old * voo[q];
        double ug = 0.;
        for (int e = 0; e < 3; ++e)
          {
            const double u = vq[e * NQ3 + q];
            ug += u * grad[e * NQ3 + q] * ih[e];
            uq[e * NQ3 + q] = u; // :389 evaluated_convection
          }
        val[q] = -(val[q] * P.weight + ug + old_value) * jxw;
        if (a.art_visc) // :387-388
          for (int e = 0; e < 3; ++e)
            grad[e * NQ3 + q] *= -nu_cell * ih[e] * ih[e] * jxw;
      }
    __syncthreads();
    if (a.art_visc)
      SF::template integrate<true, true>(Sm, Dm, val, grad, grad + NQ3, grad + 2 * NQ3, ul, tmp);
    else
      SF::template integrate<true, false>(Sm, Dm, val, nullptr, nullptr, nullptr, ul, tmp);
    scatter_cell<S, 1, NT, ZF>(a.dst, ul, cx, cy, cz, nx, ny, nz, a.brick.con_ls, a.brick.colour);
  }

  // boundary part of the stabilisation (:419-472 operator, bsign = -1; :569-617 rhs, bsign = +1):
  // dst_i += bsign (phi_i, n . nu_cell grad vec) over the boundary faces that are not symmetry
  // faces; face quadrature QIterated(QGauss<1>(2), s)^2, vec read plainly, constrained rows skipped.
  // One workgroup per cell (cells away from the boundary return at once).
  template <int S, int NT>
  __global__ __launch_bounds__(NT) void ls_advect_boundary_kernel(const LSArgs a)
  {
    using C = LSCell<S>;
    constexpr int ND = C::ND, NQ = C::NQ, ND3 = C::ND3;
    const long c   = brick_cell(a.brick, blockIdx.x, a.n_cells);
    const int  ncx = a.brick.ncell[0], ncy = a.brick.ncell[1], ncz = a.brick.ncell[2];
    const int  cc[3] = {(int)(c % ncx), (int)((c / ncx) % ncy), (int)(c / ((long)ncx * ncy))};
    const int  nc[3] = {ncx, ncy, ncz};
    unsigned   faces = 0;
    for (int d = 0; d < 3; ++d)
      {
        if (cc[d] == 0 && !(a.symmetry >> (2 * d) & 1u))
          faces |= 1u << (2 * d);
        if (cc[d] == nc[d] - 1 && !(a.symmetry >> (2 * d + 1) & 1u))
          faces |= 1u << (2 * d + 1);
      }
    if (!faces)
      return;
    extern __shared__ double lds[];
    double *Sm = lds, *wq = Sm + 2 * NQ * ND, *ul = lds + C::TABP, *flux = ul + ND3; // flux [NQ][NQ]
    const int tid = threadIdx.x;
    for (int o = tid; o < C::TAB; o += NT)
      lds[o] = a.tab[o];
    const int nx = S * ncx + 1, ny = S * ncy + 1, nz = S * ncz + 1;
    gather_cell<S, 1, NT, false>(a.src, ul, cc[0], cc[1], cc[2], nx, ny, nz, 0u);
    __syncthreads();
    const double nu = a.art_visc[c];
    for (int f = 0; f < 6; ++f)
      {
        if (!(faces >> f & 1u))
          continue;
        const int d = f / 2, side = f % 2, t1 = (d + 1) % 3, t2 = (d + 2) % 3;
        const int e1 = t1 < t2 ? t1 : t2, e2 = t1 < t2 ? t2 : t1; // tangential directions, ascending
        const double area = a.brick.h[e1] * a.brick.h[e2];
        const int    in0 = side ? S - 1 : 0; // the two nodes whose hats have a normal derivative at the face
        const int    stride[3] = {1, ND, ND * ND};
        // normal derivative of vec at the face points (hats are piecewise linear: -+ s / h between
        // the two nodes next to the face), times n, nu, JxW and the sign
        for (int q = tid; q < NQ * NQ; q += NT)
          {
            const int q1 = q % NQ, q2 = q / NQ;
            double    dn = 0.;
            for (int j2 = 0; j2 < ND; ++j2)
              for (int j1 = 0; j1 < ND; ++j1)
                {
                  const double w = Sm[q1 * ND + j1] * Sm[q2 * ND + j2];
                  if (w != 0.)
                    dn += w * (ul[j1 * stride[e1] + j2 * stride[e2] + (in0 + 1) * stride[d]] -
                               ul[j1 * stride[e1] + j2 * stride[e2] + in0 * stride[d]]);
                }
            flux[q] = a.bsign * (side ? 1. : -1.) * nu * dn * ((double)S / a.brick.h[d]) * area * wq[q1] * wq[q2];
          }
        __syncthreads();
        for (int i = tid; i < ND * ND; i += NT)
          {
            const int i1 = i % ND, i2 = i / ND;
            double    r  = 0.;
            for (int q2 = 0; q2 < NQ; ++q2)
              for (int q1 = 0; q1 < NQ; ++q1)
                r += Sm[q1 * ND + i1] * Sm[q2 * ND + i2] * flux[q2 * NQ + q1];
            int ii[3];
            ii[e1] = i1;
            ii[e2] = i2;
            ii[d]  = side ? S : 0;
            const int I = cc[0] * S + ii[0], J = cc[1] * S + ii[1], K = cc[2] * S + ii[2];
            if (!on_constrained_face(I, J, K, nx, ny, nz, a.brick.con_ls, 1, 0))
              a.dst[I + (long)nx * (J + (long)ny * K)] += r; // (cells of one colour share no node)
          }
        __syncthreads();
      }
  }

  // get_maximal_velocity (:39-68): largest |u| on the points of QIterated(QTrapezoid<1>(), k + 1)
  template <int KU, bool ZF = false>
  __global__ __launch_bounds__(64) void ls_max_velocity_kernel(const double *__restrict__ vel, const double *__restrict__ tab,
                                                               const int ncx, const int ncy, const int ncz,
                                                               unsigned long long *result)
  {
    constexpr int ND = KU + 1, NP = KU + 2, NDZ = ZF ? 1 : ND, NPZ = ZF ? 1 : NP, ND3 = ND * ND * NDZ, NP3 = NP * NP * NPZ;
    __shared__ double St[NP * ND], vl[3 * ND3], red[64];
    const int  tid = threadIdx.x;
    const long c   = blockIdx.x;
    const int  cx = c % ncx, cy = (c / ncx) % ncy, cz = c / ((long)ncx * ncy);
    for (int o = tid; o < NP * ND; o += 64)
      St[o] = tab[o];
    gather_cell<KU, 3, 64, false, ZF>(vel, vl, cx, cy, cz, KU * ncx + 1, KU * ncy + 1, ZF ? 1 : KU * ncz + 1, 0u);
    __syncthreads();
    double best = 0.;
    for (int q = tid; q < NP3; q += 64)
      {
        const int qx = q % NP, qy = (q / NP) % NP, qz = q / (NP * NP);
        double    u[3] = {0., 0., 0.};
        for (int k = 0; k < NDZ; ++k)
          for (int j = 0; j < ND; ++j)
            for (int i = 0; i < ND; ++i)
              {
                const double w = St[qx * ND + i] * St[qy * ND + j] * (ZF ? 1. : St[qz * ND + k]);
                for (int e = 0; e < 3; ++e)
                  u[e] += w * vl[e * ND3 + (k * ND + j) * ND + i];
              }
        best = fmax(best, sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
      }
    red[tid] = best;
    __syncthreads();
    if (tid == 0)
      {
        for (int o = 1; o < 64; ++o)
          best = fmax(best, red[o]);
        atomicMax(result, (unsigned long long)__double_as_longlong(best)); // (non-negative doubles order as integers)
      }
  }

  // dst[c] = diag[c] * src[c] on constrained rows (e.g. reinitialization.cc:227-230)
  __global__ __launch_bounds__(256) void ls_constrained_rows_kernel(double *dst, const double *src,
                                                                    const double *diag, const long n,
                                                                    const int nx, const int ny,
                                                                    const int nz, const uint32_t mask)
  {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
      {
        const int I = i % nx, J = (i / nx) % ny, K = i / ((long)nx * ny);
        if (on_constrained_face(I, J, K, nx, ny, nz, mask, 1, 0))
          dst[i] = diag[i] * src[i];
      }
  }

  // ------------------------------------------------------------------ launchers
  template <int S, bool ZF = false>
  static int launch_ls_s(adaflo_ctx *ctx, const int kind, LSArgs &a, const int ncomp_blocks)
  {
    constexpr int NT = S >= 3 ? 256 : 64;
    constexpr int NQ3 = 4 * S * S * (ZF ? 1 : 2 * S), ND3 = (S + 1) * (S + 1) * (ZF ? 1 : S + 1);
    if (ZF && kind == 3)
      return ADAFLO_EUNSUPPORTED; // (the boundary term of the convection stabilisation: dim = 3 only)
    const dim3    block(NT);
    hipError_t    err = hipSuccess;
    // one launch per colour of the cells (no atomics in the scatter, fe_kernels.hpp)
    for (int colour = 0; colour < 8 && err == hipSuccess; ++colour)
    {
    const long n_colour = n_cells_of_colour(a.brick.ncell, colour);
    if (n_colour == 0)
      continue;
    a.brick.colour = colour;
    const dim3 grid((unsigned)n_colour, (unsigned)ncomp_blocks);
    if (kind == 0)
      {
        const size_t lds = sizeof(double) * ls_lds_doubles<S, NT, ZF>(0);
        if (lds > 64 * 1024)
          err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ls_vmult_kernel<S, NT, ZF>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ls_vmult_kernel<S, NT, ZF>), grid, block, lds, ctx->stream, a);
      }
    else if (kind == 1)
      {
        const size_t lds = sizeof(double) * ls_lds_doubles<S, NT, ZF>(3 * (NQ3 > ND3 ? NQ3 : ND3));
        if (lds > 64 * 1024)
          err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ls_rhs_kernel<S, NT, ZF>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ls_rhs_kernel<S, NT, ZF>), grid, block, lds, ctx->stream, a);
      }
    else if (kind == 3)
      {
        const size_t lds = sizeof(double) * (LSCell<S>::TABP + ND3 + 4 * S * S);
        hipLaunchKernelGGL((ls_advect_boundary_kernel<S, 64>), dim3((unsigned)n_colour), dim3(64), lds, ctx->stream, a);
      }
    else
      {
#define ADV(KU)                                                                                      \
  {                                                                                                  \
    constexpr int    NDV3 = (KU + 1) * (KU + 1) * (ZF ? 1 : KU + 1);                                          \
    constexpr size_t tmpv = SumFac<KU + 1, 2 * S, NT, ZF>::TMP, tmps = SumFac<S + 1, 2 * S, NT, ZF>::TMP;    \
    const size_t     lds  = sizeof(double) * (ls_lds_doubles<S, NT, ZF>(0) + (tmpv > tmps ? tmpv - tmps : 0) + \
                                         2 * S * (KU + 1) + 3 * NDV3 + 5 * NQ3 + 2 * NT);             \
    if (lds > 64 * 1024)                                                                             \
      err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ls_advect_rhs_kernel<S, KU, NT, ZF>),    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
    hipLaunchKernelGGL((ls_advect_rhs_kernel<S, KU, NT, ZF>), grid, block, lds, ctx->stream, a);         \
  }
        switch (ctx->k)
          {
            case 2:
              ADV(2);
              break;
            case 3:
              ADV(3);
              break;
            case 4:
              ADV(4);
              break;
            case 5:
              ADV(5);
              break;
            default:
              return ADAFLO_EUNSUPPORTED;
          }
#undef ADV
      }
    }
    a.brick.colour = -1;
    if (err == hipSuccess)
      err = hipGetLastError();
    return err == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  int launch_ls_max_velocity(adaflo_ctx *ctx, const double *vel, const double *tab, unsigned long long *result)
  {
    const dim3 grid((unsigned)ctx->n_cells), block(64);
    switch (ctx->k)
      {
        case 2:
          if (ctx->flat)
            hipLaunchKernelGGL((ls_max_velocity_kernel<2, true>), grid, block, 0, ctx->stream, vel, tab, ctx->desc.ncell[0],
                               ctx->desc.ncell[1], 1, result);
          else
            hipLaunchKernelGGL((ls_max_velocity_kernel<2>), grid, block, 0, ctx->stream, vel, tab, ctx->desc.ncell[0],
                               ctx->desc.ncell[1], ctx->desc.ncell[2], result);
          break;
        case 3:
          if (ctx->flat)
            hipLaunchKernelGGL((ls_max_velocity_kernel<3, true>), grid, block, 0, ctx->stream, vel, tab, ctx->desc.ncell[0],
                               ctx->desc.ncell[1], 1, result);
          else
            hipLaunchKernelGGL((ls_max_velocity_kernel<3>), grid, block, 0, ctx->stream, vel, tab, ctx->desc.ncell[0],
                               ctx->desc.ncell[1], ctx->desc.ncell[2], result);
          break;
        case 4:
          hipLaunchKernelGGL((ls_max_velocity_kernel<4>), grid, block, 0, ctx->stream, vel, tab, ctx->desc.ncell[0],
                             ctx->desc.ncell[1], ctx->desc.ncell[2], result);
          break;
        case 5:
          hipLaunchKernelGGL((ls_max_velocity_kernel<5>), grid, block, 0, ctx->stream, vel, tab, ctx->desc.ncell[0],
                             ctx->desc.ncell[1], ctx->desc.ncell[2], result);
          break;
        default:
          return ADAFLO_EUNSUPPORTED;
      }
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }

  // kind: 0 operator application, 1 rhs (reinit / normal / curvature), 2 advection rhs,
  // 3 boundary part of the convection stabilisation (src = the vector, stab->bsign)
  int launch_ls(adaflo_ctx *ctx, const int kind, const int mode, const int flag, double *dst,
                const double *src, const double *src2, const double *src3, const double *vel,
                double *qstate, const int ncomp_blocks, const LSStab *stab)
  {
    LSArgs a{};
    if (stab)
      {
        a.art_visc       = stab->art_visc;
        a.vel_sum        = stab->vel_sum;
        a.ls_sum         = stab->ls_sum;
        a.old_step_inv   = stab->old_step_inv;
        a.global_scaling = stab->global_scaling;
        a.bsign          = stab->bsign;
        a.symmetry       = stab->symmetry;
      }
    a.brick   = ctx->brick;
    a.ls      = ctx->ls;
    a.src     = src;
    a.src2    = src2;
    a.src3    = src3;
    a.vel     = vel;
    a.dst     = dst;
    a.qstate  = qstate;
    a.diag    = ctx->d_ls_diag;
    a.tab     = ctx->d_tab_ls;
    a.n_cells = ctx->n_cells;
    a.n_nodes = ctx->n_nodes_ls;
    a.mode    = mode;
    a.flag    = flag;
    if (ctx->flat)
      switch (ctx->s)
        {
          case 1:
            return launch_ls_s<1, true>(ctx, kind, a, ncomp_blocks);
          case 2:
            return launch_ls_s<2, true>(ctx, kind, a, ncomp_blocks);
          case 3:
            return launch_ls_s<3, true>(ctx, kind, a, ncomp_blocks);
          case 4:
            return launch_ls_s<4, true>(ctx, kind, a, ncomp_blocks);
          default:
            return ADAFLO_EUNSUPPORTED;
        }
    switch (ctx->s)
      {
        case 1:
          return launch_ls_s<1>(ctx, kind, a, ncomp_blocks);
        case 2:
          return launch_ls_s<2>(ctx, kind, a, ncomp_blocks);
        case 3:
          return launch_ls_s<3>(ctx, kind, a, ncomp_blocks);
        case 4:
          return launch_ls_s<4>(ctx, kind, a, ncomp_blocks);
        default:
          return ADAFLO_EUNSUPPORTED;
      }
  }

  int launch_ls_constrained_rows(adaflo_ctx *ctx, double *dst, const double *src, const int nblocks)
  {
    if (!ctx->brick.con_ls)
      return 0;
    if (!ctx->d_ls_diag)
      return ADAFLO_ENOTINIT;
    const int nx = ctx->s * ctx->desc.ncell[0] + 1, ny = ctx->s * ctx->desc.ncell[1] + 1,
              nz = ctx->flat ? 1 : ctx->s * ctx->desc.ncell[2] + 1;
    long nb = (ctx->n_nodes_ls + 255) / 256;
    if (nb > 4096)
      nb = 4096;
    for (int b = 0; b < nblocks; ++b)
      hipLaunchKernelGGL(ls_constrained_rows_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream,
                         dst + (size_t)b * ctx->n_nodes_ls, src + (size_t)b * ctx->n_nodes_ls,
                         ctx->d_ls_diag, (long)ctx->n_nodes_ls, nx, ny, nz, ctx->brick.con_ls);
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }
} // namespace adaflo_hip
