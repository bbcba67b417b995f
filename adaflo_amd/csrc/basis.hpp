// basis.hpp -- host-side 1D finite element data for the HIP operator engine.
//
// Produces what deal.II's ShapeInfo would hand to FEEvaluation for the spaces
// adaflo uses:
//   FE_Q(QGaussLobatto<1>(k+1))   source/navier_stokes.cc:95-106
//   FE_Q_iso_Q1(s)                source/level_set_base.cc:58-59
//   QGauss<1>(n)                  source/navier_stokes.cc:447-448
//   QIterated(QGauss<1>(2), s)    source/two_phase_base.cc:267-268
// All on the reference interval [0,1]; matrices are row-major [q][i].
#pragma once
#include <cmath>
#include <vector>

namespace adaflo_hip
{
  struct Quadrature1D
  {
    std::vector<double> x, w;
  };

  // P_n(z) and P_n'(z) on [-1,1] by the three-term recurrence
  inline void legendre_pair(const int n, const double z, double &pn, double &dpn)
  {
    double pm = 1., p = z;
    if (n == 0)
      {
        pn  = 1.;
        dpn = 0.;
        return;
      }
    for (int j = 1; j < n; ++j)
      {
        const double pp = ((2 * j + 1) * z * p - j * pm) / (j + 1);
        pm              = p;
        p               = pp;
      }
    pn  = p;
    dpn = n * (z * p - pm) / (z * z - 1.);
  }

  inline Quadrature1D gauss(const int n)
  {
    Quadrature1D q;
    q.x.resize(n);
    q.w.resize(n);
    for (int i = 0; i < (n + 1) / 2; ++i)
      {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), pn, dpn;
        for (int it = 0; it < 50; ++it)
          {
            legendre_pair(n, z, pn, dpn);
            const double dz = pn / dpn;
            z -= dz;
            if (std::abs(dz) < 1e-17)
              break;
          }
        legendre_pair(n, z, pn, dpn);
        const double w = 2. / ((1. - z * z) * dpn * dpn);
        q.x[n - 1 - i] = 0.5 * (1. + z);
        q.x[i]         = 0.5 * (1. - z);
        q.w[n - 1 - i] = q.w[i] = 0.5 * w;
      }
    return q;
  }

  // QIterated(QGauss<1>(2), s): 2-point Gauss on each of s equal sub-intervals
  inline Quadrature1D gauss2_iterated(const int s)
  {
    const Quadrature1D g = gauss(2);
    Quadrature1D       q;
    for (int m = 0; m < s; ++m)
      for (int j = 0; j < 2; ++j)
        {
          q.x.push_back((m + g.x[j]) / s);
          q.w.push_back(g.w[j] / s);
        }
    return q;
  }

  // k+1 Gauss-Lobatto points on [0,1]: end points and the roots of P_k'
  inline std::vector<double> gauss_lobatto_points(const int k)
  {
    std::vector<double> x(k + 1);
    x[0] = 0.;
    x[k] = 1.;
    for (int i = 1; i <= k / 2; ++i)
      {
        double z = std::cos(M_PI * i / k), pn, dpn; // descending from +1
        for (int it = 0; it < 50; ++it)
          {
            legendre_pair(k, z, pn, dpn);
            const double d2 = (2. * z * dpn - k * (k + 1.) * pn) / (1. - z * z);
            const double dz = dpn / d2;
            z -= dz;
            if (std::abs(dz) < 1e-17)
              break;
          }
        x[k - i] = 0.5 * (1. + z);
        x[i]     = 0.5 * (1. - z);
      }
    if (k % 2 == 0)
      x[k / 2] = 0.5;
    return x;
  }

  struct Shape1D
  {
    int                 n_dofs = 0, n_q = 0;
    std::vector<double> S, D, w, xq; // S[q*n_dofs+i], D[q*n_dofs+i]
  };

  // Lagrange basis through `nodes`, evaluated at the quadrature points
  inline Shape1D shape_lagrange(const std::vector<double> &nodes, const Quadrature1D &quad)
  {
    Shape1D   sh;
    const int nd = nodes.size(), nq = quad.x.size();
    sh.n_dofs = nd;
    sh.n_q    = nq;
    sh.w      = quad.w;
    sh.xq     = quad.x;
    sh.S.assign(nq * nd, 0.);
    sh.D.assign(nq * nd, 0.);
    for (int q = 0; q < nq; ++q)
      for (int i = 0; i < nd; ++i)
        {
          const double x = quad.x[q];
          double       v = 1., d = 0.;
          for (int j = 0; j < nd; ++j)
            if (j != i)
              v *= (x - nodes[j]) / (nodes[i] - nodes[j]);
          for (int m = 0; m < nd; ++m)
            if (m != i)
              {
                double t = 1. / (nodes[i] - nodes[m]);
                for (int j = 0; j < nd; ++j)
                  if (j != i && j != m)
                    t *= (x - nodes[j]) / (nodes[i] - nodes[j]);
                d += t;
              }
          sh.S[q * nd + i] = v;
          sh.D[q * nd + i] = d;
        }
    return sh;
  }

  inline Shape1D shape_fe_q(const int degree, const Quadrature1D &quad)
  {
    return shape_lagrange(gauss_lobatto_points(degree), quad);
  }

  // continuous piecewise-linear hats on s sub-intervals
  inline Shape1D shape_fe_q_iso_q1(const int s, const Quadrature1D &quad)
  {
    Shape1D   sh;
    const int nd = s + 1, nq = quad.x.size();
    sh.n_dofs = nd;
    sh.n_q    = nq;
    sh.w      = quad.w;
    sh.xq     = quad.x;
    sh.S.assign(nq * nd, 0.);
    sh.D.assign(nq * nd, 0.);
    for (int q = 0; q < nq; ++q)
      {
        const double t = quad.x[q] * s;
        int          m = static_cast<int>(std::floor(t));
        if (m >= s)
          m = s - 1;
        const double xi      = t - m;
        sh.S[q * nd + m]     = 1. - xi;
        sh.S[q * nd + m + 1] = xi;
        sh.D[q * nd + m]     = -double(s);
        sh.D[q * nd + m + 1] = double(s);
      }
    return sh;
  }

  // collocation derivative at the quadrature points: Dc[q][r] = l_r'(x_q) with
  // l_r the Lagrange basis through the quadrature points themselves
  inline std::vector<double> collocation_derivative(const Quadrature1D &quad)
  {
    Quadrature1D q2 = quad;
    return shape_lagrange(quad.x, q2).D;
  }
} // namespace adaflo_hip
