/* adaflo_oracle_fast.c -- sum-factorised, OpenMP-parallel CPU restatement of
 * NavierStokesMatrix::vmult (source/navier_stokes_matrix.cc:221-262, :601-916).
 *
 * TEST INFRASTRUCTURE ONLY (see adaflo_oracle.c).  This variant exists to be
 * the TIMED CPU baseline of bench.py ("cpu_baseline.kind = port"): same data
 * flow as the reference's CPU path -- per-cell gather, sum-factorised
 * evaluate, quadrature-point loop with STORED linearisation state, integrate,
 * scatter-add -- with conflict-free 8-colouring of the cells instead of
 * deal.II's TBB partitioning, all host cores via OpenMP.  It is validated
 * against the naive oracle in tests/test_oracle_kats.py and
 * tests/test_golden_fixtures.py.  It is NOT deal.II's
 * tuned AVX-512 kernel (no cross-cell SIMD batching, no even-odd trick);
 * bench.py labels it accordingly.
 *
 * 3D only; vmult (Newton / Picard-type / Stokes branches), constant or
 * variable coefficients, canonical state layout [cell][q][12]; and (round 6)
 * NavierStokesMatrix::residual of the two fully implicit schemes
 * (source/navier_stokes_matrix.cc:266-293, :663-686, :725-732, :783-799), which WRITES that
 * state -- so that the full-size tests can hold the path bench.py times
 * (residual -> vmult on the state the residual left) against the oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct
{
  int    dim;
  int    ncell[3];
  double h[3];
  double origin[3];
} orc_mesh;

typedef struct
{
  int    physical_type, linearization;
  double beta, tau_grad_div, density, viscosity, damping, density_diff;
  double weight, weight_old, weight_old_old, tau1, extrap_old, extrap_old_old;
} orc_ns_params;

void orc_gauss_legendre(int n, double *x, double *w);
void orc_shape_1d(int fe_type, int degree, int nq, const double *xq, double *S, double *D);
void orc_ns_pressure_projection(long n_p, double *vec, const double *weights, const double *modes);

#define MAXN 7

/* out[c][b][q] = sum_i M[q][i] in[c][b][i] along direction dir of an n0 x n1 x n2 tensor */
static void apply_dir(const double *M, int nq, int nd, int transpose, int dir, const int nin[3],
                      const double *in, double *out, int add)
{
  int nout[3] = {nin[0], nin[1], nin[2]};
  const int no = transpose ? nd : nq, ni = transpose ? nq : nd;
  nout[dir] = no;
  const int sin[3] = {1, nin[0], nin[0] * nin[1]}, sout[3] = {1, nout[0], nout[0] * nout[1]};
  const int a = (dir + 1) % 3, b = (dir + 2) % 3;
  for (int ib = 0; ib < nin[b]; ++ib)
    for (int ia = 0; ia < nin[a]; ++ia)
      {
        const double *pin = in + ia * sin[a] + ib * sin[b];
        double *pout = out + ia * sout[a] + ib * sout[b];
        for (int o = 0; o < no; ++o)
          {
            double s = 0.;
            for (int i = 0; i < ni; ++i)
              s += (transpose ? M[i * nd + o] : M[o * nd + i]) * pin[i * sin[dir]];
            if (add) pout[o * sout[dir]] += s; else pout[o * sout[dir]] = s;
          }
      }
}

/* number of OpenMP threads of the following calls (the caller knows the CPU quota of its container) */
void orc_fast_set_threads(int n)
{
#ifdef _OPENMP
  if (n > 0)
    omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_fast_n_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* residual == 0: y = J x, full vmult semantics incl. zeroing, constrained rows, mean projection (lin is read);
 * residual == 1: the cell loop of NavierStokesMatrix::residual (:266-293): plain reads of the solution (:663-671),
 * values of the old solutions (:673-686, :727-732), the nonlinear term of the solution itself and the state
 * (u, grad u) or (u, div u) WRITTEN to lin (:783-799), no contribution to constrained rows; the caller negates and
 * adds the user vector */
static int fast_ns_apply(const orc_mesh *m, int k, const orc_ns_params *P, const double *src_u,
                         const double *src_p, double *dst_u, double *dst_p, const uint8_t *con_u,
                         const uint8_t *con_p, double *lin, const double *rho_q,
                         const double *mu_q, const double *damp_q, const double *weights,
                         const double *modes, int residual, const double *old_u, const double *oldold_u)
{
  if (m->dim != 3 || k + 1 > MAXN) return -1;
  if (residual && (P->linearization > 1 || !lin)) return -2; /* (the extrapolating schemes: naive oracle only) */
  const int with_old = residual && P->physical_type == 0;
  const double w1 = P->weight_old, w2 = P->weight_old_old;
  const int p = k - 1, n = k + 1, ndu = k + 1, ndp = k;
  const int nq3 = n * n * n, ndu3 = ndu * ndu * ndu;
  double xq[MAXN], wq[MAXN], Su[MAXN * MAXN], Du[MAXN * MAXN], Sp[MAXN * MAXN], Dp[MAXN * MAXN];
  double Dc[MAXN * MAXN]; /* collocation derivative at the q-points */
  orc_gauss_legendre(n, xq, wq);
  orc_shape_1d(0, k, n, xq, Su, Du);
  orc_shape_1d(0, p, n, xq, Sp, Dp);
  { /* Lagrange basis through the q-points, derivative at the q-points */
    for (int q = 0; q < n; ++q)
      for (int r = 0; r < n; ++r)
        {
          double d = 0.;
          for (int mm = 0; mm < n; ++mm)
            if (mm != r)
              {
                double t = 1. / (xq[r] - xq[mm]);
                for (int j = 0; j < n; ++j)
                  if (j != r && j != mm) t *= (xq[q] - xq[j]) / (xq[r] - xq[j]);
                d += t;
              }
          Dc[q * n + r] = d;
        }
  }
  const long nnu[3] = {(long)k * m->ncell[0] + 1, (long)k * m->ncell[1] + 1, (long)k * m->ncell[2] + 1};
  const long nnp[3] = {(long)p * m->ncell[0] + 1, (long)p * m->ncell[1] + 1, (long)p * m->ncell[2] + 1};
  const long nu = nnu[0] * nnu[1] * nnu[2] * 3, np = nnp[0] * nnp[1] * nnp[2];
  memset(dst_u, 0, sizeof(double) * nu);
  memset(dst_p, 0, sizeof(double) * np);

  const double w0 = P->physical_type == 0 ? P->weight : 0., tau1 = P->tau1, beta = P->beta;
  const int stokes = P->physical_type == 2;
  const double ih[3] = {1. / m->h[0], 1. / m->h[1], 1. / m->h[2]};
  const double det = m->h[0] * m->h[1] * m->h[2];
  const int ncx = m->ncell[0], ncy = m->ncell[1], ncz = m->ncell[2];

  for (int colour = 0; colour < 8; ++colour)
    {
      const int ox = colour & 1, oy = (colour >> 1) & 1, oz = colour >> 2;
      const int mx = (ncx - ox + 1) / 2, my = (ncy - oy + 1) / 2, mz = (ncz - oz + 1) / 2;
      const long nblk = (long)mx * my * mz;
#pragma omp parallel
      {
        double ul[3 * MAXN * MAXN * MAXN], pl[MAXN * MAXN * MAXN], t1[MAXN * MAXN * MAXN],
          t2[MAXN * MAXN * MAXN];
        double vu[3][MAXN * MAXN * MAXN], gu[3][3][MAXN * MAXN * MAXN], vp[MAXN * MAXN * MAXN];
        double vo[3][MAXN * MAXN * MAXN]; /* residual: w1 u_old + w2 u_old_old at the q-points (interpolation is linear) */
#pragma omp for schedule(static)
        for (long blk = 0; blk < nblk; ++blk)
          {
            const int cx = 2 * (int)(blk % mx) + ox, cy = 2 * (int)((blk / mx) % my) + oy,
                      cz = 2 * (int)(blk / ((long)mx * my)) + oz;
            const long c = cx + (long)ncx * (cy + (long)ncy * cz);
            /* gather (constraints resolved) */
            for (int kk = 0; kk < ndu; ++kk)
              for (int j = 0; j < ndu; ++j)
                for (int i = 0; i < ndu; ++i)
                  {
                    const long node = (cx * k + i) + nnu[0] * ((cy * k + j) + nnu[1] * (long)(cz * k + kk));
                    const int l = i + ndu * (j + ndu * kk);
                    for (int d = 0; d < 3; ++d)
                      ul[d * ndu3 + l] = (!residual && con_u && con_u[node * 3 + d]) ? 0. : src_u[node * 3 + d];
                  }
            for (int kk = 0; kk < ndp; ++kk)
              for (int j = 0; j < ndp; ++j)
                for (int i = 0; i < ndp; ++i)
                  {
                    const long node = (cx * p + i) + nnp[0] * ((cy * p + j) + nnp[1] * (long)(cz * p + kk));
                    pl[i + ndp * (j + ndp * kk)] = (!residual && con_p && con_p[node]) ? 0. : src_p[node];
                  }
            if (with_old) /* :673-686, :727-732 */
              {
                double ol[3 * MAXN * MAXN * MAXN];
                for (int kk = 0; kk < ndu; ++kk)
                  for (int j = 0; j < ndu; ++j)
                    for (int i = 0; i < ndu; ++i)
                      {
                        const long node = (cx * k + i) + nnu[0] * ((cy * k + j) + nnu[1] * (long)(cz * k + kk));
                        const int l = i + ndu * (j + ndu * kk);
                        for (int d = 0; d < 3; ++d)
                          ol[d * ndu3 + l] = w1 * old_u[node * 3 + d] + w2 * oldold_u[node * 3 + d];
                      }
                for (int d = 0; d < 3; ++d)
                  {
                    int s0[3] = {ndu, ndu, ndu}, s1[3] = {n, ndu, ndu}, s2[3] = {n, n, ndu};
                    apply_dir(Su, n, ndu, 0, 0, s0, ol + d * ndu3, t1, 0);
                    apply_dir(Su, n, ndu, 0, 1, s1, t1, t2, 0);
                    apply_dir(Su, n, ndu, 0, 2, s2, t2, vo[d], 0);
                  }
              }
            /* evaluate: interpolate to q-points, then collocation derivatives */
            for (int d = 0; d < 3; ++d)
              {
                int s0[3] = {ndu, ndu, ndu}, s1[3] = {n, ndu, ndu}, s2[3] = {n, n, ndu}, s3[3] = {n, n, n};
                apply_dir(Su, n, ndu, 0, 0, s0, ul + d * ndu3, t1, 0);
                apply_dir(Su, n, ndu, 0, 1, s1, t1, t2, 0);
                apply_dir(Su, n, ndu, 0, 2, s2, t2, vu[d], 0);
                for (int e = 0; e < 3; ++e) apply_dir(Dc, n, n, 0, e, s3, vu[d], gu[d][e], 0);
              }
            {
              int s0[3] = {ndp, ndp, ndp}, s1[3] = {n, ndp, ndp}, s2[3] = {n, n, ndp};
              apply_dir(Sp, n, ndp, 0, 0, s0, pl, t1, 0);
              apply_dir(Sp, n, ndp, 0, 1, s1, t1, t2, 0);
              apply_dir(Sp, n, ndp, 0, 2, s2, t2, vp, 0);
            }
            /* quadrature-point loop, source/navier_stokes_matrix.cc:702-893 */
            for (int q = 0; q < nq3; ++q)
              {
                const int qx = q % n, qy = (q / n) % n, qz = q / (n * n);
                const double jxw = det * wq[qx] * wq[qy] * wq[qz];
                double g[3][3], val[3], conv[3] = {0., 0., 0.};
                for (int d = 0; d < 3; ++d)
                  {
                    val[d] = vu[d][q];
                    for (int e = 0; e < 3; ++e) g[d][e] = gu[d][e][q] * ih[e];
                  }
                const double div = g[0][0] + g[1][1] + g[2][2];
                double *L = lin ? lin + ((size_t)c * nq3 + q) * 12 : NULL;
                if (!stokes)
                  {
                    const double rho = rho_q ? rho_q[c * nq3 + q] : P->density;
                    for (int d = 0; d < 3; ++d) conv[d] = val[d] * w0;
                    if (residual) /* :727-732, :783-799 */
                      {
                        for (int d = 0; d < 3; ++d)
                          {
                            if (with_old) conv[d] += vo[d][q];
                            double res = beta * div * val[d];
                            for (int e = 0; e < 3; ++e) res += val[e] * g[d][e];
                            conv[d] += tau1 * res;
                            L[d] = val[d];
                          }
                        if (P->linearization == 0)
                          for (int d = 0; d < 3; ++d)
                            for (int e = 0; e < 3; ++e) L[3 + 3 * d + e] = g[d][e];
                        else
                          L[3] = div;
                      }
                    else if (P->linearization == 0)
                      {
                        const double f1 = beta * div, f2 = beta * (L[3] + L[7] + L[11]);
                        for (int d = 0; d < 3; ++d)
                          {
                            double res = f1 * L[d] + f2 * val[d];
                            for (int e = 0; e < 3; ++e)
                              res += L[e] * g[d][e] + val[e] * L[3 + 3 * d + e];
                            conv[d] += tau1 * res;
                          }
                      }
                    else if (P->linearization != 3)
                      for (int d = 0; d < 3; ++d)
                        {
                          double res = beta * L[3] * val[d];
                          for (int e = 0; e < 3; ++e) res += L[e] * g[d][e];
                          conv[d] += tau1 * res;
                        }
                    const double damping = damp_q ? damp_q[c * nq3 + q] : P->damping;
                    for (int d = 0; d < 3; ++d) conv[d] = conv[d] * rho - damping * val[d];
                  }
                const double tmu = (mu_q ? mu_q[c * nq3 + q] : P->viscosity) * tau1;
                const double pres = vp[q];
                vp[q] = -div * jxw;
                for (int d = 0; d < 3; ++d)
                  for (int e = d + 1; e < 3; ++e)
                    {
                      const double sym = tmu * (g[d][e] + g[e][d]);
                      g[d][e] = g[e][d] = sym;
                    }
                for (int d = 0; d < 3; ++d)
                  g[d][d] = 2. * tmu * g[d][d] + P->tau_grad_div * div - pres;
                for (int d = 0; d < 3; ++d)
                  {
                    vu[d][q] = conv[d] * jxw;
                    for (int e = 0; e < 3; ++e) gu[d][e][q] = g[d][e] * (jxw * ih[e]);
                  }
              }
            /* integrate + scatter */
            for (int d = 0; d < 3; ++d)
              {
                int s3[3] = {n, n, n}, s2[3] = {n, n, n}, s1[3] = {n, n, ndu}, s0[3] = {n, ndu, ndu};
                for (int e = 0; e < 3; ++e) apply_dir(Dc, n, n, 1, e, s3, gu[d][e], vu[d], 1);
                apply_dir(Su, n, ndu, 1, 2, s2, vu[d], t1, 0);
                apply_dir(Su, n, ndu, 1, 1, s1, t1, t2, 0);
                apply_dir(Su, n, ndu, 1, 0, s0, t2, ul + d * ndu3, 0);
              }
            for (int kk = 0; kk < ndu; ++kk)
              for (int j = 0; j < ndu; ++j)
                for (int i = 0; i < ndu; ++i)
                  {
                    const long node = (cx * k + i) + nnu[0] * ((cy * k + j) + nnu[1] * (long)(cz * k + kk));
                    const int l = i + ndu * (j + ndu * kk);
                    for (int d = 0; d < 3; ++d)
                      if (!(con_u && con_u[node * 3 + d])) dst_u[node * 3 + d] += ul[d * ndu3 + l];
                  }
            if (P->linearization != 4)
              {
                int s2[3] = {n, n, n}, s1[3] = {n, n, ndp}, s0[3] = {n, ndp, ndp};
                apply_dir(Sp, n, ndp, 1, 2, s2, vp, t1, 0);
                apply_dir(Sp, n, ndp, 1, 1, s1, t1, t2, 0);
                apply_dir(Sp, n, ndp, 1, 0, s0, t2, pl, 0);
                for (int kk = 0; kk < ndp; ++kk)
                  for (int j = 0; j < ndp; ++j)
                    for (int i = 0; i < ndp; ++i)
                      {
                        const long node = (cx * p + i) + nnp[0] * ((cy * p + j) + nnp[1] * (long)(cz * p + kk));
                        if (!(con_p && con_p[node])) dst_p[node] += pl[i + ndp * (j + ndp * kk)];
                      }
              }
          }
      }
    }
  if (residual)
    return 0;
  if (con_u)
    {
#pragma omp parallel for
      for (long i = 0; i < nu; ++i) if (con_u[i]) dst_u[i] = src_u[i];
    }
  if (con_p)
    for (long i = 0; i < np; ++i) if (con_p[i]) dst_p[i] = -src_p[i];
  if (weights && P->linearization != 4 && P->physical_type != 1)
    orc_ns_pressure_projection(np, dst_p, weights, modes);
  return 0;
}

int orc_fast_ns_vmult(const orc_mesh *m, int k, const orc_ns_params *P, const double *src_u,
                      const double *src_p, double *dst_u, double *dst_p, const uint8_t *con_u,
                      const uint8_t *con_p, const double *lin, const double *rho_q,
                      const double *mu_q, const double *damp_q, const double *weights,
                      const double *modes)
{
  return fast_ns_apply(m, k, P, src_u, src_p, dst_u, dst_p, con_u, con_p, (double *)lin, rho_q, mu_q, damp_q,
                       weights, modes, 0, NULL, NULL);
}

/* NavierStokesMatrix::residual (source/navier_stokes_matrix.cc:266-293) for the fully implicit schemes
 * (Newton: lin = (u, grad u); Picard-type: lin = (u, div u)): rhs = user - (cell loop), lin overwritten.
 * Returns -2 for the schemes that extrapolate the old velocity (the naive oracle has them). */
int orc_fast_ns_residual(const orc_mesh *m, int k, const orc_ns_params *P, const double *src_u,
                         const double *src_p, double *rhs_u, double *rhs_p, const double *user_u,
                         const double *user_p, const uint8_t *con_u, const uint8_t *con_p, double *lin,
                         const double *rho_q, const double *mu_q, const double *damp_q,
                         const double *old_u, const double *oldold_u)
{
  const int rc = fast_ns_apply(m, k, P, src_u, src_p, rhs_u, rhs_p, con_u, con_p, lin, rho_q, mu_q, damp_q,
                               NULL, NULL, 1, old_u, oldold_u);
  if (rc)
    return rc;
  const long nu = ((long)k * m->ncell[0] + 1) * ((long)k * m->ncell[1] + 1) * ((long)k * m->ncell[2] + 1) * 3;
  const long np = ((long)(k - 1) * m->ncell[0] + 1) * ((long)(k - 1) * m->ncell[1] + 1) * ((long)(k - 1) * m->ncell[2] + 1);
#pragma omp parallel for
  for (long i = 0; i < nu; ++i) rhs_u[i] = -rhs_u[i] + (user_u ? user_u[i] : 0.);
#pragma omp parallel for
  for (long i = 0; i < np; ++i) rhs_p[i] = -rhs_p[i] + (user_p ? user_p[i] : 0.);
  return 0;
}
