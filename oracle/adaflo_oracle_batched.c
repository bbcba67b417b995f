/* adaflo_oracle_batched.c -- CELL-BATCHED CPU restatement of NavierStokesMatrix::vmult
 * (source/navier_stokes_matrix.cc:221-262, :601-916): the data flow of the reference's own CPU path.
 *
 * TEST INFRASTRUCTURE ONLY (see adaflo_oracle.c): this is the second TIMED CPU baseline of bench.py.
 *
 * What deal.II's FEEvaluation does on the CPU and adaflo_oracle_fast.c does not:
 *   - W cells per SIMD register (VectorizedArray<double>, navier_stokes_matrix.cc:610): every local array is an
 *     array of `double[W]`, W = 8 with AVX-512, 4 with AVX2 (compile-time, -march=native; orc_batched_isa() reports it);
 *   - the linearisation state lives in that layout already ([batch][q][12][W], the Table<2, Tensor<.., VectorizedArray>>
 *     of navier_stokes_matrix.h:54-56), written once by orc_batched_prepare as MatrixFree::reinit / the residual would;
 *   - every loop bound is a compile-time constant (one instantiation per degree, adaflo_oracle_batched_body.h);
 *   - even-odd decomposition of the symmetric 1D matrices (optional: `even_odd`), half the multiplications per line.
 * Cells are 8-coloured (no write conflicts inside a colour), batches of one colour are spread over the OpenMP threads.
 * Validated against the naive oracle in tests/test_oracle_kats.py.  3D, constant coefficients, vmult only. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(__AVX512F__)
#define W 8
#define ISA_NAME "avx512"
#elif defined(__AVX2__)
#define W 4
#define ISA_NAME "avx2"
#else
#define W 2
#define ISA_NAME "sse2"
#endif
typedef double vd __attribute__((vector_size(8 * W)));

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

#define MAXN 6

/* a 1D matrix [nq][nd] with M[q][i] = sigma M[nq-1-q][nd-1-i], forward and transposed, plain and in even-odd form */
typedef struct
{
  int    nq, nd;
  double sigma;
  double f[MAXN * MAXN], t[MAXN * MAXN];   /* [nq][nd], [nd][nq] */
  double fe[MAXN * MAXN], fo[MAXN * MAXN]; /* [nq][(nd+1)/2], [nq][nd/2] */
  double te[MAXN * MAXN], to[MAXN * MAXN]; /* [nd][(nq+1)/2], [nd][nq/2] */
} mat1d;

static void even_odd_tables(const double *A, const int no, const int ni, double *Ae, double *Ao)
{
  const int hi = ni / 2, he = (ni + 1) / 2;
  for (int q = 0; q < no; ++q)
    {
      for (int i = 0; i < hi; ++i)
        {
          Ae[q * he + i] = 0.5 * (A[q * ni + i] + A[q * ni + ni - 1 - i]);
          Ao[q * hi + i] = 0.5 * (A[q * ni + i] - A[q * ni + ni - 1 - i]);
        }
      if (ni & 1)
        Ae[q * he + hi] = A[q * ni + hi];
    }
}

static void mat1d_init(mat1d *M, const int nq, const int nd, const double sigma, const double *A)
{
  M->nq = nq, M->nd = nd, M->sigma = sigma;
  for (int q = 0; q < nq; ++q)
    for (int i = 0; i < nd; ++i)
      M->f[q * nd + i] = M->t[i * nq + q] = A[q * nd + i];
  even_odd_tables(M->f, nq, nd, M->fe, M->fo);
  even_odd_tables(M->t, nd, nq, M->te, M->to);
}

/* out = M in (or M^T in) along direction dir of an n0 x n1 x n2 tensor of vd; all sizes are compile-time constants at
 * the call sites, so the loops below unroll */
static inline __attribute__((always_inline)) void apply_line(const mat1d *M, const int transpose, const int even_odd,
                                                             const int nq, const int nd, const int dir, const int n0,
                                                             const int n1, const int n2, const vd *in, vd *out,
                                                             const int add)
{
  const int     ni = transpose ? nq : nd, no = transpose ? nd : nq;
  const int     nin[3]  = {n0, n1, n2};
  const int     o0 = dir == 0 ? no : n0, o1 = dir == 1 ? no : n1;
  const int     sin[3]  = {1, n0, n0 * n1};
  const int     sout[3] = {1, o0, o0 * o1};
  const int     a = (dir + 1) % 3, b = (dir + 2) % 3;
  const double *A = transpose ? M->t : M->f, *Ae = transpose ? M->te : M->fe, *Ao = transpose ? M->to : M->fo;
  const int     hi = ni / 2, he = (ni + 1) / 2;
  for (int ib = 0; ib < nin[b]; ++ib)
    for (int ia = 0; ia < nin[a]; ++ia)
      {
        const vd *pin  = in + ia * sin[a] + ib * sin[b];
        vd       *pout = out + ia * sout[a] + ib * sout[b];
        if (!even_odd)
          for (int o = 0; o < no; ++o)
            {
              vd s = A[o * ni] * pin[0];
              for (int i = 1; i < ni; ++i)
                s += A[o * ni + i] * pin[i * sin[dir]];
              if (add)
                pout[o * sout[dir]] += s;
              else
                pout[o * sout[dir]] = s;
            }
        else
          {
            vd xe[MAXN], xo[MAXN];
            for (int i = 0; i < hi; ++i)
              {
                xe[i] = pin[i * sin[dir]] + pin[(ni - 1 - i) * sin[dir]];
                xo[i] = pin[i * sin[dir]] - pin[(ni - 1 - i) * sin[dir]];
              }
            if (ni & 1)
              xe[hi] = pin[hi * sin[dir]];
            for (int o = 0; o < (no + 1) / 2; ++o)
              {
                vd ev = Ae[o * he] * xe[0];
                for (int i = 1; i < he; ++i)
                  ev += Ae[o * he + i] * xe[i];
                vd od = xe[0] * 0.;
                if (hi > 0)
                  {
                    od = Ao[o * hi] * xo[0];
                    for (int i = 1; i < hi; ++i)
                      od += Ao[o * hi + i] * xo[i];
                  }
                const vd lo = ev + od, up = M->sigma * (ev - od);
                if (add)
                  {
                    pout[o * sout[dir]] += lo;
                    if (o != no - 1 - o)
                      pout[(no - 1 - o) * sout[dir]] += up;
                  }
                else
                  {
                    pout[o * sout[dir]] = lo;
                    if (o != no - 1 - o)
                      pout[(no - 1 - o) * sout[dir]] = up;
                  }
              }
          }
      }
}

typedef struct
{
  int    stokes, linearization;
  double w0, tau1, beta, density, damping, tmu, tau_grad_div, ih[3];
} batched_consts;

typedef struct
{
  orc_mesh       mesh;
  int            k;
  long           nnu[3], nnp[3], nu, np;
  long           n_batches, colour_first[9];
  int           *batch_cells;       /* [batch][lane][3] */
  int           *batch_lanes;       /* valid lanes of the batch */
  uint8_t       *batch_constrained; /* some cell of the batch has a constrained velocity dof */
  const uint8_t *con_u, *con_p;     /* borrowed from the caller */
  vd            *lin;               /* [batch][q][12], lanes = cells */
  mat1d          Su, Dc, Sp;
  double         jxw[MAXN * MAXN * MAXN];
} batched_handle;

#define BK 2
#define BEO 0
#include "adaflo_oracle_batched_body.h"
#undef BK
#undef BEO
#define BK 2
#define BEO 1
#include "adaflo_oracle_batched_body.h"
#undef BK
#undef BEO
#define BK 3
#define BEO 0
#include "adaflo_oracle_batched_body.h"
#undef BK
#undef BEO
#define BK 3
#define BEO 1
#include "adaflo_oracle_batched_body.h"
#undef BK
#undef BEO
#define BK 4
#define BEO 0
#include "adaflo_oracle_batched_body.h"
#undef BK
#undef BEO
#define BK 4
#define BEO 1
#include "adaflo_oracle_batched_body.h"
#undef BK
#undef BEO
#define BK 5
#define BEO 0
#include "adaflo_oracle_batched_body.h"
#undef BK
#undef BEO
#define BK 5
#define BEO 1
#include "adaflo_oracle_batched_body.h"
#undef BK
#undef BEO

const char *orc_batched_isa(void) { return ISA_NAME; }
int         orc_batched_width(void) { return W; }

void orc_batched_free(void *handle)
{
  batched_handle *h = (batched_handle *)handle;
  if (!h)
    return;
  free(h->batch_cells);
  free(h->batch_lanes);
  free(h->batch_constrained);
  free(h->lin);
  free(h);
}

/* the set-up deal.II does in MatrixFree::reinit + the state the residual evaluation would have stored:
 * batches of W same-coloured cells, per-batch constraint flag, the canonical state [cell][q][12] re-laid per batch */
void *orc_batched_prepare(const orc_mesh *m, const int k, const uint8_t *con_u, const uint8_t *con_p, const double *lin)
{
  if (m->dim != 3 || k < 2 || k + 1 > MAXN)
    return NULL;
  batched_handle *h = (batched_handle *)calloc(1, sizeof(batched_handle));
  h->mesh = *m, h->k = k, h->con_u = con_u, h->con_p = con_p;
  const int n = k + 1, p = k - 1, n3 = n * n * n;
  for (int d = 0; d < 3; ++d)
    {
      h->nnu[d] = (long)k * m->ncell[d] + 1;
      h->nnp[d] = (long)p * m->ncell[d] + 1;
    }
  h->nu = h->nnu[0] * h->nnu[1] * h->nnu[2] * 3, h->np = h->nnp[0] * h->nnp[1] * h->nnp[2];
  double xq[MAXN], wq[MAXN], S[MAXN * MAXN], D[MAXN * MAXN], Dc[MAXN * MAXN];
  orc_gauss_legendre(n, xq, wq);
  orc_shape_1d(0, k, n, xq, S, D);
  mat1d_init(&h->Su, n, n, 1., S);
  orc_shape_1d(0, p, n, xq, S, D);
  mat1d_init(&h->Sp, n, k, 1., S);
  for (int q = 0; q < n; ++q) /* derivative of the Lagrange basis through the quadrature points, at those points */
    for (int r = 0; r < n; ++r)
      {
        double d = 0.;
        for (int mm = 0; mm < n; ++mm)
          if (mm != r)
            {
              double t = 1. / (xq[r] - xq[mm]);
              for (int j = 0; j < n; ++j)
                if (j != r && j != mm)
                  t *= (xq[q] - xq[j]) / (xq[r] - xq[j]);
              d += t;
            }
        Dc[q * n + r] = d;
      }
  mat1d_init(&h->Dc, n, n, -1., Dc);
  const double det = m->h[0] * m->h[1] * m->h[2];
  for (int q = 0; q < n3; ++q)
    h->jxw[q] = det * wq[q % n] * wq[(q / n) % n] * wq[q / (n * n)];
  /* batches */
  const int ncx = m->ncell[0], ncy = m->ncell[1], ncz = m->ncell[2];
  long      nb = 0;
  for (int colour = 0; colour < 8; ++colour)
    {
      const int  ox = colour & 1, oy = (colour >> 1) & 1, oz = colour >> 2;
      const long nblk = (long)((ncx - ox + 1) / 2) * ((ncy - oy + 1) / 2) * ((ncz - oz + 1) / 2);
      h->colour_first[colour] = nb;
      nb += (nblk + W - 1) / W;
    }
  h->colour_first[8] = h->n_batches = nb;
  h->batch_cells                    = (int *)malloc(sizeof(int) * 3 * W * (nb > 0 ? nb : 1));
  h->batch_lanes                    = (int *)malloc(sizeof(int) * (nb > 0 ? nb : 1));
  h->batch_constrained              = (uint8_t *)calloc(nb > 0 ? nb : 1, 1);
  if (lin)
    h->lin = (vd *)aligned_alloc(64, sizeof(vd) * (size_t)(nb > 0 ? nb : 1) * n3 * 12);
  for (int colour = 0; colour < 8; ++colour)
    {
      const int  ox = colour & 1, oy = (colour >> 1) & 1, oz = colour >> 2;
      const int  mx = (ncx - ox + 1) / 2, my = (ncy - oy + 1) / 2, mz = (ncz - oz + 1) / 2;
      const long nblk = (long)mx * my * mz;
#pragma omp parallel for schedule(static)
      for (long b = h->colour_first[colour]; b < h->colour_first[colour + 1]; ++b)
        {
          const long first = (b - h->colour_first[colour]) * W;
          const int  lanes = (int)((nblk - first) < W ? (nblk - first) : W);
          h->batch_lanes[b] = lanes;
          for (int l = 0; l < W; ++l)
            {
              const long blk = first + (l < lanes ? l : 0);
              const int  cx = 2 * (int)(blk % mx) + ox, cy = 2 * (int)((blk / mx) % my) + oy,
                        cz = 2 * (int)(blk / ((long)mx * my)) + oz;
              int *c = h->batch_cells + (b * W + l) * 3;
              c[0] = cx, c[1] = cy, c[2] = cz;
              const long cell = cx + (long)ncx * (cy + (long)ncy * cz);
              if (lin)
                for (int q = 0; q < n3; ++q)
                  for (int e = 0; e < 12; ++e)
                    h->lin[((size_t)b * n3 + q) * 12 + e][l] = lin[((size_t)cell * n3 + q) * 12 + e];
              if (con_u && !h->batch_constrained[b])
                for (int kk = 0; kk < n && !h->batch_constrained[b]; ++kk)
                  for (int j = 0; j < n && !h->batch_constrained[b]; ++j)
                    for (int i = 0; i < n; ++i)
                      {
                        const long node = (cx * k + i) + h->nnu[0] * ((cy * k + j) + h->nnu[1] * (long)(cz * k + kk));
                        if (con_u[node * 3] | con_u[node * 3 + 1] | con_u[node * 3 + 2])
                          {
                            h->batch_constrained[b] = 1;
                            break;
                          }
                      }
            }
        }
    }
  return h;
}

/* y = J x with the full vmult semantics (zeroing, constrained rows, mean-value projection) */
int orc_batched_ns_vmult(void *handle, const orc_ns_params *P, const double *src_u, const double *src_p, double *dst_u,
                         double *dst_p, const double *weights, const double *modes, const int even_odd)
{
  batched_handle *h = (batched_handle *)handle;
  if (!h)
    return -1;
  if (P->physical_type != 2 && P->linearization != 3 && P->linearization != 4 && !h->lin)
    return -2;
  batched_consts c;
  c.stokes = P->physical_type == 2, c.linearization = P->linearization;
  c.w0 = P->physical_type == 0 ? P->weight : 0., c.tau1 = P->tau1, c.beta = P->beta;
  c.density = P->density, c.damping = P->damping, c.tmu = P->viscosity * P->tau1, c.tau_grad_div = P->tau_grad_div;
  for (int d = 0; d < 3; ++d)
    c.ih[d] = 1. / h->mesh.h[d];
#pragma omp parallel for schedule(static)
  for (long i = 0; i < h->nu; ++i)
    dst_u[i] = 0.;
  memset(dst_p, 0, sizeof(double) * h->np);
  void (*cells)(const batched_handle *, const batched_consts *, const double *, const double *, double *, double *, int) = NULL;
  switch (h->k * 2 + (even_odd ? 1 : 0))
    {
      case 4: cells = batched_cells_k2_eo0; break;
      case 5: cells = batched_cells_k2_eo1; break;
      case 6: cells = batched_cells_k3_eo0; break;
      case 7: cells = batched_cells_k3_eo1; break;
      case 8: cells = batched_cells_k4_eo0; break;
      case 9: cells = batched_cells_k4_eo1; break;
      case 10: cells = batched_cells_k5_eo0; break;
      case 11: cells = batched_cells_k5_eo1; break;
      default: return -3;
    }
  for (int colour = 0; colour < 8; ++colour)
    cells(h, &c, src_u, src_p, dst_u, dst_p, colour);
  if (h->con_u)
    {
#pragma omp parallel for schedule(static)
      for (long i = 0; i < h->nu; ++i)
        if (h->con_u[i])
          dst_u[i] = src_u[i];
    }
  if (h->con_p)
    for (long i = 0; i < h->np; ++i)
      if (h->con_p[i])
        dst_p[i] = -src_p[i];
  if (weights && P->linearization != 4 && P->physical_type != 1)
    orc_ns_pressure_projection(h->np, dst_p, weights, modes);
  return 0;
}
