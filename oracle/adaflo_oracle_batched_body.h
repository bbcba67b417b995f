/* adaflo_oracle_batched_body.h -- body of the cell-batched CPU vmult for ONE velocity degree BK and ONE form of the 1D
 * kernels BEO (0 plain, 1 even-odd); included once per pair by adaflo_oracle_batched.c, so that every loop bound and the
 * choice of the 1D kernel are compile-time constants also inside the outlined OpenMP regions.
 * TEST INFRASTRUCTURE ONLY. */
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(CAT(CAT(name, _k), BK), CAT(_eo, BEO))

enum
{
  FN(K)   = BK,
  FN(N)   = BK + 1, /* quadrature points and velocity nodes per direction */
  FN(NP)  = BK,     /* pressure nodes per direction */
  FN(N3)  = (BK + 1) * (BK + 1) * (BK + 1),
  FN(NP3) = BK * BK * BK
};

/* quadrature-point operation on W cells at once, source/navier_stokes_matrix.cc:702-893 (constant coefficients) */
static inline __attribute__((always_inline)) void FN(quad_point)(const batched_consts *c, const vd *L, const double jxw,
                                                                 vd val[3], vd g[3][3], vd *pres_io)
{
  const vd div = g[0][0] + g[1][1] + g[2][2];
  vd       conv[3];
  for (int d = 0; d < 3; ++d)
    conv[d] = val[d] * 0.;
  if (!c->stokes)
    {
      for (int d = 0; d < 3; ++d)
        conv[d] = val[d] * c->w0;
      if (c->linearization == 0)
        {
          const vd f1 = c->beta * div, f2 = c->beta * (L[3] + L[7] + L[11]);
          for (int d = 0; d < 3; ++d)
            {
              vd res = f1 * L[d] + f2 * val[d];
              for (int e = 0; e < 3; ++e)
                res += L[e] * g[d][e] + val[e] * L[3 + 3 * d + e];
              conv[d] += c->tau1 * res;
            }
        }
      else if (c->linearization != 3)
        for (int d = 0; d < 3; ++d)
          {
            vd res = c->beta * L[3] * val[d];
            for (int e = 0; e < 3; ++e)
              res += L[e] * g[d][e];
            conv[d] += c->tau1 * res;
          }
      for (int d = 0; d < 3; ++d)
        conv[d] = conv[d] * c->density - c->damping * val[d];
    }
  const vd pres = *pres_io;
  *pres_io      = -div * jxw;
  for (int d = 0; d < 3; ++d)
    for (int e = d + 1; e < 3; ++e)
      {
        const vd sym = c->tmu * (g[d][e] + g[e][d]);
        g[d][e] = g[e][d] = sym;
      }
  for (int d = 0; d < 3; ++d)
    g[d][d] = 2. * c->tmu * g[d][d] + c->tau_grad_div * div - pres;
  for (int d = 0; d < 3; ++d)
    val[d] = conv[d] * jxw;
}

static void FN(batched_cells)(const batched_handle *h, const batched_consts *c, const double *src_u, const double *src_p,
                              double *dst_u, double *dst_p, const int colour)
{
  enum
  {
    k   = FN(K),
    n   = FN(N),
    ndp = FN(NP),
    n3  = FN(N3),
    np3 = FN(NP3)
  };
  const int   p   = k - 1;
  const long *nnu = h->nnu, *nnp = h->nnp;
  const long  b0 = h->colour_first[colour], b1 = h->colour_first[colour + 1];
#pragma omp parallel
  {
    vd ul[3][n3], pl[n3], t1[n3], t2[n3], vu[3][n3], gu[3][3][n3], vp[n3];
#pragma omp for schedule(static)
    for (long b = b0; b < b1; ++b)
      {
        const int  *cells = h->batch_cells + b * 3 * W; /* [lane][cx cy cz]; padded lanes repeat lane 0 */
        const int   nlane = h->batch_lanes[b];
        const int   plain = !h->batch_constrained[b];
        long        ubase[W], pbase[W];
        for (int l = 0; l < W; ++l)
          {
            const int cx = cells[3 * l], cy = cells[3 * l + 1], cz = cells[3 * l + 2];
            ubase[l] = ((long)cx * k + nnu[0] * ((long)cy * k + nnu[1] * (long)cz * k)) * 3;
            pbase[l] = (long)cx * p + nnp[0] * ((long)cy * p + nnp[1] * (long)cz * p);
          }
        /* gather (constraints resolved: read_dof_values) */
        for (int kk = 0; kk < n; ++kk)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
              {
                const long off = (i + nnu[0] * (j + nnu[1] * (long)kk)) * 3;
                const int  li  = i + n * (j + n * kk);
                for (int l = 0; l < W; ++l)
                  {
                    const double *s = src_u + ubase[l] + off;
                    if (plain)
                      {
                        ul[0][li][l] = s[0];
                        ul[1][li][l] = s[1];
                        ul[2][li][l] = s[2];
                      }
                    else
                      {
                        const uint8_t *m = h->con_u + ubase[l] + off;
                        ul[0][li][l]     = m[0] ? 0. : s[0];
                        ul[1][li][l]     = m[1] ? 0. : s[1];
                        ul[2][li][l]     = m[2] ? 0. : s[2];
                      }
                  }
              }
        for (int kk = 0; kk < ndp; ++kk)
          for (int j = 0; j < ndp; ++j)
            for (int i = 0; i < ndp; ++i)
              {
                const long off = i + nnp[0] * (j + nnp[1] * (long)kk);
                const int  li  = i + ndp * (j + ndp * kk);
                for (int l = 0; l < W; ++l)
                  pl[li][l] = (h->con_p && h->con_p[pbase[l] + off]) ? 0. : src_p[pbase[l] + off];
              }
        /* evaluate: values at the quadrature points, then collocation derivatives */
        for (int d = 0; d < 3; ++d)
          {
            apply_line(&h->Su, 0, BEO, n, n, 0, n, n, n, ul[d], t1, 0);
            apply_line(&h->Su, 0, BEO, n, n, 1, n, n, n, t1, t2, 0);
            apply_line(&h->Su, 0, BEO, n, n, 2, n, n, n, t2, vu[d], 0);
            for (int e = 0; e < 3; ++e)
              apply_line(&h->Dc, 0, BEO, n, n, e, n, n, n, vu[d], gu[d][e], 0);
          }
        apply_line(&h->Sp, 0, BEO, n, ndp, 0, ndp, ndp, ndp, pl, t1, 0);
        apply_line(&h->Sp, 0, BEO, n, ndp, 1, n, ndp, ndp, t1, t2, 0);
        apply_line(&h->Sp, 0, BEO, n, ndp, 2, n, n, ndp, t2, vp, 0);
        /* quadrature-point loop on the stored state of the batch */
        const vd *Lb = h->lin ? h->lin + (size_t)b * n3 * 12 : NULL;
        for (int q = 0; q < n3; ++q)
          {
            vd val[3], g[3][3];
            for (int d = 0; d < 3; ++d)
              {
                val[d] = vu[d][q];
                for (int e = 0; e < 3; ++e)
                  g[d][e] = gu[d][e][q] * c->ih[e];
              }
            FN(quad_point)(c, Lb ? Lb + (size_t)q * 12 : NULL, h->jxw[q], val, g, &vp[q]);
            for (int d = 0; d < 3; ++d)
              {
                vu[d][q] = val[d];
                for (int e = 0; e < 3; ++e)
                  gu[d][e][q] = g[d][e] * (h->jxw[q] * c->ih[e]);
              }
          }
        /* integrate */
        for (int d = 0; d < 3; ++d)
          {
            for (int e = 0; e < 3; ++e)
              apply_line(&h->Dc, 1, BEO, n, n, e, n, n, n, gu[d][e], vu[d], 1);
            apply_line(&h->Su, 1, BEO, n, n, 2, n, n, n, vu[d], t1, 0);
            apply_line(&h->Su, 1, BEO, n, n, 1, n, n, n, t1, t2, 0);
            apply_line(&h->Su, 1, BEO, n, n, 0, n, n, n, t2, ul[d], 0);
          }
        /* scatter-add (cells of one colour share no node; padded lanes are skipped) */
        for (int kk = 0; kk < n; ++kk)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
              {
                const long off = (i + nnu[0] * (j + nnu[1] * (long)kk)) * 3;
                const int  li  = i + n * (j + n * kk);
                for (int l = 0; l < nlane; ++l)
                  {
                    double *t = dst_u + ubase[l] + off;
                    if (plain)
                      {
                        t[0] += ul[0][li][l];
                        t[1] += ul[1][li][l];
                        t[2] += ul[2][li][l];
                      }
                    else
                      {
                        const uint8_t *m = h->con_u + ubase[l] + off;
                        for (int d = 0; d < 3; ++d)
                          if (!m[d])
                            t[d] += ul[d][li][l];
                      }
                  }
              }
        if (c->linearization != 4)
          {
            apply_line(&h->Sp, 1, BEO, n, ndp, 2, n, n, n, vp, t1, 0);
            apply_line(&h->Sp, 1, BEO, n, ndp, 1, n, n, ndp, t1, t2, 0);
            apply_line(&h->Sp, 1, BEO, n, ndp, 0, n, ndp, ndp, t2, pl, 0);
            for (int kk = 0; kk < ndp; ++kk)
              for (int j = 0; j < ndp; ++j)
                for (int i = 0; i < ndp; ++i)
                  {
                    const long off = i + nnp[0] * (j + nnp[1] * (long)kk);
                    const int  li  = i + ndp * (j + ndp * kk);
                    for (int l = 0; l < nlane; ++l)
                      if (!(h->con_p && h->con_p[pbase[l] + off]))
                        dst_p[pbase[l] + off] += pl[li][l];
                  }
          }
      }
  }
}
#undef FN
#undef CAT
#undef CAT2
