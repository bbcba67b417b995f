// Fast cosine transforms for the fast-diagonalisation inverses of degree-1 spaces (csrc/fdm.hip).
//
// On a uniform grid with natural ends the generalised eigenvectors of the 1D pair (K, M) of FE_Q(1) are
// v_k(j) = cos(pi j k / N), j, k = 0 .. N (fdm.hip: linear_eig), so nodes -> modes and modes -> nodes are both the
// plain cosine sum  y_k = sum_{j = 0}^{N} cos(pi j k / N) x_j  (DCT-I without the customary half weights at the ends)
// followed / preceded by a scaling.  For N a power of two this kernel computes it in O(N log N) instead of the
// (N + 1)^2 of the matrix product:
//   e = even extension of x to length 2N;  z_j = e_{2j} + i e_{2j+1}, j = 0 .. N-1;  Z = FFT_N(z)
//   E_k = (Z_k + conj Z_{N-k}) / 2 + w^k (Z_k - conj Z_{N-k}) / (2i),  w = exp(-i pi / N)   (real: e is real and even)
//   y_k = (E_k + x_0 + (-1)^k x_N) / 2
// The FFT is an in-place decimation-in-frequency transform in LDS, radix 4 (one radix-2 stage at the end if log2 N is
// odd); its result stays in digit-reversed order and the step that forms E reads Z_k where it lies.
//
// A workgroup of 256 threads transforms LB = 4096 / N lines at once (64 KB of complex numbers).  Lines are contiguous
// (axis 0), strided by nx (axis 1: line = x + nx z) or by nx ny (axis 2: line = x + nx y); the strided passes read and
// write LB neighbouring lines as runs of LB consecutive doubles.  The z pass can be FUSED: forward transform, the scaling
// of the mode coefficients 1 / (c_m + c_l (lx + ly + lz)) times the squared normalisation of the three eigenvectors, and
// the transform back, with the line never leaving LDS.
//
// This header includes nothing: csrc/fdm.hip compiles it for the device, tests/emu/dct_emu.cpp on the host lane emulator.
#pragma once

namespace adaflo_hip
{
  namespace dct
  {
    constexpr int NT    = 256;  // threads of a workgroup
    constexpr int NCPLX = 4096; // complex numbers of a batch: LB lines x N

    struct alignas(16) cplx
    {
      double re, im;
    };

    struct DctArgs
    {
      const double *in;
      double       *out;
      const double *tw;      // [N + 1][2]: exp(-i pi m / N)
      long          n_lines; // lines of the whole field
      int           axis;    // 0: x (contiguous), 1: y, 2: z
      int           nx, ny, nz;
      // FUSED (axis 2): scaling between the two transforms; a*: squared normalisation of mode k, l*: eigenvalue
      const double *lx, *ly, *lz, *ax, *ay, *az;
      double        cm, cl, eps;
    };

    template <int LOG2N>
    struct Geo
    {
      static constexpr int N = 1 << LOG2N, n = N + 1, LB = NCPLX / N;
      static constexpr int n_r4 = LOG2N / 2, n_stages = n_r4 + (LOG2N & 1);
      static_assert(LOG2N >= 6 && LOG2N <= 10, "64 .. 1024 intervals per line");
      // LDS, in doubles: Z (the raw lines R[LB][n] overlay its start), twiddles, the two end values of every line
      static constexpr int L_Z = 0, L_T = 2 * NCPLX, L_X0 = L_T + 2 * (N + 1), L_XN = L_X0 + LB, L_TOTAL = L_XN + LB;
      static_assert(LB * n <= 2 * NCPLX, "raw overlay");
      // where X[k] lies after the stages (digit reversal of the mixed radix 4, 4, ..., [2])
      static constexpr int pos_of(int k)
      {
        int p = 0, span = N;
        for (int s = 0; s < n_stages; ++s)
          {
            const int r = s < n_r4 ? 4 : 2;
            span /= r;
            p += (k % r) * span;
            k /= r;
          }
        return p;
      }
    };

    __device__ __forceinline__ cplx cmul(const cplx a, const cplx b)
    {
      return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    }

    // exp(-2 pi i m / N), 0 <= m < N, from the half-circle table T[j] = exp(-i pi j / N), j = 0 .. N
    template <int N>
    __device__ __forceinline__ cplx twiddle(const cplx *T, const int m)
    {
      const int idx = 2 * m;
      if (idx <= N)
        return T[idx];
      const cplx t = T[idx - N];
      return cplx{-t.re, -t.im};
    }

    // one radix-4 butterfly (forward transform) with the twiddles of position j in a sub-transform of length L
    template <int N, int L>
    __device__ __forceinline__ void radix4(cplx (&a)[4], const cplx *T, const int j)
    {
      const cplx b0{a[0].re + a[2].re, a[0].im + a[2].im}, b1{a[0].re - a[2].re, a[0].im - a[2].im};
      const cplx b2{a[1].re + a[3].re, a[1].im + a[3].im};
      const cplx b3{a[1].im - a[3].im, -(a[1].re - a[3].re)}; // -i (a1 - a3)
      a[0] = cplx{b0.re + b2.re, b0.im + b2.im};
      cplx y1{b1.re + b3.re, b1.im + b3.im}, y2{b0.re - b2.re, b0.im - b2.im}, y3{b1.re - b3.re, b1.im - b3.im};
      if (L > 4)
        {
          const int m = j * (N / L);
          y1          = cmul(y1, twiddle<N>(T, m));
          y2          = cmul(y2, twiddle<N>(T, 2 * m));
          y3          = cmul(y3, twiddle<N>(T, 3 * m));
        }
      a[1] = y1, a[2] = y2, a[3] = y3;
    }

    // stages 1 .. of the FFT, in place in Z (stage 0 is part of dct_lines: it reads the raw lines)
    template <int LOG2N, int S>
    __device__ __forceinline__ void fft_stage(cplx *Z, const cplx *T)
    {
      using G             = Geo<LOG2N>;
      constexpr int N     = G::N;
      constexpr int L     = N >> (2 * S); // length of the sub-transforms this stage splits
      const int     t     = threadIdx.x;
      if constexpr (S < G::n_r4)
        {
#pragma unroll
          for (int i = 0; i < NCPLX / 4 / NT; ++i)
            {
              const int b = t + NT * i, line = b >> (LOG2N - 2), r = b & (N / 4 - 1);
              const int j = r & (L / 4 - 1), g = (r / (L / 4)) * L;
              cplx     *z = Z + line * N + g + j;
              cplx      a[4];
#pragma unroll
              for (int p = 0; p < 4; ++p)
                a[p] = z[p * (L / 4)];
              radix4<N, L>(a, T, j);
#pragma unroll
              for (int p = 0; p < 4; ++p)
                z[p * (L / 4)] = a[p];
            }
        }
      else // the radix-2 stage at the end: L = 2, no twiddles
        {
#pragma unroll
          for (int i = 0; i < NCPLX / 2 / NT; ++i)
            {
              const int  b = t + NT * i;
              cplx      *z = Z + 2 * b;
              const cplx a0 = z[0], a1 = z[1];
              z[0] = cplx{a0.re + a1.re, a0.im + a1.im};
              z[1] = cplx{a0.re - a1.re, a0.im - a1.im};
            }
        }
      __syncthreads();
    }
    template <int LOG2N, int S>
    struct Stages
    {
      static __device__ __forceinline__ void run(cplx *Z, const cplx *T)
      {
        if constexpr (S < Geo<LOG2N>::n_stages)
          {
            fft_stage<LOG2N, S>(Z, T);
            Stages<LOG2N, S + 1>::run(Z, T);
          }
      }
    };

    // The cosine sums of the LB raw lines R[line][n] at the start of the LDS area; the results are returned in
    // registers: item it = t + NT i, i < NI, is (line, k) = (it / (N/2 + 1), it % (N/2 + 1)) and carries y_k and y_{N-k}.
    // On return every thread has passed a barrier after its last LDS read: the caller may overwrite R.
    template <int LOG2N>
    struct Items
    {
      static constexpr int N = 1 << LOG2N, per_line = N / 2 + 1, total = Geo<LOG2N>::LB * per_line, NI = (total + NT - 1) / NT;
    };
    template <int LOG2N>
    __device__ __forceinline__ void dct_lines(double *lds, double (&yk)[Items<LOG2N>::NI], double (&ym)[Items<LOG2N>::NI])
    {
      using G         = Geo<LOG2N>;
      using I         = Items<LOG2N>;
      constexpr int N = G::N, n = G::n;
      const int     t = threadIdx.x;
      double       *R = lds + G::L_Z;
      cplx         *Z = reinterpret_cast<cplx *>(lds + G::L_Z);
      const cplx   *T = reinterpret_cast<const cplx *>(lds + G::L_T);
      // ---- stage 0 on z_j = e_{2j} + i e_{2j+1} taken from the raw lines (e_m = x_m, m <= N; x_{2N-m} beyond)
      {
        cplx a[NCPLX / 4 / NT][4];
#pragma unroll
        for (int i = 0; i < NCPLX / 4 / NT; ++i)
          {
            const int     b = t + NT * i, line = b >> (LOG2N - 2), j = b & (N / 4 - 1);
            const double *x = R + line * n;
#pragma unroll
            for (int p = 0; p < 4; ++p)
              {
                const int m0 = 2 * (j + p * (N / 4)), m1 = m0 + 1;
                a[i][p]      = cplx{x[m0 <= N ? m0 : 2 * N - m0], x[m1 <= N ? m1 : 2 * N - m1]};
              }
            radix4<N, N>(a[i], T, j);
          }
        if (t < G::LB)
          {
            lds[G::L_X0 + t] = R[t * n];
            lds[G::L_XN + t] = R[t * n + N];
          }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCPLX / 4 / NT; ++i)
          {
            const int b = t + NT * i, line = b >> (LOG2N - 2), j = b & (N / 4 - 1);
#pragma unroll
            for (int p = 0; p < 4; ++p)
              Z[line * N + j + p * (N / 4)] = a[i][p];
          }
        __syncthreads();
      }
      Stages<LOG2N, 1>::run(Z, T);
      // ---- E_k, E_{N-k} from Z_k and Z_{N-k} (digit-reversed positions), then y
#pragma unroll
      for (int i = 0; i < I::NI; ++i)
        {
          const int it = t + NT * i;
          yk[i] = ym[i] = 0.;
          if (it < I::total)
            {
              const int    line = it / I::per_line, k = it - line * I::per_line;
              const cplx   zk = Z[line * N + G::pos_of(k)], zm = Z[line * N + G::pos_of((N - k) & (N - 1))];
              const cplx   w   = T[k];
              const double are = 0.5 * (zk.re + zm.re), bre = 0.5 * (zk.im + zm.im), bim = -0.5 * (zk.re - zm.re);
              const double p   = w.re * bre - w.im * bim;
              const double x0 = lds[G::L_X0 + line], xn = lds[G::L_XN + line];
              const double ends = (k & 1) ? x0 - xn : x0 + xn;
              yk[i]             = 0.5 * (are + p + ends);
              ym[i]             = 0.5 * (are - p + ends);
            }
        }
      __syncthreads();
    }

    template <int LOG2N, bool FUSED>
    __device__ __forceinline__ void dct_body(const DctArgs &A, double *lds)
    {
      using G         = Geo<LOG2N>;
      using I         = Items<LOG2N>;
      constexpr int N = G::N, n = G::n, LB = G::LB;
      const int     t = threadIdx.x;
      double       *R = lds + G::L_Z;
      const long    l0 = (long)blockIdx.x * LB;
      const int     nl = (int)(A.n_lines - l0 < LB ? A.n_lines - l0 : LB);
      for (int e = t; e < 2 * (N + 1); e += NT)
        lds[G::L_T + e] = A.tw[e];
      // ---- the lines of this batch -> R[line][n] (absent lines: zeros)
      long      base   = 0; // axes 1, 2: a thread serves ONE line c = t % LB of the batch (LB divides NT)
      long      stride = 1;
      const int c = t & (LB - 1), r0 = t / LB;
      if (A.axis == 0)
        {
          const long    count = (long)nl * n;
          const double *src   = A.in + l0 * n;
          for (int e = t; e < LB * n; e += NT)
            R[e] = e < count ? src[e] : 0.;
        }
      else
        {
          const long l = l0 + c;
          if (A.axis == 1)
            base = (l / A.nx) * ((long)A.nx * A.ny) + l % A.nx, stride = A.nx;
          else
            base = l, stride = (long)A.nx * A.ny;
          for (int r = r0; r < n; r += NT / LB)
            R[c * n + r] = c < nl ? A.in[base + r * stride] : 0.;
        }
      __syncthreads();
      double yk[I::NI], ym[I::NI];
      dct_lines<LOG2N>(lds, yk, ym);
      if (FUSED)
        {
          // scale the mode coefficients and transform back
#pragma unroll
          for (int i = 0; i < I::NI; ++i)
            {
              const int it = t + NT * i;
              if (it < I::total)
                {
                  const int  line = it / I::per_line, k = it - line * I::per_line;
                  const long l = l0 + line;
                  double     sk = 0., sm = 0.;
                  if (line < nl)
                    {
                      const int    x = (int)(l % A.nx), y = (int)(l / A.nx);
                      const double lxy = A.lx[x] + A.ly[y], axy = A.ax[x] * A.ay[y];
                      const double dk = A.cm + A.cl * (lxy + A.lz[k]), dm = A.cm + A.cl * (lxy + A.lz[N - k]);
                      sk = (dk > A.eps || dk < -A.eps) ? axy * A.az[k] / dk : 0.;
                      sm = (dm > A.eps || dm < -A.eps) ? axy * A.az[N - k] / dm : 0.;
                    }
                  R[line * n + k]     = yk[i] * sk;
                  R[line * n + N - k] = ym[i] * sm;
                }
            }
          __syncthreads();
          dct_lines<LOG2N>(lds, yk, ym);
        }
#pragma unroll
      for (int i = 0; i < I::NI; ++i)
        {
          const int it = t + NT * i;
          if (it < I::total)
            {
              const int line = it / I::per_line, k = it - line * I::per_line;
              R[line * n + k]     = yk[i];
              R[line * n + N - k] = ym[i];
            }
        }
      __syncthreads();
      if (A.axis == 0)
        {
          const long count = (long)nl * n;
          double    *dst   = A.out + l0 * n;
          for (int e = t; e < count; e += NT)
            dst[e] = R[e];
        }
      else if (c < nl)
        for (int r = r0; r < n; r += NT / LB)
          A.out[base + r * stride] = R[c * n + r];
    }
  } // namespace dct
} // namespace adaflo_hip
