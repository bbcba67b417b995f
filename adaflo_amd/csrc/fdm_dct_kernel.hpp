// Fast cosine transforms for the fast-diagonalisation inverses of degree-1 spaces (csrc/fdm.hip).
//
// On a uniform grid with natural ends the generalised eigenvectors of the 1D pair (K, M) of FE_Q(1) are
// v_k(j) = cos(pi j k / N), j, k = 0 .. N (fdm.hip: linear_eig), so nodes -> modes and modes -> nodes are both the
// plain cosine sum  y_k = sum_{j = 0}^{N} cos(pi j k / N) x_j  (DCT-I without the customary half weights at the ends)
// followed / preceded by a scaling.  For N = 2^m, 3 2^m or 5 2^m this kernel computes it in O(N log N) instead of the
// (N + 1)^2 of the matrix product:
//   e = even extension of x to length 2N;  z_j = e_{2j} + i e_{2j+1}, j = 0 .. N-1;  Z = FFT_N(z)
//   E_k = (Z_k + conj Z_{N-k}) / 2 + w^k (Z_k - conj Z_{N-k}) / (2i),  w = exp(-i pi / N)   (real: e is real and even)
//   y_k = (E_k + x_0 + (-1)^k x_N) / 2
// The FFT is an in-place decimation-in-frequency transform in LDS: a radix-3 / radix-5 stage if 3 / 5 divides N, then radix 4 (two
// stages per LDS round trip, one radix-2 stage at the end if needed); its result stays in digit-reversed order and the
// step that forms E reads Z_k where it lies.
//
// A workgroup of 256 threads transforms LB = 4096 / N (rounded down) lines at once (68 KB of complex numbers, two stages per LDS round
// trip).  Lines are contiguous (axis 0), strided by the row pitch (axis 1: line = x + pitch z) or by pitch ny (axis 2:
// line = x + pitch y); the strided passes read and write LB neighbouring lines as runs of LB consecutive doubles.  The z
// pass can be FUSED: forward transform, the scaling of the mode coefficients 1 / (c_m + c_l (lx + ly + lz)) times the
// squared normalisation of the three eigenvectors, and the transform back, with the line never leaving LDS.  The memory
// stream is software-pipelined around the transform (dct_body).  Measurements: DESIGN.md 4.7.
//
// This header includes nothing: csrc/fdm.hip compiles it for the device, tests/emu/dct_emu.cpp on the host lane emulator.
#pragma once

namespace adaflo_hip
{
  namespace dct
  {
#ifndef DCT_EXP
#define DCT_EXP 0 // diagnostic builds (scripts/dev/dct_probe.hip): 1 no transform, 2 no stages after the first pair, 4 no
                  // digit-reversed reads, 8 no global loads, 16 no global stores -- results wrong by construction
#endif
    constexpr int NT    = 256;  // threads of a workgroup
    constexpr int NCPLX = 4096; // complex numbers of a batch: LB lines x N

    struct alignas(16) cplx
    {
      double re, im;
    };
    // hides a value from the optimiser (no instruction): what is derived from it is not hoisted out of a loop
#if defined(__HIPCC__)
    __device__ __forceinline__ void opaque(int &v) { asm volatile("" : "+v"(v)); }
#else
    inline void opaque(int &) {}
#endif
    // 1 / d to the last bit or two: hardware estimate + two Newton steps (the IEEE division sequence, unrolled 17 times
    // in the scaling of the fused pass, needed 450 B of scratch per lane)
#if defined(__HIPCC__)
    __device__ __forceinline__ double fast_rcp(const double d)
    {
      double r = __builtin_amdgcn_rcp(d);
      r        = r * (2. - d * r);
      return r * (2. - d * r);
    }
#else
    inline double fast_rcp(const double d) { return 1. / d; }
#endif

    struct DctArgs
    {
      const double *in;
      double       *out;
      const double *tw;      // [N + 1][2]: exp(-i pi m / N)
      long          n_lines; // lines of the whole field
      int           axis;    // 0: x (contiguous), 1: y, 2: z
      int           nx, ny, nz;
      // elements per x-row of the input and of the output array (>= nx).  The intermediate arrays of an application are
      // padded to a multiple of 16 so that the runs of LB consecutive doubles the strided passes move are aligned; a
      // strided pass has ONE pitch (pitch_in == pitch_out) and enumerates its lines over the padded rows, lines with
      // x >= nx do not exist
      int           pitch_in, pitch_out;
      // FUSED (axis 2): scaling between the two transforms; a*: squared normalisation of mode k, l*: eigenvalue
      const double *lx, *ly, *lz, *ax, *ay, *az;
      double        cm, cl, eps;
      // a second inverse applied to the same source and added (cm2 == cl2 == 0: none): both are diagonal in the same modes
      double        cm2, cl2, eps2;
    };

    // N = F 2^m intervals per line, F = 1, 3 or 5 (the reference's meshes are 5 x 10 coarse cells refined: 5 2^m), 2^m >= 16:
    // an optional radix-3 / radix-5 stage, then radix-4 stages in pairs, then what is left (radix 4, radix 2)
    constexpr int dct_odd_factor(const int N) { return N % 5 == 0 ? 5 : (N % 3 == 0 ? 3 : 1); }
    constexpr bool dct_length_supported(const int N)
    {
      const int M = N / dct_odd_factor(N);
      return N >= 64 && N <= 1024 && M >= 16 && (M & (M - 1)) == 0;
    }
    template <int N_>
    struct Geo
    {
      static constexpr int N = N_, n = N + 1, F = dct_odd_factor(N_), M = N / F;
      static_assert(dct_length_supported(N), "N = 2^m, 3 2^m or 5 2^m, 64 <= N <= 1024, 2^m >= 16");
      static constexpr int log2(const int v) { return v <= 1 ? 0 : 1 + log2(v / 2); }
      static constexpr int LOG2M = log2(M), n_r4 = LOG2M / 2, n_stages = n_r4 + (LOG2M & 1); // stages of the length-M part
      // a thread serves one line: TPL threads per line, LB lines per batch, LB TPL <= NT threads are active
      // (F = 5: at most three radix-5 butterflies per thread -- their results wait in registers for a barrier; F = 3: five)
      static constexpr int TPL = N / 16, LB_ = NT / TPL, NBMAX = F == 5 ? 3 : 5,
                           LB = F > 1 && LB_ > NBMAX * NT * F / N ? NBMAX * NT * F / N : LB_, NACT = LB * TPL, ZC = LB * N;
      static_assert(N % 16 == 0 && ZC <= NCPLX && LB >= 1, "batch");
      // LDS, in doubles: Z (one complex number of padding behind every 16: the strides 16, 64, ... of the later stages and
      // of the digit-reversed reads would otherwise fall on one bank; the raw lines R[LB][n] overlay its start), twiddles,
      // (x_0 + x_N, x_0 - x_N) of every line, (lx + ly, ax ay) of every line (fused pass)
      static constexpr int L_Z = 0, L_T = 2 * (ZC + ZC / 16 + 1), L_E = L_T + 2 * (N + 1), L_F = L_E + 2 * LB, L_DUMP = L_F + 2 * LB,
                           L_TOTAL = L_DUMP + 2;
      static_assert(LB * n <= 2 * ZC, "raw overlay");
      static __device__ __forceinline__ constexpr int pad(const int i) { return i + (i >> 4); }
      // where X[k] lies after the stages (digit reversal of the mixed radix [5], 4, 4, ..., [2])
      static __device__ __forceinline__ constexpr int pos_of_m(int k) // the length-M part
      {
        int p = 0, span = M;
        for (int s = 0; s < n_stages; ++s)
          {
            const int r = s < n_r4 ? 4 : 2;
            span /= r;
            p += (k % r) * span;
            k /= r;
          }
        return p;
      }
      static __device__ __forceinline__ constexpr int pos_of(const int k)
      {
        return F == 1 ? pos_of_m(k) : (k % F) * M + pos_of_m(k / F);
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

    // radix-4 butterfly of the forward transform, twiddles not applied: a <- (y0, y1, y2, y3)
    __device__ __forceinline__ void radix4(cplx &a0, cplx &a1, cplx &a2, cplx &a3)
    {
      const cplx b0{a0.re + a2.re, a0.im + a2.im}, b1{a0.re - a2.re, a0.im - a2.im};
      const cplx b2{a1.re + a3.re, a1.im + a3.im};
      const cplx b3{a1.im - a3.im, -(a1.re - a3.re)}; // -i (a1 - a3)
      a0 = cplx{b0.re + b2.re, b0.im + b2.im};
      a1 = cplx{b1.re + b3.re, b1.im + b3.im};
      a2 = cplx{b0.re - b2.re, b0.im - b2.im};
      a3 = cplx{b1.re - b3.re, b1.im - b3.im};
    }
    // exp(-2 pi i e / 16), e = 0 .. 9
    __device__ __forceinline__ cplx w16(const int e)
    {
      constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, r = 0.70710678118654752440;
      switch (e)
        {
          case 0:
            return cplx{1., 0.};
          case 1:
            return cplx{c1, -s1};
          case 2:
            return cplx{r, -r};
          case 3:
            return cplx{s1, -c1};
          case 4:
            return cplx{0., -1.};
          case 5:
            return cplx{-s1, -c1};
          case 6:
            return cplx{-r, -r};
          case 7:
            return cplx{-c1, -s1};
          case 8:
            return cplx{-1., 0.};
          default:
            return cplx{-c1, s1};
        }
    }

    // TWO radix-4 stages (sub-transform lengths L and L / 4) on 16 numbers in registers: a[p][q] is the element at
    // g + j + p L/4 + q L/16, 0 <= j < L/16.  Stage A: for every q the butterfly over p at position j + q L/16 of the
    // length-L transform, whose twiddles are w_L^(j m) w_16^(q m) (the second factor a constant); stage B: for every p the
    // butterfly over q at position j of a length-L/4 transform.
    template <int N, int L>
    __device__ __forceinline__ void radix4x4(cplx (&a)[4][4], const cplx *T, const int j)
    {
      cplx tw[3];
      if (L > 16)
        {
          const int m = j * (N / L);
          tw[0] = twiddle<N>(T, m), tw[1] = twiddle<N>(T, 2 * m), tw[2] = twiddle<N>(T, 3 * m);
        }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        {
          radix4(a[0][q], a[1][q], a[2][q], a[3][q]);
#pragma unroll
          for (int m = 1; m < 4; ++m)
            {
              if (q > 0)
                a[m][q] = cmul(a[m][q], w16(q * m));
              if (L > 16)
                a[m][q] = cmul(a[m][q], tw[m - 1]);
            }
        }
      if (L / 4 > 4)
        {
          const int m = j * (N / (L / 4));
          tw[0] = twiddle<N>(T, m), tw[1] = twiddle<N>(T, 2 * m), tw[2] = twiddle<N>(T, 3 * m);
        }
#pragma unroll
      for (int p = 0; p < 4; ++p)
        {
          radix4(a[p][0], a[p][1], a[p][2], a[p][3]);
          if (L / 4 > 4)
            {
#pragma unroll
              for (int m = 1; m < 4; ++m)
                a[p][m] = cmul(a[p][m], tw[m - 1]);
            }
        }
    }

    // radix-5 butterfly of the forward transform, twiddles not applied
    __device__ __forceinline__ void radix5(cplx (&a)[5])
    {
      constexpr double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410; // cos(2 pi / 5), cos(4 pi / 5)
      constexpr double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;  // sin(2 pi / 5), sin(4 pi / 5)
      const cplx t1{a[1].re + a[4].re, a[1].im + a[4].im}, t2{a[2].re + a[3].re, a[2].im + a[3].im};
      const cplx d1{a[1].re - a[4].re, a[1].im - a[4].im}, d2{a[2].re - a[3].re, a[2].im - a[3].im};
      const cplx m1{a[0].re + c1 * t1.re + c2 * t2.re, a[0].im + c1 * t1.im + c2 * t2.im};
      const cplx m2{a[0].re + c2 * t1.re + c1 * t2.re, a[0].im + c2 * t1.im + c1 * t2.im};
      // -i (s1 d1 + s2 d2) and -i (s2 d1 - s1 d2)
      const cplx n1{s1 * d1.im + s2 * d2.im, -(s1 * d1.re + s2 * d2.re)}, n2{s2 * d1.im - s1 * d2.im, -(s2 * d1.re - s1 * d2.re)};
      a[0] = cplx{a[0].re + t1.re + t2.re, a[0].im + t1.im + t2.im};
      a[1] = cplx{m1.re + n1.re, m1.im + n1.im};
      a[4] = cplx{m1.re - n1.re, m1.im - n1.im};
      a[2] = cplx{m2.re + n2.re, m2.im + n2.im};
      a[3] = cplx{m2.re - n2.re, m2.im - n2.im};
    }

    // radix-3 butterfly of the forward transform, twiddles not applied
    __device__ __forceinline__ void radix3(cplx (&a)[3])
    {
      constexpr double s = 0.86602540378443864676; // sin(2 pi / 3)
      const cplx t{a[1].re + a[2].re, a[1].im + a[2].im}, d{a[1].re - a[2].re, a[1].im - a[2].im};
      const cplx m{a[0].re - 0.5 * t.re, a[0].im - 0.5 * t.im}, n{s * d.im, -s * d.re}; // -i s d
      a[0] = cplx{a[0].re + t.re, a[0].im + t.im};
      a[1] = cplx{m.re + n.re, m.im + n.im};
      a[2] = cplx{m.re - n.re, m.im - n.im};
    }
    __device__ __forceinline__ void radix_odd(cplx (&a)[3]) { radix3(a); }
    __device__ __forceinline__ void radix_odd(cplx (&a)[5]) { radix5(a); }

    // stages S, S + 1 (radix 4 x 4) of the length-M part, in place in Z
    template <int N, int S>
    __device__ __forceinline__ void fft_double_stage(cplx *Z, const cplx *T)
    {
      using G         = Geo<N>;
      constexpr int L = G::M >> (2 * S);
      static_assert(L >= 16, "two radix-4 stages");
      int t = threadIdx.x; // (opaque: the index arithmetic below depends on the thread only and would be hoisted out
      opaque(t);           //  of the batch loop of dct_body -- dozens of registers held, or spilled, for the whole kernel)
      if (G::ZC / 16 == NT || t < G::ZC / 16) // (one 16-point group per thread)
        {
          const int line = t / (N / 16), r = t - line * (N / 16);
          const int j = r & (L / 16 - 1), base = line * N + (r / (L / 16)) * L + j;
          cplx      a[4][4];
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              a[p][q] = Z[G::pad(base + p * (L / 4) + q * (L / 16))];
          radix4x4<N, L>(a, T, j);
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              Z[G::pad(base + p * (L / 4) + q * (L / 16))] = a[p][q];
        }
      __syncthreads();
    }
    // one radix-4 stage S of the length-M part, or the radix-2 stage at its end (S = n_r4), in place in Z
    template <int N, int S>
    __device__ __forceinline__ void fft_stage(cplx *Z, const cplx *T)
    {
      using G         = Geo<N>;
      constexpr int L = G::M >> (2 * S); // length of the sub-transforms this stage splits
      int t = threadIdx.x; // (opaque: the index arithmetic below depends on the thread only and would be hoisted out
      opaque(t);           //  of the batch loop of dct_body -- dozens of registers held, or spilled, for the whole kernel)
      if constexpr (S < G::n_r4)
        {
          constexpr int NB = G::ZC / 4;
#pragma unroll
          for (int i = 0; i < (NB + NT - 1) / NT; ++i)
            {
              const int b = t + NT * i;
              if (NB % NT == 0 || b < NB)
                {
                  const int line = b / (N / 4), r = b - line * (N / 4);
                  const int j = r & (L / 4 - 1), base = line * N + (r / (L / 4)) * L + j;
                  cplx      a[4];
#pragma unroll
                  for (int p = 0; p < 4; ++p)
                    a[p] = Z[G::pad(base + p * (L / 4))];
                  radix4(a[0], a[1], a[2], a[3]);
                  if (L > 4)
                    {
                      const int m = j * (N / L);
#pragma unroll
                      for (int q = 1; q < 4; ++q)
                        a[q] = cmul(a[q], twiddle<N>(T, q * m));
                    }
#pragma unroll
                  for (int p = 0; p < 4; ++p)
                    Z[G::pad(base + p * (L / 4))] = a[p];
                }
            }
        }
      else // the radix-2 stage at the end: L = 2, no twiddles
        {
          constexpr int NB = G::ZC / 2;
#pragma unroll
          for (int i = 0; i < (NB + NT - 1) / NT; ++i)
            {
              const int b = t + NT * i;
              if (NB % NT == 0 || b < NB)
                {
                  const cplx a0 = Z[G::pad(2 * b)], a1 = Z[G::pad(2 * b + 1)];
                  Z[G::pad(2 * b)]     = cplx{a0.re + a1.re, a0.im + a1.im};
                  Z[G::pad(2 * b + 1)] = cplx{a0.re - a1.re, a0.im - a1.im};
                }
            }
        }
      __syncthreads();
    }
    // stages S ... of the length-M part: pairs of radix-4 stages as long as there are two, then what is left
    template <int N, int S>
    struct Stages
    {
      static __device__ __forceinline__ void run(cplx *Z, const cplx *T)
      {
        using G = Geo<N>;
        if constexpr (S + 1 < G::n_r4)
          {
            fft_double_stage<N, S>(Z, T);
            Stages<N, S + 2>::run(Z, T);
          }
        else if constexpr (S < G::n_stages)
          {
            fft_stage<N, S>(Z, T);
            Stages<N, S + 1>::run(Z, T);
          }
      }
    };

    // The FFT of the LB raw lines R[line][n] at the start of the LDS area: Z_k of line c lies at pad(c N + pos_of(k)),
    // (x_0 + x_N, x_0 - x_N) of the line in E[c].  On return every thread has passed a barrier after its last access.
    template <int N>
    __device__ __forceinline__ void dct_lines(double *lds)
    {
      using G         = Geo<N>;
      constexpr int n = G::n, M = G::M;
      int t = threadIdx.x; // (opaque: the index arithmetic below depends on the thread only and would be hoisted out
      opaque(t);           //  of the batch loop of dct_body -- dozens of registers held, or spilled, for the whole kernel)
      double       *R = lds + G::L_Z;
      cplx         *Z = reinterpret_cast<cplx *>(lds + G::L_Z);
      const cplx   *T = reinterpret_cast<const cplx *>(lds + G::L_T);
      cplx         *E = reinterpret_cast<cplx *>(lds + G::L_E);
      if (DCT_EXP & 1)
        {
          __syncthreads();
          return;
        }
      // z_j = e_{2j} + i e_{2j+1} from the raw lines (e_m = x_m, m <= N; x_{2N-m} beyond)
      auto zraw = [&](const double *x, const int j) {
        const int m0 = 2 * j, m1 = m0 + 1;
        return cplx{x[m0 <= N ? m0 : 2 * N - m0], x[m1 <= N ? m1 : 2 * N - m1]};
      };
      cplx ends{0., 0.};
      if (t < G::LB)
        ends = cplx{R[t * n] + R[t * n + N], R[t * n] - R[t * n + N]};
      if constexpr (G::F == 1)
        {
          // ---- stages 0, 1 on the raw lines
          const int line = t / (N / 16), j = t - line * (N / 16);
          cplx      a[4][4];
          if (G::ZC / 16 == NT || t < G::ZC / 16)
            {
#pragma unroll
              for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  a[p][q] = zraw(R + line * n, j + p * (N / 4) + q * (N / 16));
              radix4x4<N, N>(a, T, j);
            }
          __syncthreads();
          if (t < G::LB)
            E[t] = ends;
          if (G::ZC / 16 == NT || t < G::ZC / 16)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                Z[G::pad(line * N + j + p * (N / 4) + q * (N / 16))] = a[p][q];
          __syncthreads();
          if (!(DCT_EXP & 2))
            Stages<N, 2>::run(Z, T);
        }
      else
        {
          // ---- the radix-3 / radix-5 stage on the raw lines: sub-transform q of length M gets y_q w_N^(q j) at position j
          constexpr int F = G::F, NB = G::ZC / F, NIT = (NB + NT - 1) / NT;
          cplx          a[NIT][F];
#pragma unroll
          for (int i = 0; i < NIT; ++i)
            {
              const int b = t + NT * i;
              if (b < NB)
                {
                  const int line = b / M, j = b - line * M;
#pragma unroll
                  for (int p = 0; p < F; ++p)
                    a[i][p] = zraw(R + line * n, j + p * M);
                  radix_odd(a[i]);
#pragma unroll
                  for (int q = 1; q < F; ++q)
                    a[i][q] = cmul(a[i][q], twiddle<N>(T, q * j));
                }
            }
          __syncthreads();
          if (t < G::LB)
            E[t] = ends;
#pragma unroll
          for (int i = 0; i < NIT; ++i)
            {
              const int b = t + NT * i;
              if (b < NB)
                {
                  const int line = b / M, j = b - line * M;
#pragma unroll
                  for (int q = 0; q < F; ++q)
                    Z[G::pad(line * N + q * M + j)] = a[i][q];
                }
            }
          __syncthreads();
          if (!(DCT_EXP & 2))
            Stages<N, 0>::run(Z, T);
        }
    }

    // One workgroup, a grid-stride loop over the batches.  The memory stream is software-pipelined around the transform:
    // the lines of batch b + 1 are loaded into registers and the results of batch b - 1 are stored from registers at the
    // START of the work on batch b, so that both have the whole transform to complete (vector-memory operations retire in
    // order: a wait for the loads is a wait for every older store as well) and the workgroup never sits on its own latency.
    // A thread serves ONE line c of a batch, rows r0, r0 + TPL, ... (TPL = NT / LB = N / 16 threads per line): along the
    // line for AXIS 0 (consecutive lanes read consecutive doubles of a line), across the lines for the strided axes
    // (consecutive lanes read the same row of consecutive lines); every index is base + constant, nothing to keep in
    // registers.  AXIS is a template argument and every load is unconditional -- an absent value is read from element 0,
    // an absent LDS slot is the dump slot: with run-time branches around the loads the same code
    // had 400 basic blocks and the compiler waited for every load where it was issued.
    template <int N, bool FUSED, int AXIS>
    __device__ __forceinline__ void dct_body(const DctArgs &A, double *lds)
    {
      using G           = Geo<N>;
      constexpr int n = G::n, LB = G::LB, TPL = G::TPL;
      static_assert(!FUSED || AXIS == 2, "the fused pass is the z pass");
      constexpr int NLD = 17; // values of a line per thread
      static_assert((NLD - 1) * TPL == N, "rows r0 + i TPL: only r0 = 0 has a last one (r = N)");
      const int  t = threadIdx.x;
      double    *R = lds + G::L_Z;
      const long nbatch = (A.n_lines + LB - 1) / LB;
      for (int e = t; e < 2 * (N + 1); e += NT)
        lds[G::L_T + e] = A.tw[e];
      // (LB TPL <= NT: the last threads of a workgroup may have no line)
      const bool active = G::NACT == NT || t < G::NACT;
      const int  c = !active ? 0 : (AXIS == 0 ? t / TPL : t % LB), r0 = !active ? 1 : (AXIS == 0 ? t % TPL : t / LB);
      const int  pitch  = A.pitch_in;
      const long stride = AXIS == 0 ? 1 : (AXIS == 1 ? (long)pitch : (long)pitch * A.ny);
      struct Pos
      {
        long in, out; // first value of this thread's line; -1: the line does not exist
        int  nl;      // lines of the batch that exist
      };
      auto batch_pos = [&](const long b) {
        const long l0 = b * LB;
        Pos        P;
        P.nl = (int)(A.n_lines - l0 < LB ? A.n_lines - l0 : LB);
        if (AXIS == 0)
          P.in = (l0 + c) * A.pitch_in, P.out = (l0 + c) * A.pitch_out;
        else
          {
            const unsigned l = (unsigned)l0 + c, x = l % (unsigned)pitch, zz = l / (unsigned)pitch; // (fewer than 2^31 lines)
            P.in = P.out = AXIS == 1 ? zz * ((long)pitch * A.ny) + x : (long)l;
            if ((int)x >= A.nx)
              P.in = -1;
          }
        if (c >= P.nl || !active)
          P.in = -1;
        return P;
      };
      auto load = [&](const Pos &P, double (&v)[NLD]) {
        const double *src = A.in + (P.in >= 0 ? P.in + r0 * stride : 0);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
          {
            // (no select on the VALUE: it would be scheduled right behind the load and wait for it.  A line that does not
            // exist transforms whatever element 0 holds; lines do not mix and its results are never stored)
            const bool valid = P.in >= 0 && (i < NLD - 1 || r0 == 0);
            v[i]             = (DCT_EXP & 8) ? 0. : src[valid ? i * TPL * stride : 0];
          }
      };
      auto store = [&](const Pos &P, const double (&v)[NLD]) {
        if (DCT_EXP & 16)
          {
            if (v[0] == 1.2345)
              A.out[t] = 0.;
            return;
          }
        if (P.in < 0)
          return;
        double *dst = A.out + P.out + r0 * stride;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
          if (i < NLD - 1 || r0 == 0)
            dst[i * TPL * stride] = v[i];
      };
      // the same values <-> the raw lines R[line][n] in LDS, and where dct_lines left the results
      double *Rc = R + c * n + r0;
      auto    to_lds = [&](const double (&v)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
          ((i < NLD - 1 || r0 == 0) && active ? Rc[i * TPL] : lds[G::L_DUMP]) = v[i];
      };
      // The cosine sums of this thread's values from the FFT:  E_k = (Z_k + conj Z_{N-k}) / 2 + w^k (Z_k - conj Z_{N-k}) / (2i)
      // (real), y_k = (E_k + x_0 + (-1)^k x_N) / 2 for k = r0 + i TPL < N, and y_N = (Re Z_0 - Im Z_0 + x_0 + x_N) / 2.
      // F = 1: pos_of is linear over disjoint bit fields, so the slots of k and of N - k = (15 - i [+ 1 if r0 = 0]) TPL +
      // (TPL - r0) are a per-thread base plus a constant.
      const cplx *Zc = reinterpret_cast<const cplx *>(lds + G::L_Z);
      const cplx *Tc = reinterpret_cast<const cplx *>(lds + G::L_T), *Ec = reinterpret_cast<const cplx *>(lds + G::L_E);
      auto        finish = [&](double (&v)[NLD]) {
        // (the 51 LDS addresses below depend on the thread only: computed outside the batch loop they would occupy 51
        // registers for the whole kernel -- with them the fused pass spilled 500 B per lane)
        int c_ = c, r0_ = r0;
        opaque(c_), opaque(r0_);
        const int c = c_, r0 = r0_;
        // k = r0 + i TPL with TPL = F Q, Q a power of two: k mod F = r0 mod F and k / F = r0 / F + i Q (disjoint bit
        // fields), so the slot of Z_k is a per-thread base plus a constant; the same for N - k = u TPL + rm
        constexpr int F = G::F, Q = TPL / F;
        const int     rm = r0 ? TPL - r0 : 0;
        const int     posk0 = c * N + (r0 % F) * G::M + G::pos_of_m(r0 / F), posm0 = c * N + (rm % F) * G::M + G::pos_of_m(rm / F);
        if (DCT_EXP & 4)
          {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
              v[i] = lds[G::L_Z + c * N + r0 + i];
            return;
          }
        const cplx e = Ec[c];
#pragma unroll
        for (int i = 0; i < NLD - 1; ++i)
          {
            const int k  = r0 + i * TPL;
            const int pk = posk0 + G::pos_of_m(i * Q);
            const int pm = posm0 + (r0 ? G::pos_of_m((15 - i) * Q) : G::pos_of_m(((16 - i) & 15) * Q));
            const cplx   zk = Zc[G::pad(pk)], zm = Zc[G::pad(pm)], w = Tc[k];
            const double are = 0.5 * (zk.re + zm.re), bre = 0.5 * (zk.im + zm.im), bim = -0.5 * (zk.re - zm.re);
            v[i] = 0.5 * (are + (w.re * bre - w.im * bim) + ((k & 1) ? e.im : e.re));
          }
        const cplx z0 = Zc[G::pad(c * N)];
        v[NLD - 1]    = 0.5 * (z0.re - z0.im + e.re); // y_N (used by r0 = 0 only)
      };
      double pre[NLD], res[NLD];
      long   b = blockIdx.x;
      if (b >= nbatch)
        return;
      Pos P = batch_pos(b), Pprev = P;
      load(P, pre);
      bool have_prev = false;
      while (true)
        {
          to_lds(pre);
          __syncthreads();
          if (have_prev)
            store(Pprev, res);
          const long bn = b + gridDim.x;
          const Pos  Pn = batch_pos(bn < nbatch ? bn : b);
          if (bn < nbatch)
            load(Pn, pre);
          dct_lines<N>(lds);
          if (FUSED)
            {
              // scale the mode coefficients and transform back; the factors that belong to the line (x, y) first
              cplx *F = reinterpret_cast<cplx *>(lds + G::L_F);
              if (t < LB)
                {
                  const unsigned l = (unsigned)(b * LB) + t, x = l % (unsigned)pitch, y = l / (unsigned)pitch;
                  const bool     live = t < P.nl && (int)x < A.nx;
                  F[t] = live ? cplx{A.lx[x] + A.ly[y], A.ax[x] * A.ay[y]} : cplx{0., 0.};
                }
              finish(res);
              __syncthreads();
              const cplx f = F[c]; // (an absent line: ax ay = 0)
              const bool second = A.cm2 != 0. || A.cl2 != 0.;
#pragma unroll
              for (int i = 0; i < NLD; ++i)
                {
                  const int    k = i < NLD - 1 ? r0 + i * TPL : N;
                  const double lam = f.re + A.lz[k], d = A.cm + A.cl * lam, d2 = A.cm2 + A.cl2 * lam;
                  const double az = A.az[k]; // (unconditional: a load behind the condition is a branch)
                  // (selects only: seventeen branches here put 350 B per lane into scratch)
                  const double r1 = fast_rcp(d), r2 = fast_rcp(d2);
                  const double sc = ((d > A.eps || d < -A.eps) ? r1 : 0.) + ((second && (d2 > A.eps2 || d2 < -A.eps2)) ? r2 : 0.);
                  res[i] *= f.im * az * sc;
                }
              to_lds(res);
              __syncthreads();
              dct_lines<N>(lds);
            }
          finish(res);
          __syncthreads();
          Pprev = P, have_prev = true;
          if (bn >= nbatch)
            break;
          b = bn, P = Pn;
        }
      store(Pprev, res);
    }
  } // namespace dct
} // namespace adaflo_hip
