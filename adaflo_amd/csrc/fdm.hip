// fdm.hip -- fast-diagonalisation inverses for the inner solves of the block preconditioner.
//
// The reference preconditions the inner solves of NavierStokesPreconditioner::vmult
// (source/navier_stokes_preconditioner.cc:595-737) with ILU / AMG of ASSEMBLED matrices
// (Trilinos); SURVEY 8f rank 3 asks for matrix-free replacements.  On a structured brick with
// constant coefficients the operators behind those matrices are sums of Kronecker products of 1D
// finite element matrices,
//     c_m  Mx (x) My (x) Mz  +  c_l (Kx (x) My (x) Mz + Mx (x) Ky (x) Mz + Mx (x) My (x) Kz),
// (pressure mass: c_l = 0; pressure Poisson: c_m = 0; velocity block: its mass + vector-Laplace
// part), and are inverted EXACTLY by the fast diagonalisation method (Lynch, Rice, Thomas 1964):
// with the generalised eigenpairs  K_d S_d = M_d S_d Lambda_d,  S_d^T M_d S_d = I  of the three 1D
// problems (Dirichlet nodes eliminated),
//     A^{-1} = (Sx (x) Sy (x) Sz)  diag(c_m + c_l (lx + ly + lz))^{-1}  (Sx (x) Sy (x) Sz)^T.
// One application = six dense 1D transforms: GEMMs of n x n matrices against the n^2 lines of the
// brick -- 12 n^4 flops per scalar field, the kind of dense FP64 work the chip is good at (tiled
// LDS kernel below; rocBLAS is deliberately not linked).  Singular cases (pure Neumann Poisson
// problem) use the pseudo-inverse: the null mode is dropped.
// Degree-1 spaces with natural ends and 2^m intervals per direction (the level-set grid and the Q1 pressure grid of
// the two-phase runs) do not need the matrices at all: their eigenvectors are cosines and the transforms are fast
// cosine transforms in LDS, five memory-bound passes per application (fdm_dct_kernel.hpp).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "kernels.hpp"

#include "fdm_dct_kernel.hpp" // (after the HIP runtime: the header includes nothing itself)

namespace adaflo_hip
{
  namespace
  {
    // ---- host: 1D finite element matrices and their generalised eigen-decomposition ----------
    struct Eig1D
    {
      int     n      = 0;
      double *d_St   = nullptr; // [mode][node]  (forward transform: modes = St . nodes)
      double *d_S    = nullptr; // [node][mode]  (backward transform)
      double *d_lam  = nullptr; // eigenvalue per mode; < 0 marks a padding mode (constrained node)
      // A 1D problem that is symmetric about its midpoint (same kind of end on both sides) has eigenvectors that are
      // even or odd about it.  The modes are then ordered [even | odd] and the transforms fold the nodes j and n-1-j
      // onto each other: two products of half the size instead of one, half the flops (fdm_apply).
      bool sym = false;
      int  n_even = 0; // number of even modes (they come first)
      // degree 1, natural ends, n - 1 = 2^m, 3 2^m or 5 2^m intervals: the eigenvectors are sqrt(a2[k]) cos(pi j k / (n - 1)) and the
      // transforms run as fast cosine transforms (fdm_dct_kernel.hpp); modes in natural order k = 0 .. n - 1 there
      int     nfft = 0;          // n - 1; 0: not available
      double *d_tw  = nullptr;   // [n][2]: exp(-i pi m / (n - 1))
      double *d_a2  = nullptr;   // squared normalisation of mode k
      double *d_lamn = nullptr;  // eigenvalue of mode k
    };

    // M, K of FE_Q(degree) on ncell cells of size h, quadrature QGauss(nq); dense n x n, row-major
    void assemble_1d(const int degree, const int ncell, const double h, const int nq, std::vector<double> &M,
                     std::vector<double> &K)
    {
      const Quadrature1D q  = gauss(nq);
      const Shape1D      sh = shape_fe_q(degree, q);
      const int          nd = degree + 1, n = degree * ncell + 1;
      M.assign((size_t)n * n, 0.);
      K.assign((size_t)n * n, 0.);
      for (int c = 0; c < ncell; ++c)
        for (int i = 0; i < nd; ++i)
          for (int j = 0; j < nd; ++j)
            {
              double m = 0., k = 0.;
              for (int p = 0; p < nq; ++p)
                {
                  m += q.w[p] * sh.S[p * nd + i] * sh.S[p * nd + j];
                  k += q.w[p] * sh.D[p * nd + i] * sh.D[p * nd + j];
                }
              M[(size_t)(c * degree + i) * n + c * degree + j] += h * m;
              K[(size_t)(c * degree + i) * n + c * degree + j] += k / h;
            }
    }

    // symmetric eigenproblem by cyclic Jacobi rotations: A (n x n, destroyed) = Q diag(w) Q^T
    void jacobi_eig(const int n, std::vector<double> &A, std::vector<double> &Q, std::vector<double> &w)
    {
      Q.assign((size_t)n * n, 0.);
      for (int i = 0; i < n; ++i)
        Q[(size_t)i * n + i] = 1.;
      double norm = 0.;
      for (double v : A)
        norm += v * v;
      norm = std::sqrt(norm);
      for (int sweep = 0; sweep < 60; ++sweep)
        {
          double off = 0.;
          for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q)
              off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
          if (std::sqrt(2. * off) <= 1e-15 * norm)
            break;
          for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q)
              {
                const double apq = A[(size_t)p * n + q];
                if (std::abs(apq) <= 1e-300)
                  continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                const double theta = (aqq - app) / (2. * apq);
                const double t     = (theta >= 0. ? 1. : -1.) / (std::abs(theta) + std::sqrt(theta * theta + 1.));
                const double c = 1. / std::sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < n; ++k) // columns p, q
                  {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                  }
                for (int k = 0; k < n; ++k) // rows p, q
                  {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                  }
                for (int k = 0; k < n; ++k)
                  {
                    const double qkp = Q[(size_t)k * n + p], qkq = Q[(size_t)k * n + q];
                    Q[(size_t)k * n + p] = c * qkp - s * qkq;
                    Q[(size_t)k * n + q] = s * qkp + c * qkq;
                  }
              }
        }
      w.resize(n);
      for (int i = 0; i < n; ++i)
        w[i] = A[(size_t)i * n + i];
    }

    // K v = lambda M v on the free nodes; S (n x n, [node][mode]) with S^T M S = I on the free
    // block, zero rows at constrained nodes, padding modes (lambda = -1) beyond the free count
    void generalized_eig(const int n, const std::vector<double> &K, const std::vector<double> &M,
                         const std::vector<char> &constrained, std::vector<double> &S, std::vector<double> &lam)
    {
      std::vector<int> fr;
      for (int i = 0; i < n; ++i)
        if (!constrained[i])
          fr.push_back(i);
      const int m = (int)fr.size();
      // Cholesky M_ff = L L^T
      std::vector<double> L((size_t)m * m, 0.);
      for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j)
          {
            double s = M[(size_t)fr[i] * n + fr[j]];
            for (int k = 0; k < j; ++k)
              s -= L[(size_t)i * m + k] * L[(size_t)j * m + k];
            L[(size_t)i * m + j] = i == j ? std::sqrt(s) : s / L[(size_t)j * m + j];
          }
      // C = L^-1 K_ff L^-T
      std::vector<double> B((size_t)m * m), C((size_t)m * m);
      for (int col = 0; col < m; ++col) // B = L^-1 K_ff
        for (int i = 0; i < m; ++i)
          {
            double s = K[(size_t)fr[i] * n + fr[col]];
            for (int k = 0; k < i; ++k)
              s -= L[(size_t)i * m + k] * B[(size_t)k * m + col];
            B[(size_t)i * m + col] = s / L[(size_t)i * m + i];
          }
      for (int row = 0; row < m; ++row) // C = B L^-T:  C[row][:] solves L C[row][:]^T = B[row][:]^T
        for (int i = 0; i < m; ++i)
          {
            double s = B[(size_t)row * m + i];
            for (int k = 0; k < i; ++k)
              s -= L[(size_t)i * m + k] * C[(size_t)row * m + k];
            C[(size_t)row * m + i] = s / L[(size_t)i * m + i];
          }
      for (int i = 0; i < m; ++i) // symmetrise the round-off
        for (int j = i + 1; j < m; ++j)
          C[(size_t)i * m + j] = C[(size_t)j * m + i] = 0.5 * (C[(size_t)i * m + j] + C[(size_t)j * m + i]);
      std::vector<double> Q, w;
      jacobi_eig(m, C, Q, w);
      // S_ff = L^-T Q
      S.assign((size_t)n * n, 0.);
      lam.assign(n, -1.);
      for (int col = 0; col < m; ++col)
        {
          std::vector<double> x(m);
          for (int i = m - 1; i >= 0; --i)
            {
              double s = Q[(size_t)i * m + col];
              for (int k = i + 1; k < m; ++k)
                s -= L[(size_t)k * m + i] * x[k];
              x[i] = s / L[(size_t)i * m + i];
            }
          for (int i = 0; i < m; ++i)
            S[(size_t)fr[i] * n + col] = x[i];
          lam[col] = std::max(w[col], 0.);
        }
    }

    // Degree 1 on a uniform grid: M = h/6 tridiag(1, 4, 1), K = 1/h tridiag(-1, 2, -1) (half rows at a
    // natural end) have the cosine / sine vectors as common eigenvectors for all four combinations of
    // Dirichlet and natural ends,
    //   natural - natural    v_k(j) = cos(k pi j / N),        k = 0 .. N
    //   Dirichlet - Dirichlet v_k(j) = sin(k pi j / N),        k = 1 .. N - 1
    //   Dirichlet - natural  v_k(j) = sin((k + 1/2) pi j / N), k = 0 .. N - 1
    //   natural - Dirichlet  v_k(j) = cos((k + 1/2) pi j / N), k = 0 .. N - 1
    // with K v = lambda M v, lambda = 6 (1 - cos t) / (h^2 (2 + cos t)), t the frequency times pi / N.
    // The level-set space has up to 513 nodes per direction, where the cyclic Jacobi solver above would take
    // seconds.  Same output convention as generalized_eig.
    void linear_eig(const int n, const double h, const bool lo, const bool hi, std::vector<double> &S, std::vector<double> &lam,
                    std::vector<double> *norm2 = nullptr)
    {
      if (norm2)
        norm2->assign(n, 0.);
      const int    N  = n - 1;
      const double pi = 3.14159265358979323846;
      S.assign((size_t)n * n, 0.);
      lam.assign(n, -1.);
      const int j0 = lo ? 1 : 0, j1 = hi ? N - 1 : N, m = j1 - j0 + 1;
      for (int col = 0; col < m; ++col)
        {
          const double f = (lo == hi) ? (double)(col + (lo ? 1 : 0)) : col + 0.5; // frequency
          const double t = f * pi / N;
          std::vector<double> v(n, 0.);
          for (int j = j0; j <= j1; ++j)
            v[j] = lo ? std::sin(t * j) : std::cos(t * j);
          // v^T M v with the free block of M
          double nrm = 0.;
          for (int j = j0; j <= j1; ++j)
            {
              const double diag = ((j > 0 ? 1. : 0.) + (j < N ? 1. : 0.)) * h / 3.;
              double       mv   = diag * v[j];
              if (j > j0)
                mv += h / 6. * v[j - 1];
              if (j < j1)
                mv += h / 6. * v[j + 1];
              nrm += v[j] * mv;
            }
          const double sc = 1. / std::sqrt(nrm);
          if (norm2)
            (*norm2)[col] = 1. / nrm;
          for (int j = j0; j <= j1; ++j)
            S[(size_t)j * n + col] = v[j] * sc;
          const double c = std::cos(t);
          lam[col]       = std::max(6. * (1. - c) / (h * h * (2. + c)), 0.);
        }
    }

    // ---- device: strided batched f64 GEMM  C[b] = A[b?] . B[b] on the matrix cores ----------------
    // C[i][j] = sum_k A[i * rsA + k * csA] * B[k * rsB + j * csB].  A is the small 1D matrix (shared by all
    // batches) or the field (x transform), B the other one.  64 x 64 tiles of C per workgroup, K-steps of 16
    // staged in LDS; each of the four waves owns a 32 x 32 quadrant = 2 x 2 tiles of v_mfma_f64_16x16x4_f64
    // (operands: lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15], result register r of
    // lane l is C[(l >> 4) + 4 r][l & 15]; cdna_hip_programming.md, fragment layout).  One LDS read feeds
    // 16 FMAs per lane -- the register-tiled vector version (4 x 4 outputs per thread, one read per two
    // FMAs) ran at 11 TFLOP/s, bound by the LDS instruction rate.
    // Tile shapes: a wave owns TMS x TNS MFMA tiles, WM x WN waves form the workgroup tile
    // (16 TMS WM) x (16 TNS WN); 64 x 64, 48 x 64, 64 x 48 and 48 x 48 are instantiated and the launcher takes
    // 48 for a dimension that it pads less (129 = 2 * 64 + 1 nodes would waste a third of a 64-wide tiling).
#ifndef GEMM_BATCH
#define GEMM_BATCH 2
#endif
    constexpr int GK = 8, GLD = 64 + 16; // row stride 80 doubles: the two k-rows of a half-wave hit disjoint banks
    struct GemmArgs
    {
      int           M, N, K;
      long          rsA, csA, bsA, rsB, csB, bsB, rsC, csC, bsC;
      const double *A, *B;
      double       *C;
      // folded transforms: the operand that holds the field is read as X[k] + fold X[nfold-1-k] along the contraction
      // index (fold = +1 / -1; the midpoint of an odd length counts once); the result is stored plainly (mirror = 0),
      // to an index and its mirror image nmirror-1-index (1), or added there / subtracted at the mirror image (2)
      int fold = 0, fold_on_A = 0, nfold = 0, mirror = 0, mirror_on_i = 0, nmirror = 0;
      // scaling of the mode coefficients fused into the forward z transform (rows i = z modes, columns j = x + nx y, batch =
      // stacked component): C *= 1 / (c_m + c_l (lx[x] + ly[y] + lz[i0 + i])), padding modes and the null mode give 0
      const double *lx = nullptr, *ly = nullptr, *lz = nullptr;
      int           scale = 0, scale_nx = 0, scale_i0 = 0;
      double        scale_cm = 0., scale_cl = 0., scale_eps = 0.;
      double        scale_cm2 = 0., scale_cl2 = 0.; // a second inverse of the same modes, added (both 0: none)
      double        scale_eps2 = 0.;                // ... with the null-mode threshold of ITS scale
    };
    typedef double d4_t __attribute__((ext_vector_type(4)));
    // FM: 0 plain, 1 / 2 the field operand A / B is folded on load, 3 / 4 mirrored stores along j / i (GemmArgs)
    template <int TMS, int TNS, int WM, int WN, int FM>
    __global__ __launch_bounds__(64 * WM *WN) void fdm_gemm_kernel(const GemmArgs g)
    {
      constexpr int NT = 64 * WM * WN, TM = 16 * TMS * WM, TN = 16 * TNS * WN;
      __shared__ double As[GK][GLD], Bs[GK][GLD];
      const int     tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
      const int     i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
      const int     wi = 16 * TMS * (wave / WN), wj = 16 * TNS * (wave % WN); // sub-tile of this wave
      const int     fl = lane & 15, fk = lane >> 4;                          // fragment coordinates of this lane
      const double *A = g.A + (long)blockIdx.z * g.bsA, *B = g.B + (long)blockIdx.z * g.bsB;
      double       *C = g.C + (long)blockIdx.z * g.bsC;
      d4_t          acc[TMS][TNS];
#pragma unroll
      for (int r = 0; r < TMS; ++r)
#pragma unroll
        for (int c = 0; c < TNS; ++c)
          acc[r][c] = d4_t{0., 0., 0., 0.};
      // loader mapping: the faster-varying global index gets consecutive threads.  The loads of TWO k-tiles are
      // issued together into two register sets before the first is used: a workgroup walks few k-tiles with
      // little work each, so the load latency is paid once per pair; more tiles per batch cost occupancy and
      // were slower (3, 4, 5, 9 tiles), as was a rolling two-tile pipeline.  129^3 nodes, same box: 50.4 us per
      // GEMM with the register-tiled vector kernel, 40.0 us with this one.  Round 3: k-tiles of 8 instead of 16
      // (less LDS and fewer staging registers per workgroup, more workgroups in flight) are 5-8 % faster on every
      // transform size (32: 30 % slower; 4: as 8; 3 or 4 tiles of 8 in flight: as 2; one wave per 48 x 48 tile with nine
      // accumulators and no workgroup barrier: 14 % slower).
      const bool    a_k_fast = g.csA == 1, b_j_fast = g.csB == 1;
      constexpr bool fold_a = FM == 1, fold_b = FM == 2 || FM == 6, scaled = FM == 5 || FM == 6;
      const double  fold_sign = g.fold;
      const int     fold_mid = (g.fold > 0 && (g.nfold & 1)) ? (g.nfold - 1) / 2 : -1;
      constexpr int LA = (TM * GK + NT - 1) / NT, LB = (TN * GK + NT - 1) / NT;
      auto fetch = [&](const int k0, double (&ra)[LA], double (&rb)[LB]) {
#pragma unroll
        for (int u = 0; u < LA; ++u)
          {
            const int e = tid + u * NT;
            const int ii = a_k_fast ? e / GK : e % TM, kk = a_k_fast ? e % GK : e / TM;
            const int i = i0 + ii, k = k0 + kk;
            double    v = (e < TM * GK && i < g.M && k < g.K) ? A[i * g.rsA + k * g.csA] : 0.;
            if (fold_a && e < TM * GK && i < g.M && k < g.K && k != fold_mid)
              v += fold_sign * A[i * g.rsA + (g.nfold - 1 - k) * g.csA];
            ra[u] = v;
          }
#pragma unroll
        for (int u = 0; u < LB; ++u)
          {
            const int e = tid + u * NT;
            const int jj = b_j_fast ? e % TN : e / GK, kk = b_j_fast ? e / TN : e % GK;
            const int j = j0 + jj, k = k0 + kk;
            double    v = (e < TN * GK && j < g.N && k < g.K) ? B[k * g.rsB + j * g.csB] : 0.;
            if (fold_b && e < TN * GK && j < g.N && k < g.K && k != fold_mid)
              v += fold_sign * B[(g.nfold - 1 - k) * g.rsB + j * g.csB];
            rb[u] = v;
          }
      };
      auto commit = [&](const double (&ra)[LA], const double (&rb)[LB]) {
#pragma unroll
        for (int u = 0; u < LA; ++u)
          {
            const int e = tid + u * NT;
            if (e < TM * GK)
              As[a_k_fast ? e % GK : e / TM][a_k_fast ? e / GK : e % TM] = ra[u];
          }
#pragma unroll
        for (int u = 0; u < LB; ++u)
          {
            const int e = tid + u * NT;
            if (e < TN * GK)
              Bs[b_j_fast ? e / TN : e % GK][b_j_fast ? e % TN : e / GK] = rb[u];
          }
      };
      auto multiply = [&]() {
#pragma unroll
        for (int ks = 0; ks < GK; ks += 4)
          {
            double a[TMS], b[TNS];
#pragma unroll
            for (int r = 0; r < TMS; ++r)
              a[r] = As[ks + fk][wi + 16 * r + fl];
#pragma unroll
            for (int c = 0; c < TNS; ++c)
              b[c] = Bs[ks + fk][wj + 16 * c + fl];
#pragma unroll
            for (int r = 0; r < TMS; ++r)
#pragma unroll
              for (int c = 0; c < TNS; ++c)
                acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], b[c], acc[r][c], 0, 0, 0);
          }
      };
      constexpr int NB = GEMM_BATCH; // k-tiles whose loads are issued together
      double        ra[NB][LA], rb[NB][LB];
      for (int k0 = 0; k0 < g.K; k0 += NB * GK)
        {
#pragma unroll
          for (int t = 0; t < NB; ++t)
            if (k0 + t * GK < g.K)
              fetch(k0 + t * GK, ra[t], rb[t]);
#pragma unroll
          for (int t = 0; t < NB; ++t)
            if (k0 + t * GK < g.K)
              {
                commit(ra[t], rb[t]);
                __syncthreads();
                multiply();
                __syncthreads();
              }
        }
#pragma unroll
      for (int r = 0; r < TMS; ++r)
#pragma unroll
        for (int c = 0; c < TNS; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            {
              const int i = i0 + wi + 16 * r + fk + 4 * q, j = j0 + wj + 16 * c + fl;
              if (i < g.M && j < g.N)
                {
                  const long idx = i * g.rsC + j * g.csC;
                  if (scaled)
                    {
                      const double a = g.lx[j % g.scale_nx], b = g.ly[j / g.scale_nx], l = g.lz[g.scale_i0 + i];
                      const double d = g.scale_cm + g.scale_cl * (a + b + l), d2 = g.scale_cm2 + g.scale_cl2 * (a + b + l);
                      const bool   pad = a < 0. || b < 0. || l < 0.;
                      double       v = (pad || std::abs(d) <= g.scale_eps) ? 0. : acc[r][c][q] / d;
                      if (g.scale_cm2 != 0. || g.scale_cl2 != 0.)
                        v += (pad || std::abs(d2) <= g.scale_eps2) ? 0. : acc[r][c][q] / d2;
                      C[idx] = v;
                    }
                  else if (FM < 3)
                    C[idx] = acc[r][c][q];
                  else
                    {
                      const int  o = FM == 4 ? i : j, om = g.nmirror - 1 - o;
                      const long idm = FM == 4 ? om * g.rsC + j * g.csC : i * g.rsC + om * g.csC;
                      if (g.mirror == 1)
                        {
                          C[idx] = acc[r][c][q];
                          if (om != o)
                            C[idm] = acc[r][c][q];
                        }
                      else
                        {
                          C[idx] += acc[r][c][q];
                          C[idm] -= acc[r][c][q];
                        }
                    }
                }
            }
    }

    int gemm(adaflo_ctx *ctx, const GemmArgs &g, const int batch)
    {
      // 48 or 64 per direction: whichever pads the dimension less (ties: 64)
      auto tile = [](const int n) { return (n + 47) / 48 * 48 < (n + 63) / 64 * 64 ? 48 : 64; };
      const int  tm = tile(g.M), tn = tile(g.N);
      const dim3 grid((g.N + tn - 1) / tn, (g.M + tm - 1) / tm, batch);
      const int fm = g.scale ? (g.fold != 0 ? 6 : 5) : (g.fold != 0 ? (g.fold_on_A ? 1 : 2) : (g.mirror != 0 ? (g.mirror_on_i ? 4 : 3) : 0));
#define FDM_GEMM(FM)                                                                                                  \
  {                                                                                                                   \
    if (tm == 64 && tn == 64)                                                                                         \
      hipLaunchKernelGGL((fdm_gemm_kernel<2, 2, 2, 2, FM>), grid, dim3(256), 0, ctx->stream, g);                       \
    else if (tm == 48 && tn == 64)                                                                                    \
      hipLaunchKernelGGL((fdm_gemm_kernel<3, 1, 1, 4, FM>), grid, dim3(256), 0, ctx->stream, g);                       \
    else if (tm == 64 && tn == 48)                                                                                    \
      hipLaunchKernelGGL((fdm_gemm_kernel<1, 3, 4, 1, FM>), grid, dim3(256), 0, ctx->stream, g);                       \
    else                                                                                                              \
      hipLaunchKernelGGL((fdm_gemm_kernel<3, 1, 1, 3, FM>), grid, dim3(192), 0, ctx->stream, g);                       \
  }
      switch (fm)
        {
          case 0:
            FDM_GEMM(0) break;
          case 1:
            FDM_GEMM(1) break;
          case 2:
            FDM_GEMM(2) break;
          case 3:
            FDM_GEMM(3) break;
          case 4:
            FDM_GEMM(4) break;
          case 5:
            FDM_GEMM(5) break;
          default:
            FDM_GEMM(6) break;
        }
#undef FDM_GEMM
      return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
    }


    // scalar field <- component c of an interleaved vector, and back (constrained rows: dst = src)
    __global__ __launch_bounds__(256) void fdm_take_kernel(double *__restrict__ w, const double *__restrict__ v,
                                                           const long n, const int ncomp, const int comp)
    {
      for (long t = blockIdx.x * 256L + threadIdx.x; t < n; t += (long)gridDim.x * 256)
        w[t] = v[t * ncomp + comp];
    }
    __global__ __launch_bounds__(256) void fdm_put_kernel(double *__restrict__ dst, const double *__restrict__ w,
                                                          const double *__restrict__ src, const long n, const int ncomp,
                                                          const int comp, const int nx, const int ny, const int nz,
                                                          const uint32_t mask, const int stride)
    {
      for (long t = blockIdx.x * 256L + threadIdx.x; t < n; t += (long)gridDim.x * 256)
        {
          const int  I = t % nx, J = (t / nx) % ny, K = t / ((long)nx * ny);
          const bool con = mask != 0u && on_constrained_face(I, J, K, nx, ny, nz, mask, stride, comp);
          dst[t * ncomp + comp] = con ? src[t * ncomp + comp] : w[t];
        }
    }

    // all components at once: w [c][node] <-> v [node][c]
    __global__ __launch_bounds__(256) void fdm_take_all_kernel(double *__restrict__ w, const double *__restrict__ v,
                                                               const long n, const int ncomp)
    {
      for (long t = blockIdx.x * 256L + threadIdx.x; t < n * ncomp; t += (long)gridDim.x * 256)
        w[(t % ncomp) * n + t / ncomp] = v[t];
    }
    __global__ __launch_bounds__(256) void fdm_put_all_kernel(double *__restrict__ dst, const double *__restrict__ w,
                                                              const double *__restrict__ src, const long n, const int ncomp,
                                                              const int nx, const int ny, const int nz, const uint32_t mask,
                                                              const int stride)
    {
      for (long t = blockIdx.x * 256L + threadIdx.x; t < n * ncomp; t += (long)gridDim.x * 256)
        {
          const long node = t / ncomp;
          const int  c = (int)(t % ncomp), I = node % nx, J = (node / nx) % ny, K = node / ((long)nx * ny);
          const bool con = mask != 0u && on_constrained_face(I, J, K, nx, ny, nz, mask, stride, c);
          dst[t]         = con ? src[t] : w[c * n + node];
        }
    }

    // ---- fast cosine transforms (fdm_dct_kernel.hpp) -----------------------------------------------------------
    template <int N, bool FUSED, int AXIS>
    __global__ __launch_bounds__(dct::NT, 2) void fdm_dct_kernel(const dct::DctArgs A) // (two workgroups per CU: 256 registers)
    {
      extern __shared__ double dct_lds[];
      dct::dct_body<N, FUSED, AXIS>(A, dct_lds);
    }
    template <int N, bool FUSED, int AXIS>
    int launch_dct_t(adaflo_ctx *ctx, const dct::DctArgs &A)
    {
      using G            = dct::Geo<N>;
      const size_t lds   = sizeof(double) * G::L_TOTAL;
      static bool  attr_set = false;
      if (!attr_set)
        {
          if (hipFuncSetAttribute(reinterpret_cast<const void *>(&fdm_dct_kernel<N, FUSED, AXIS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ADAFLO_EHIP;
          attr_set = true;
        }
      // persistent workgroups (grid-stride loop over the batches of LB lines), as many as are resident at once
      static int resident = 0;
      if (resident == 0)
        {
          int dev = 0, cus = 0;
          if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return ADAFLO_EHIP;
          resident = std::max(1, cus) * std::max(1, (int)((160 * 1024) / lds));
        }
      const long nb = std::min<long>((A.n_lines + G::LB - 1) / G::LB, resident);
      hipLaunchKernelGGL((fdm_dct_kernel<N, FUSED, AXIS>), dim3((unsigned)nb), dim3(dct::NT), lds, ctx->stream, A);
      return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
    }
    template <int N>
    int launch_dct_n(adaflo_ctx *ctx, const bool fused, const dct::DctArgs &A)
    {
      if (fused)
        return launch_dct_t<N, true, 2>(ctx, A);
      if (A.axis == 0)
        return launch_dct_t<N, false, 0>(ctx, A);
      return A.axis == 1 ? launch_dct_t<N, false, 1>(ctx, A) : launch_dct_t<N, false, 2>(ctx, A);
    }
    int launch_dct(adaflo_ctx *ctx, const int nfft, const bool fused, const dct::DctArgs &A)
    {
      switch (nfft)
        {
#define DCT_CASE(N_) \
  case N_:           \
    return launch_dct_n<N_>(ctx, fused, A);
          DCT_CASE(64)
          DCT_CASE(128)
          DCT_CASE(256)
          DCT_CASE(512)
          DCT_CASE(1024)
          DCT_CASE(80)
          DCT_CASE(160)
          DCT_CASE(320)
          DCT_CASE(640)
          DCT_CASE(96)
          DCT_CASE(192)
          DCT_CASE(384)
          DCT_CASE(768)
#undef DCT_CASE
        }
      return ADAFLO_EINVAL;
    }

    struct FieldFdm
    {
      int   degree = 0, ncomp = 0, nn[3] = {0, 0, 0};
      Eig1D e[3][3]; // [component][direction]
    };
    struct Fdm
    {
      FieldFdm                        field[3];
      bool                            ready[3] = {false, false, false};
      std::map<std::vector<long>, Eig1D> cache;
      double                         *w0 = nullptr, *w1 = nullptr;
      size_t                          wcount = 0;
    };

    // modes even / odd about the midpoint first / last (padding modes, whose columns never matter -- the scaling
    // sets their coefficient to zero --, are zeroed and fill up the even block); false: the problem is not symmetric
    bool order_modes_by_symmetry(const int n, std::vector<double> &S, std::vector<double> &lam, int &n_even)
    {
      std::vector<int> kind(n); // +1 even, -1 odd, 0 padding
      for (int m = 0; m < n; ++m)
        {
          if (lam[m] < 0.)
            {
              kind[m] = 0;
              continue;
            }
          double p = 0., q = 0.;
          for (int j = 0; j < n; ++j)
            {
              p += S[(size_t)j * n + m] * S[(size_t)(n - 1 - j) * n + m];
              q += S[(size_t)j * n + m] * S[(size_t)j * n + m];
            }
          if (q <= 0. || std::abs(std::abs(p / q) - 1.) > 1e-9)
            return false;
          kind[m] = p > 0. ? 1 : -1;
        }
      std::vector<int> order;
      for (int m = 0; m < n; ++m)
        if (kind[m] == 1)
          order.push_back(m);
      for (int m = 0; m < n; ++m) // padding modes behind the even ones, up to half of the nodes
        if (kind[m] == 0 && (int)order.size() < (n + 1) / 2)
          {
            order.push_back(m);
            kind[m] = 2;
          }
      n_even = (int)order.size();
      for (int m = 0; m < n; ++m)
        if (kind[m] == -1 || kind[m] == 0)
          order.push_back(m);
      std::vector<double> S2((size_t)n * n), lam2(n);
      for (int c = 0; c < n; ++c)
        {
          const int m = order[c];
          lam2[c]     = lam[m];
          for (int j = 0; j < n; ++j)
            S2[(size_t)j * n + c] = lam[m] < 0. ? 0. : S[(size_t)j * n + m];
        }
      S.swap(S2);
      lam.swap(lam2);
      return true;
    }

    int upload_eig(Eig1D &E, const int n, std::vector<double> S, std::vector<double> lam)
    {
      E.sym = order_modes_by_symmetry(n, S, lam, E.n_even);
      std::vector<double> St((size_t)n * n);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
          St[(size_t)j * n + i] = S[(size_t)i * n + j];
      E.n = n;
      if (hipMalloc(&E.d_S, sizeof(double) * n * n) != hipSuccess || hipMalloc(&E.d_St, sizeof(double) * n * n) != hipSuccess ||
          hipMalloc(&E.d_lam, sizeof(double) * n) != hipSuccess)
        return ADAFLO_ENOMEM;
      if (copy_to_device_now(E.d_S, S.data(), sizeof(double) * n * n) != hipSuccess ||
          copy_to_device_now(E.d_St, St.data(), sizeof(double) * n * n) != hipSuccess ||
          copy_to_device_now(E.d_lam, lam.data(), sizeof(double) * n) != hipSuccess)
        return ADAFLO_EHIP;
      return 0;
    }
  } // namespace

  void fdm_destroy(adaflo_ctx *ctx)
  {
    Fdm *F = static_cast<Fdm *>(ctx->fdm);
    if (!F)
      return;
    for (auto &kv : F->cache)
      for (double *p : {kv.second.d_S, kv.second.d_St, kv.second.d_lam, kv.second.d_tw, kv.second.d_a2, kv.second.d_lamn})
        if (p)
          (void)hipFree(p);
    for (double *p : {F->w0, F->w1})
      if (p)
        (void)hipFree(p);
    delete F;
    ctx->fdm = nullptr;
  }

  // field 0: velocity space FE_Q(k)^3 with quad_index_u, field 1: pressure space FE_Q(k-1) with
  // quad_index_p, field 2: level-set space FE_Q_iso_Q1(s) = FE_Q(1) on the s-times refined grid with the
  // 2-point rule; Dirichlet nodes as in the brick's constraint masks.  Fields are set up on first use.
  static int fdm_setup_field(adaflo_ctx *ctx, const int f)
  {
    if (!ctx->fdm)
      ctx->fdm = new Fdm;
    Fdm *F = static_cast<Fdm *>(ctx->fdm);
    if (F->ready[f])
      return 0;
    if (f == 2 && ctx->s <= 0)
      return ADAFLO_ENOTINIT;
    FieldFdm &fd = F->field[f];
    fd.degree    = f == 0 ? ctx->k : (f == 1 ? ctx->k - 1 : 1);
    fd.ncomp     = f == 0 ? 3 : 1;
    const int      sub  = f == 2 ? ctx->s : 1;
    const int      nq   = f == 0 ? ctx->k + 1 : (f == 1 ? ctx->k : 2);
    const uint32_t mask = f == 0 ? ctx->brick.con_u : (f == 1 ? ctx->brick.con_p : ctx->brick.con_ls);
    for (int d = 0; d < 3; ++d)
      fd.nn[d] = fd.degree * sub * ctx->desc.ncell[d] + 1;
    for (int c = 0; c < fd.ncomp; ++c)
      for (int d = 0; d < 3; ++d)
        {
          const int    stride = fd.ncomp == 1 ? 1 : 3, ncell = sub * ctx->desc.ncell[d];
          const double h      = ctx->desc.h[d] / sub;
          const bool   lo = mask >> (stride * (2 * d) + c) & 1u, hi = mask >> (stride * (2 * d + 1) + c) & 1u;
          const std::vector<long> key = {fd.degree, ncell, (long)std::llround(h * 1e15), nq, lo, hi};
          auto it = F->cache.find(key);
          if (it == F->cache.end())
            {
              const int           n = fd.nn[d];
              std::vector<double> S, lam, norm2;
              if (fd.degree == 1)
                linear_eig(n, h, lo, hi, S, lam, &norm2);
              else
                {
                  std::vector<double> M, K;
                  assemble_1d(fd.degree, ncell, h, nq, M, K);
                  std::vector<char> con(n, 0);
                  con[0]     = lo;
                  con[n - 1] = hi;
                  generalized_eig(n, K, M, con, S, lam);
                }
              Eig1D E;
              if (fd.degree == 1 && !lo && !hi && dct::dct_length_supported(n - 1))
                {
                  // natural order of the modes here (upload_eig reorders its copy by symmetry)
                  std::vector<double> tw(2 * (size_t)n);
                  for (int m = 0; m < n; ++m)
                    {
                      tw[2 * m]     = std::cos(3.14159265358979323846 * m / (n - 1));
                      tw[2 * m + 1] = -std::sin(3.14159265358979323846 * m / (n - 1));
                    }
                  if (hipMalloc(&E.d_tw, sizeof(double) * 2 * n) != hipSuccess || hipMalloc(&E.d_a2, sizeof(double) * n) != hipSuccess ||
                      hipMalloc(&E.d_lamn, sizeof(double) * n) != hipSuccess)
                    return ADAFLO_ENOMEM;
                  if (copy_to_device_now(E.d_tw, tw.data(), sizeof(double) * 2 * n) != hipSuccess ||
                      copy_to_device_now(E.d_a2, norm2.data(), sizeof(double) * n) != hipSuccess ||
                      copy_to_device_now(E.d_lamn, lam.data(), sizeof(double) * n) != hipSuccess)
                    return ADAFLO_EHIP;
                  E.nfft = n - 1;
                }
              if (int e = upload_eig(E, n, S, lam))
                return e;
              it = F->cache.emplace(key, E).first;
            }
          fd.e[c][d] = it->second;
        }
    // (all components at once, fdm_apply; rows padded to 16 doubles for the cosine transforms)
    const size_t need = (size_t)((fd.nn[0] + 15) / 16 * 16) * fd.nn[1] * fd.nn[2] * fd.ncomp;
    if (need > F->wcount)
      {
        for (double *p : {F->w0, F->w1})
          if (p)
            (void)hipFree(p);
        F->w0 = F->w1 = nullptr;
        F->wcount     = 0;
        if (hipMalloc(&F->w0, need * sizeof(double)) != hipSuccess || hipMalloc(&F->w1, need * sizeof(double)) != hipSuccess)
          return ADAFLO_ENOMEM;
        F->wcount = need;
      }
    F->ready[f] = true;
    return 0;
  }

  int fdm_setup(adaflo_ctx *ctx)
  {
    if (int e = fdm_setup_field(ctx, 0))
      return e;
    return fdm_setup_field(ctx, 1);
  }

  // One 1D transform of the field in [z][y][x] along `axis`: nodes -> modes with St (forward) or modes -> nodes with S
  // (backward), as strided (batched) GEMM.  For a symmetric 1D problem with at least FOLD_MIN nodes (the level-set grids) the transform is
  // folded (Eig1D): forward, the even modes see w_j + w_{n-1-j} and the odd ones w_j - w_{n-1-j} for j in the lower
  // half only; backward, the even and the odd sums u_e, u_o of the lower half give w_j = u_e + u_o and
  // w_{n-1-j} = u_e - u_o.  Two launches of a quarter of the flops each.
  struct ModeScaling // w *= 1 / (c_m + c_l (lx + ly + lz)), padding modes and the null mode give 0: epilogue of the forward z transform
  {
    const double *lx, *ly, *lz;
    double        cm, cl, eps;
    double        cm2 = 0., cl2 = 0.; // a second inverse applied to the same source and added
    double        eps2 = 0.;          // null-mode threshold of the second operator (its own scale)
  };
  static int transform_axis(adaflo_ctx *ctx, const int axis, const bool backward, const Eig1D &E, const double *in, double *out,
                            const int nx, const int ny, const int nz, const int nstack = 1, const ModeScaling *sc = nullptr)
  {
    // (directions of 129 nodes: a loss on the 65 x 65 x 129 pressure grid, where a launch is 10-16 us whatever it computes)
    const int FOLD_MIN = (long)nx * ny * nz * nstack >= 4000000 ? 96 : 192;
    const int     n = axis == 0 ? nx : (axis == 1 ? ny : nz);
    const double *T = backward ? E.d_S : E.d_St; // [out index][contraction index], row-major n x n
    GemmArgs      g{};
    int           batch = 1;
    // (nstack fields of the same shape one behind the other: more rows in x, more planes in y, a batch in z)
    if (axis == 0) // C[r][i] = sum_k W[r][k] T[i][k]
      {
        g.M = ny * nz * nstack, g.N = nx, g.K = nx;
        g.A = in, g.rsA = nx, g.csA = 1;
        g.B = T, g.rsB = 1, g.csB = nx;
        g.C = out, g.rsC = nx, g.csC = 1;
      }
    else if (axis == 1) // per z-plane C[i][j] = sum_k T[i][k] W[k][j]
      {
        g.M = ny, g.N = nx, g.K = ny;
        g.A = T, g.rsA = ny, g.csA = 1, g.bsA = 0;
        g.B = in, g.rsB = nx, g.csB = 1, g.bsB = (long)nx * ny;
        g.C = out, g.rsC = nx, g.csC = 1, g.bsC = (long)nx * ny;
        batch = nz * nstack;
      }
    else // C[i][j] = sum_k T[i][k] W[k][j], j over the plane
      {
        g.M = nz, g.N = nx * ny, g.K = nz;
        g.A = T, g.rsA = nz, g.csA = 1;
        g.B = in, g.rsB = (long)nx * ny, g.csB = 1, g.bsB = (long)nx * ny * nz;
        g.C = out, g.rsC = (long)nx * ny, g.csC = 1, g.bsC = (long)nx * ny * nz;
        batch = nstack;
        if (sc && !backward)
          {
            g.scale = 1, g.scale_nx = nx, g.scale_i0 = 0;
            g.lx = sc->lx, g.ly = sc->ly, g.lz = sc->lz;
            g.scale_cm = sc->cm, g.scale_cl = sc->cl, g.scale_eps = sc->eps;
            g.scale_cm2 = sc->cm2, g.scale_cl2 = sc->cl2, g.scale_eps2 = sc->eps2;
          }
      }
    static const bool no_fold = getenv("ADAFLO_FDM_NO_FOLD") != nullptr; // (tests / timing of the plain transforms)
    if (!E.sym || n < FOLD_MIN || no_fold)
      return gemm(ctx, g, batch);
    const int  nE = E.n_even, nO = n - nE, lo_e = (n + 1) / 2, lo_o = n / 2; // modes; nodes of the lower half (with / without midpoint)
    const long sw = axis == 0 ? 1 : (axis == 1 ? nx : (long)nx * ny);         // stride of the field along the axis
    for (int part = 0; part < 2; ++part) // even, odd
      {
        GemmArgs h = g;
        const int n_out = backward ? (part == 0 ? lo_e : lo_o) : (part == 0 ? nE : nO); // output indices of this launch
        const int n_con = backward ? (part == 0 ? nE : nO) : (part == 0 ? lo_e : lo_o); // contraction length
        const int o_out = backward ? 0 : (part == 0 ? 0 : nE);                          // first output index (forward: modes)
        const int o_con = backward ? (part == 0 ? 0 : nE) : 0;                          // first contraction index (backward: modes)
        if (n_out == 0 || n_con == 0)
          continue;
        h.K = n_con;
        if (axis == 0)
          {
            h.N = n_out;
            h.A = in + o_con;                           // field, contraction along its rows
            h.B = T + (long)o_out * n + o_con;          // B[k][j] = T[o_out + j][o_con + k]
            h.C = out + o_out;
          }
        else
          {
            h.M = n_out;
            h.A = T + (long)o_out * n + o_con;          // A[i][k] = T[o_out + i][o_con + k]
            h.B = in + (long)o_con * sw;                // field, contraction along its columns
            h.C = out + (long)o_out * sw;
          }
        if (!backward)
          {
            h.fold = part == 0 ? 1 : -1, h.fold_on_A = axis == 0, h.nfold = n;
            h.scale_i0 = o_out; // (rows of this launch are the modes o_out ...)
          }
        else
          {
            h.mirror = part == 0 ? 1 : 2, h.mirror_on_i = axis != 0, h.nmirror = n;
          }
        if (int e = gemm(ctx, h, batch))
          return e;
      }
    return 0;
  }

  // dst = (c_mass M + c_lap K)^-1 src on the free rows (pseudo-inverse if singular), dst = src on
  // the constrained rows; dst == src allowed.  c_mass2 / c_lap2 (not both 0): dst = [(c_mass M + c_lap K)^-1 +
  // (c_mass2 M + c_lap2 K)^-1] src -- both inverses are diagonal in the same modes, one application serves the
  // pressure mass and the pressure Poisson inverse of the Schur complement approximation
  int fdm_apply(adaflo_ctx *ctx, const int field, double *dst, const double *src, const double c_mass, const double c_lap,
                const double c_mass2, const double c_lap2)
  {
    if (field < 0 || field > 2)
      return ADAFLO_EINVAL;
    if (int e = fdm_setup_field(ctx, field))
      return e;
    Fdm            *F  = static_cast<Fdm *>(ctx->fdm);
    const FieldFdm &fd = F->field[field];
    const int       nx = fd.nn[0], ny = fd.nn[1], nz = fd.nn[2];
    const long      n = (long)nx * ny * nz;
    const uint32_t  mask = field == 0 ? ctx->brick.con_u : (field == 1 ? ctx->brick.con_p : ctx->brick.con_ls);
    const unsigned  nb = (unsigned)std::min<long>((n + 255) / 256, 16384);
    // the null mode of a singular operator: |c_m + c_l sum(lambda)| relative to c_l lambda_max
    // (each operator against its OWN scale: with one shared threshold the round-off null eigenvalue of a singular
    // Poisson part of pressure degree >= 2, ~1e-16 lambda_max, can pass the smaller threshold of the mass part)
    const double eps = 1e-10 * (std::abs(c_mass) + std::abs(c_lap) * 12. / (ctx->desc.h[0] * ctx->desc.h[0]));
    const double eps2 = 1e-10 * (std::abs(c_mass2) + std::abs(c_lap2) * 12. / (ctx->desc.h[0] * ctx->desc.h[0]));
    // components with the same 1D problems in all directions (the same kind of boundary for every component) go through
    // the transforms together: a third of the launches, three times the work per launch (129^3 velocity nodes:
    // 27 launches of 10-37 us -> 9)
    bool together = fd.ncomp > 1;
    for (int c = 1; c < fd.ncomp; ++c)
      for (int d = 0; d < 3; ++d)
        together = together && fd.e[c][d].d_S == fd.e[0][d].d_S;
    const int nstack = together ? fd.ncomp : 1;
    if (fd.ncomp == 1)
      {
        // scalar field: no re-layout on the way in, and none on the way out when no row is constrained -- the first
        // transform reads src, the last one writes dst (dst == src is fine: src is only read by the first)
        const Eig1D &ex = fd.e[0][0], &ey = fd.e[0][1], &ez = fd.e[0][2];
        double      *B = F->w1, *Cb = F->w0;
        const bool no_dct = getenv("ADAFLO_FDM_NO_DCT") != nullptr; // (tests / timing of the matrix products; read per call)
        if (mask == 0u && ex.nfft && ey.nfft && ez.nfft && !no_dct)
          {
            // cosine modes in all directions: x, y forward, z forward + scaling + z back in one pass, y, x back
            // (the intermediate arrays have rows padded to 16 doubles: aligned runs in the strided passes)
            const int    P = (nx + 15) / 16 * 16;
            dct::DctArgs A{};
            A.nx = nx, A.ny = ny, A.nz = nz;
            A.lx = ex.d_lamn, A.ly = ey.d_lamn, A.lz = ez.d_lamn, A.ax = ex.d_a2, A.ay = ey.d_a2, A.az = ez.d_a2;
            A.cm = c_mass, A.cl = c_lap, A.eps = eps, A.cm2 = c_mass2, A.cl2 = c_lap2, A.eps2 = eps2;
            const double *in[5]  = {src, B, Cb, B, Cb};
            double       *out[5] = {B, Cb, B, Cb, dst};
            const int     axis[5] = {0, 1, 2, 1, 0};
            for (int pass = 0; pass < 5; ++pass)
              {
                const Eig1D &E = axis[pass] == 0 ? ex : (axis[pass] == 1 ? ey : ez);
                A.in = in[pass], A.out = out[pass], A.tw = E.d_tw, A.axis = axis[pass];
                A.pitch_in = pass == 0 ? nx : P, A.pitch_out = pass == 4 ? nx : P;
                A.n_lines = axis[pass] == 0 ? (long)ny * nz : (axis[pass] == 1 ? (long)P * nz : (long)P * ny);
                if (int e = launch_dct(ctx, E.nfft, pass == 2, A))
                  return e;
              }
            return 0;
          }
        if (int e = transform_axis(ctx, 0, false, ex, src, B, nx, ny, nz))
          return e;
        if (int e = transform_axis(ctx, 1, false, ey, B, Cb, nx, ny, nz))
          return e;
        const ModeScaling sc{ex.d_lam, ey.d_lam, ez.d_lam, c_mass, c_lap, eps, c_mass2, c_lap2, eps2};
        if (int e = transform_axis(ctx, 2, false, ez, Cb, B, nx, ny, nz, 1, &sc))
          return e;
        if (int e = transform_axis(ctx, 0, true, ex, B, Cb, nx, ny, nz))
          return e;
        if (int e = transform_axis(ctx, 1, true, ey, Cb, B, nx, ny, nz))
          return e;
        if (mask == 0u)
          return transform_axis(ctx, 2, true, ez, B, dst, nx, ny, nz);
        if (int e = transform_axis(ctx, 2, true, ez, B, Cb, nx, ny, nz))
          return e;
        hipLaunchKernelGGL(fdm_put_kernel, dim3(nb), dim3(256), 0, ctx->stream, dst, Cb, src, n, 1, 0, nx, ny, nz, mask, 1);
        return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
      }
    for (int c = 0; c < fd.ncomp; c += nstack)
      {
        double        *w0 = F->w0, *w1 = F->w1;
        const long     ns = n * nstack;
        const unsigned nbs = (unsigned)std::min<long>((ns + 255) / 256, 16384);
        if (together)
          hipLaunchKernelGGL(fdm_take_all_kernel, dim3(nbs), dim3(256), 0, ctx->stream, w0, src, n, fd.ncomp);
        else
          hipLaunchKernelGGL(fdm_take_kernel, dim3(nb), dim3(256), 0, ctx->stream, w0, src, n, fd.ncomp, c);
        const Eig1D &ex = fd.e[c][0], &ey = fd.e[c][1], &ez = fd.e[c][2];
        for (int dir = 0; dir < 2; ++dir) // 0: nodes -> modes (S^T), 1: modes -> nodes (S)
          {
            if (int e = transform_axis(ctx, 0, dir == 1, ex, w0, w1, nx, ny, nz, nstack))
              return e;
            if (int e = transform_axis(ctx, 1, dir == 1, ey, w1, w0, nx, ny, nz, nstack))
              return e;
            const ModeScaling sc{ex.d_lam, ey.d_lam, ez.d_lam, c_mass, c_lap, eps, c_mass2, c_lap2, eps2};
            if (int e = transform_axis(ctx, 2, dir == 1, ez, w0, w1, nx, ny, nz, nstack, dir == 0 ? &sc : nullptr))
              return e;
            std::swap(w0, w1); // the result of this direction is the input of the next
          }
        if (together)
          hipLaunchKernelGGL(fdm_put_all_kernel, dim3(nbs), dim3(256), 0, ctx->stream, dst, w0, src, n, fd.ncomp, nx, ny, nz,
                             mask, 3);
        else
          hipLaunchKernelGGL(fdm_put_kernel, dim3(nb), dim3(256), 0, ctx->stream, dst, w0, src, n, fd.ncomp, c, nx, ny, nz,
                             mask, fd.ncomp == 1 ? 1 : 3);
      }
    return hipGetLastError() == hipSuccess ? 0 : ADAFLO_EHIP;
  }
} // namespace adaflo_hip
