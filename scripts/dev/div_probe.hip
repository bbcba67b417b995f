// development probe: q2q1_divergence_kernel alone on n^3 cells (the DIV_EXP variants -- loads with trivial
// sums, arithmetic without the loads -- were temporary edits of the kernel)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I adaflo_amd/csrc -I include scripts/dev/div_probe.hip -L adaflo_amd/lib -ladaflo_hip
#include "../../adaflo_amd/csrc/ns_divergence.hip"
#include <cstdio>
#include <cstdlib>
using namespace adaflo_hip;
int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 128, nz = argc > 2 ? atoi(argv[2]) : n;
  DivArgs A{};
  A.npx = A.npy = n + 1, A.npz = nz + 1, A.nvx = A.nvy = 2 * n + 1, A.nvz = 2 * nz + 1;
  A.flat = (long)A.npx * A.npy;
  A.blocks_per_chunk = (int)(((A.flat + DSW - 1) / DSW + 3) / 4);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 3; ++b)
      {
        for (int d = 0; d < 3; ++d)
          A.m[d][a][b] = .1 + a + b;
        A.c[a][b] = a - b;
      }
  A.weight = -1;
  const long nu = 3L * A.nvx * A.nvy * A.nvz, np = A.flat * A.npz;
  double    *u, *p;
  hipMalloc(&u, nu * 8);
  hipMalloc(&p, np * 8);
  hipMemset(u, 0, nu * 8);
  hipMemset(p, 0, np * 8);
  A.src_u = u, A.dst_p = p;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int lz : {4, 6, 8, 12, 16})
    {
      A.LZ = lz, A.n_chunks = (A.npz + lz - 1) / lz;
      const unsigned grid = A.blocks_per_chunk * A.n_chunks;
      for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(q2q1_divergence_kernel, dim3(grid), dim3(256), 0, 0, A);
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i)
        hipLaunchKernelGGL(q2q1_divergence_kernel, dim3(grid), dim3(256), 0, 0, A);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 8. * nu + 16. * np;
      printf("EXP %d LZ %2d: %.4f ms  %.2f TB/s  %s\n",
#ifdef DIV_EXP
             DIV_EXP,
#else
             0,
#endif
             lz, ms / 20, bytes / (ms / 20 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    }
  return 0;
}
