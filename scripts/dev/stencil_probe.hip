// development probe: q1_stencil_kernel alone on 257 x 257 x 513 nodes, with parts switched off by
// -DQ1S_EXP=1 (no stores)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I adaflo_amd/csrc scripts/dev/stencil_probe.hip -L adaflo_amd/lib -ladaflo_hip
#include "../../adaflo_amd/csrc/q1_sweep.hip"
#include <cstdio>
#include <cstdlib>
using namespace adaflo_hip;
int main(int argc, char **argv)
{
  const int nx = argc > 1 ? atoi(argv[1]) : 257, ny = nx, nz = argc > 2 ? atoi(argv[2]) : 513;
  StencilArgs S{};
  S.nnx = nx, S.nny = ny, S.nnz = nz;
  S.plane = (long)nx * ny, S.comp_stride = S.plane * nz;
  S.flat = (long)((ny + FSR - 1) / FSR) * nx;
  S.blocks_per_plane = (int)(((S.flat + FSW - 1) / FSW + 3) / 4);
  for (int d = 0; d < 3; ++d)
    S.m_off[d] = 1. / 6, S.m_ctr[d] = 1. / 3, S.k_off[d] = -1, S.k_ctr[d] = 1;
  S.c_mass = 1, S.c_lap = .1;
  double *a, *b;
  hipMalloc(&a, S.comp_stride * 8);
  hipMalloc(&b, S.comp_stride * 8);
  hipMemset(a, 0, S.comp_stride * 8);
  S.src = a, S.dst = b;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int lz : {4, 6, 8, 12, 16, 32})
    {
      S.LZ = lz, S.n_chunks = (nz + lz - 1) / lz;
      for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(q1_stencil_kernel, dim3(S.blocks_per_plane * S.n_chunks, 1), dim3(256), 0, 0, S);
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i)
        hipLaunchKernelGGL(q1_stencil_kernel, dim3(S.blocks_per_plane * S.n_chunks, 1), dim3(256), 0, 0, S);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("EXP %d rows %d LZ %2d: %.4f ms  %.2f TB/s (16 B per node)  %s\n",
#ifdef Q1S_EXP
             Q1S_EXP,
#else
             0,
#endif
             FSR, lz, ms / 20, 16.0 * S.comp_stride / (ms / 20 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    }
  return 0;
}
