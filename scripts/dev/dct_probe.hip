// dct_probe.hip -- development probe: one pass of the fast cosine transform (csrc/fdm_dct_kernel.hpp) over a
// 257 x 257 x 513 field along every axis, timed with HIP events; build with -DDCT_EXP=<bits> to switch parts off.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDCT_EXP=0 scripts/dev/dct_probe.hip -o /tmp/dct_probe && /tmp/dct_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../adaflo_amd/csrc/fdm_dct_kernel.hpp"
using namespace adaflo_hip::dct;

template <int N, bool FUSED, int AXIS>
__global__ __launch_bounds__(NT, 2) void k(const DctArgs A)
{
  extern __shared__ double lds[];
  dct_body<N, FUSED, AXIS>(A, lds);
}

template <int N, bool FUSED, int AXIS>
float run(DctArgs A, const int n)
{
  using G          = Geo<N>;
  const size_t lds = sizeof(double) * G::L_TOTAL;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<N, FUSED, AXIS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<double> tw(2 * (n));
  for (int m = 0; m < n; ++m)
    tw[2 * m] = std::cos(M_PI * m / (n - 1)), tw[2 * m + 1] = -std::sin(M_PI * m / (n - 1));
  double *d_tw;
  hipMalloc(&d_tw, tw.size() * 8);
  hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice);
  A.tw          = d_tw;
  long nb = (A.n_lines + G::LB - 1) / G::LB;
  nb      = nb > 512 ? 512 : nb; // two workgroups per CU, grid-stride loop over the batches
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((k<N, FUSED, AXIS>), dim3((unsigned)nb), dim3(NT), lds, 0, A);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i)
    hipLaunchKernelGGL((k<N, FUSED, AXIS>), dim3((unsigned)nb), dim3(NT), lds, 0, A);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d_tw);
  return ms / 10;
}

int main()
{
  const int  nx = 257, ny = 257, nz = 513;
  const int  P  = (nx + 15) / 16 * 16; // padded rows, as the intermediate arrays of fdm_apply
  const long n  = (long)P * ny * nz;
  double    *a, *b, *aux;
  hipMalloc(&a, n * 8), hipMalloc(&b, n * 8), hipMalloc(&aux, 8 * 2048);
  hipMemset(a, 0, n * 8), hipMemset(aux, 0, 8 * 2048);
  DctArgs A{};
  A.in = a, A.out = b, A.nx = nx, A.ny = ny, A.nz = nz;
  A.lx = A.ly = A.lz = A.ax = A.ay = A.az = aux, A.cm = 1., A.cl = 1., A.eps = 1e-9;
  A.axis = 0, A.n_lines = (long)ny * nz;
  A.pitch_in = A.pitch_out = nx;
  const float t0a = run<256, false, 0>(A, nx);
  A.pitch_in = nx, A.pitch_out = P;
  const float t0b = run<256, false, 0>(A, nx);
  A.pitch_in = A.pitch_out = P;
  const float t0 = run<256, false, 0>(A, nx);
  std::printf("x pass: contiguous %.3f, contiguous -> padded %.3f, padded %.3f ms\n", t0a, t0b, t0);
  A.axis = 1, A.n_lines = (long)P * nz;
  const float t1 = run<256, false, 1>(A, ny);
  A.axis = 2, A.n_lines = (long)P * ny;
  const float t2 = run<512, false, 2>(A, nz);
  const float t3 = run<512, true, 2>(A, nz);
  std::printf("DCT_EXP=%d  x %.3f  y %.3f  z %.3f  z fused %.3f ms   (%.0f MB per pass)\n", DCT_EXP, t0, t1, t2, t3, 2. * nx * ny * nz * 8 / 1e6);
  return 0;
}
