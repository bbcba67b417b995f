// development probe: HBM rate of the z-marching access pattern of q1_stencil_kernel
// (a block owns 256 consecutive nodes of a plane and walks LZ planes) with 1 load + 1 store per node,
// against a flat grid-stride copy.      hipcc --offload-arch=gfx950 -O3 march_copy.hip -o march_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void march(const double *__restrict__ src, double *__restrict__ dst, long plane,
                                             int nnz, int LZ, int bpp, int nloads)
{
  const long wg = blockIdx.x;
  const int  chunk = (int)(wg / bpp);
  long       p     = (wg % bpp) * 256 + threadIdx.x;
  if (p >= plane)
    p = plane - 1;
  const int k0 = chunk * LZ, k1 = min(k0 + LZ, nnz);
  for (int k = k0; k < k1; ++k)
    {
      const double *s = src + (long)k * plane + p;
      double        v = s[0];
      if (nloads >= 3)
        v += s[p > 0 ? -1 : 0] + s[p < plane - 1 ? 1 : 0];
      if (nloads >= 9)
        {
          const long up = p + 257 < plane ? 257 : 0, dn = p >= 257 ? -257 : 0;
          v += s[up] + s[dn] + s[up + (p + 258 < plane ? 1 : 0)] + s[dn + (p > 257 ? -1 : 0)] + s[up - (p + 256 < plane && p > 0 ? 1 : 0)] +
               s[dn + (p >= 256 ? 1 : 0)];
        }
      __builtin_nontemporal_store(v, dst + (long)k * plane + p);
    }
}

__global__ __launch_bounds__(256) void flat(const double *__restrict__ src, double *__restrict__ dst, long n)
{
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    __builtin_nontemporal_store(src[i], dst + i);
}

int main()
{
  const int  nx = 257, ny = 257, nz = 513;
  const long plane = (long)nx * ny, n = plane * nz;
  double    *a, *b;
  hipMalloc(&a, n * 8);
  hipMalloc(&b, n * 8);
  hipMemset(a, 0, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int bpp = (int)((plane + 255) / 256);
  auto      run = [&](const char *name, auto launch) {
    for (int i = 0; i < 3; ++i)
      launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i)
      launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  %.2f TB/s (16 B per node)\n", name, ms / 20, 16.0 * n / (ms / 20 * 1e-3) / 1e12);
  };
  run("flat copy", [&] { hipLaunchKernelGGL(flat, dim3(16384), dim3(256), 0, 0, a, b, n); });
  for (int lz : {8, 32, 128})
    for (int nl : {1, 3, 9})
      {
        char name[64];
        snprintf(name, 64, "march LZ=%d loads=%d", lz, nl);
        const int chunks = (nz + lz - 1) / lz;
        run(name, [&] { hipLaunchKernelGGL(march, dim3(bpp * chunks), dim3(256), 0, 0, a, b, plane, nz, lz, bpp, nl); });
      }
  return 0;
}
