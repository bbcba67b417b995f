// development probe: HBM rate of a read stream with a small share of writes (the Q2/Q1 sweep kernel reads
// 6.27 GB of state + vectors and writes 0.47 GB per application at 128^3).
//   hipcc --offload-arch=gfx950 -O3 mix_probe.hip -o mix_probe && ./mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double2v __attribute__((ext_vector_type(2)));

// every workgroup streams `per_wg` 16-byte elements; after every `ratio` loaded elements per thread one
// 16-byte element is stored (ratio = 0: no stores); `shift` doubles of misalignment of the store runs
template <int UN>
__global__ __launch_bounds__(256, 2) void mix(const double2v *__restrict__ src, double2v *__restrict__ dst,
                                              const long per_wg, const int ratio, const int shift, double *sink)
{
  const double2v *s = src + (long)blockIdx.x * per_wg;
  double2v       *d = reinterpret_cast<double2v *>(reinterpret_cast<double *>(dst) + shift) + (long)blockIdx.x * (ratio ? per_wg / ratio : 0);
  double2v        acc = {0., 0.};
  long            w = threadIdx.x;
  int             cnt = 0;
  for (long i = threadIdx.x; i + (UN - 1) * 256 < per_wg; i += UN * 256)
    {
      double2v v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u)
        v[u] = __builtin_nontemporal_load(s + i + u * 256);
#pragma unroll
      for (int u = 0; u < UN; ++u)
        acc += v[u];
      cnt += UN;
      if (ratio && cnt >= ratio)
        {
          cnt -= ratio;
          __builtin_nontemporal_store(acc, d + w);
          w += 256;
        }
    }
  if (acc.x == 1.2345e300)
    sink[0] = acc.y;
}

int main()
{
  const long n16 = 6274678784L / 16; // 16-byte elements read
  const int  nwg = 2048;
  const long per_wg = n16 / nwg / 2048 * 2048;
  double2v  *a, *b;
  double    *sink;
  hipMalloc(&a, n16 * 16);
  hipMalloc(&b, n16 * 16 + 65536);
  hipMalloc(&sink, 64);
  hipMemset(a, 0, n16 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char *name, int ratio, int shift) {
    for (int i = 0; i < 3; ++i)
      hipLaunchKernelGGL(mix<8>, dim3(nwg), dim3(256), 0, 0, a, b, per_wg, ratio, shift, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i)
      hipLaunchKernelGGL(mix<8>, dim3(nwg), dim3(256), 0, 0, a, b, per_wg, ratio, shift, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double rb = (double)per_wg * nwg * 16, wb = ratio ? rb / ratio : 0.;
    std::printf("%-34s %.3f ms  read %.2f GB + write %.2f GB -> %.2f TB/s total\n", name, ms, rb / 1e9, wb / 1e9, (rb + wb) / ms / 1e9);
  };
  run("read only", 0, 0);
  run("read + 1/16 writes, aligned", 16, 0);
  run("read + 1/16 writes, shifted 8 B", 16, 1);
  run("read + 1/8 writes, aligned", 8, 0);
  run("read + 1/32 writes, aligned", 32, 0);
  run("read + 1/1 writes (copy)", 8 / 8, 0);
  return 0;
}
