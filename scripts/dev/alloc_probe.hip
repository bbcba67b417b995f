// development probe: does the streaming-read rate of a 6.2 GB buffer depend on which allocation it is?
// (the Q2/Q1 kernel's time moves by 10 % between processes; 93 % of its traffic is the state stream)
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/alloc_probe.hip -o scripts/dev/build/alloc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void stream_read(const double2 *__restrict__ a, const long n, double *out)
{
  double s = 0.;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    {
      const double2 v = a[i];
      s += v.x + v.y;
    }
  if (s == 1.2345e300)
    out[0] = s;
}
int main()
{
  const long bytes = 6274678784L, n = bytes / 16;
  std::vector<double2 *> buf;
  double *out;
  hipMalloc(&out, 8);
  for (int b = 0; b < 8; ++b)
    {
      double2 *p = nullptr;
      if (hipMalloc(&p, bytes) != hipSuccess)
        break;
      hipMemset(p, 0, bytes);
      buf.push_back(p);
    }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep)
    for (size_t b = 0; b < buf.size(); ++b)
      {
        for (int i = 0; i < 2; ++i)
          hipLaunchKernelGGL(stream_read, dim3(256 * 16), dim3(256), 0, 0, buf[b], n, out);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i)
          hipLaunchKernelGGL(stream_read, dim3(256 * 16), dim3(256), 0, 0, buf[b], n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("buffer %zu (%p): %.4f ms  %.2f TB/s\n", b, (void *)buf[b], ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
      }
  return 0;
}
