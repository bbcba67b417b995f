// development probe: what does the FETCH_SIZE counter of rocprofv3 report on gfx950 for reads of a KNOWN number of
// bytes?  (MI355X_MICROARCH.md: exactly half for 16-B-per-lane streaming reads.)  The Q3..Q5 sweep kernel reads its
// linearisation state 8 B per lane in runs of (k+1)^2 lanes: this probe calibrates the correction for that width.
//   hipcc --offload-arch=gfx950 -O3 fetch_probe.hip -o fetch_probe
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o fetch -- ./fetch_probe
// every kernel reads exactly NBYTES (printed); compare with FETCH_SIZE (KB) per kernel name.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr long NBYTES = 2L << 30;

__global__ __launch_bounds__(256) void read16_contiguous(const double2 *__restrict__ a, double *sink, const long n16)
{
  double s = 0.;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256)
    {
      const double2 v = a[i];
      s += v.x + v.y;
    }
  if (s == 1.2345e300)
    sink[0] = s;
}
__global__ __launch_bounds__(256) void read8_contiguous(const double *__restrict__ a, double *sink, const long n8)
{
  double s = 0.;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256)
    s += a[i];
  if (s == 1.2345e300)
    sink[0] = s;
}
// the access shape of ns_ho_kernel<4>: a wave reads two runs of 25 doubles (two cells) 125 * 12 doubles apart,
// twelve such instructions per quadrature plane, planes 25 doubles apart: every byte of the array is read once
__global__ __launch_bounds__(256) void read8_runs_of_25(const double *__restrict__ a, double *sink, const long ncell)
{
  const int lane = threadIdx.x & 63, cw = lane >> 5, l = lane & 31;
  double    s    = 0.;
  for (long pair = blockIdx.x * 4L + (threadIdx.x >> 6); 2 * pair + 1 < ncell; pair += (long)gridDim.x * 4)
    {
      const double *cell = a + (2 * pair + cw) * (12 * 125);
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int e = 0; e < 12; ++e)
          if (l < 25)
            s += cell[e * 125 + c * 25 + l];
    }
  if (s == 1.2345e300)
    sink[0] = s;
}

int main()
{
  double *a, *sink;
  hipMalloc(&a, NBYTES);
  hipMalloc(&sink, 64);
  hipMemset(a, 0, NBYTES);
  const long ncell = NBYTES / (12 * 125 * 8);
  for (int rep = 0; rep < 3; ++rep)
    {
      hipLaunchKernelGGL(read16_contiguous, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const double2 *>(a), sink, NBYTES / 16);
      hipLaunchKernelGGL(read8_contiguous, dim3(4096), dim3(256), 0, 0, a, sink, NBYTES / 8);
      hipLaunchKernelGGL(read8_runs_of_25, dim3(4096), dim3(256), 0, 0, a, sink, ncell);
    }
  hipDeviceSynchronize();
  std::printf("fetch_probe: read16_contiguous %ld bytes, read8_contiguous %ld bytes, read8_runs_of_25 %ld bytes per launch\n",
              NBYTES, NBYTES, (ncell / 2) * 2 * 12 * 125 * 8);
  return 0;
}
