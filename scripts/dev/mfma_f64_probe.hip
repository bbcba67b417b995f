// mfma_f64_probe.hip -- settles the "FP64 MFMA for the (k+1)x(k+1) 1D contractions" question
// with numbers.  Measures on one MI355X:
//   (a) issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 (cycles per instruction
//       per SIMD, 4 independent accumulator chains per wave, 1 and 2 waves per SIMD, all CUs),
//   (b) the f64 vector FMA rate with the same launch shape,
//   (c) the rate of USEFUL flops when the A operand is the block-diagonal packing of three 5x5
//       matrices (the best packing for k = 4: 15 of 16 rows, K = 15 of 16).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/dev/mfma_f64_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double double4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma16_kernel(double *out, const int iters, long long *cyc)
{
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  double4v     c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i)
    {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *cyc = t1 - t0;
}

__global__ __launch_bounds__(256) void mfma4_kernel(double *out, const int iters, long long *cyc)
{
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  double       c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i)
    {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *cyc = t1 - t0;
}

__global__ __launch_bounds__(256) void vfma_kernel(double *out, const int iters, long long *cyc)
{
  const double a = 1.0 + 1e-9 * threadIdx.x;
  double       c[8];
  for (int j = 0; j < 8; ++j)
    c[j] = j;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      c[j] = __builtin_fma(a, c[j], 1e-3);
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int j = 0; j < 8; ++j)
    s += c[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *cyc = t1 - t0;
}

template <typename K>
static void run(const char *name, K kernel, const int blocks, const int iters, const double flop_per_wave_iter,
                const int instr_per_iter, const double useful_fraction)
{
  double    *out;
  long long *cyc, hcyc = 0;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipMalloc(&cyc, sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&hcyc, cyc, sizeof(hcyc), hipMemcpyDeviceToHost);
  const double waves = blocks * 4.0;
  const double tf    = flop_per_wave_iter * iters * waves / (ms * 1e-3) / 1e12;
  // s_memtime ticks at 100 MHz on this part: report event time and the per-instruction time instead
  const double waves_per_simd = waves / (256.0 * 4.0);
  const double ns_per_instr   = ms * 1e6 / ((double)iters * instr_per_iter * (waves_per_simd < 1 ? 1 : waves_per_simd));
  printf("%-34s blocks %5d  %8.3f ms  %7.2f TFLOP/s issued  %7.2f TFLOP/s useful  %6.2f ns per instr per SIMD  (memtime ticks %lld)\n",
         name, blocks, ms, tf, tf * useful_fraction, ns_per_instr, hcyc);
  hipFree(out);
  hipFree(cyc);
}

int main()
{
  const int iters = 20000;
  // 16x16x4 f64: 2*16*16*4 = 2048 flop per wave-instruction, 4 per iteration
  for (int blocks : {256, 512})
    run("v_mfma_f64_16x16x4_f64", mfma16_kernel, blocks, iters, 4 * 2048.0, 4,
        3.0 * 5 * 5 / (16.0 * 16.0)); // three 5x5 blocks on the diagonal of a 16x16 A
  // 4x4x4 (4 blocks): 4 * 2*4*4*4 = 512 flop per wave-instruction
  for (int blocks : {256, 512})
    run("v_mfma_f64_4x4x4_4b_f64", mfma4_kernel, blocks, iters, 4 * 512.0, 4, 1.0);
  // vector FMA: 64 lanes * 2 flop, 8 per iteration
  for (int blocks : {256, 512})
    run("v_fma_f64 (vector)", vfma_kernel, blocks, iters, 8 * 128.0, 8, 1.0);
  return 0;
}
