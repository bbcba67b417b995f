// probe: semantics of global_load_lds_dwordx3 on gfx950 (LDS address per lane, alignment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* g, float* out, int lds_off_dw)
{
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
  __syncthreads();
  const unsigned voff = threadIdx.x * 12u + 4u;  // global byte offset per lane (4-B aligned only)
  const unsigned ldsb = (unsigned)(size_t)(lds + lds_off_dw);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_waitcnt vmcnt(0)" :: "s"(ldsb), "v"(voff), "s"(g) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main()
{
  std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = i;
  float *g, *o; hipMalloc(&g, 4096*4); hipMalloc(&o, 1024*4);
  hipMemcpy(g, h.data(), 4096*4, hipMemcpyHostToDevice);
  for (int off : {0, 1, 2, 3}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o, off);
    std::vector<float> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 192; ++i) if (r[off + i] != (float)(i + 1)) ++bad;
    printf("lds_off %d dwords: mismatches %d of 192; first words: %g %g %g %g %g | after: %g %g\n", off, bad, r[off], r[off+1], r[off+2], r[off+3], r[off+4], r[off+192], r[off+193]);
  }
  return 0;
}
