// development (round 6): does gfx950 need wait states (a) between an SALU write of EXEC and a DPP move, (b) between an f64
// VALU result and a DPP move that reads it, (c) between v_accvgpr_read and a DPP move?  Every sequence is ONE asm statement
// (nothing of the compiler in between).  Each test runs the sequence in a loop on every wave of a grid that fills the chip
// (at eight waves and at one wave per SIMD) and counts the lanes whose DPP result differs from the same broadcast done
// through the LDS crossbar (__shfl).
//   hipcc --offload-arch=gfx950 -O2 scripts/dev/hazard_probe.hip -o gpurun_out/hazard_probe   (here; travels? no ->
//   build on the box)        gpurun -- 'hipcc ... && ./hazard_probe'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NOP0 ""
#define NOP1 "s_nop 0\n\t"
#define NOP2 "s_nop 1\n\t"
#define NOP3 "s_nop 2\n\t"
#define NOP4 "s_nop 3\n\t"
#define NOP5 "s_nop 4\n\t"
#define NOP6 "s_nop 5\n\t"

#define DPP3 "quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"

struct Args
{
  const unsigned *in;
  unsigned       *out;
  double         *dump;
  unsigned       *bad;
  int             iters;
};

#define KERNEL_HEAD(name)                                                               \
  __global__ void name(const Args A)                                                    \
  {                                                                                     \
    extern __shared__ double lds[];                                                     \
    const unsigned           tid  = blockIdx.x * blockDim.x + threadIdx.x;              \
    unsigned                 x    = A.in[tid], acc = 0;                                 \
    const unsigned long long mask = 0x7777777777777777ull;                              \
    unsigned long long       saved = 0;                                                 \
    double                  *p     = A.dump + tid;                                      \
    double                   v     = (double)tid;                                       \
    (void)saved, (void)p, (void)v, (void)mask;                                          \
    if (A.iters < 0)                                                                    \
      lds[threadIdx.x] = 1.;                                                            \
    for (int it = 0; it < A.iters; ++it)                                                \
      {                                                                                 \
        unsigned r, want;                                                               \
        x = x * 1664525u + 1013904223u;

#define KERNEL_TAIL                                                                     \
        acc += (r != want) ? 1u : 0u;                                                   \
      }                                                                                 \
    if (acc)                                                                            \
      atomicAdd(A.bad, acc);                                                            \
    A.out[tid] = x;                                                                     \
  }

// legend: a/b/c = EXEC restored by SALU, <n> wait states, DPP; d = v_fma_f64 -> DPP(lo), g = v_fma_f64 -> DPP(hi), h = v_mul_f64,
// i = v_add_f64, j = v_fmac_f64 (VOP2), f = v_add_u32 (control: documented 2 wait states), e = v_accvgpr_read -> DPP
// A: s_mov exec, mask; store; s_mov exec, -1; nops; DPP from lane 3 (off in the narrowed mask)
#define TEST_A(n)                                                                       \
  KERNEL_HEAD(test_a##n)                                                                \
  want = __shfl(x, (threadIdx.x & ~3u) | 3u, 64);                                       \
  asm volatile("s_mov_b64 exec, %3\n\t"                                                 \
               "global_store_dwordx2 %1, %2, off\n\t"                                   \
               "s_mov_b64 exec, -1\n\t" NOP##n "v_mov_b32_dpp %0, %4 " DPP3             \
               : "=&v"(r)                                                               \
               : "v"(p), "v"(v), "s"(mask), "v"(x)                                      \
               : "memory");                                                             \
  KERNEL_TAIL
// B: the form of store_b128_dst: s_and_saveexec; store; s_mov exec, saved; nops; DPP
#define TEST_B(n)                                                                       \
  KERNEL_HEAD(test_b##n)                                                                \
  want = __shfl(x, (threadIdx.x & ~3u) | 3u, 64);                                       \
  asm volatile("s_and_saveexec_b64 %1, %4\n\t"                                          \
               "global_store_dwordx2 %2, %3, off\n\t"                                   \
               "s_mov_b64 exec, %1\n\t" NOP##n "v_mov_b32_dpp %0, %5 " DPP3             \
               : "=&v"(r), "=&s"(saved)                                                 \
               : "v"(p), "v"(v), "s"(mask), "v"(x)                                      \
               : "memory", "scc");                                                      \
  KERNEL_TAIL
// C: no memory instruction between the two EXEC writes
#define TEST_C(n)                                                                       \
  KERNEL_HEAD(test_c##n)                                                                \
  want = __shfl(x, (threadIdx.x & ~3u) | 3u, 64);                                       \
  asm volatile("s_mov_b64 exec, %1\n\t"                                                 \
               "s_mov_b64 exec, -1\n\t" NOP##n "v_mov_b32_dpp %0, %2 " DPP3             \
               : "=&v"(r)                                                               \
               : "s"(mask), "v"(x)                                                      \
               : "memory");                                                             \
  KERNEL_TAIL
// E: v_accvgpr_read -> DPP
#define TEST_E(n)                                                                       \
  KERNEL_HEAD(test_e##n)                                                                \
  want = __shfl(x, (threadIdx.x & ~3u) | 3u, 64);                                       \
  unsigned t;                                                                           \
  asm volatile("v_accvgpr_write_b32 a0, %2\n\t"                                         \
               "s_nop 4\n\t"                                                            \
               "v_accvgpr_read_b32 %1, a0\n\t" NOP##n "v_mov_b32_dpp %0, %1 " DPP3      \
               : "=&v"(r), "=&v"(t)                                                     \
               : "v"(x)                                                                 \
               : "a0");                                                                 \
  KERNEL_TAIL

// D*: f64 producer -> DPP of one half of its result (the compiler's rule: 2 wait states).  OPSTR = the producing instruction
#define TEST_F64(name, OPSTR, HALF, n)                                                  \
  KERNEL_HEAD(name##n)                                                                  \
  double   f = (double)x * 1.25 + 3.;                                                   \
  unsigned lo, hi;                                                                      \
  asm volatile("v_mov_b32 v200, %5\n\tv_mov_b32 v201, %6\n\ts_nop 4\n\t"              \
               OPSTR "\n\t" NOP##n "v_mov_b32_dpp %0, " HALF " " DPP3 "\n\t"            \
               "s_nop 4\n\tv_mov_b32 %1, v200\n\tv_mov_b32 %2, v201"                   \
               : "=&v"(r), "=&v"(lo), "=&v"(hi)                                         \
               : "v"(f), "v"(1.5), "v"(__double2loint(v)), "v"(__double2hiint(v))       \
               : "v200", "v201");                                                       \
  want = __shfl(HALF[3] == '0' ? lo : hi, (threadIdx.x & ~3u) | 3u, 64);                \
  KERNEL_TAIL
#define TEST_D(n) TEST_F64(test_d, "v_fma_f64 v[200:201], %3, %4, v[200:201]", "v200", n)
#define TEST_G(n) TEST_F64(test_g, "v_fma_f64 v[200:201], %3, %4, v[200:201]", "v201", n)
#define TEST_H(n) TEST_F64(test_h, "v_mul_f64 v[200:201], %3, %4", "v200", n)
#define TEST_I(n) TEST_F64(test_i, "v_add_f64 v[200:201], %3, %4", "v200", n)
#define TEST_J(n) TEST_F64(test_j, "v_fmac_f64_e32 v[200:201], %3, %4", "v200", n)
// F: control -- a 32-bit producer (the documented hazard: 2 wait states)
#define TEST_F(n)                                                                       \
  KERNEL_HEAD(test_f##n)                                                                \
  unsigned t;                                                                           \
  asm volatile("v_add_u32 %1, %2, %2\n\t" NOP##n "v_mov_b32_dpp %0, %1 " DPP3           \
               : "=&v"(r), "=&v"(t)                                                     \
               : "v"(x));                                                               \
  want = __shfl(t, (threadIdx.x & ~3u) | 3u, 64);                                       \
  KERNEL_TAIL

#define ALL(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6)
ALL(TEST_A)
ALL(TEST_B)
ALL(TEST_C)
ALL(TEST_D)
ALL(TEST_G)
ALL(TEST_H)
ALL(TEST_I)
ALL(TEST_J)
ALL(TEST_F)
ALL(TEST_E)

typedef void (*kern_t)(const Args);
struct Entry
{
  const char *name;
  kern_t      k;
};
#define ENT(t, n) {#t #n, t##n},
#define ENTS(t) ENT(t, 0) ENT(t, 1) ENT(t, 2) ENT(t, 3) ENT(t, 4) ENT(t, 5) ENT(t, 6)
static Entry table[] = {ENTS(test_a) ENTS(test_b) ENTS(test_c) ENTS(test_d) ENTS(test_g) ENTS(test_h) ENTS(test_i) ENTS(test_j) ENTS(test_f) ENTS(test_e)};

int main()
{
  const int blocks = 256 * 8, threads = 256, n = blocks * threads, iters = 2000;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i)
    h[i] = 2654435761u * (unsigned)(i + 1);
  Args A;
  (void)hipMalloc((void **)&A.in, n * 4);
  (void)hipMalloc((void **)&A.out, n * 4);
  (void)hipMalloc((void **)&A.dump, n * 8);
  (void)hipMalloc((void **)&A.bad, 4);
  (void)hipMemcpy((void *)A.in, h.data(), n * 4, hipMemcpyHostToDevice);
  A.iters = iters;
  printf("%-10s %14s %14s   (lanes x iterations with a wrong DPP result, of %ld)\n", "test", "8 waves/SIMD", "1 wave/SIMD", (long)n * iters);
  for (const Entry &e : table)
    {
      unsigned bad[2];
      for (int occ = 0; occ < 2; ++occ)
        {
          (void)hipMemset(A.bad, 0, 4);
          const size_t lds = occ ? 100 * 1024 : 0;
          if (lds)
            (void)hipFuncSetAttribute((const void *)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), lds, 0, A);
          hipError_t err = hipDeviceSynchronize();
          if (err != hipSuccess)
            printf("  %s: %s\n", e.name, hipGetErrorString(err));
          (void)hipMemcpy(&bad[occ], A.bad, 4, hipMemcpyDeviceToHost);
        }
      printf("%-10s %14u %14u\n", e.name, bad[0], bad[1]);
    }
  return 0;
}
