// hip_emu.hpp -- TEST INFRASTRUCTURE ONLY.  A host-side lane emulator that runs the *device source* of a HIP
// kernel header (e.g. adaflo_amd/csrc/ns_hox_kernel.hpp) on the CPU so that index logic, ownership rules and
// LDS hand-offs can be checked against the oracle in this GPU-less container.  Nothing under adaflo_amd/ includes
// or loads this file; the product path is the gfx950 code object and fails without a GPU.
//
// Model: one workgroup at a time; every thread of the workgroup is a ucontext fiber.  A fiber runs until it
// reaches a barrier: __syncthreads() (all live fibers of the workgroup) or wave_sync() (the 64 fibers of its
// wave).  On the hardware the LDS operations of one wave execute in program order, so a wave's lanes need no
// s_barrier between an LDS write phase and the read phase that follows; the kernel marks those points with
// wave_sync() (a compiler fence on the GPU), and the emulator turns them into a wave-wide rendezvous.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <utility>
#include <vector>

struct emu_dim3
{
  unsigned x = 1, y = 1, z = 1;
  emu_dim3() = default;
  emu_dim3(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace emu
{
  inline emu_dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
  constexpr size_t LDS_BYTES = 160 * 1024;
  alignas(64) inline char g_lds[LDS_BYTES];

  struct Fiber
  {
    ucontext_t ctx;
    char      *stack = nullptr;
    int        state = 0; // 0 runnable, 1 at wave barrier, 2 at block barrier, 3 done
  };
  inline std::vector<Fiber>     g_fibers;
  inline ucontext_t             g_sched;
  inline int                    g_current = 0;
  inline std::function<void()> *g_body    = nullptr;
  constexpr size_t              STACK     = 512 * 1024;

  inline void fiber_main()
  {
    (*g_body)();
    g_fibers[g_current].state = 3;
    swapcontext(&g_fibers[g_current].ctx, &g_sched);
  }
  inline void yield(const int st)
  {
    Fiber &f = g_fibers[g_current];
    f.state  = st;
    swapcontext(&f.ctx, &g_sched);
  }

  inline void run_block(const unsigned nthreads)
  {
    if (g_fibers.size() < nthreads)
      g_fibers.resize(nthreads);
    for (unsigned t = 0; t < nthreads; ++t)
      {
        Fiber &f = g_fibers[t];
        if (!f.stack)
          f.stack = (char *)std::malloc(STACK);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp   = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link          = &g_sched;
        makecontext(&f.ctx, (void (*)())fiber_main, 0);
        f.state = 0;
      }
    for (;;)
      {
        bool progress = false, all_done = true;
        for (unsigned t = 0; t < nthreads; ++t)
          if (g_fibers[t].state == 0)
            {
              g_current     = (int)t;
              g_threadIdx.x = t;
              swapcontext(&g_sched, &g_fibers[t].ctx);
              progress = true;
            }
        // release wave barriers
        for (unsigned w = 0; w * 64 < nthreads; ++w)
          {
            bool all = true, any = false;
            for (unsigned t = w * 64; t < nthreads && t < (w + 1) * 64; ++t)
              {
                if (g_fibers[t].state == 1)
                  any = true;
                else if (g_fibers[t].state != 3)
                  all = false;
              }
            if (any && all)
              for (unsigned t = w * 64; t < nthreads && t < (w + 1) * 64; ++t)
                if (g_fibers[t].state == 1)
                  {
                    g_fibers[t].state = 0;
                    progress          = true;
                  }
          }
        bool all_blk = true, any_blk = false;
        for (unsigned t = 0; t < nthreads; ++t)
          {
            if (g_fibers[t].state == 2)
              any_blk = true;
            else if (g_fibers[t].state != 3)
              all_blk = false;
            if (g_fibers[t].state != 3)
              all_done = false;
          }
        if (any_blk && all_blk)
          for (unsigned t = 0; t < nthreads; ++t)
            if (g_fibers[t].state == 2)
              {
                g_fibers[t].state = 0;
                progress          = true;
              }
        if (all_done)
          break;
        if (!progress)
          {
            std::fprintf(stderr, "hip_emu: deadlock in block %u (divergent barriers)\n", g_blockIdx.x);
            std::abort();
          }
      }
  }

  // launch<<<grid, block>>>: `body` is called once per thread with threadIdx / blockIdx set
  template <class F>
  void launch(const unsigned grid, const unsigned block, F &&body)
  {
    std::function<void()> fn = body;
    g_body                   = &fn;
    g_gridDim                = emu_dim3(grid);
    g_blockDim               = emu_dim3(block);
    for (unsigned b = 0; b < grid; ++b)
      {
        g_blockIdx = emu_dim3(b);
        std::memset(g_lds, 0xff, LDS_BYTES); // NaN pattern: reads of never-written LDS show up
        run_block(block);
      }
    g_body = nullptr;
  }
} // namespace emu

// ---- the HIP spellings the kernel headers use -------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
#define threadIdx emu::g_threadIdx
#define blockIdx emu::g_blockIdx
#define blockDim emu::g_blockDim
#define gridDim emu::g_gridDim
inline void __syncthreads() { emu::yield(2); }
inline int  __builtin_amdgcn_readfirstlane(const int v) { return v; } // (callers pass wave-uniform values)
inline void __builtin_amdgcn_sched_barrier(int) {}
template <class T>
inline T min(const T a, const T b)
{
  return a < b ? a : b;
}
template <class T>
inline T max(const T a, const T b)
{
  return a > b ? a : b;
}

// ---- the helper layer of csrc/hox_intrin.hpp, host flavour -------------------------------------------------
namespace adaflo_hip
{
  typedef const double *ctab_t;
  inline ctab_t as_ctab(const double *p) { return p; }
  inline void   opaque(ctab_t &) {}
  inline void   opaque(int &) {}
  inline void   opaque(unsigned &) {}
  inline void   opaque(double &) {}
  inline void   opaque_ptr(const double *&) {}
  inline void   opaque_s(int &) {}
  inline void   pin_after(unsigned &, const double) {}
  inline void   sink(const double) {}
  inline unsigned long long clock_now() { return 0; }
  inline void   wave_sync() { emu::yield(1); }
  inline void   wave_fence() { emu::yield(1); }
  inline void   emu_sync() { emu::yield(1); }
  // cross-lane moves of ns_hop_kernel (v_permlane16_swap / v_permlane32_swap on the device): every lane of the wave
  // deposits its value, the wave meets, every lane picks up its partner's.  Lanes that have left the kernel hold no
  // value anybody reads.
  inline double g_xlane[1024];
  inline double xlane_from(const double v, const int delta)
  {
    const int t = (int)emu::g_threadIdx.x, l = t & 63;
    g_xlane[t]  = v;
    emu::yield(1);
    const double r = l >= delta && ((l / delta) & 1) ? g_xlane[t - delta] : v;
    emu::yield(1);
    return r;
  }
  inline double from_row_below(const double v) { return xlane_from(v, 16); }  // rows 1, 3 <- rows 0, 2
  inline double from_half_below(const double v) { return xlane_from(v, 32); } // lanes 32..63 <- lanes 0..31
  inline void   lds_barrier() { emu::yield(2); }
  inline void   lds_flag_set(const unsigned a, const int v) { *reinterpret_cast<volatile int *>(emu::g_lds + a) = v; }
  inline void   lds_flag_wait(const unsigned a, const int target)
  {
    while (*reinterpret_cast<volatile int *>(emu::g_lds + a) < target)
      emu::yield(0); // (cooperative spin: the other fibers run in between)
  }
  inline double *dyn_lds() { return reinterpret_cast<double *>(emu::g_lds); } // the workgroup's dynamic LDS
  inline unsigned lds_byte_addr(const void *p) { return (unsigned)((const char *)p - emu::g_lds); }
  template <int OFF>
  inline double ds_rd(const unsigned a)
  {
    double v;
    std::memcpy(&v, emu::g_lds + a + OFF, 8);
    return v;
  }
  template <int OFF>
  inline void ds_wr(const unsigned a, const double v)
  {
    std::memcpy(emu::g_lds + a + OFF, &v, 8);
  }
  template <int CNT, int NM>
  inline void ds_wait(double (&)[NM])
  {}
  // LDS-DMA, synchronous here: lane l copies 16 bytes to lds_byte + 16 * l
  inline void dma_b128(const double *sbase, const unsigned voff, const unsigned lds_byte)
  {
    std::memcpy(emu::g_lds + lds_byte + 16 * (emu::g_threadIdx.x & 63), reinterpret_cast<const char *>(sbase) + voff, 16);
  }
  template <int N>
  inline void wait_vmcnt()
  {}
  inline const double *uniform_ptr(const double *p) { return p; }
  struct hox_double2
  {
    double x, y;
  };
  template <int OFF>
  inline void ds_rd128(const unsigned a, hox_double2 &v)
  {
    std::memcpy(&v, emu::g_lds + a + OFF, 16);
  }
  inline void lds_arrived(hox_double2 &, hox_double2 &) {}
  inline long xcd_remap(const long b, const long n)
  {
    const long per = n / 8;
    if (b >= per * 8)
      return b;
    return (b % 8) * per + b / 8;
  }
  inline bool on_constrained_face(const int I, const int J, const int K, const int nnx, const int nny, const int nnz,
                                  const uint32_t mask, const int stride, const int comp)
  {
    uint32_t f = 0;
    f |= (I == 0) ? (1u << (stride * 0 + comp)) : 0u;
    f |= (I == nnx - 1) ? (1u << (stride * 1 + comp)) : 0u;
    f |= (J == 0) ? (1u << (stride * 2 + comp)) : 0u;
    f |= (J == nny - 1) ? (1u << (stride * 3 + comp)) : 0u;
    f |= (K == 0) ? (1u << (stride * 4 + comp)) : 0u;
    f |= (K == nnz - 1) ? (1u << (stride * 5 + comp)) : 0u;
    return (f & mask) != 0u;
  }
} // namespace adaflo_hip
