// hox_intrin.hpp -- the gfx950-specific primitives the x-marching high-order kernel (ns_hox_kernel.hpp) is
// written against: hand-issued ds_read_b64 with counted waits, the wave-level LDS fence, scalar-load tables.
// (tests/emu/hip_emu.hpp provides host stand-ins with the same names so that the kernel source can be run
// lane by lane on the CPU against the oracle; that is test infrastructure, never part of the product path.)
#pragma once
#include <hip/hip_runtime.h>

#include "fe_kernels.hpp"

namespace adaflo_hip
{
  // 1D matrices are wave-uniform: read them with scalar loads through the constant address space
  typedef const double __attribute__((address_space(4))) *ctab_t;
  __device__ __forceinline__ ctab_t as_ctab(const double *p) { return (ctab_t)p; }
  // opaque copies keep the compiler from hoisting re-computable values out of the marching loop
  __device__ __forceinline__ void opaque(ctab_t &t) { asm volatile("" : "+s"(t)); }
  __device__ __forceinline__ void opaque(int &v) { asm volatile("" : "+v"(v)); }
  __device__ __forceinline__ void opaque(unsigned &v) { asm volatile("" : "+v"(v)); }

  // the offset becomes known only after `v` has been computed: pins a prefetch behind the arithmetic that frees its
  // destination registers (otherwise all loads of an unrolled loop are hoisted to its top and stay live together).
  // (An offset, not the pointer: behind an opaque pointer the compiler no longer knows the address space.)
  __device__ __forceinline__ void pin_after(unsigned &off, const double v) { asm volatile("" : "+v"(off) : "v"(v)); }

  // the workgroup's dynamic LDS
  __device__ __forceinline__ double *dyn_lds()
  {
    extern __shared__ double adaflo_dyn_lds[];
    return adaflo_dyn_lds;
  }

  // compiler-only fence between a wave's LDS write and read phases (the hardware executes one wave's LDS
  // operations in order; this keeps the compiler from moving them across)
  __device__ __forceinline__ void wave_sync()
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0); // phases stay phases: the scheduler otherwise interleaves them for ILP and the
                                       // kernel needs 440 instead of ~230 registers
  }

  __device__ __forceinline__ unsigned lds_byte_addr(const void *p)
  {
    return (unsigned)(size_t)p; // LDS aperture: the low 32 bits are the LDS byte address
  }
  // hand-issued ds_read_b64 (the compiler would pair strided reads into ds_read2_b64 = half the LDS rate,
  // MI355X_MICROARCH.md LDS table); results become usable after ds_wait<N> with N = younger reads in flight
  template <int OFF>
  __device__ __forceinline__ double ds_rd(const unsigned a)
  {
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
  }
  template <int OFF>
  __device__ __forceinline__ void ds_wr(const unsigned a, const double v)
  {
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(a), "v"(v), "n"(OFF) : "memory");
  }
  template <int CNT, int NM>
  __device__ __forceinline__ void ds_wait(double (&x)[NM])
  {
    constexpr int C = CNT > 15 ? 15 : CNT;
    static_assert(NM >= 3 && NM <= 6, "line length");
    if constexpr (NM == 3)
      asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "n"(C));
    else if constexpr (NM == 4)
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(C));
    else if constexpr (NM == 5)
      asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]) : "n"(C));
    else
      asm volatile("s_waitcnt lgkmcnt(%6)"
                   : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5])
                   : "n"(C));
  }
  template <int CNT>
  __device__ __forceinline__ void ds_wait1(double &x)
  {
    constexpr int C = CNT > 15 ? 15 : CNT;
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(C));
  }
} // namespace adaflo_hip
