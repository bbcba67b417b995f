// Hand-issued asynchronous global -> LDS copies (LDS-DMA of gfx950) for the stencil kernels.
//
// The copies are invisible to the compiler's s_waitcnt bookkeeping: every consumer is guarded by an
// explicit wait_vmcnt<N>() (VMEM operations of a wave retire in issue order, so "at most N
// outstanding" means everything older than the N youngest has landed) followed by lds_barrier().
// Loads the compiler can see must not sit in the same loop: it would answer them with
// s_waitcnt vmcnt(0) at the next join and empty the queue in every iteration --
// rare ones go through load_now().
#pragma once
#include <hip/hip_runtime.h>

namespace adaflo_hip
{
  namespace lds_dma
  {
    // LDS-only workgroup barrier: unlike __syncthreads() it does not wait for vmcnt
    __device__ __forceinline__ void lds_barrier()
    {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    template <int N>
    __device__ __forceinline__ void wait_vmcnt()
    {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    }

    __device__ __forceinline__ unsigned uniform32(const unsigned x)
    {
      return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
    }
    __device__ __forceinline__ unsigned long long uniform64(const unsigned long long x)
    {
      return ((unsigned long long)uniform32((unsigned)(x >> 32)) << 32) | uniform32((unsigned)x);
    }

    // every lane copies 4 bytes from sbase + voff to LDS byte lds_byte + 4 * lane (lds_byte and sbase
    // wave-uniform); EXEC is all ones where this is called
    __device__ __forceinline__ void copy_b32(const void *sbase, const unsigned voff, const unsigned lds_byte)
    {
      asm volatile("s_mov_b32 m0, %0\n\t"
                   "global_load_lds_dword %1, %2" ::"s"(lds_byte),
                   "v"(voff), "s"(sbase)
                   : "memory");
    }

    // a load in a rarely taken branch, with its own wait inside the branch
    __device__ __forceinline__ double load_now(const double *p)
    {
      double v;
      asm volatile("global_load_dwordx2 %0, %1, off\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=v"(v)
                   : "v"(p)
                   : "memory");
      return v;
    }

    __device__ __forceinline__ unsigned lds_addr(const void *p)
    {
      return (unsigned)(size_t)p; // LDS aperture: the low 32 bits are the LDS byte address
    }
  } // namespace lds_dma
} // namespace adaflo_hip
