// A load the compiler must not see (stencil kernels of q1_sweep.hip).
//
// The hand-issued asynchronous global -> LDS copies (LDS-DMA of gfx950) of the streaming kernels are invisible to the
// compiler's s_waitcnt bookkeeping; loads the compiler CAN see must not sit in the same loop: it would answer them
// with s_waitcnt vmcnt(0) at the next join and empty the queue in every iteration -- rare ones go
// through load_now().  The DMA helpers themselves live next to their kernels: ns_q2.hip (dma_b128 / dma_b32) and
// hox_intrin.hpp (dma_b128).  They write M0 inside one asm statement (s_mov_b32 m0 + global_load_lds) without naming
// it as clobbered: clang rejects "m0" in a clobber list as a reserved register; the compiler itself only ever sets M0
// immediately before an instruction of its own that reads it (no such instruction -- movrel, sendmsg, LDS parameter
// loads, GWS -- is generated for these kernels; checked in the ISA of every round).
#pragma once
#include <hip/hip_runtime.h>

namespace adaflo_hip
{
  namespace lds_dma
  {
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
  } // namespace lds_dma
} // namespace adaflo_hip
