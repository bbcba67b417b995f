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
  __device__ __forceinline__ void opaque(double &v) { asm volatile("" : "+v"(v)); } // (also: "this load has arrived")

  __device__ __forceinline__ void opaque_ptr(const double *&p) { asm volatile("" : "+v"(p)); }
  __device__ __forceinline__ void opaque_s(int &v) { asm volatile("" : "+s"(v)); } // (a wave-uniform value: stays scalar)

  // a cell = a DPP row of 16 lanes (ns_hop_kernel): the value the same lane position holds one row below (rows 1, 3
  // receive rows 0, 2; the even rows get their own value back) and one half-wave below (lanes 32..63 receive 0..31) --
  // gfx950's v_permlane16_swap / v_permlane32_swap, two per double, no LDS
  __device__ __forceinline__ double from_row_below(const double v)
  {
    const unsigned long long b  = (unsigned long long)__double_as_longlong(v);
    const unsigned           lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto               r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto               r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0]));
  }
  __device__ __forceinline__ double from_half_below(const double v)
  {
    const unsigned long long b  = (unsigned long long)__double_as_longlong(v);
    const unsigned           lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const auto               r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto               r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0]));
  }

  // the offset becomes known only after `v` has been computed: pins a prefetch behind the arithmetic that frees its
  // destination registers (otherwise all loads of an unrolled loop are hoisted to its top and stay live together).
  // (An offset, not the pointer: behind an opaque pointer the compiler no longer knows the address space.)
  __device__ __forceinline__ void pin_after(unsigned &off, const double v) { asm volatile("" : "+v"(off) : "v"(v)); }

  // diagnostic builds (-DHOX_STAMP): shader clock
  __device__ __forceinline__ unsigned long long clock_now() { return __builtin_amdgcn_s_memtime(); }
  // diagnostic builds (-DHOX_EXP=...): keeps a value alive without storing it anywhere
  __device__ __forceinline__ void sink(const double v) { asm volatile("" : : "v"(v)); }

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

  // the same without the scheduling barrier (ns_hop_kernel: the exchanges of consecutive quadrature points may overlap);
  // between an LDS write and the reads of OTHER lanes of the wave that depend on it
  __device__ __forceinline__ void wave_fence()
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // before an LDS write to a location other lanes of the wave may still have to READ: nothing on the hardware (the LDS
  // executes one wave's instructions in order and the compiler keeps may-alias accesses in order); the host emulator,
  // which runs the lanes one after the other, meets here
  __device__ __forceinline__ void emu_sync() {}

  // workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load and store of
  // the wave (vmcnt(0)) -- that would drain the prefetches in flight at every step of the marching loop
  __device__ __forceinline__ void lds_barrier()
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);
  }

  // LDS-DMA: every active lane copies 16 bytes from sbase + voff to LDS byte lds_byte + 16 * lane (sbase, lds_byte
  // wave-uniform; `nt`: a once-read stream).  Asynchronous and invisible to the compiler's vmcnt bookkeeping: the
  // consumer waits with wait_vmcnt<N>() (VMEM operations of a wave retire in issue order: "at most N outstanding"
  // means everything older than the N youngest has landed) before it reads the slot with ds_rd128
  __device__ __forceinline__ void dma_b128(const double *sbase, const unsigned voff, const unsigned lds_byte)
  {
    // wait states the compiler cannot insert into the statement: the scalar base may be fresh from v_readfirstlane
    // (uniform_ptr; a VMEM instruction must be five wait states behind a VALU write of a scalar register it reads), and an
    // LDS-DMA must be one wait state behind the SALU write of M0 (cdna_hip_programming.md, "What hipcc does not do" 2)
    asm volatile("s_nop 3\n\t"
                 "s_mov_b32 m0, %0\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 nt" ::"s"(lds_byte),
                 "v"(voff), "s"(sbase)
                 : "memory"); // (m0 cannot be named as a clobber; on gfx950 the compiler itself loads M0 right before
                              // each of the few instructions that read it and keeps nothing in it)
  }
  template <int N>
  __device__ __forceinline__ void wait_vmcnt()
  {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  }
  // a pointer the compiler must keep in scalar registers (operand of dma_b128)
  __device__ __forceinline__ const double *uniform_ptr(const double *p)
  {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned           lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v),
                   hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const double *>(((unsigned long long)hi << 32) | lo);
  }
  // 16 bytes of LDS -> two doubles; usable after lds_arrived() on the pair
  typedef double hox_double2 __attribute__((ext_vector_type(2)));
  template <int OFF>
  __device__ __forceinline__ void ds_rd128(const unsigned a, hox_double2 &v)
  {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
  }
  // all LDS reads of the wave have returned; the values listed become usable
  __device__ __forceinline__ void lds_arrived(hox_double2 &a, hox_double2 &b)
  {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
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
  // point-to-point hand-off between the waves of a workgroup through a 32-bit flag in LDS (ns_hox_kernel, HOX_FLAGS):
  // the LDS pipeline serves the requests of a CU in order, so data written before the flag is visible to a wave that has
  // read the flag
  __device__ __forceinline__ void lds_flag_set(const unsigned byte_addr, const int value)
  {
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" : : "v"(byte_addr), "v"(value) : "memory");
  }
  __device__ __forceinline__ void lds_flag_wait(const unsigned byte_addr, const int target)
  {
    int v;
    do
      {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
        v = __builtin_amdgcn_readfirstlane(v);
        if (v < target)
          __builtin_amdgcn_s_sleep(1);
      }
    while (v < target);
  }
  template <int CNT>
  __device__ __forceinline__ void ds_wait1(double &x)
  {
    constexpr int C = CNT > 15 ? 15 : CNT;
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(C));
  }
} // namespace adaflo_hip
