// Building blocks of the PERSISTENT, software-pipelined row kernels of the unwrap (gpa_unwrap_rowpers.hip,
// gpa_unwrap_rowhalfpers.hip): LDS-DMA of the next row(s) while the current ones are transformed.
#pragma once
#include "gpa_unwrap_impl.h"

namespace gpa {
namespace {

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
// one LDS-DMA wave-instruction: lane l copies 16 bytes from its own `gsrc` to LDS byte lds_dst + 16 l (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// the same with the global address as a wave-uniform base (scalar register pair) + a 32-bit lane offset: no 64-bit address
// arithmetic in vector registers (the 16384-point kernels have none to spare)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
// opaque use of a register: whatever load produces it has been waited for when this returns
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }
// workgroup barrier that leaves vector-memory operations (the DMA, the stores) in flight
#define GPA_PBAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// resident workgroups per CU x CUs of the CURRENT device
inline int pers_grid(int per_cu) { return per_cu * device_cus(); }

}  // namespace
}  // namespace gpa
