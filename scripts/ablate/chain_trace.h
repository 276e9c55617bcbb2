// chain_trace.h -- measurement build of csrc/sa_chain.hip only (scripts/wg_timeline.py): per-workgroup start / end stamps
// (100 MHz real-time counter) and placement (HW_ID, XCC_ID) of sa_chain_kernel, written to a buffer set through the debug
// export below.  Included by sa_chain.hip when built with -DCH_TRACE_H='"<this file>"'; never part of the product library.
#pragma once
__device__ unsigned long long* g_ch_trace = nullptr;
extern "C" int regnet_debug_set_chain_trace(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ch_trace), &buf, sizeof(buf));
}
#define CH_TRACE_BEGIN() const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime()
// behind the barrier that ends the set-up (weights staged, neighbourhood gathered): its distance from the start goes into bits 8.. of word 3
#define CH_TRACE_MID() const unsigned long long tr_tm = __builtin_amdgcn_s_memrealtime()
#define CH_TRACE_END()                                                                      \
  do {                                                                                      \
    if (g_ch_trace && threadIdx.x == 0) {                                                   \
      unsigned long long* t = g_ch_trace + (long long)blockIdx.x * 4;                       \
      t[0] = tr_t0; t[1] = __builtin_amdgcn_s_memrealtime();                                \
      t[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    /* HW_ID */                      \
      t[3] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) | ((tr_tm - tr_t0) << 8);   /* XCC_ID | set-up ticks */                     \
    }                                                                                       \
  } while (0)
