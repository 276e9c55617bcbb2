// Shared helpers for libregnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/regnet_hip.h"

#define REGNET_LAUNCH_CHECK()                      \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Canonical squared distance: three individually rounded products summed left to right,
// ((dx*dx)+(dy*dy))+(dz*dz).  The translation unit using it MUST be built with
// -ffp-contract=off (see csrc/build.py); the oracle (oracle/pn2_oracle.c) does the same.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  float s = xx + yy;
  return s + zz;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
