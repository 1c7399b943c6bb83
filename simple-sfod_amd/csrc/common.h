// Shared helpers for libsfod_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sfod_hip.h"

#define SFOD_EBADARG (-1000)

void sfod_set_error(const char* fmt, ...);

static inline int sfod_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sfod_set_error("%s: %s", what, hipGetErrorString(e));
    return -(int)e;
  }
  return 0;
}

#define SFOD_REQUIRE(cond, msg)                      \
  do {                                               \
    if (!(cond)) {                                   \
      sfod_set_error("bad argument: %s", msg);       \
      return SFOD_EBADARG;                           \
    }                                                \
  } while (0)

typedef __bf16 bf16_t;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
