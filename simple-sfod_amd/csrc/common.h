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

// Entry-point hygiene (tests/test_abi.py fuzzes the host side under ASan / UBSan): every extent of a call is
// non-negative and small enough that the +255 / +63 roundings of the launch arithmetic cannot wrap an int (int64
// element / byte counts: below 2^40); products of extents that the host code forms are checked where they are formed
// (sfod_prod_fits) -- otherwise SFOD_EBADARG before anything is computed or launched.
#include <initializer_list>
static inline bool sfod_ints_ok(std::initializer_list<long long> dims) {
  for (long long d : dims)
    if (d < 0 || d > 2147483647LL - 65536) return false;
  return true;
}
static inline bool sfod_i64s_ok(std::initializer_list<long long> dims) {
  for (long long d : dims)
    if (d < 0 || d > (1LL << 40)) return false;
  return true;
}
// does the product of (already non-negative) extents stay <= limit?  saturating, no overflow on the way
static inline bool sfod_prod_fits(std::initializer_list<long long> dims, long long limit = 2147483647LL - 65536) {
  unsigned long long p = 1;
  bool over = false;
  for (long long d : dims) {
    if (d < 0) return false;
    if (d == 0) return true;                          // an empty tensor, whatever the other extents say
    if (over || p > (unsigned long long)limit / (unsigned long long)d) over = true;
    else p *= (unsigned long long)d;
  }
  return !over;
}
#define SFOD_REQUIRE_EXTENTS(what, ...) SFOD_REQUIRE(sfod_ints_ok({__VA_ARGS__}), what ": negative or oversized extent")

typedef __bf16 bf16_t;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---- SFOD_BF16X3 storage ("split" tensors, include/sfod_hip.h): a logical fp32 element is the pair
// hi = bf16(v), lo = bf16(v - hi); a group of 8 consecutive channels occupies 32 bytes = 8 hi then 8 lo.
// split_t is a 4-byte tag type: pointer arithmetic on split_t* counts LOGICAL elements, and a pointer to
// the first element of an 8-group addresses the group's 32 bytes.  hi + lo is exact in fp32.
struct split_t { uint32_t raw; };

__device__ __forceinline__ void split_load8(const split_t* p, float* out) {   // p: 32-byte aligned group
  const uint4 hv = reinterpret_cast<const uint4*>(p)[0], lv = reinterpret_cast<const uint4*>(p)[1];
  const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w}, lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(hw[i] << 16) + __uint_as_float(lw[i] << 16);
    out[2 * i + 1] = __uint_as_float(hw[i] & 0xffff0000u) + __uint_as_float(lw[i] & 0xffff0000u);
  }
}
// 16-byte store, optionally non-temporal (`global_store_dwordx4 ... nt`): a write-once output stream that nobody re-reads
// soon should not push re-used lines (a feature map gathered by many workgroups) out of L2
typedef uint32_t sfod_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void store16(void* p, uint4 v) {
  if constexpr (NT) {
    sfod_u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<sfod_u32x4*>(p));
  } else {
    *reinterpret_cast<uint4*>(p) = v;
  }
}
template <bool NT = false>
__device__ __forceinline__ void split_store8(split_t* p, const float* in) {
  union { bf16_t h[8]; uint4 v; } hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi.h[i] = (bf16_t)in[i];
    lo.h[i] = (bf16_t)(in[i] - (float)hi.h[i]);
  }
  store16<NT>(p, hi.v);
  store16<NT>(reinterpret_cast<uint4*>(p) + 1, lo.v);
}
// ---- SFOD_F16X3 storage: the same layout with IEEE half pairs, hi = f16(v), lo = f16(v - hi) (include/sfod_hip.h).
// |v| beyond the half range (infinities included) saturates at exactly +-65504, lo = 0 (NaN stays NaN).
typedef _Float16 f16_t;
struct splith_t { uint32_t raw; };

// Saturation is REPORTED: a finite value beyond half's range raises this word (one copy per translation unit that
// produces half pairs; sfod_f16x3_poll ORs them into the caller's device word and clears them), so that a model whose
// activations or scaled weights leave the window the mode assumes fails loudly at the trainer's next metrics flush
// instead of training on clamped values.
static __device__ unsigned g_f16_sat;
#define SFOD_DEFINE_F16_POLL(fn)                                                                            \
  static __global__ void fn##_k(unsigned* out) { if (atomicExch(&g_f16_sat, 0u)) atomicOr(out, 1u); }       \
  void fn(unsigned* out, hipStream_t s) { hipLaunchKernelGGL(fn##_k, dim3(1), dim3(1), 0, s, out); }

__device__ __forceinline__ void f16_pair(float v, f16_t& h, f16_t& l) {
  const float lim = 65504.f;
  const bool over = fabsf(v) > lim;                 // NaN compares false: passes through as NaN in both halves
  if (over) g_f16_sat = 1u;                         // finite or infinite: the value left half's range -- reported
  const float c = over ? copysignf(lim, v) : v;
  h = (f16_t)c;
  l = (f16_t)(c - (float)h);                        // residual of the CLAMPED value: a saturated element is exactly +-65504
}
__device__ __forceinline__ void split_load8(const splith_t* p, float* out) {
  union { uint4 v; f16_t h[8]; } hi, lo;
  hi.v = reinterpret_cast<const uint4*>(p)[0];
  lo.v = reinterpret_cast<const uint4*>(p)[1];
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)hi.h[i] + (float)lo.h[i];
}
template <bool NT = false>
__device__ __forceinline__ void split_store8(splith_t* p, const float* in) {
  union { f16_t h[8]; uint4 v; } hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) f16_pair(in[i], hi.h[i], lo.h[i]);
  store16<NT>(p, hi.v);
  store16<NT>(reinterpret_cast<uint4*>(p) + 1, lo.v);
}

// scalar access to logical element idx of a split tensor whose 8-groups start at `base`
__device__ __forceinline__ void split_put(split_t* base, int64_t idx, float v) {
  bf16_t* b = reinterpret_cast<bf16_t*>(base) + (idx >> 3) * 16 + (idx & 7);
  const bf16_t h = (bf16_t)v;
  b[0] = h;
  b[8] = (bf16_t)(v - (float)h);
}
__device__ __forceinline__ float split_get(const split_t* base, int64_t idx) {
  const bf16_t* b = reinterpret_cast<const bf16_t*>(base) + (idx >> 3) * 16 + (idx & 7);
  return (float)b[0] + (float)b[8];
}
__device__ __forceinline__ void split_put(splith_t* base, int64_t idx, float v) {
  f16_t* b = reinterpret_cast<f16_t*>(base) + (idx >> 3) * 16 + (idx & 7);
  f16_pair(v, b[0], b[8]);
}
__device__ __forceinline__ float split_get(const splith_t* base, int64_t idx) {
  const f16_t* b = reinterpret_cast<const f16_t*>(base) + (idx >> 3) * 16 + (idx & 7);
  return (float)b[0] + (float)b[8];
}
// out[idx] = v for any storage type (fp32 / bf16: plain element; split: the (hi, lo) pair)
template <typename T> __device__ __forceinline__ void put_elem(T* base, int64_t idx, float v) { base[idx] = from_f32<T>(v); }
template <> __device__ __forceinline__ void put_elem<split_t>(split_t* base, int64_t idx, float v) { split_put(base, idx, v); }
template <> __device__ __forceinline__ void put_elem<splith_t>(splith_t* base, int64_t idx, float v) { split_put(base, idx, v); }
template <typename T> __device__ __forceinline__ float get_elem(const T* base, int64_t idx) { return to_f32(base[idx]); }
template <> __device__ __forceinline__ float get_elem<split_t>(const split_t* base, int64_t idx) { return split_get(base, idx); }
template <> __device__ __forceinline__ float get_elem<splith_t>(const splith_t* base, int64_t idx) { return split_get(base, idx); }

// the split-precision product's MFMA on either pair format (FMT: 1 = bf16 pairs, 2 = f16 pairs; same registers, same rate)
typedef __attribute__((ext_vector_type(8))) __bf16 sfod_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 sfod_f16x8;
typedef __attribute__((ext_vector_type(16))) float sfod_f32x16;
template <int FMT>
__device__ __forceinline__ sfod_f32x16 mfma_pairs(sfod_bf16x8 a, sfod_bf16x8 b, sfod_f32x16 c) {
  if constexpr (FMT == 2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sfod_f16x8, a), __builtin_bit_cast(sfod_f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// SFOD_F16X3 weights are stored multiplied by a power of two s chosen per packed tensor so that max|w| * s lands in
// [2^13, 2^14): half's full 22-bit pair precision needs |v| >= 2^-3 (below, lo is a subnormal with absolute spacing 2^-24),
// and weights are uniformly small (kaiming: 0.02 ... 0.09).  The packer publishes the bit pattern of max|w| in a device
// word; the packing kernels derive s from it, the forward kernels derive 1 / s and apply it to their accumulators (exact:
// powers of two).  Zero / non-finite maxima leave the tensor unscaled.
__device__ __forceinline__ int wscale_biased_exp(unsigned absmax_bits) {
  const int e = (int)(absmax_bits >> 23) & 0xff;
  if (e == 0 || e == 255) return 127;
  const int se = 127 + 13 - (e - 127);
  return se < 8 ? 8 : (se > 246 ? 246 : se);
}
__device__ __forceinline__ float wscale_from_absmax(unsigned absmax_bits) {
  return __uint_as_float((unsigned)wscale_biased_exp(absmax_bits) << 23);
}
__device__ __forceinline__ float winv_from_absmax(unsigned absmax_bits) {
  return __uint_as_float((unsigned)(254 - wscale_biased_exp(absmax_bits)) << 23);
}
static inline bool sfod_is_pairs(int dt) { return dt == SFOD_BF16X3 || dt == SFOD_F16X3; }

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
