// Stable descending segmented sort (scores -> ranking) used for the RPN pre-NMS top-k
// (d2 find_top_rpn_proposals) and the Fast R-CNN inference candidates (A.13).
// Radix sort is stable, so equal scores keep ascending original index: the tie rule the
// oracle defines (torch.sort(stable=True, descending=True)).
// Segments of up to 16384 keys (the hot path: 9990 / 12000 anchors, 16000 ROI candidates) are sorted
// by one workgroup each with a bitonic network over 64-bit (score, index) keys held in LDS; larger
// segments fall back to rocPRIM's device-wide segmented radix sort (via hipCUB).
#include <hipcub/hipcub.hpp>

#include "common.h"

static inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

__global__ void k_sort_setup(int32_t* offsets, int32_t* idx_in, int B, int n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= B) offsets[i] = (int32_t)(i * n);
  if (i < (int64_t)B * n) idx_in[i] = (int32_t)(i % n);
}

static size_t cub_temp_bytes(int B, int n) {
  size_t temp = 0;
  hipcub::DeviceSegmentedRadixSort::SortPairsDescending(
      nullptr, temp, (const float*)nullptr, (float*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
      B * n, B, (const int32_t*)nullptr, (const int32_t*)nullptr, 0, 32, (hipStream_t)0);
  return temp;
}

// ---- one-workgroup bitonic sort --------------------------------------------------------------------
// key = (~orderable(score)) << 32 | index: ascending u64 order == descending score, ascending index on
// ties (torch.sort(stable=True, descending=True)); -0.0 is canonicalised to +0.0 like torch's compare.
#define BITONIC_MAX 16384

__device__ __forceinline__ uint32_t f32_orderable(float f) {
  f = f + 0.0f;
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(1024)
k_bitonic_sort_desc(const float* __restrict__ keys, int n, int npow2, float* __restrict__ out_keys,
                    int32_t* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* src = keys + (int64_t)b * n;
  for (int i = tid; i < npow2; i += 1024)
    sk[i] = (i < n) ? (((unsigned long long)(~f32_orderable(src[i])) << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (npow2 >> 1); t += 1024) {
        // element pair (i, i ^ j) with i the one whose bit j is clear
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const unsigned long long a = sk[i], c = sk[p];
        const bool up = (i & k) == 0;
        if ((a > c) == up) { sk[i] = c; sk[p] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += 1024) {
    const unsigned long long v = sk[i];
    const int idx = (int)(v & 0xffffffffull);
    out_idx[(int64_t)b * n + i] = idx;
    out_keys[(int64_t)b * n + i] = src[idx];
  }
}

extern "C" int64_t sfod_sort_ws_bytes(int B, int n) {
  return align256(sizeof(int32_t) * (B + 1)) + align256(sizeof(int32_t) * (int64_t)B * n) +
         align256((int64_t)cub_temp_bytes(B, n)) + 256;
}

extern "C" int sfod_segmented_sort_desc(const float* keys, int B, int n, float* out_keys, int32_t* out_idx,
                                        void* ws, int64_t ws_bytes, void* stream) {
  SFOD_REQUIRE(B >= 1 && n >= 1, "sort sizes");
  SFOD_REQUIRE(ws_bytes >= sfod_sort_ws_bytes(B, n), "sort workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (n <= BITONIC_MAX) {
    int npow2 = 2;
    while (npow2 < n) npow2 <<= 1;
    const int lds = npow2 * 8;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bitonic_sort_desc),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, BITONIC_MAX * 8);
      if (e != hipSuccess) { sfod_set_error("hipFuncSetAttribute(sort): %s", hipGetErrorString(e)); return -(int)e; }
      attr_set = true;
    }
    hipLaunchKernelGGL(k_bitonic_sort_desc, dim3(B), dim3(1024), lds, s, keys, n, npow2, out_keys, out_idx);
    return sfod_check_launch("bitonic_sort");
  }
  char* p = reinterpret_cast<char*>(ws);
  int32_t* offsets = reinterpret_cast<int32_t*>(p);
  p += align256(sizeof(int32_t) * (B + 1));
  int32_t* idx_in = reinterpret_cast<int32_t*>(p);
  p += align256(sizeof(int32_t) * (int64_t)B * n);
  size_t temp = cub_temp_bytes(B, n);
  const int64_t tot = (int64_t)B * n + 1;
  hipLaunchKernelGGL(k_sort_setup, dim3(cdiv(tot, 256)), dim3(256), 0, s, offsets, idx_in, B, n);
  int rc = sfod_check_launch("sort_setup");
  if (rc) return rc;
  hipError_t e = hipcub::DeviceSegmentedRadixSort::SortPairsDescending(
      (void*)p, temp, keys, out_keys, (const int32_t*)idx_in, out_idx, B * n, B, offsets, offsets + 1, 0,
      32, s);
  if (e != hipSuccess) {
    sfod_set_error("segmented sort: %s", hipGetErrorString(e));
    return -(int)e;
  }
  return 0;
}
