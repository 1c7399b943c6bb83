// Stable descending segmented sort (scores -> ranking) used for the RPN pre-NMS top-k
// (d2 find_top_rpn_proposals) and the Fast R-CNN inference candidates (A.13).
// Radix sort is stable, so equal scores keep ascending original index: the tie rule the
// oracle defines (torch.sort(stable=True, descending=True)).
// The device-wide radix sort itself is rocPRIM's (via hipCUB); everything around it is ours.
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

extern "C" int64_t sfod_sort_ws_bytes(int B, int n) {
  return align256(sizeof(int32_t) * (B + 1)) + align256(sizeof(int32_t) * (int64_t)B * n) +
         align256((int64_t)cub_temp_bytes(B, n)) + 256;
}

extern "C" int sfod_segmented_sort_desc(const float* keys, int B, int n, float* out_keys, int32_t* out_idx,
                                        void* ws, int64_t ws_bytes, void* stream) {
  SFOD_REQUIRE(B >= 1 && n >= 1, "sort sizes");
  SFOD_REQUIRE(ws_bytes >= sfod_sort_ws_bytes(B, n), "sort workspace too small");
  hipStream_t s = (hipStream_t)stream;
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
