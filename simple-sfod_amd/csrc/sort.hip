// Stable descending segmented sort (scores -> ranking) used for the RPN pre-NMS top-k
// (d2 find_top_rpn_proposals) and the Fast R-CNN inference candidates (A.13).
// Every key is the 64-bit pair (~orderable(score), original index): ascending u64 order == descending score with
// ascending original index on ties -- the rule the oracle defines (torch.sort(stable=True, descending=True)) -- and
// all keys are distinct, so any comparison sort gives the one stable result.
// Segments of up to 4096 keys are sorted by one workgroup each with a bitonic network over the keys held in LDS.
// Larger segments are cut into chunks, each chunk is sorted by one workgroup with the same network, and the sorted
// chunks are merged pairwise by merge-path passes (every thread finds its diagonal's split by binary search and merges
// 8 outputs): 1 + ceil(log2(chunks)) + 1 launches.  Chunk size: 4096 keys for segments up to 131072 (the hot path at
// 600x1200: 9990 / 12000 anchors, 16000 ROI candidates -- B = 8 segments as 32 workgroups of 78 network passes + 2
// merge passes instead of 8 workgroups of 105 passes over 4x the keys per thread: 156 -> 63 us; 1024x2048 frames:
// 30 720 anchors 173 -> 74 us; ResNet-C4: 34 200 / 98 304 keys 185 -> 85 / 212 -> 125 us), 16384-key chunks beyond.
#include "common.h"

static inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

// -0.0 is canonicalised to +0.0 like torch's compare
__device__ __forceinline__ uint32_t f32_orderable(float f) {
  f = f + 0.0f;
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

#define BITONIC_MAX 16384          // LDS capacity of one workgroup's network (keys)
#define SINGLE_MAX 4096            // segments up to this size: one workgroup, one launch
#define SORT_CHUNK_BIG 16384
#define SORT_CHUNK_SMALL 4096
static inline int sort_chunk_of(int n) { return n <= 131072 ? SORT_CHUNK_SMALL : SORT_CHUNK_BIG; }
#define MERGE_ITEMS 8
#define MERGE_THREADS 256

// the bitonic network over npow2 keys in LDS (1024 threads)
__device__ __forceinline__ void bitonic_network(unsigned long long* sk, int npow2, int tid) {
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
}

// ---- one workgroup per segment (n <= 16384) ----------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_bitonic_sort_desc(const float* __restrict__ keys, int n, int npow2, float* __restrict__ out_keys,
                    int32_t* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* src = keys + (int64_t)b * n;
  for (int i = tid; i < npow2; i += 1024)
    sk[i] = (i < n) ? (((unsigned long long)(~f32_orderable(src[i])) << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  bitonic_network(sk, npow2, tid);
  for (int i = tid; i < n; i += 1024) {
    const unsigned long long v = sk[i];
    const int idx = (int)(v & 0xffffffffull);
    out_idx[(int64_t)b * n + i] = idx;
    out_keys[(int64_t)b * n + i] = src[idx];
  }
}

// ---- larger segments: sorted chunks + merge-path passes ----------------------------------------------------------
// chunk c of segment b: its <= CH keys sorted ascending into dst[b][c * CH ..]; positions beyond n
// hold ~0 (they sort last and are never emitted)
__global__ void __launch_bounds__(1024)
k_bitonic_chunk(const float* __restrict__ keys, int n, int nchunks, int CH, unsigned long long* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const int b = blockIdx.x / nchunks, c = blockIdx.x % nchunks, tid = threadIdx.x;
  const float* src = keys + (int64_t)b * n;
  const int base = c * CH;
  for (int i = tid; i < CH; i += 1024) {
    const int g = base + i;
    sk[i] = (g < n) ? (((unsigned long long)(~f32_orderable(src[g])) << 32) | (unsigned)g) : ~0ull;
  }
  __syncthreads();
  bitonic_network(sk, CH, tid);
  unsigned long long* out = dst + ((int64_t)b * nchunks + c) * CH;
  for (int i = tid; i < CH; i += 1024) out[i] = sk[i];
}

// one merge pass over the sorted runs of length L (a multiple of the chunk size) inside every segment of
// NP = nchunks * chunk keys: runs (2r, 2r+1) -> one run of length 2L; a last run without partner is copied
__global__ void __launch_bounds__(MERGE_THREADS)
k_merge_pass(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int NP, int L) {
  const int b = blockIdx.y;
  const unsigned long long* S = src + (int64_t)b * NP;
  unsigned long long* D = dst + (int64_t)b * NP;
  const int o0 = (blockIdx.x * MERGE_THREADS + threadIdx.x) * MERGE_ITEMS;     // first output position of this thread
  if (o0 >= NP) return;
  const int pair = o0 / (2 * L);
  const int a0 = pair * 2 * L;                       // start of run A
  const int lenA = min(L, NP - a0);
  const int b0 = a0 + lenA;
  const int lenB = max(0, min(L, NP - b0));
  const unsigned long long* A = S + a0;
  const unsigned long long* Bp = S + b0;
  const int d = o0 - a0;                             // diagonal: outputs [d, d + MERGE_ITEMS) of this pair
  // merge path: the largest i in [lo, hi] with A[i-1] < B[d-i] (keys are distinct: strict order)
  int lo = max(0, d - lenB), hi = min(d, lenA);
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (A[mid - 1] < Bp[d - mid]) lo = mid; else hi = mid - 1;
  }
  int i = lo, j = d - lo;
#pragma unroll
  for (int k = 0; k < MERGE_ITEMS; ++k) {
    const int o = d + k;
    if (o >= lenA + lenB) break;
    const bool takeA = (j >= lenB) || (i < lenA && A[i] < Bp[j]);
    D[a0 + o] = takeA ? A[i] : Bp[j];
    i += takeA ? 1 : 0;
    j += takeA ? 0 : 1;
  }
}

__global__ void __launch_bounds__(256)
k_sort_emit(const unsigned long long* __restrict__ sorted, const float* __restrict__ keys, int n, int NP,
            float* __restrict__ out_keys, int32_t* __restrict__ out_idx) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int idx = (int)(sorted[(int64_t)b * NP + i] & 0xffffffffull);
  out_idx[(int64_t)b * n + i] = idx;
  out_keys[(int64_t)b * n + i] = keys[(int64_t)b * n + idx];
}

extern "C" int64_t sfod_sort_ws_bytes(int B, int n) {
  if (!sfod_ints_ok({B, n}) || !sfod_prod_fits({B, n, 32}, 1LL << 40)) return 0;      // hostile extents: not served / nothing
  if (n <= SINGLE_MAX) return 256;
  const int ch = sort_chunk_of(n);
  const int64_t NP = (int64_t)((n + ch - 1) / ch) * ch;
  return align256(2 * (int64_t)B * NP * 8) + 256;       // two key buffers (ping-pong of the merge passes)
}

static int sort_set_lds_attr(const void* kern) {
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, BITONIC_MAX * 8);
  if (e != hipSuccess) { sfod_set_error("hipFuncSetAttribute(sort): %s", hipGetErrorString(e)); return -(int)e; }
  return 0;
}

extern "C" int sfod_segmented_sort_desc(const float* keys, int B, int n, float* out_keys, int32_t* out_idx,
                                        void* ws, int64_t ws_bytes, void* stream) {
  SFOD_REQUIRE(sfod_ints_ok({B, n}) && sfod_i64s_ok({B, n, ws_bytes}), "segmented_sort_desc: negative or oversized extent");
  SFOD_REQUIRE(B >= 1 && n >= 1, "sort sizes");
  SFOD_REQUIRE(keys != nullptr && out_keys != nullptr && out_idx != nullptr && ws != nullptr, "sort: null argument");
  SFOD_REQUIRE(ws_bytes >= sfod_sort_ws_bytes(B, n), "sort workspace too small");
  hipStream_t s = (hipStream_t)stream;
  // (idempotent attribute calls; repeating them from two threads is harmless)
  static const int attr_rc = []() {
    int rc = sort_set_lds_attr(reinterpret_cast<const void*>(k_bitonic_sort_desc));
    return rc ? rc : sort_set_lds_attr(reinterpret_cast<const void*>(k_bitonic_chunk));
  }();
  if (attr_rc) return attr_rc;
  if (n <= SINGLE_MAX) {
    int npow2 = 2;
    while (npow2 < n) npow2 <<= 1;
    hipLaunchKernelGGL(k_bitonic_sort_desc, dim3(B), dim3(1024), npow2 * 8, s, keys, n, npow2, out_keys, out_idx);
    return sfod_check_launch("bitonic_sort");
  }
  SFOD_REQUIRE(ws != nullptr, "sort: workspace required above 4096 keys per segment");
  const int CH = sort_chunk_of(n);
  const int nchunks = (n + CH - 1) / CH;
  const int NP = nchunks * CH;
  unsigned long long* src = reinterpret_cast<unsigned long long*>(ws);
  unsigned long long* dst = src + (int64_t)B * NP;
  hipLaunchKernelGGL(k_bitonic_chunk, dim3(B * nchunks), dim3(1024), CH * 8, s, keys, n, nchunks, CH, src);
  int rc = sfod_check_launch("bitonic_chunk");
  if (rc) return rc;
  const dim3 mgrid(cdiv(NP, MERGE_THREADS * MERGE_ITEMS), B);
  for (int L = CH; L < NP; L *= 2) {
    hipLaunchKernelGGL(k_merge_pass, mgrid, dim3(MERGE_THREADS), 0, s, (const unsigned long long*)src, dst, NP, L);
    rc = sfod_check_launch("merge_pass");
    if (rc) return rc;
    unsigned long long* t = src;
    src = dst;
    dst = t;
  }
  hipLaunchKernelGGL(k_sort_emit, dim3(cdiv(n, 256), B), dim3(256), 0, s, (const unsigned long long*)src, keys, n, NP,
                     out_keys, out_idx);
  return sfod_check_launch("sort_emit");
}
