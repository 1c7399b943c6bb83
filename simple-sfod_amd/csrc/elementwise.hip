// HBM-bound kernels of the hot path: preprocess, train-mode BatchNorm + ReLU + max-pool
// (forward and backward), weight (re)packing, bias gradients, fused SGD + EMA.  gfx950.
//
// Activations are NHWC; every kernel reads/writes 16 bytes per lane along the channel axis
// (4 x fp32 or 8 x bf16) so a wavefront covers 1 KiB of contiguous channels per instruction.
#include "conv_internal.h"
#include <type_traits>
#include <atomic>
#include <stdlib.h>

template <typename T> struct VecT;
template <> struct VecT<float> {
  static constexpr int N = 4;
  typedef float4 raw;
};
template <> struct VecT<bf16_t> {
  static constexpr int N = 8;
  typedef uint4 raw;
};

template <> struct VecT<split_t> {
  static constexpr int N = 8;
  typedef uint4 raw;
};
template <> struct VecT<splith_t> {
  static constexpr int N = 8;
  typedef uint4 raw;
};

template <typename T> __device__ __forceinline__ void load_vec(const T* p, float* out);
template <> __device__ __forceinline__ void load_vec<float>(const float* p, float* out) {
  float4 v = *reinterpret_cast<const float4*>(p);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <> __device__ __forceinline__ void load_vec<bf16_t>(const bf16_t* p, float* out) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(w[i] << 16);
    out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void load_vec<split_t>(const split_t* p, float* out) { split_load8(p, out); }
template <> __device__ __forceinline__ void load_vec<splith_t>(const splith_t* p, float* out) { split_load8(p, out); }
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float* in);
template <> __device__ __forceinline__ void store_vec<split_t>(split_t* p, const float* in) { split_store8(p, in); }
template <> __device__ __forceinline__ void store_vec<splith_t>(splith_t* p, const float* in) { split_store8(p, in); }
template <> __device__ __forceinline__ void store_vec<float>(float* p, const float* in) {
  *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
}
template <> __device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float* in) {
  union { bf16_t h[8]; uint4 v; } u;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.h[i] = (bf16_t)in[i];
  *reinterpret_cast<uint4*>(p) = u.v;
}

// V-element forms (V a multiple of the type's native vector width): a kernel that reads fp32 and writes
// SFOD_BF16X3 pairs handles 8 channels per thread (two float4 in, one 32-byte group out)
template <typename T, int V> __device__ __forceinline__ void load_n(const T* p, float* out) {
  constexpr int N = VecT<T>::N;
  static_assert(V % N == 0, "vector width");
#pragma unroll
  for (int k = 0; k < V / N; ++k) load_vec<T>(p + k * N, out + k * N);
}
template <typename T, int V> __device__ __forceinline__ void store_n(T* p, const float* in) {
  constexpr int N = VecT<T>::N;
  static_assert(V % N == 0, "vector width");
#pragma unroll
  for (int k = 0; k < V / N; ++k) store_vec<T>(p + k * N, in + k * N);
}
template <typename TI, typename TO> struct VecW {
  static constexpr int N = VecT<TI>::N > VecT<TO>::N ? VecT<TI>::N : VecT<TO>::N;
};

// ---------------------------------------------------------------------------------------------
// K1 preprocess
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_preprocess(const uint8_t* const* __restrict__ imgs, const int32_t* __restrict__ sizes,
                             int Hp, int Wp, int Cpad, float m0, float m1, float m2, float s0, float s1,
                             float s2, T* __restrict__ out) {
  const int b = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= Wp) return;
  const int h = sizes[b * 2], w = sizes[b * 2 + 1];
  T* o = out + (((int64_t)b * Hp + y) * Wp + x) * Cpad;
  float v[3] = {0.f, 0.f, 0.f};
  if (y < h && x < w) {
    const uint8_t* im = imgs[b];
    const int64_t plane = (int64_t)h * w;
    v[0] = ((float)im[(int64_t)y * w + x] - m0) / s0;
    v[1] = ((float)im[plane + (int64_t)y * w + x] - m1) / s1;
    v[2] = ((float)im[2 * plane + (int64_t)y * w + x] - m2) / s2;
  }
  constexpr int N = VecT<T>::N;
  if (Cpad == N) {      // the usual case (one 16-byte chunk / one 8-channel pair group per pixel): one or two 16-byte stores
    float f[N];
#pragma unroll
    for (int c = 0; c < N; ++c) f[c] = c < 3 ? v[c] : 0.f;
    store_vec<T>(o, f);
    return;
  }
  for (int c = 0; c < Cpad; ++c) put_elem<T>(o, c, c < 3 ? v[c] : 0.f);
}

extern "C" int sfod_preprocess(const void* const* img_ptrs, const int32_t* sizes, int B, int Hp, int Wp,
                               int Cpad, const float* mean3, const float* std3, void* out, int dt,
                               void* stream) {
  SFOD_REQUIRE_EXTENTS("preprocess", B, Hp, Wp, Cpad);
  SFOD_REQUIRE(Cpad >= 3, "Cpad < 3");
  dim3 grid(cdiv(Wp, 256), Hp, B);
  hipStream_t s = (hipStream_t)stream;
  if (sfod_is_pairs(dt)) SFOD_REQUIRE(Cpad % 8 == 0, "preprocess: operand-pair output needs Cpad % 8 == 0");
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_preprocess<float>, grid, dim3(256), 0, s, (const uint8_t* const*)img_ptrs, sizes,
                       Hp, Wp, Cpad, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (float*)out);
  else if (dt == SFOD_BF16X3)
    hipLaunchKernelGGL(k_preprocess<split_t>, grid, dim3(256), 0, s, (const uint8_t* const*)img_ptrs, sizes,
                       Hp, Wp, Cpad, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (split_t*)out);
  else if (dt == SFOD_F16X3)
    hipLaunchKernelGGL(k_preprocess<splith_t>, grid, dim3(256), 0, s, (const uint8_t* const*)img_ptrs, sizes,
                       Hp, Wp, Cpad, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (splith_t*)out);
  else
    hipLaunchKernelGGL(k_preprocess<bf16_t>, grid, dim3(256), 0, s, (const uint8_t* const*)img_ptrs, sizes,
                       Hp, Wp, Cpad, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (bf16_t*)out);
  return sfod_check_launch("preprocess");
}

static inline int ew_grid(int64_t total);

// horizontal flip of a uint8 [C,H,W] image (d2 RandomFlip in the weak augmentation): 16 output
// bytes per thread, read as the mirrored 16 bytes and byte-reversed
__global__ void __launch_bounds__(256)
k_hflip_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int W) {
  const int chunks = (W + 15) / 16;
  const int64_t total = (int64_t)rows * chunks;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(t / chunks), c = (int)(t % chunks);
    const int x0 = c * 16;
    const uint8_t* s = src + (int64_t)r * W;
    uint8_t* d = dst + (int64_t)r * W;
    const int n = min(16, W - x0);
    uint8_t v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (i < n) ? s[W - 1 - (x0 + i)] : (uint8_t)0;
    if (n == 16 && ((W & 15) == 0)) {
      *reinterpret_cast<uint4*>(d + x0) = *reinterpret_cast<const uint4*>(v);
    } else {
      for (int i = 0; i < n; ++i) d[x0 + i] = v[i];
    }
  }
}

extern "C" int sfod_hflip_u8(const void* src, void* dst, int C, int H, int W, void* stream) {
  SFOD_REQUIRE_EXTENTS("hflip_u8", C, H, W);
  SFOD_REQUIRE(sfod_prod_fits({C, H}) && sfod_prod_fits({C, H, W}, 1LL << 40), "hflip_u8: oversized image");
  if ((int64_t)C * H * W == 0) return 0;
  const int64_t total = (int64_t)C * H * ((W + 15) / 16);
  hipLaunchKernelGGL(k_hflip_u8, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src,
                     (uint8_t*)dst, C * H, W);
  return sfod_check_launch("hflip_u8");
}

// ---------------------------------------------------------------------------------------------
// ResizeShortestEdge on uint8 frames, bit-exact with Pillow's ImagingResample (BILINEAR, 8 bits per
// channel): two separable passes with a support-scaled triangle filter, fixed-point coefficients
// (22 fractional bits) and a uint8 clip BETWEEN the passes.  Fused here: every output pixel recomputes
// the <= KS horizontally filtered values of its KS source rows (each rounded and clipped exactly like
// Pillow's temporary image), then filters them vertically.  The coefficient tables are built on the host
// (float64, same operation order as Pillow's precompute_coeffs / normalize_coeffs_8bpc).  flip != 0
// additionally mirrors the output (RandomFlip comes after the resize in the mapper's augmentation list).
// ---------------------------------------------------------------------------------------------
#define RS_PREC 22
__device__ __forceinline__ int rs_clip8(int v) {
  v >>= RS_PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// One thread per output pixel (all channels): the horizontal / vertical coefficients are loaded once per
// pixel, and a source row's <= 8 horizontal taps come in with ONE unaligned 8-byte load (one load per tap,
// coefficient and channel made the first version 55 loads per output byte: 77 us per 1024x2048 frame).
template <bool WIDE8>
__global__ void __launch_bounds__(256)
k_resize_bilinear_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int C, int H, int W,
                     int h, int w, const int32_t* __restrict__ hb, const int32_t* __restrict__ hk, int ksh,
                     const int32_t* __restrict__ vb, const int32_t* __restrict__ vk, int ksv, int flip) {
  const int64_t total = (int64_t)h * w;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(t % w);
    const int y = (int)(t / w);
    const int xmin = hb[2 * x], xn = hb[2 * x + 1];
    const int ymin = vb[2 * y], yn = vb[2 * y + 1];
    const int xo = flip ? (w - 1 - x) : x;
    if constexpr (WIDE8) {
      int kh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) kh[i] = (i < xn) ? hk[x * ksh + i] : 0;
      // the 8-byte window must stay inside the plane: shift it left at the right border
      const int64_t plane_bytes = (int64_t)H * W;
      if (C == 3 && ksv <= 6) {
        // the common shape (RGB frames, <= 2.5x downscale): all C x ksv row windows are fetched up front (rows beyond
        // this pixel's vertical support re-read its last row and carry weight 0) -- 18 independent 8-byte loads in
        // flight per thread instead of a chain of dependent ones (92 -> see profiles: us per 1024x2048 frame)
        int kv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) kv[j] = (j < yn) ? vk[y * ksv + j] : 0;
        unsigned long long v[3][6];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int jj = j < yn ? j : (yn > 0 ? yn - 1 : 0);
            const int64_t off = (int64_t)(ymin + jj) * W + xmin;
            const int64_t lim = (int64_t)(3 - c) * plane_bytes - 8;       // last legal start in the tensor, from this plane
            const int64_t o2 = off <= lim ? off : lim;
            unsigned long long t8;
            __builtin_memcpy(&t8, src + (int64_t)c * plane_bytes + o2, 8);
            v[c][j] = t8 >> ((int)(off - o2) * 8);                        // taps beyond the tensor's end carry weight 0
          }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          int acc = 1 << (RS_PREC - 1);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            int sh = 1 << (RS_PREC - 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) sh += (int)((v[c][j] >> (8 * i)) & 0xffull) * kh[i];
            acc += rs_clip8(sh) * kv[j];
          }
          dst[((int64_t)c * h + y) * w + xo] = (uint8_t)rs_clip8(acc);
        }
        continue;
      }
      for (int c = 0; c < C; ++c) {
        const uint8_t* plane = src + (int64_t)c * plane_bytes;
        int acc = 1 << (RS_PREC - 1);
        for (int j = 0; j < yn; ++j) {
          const int64_t off = (int64_t)(ymin + j) * W + xmin;
          const int64_t lim = (int64_t)C * plane_bytes - 8 - (int64_t)c * plane_bytes;   // last legal start in the tensor
          const int64_t o2 = off <= lim ? off : lim;
          const int sh8 = (int)(off - o2) * 8;
          unsigned long long v;
          __builtin_memcpy(&v, plane + o2, 8);
          v >>= sh8;                                      // taps beyond the tensor's end carry weight 0
          int sh = 1 << (RS_PREC - 1);
#pragma unroll
          for (int i = 0; i < 8; ++i) sh += (int)((v >> (8 * i)) & 0xffull) * kh[i];
          acc += rs_clip8(sh) * vk[y * ksv + j];
        }
        dst[((int64_t)c * h + y) * w + xo] = (uint8_t)rs_clip8(acc);
      }
    } else {
      for (int c = 0; c < C; ++c) {
        const uint8_t* plane = src + (int64_t)c * H * W;
        int acc = 1 << (RS_PREC - 1);
        for (int j = 0; j < yn; ++j) {
          const uint8_t* row = plane + (int64_t)(ymin + j) * W + xmin;
          int sh = 1 << (RS_PREC - 1);
          for (int i = 0; i < xn; ++i) sh += (int)row[i] * hk[x * ksh + i];
          acc += rs_clip8(sh) * vk[y * ksv + j];
        }
        dst[((int64_t)c * h + y) * w + xo] = (uint8_t)rs_clip8(acc);
      }
    }
  }
}

// Tiled form of the common shape (3 channels, <= 2.5x downscale): the two passes really are separate -- a workgroup owns a
// 16 x 64 output tile, phase A writes the horizontally filtered, rounded and clipped rows the tile needs (Pillow's temporary
// image, ~1.7 x 16 + 5 rows here) into LDS ONCE, phase B filters them vertically.  The per-pixel kernel above recomputes the
// horizontal pass of every source row for every output row that touches it: 2.8x the loads and 2.5x the multiply-adds.
// Same integer arithmetic per value, so the result stays bit-identical to Pillow.
#define RS_TH 16
#define RS_TW 64
#define RS_ROWS 48
__global__ void __launch_bounds__(256)
k_resize_bilinear_u8_tiled(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int h, int w,
                           const int32_t* __restrict__ hb, const int32_t* __restrict__ hk, int ksh,
                           const int32_t* __restrict__ vb, const int32_t* __restrict__ vk, int ksv, int flip,
                           int tiles_x) {
  __shared__ uint8_t tmp[3][RS_ROWS][RS_TW];
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * RS_TH, x0 = tx * RS_TW;
  const int y1 = min(y0 + RS_TH, h) - 1;
  const int rmin = vb[2 * y0];
  const int nrows = min(vb[2 * y1] + vb[2 * y1 + 1] - rmin, RS_ROWS);
  const int64_t plane_bytes = (int64_t)H * W;
  // ---- phase A: thread = (column of the tile, row group); the column's taps are loaded once --------------------------
  {
    const int xl = threadIdx.x & (RS_TW - 1), rg = threadIdx.x / RS_TW;      // 64 columns x 4 row groups
    const int x = x0 + xl;
    if (x < w) {
      const int xmin = hb[2 * x], xn = hb[2 * x + 1];
      int kh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) kh[i] = (i < xn) ? hk[x * ksh + i] : 0;
      for (int r = rg; r < nrows; r += 256 / RS_TW) {
        const int64_t off = (int64_t)(rmin + r) * W + xmin;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          // the 8-byte window must stay inside the tensor: shift it left at the very end (taps beyond carry weight 0)
          const int64_t lim = (int64_t)(3 - c) * plane_bytes - 8;
          const int64_t o2 = off <= lim ? off : lim;
          unsigned long long t8;
          __builtin_memcpy(&t8, src + (int64_t)c * plane_bytes + o2, 8);
          t8 >>= ((int)(off - o2) * 8);
          int sh = 1 << (RS_PREC - 1);
#pragma unroll
          for (int i = 0; i < 8; ++i) sh += (int)((t8 >> (8 * i)) & 0xffull) * kh[i];
          tmp[c][r][xl] = (uint8_t)rs_clip8(sh);
        }
      }
    }
  }
  __syncthreads();
  // ---- phase B: thread = output pixels (xl, yl), yl = row group + 4 k ------------------------------------------------------
  {
    const int xl = threadIdx.x & (RS_TW - 1), rg = threadIdx.x / RS_TW;
    const int x = x0 + xl;
    if (x < w) {
      const int xo = flip ? (w - 1 - x) : x;
      for (int yl = rg; yl < RS_TH; yl += 256 / RS_TW) {
        const int y = y0 + yl;
        if (y >= h) break;
        const int r0 = vb[2 * y] - rmin, yn = vb[2 * y + 1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          int acc = 1 << (RS_PREC - 1);
          for (int j = 0; j < yn; ++j) acc += (int)tmp[c][r0 + j][xl] * vk[y * ksv + j];
          dst[((int64_t)c * h + y) * w + xo] = (uint8_t)rs_clip8(acc);
        }
      }
    }
  }
}

extern "C" int sfod_resize_bilinear_u8(const void* src, void* dst, int C, int H, int W, int h, int w,
                                       const int32_t* hbounds, const int32_t* hcoef, int ksize_h,
                                       const int32_t* vbounds, const int32_t* vcoef, int ksize_v, int flip,
                                       void* stream) {
  SFOD_REQUIRE_EXTENTS("resize_bilinear_u8", C, H, W, h, w, ksize_h, ksize_v);
  SFOD_REQUIRE(sfod_prod_fits({C, H, W}, 1LL << 40) && sfod_prod_fits({C, h, w}, 1LL << 40) &&
               sfod_prod_fits({C, H, w}, 1LL << 40), "resize: oversized image");
  if ((int64_t)C * h * w == 0) return 0;
  SFOD_REQUIRE(ksize_h >= 1 && ksize_v >= 1 && H >= 1 && W >= 1, "resize: bad sizes");
  // rows a 16-row output tile needs: (its first and last rows' supports) <= 16 * H / h + ksize_v + 1
  static const int tiled_on = []() { const char* e = getenv("SFOD_RESIZE_TILED"); return e ? atoi(e) : 1; }();    // A/B hook
  const int64_t rows_needed = ((int64_t)RS_TH * H + h - 1) / h + ksize_v + 2;
  if (tiled_on && C == 3 && ksize_h <= 8 && ksize_v <= 8 && rows_needed <= RS_ROWS && (int64_t)C * H * W >= 8 &&
      sfod_prod_fits({(h + RS_TH - 1) / RS_TH, (w + RS_TW - 1) / RS_TW})) {
    const int tiles_x = (w + RS_TW - 1) / RS_TW, tiles_y = (h + RS_TH - 1) / RS_TH;
    hipLaunchKernelGGL(k_resize_bilinear_u8_tiled, dim3(tiles_x * tiles_y), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)src, (uint8_t*)dst, H, W, h, w, hbounds, hcoef, ksize_h, vbounds, vcoef, ksize_v, flip,
                       tiles_x);
  } else if (ksize_h <= 8 && (int64_t)C * H * W >= 8)
    hipLaunchKernelGGL(k_resize_bilinear_u8<true>, dim3(ew_grid((int64_t)h * w)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)src, (uint8_t*)dst, C, H, W, h, w, hbounds, hcoef, ksize_h, vbounds, vcoef,
                       ksize_v, flip);
  else
    hipLaunchKernelGGL(k_resize_bilinear_u8<false>, dim3(ew_grid((int64_t)h * w)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)src, (uint8_t*)dst, C, H, W, h, w, hbounds, hcoef, ksize_h, vbounds, vcoef,
                       ksize_v, flip);
  return sfod_check_launch("resize_bilinear_u8");
}

// ---------------------------------------------------------------------------------------------
// K3 BatchNorm statistics finalize.  stats[blk][0][c] = sum over the block's rows, stats[blk][1][c]
// = sum of squared deviations from the block mean, counts[blk] (behind the sums) = rows of the block
// (all written by the conv epilogue).  Combined in
// fp64:  M2 = sum_b M2_b + sum_b s_b^2/n_b - (sum_b s_b)^2 / M .  Two launches: BNF_SPLITS x C/64
// workgroups reduce slices of the block axis, one small kernel combines them.
// ---------------------------------------------------------------------------------------------
#define BNF_SPLITS 64

__global__ void __launch_bounds__(1024)
k_bn_partial(const float* __restrict__ stats, int nblocks, int C, double* __restrict__ part) {
  // 16 waves split the blocks of this slice; lanes are consecutive channels (coalesced rows)
  __shared__ double red[16][3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int per = (nblocks + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
  const float* counts = stats + (int64_t)nblocks * 2 * C;
  double a = 0.0, q = 0.0, m2 = 0.0;
  if (c < C)
    for (int blk = b0 + wave; blk < b1; blk += 16) {
      const double nb = (double)counts[blk];
      const double sb = (double)stats[((int64_t)blk * 2) * C + c];
      const double mb = (double)stats[((int64_t)blk * 2 + 1) * C + c];
      if (nb > 0.0) {
        a += sb;
        q += sb * sb / nb;
        m2 += mb;
      }
    }
  red[wave][0][lane] = a; red[wave][1][lane] = q; red[wave][2][lane] = m2;
  __syncthreads();
  if (wave < 3 && c < C) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][wave][lane];
    part[((int64_t)blockIdx.y * 3 + wave) * C + c] = v;
  }
}

__global__ void __launch_bounds__(256)
k_bn_final(const double* __restrict__ part, int nsplit, int M, int C, float* __restrict__ mean,
           float* __restrict__ invstd, float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
           float eps, int update_running, long long* __restrict__ nbt) {
  __shared__ double red[4][3][64];
  if (nbt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *nbt += update_running;   // BatchNorm's num_batches_tracked
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double a = 0.0, q = 0.0, m2 = 0.0;
  if (c < C)
    for (int s = wave; s < nsplit; s += 4) {
      a += part[((int64_t)s * 3 + 0) * C + c];
      q += part[((int64_t)s * 3 + 1) * C + c];
      m2 += part[((int64_t)s * 3 + 2) * C + c];
    }
  red[wave][0][lane] = a; red[wave][1][lane] = q; red[wave][2][lane] = m2;
  __syncthreads();
  if (wave != 0 || c >= C) return;
  a = red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
  q = red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
  m2 = red[0][2][lane] + red[1][2][lane] + red[2][2][lane] + red[3][2][lane];
  const double mu = a / (double)M;
  double tot = m2 + q - a * a / (double)M;
  if (tot < 0.0) tot = 0.0;
  const double var = tot / (double)M;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (update_running) {
    // update_running = k >= 1 momentum updates with the SAME batch statistics folded into this call (k forward passes
    // over one batch between two optimiser steps), each rounded to fp32 like k separate BatchNorm calls
    const double unbiased = (M > 1) ? tot / (double)(M - 1) : var;
    float rm = rmean[c], rv = rvar[c];
    for (int k = 0; k < update_running; ++k) {
      rm = (float)((1.0 - (double)momentum) * (double)rm + (double)momentum * mu);
      rv = (float)((1.0 - (double)momentum) * (double)rv + (double)momentum * unbiased);
    }
    rmean[c] = rm;
    rvar[c] = rv;
  }
}

// <= 64 statistics blocks (one slice: every layer of a one-frame-per-GPU step below the first stages, every live BatchNorm of
// ResNet-101-C4 at that size) in ONE launch: slice 0's sums in k_bn_partial's order (16 waves stride the blocks, combined
// w = 0 .. 15), then k_bn_final's arithmetic for nsplit = 1 -- bit-identical to the two launches
// (sfod_set_bn_finalize_fused(0) keeps them, for the A/B and the test), one 5 us launch less per BatchNorm.
__global__ void __launch_bounds__(1024)
k_bn_finalize_one(const float* __restrict__ stats, int nblocks, int M, int C, float* __restrict__ mean,
                  float* __restrict__ invstd, float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                  float eps, int update_running, long long* __restrict__ nbt) {
  __shared__ double red[16][3][64];
  if (nbt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *nbt += update_running;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const float* counts = stats + (int64_t)nblocks * 2 * C;
  double a = 0.0, q = 0.0, m2 = 0.0;
  if (c < C)
    for (int blk = wave; blk < nblocks; blk += 16) {
      const double nb = (double)counts[blk];
      const double sb = (double)stats[((int64_t)blk * 2) * C + c];
      const double mb = (double)stats[((int64_t)blk * 2 + 1) * C + c];
      if (nb > 0.0) {
        a += sb;
        q += sb * sb / nb;
        m2 += mb;
      }
    }
  red[wave][0][lane] = a; red[wave][1][lane] = q; red[wave][2][lane] = m2;
  __syncthreads();
  if (wave != 0 || c >= C) return;
  double p[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][k][lane];
    p[k] = v;                       // = part[(0 * 3 + k) * C + c] of the two-launch form
  }
  // k_bn_final, nsplit = 1: wave 0 adds part[0] to 0.0, waves 1 .. 3 hold 0.0, summed ((w0 + w1) + w2) + w3
  a = (0.0 + p[0]) + 0.0 + 0.0 + 0.0;
  q = (0.0 + p[1]) + 0.0 + 0.0 + 0.0;
  m2 = (0.0 + p[2]) + 0.0 + 0.0 + 0.0;
  const double mu = a / (double)M;
  double tot = m2 + q - a * a / (double)M;
  if (tot < 0.0) tot = 0.0;
  const double var = tot / (double)M;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (update_running) {
    const double unbiased = (M > 1) ? tot / (double)(M - 1) : var;
    float rm = rmean[c], rv = rvar[c];
    for (int k = 0; k < update_running; ++k) {
      rm = (float)((1.0 - (double)momentum) * (double)rm + (double)momentum * mu);
      rv = (float)((1.0 - (double)momentum) * (double)rv + (double)momentum * unbiased);
    }
    rmean[c] = rm;
    rvar[c] = rv;
  }
}

static std::atomic<int> g_bn_finalize_fused{-1};      // -1: not initialised (environment SFOD_BN_FINALIZE_FUSED, default on)
extern "C" int sfod_set_bn_finalize_fused(int on) {
  g_bn_finalize_fused.store(on ? 1 : 0);
  return 0;
}
static bool bn_finalize_fused() {
  int v = g_bn_finalize_fused.load();
  if (v < 0) {
    const char* e = getenv("SFOD_BN_FINALIZE_FUSED");
    v = (e != nullptr && e[0] == '0') ? 0 : 1;
    g_bn_finalize_fused.store(v);
  }
  return v != 0;
}

extern "C" int sfod_bn_finalize_ws_floats(int C) {
  if (!sfod_prod_fits({BNF_SPLITS * 6, C})) return 0;
  return BNF_SPLITS * 3 * C * 2;
}

extern "C" int sfod_bn_finalize(const float* stats, int nblocks, int M, int C,
                                float* mean, float* invstd, float* running_mean, float* running_var,
                                float momentum, float eps, int update_running, int64_t* num_batches_tracked,
                                float* ws, void* stream) {
  SFOD_REQUIRE_EXTENTS("bn_finalize", nblocks, M, C);
  SFOD_REQUIRE(ws != nullptr && ((uintptr_t)ws & 7) == 0, "bn_finalize: 8-byte aligned workspace required");
  hipStream_t s = (hipStream_t)stream;
  int nsplit = (nblocks + 63) / 64;   // >= 64 blocks (4 per wave) per slice
  if (nsplit > BNF_SPLITS) nsplit = BNF_SPLITS;
  if (nsplit < 1) nsplit = 1;
  if (nsplit == 1 && bn_finalize_fused()) {
    hipLaunchKernelGGL(k_bn_finalize_one, dim3(cdiv(C, 64)), dim3(1024), 0, s, stats, nblocks, M, C, mean, invstd,
                       running_mean, running_var, momentum, eps, update_running,
                       update_running ? (long long*)num_batches_tracked : (long long*)nullptr);
    return sfod_check_launch("bn_finalize");
  }
  double* part = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(k_bn_partial, dim3(cdiv(C, 64), nsplit), dim3(1024), 0, s, stats, nblocks, C, part);
  hipLaunchKernelGGL(k_bn_final, dim3(cdiv(C, 64)), dim3(256), 0, s, part, nsplit, M, C, mean, invstd,
                     running_mean, running_var, momentum, eps, update_running,
                     update_running ? (long long*)num_batches_tracked : (long long*)nullptr);
  return sfod_check_launch("bn_finalize");
}

// ---------------------------------------------------------------------------------------------
// K3/K4 forward: z = relu(gamma * (y - mean) * invstd + beta), optional 2x2/2 max-pool
// ---------------------------------------------------------------------------------------------
// DUAL: the same values once more as SFOD_BF16X3 pairs (z2) -- "f16x3" mode, student pass: the forward products read
// half pairs, the weight gradient of the consuming layer reads bf16 pairs of the same activation
template <typename T, int POOL, int RELU, typename TO = T, bool DUAL = false>
__global__ void __launch_bounds__(256)
k_bn_relu_pool_fwd(const T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                   const float* __restrict__ gamma, const float* __restrict__ beta, TO* __restrict__ z,
                   int B, int H, int W, int C, split_t* __restrict__ z2 = nullptr) {
  constexpr int V = VecW<T, TO>::N;
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int cv = C / V;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  if constexpr (!POOL) {
    // no pooling: z and y share one layout, and when the grid stride is a multiple of the channel-vector
    // count a thread keeps ONE channel vector for its whole grid-stride walk: no index arithmetic, the
    // affine coefficients live in registers
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (stride % cv == 0) {
      int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
      const int c0 = (int)(t % cv) * V;
      float sc[V], sh[V], mu[V];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        sc[i] = invstd[c0 + i] * gamma[c0 + i];
        mu[i] = mean[c0 + i];
        sh[i] = beta[c0 + i];
      }
      for (; t < total; t += 2 * stride) {
        float v0[V], v1[V];
        const bool two = t + stride < total;
        load_n<T, V>(y + t * V, v0);
        if (two) load_n<T, V>(y + (t + stride) * V, v1);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float a = (v0[i] - mu[i]) * sc[i] + sh[i], b2 = (v1[i] - mu[i]) * sc[i] + sh[i];
          v0[i] = RELU ? fmaxf(a, 0.f) : a;
          v1[i] = RELU ? fmaxf(b2, 0.f) : b2;
        }
        store_n<TO, V>(z + t * V, v0);
        if (two) store_n<TO, V>(z + (t + stride) * V, v1);
        if constexpr (DUAL) {
          store_n<split_t, V>(z2 + t * V, v0);
          if (two) store_n<split_t, V>(z2 + (t + stride) * V, v1);
        }
      }
      return;
    }
  }
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(t % cv) * V;
    int64_t pix = t / cv;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    float sc[V], sh[V], mu[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      sc[i] = invstd[c0 + i] * gamma[c0 + i];
      mu[i] = mean[c0 + i];
      sh[i] = beta[c0 + i];
    }
    float r[V];
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = RELU ? 0.f : -3.0e38f;  // relu floor doubles as the max identity
    const int win = POOL ? 2 : 1;
#pragma unroll
    for (int dy = 0; dy < win; ++dy)
#pragma unroll
      for (int dx = 0; dx < win; ++dx) {
        const int iy = POOL ? oy * 2 + dy : oy, ix = POOL ? ox * 2 + dx : ox;
        float v[V];
        load_n<T, V>(y + (((int64_t)b * H + iy) * W + ix) * C + c0, v);
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = fmaxf(r[i], (v[i] - mu[i]) * sc[i] + sh[i]);
      }
    store_n<TO, V>(z + (((int64_t)b * Ho + oy) * Wo + ox) * C + c0, r);
    if constexpr (DUAL) store_n<split_t, V>(z2 + (((int64_t)b * Ho + oy) * Wo + ox) * C + c0, r);
  }
}

// Bottleneck tail in one pass: z = relu(bn(y) + residual) (d2 BottleneckBlock.forward: out = conv3(out);
// out += shortcut; relu) -- the normalised tensor is never written.
template <typename T>
__global__ void __launch_bounds__(256)
k_bn_add_relu_fwd(const T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                  const float* __restrict__ gamma, const float* __restrict__ beta, const T* __restrict__ res,
                  T* __restrict__ z, int64_t rows, int C) {
  constexpr int V = VecT<T>::N;
  const int cv = C / V;
  const int64_t total = rows * cv;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(t % cv) * V;
    float v[V], r[V];
    load_vec<T>(y + t * V, v);
    load_vec<T>(res + t * V, r);
#pragma unroll
    for (int i = 0; i < V; ++i)
      v[i] = fmaxf((v[i] - mean[c0 + i]) * (invstd[c0 + i] * gamma[c0 + i]) + beta[c0 + i] + r[i], 0.f);
    store_vec<T>(z + t * V, v);
  }
}

// fp32 in / out plus the same values as SFOD_BF16X3 operand pairs (the next convolution's input): the bottleneck output
// feeds both the residual stream (fp32) and conv1 of the next block, whose operand would otherwise be made by a separate
// conversion pass re-reading the tensor
template <typename PT>
__global__ void __launch_bounds__(256)
k_bn_add_relu_fwd_dual(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                       const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res,
                       float* __restrict__ z, PT* __restrict__ zp, int64_t rows, int C, split_t* __restrict__ zg) {
  // zg (may be null; f16x3, differentiated pass): the same values once more as bf16 pairs -- the next block's conv1 /
  // shortcut weight gradients read those
  constexpr int V = 8;
  const int cv = C / V;
  const int64_t total = rows * cv;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (stride % cv == 0) {
    // a thread keeps ONE channel group for its whole grid-stride walk: the affine lives in registers (as in
    // k_bn_relu_pool_fwd), two rows in flight per iteration
    const int c0 = (int)(t % cv) * V;
    float sc[V], mu[V], be[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      sc[i] = invstd[c0 + i] * gamma[c0 + i];
      mu[i] = mean[c0 + i];
      be[i] = beta[c0 + i];
    }
    for (; t < total; t += 2 * stride) {
      const bool two = t + stride < total;
      float v0[V], r0[V], v1[V], r1[V];
      load_n<float, V>(y + t * V, v0);
      load_n<float, V>(res + t * V, r0);
      if (two) {
        load_n<float, V>(y + (t + stride) * V, v1);
        load_n<float, V>(res + (t + stride) * V, r1);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        v0[i] = fmaxf((v0[i] - mu[i]) * sc[i] + be[i] + r0[i], 0.f);
        if (two) v1[i] = fmaxf((v1[i] - mu[i]) * sc[i] + be[i] + r1[i], 0.f);
      }
      store_n<float, V>(z + t * V, v0);
      store_n<PT, V>(zp + t * V, v0);
      if (zg != nullptr) store_n<split_t, V>(zg + t * V, v0);
      if (two) {
        store_n<float, V>(z + (t + stride) * V, v1);
        store_n<PT, V>(zp + (t + stride) * V, v1);
        if (zg != nullptr) store_n<split_t, V>(zg + (t + stride) * V, v1);
      }
    }
    return;
  }
  for (; t < total; t += stride) {
    const int c0 = (int)(t % cv) * V;
    float v[V], r[V];
    load_n<float, V>(y + t * V, v);
    load_n<float, V>(res + t * V, r);
#pragma unroll
    for (int i = 0; i < V; ++i)
      v[i] = fmaxf((v[i] - mean[c0 + i]) * (invstd[c0 + i] * gamma[c0 + i]) + beta[c0 + i] + r[i], 0.f);
    store_n<float, V>(z + t * V, v);
    store_n<PT, V>(zp + t * V, v);
    if (zg != nullptr) store_n<split_t, V>(zg + t * V, v);
  }
}

template <typename PT>
__global__ void __launch_bounds__(256)
k_add_act_dual(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
               PT* __restrict__ op, int64_t nvec, int act, split_t* __restrict__ og) {
  constexpr int V = 8;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
       t += (int64_t)gridDim.x * blockDim.x) {
    float x[V], y[V];
    load_n<float, V>(a + t * V, x);
    load_n<float, V>(b + t * V, y);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float v = x[i] + y[i];
      x[i] = act ? fmaxf(v, 0.f) : v;
    }
    store_n<float, V>(o + t * V, x);
    store_n<PT, V>(op + t * V, x);
    if (og != nullptr) store_n<split_t, V>(og + t * V, x);
  }
}

static inline int ew_grid(int64_t total) {
  int64_t g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int sfod_bn_add_relu_fwd(const void* y, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, const void* residual, void* z, void* z_pairs, void* z_pairs2,
                                    int64_t rows, int C, int dt, int pairs_dt, void* stream) {
  SFOD_REQUIRE(sfod_ints_ok({C}) && sfod_i64s_ok({rows, C}), "bn_add_relu_fwd: negative or oversized extent");
  SFOD_REQUIRE(z_pairs2 == nullptr || (z_pairs != nullptr && pairs_dt == SFOD_F16X3),
               "bn_add_relu: the second (bf16-pair) copy accompanies SFOD_F16X3 pairs");
  const int V = (dt == SFOD_F32 && z_pairs == nullptr) ? 4 : 8;
  SFOD_REQUIRE(C % V == 0, "bn_add_relu: C not a multiple of the vector width");
  SFOD_REQUIRE(z_pairs == nullptr || (dt == SFOD_F32 && sfod_is_pairs(pairs_dt)),
               "bn_add_relu: the operand-pair copy (SFOD_BF16X3 / SFOD_F16X3) is made from fp32 data");
  if (rows == 0) return 0;
  const int grid = ew_grid(rows * (C / V));
  hipStream_t s = (hipStream_t)stream;
  if (z_pairs != nullptr && pairs_dt == SFOD_F16X3)
    hipLaunchKernelGGL(k_bn_add_relu_fwd_dual<splith_t>, dim3(grid), dim3(256), 0, s, (const float*)y, mean, invstd, gamma,
                       beta, (const float*)residual, (float*)z, (splith_t*)z_pairs, rows, C, (split_t*)z_pairs2);
  else if (z_pairs != nullptr)
    hipLaunchKernelGGL(k_bn_add_relu_fwd_dual<split_t>, dim3(grid), dim3(256), 0, s, (const float*)y, mean, invstd, gamma,
                       beta, (const float*)residual, (float*)z, (split_t*)z_pairs, rows, C, (split_t*)nullptr);
  else if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_bn_add_relu_fwd<float>, dim3(grid), dim3(256), 0, s, (const float*)y, mean, invstd, gamma,
                       beta, (const float*)residual, (float*)z, rows, C);
  else
    hipLaunchKernelGGL(k_bn_add_relu_fwd<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, mean, invstd, gamma,
                       beta, (const bf16_t*)residual, (bf16_t*)z, rows, C);
  return sfod_check_launch("bn_add_relu_fwd");
}

extern "C" int sfod_bn_relu_pool_fwd2(const void* y, const float* mean, const float* invstd,
                                      const float* gamma, const float* beta, void* z, void* z2, int B, int H, int W,
                                      int C, int pool_flags, int dt, int out_dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("bn_relu_pool_fwd2", B, H, W, C, pool_flags);
  SFOD_REQUIRE(sfod_prod_fits({B, H, W}) && sfod_prod_fits({B, H, W, C}, 1LL << 40), "bn_relu_pool_fwd2: oversized tensor");
  hipStream_t s = (hipStream_t)stream;
  const int pool = pool_flags & 1, norelu = (pool_flags >> 1) & 1;
  SFOD_REQUIRE(out_dt == dt || (dt == SFOD_F32 && sfod_is_pairs(out_dt)),
               "bn: output type must equal the input type, or fp32 -> operand pairs");
  SFOD_REQUIRE(!sfod_is_pairs(dt), "bn: reads fp32 or bf16 (convolutions on operand pairs write fp32)");
  SFOD_REQUIRE(z2 == nullptr || out_dt == SFOD_F16X3, "bn: the second (bf16-pair) output accompanies SFOD_F16X3 pairs");
  const int V = (out_dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(C % V == 0, "bn: C not a multiple of the vector width");
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  const int grid = ew_grid((int64_t)B * Ho * Wo * (C / V));
#define LAUNCH(T, P, R, TO)                                                                                  \
  hipLaunchKernelGGL((k_bn_relu_pool_fwd<T, P, R, TO>), dim3(grid), dim3(256), 0, s, (const T*)y, mean, invstd, \
                     gamma, beta, (TO*)z, B, H, W, C, (split_t*)nullptr)
#define LAUNCH_DUAL(P, R)                                                                                            \
  hipLaunchKernelGGL((k_bn_relu_pool_fwd<float, P, R, splith_t, true>), dim3(grid), dim3(256), 0, s, (const float*)y, \
                     mean, invstd, gamma, beta, (splith_t*)z, B, H, W, C, (split_t*)z2)
#define LAUNCH4(T, TO)                                                           \
  do {                                                                           \
    if (norelu) { if (pool) LAUNCH(T, 1, 0, TO); else LAUNCH(T, 0, 0, TO); }     \
    else { if (pool) LAUNCH(T, 1, 1, TO); else LAUNCH(T, 0, 1, TO); }            \
  } while (0)
  if (z2 != nullptr) {
    if (norelu) { if (pool) LAUNCH_DUAL(1, 0); else LAUNCH_DUAL(0, 0); }
    else { if (pool) LAUNCH_DUAL(1, 1); else LAUNCH_DUAL(0, 1); }
  } else if (out_dt == SFOD_BF16X3) LAUNCH4(float, split_t);
  else if (out_dt == SFOD_F16X3) LAUNCH4(float, splith_t);
  else if (dt == SFOD_F32) LAUNCH4(float, float);
  else LAUNCH4(bf16_t, bf16_t);
#undef LAUNCH4
#undef LAUNCH_DUAL
#undef LAUNCH
  return sfod_check_launch("bn_relu_pool_fwd");
}

extern "C" int sfod_bn_relu_pool_fwd(const void* y, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, void* z, int B, int H, int W,
                                     int C, int pool_flags, int dt, int out_dt, void* stream) {
  return sfod_bn_relu_pool_fwd2(y, mean, invstd, gamma, beta, z, nullptr, B, H, W, C, pool_flags, dt, out_dt, stream);
}

// ---------------------------------------------------------------------------------------------
// K3/K4 backward.  Units: a 2x2 window (pool) or one pixel; leftover pixels of odd H/W (not
// covered by a window) take part in the batch statistics with zero upstream gradient.
//   g     = dz routed to the first max of the window, gated by relu (z_pre > 0)
//   dbeta = sum g ; dgamma = sum g * xhat ; dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M)
// pass 1 writes per-workgroup partial sums, pass 2 reduces them (fp64, fixed order), pass 3
// writes dy.
// ---------------------------------------------------------------------------------------------
#define BNB_ROWS 64  // units per workgroup in pass 1

template <typename T, int POOL, int RELU, int V = VecT<T>::N>
__device__ __forceinline__ void bn_unit_grad(const T* __restrict__ y, const T* __restrict__ dz, int b,
                                             int oy, int ox, int H, int W, int C, int c0,
                                             const float* mu, const float* sc, const float* sh,
                                             float (*xhat_out)[V], float (*g_out)[V],
                                             const float* invs) {
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  float gz[V];
  load_n<T, V>(dz + (((int64_t)b * Ho + oy) * Wo + ox) * C + c0, gz);
  constexpr int NW = POOL ? 4 : 1;
  float zp[NW][V];
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int iy = POOL ? oy * 2 + (k >> 1) : oy, ix = POOL ? ox * 2 + (k & 1) : ox;
    float v[V];
    load_n<T, V>(y + (((int64_t)b * H + iy) * W + ix) * C + c0, v);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      xhat_out[k][i] = (v[i] - mu[i]) * invs[i];
      zp[k][i] = (v[i] - mu[i]) * sc[i] + sh[i];
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    int arg = 0;
    float best = zp[0][i];
#pragma unroll
    for (int k = 1; k < NW; ++k)
      if (zp[k][i] > best) { best = zp[k][i]; arg = k; }
#pragma unroll
    for (int k = 0; k < NW; ++k) g_out[k][i] = (k == arg && (!RELU || best > 0.f)) ? gz[i] : 0.f;
  }
}


#define BNB_GRID_MAX 1024

template <typename T, int POOL, int RELU>
__global__ void __launch_bounds__(256)
k_bn_bwd_reduce(const T* __restrict__ dz, const T* __restrict__ y, const float* __restrict__ mean,
                const float* __restrict__ invstd, const float* __restrict__ gamma,
                const float* __restrict__ beta, float* __restrict__ ws, int B, int H, int W, int C) {
  constexpr int V = VecT<T>::N;
  constexpr int NW = POOL ? 4 : 1;
  extern __shared__ __attribute__((aligned(16))) float sred[];  // [UL][2][C]
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int cv = C / V;
  const int UL = blockDim.x / cv;  // unit lanes per workgroup (cv <= 256 required)
  const int cl = threadIdx.x % cv, ul = threadIdx.x / cv;
  const int c0 = cl * V;
  const int64_t units = (int64_t)B * Ho * Wo;
  float mu[V], sc[V], sh[V], invs[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    invs[i] = invstd[c0 + i];
    sc[i] = invs[i] * gamma[c0 + i];
    mu[i] = mean[c0 + i];
    sh[i] = beta[c0 + i];
  }
  float db[V], dg[V];
#pragma unroll
  for (int i = 0; i < V; ++i) { db[i] = 0.f; dg[i] = 0.f; }
  if (ul < UL) {
    for (int64_t u = (int64_t)blockIdx.x * UL + ul; u < units; u += (int64_t)gridDim.x * UL) {
      if constexpr (!POOL) {   // flat: unit u is pixel u
        float v[V], gz[V];
        load_vec<T>(y + u * C + c0, v);
        load_vec<T>(dz + u * C + c0, gz);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float xh = (v[i] - mu[i]) * invs[i];
          const float zp = (v[i] - mu[i]) * sc[i] + sh[i];
          const float gg = (!RELU || zp > 0.f) ? gz[i] : 0.f;
          db[i] += gg;
          dg[i] += gg * xh;
        }
        continue;
      }
      const int ox = (int)(u % Wo);
      const int64_t t = u / Wo;
      const int oy = (int)(t % Ho);
      const int b = (int)(t / Ho);
      float xh[NW][V], g[NW][V];
      bn_unit_grad<T, POOL, RELU>(y, dz, b, oy, ox, H, W, C, c0, mu, sc, sh, xh, g, invs);
#pragma unroll
      for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) { db[i] += g[k][i]; dg[i] += g[k][i] * xh[k][i]; }
    }
    for (int i = 0; i < V; ++i) {
      sred[(ul * 2 + 0) * C + c0 + i] = db[i];
      sred[(ul * 2 + 1) * C + c0 + i] = dg[i];
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * C; t += blockDim.x) {
    float a = 0.f;
    for (int l = 0; l < UL; ++l) a += sred[l * 2 * C + t];
    ws[(int64_t)blockIdx.x * 2 * C + t] = a;
  }
}

__global__ void __launch_bounds__(1024)
k_bn_bwd_finalize(const float* __restrict__ ws, int nblk, int C, float* __restrict__ dgamma,
                  float* __restrict__ dbeta, float* __restrict__ dgamma_acc, float* __restrict__ dbeta_acc) {
  // 64 consecutive columns of the [nblk][2C] partial matrix per workgroup; 16 waves split the rows
  // (4 independent loads in flight per lane), fixed summation order
  __shared__ double red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;  // over 2*C
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (t < 2 * C) {
    int b = wave;
    for (; b + 48 < nblk; b += 64) {
      a0 += (double)ws[(int64_t)b * 2 * C + t];
      a1 += (double)ws[(int64_t)(b + 16) * 2 * C + t];
      a2 += (double)ws[(int64_t)(b + 32) * 2 * C + t];
      a3 += (double)ws[(int64_t)(b + 48) * 2 * C + t];
    }
    for (; b < nblk; b += 16) a0 += (double)ws[(int64_t)b * 2 * C + t];
  }
  red[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && t < 2 * C) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][lane];
    if (t < C) {
      dbeta[t] = (float)v;
      if (dbeta_acc != nullptr) dbeta_acc[t] += (float)v;
    } else {
      dgamma[t - C] = (float)v;
      if (dgamma_acc != nullptr) dgamma_acc[t - C] += (float)v;
    }
  }
}

// First stage for long partial matrices (the fused data-gradient epilogue writes one row per pixel tile: 22 800 rows for
// conv1_1 at B = 8, 600x1200): slice s of the rows -> out[s][2C], so that the finalize kernel above (2C / 64 workgroups)
// only sees BNB_SLICES rows.  Same fixed summation order inside a slice.
#define BNB_SLICES 32
static_assert(BNB_SLICES == SFOD_BN_BWD_SCRATCH_ROWS, "scratch rows of the pre-reduced workspace");
__global__ void __launch_bounds__(1024)
k_bn_bwd_slices(const float* __restrict__ ws, int nblk, int C, float* __restrict__ out) {
  __shared__ double red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;  // over 2*C
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  double a0 = 0.0, a1 = 0.0;
  if (t < 2 * C) {
    int b = b0 + wave;
    for (; b + 16 < b1; b += 32) {
      a0 += (double)ws[(int64_t)b * 2 * C + t];
      a1 += (double)ws[(int64_t)(b + 16) * 2 * C + t];
    }
    for (; b < b1; b += 16) a0 += (double)ws[(int64_t)b * 2 * C + t];
  }
  red[wave][lane] = a0 + a1;
  __syncthreads();
  if (wave == 0 && t < 2 * C) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w][lane];
    out[(int64_t)blockIdx.y * 2 * C + t] = (float)v;
  }
}

template <typename T, int POOL, int RELU, typename TO = T>
__global__ void __launch_bounds__(256)
k_bn_bwd_apply(const T* __restrict__ dz, const T* __restrict__ y, const float* __restrict__ mean,
               const float* __restrict__ invstd, const float* __restrict__ gamma,
               const float* __restrict__ beta, const float* __restrict__ dgamma,
               const float* __restrict__ dbeta, TO* __restrict__ dy, int B, int H, int W, int C) {
  constexpr int V = VecW<T, TO>::N;
  constexpr int NW = POOL ? 4 : 1;
  const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
  const int cv = C / V;
  const float invM = 1.f / (float)((int64_t)B * H * W);
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  if constexpr (!POOL) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (stride % cv == 0) {   // flat walk with a fixed channel vector per thread (see the forward kernel)
      int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
      const int c0 = (int)(t % cv) * V;
      float mu[V], sc[V], sh[V], invs[V], k1[V], k2[V];
#pragma unroll
      for (int i = 0; i < V; ++i) {
        invs[i] = invstd[c0 + i];
        sc[i] = invs[i] * gamma[c0 + i];
        mu[i] = mean[c0 + i];
        sh[i] = beta[c0 + i];
        k1[i] = dbeta[c0 + i] * invM;
        k2[i] = dgamma[c0 + i] * invM;
      }
      for (; t < total; t += stride) {
        float v[V], gz[V], o[V];
        load_n<T, V>(y + t * V, v);
        load_n<T, V>(dz + t * V, gz);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float xh = (v[i] - mu[i]) * invs[i];
          const float zp = (v[i] - mu[i]) * sc[i] + sh[i];
          const float g = (!RELU || zp > 0.f) ? gz[i] : 0.f;
          o[i] = sc[i] * (g - k1[i] - xh * k2[i]);
        }
        store_n<TO, V>(dy + t * V, o);
      }
      return;
    }
  }
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(t % cv) * V;
    int64_t pix = t / cv;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    float mu[V], sc[V], sh[V], invs[V], k1[V], k2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      invs[i] = invstd[c0 + i];
      sc[i] = invs[i] * gamma[c0 + i];
      mu[i] = mean[c0 + i];
      sh[i] = beta[c0 + i];
      k1[i] = dbeta[c0 + i] * invM;
      k2[i] = dgamma[c0 + i] * invM;
    }
    float xh[NW][V], g[NW][V];
    bn_unit_grad<T, POOL, RELU, V>(y, dz, b, oy, ox, H, W, C, c0, mu, sc, sh, xh, g, invs);
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int iy = POOL ? oy * 2 + (k >> 1) : oy, ix = POOL ? ox * 2 + (k & 1) : ox;
      float o[V];
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = sc[i] * (g[k][i] - k1[i] - xh[k][i] * k2[i]);
      store_n<TO, V>(dy + (((int64_t)b * H + iy) * W + ix) * C + c0, o);
    }
  }
}

// pixels of an odd-sized map that no 2x2 window covers: zero upstream gradient
template <typename T, typename TO = T>
__global__ void __launch_bounds__(256)
k_bn_bwd_leftover(const T* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                  const float* __restrict__ gamma, const float* __restrict__ dgamma,
                  const float* __restrict__ dbeta, TO* __restrict__ dy, int B, int H, int W, int C) {
  constexpr int V = VecW<T, TO>::N;
  const int Ho = H / 2, Wo = W / 2;
  const int cv = C / V;
  const int la = 2 * Ho * (W - 2 * Wo);      // right column (x = W-1) for y < 2Ho
  const int L = la + (H - 2 * Ho) * W;       // + bottom row
  const float invM = 1.f / (float)((int64_t)B * H * W);
  const int64_t total = (int64_t)B * L * cv;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(t % cv) * V;
    const int64_t q = t / cv;
    const int l = (int)(q % L);
    const int b = (int)(q / L);
    int iy, ix;
    if (l < la) { iy = l; ix = 2 * Wo; } else { iy = 2 * Ho; ix = l - la; }
    float v[V], o[V];
    const int64_t off = (((int64_t)b * H + iy) * W + ix) * C + c0;
    load_n<T, V>(y + off, v);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float invs = invstd[c0 + i];
      const float xh = (v[i] - mean[c0 + i]) * invs;
      o[i] = invs * gamma[c0 + i] * (0.f - dbeta[c0 + i] * invM - xh * dgamma[c0 + i] * invM);
    }
    store_n<TO, V>(dy + off, o);
  }
}

extern "C" int sfod_bn_bwd_ws_floats(int M, int C) {
  if (!sfod_ints_ok({M, C}) || !sfod_prod_fits({BNB_GRID_MAX * 2, C})) return 0;      // hostile extents: not served / nothing
  (void)M;
  return BNB_GRID_MAX * 2 * C;
}

extern "C" int sfod_bn_relu_pool_bwd(const void* dz, const void* y, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, void* dy, float* dgamma,
                                     float* dbeta, float* dgamma_acc, float* dbeta_acc, float* ws, int B, int H,
                                     int W, int C, int pool_flags, int dt, int dy_dt, int reduced_blocks,
                                     void* stream) {
  SFOD_REQUIRE_EXTENTS("bn_relu_pool_bwd", B, H, W, C, pool_flags, dy_dt, reduced_blocks);
  SFOD_REQUIRE(sfod_prod_fits({B, H, W}) && sfod_prod_fits({B, H, W, C}, 1LL << 40), "bn_relu_pool_bwd: oversized tensor");
  hipStream_t s = (hipStream_t)stream;
  const int pool = pool_flags & 1, norelu = (pool_flags >> 1) & 1;
  SFOD_REQUIRE(reduced_blocks >= 0 && (reduced_blocks == 0 || (!pool && !norelu)),
               "bn_bwd: a pre-reduced workspace (sfod_conv_dgrad_bnred) is for ReLU layers without pooling");
  SFOD_REQUIRE(dt != SFOD_BF16X3 && (dy_dt == dt || (dt == SFOD_F32 && dy_dt == SFOD_BF16X3)),
               "bn_bwd: dz / y are fp32 or bf16; dy has the same type, or bf16x3 from fp32");
  const int V = (dt == SFOD_F32) ? 4 : 8;            // reduce pass (reads only)
  const int VO = (dy_dt == SFOD_F32) ? 4 : 8;        // apply pass
  SFOD_REQUIRE(C >= 8 && C % VO == 0 && C / V <= 256, "bn_bwd: unsupported channel count");
  SFOD_REQUIRE(dz && y && mean && invstd && gamma && beta && ws, "bn_bwd: null argument");
  const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  const int cv = C / V, UL = 256 / cv;
  const int64_t units = (int64_t)B * Ho * Wo;
  int grid1 = (int)((units + UL - 1) / UL);
  if (grid1 > BNB_GRID_MAX) grid1 = BNB_GRID_MAX;
  if (grid1 < 1) grid1 = 1;
  const size_t lds = sizeof(float) * UL * 2 * C;
  const int grid3 = ew_grid(units * (C / VO));
#define LAUNCH(T, P, R, TO)                                                                            \
  do {                                                                                                 \
    if (reduced_blocks == 0)                                                                           \
      hipLaunchKernelGGL((k_bn_bwd_reduce<T, P, R>), dim3(grid1), dim3(256), lds, s, (const T*)dz,     \
                         (const T*)y, mean, invstd, gamma, beta, ws, B, H, W, C);                      \
    if (reduced_blocks > 2048) {  /* long pre-reduced matrix: slice it first (scratch rows behind it) */   \
      float* sl = ws + (int64_t)reduced_blocks * 2 * C;                                                \
      hipLaunchKernelGGL(k_bn_bwd_slices, dim3(cdiv(2 * C, 64), BNB_SLICES), dim3(1024), 0, s, ws,     \
                         reduced_blocks, C, sl);                                                       \
      hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(cdiv(2 * C, 64)), dim3(1024), 0, s, sl, BNB_SLICES, C, \
                         dgamma, dbeta, dgamma_acc, dbeta_acc);                                        \
    } else                                                                                             \
      hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(cdiv(2 * C, 64)), dim3(1024), 0, s, ws,               \
                         reduced_blocks ? reduced_blocks : grid1, C, dgamma, dbeta, dgamma_acc, dbeta_acc); \
    hipLaunchKernelGGL((k_bn_bwd_apply<T, P, R, TO>), dim3(grid3), dim3(256), 0, s, (const T*)dz,      \
                       (const T*)y, mean, invstd, gamma, beta, dgamma, dbeta, (TO*)dy, B, H, W, C);    \
  } while (0)
#define LAUNCH4(T, TO)                                                           \
  do {                                                                           \
    if (norelu) { if (pool) LAUNCH(T, 1, 0, TO); else LAUNCH(T, 0, 0, TO); }     \
    else { if (pool) LAUNCH(T, 1, 1, TO); else LAUNCH(T, 0, 1, TO); }            \
  } while (0)
  if (dy_dt == SFOD_BF16X3) LAUNCH4(float, split_t);
  else if (dt == SFOD_F32) LAUNCH4(float, float);
  else LAUNCH4(bf16_t, bf16_t);
#undef LAUNCH4
#undef LAUNCH
  int rc = sfod_check_launch("bn_relu_pool_bwd");
  if (rc) return rc;
  if (pool && ((H & 1) || (W & 1))) {
    const int L = 2 * Ho * (W - 2 * Wo) + (H - 2 * Ho) * W;
    const int grid = ew_grid((int64_t)B * L * (C / VO));
    if (dy_dt == SFOD_BF16X3)
      hipLaunchKernelGGL((k_bn_bwd_leftover<float, split_t>), dim3(grid), dim3(256), 0, s, (const float*)y, mean,
                         invstd, gamma, dgamma, dbeta, (split_t*)dy, B, H, W, C);
    else if (dt == SFOD_F32)
      hipLaunchKernelGGL((k_bn_bwd_leftover<float, float>), dim3(grid), dim3(256), 0, s, (const float*)y, mean,
                         invstd, gamma, dgamma, dbeta, (float*)dy, B, H, W, C);
    else
      hipLaunchKernelGGL((k_bn_bwd_leftover<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)y, mean,
                         invstd, gamma, dgamma, dbeta, (bf16_t*)dy, B, H, W, C);
    rc = sfod_check_launch("bn_bwd_leftover");
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------
// activation backward / add
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_act_bwd(T* __restrict__ dy, const T* __restrict__ y, int64_t nvec, int act) {
  constexpr int V = VecT<T>::N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
       t += (int64_t)gridDim.x * blockDim.x) {
    float g[V], v[V];
    load_vec<T>(dy + t * V, g);
    load_vec<T>(y + t * V, v);
#pragma unroll
    for (int i = 0; i < V; ++i) g[i] = v[i] > 0.f ? g[i] : (act == 2 ? 0.2f * g[i] : 0.f);
    store_vec<T>(dy + t * V, g);
  }
}

extern "C" int sfod_act_bwd(void* dy, const void* y, int64_t n, int act, int dt, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "act_bwd: negative or oversized extent");
  const int V = (dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(n % V == 0, "act_bwd: n not a multiple of the vector width");
  const int64_t nvec = n / V;
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_act_bwd<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (float*)dy,
                       (const float*)y, nvec, act);
  else
    hipLaunchKernelGGL(k_act_bwd<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)dy, (const bf16_t*)y, nvec, act);
  return sfod_check_launch("act_bwd");
}

template <typename T>
__global__ void k_add_inplace(T* __restrict__ a, const T* __restrict__ b, int64_t nvec) {
  constexpr int V = VecT<T>::N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
       t += (int64_t)gridDim.x * blockDim.x) {
    float x[V], yv[V];
    load_vec<T>(a + t * V, x);
    load_vec<T>(b + t * V, yv);
#pragma unroll
    for (int i = 0; i < V; ++i) x[i] += yv[i];
    store_vec<T>(a + t * V, x);
  }
}

extern "C" int sfod_add_inplace(void* a, const void* b, int64_t n, int dt, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "add_inplace: negative or oversized extent");
  const int V = (dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(n % V == 0, "add_inplace: n not a multiple of the vector width");
  const int64_t nvec = n / V;
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_add_inplace<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (float*)a, (const float*)b, nvec);
  else
    hipLaunchKernelGGL(k_add_inplace<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)a, (const bf16_t*)b, nvec);
  return sfod_check_launch("add_inplace");
}

// a = (a + b) * [y > 0]: the residual join's two gradient branches summed and taken through the ReLU of the block
// below in one pass (was k_add_inplace followed by that block's k_act_bwd: one read and one write of the tensor less)
template <typename T>
__global__ void k_add_act_bwd(T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ y, int64_t nvec) {
  constexpr int V = VecT<T>::N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
       t += (int64_t)gridDim.x * blockDim.x) {
    float x[V], bv[V], yv[V];
    load_vec<T>(a + t * V, x);
    load_vec<T>(b + t * V, bv);
    load_vec<T>(y + t * V, yv);
#pragma unroll
    for (int i = 0; i < V; ++i) x[i] = yv[i] > 0.f ? x[i] + bv[i] : 0.f;
    store_vec<T>(a + t * V, x);
  }
}

extern "C" int sfod_add_act_bwd(void* a, const void* b, const void* y, int64_t n, int dt, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "add_act_bwd: negative or oversized extent");
  const int V = (dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(n % V == 0, "add_act_bwd: n not a multiple of the vector width");
  const int64_t nvec = n / V;
  if (nvec == 0) return 0;
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_add_act_bwd<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (float*)a,
                       (const float*)b, (const float*)y, nvec);
  else
    hipLaunchKernelGGL(k_add_act_bwd<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)a,
                       (const bf16_t*)b, (const bf16_t*)y, nvec);
  return sfod_check_launch("add_act_bwd");
}

// a *= mask * scale  (dropout forward / backward: mask is a 0/1 byte per element, scale = 1 / (1 - p))
template <typename T>
__global__ void k_mul_mask(T* __restrict__ a, const uint8_t* __restrict__ m, int64_t nvec, float scale) {
  constexpr int V = VecT<T>::N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
       t += (int64_t)gridDim.x * blockDim.x) {
    float x[V];
    load_vec<T>(a + t * V, x);
#pragma unroll
    for (int i = 0; i < V; ++i) x[i] = m[t * V + i] ? x[i] * scale : 0.f;
    store_vec<T>(a + t * V, x);
  }
}

extern "C" int sfod_mul_mask(void* a, const uint8_t* mask, int64_t n, float scale, int dt, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "mul_mask: negative or oversized extent");
  const int V = (dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(n % V == 0, "mul_mask: n not a multiple of the vector width");
  const int64_t nvec = n / V;
  if (nvec == 0) return 0;
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_mul_mask<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (float*)a, mask,
                       nvec, scale);
  else
    hipLaunchKernelGGL(k_mul_mask<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)a, mask,
                       nvec, scale);
  return sfod_check_launch("mul_mask");
}

// ---------------------------------------------------------------------------------------------
// ResNet-C4 helpers (d2 build_resnet_backbone: BasicStem + BottleneckBlock)
// ---------------------------------------------------------------------------------------------
// out = act(a + b): the residual join of a bottleneck block (relu(conv3_bn(x) + shortcut))
template <typename T>
__global__ void k_add_act(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t nvec, int act) {
  constexpr int V = VecT<T>::N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nvec;
       t += (int64_t)gridDim.x * blockDim.x) {
    float x[V], y[V];
    load_vec<T>(a + t * V, x);
    load_vec<T>(b + t * V, y);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float v = x[i] + y[i];
      x[i] = act ? fmaxf(v, 0.f) : v;
    }
    store_vec<T>(o + t * V, x);
  }
}

extern "C" int sfod_add_act(const void* a, const void* b, void* out, void* out_pairs, void* out_pairs2, int64_t n, int act,
                            int dt, int pairs_dt, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "add_act: negative or oversized extent");
  SFOD_REQUIRE(out_pairs2 == nullptr || (out_pairs != nullptr && pairs_dt == SFOD_F16X3),
               "add_act: the second (bf16-pair) copy accompanies SFOD_F16X3 pairs");
  const int V = (dt == SFOD_F32 && out_pairs == nullptr) ? 4 : 8;
  SFOD_REQUIRE(n % V == 0, "add_act: n not a multiple of the vector width");
  SFOD_REQUIRE(out_pairs == nullptr || (dt == SFOD_F32 && sfod_is_pairs(pairs_dt)),
               "add_act: the operand-pair copy (SFOD_BF16X3 / SFOD_F16X3) is made from fp32 data");
  const int64_t nvec = n / V;
  if (nvec == 0) return 0;
  if (out_pairs != nullptr && pairs_dt == SFOD_F16X3)
    hipLaunchKernelGGL(k_add_act_dual<splith_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)a,
                       (const float*)b, (float*)out, (splith_t*)out_pairs, nvec, act, (split_t*)out_pairs2);
  else if (out_pairs != nullptr)
    hipLaunchKernelGGL(k_add_act_dual<split_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)a,
                       (const float*)b, (float*)out, (split_t*)out_pairs, nvec, act, (split_t*)nullptr);
  else if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_add_act<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)a,
                       (const float*)b, (float*)out, nvec, act);
  else
    hipLaunchKernelGGL(k_add_act<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, nvec, act);
  return sfod_check_launch("add_act");
}

// stride-2 pixel subsampling (the data movement of a 1x1 stride-2 convolution) and its adjoint
template <typename T, int BWD>
__global__ void k_subsample2(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int C) {
  // fwd: dst [B,Ho,Wo,C] = src [B,H,W,C] at even pixels; bwd: dst [B,H,W,C] = src [B,Ho,Wo,C] at even pixels, else 0
  constexpr int V = VecT<T>::N;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int cv = C / V;
  const int64_t total = BWD ? (int64_t)B * H * W * cv : (int64_t)B * Ho * Wo * cv;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(t % cv) * V;
    int64_t pix = t / cv;
    const int wd = BWD ? W : Wo, hd = BWD ? H : Ho;
    const int x = (int)(pix % wd);
    pix /= wd;
    const int y = (int)(pix % hd);
    const int b = (int)(pix / hd);
    float v[V];
    if (BWD) {
      if ((x & 1) == 0 && (y & 1) == 0) load_vec<T>(src + (((int64_t)b * Ho + y / 2) * Wo + x / 2) * C + c0, v);
      else {
#pragma unroll
        for (int i = 0; i < V; ++i) v[i] = 0.f;
      }
      store_vec<T>(dst + (((int64_t)b * H + y) * W + x) * C + c0, v);
    } else {
      load_vec<T>(src + (((int64_t)b * H + 2 * y) * W + 2 * x) * C + c0, v);
      store_vec<T>(dst + (((int64_t)b * Ho + y) * Wo + x) * C + c0, v);
    }
  }
}

extern "C" int sfod_subsample2(const void* src, void* dst, int B, int H, int W, int C, int backward, int dt,
                               void* stream) {
  SFOD_REQUIRE_EXTENTS("subsample2", B, H, W, C);
  SFOD_REQUIRE(sfod_prod_fits({B, H, W}) && sfod_prod_fits({B, H, W, C}, 1LL << 40), "subsample2: oversized tensor");
  const int V = (dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(C % V == 0, "subsample2: C not a multiple of the vector width");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t total = backward ? (int64_t)B * H * W * (C / V) : (int64_t)B * Ho * Wo * (C / V);
  if (total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(ew_grid(total)), blk(256);
  if (dt == SFOD_F32) {
    if (backward) hipLaunchKernelGGL((k_subsample2<float, 1>), g, blk, 0, s, (const float*)src, (float*)dst, B, H, W, C);
    else hipLaunchKernelGGL((k_subsample2<float, 0>), g, blk, 0, s, (const float*)src, (float*)dst, B, H, W, C);
  } else {
    if (backward) hipLaunchKernelGGL((k_subsample2<bf16_t, 1>), g, blk, 0, s, (const bf16_t*)src, (bf16_t*)dst, B, H, W, C);
    else hipLaunchKernelGGL((k_subsample2<bf16_t, 0>), g, blk, 0, s, (const bf16_t*)src, (bf16_t*)dst, B, H, W, C);
  }
  return sfod_check_launch("subsample2");
}

// im2col of the 7x7 stride-2 pad-3 stem convolution: out [B,Ho,Wo,Kpad], k = (ky*7+kx)*3 + c for the
// first 147 columns, zeros up to Kpad.  The stem is frozen (FREEZE_AT=2), so only the forward exists;
// the GEMM itself (K = Kpad, N = 64, FrozenBN folded into the weights, ReLU) runs on sfod_conv_fwd.
// column k -> (ky, kx, c) packed as ky | kx << 8 | c << 16 (0xffffffff: padding column): a per-workgroup LDS table -- the
// per-element divisions were the kernel's cost (8 scattered 4-byte loads per thread otherwise hit L1 / L2); lanes of a
// wave index it at different k, which a __constant__ table serialises (tried: 630 -> 760 us)
template <typename T, typename TO = T>
__global__ void k_im2col_stem(const T* __restrict__ x, TO* __restrict__ out, int B, int H, int W, int Cp, int Kpad) {
  __shared__ unsigned lut[256];
  for (int k = threadIdx.x; k < 256; k += blockDim.x) {
    unsigned code = 0xffffffffu;
    if (k < 147) {
      const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
      code = (unsigned)ky | ((unsigned)kx << 8) | ((unsigned)c << 16);
    }
    lut[k] = code;
  }
  __syncthreads();
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int kv = Kpad / 8;   // 8-element groups per row
  const int64_t total = (int64_t)B * Ho * Wo * kv;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(t % kv);
    int64_t pix = t / kv;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const T* xb = x + (int64_t)b * H * W * Cp;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = g * 8 + e;
      const unsigned code = k < 256 ? lut[k] : 0xffffffffu;
      float val = 0.f;
      if (code != 0xffffffffu) {
        const int iy = 2 * oy - 3 + (int)(code & 0xffu), ix = 2 * ox - 3 + (int)((code >> 8) & 0xffu);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = to_f32(xb[((int64_t)iy * W + ix) * Cp + (code >> 16)]);
      }
      v[e] = val;
    }
    store_n<TO, 8>(out + t * 8, v);       // 16 / 32-byte stores; TO = split_t: the GEMM's operand pairs directly
  }
}

extern "C" int sfod_im2col_stem(const void* x, void* out, int B, int H, int W, int Cp, int Kpad, int dt, int out_dt,
                                void* stream) {
  SFOD_REQUIRE_EXTENTS("im2col_stem", B, H, W, Cp, Kpad);
  SFOD_REQUIRE(sfod_prod_fits({B, H, W}) && sfod_prod_fits({B, H, W, Cp > Kpad ? Cp : Kpad}, 1LL << 40), "im2col_stem: oversized tensor");
  SFOD_REQUIRE(Kpad >= 152 && Kpad % 8 == 0 && Cp >= 3, "im2col_stem: Kpad must be a multiple of 8 >= 152");
  SFOD_REQUIRE(out_dt == dt || (dt == SFOD_F32 && sfod_is_pairs(out_dt)), "im2col_stem: output is the input type, or operand pairs from fp32");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (Kpad / 8);
  if (total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (out_dt == SFOD_BF16X3)
    hipLaunchKernelGGL((k_im2col_stem<float, split_t>), dim3(ew_grid(total)), dim3(256), 0, s, (const float*)x,
                       (split_t*)out, B, H, W, Cp, Kpad);
  else if (out_dt == SFOD_F16X3)
    hipLaunchKernelGGL((k_im2col_stem<float, splith_t>), dim3(ew_grid(total)), dim3(256), 0, s, (const float*)x,
                       (splith_t*)out, B, H, W, Cp, Kpad);
  else if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_im2col_stem<float>, dim3(ew_grid(total)), dim3(256), 0, s, (const float*)x, (float*)out, B, H,
                       W, Cp, Kpad);
  else
    hipLaunchKernelGGL(k_im2col_stem<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B,
                       H, W, Cp, Kpad);
  return sfod_check_launch("im2col_stem");
}

// max-pool 3x3 stride 2 pad 1 (BasicStem), forward only
template <typename T>
__global__ void k_maxpool3s2(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int V = VecT<T>::N;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int cv = C / V;
  const int64_t total = (int64_t)B * Ho * Wo * cv;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(t % cv) * V;
    int64_t pix = t / cv;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    float r[V];
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = -3.0e38f;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int iy = 2 * oy + dy, ix = 2 * ox + dx;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        float v[V];
        load_vec<T>(x + (((int64_t)b * H + iy) * W + ix) * C + c0, v);
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = fmaxf(r[i], v[i]);
      }
    store_vec<T>(y + (((int64_t)b * Ho + oy) * Wo + ox) * C + c0, r);
  }
}

extern "C" int sfod_maxpool3s2(const void* x, void* y, int B, int H, int W, int C, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("maxpool3s2", B, H, W, C);
  SFOD_REQUIRE(sfod_prod_fits({B, H, W}) && sfod_prod_fits({B, H, W, C}, 1LL << 40), "maxpool3s2: oversized tensor");
  if (B == 0 || H == 0 || W == 0 || C == 0) return 0;
  const int V = (dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(C % V == 0, "maxpool3s2: C not a multiple of the vector width");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (C / V);
  if (total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_maxpool3s2<float>, dim3(ew_grid(total)), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C);
  else
    hipLaunchKernelGGL(k_maxpool3s2<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
  return sfod_check_launch("maxpool3s2");
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
// max|w| of fp32 tensors as the bit pattern of a non-negative float (orders like an unsigned; a NaN wins), combined
// with atomicMax into amax[blockIdx.y] (zeroed by the caller): the per-tensor scale of SFOD_F16X3 weights (common.h).
// desc == nullptr: one tensor (w, count); else entry blockIdx.y of a sfod_pack_conv_weights_multi table.
__global__ void __launch_bounds__(256)
k_weight_absmax(const float* __restrict__ w, int64_t count, const long long* __restrict__ desc, unsigned* __restrict__ amax) {
  if (desc != nullptr) {
    const int e = blockIdx.y;
    w = reinterpret_cast<const float*>(desc[e * 8 + 0]);
    count = (int64_t)desc[e * 8 + 2] * desc[e * 8 + 3] * desc[e * 8 + 4] * desc[e * 8 + 4];
  }
  unsigned m = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {       // 16-byte loads (fc1: 100-400 MB per call)
    const int64_t n4 = count >> 2;
    const uint4* w4 = reinterpret_cast<const uint4*>(w);
    for (int64_t i = tid; i < n4; i += stride) {
      const uint4 v = w4[i];
      m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    }
    for (int64_t i = (n4 << 2) + tid; i < count; i += stride) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
  } else {
    for (int64_t i = tid; i < count; i += stride) m = max(m, __float_as_uint(w[i]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m != 0) atomicMax(amax + blockIdx.y, m);
}
static int launch_weight_absmax(const float* w, int64_t count, const long long* desc, int n, unsigned* amax, hipStream_t s) {
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(unsigned) * n, s);
  if (e != hipSuccess) { sfod_set_error("weight absmax: %s", hipGetErrorString(e)); return -(int)e; }
  int gx = desc != nullptr ? 32 : (int)((count + 256 * 64 - 1) / (256 * 64));      // >= 64 elements per thread
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_weight_absmax, dim3(gx, n), dim3(256), 0, s, w, count, desc, amax);
  return sfod_check_launch("weight_absmax");
}

template <typename T>
__global__ void k_pack_conv_weight(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin,
                                   int ks, int innerPad, int rot180, const unsigned* __restrict__ amax) {
  const float ws = amax != nullptr ? wscale_from_absmax(*amax) : 1.0f;     // SFOD_F16X3: per-tensor power of two (common.h)
  // normal : out[co][tap][ci]  (inner = Cin  -> innerPad)
  // rot180 : out[ci][tap'][co] (inner = Cout -> innerPad), tap' = flipped tap
  const int taps = ks * ks;
  const int rows = rot180 ? Cin : Cout;
  const int64_t total = (int64_t)rows * taps * innerPad;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int inner = (int)(t % innerPad);
    const int tap = (int)((t / innerPad) % taps);
    const int row = (int)(t / ((int64_t)innerPad * taps));
    float v = 0.f;
    const int innerN = rot180 ? Cout : Cin;
    if (inner < innerN) {
      const int co = rot180 ? inner : row, ci = rot180 ? row : inner;
      const int st = rot180 ? (taps - 1 - tap) : tap;
      v = w[((int64_t)co * Cin + ci) * taps + st];
    }
    put_elem<T>(out, t, v * ws);
  }
}

extern "C" int sfod_pack_conv_weight_ws(const float* w_oihw, void* w_packed, uint32_t* absmax, int Cout, int Cin,
                                        int ksize, int CinPad, int rot180, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("pack_conv_weight_ws", Cout, Cin, ksize, CinPad);
  SFOD_REQUIRE(sfod_prod_fits({Cout > Cin ? Cout : Cin, Cin > CinPad ? Cin : CinPad, ksize, ksize}, 1LL << 40),
               "pack_conv_weight: oversized tensor");
  const int rows = rot180 ? Cin : Cout;
  const int64_t total = (int64_t)rows * ksize * ksize * CinPad;
  hipStream_t s = (hipStream_t)stream;
  SFOD_REQUIRE(absmax == nullptr || dt == SFOD_F16X3, "pack_conv_weight: the per-tensor scale belongs to SFOD_F16X3");
  if (absmax != nullptr) {
    const int rc = launch_weight_absmax(w_oihw, (int64_t)Cout * Cin * ksize * ksize, nullptr, 1, absmax, s);
    if (rc) return rc;
  }
  if (sfod_is_pairs(dt)) {
    SFOD_REQUIRE(CinPad % 8 == 0, "pack_conv_weight: operand pairs need an inner size that is a multiple of 8");
    if (dt == SFOD_F16X3)
      hipLaunchKernelGGL(k_pack_conv_weight<splith_t>, dim3(ew_grid(total)), dim3(256), 0, s,
                         w_oihw, (splith_t*)w_packed, Cout, Cin, ksize, CinPad, rot180, (const unsigned*)absmax);
    else
      hipLaunchKernelGGL(k_pack_conv_weight<split_t>, dim3(ew_grid(total)), dim3(256), 0, s,
                         w_oihw, (split_t*)w_packed, Cout, Cin, ksize, CinPad, rot180, (const unsigned*)nullptr);
  } else if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_pack_conv_weight<float>, dim3(ew_grid(total)), dim3(256), 0, s,
                       w_oihw, (float*)w_packed, Cout, Cin, ksize, CinPad, rot180, (const unsigned*)nullptr);
  else
    hipLaunchKernelGGL(k_pack_conv_weight<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s,
                       w_oihw, (bf16_t*)w_packed, Cout, Cin, ksize, CinPad, rot180, (const unsigned*)nullptr);
  return sfod_check_launch("pack_conv_weight");
}

extern "C" int sfod_pack_conv_weight(const float* w_oihw, void* w_packed, int Cout, int Cin, int ksize,
                                     int CinPad, int rot180, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("pack_conv_weight", Cout, Cin, ksize, CinPad);
  return sfod_pack_conv_weight_ws(w_oihw, w_packed, nullptr, Cout, Cin, ksize, CinPad, rot180, dt, stream);
}

// All conv weights of a model in ONE launch.  desc: n entries of 8 int64
// {src (fp32 OIHW), dst (packed, dt), Cout, Cin, ks, innerPad, rot180, first_block}; entry e owns the
// workgroups [first_block[e], first_block[e+1]) (desc[n*8 + 7] = total).  A workgroup moves one 32 (co) x 32 (ci)
// x taps tile through LDS: the OIHW side is read in contiguous runs of 32 x taps floats per output channel, the
// packed side ([co][tap][ci] or, rotated, [ci][tap'][co]) is written in 64-byte runs -- the elementwise
// version gathered fp32 words at a stride of taps (or Cin x taps) floats and ran at 1.2 TB/s.
#define PACK_T 32
template <typename T>
__global__ void __launch_bounds__(256)
k_pack_conv_weights_multi(const long long* __restrict__ desc, int n, const unsigned* __restrict__ amax) {
  __shared__ float tile[PACK_T][PACK_T * 9 + 1];
  // the entry that owns this workgroup: last e with first_block[e] <= blockIdx.x (binary search over uniform scalar
  // loads; the linear scan cost ResNet-101's 200-entry table 3 ms per launch)
  int e = 0, hi = n - 1;
  while (e < hi) {
    const int mid = (e + hi + 1) >> 1;
    if ((int)desc[mid * 8 + 7] <= (int)blockIdx.x) e = mid; else hi = mid - 1;
  }
  const float* __restrict__ w = reinterpret_cast<const float*>(desc[e * 8 + 0]);
  T* __restrict__ out = reinterpret_cast<T*>(desc[e * 8 + 1]);
  const int Cout = (int)desc[e * 8 + 2], Cin = (int)desc[e * 8 + 3], ks = (int)desc[e * 8 + 4];
  const int innerPad = (int)desc[e * 8 + 5], rot180 = (int)desc[e * 8 + 6];
  const float ws = amax != nullptr ? wscale_from_absmax(amax[e]) : 1.0f;   // SFOD_F16X3: per-tensor power of two (common.h)
  // tile grid: rows of the packed tensor x inner (padded) axis
  const int rowsN = rot180 ? Cin : Cout;
  const int tiles_inner = (innerPad + PACK_T - 1) / PACK_T;
  const int tb = (int)blockIdx.x - (int)desc[e * 8 + 7];
  const int r0 = (tb / tiles_inner) * PACK_T, i0 = (tb % tiles_inner) * PACK_T;   // packed row / inner origin
  const int co0 = rot180 ? i0 : r0, ci0 = rot180 ? r0 : i0;
  const int nco = min(PACK_T, Cout - co0), nci = min(PACK_T, Cin - ci0);           // may be <= 0 in the padding
  auto body = [&](auto tc) {
    constexpr int taps = decltype(tc)::value;      // compile-time: the index arithmetic below is all constant divisions
    const int run = max(nci, 0) * taps;            // contiguous floats per output channel
    for (int idx = threadIdx.x; idx < PACK_T * PACK_T * taps; idx += 256) {
      const int col = idx / (PACK_T * taps), j = idx - col * (PACK_T * taps);
      float v = 0.f;
      if (col < nco && j < run) v = w[((int64_t)(co0 + col) * Cin + ci0) * taps + j];
      tile[col][j] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < PACK_T * taps * PACK_T; idx += 256) {
      const int il = idx % PACK_T;                 // inner index (fastest: contiguous in the output)
      const int tap = (idx / PACK_T) % taps;
      const int rl = idx / (PACK_T * taps);        // packed row inside the tile
      const int row = r0 + rl, inner = i0 + il;
      if (row >= rowsN || inner >= innerPad) continue;
      const int col = rot180 ? il : rl, cl = rot180 ? rl : il;
      const int st = rot180 ? (taps - 1 - tap) : tap;
      put_elem<T>(out, ((int64_t)row * taps + tap) * innerPad + inner, tile[col][cl * taps + st] * ws);
    }
  };
  if (ks == 3) body(std::integral_constant<int, 9>{});
  else body(std::integral_constant<int, 1>{});
}

extern "C" int sfod_pack_conv_weights_blocks(int Cout, int Cin, int ksize, int innerPad, int rot180) {
  if (!sfod_ints_ok({Cout, Cin, ksize, innerPad})) return 0;      // hostile extents: not served / nothing
  if (!sfod_prod_fits({(rot180 ? Cin : Cout) / PACK_T + 1, innerPad / PACK_T + 1})) return 0;
  (void)ksize;
  const int rowsN = rot180 ? Cin : Cout;
  return ((rowsN + PACK_T - 1) / PACK_T) * ((innerPad + PACK_T - 1) / PACK_T);
}

extern "C" int sfod_pack_conv_weights_multi_ws(const int64_t* desc, int n, int total_blocks, int dt, uint32_t* absmax,
                                               void* stream) {
  SFOD_REQUIRE_EXTENTS("pack_conv_weights_multi_ws", n, total_blocks);
  SFOD_REQUIRE(n >= 1 && total_blocks >= 1, "pack_multi: empty table (kernel sizes 1 and 3 only)");
  SFOD_REQUIRE(absmax == nullptr || dt == SFOD_F16X3, "pack_multi: the per-tensor scales belong to SFOD_F16X3");
  hipStream_t s = (hipStream_t)stream;
  if (absmax != nullptr) {
    const int rc = launch_weight_absmax(nullptr, 0, (const long long*)desc, n, absmax, s);
    if (rc) return rc;
  }
  const unsigned* am = absmax;
  const unsigned* none = nullptr;
  if (dt == SFOD_BF16X3)    // every innerPad of the table must be a multiple of 8 (caller's contract)
    hipLaunchKernelGGL(k_pack_conv_weights_multi<split_t>, dim3(total_blocks), dim3(256), 0, s, (const long long*)desc, n, none);
  else if (dt == SFOD_F16X3)
    hipLaunchKernelGGL(k_pack_conv_weights_multi<splith_t>, dim3(total_blocks), dim3(256), 0, s, (const long long*)desc, n, am);
  else if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_pack_conv_weights_multi<float>, dim3(total_blocks), dim3(256), 0, s, (const long long*)desc, n, none);
  else
    hipLaunchKernelGGL(k_pack_conv_weights_multi<bf16_t>, dim3(total_blocks), dim3(256), 0, s, (const long long*)desc, n, none);
  return sfod_check_launch("pack_conv_weights_multi");
}

extern "C" int sfod_pack_conv_weights_multi(const int64_t* desc, int n, int total_blocks, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("pack_conv_weights_multi", n, total_blocks);
  return sfod_pack_conv_weights_multi_ws(desc, n, total_blocks, dt, nullptr, stream);
}

__global__ void k_unpack_conv_wgrad(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin,
                                    int ks, int CinPad, int accumulate) {
  const int taps = ks * ks;
  const int64_t total = (int64_t)Cout * Cin * taps;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(t % taps);
    const int ci = (int)((t / taps) % Cin);
    const int co = (int)(t / ((int64_t)taps * Cin));
    const float v = dwp[((int64_t)co * taps + tap) * CinPad + ci];
    dw[t] = accumulate ? dw[t] + v : v;
  }
}

extern "C" int sfod_unpack_conv_wgrad(const float* dw_packed, float* dw_oihw, int Cout, int Cin, int ksize,
                                      int CinPad, int accumulate, void* stream) {
  SFOD_REQUIRE_EXTENTS("unpack_conv_wgrad", Cout, Cin, ksize, CinPad);
  SFOD_REQUIRE(sfod_prod_fits({Cout, Cin > CinPad ? Cin : CinPad, ksize, ksize}, 1LL << 40), "unpack_conv_wgrad: oversized tensor");
  const int64_t total = (int64_t)Cout * Cin * ksize * ksize;
  hipLaunchKernelGGL(k_unpack_conv_wgrad, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dw_packed,
                     dw_oihw, Cout, Cin, ksize, CinPad, accumulate);
  return sfod_check_launch("unpack_conv_wgrad");
}

// nn.Linear weight [N][K]; K axis optionally permuted (c,p)->(p,c); optional transpose.
template <typename T>
__global__ void k_pack_fc_weight(const float* __restrict__ w, T* __restrict__ out, int N, int K, int chw_c,
                                 int transpose, int ld, const unsigned* __restrict__ amax) {
  const float ws = amax != nullptr ? wscale_from_absmax(*amax) : 1.0f;
  const int rows = transpose ? K : N;
  const int64_t total = (int64_t)rows * ld;
  const int PP = chw_c > 0 ? K / chw_c : 1;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int inner = (int)(t % ld);
    const int row = (int)(t / ld);
    const int n = transpose ? inner : row;
    const int kp = transpose ? row : inner;  // permuted k index
    float v = 0.f;
    if (n < N && kp < K) {
      int k = kp;
      if (chw_c > 0) { const int p = kp / chw_c, c = kp % chw_c; k = c * PP + p; }
      v = w[(int64_t)n * K + k];
    }
    put_elem<T>(out, t, v * ws);
  }
}

extern "C" int sfod_pack_fc_weight(const float* w, void* out, int N, int K, int chw_c, int transpose,
                                   int dt, void* stream);

// fc1-shaped weights (K = C x PP in (c, p) order in the state dict, (p, c) in the kernels): the naive gather
// above reads with a stride of PP elements.  These two kernels go through an LDS tile so that both the read
// (contiguous (c, p) runs of one output row) and the write (contiguous c / contiguous n) are coalesced.
template <typename T>
__global__ void __launch_bounds__(256)
k_pack_fc_chw(const float* __restrict__ w, T* __restrict__ out, int N, int C, int PP, int ld,
              const unsigned* __restrict__ amax) {
  const float ws = amax != nullptr ? wscale_from_absmax(*amax) : 1.0f;
  // block = (row n, 64 channels): tile[c][p], out[n][p*C + c]
  extern __shared__ float tile[];     // [64][PP]
  const int n = blockIdx.y, c0 = blockIdx.x * 64;
  const int cw = min(64, C - c0);
  const float* src = w + (int64_t)n * C * PP + (int64_t)c0 * PP;
  for (int i = threadIdx.x; i < cw * PP; i += 256) tile[i] = src[i];
  __syncthreads();
  T* dst = out + (int64_t)n * ld;
  for (int j = threadIdx.x; j < PP * 64; j += 256) {
    const int p = j >> 6, cl = j & 63;
    if (cl < cw) put_elem<T>(dst, (int64_t)p * C + c0 + cl, tile[cl * PP + p] * ws);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_pack_fc_chw_t(const float* __restrict__ w, T* __restrict__ out, int N, int C, int PP, int ld,
                const unsigned* __restrict__ amax) {
  const float ws = amax != nullptr ? wscale_from_absmax(*amax) : 1.0f;
  // block = (64 rows n, 4 channels): tile[n][c][p], out[(p*C + c)][n]
  extern __shared__ float tile[];     // [64][4*PP + 1]
  const int n0 = blockIdx.y * 64, c0 = blockIdx.x * 4;
  const int nw = min(64, N - n0), cw = min(4, C - c0);
  const int run = cw * PP, pitch = 4 * PP + 1;
  for (int i = threadIdx.x; i < nw * run; i += 256) {
    const int nl = i / run, r = i - nl * run;
    tile[nl * pitch + r] = w[(int64_t)(n0 + nl) * C * PP + (int64_t)c0 * PP + r];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < run * 64; j += 256) {
    const int r = j >> 6, nl = j & 63;          // r = cl * PP + p
    if (nl < nw) {
      const int cl = r / PP, p = r - cl * PP;
      put_elem<T>(out, ((int64_t)p * C + c0 + cl) * ld + n0 + nl, tile[nl * pitch + r] * ws);
    }
  }
}

extern "C" int sfod_pack_fc_weight_ld_ws(const float* w, void* out, uint32_t* absmax, int N, int K, int chw_c,
                                         int transpose, int ld, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("pack_fc_weight_ld_ws", N, K, chw_c, ld);
  SFOD_REQUIRE(ld >= (transpose ? N : K), "pack_fc_weight: ld too small");
  SFOD_REQUIRE(!sfod_is_pairs(dt) || ld % 8 == 0, "pack_fc_weight: operand pairs need ld % 8 == 0");
  SFOD_REQUIRE(absmax == nullptr || dt == SFOD_F16X3, "pack_fc_weight: the per-tensor scale belongs to SFOD_F16X3");
  hipStream_t s = (hipStream_t)stream;
  if (absmax != nullptr) {
    const int rc = launch_weight_absmax(w, (int64_t)N * K, nullptr, 1, absmax, s);
    if (rc) return rc;
  }
  const unsigned* am = absmax;
  const unsigned* none = nullptr;
  if (chw_c > 0 && K % chw_c == 0 && K / chw_c <= 64 && (int64_t)N * K >= (1 << 20) &&
      ld == (transpose ? N : K)) {
    const int C = chw_c, PP = K / chw_c;
    if (!transpose) {
      const dim3 grid(cdiv(C, 64), N);
      const size_t lds = (size_t)64 * PP * 4;
      if (dt == SFOD_F32) hipLaunchKernelGGL(k_pack_fc_chw<float>, grid, dim3(256), lds, s, w, (float*)out, N, C, PP, ld, none);
      else if (dt == SFOD_BF16X3) hipLaunchKernelGGL(k_pack_fc_chw<split_t>, grid, dim3(256), lds, s, w, (split_t*)out, N, C, PP, ld, none);
      else if (dt == SFOD_F16X3) hipLaunchKernelGGL(k_pack_fc_chw<splith_t>, grid, dim3(256), lds, s, w, (splith_t*)out, N, C, PP, ld, am);
      else hipLaunchKernelGGL(k_pack_fc_chw<bf16_t>, grid, dim3(256), lds, s, w, (bf16_t*)out, N, C, PP, ld, none);
    } else {
      const dim3 grid(cdiv(C, 4), cdiv(N, 64));
      const size_t lds = (size_t)64 * (4 * PP + 1) * 4;
      if (dt == SFOD_F32) hipLaunchKernelGGL(k_pack_fc_chw_t<float>, grid, dim3(256), lds, s, w, (float*)out, N, C, PP, ld, none);
      else if (dt == SFOD_BF16X3) hipLaunchKernelGGL(k_pack_fc_chw_t<split_t>, grid, dim3(256), lds, s, w, (split_t*)out, N, C, PP, ld, none);
      else if (dt == SFOD_F16X3) hipLaunchKernelGGL(k_pack_fc_chw_t<splith_t>, grid, dim3(256), lds, s, w, (splith_t*)out, N, C, PP, ld, am);
      else hipLaunchKernelGGL(k_pack_fc_chw_t<bf16_t>, grid, dim3(256), lds, s, w, (bf16_t*)out, N, C, PP, ld, none);
    }
    return sfod_check_launch("pack_fc_weight(chw)");
  }
  const int64_t total = (int64_t)(transpose ? K : N) * ld;
  if (dt == SFOD_BF16X3)
    hipLaunchKernelGGL(k_pack_fc_weight<split_t>, dim3(ew_grid(total)), dim3(256), 0, s, w, (split_t*)out, N, K, chw_c, transpose, ld, none);
  else if (dt == SFOD_F16X3)
    hipLaunchKernelGGL(k_pack_fc_weight<splith_t>, dim3(ew_grid(total)), dim3(256), 0, s, w, (splith_t*)out, N, K, chw_c, transpose, ld, am);
  else if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_pack_fc_weight<float>, dim3(ew_grid(total)), dim3(256), 0, s, w, (float*)out, N, K, chw_c, transpose, ld, none);
  else
    hipLaunchKernelGGL(k_pack_fc_weight<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, s, w, (bf16_t*)out, N, K, chw_c, transpose, ld, none);
  return sfod_check_launch("pack_fc_weight");
}

extern "C" int sfod_pack_fc_weight_ld(const float* w, void* out, int N, int K, int chw_c, int transpose,
                                      int ld, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("pack_fc_weight_ld", N, K, chw_c, ld);
  return sfod_pack_fc_weight_ld_ws(w, out, nullptr, N, K, chw_c, transpose, ld, dt, stream);
}

extern "C" int sfod_pack_fc_weight(const float* w, void* out, int N, int K, int chw_c, int transpose,
                                   int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("pack_fc_weight", N, K, chw_c);
  return sfod_pack_fc_weight_ld(w, out, N, K, chw_c, transpose, transpose ? N : K, dt, stream);
}

__global__ void k_unpack_fc_wgrad(const float* __restrict__ dwp, float* __restrict__ dw, int N, int K,
                                  int chw_c, int ld, int accumulate) {
  const int64_t total = (int64_t)N * K;
  const int PP = chw_c > 0 ? K / chw_c : 1;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(t % K);
    const int n = (int)(t / K);
    int kp = k;
    if (chw_c > 0) { const int c = k / PP, p = k % PP; kp = p * chw_c + c; }
    const float v = dwp[(int64_t)n * ld + kp];
    dw[t] = accumulate ? dw[t] + v : v;
  }
}

__global__ void __launch_bounds__(256)
k_unpack_fc_chw(const float* __restrict__ dwp, float* __restrict__ dw, int C, int PP, int ld, int accumulate) {
  // block = (row n, 64 channels): read dwp[n][p*C + c] (coalesced over c), write dw[n][c*PP + p] (one contiguous run)
  extern __shared__ float tile[];     // [64][PP]
  const int n = blockIdx.y, c0 = blockIdx.x * 64;
  const int cw = min(64, C - c0);
  const float* src = dwp + (int64_t)n * ld;
  for (int j = threadIdx.x; j < PP * 64; j += 256) {
    const int p = j >> 6, cl = j & 63;
    if (cl < cw) tile[cl * PP + p] = src[(int64_t)p * C + c0 + cl];
  }
  __syncthreads();
  float* dst = dw + (int64_t)n * C * PP + (int64_t)c0 * PP;
  for (int i = threadIdx.x; i < cw * PP; i += 256) dst[i] = accumulate ? dst[i] + tile[i] : tile[i];
}

extern "C" int sfod_unpack_fc_wgrad_ld(const float* dw_packed, float* dw, int N, int K, int chw_c, int ld,
                                       int accumulate, void* stream) {
  SFOD_REQUIRE_EXTENTS("unpack_fc_wgrad_ld", N, K, chw_c, ld);
  if (chw_c > 0 && K % chw_c == 0 && K / chw_c <= 64 && (int64_t)N * K >= (1 << 20)) {
    const int C = chw_c, PP = K / chw_c;
    hipLaunchKernelGGL(k_unpack_fc_chw, dim3(cdiv(C, 64), N), dim3(256), (size_t)64 * PP * 4, (hipStream_t)stream,
                       dw_packed, dw, C, PP, ld, accumulate);
    return sfod_check_launch("unpack_fc_wgrad(chw)");
  }
  const int64_t total = (int64_t)N * K;
  hipLaunchKernelGGL(k_unpack_fc_wgrad, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dw_packed,
                     dw, N, K, chw_c, ld, accumulate);
  return sfod_check_launch("unpack_fc_wgrad");
}

extern "C" int sfod_unpack_fc_wgrad(const float* dw_packed, float* dw, int N, int K, int chw_c,
                                    int accumulate, void* stream) {
  SFOD_REQUIRE_EXTENTS("unpack_fc_wgrad", N, K, chw_c);
  return sfod_unpack_fc_wgrad_ld(dw_packed, dw, N, K, chw_c, K, accumulate, stream);
}

// ---------------------------------------------------------------------------------------------
// bias gradient: column sums.  grid = (64-column groups, row slices); 4 waves split a slice's
// rows, lanes are consecutive columns (coalesced); slices are combined with one float atomic per
// column (<= 64 adds per address).
// ---------------------------------------------------------------------------------------------
#define BG_SLICES 64

template <typename T>
__global__ void __launch_bounds__(256)
k_bias_grad(const T* __restrict__ dy, float* __restrict__ db, int M, int N, int ld, int rows_per_slice) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * rows_per_slice, m1 = min(M, m0 + rows_per_slice);
  float acc = 0.f;
  if (n < N)
    for (int m = m0 + wave; m < m1; m += 4) acc += to_f32(dy[(int64_t)m * ld + n]);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && n < N) atomicAdd(db + n, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

extern "C" int sfod_bias_grad(const void* dy, float* db, int M, int N, int ld, int accumulate, int dt,
                              void* stream) {
  SFOD_REQUIRE_EXTENTS("bias_grad", M, N, ld);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) (void)hipMemsetAsync(db, 0, sizeof(float) * N, s);
  if (M == 0) return 0;
  int slices = (M + 63) / 64;
  if (slices > BG_SLICES) slices = BG_SLICES;
  if (sfod_deterministic()) slices = 1;      // one workgroup per 64 columns walks every row: a single add per address
  const int rps = (M + slices - 1) / slices;
  dim3 grid(cdiv(N, 64), cdiv(M, rps));
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_bias_grad<float>, grid, dim3(256), 0, s, (const float*)dy, db, M, N, ld, rps);
  else
    hipLaunchKernelGGL(k_bias_grad<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, db, M, N, ld, rps);
  return sfod_check_launch("bias_grad");
}

// ---------------------------------------------------------------------------------------------
// K20 + K21: fused SGD(momentum, weight decay) + EMA over flat fp32 arrays, 16 B per lane
// ---------------------------------------------------------------------------------------------
// student * (1 - k) + teacher * k with the reference's roundings: two rounded products, then a rounded sum.  The contract
// flag is attached where an operator is WRITTEN, so plain * and + under the pragma (HIP's __fmul_rn / __fadd_rn are plain
// operators compiled under the header's default, contraction allowed, and did fuse into v_fmac in one of the kernels).
__device__ __forceinline__ float ema_mix(float student, float one_minus_keep, float teacher, float keep) {
#pragma clang fp contract(off)
  const float a = student * one_minus_keep;
  const float b = teacher * keep;
  return a + b;
}

__global__ void __launch_bounds__(256)
k_sgd_ema(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ t,
          int64_t n, const float* __restrict__ lr_ptr, float momentum, float wd, float gscale, float keep,
          float one_minus_keep, int first) {
  const float lr = lr_ptr[0];
  const int64_t nvec = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(m)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w};
    const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = gg[k] * gscale + wd * pp[k];
      mm[k] = first ? gr : mm[k] * momentum + gr;
      pp[k] = pp[k] - lr * mm[k];
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    if (t) {
      float4 tv = reinterpret_cast<float4*>(t)[i];
      // the reference's operation order (student * (1 - k) + teacher * k in torch ops): two rounded products, a rounded
      // sum, no contraction -- bit-identical to _update_teacher_model (tests/golden/glue_ref.npz)
      tv.x = ema_mix(pp[0], one_minus_keep, tv.x, keep);
      tv.y = ema_mix(pp[1], one_minus_keep, tv.y, keep);
      tv.z = ema_mix(pp[2], one_minus_keep, tv.z, keep);
      tv.w = ema_mix(pp[3], one_minus_keep, tv.w, keep);
      reinterpret_cast<float4*>(t)[i] = tv;
    }
  }
  // scalar tail
  const int64_t tail0 = nvec * 4;
  if (blockIdx.x == 0 && threadIdx.x < (n - tail0)) {
    const int64_t i = tail0 + threadIdx.x;
    const float gr = g[i] * gscale + wd * p[i];
    const float mm = first ? gr : m[i] * momentum + gr;
    const float pp = p[i] - lr * mm;
    p[i] = pp;
    m[i] = mm;
    if (t) t[i] = ema_mix(pp, one_minus_keep, t[i], keep);
  }
}

extern "C" int sfod_sgd_ema(float* param, const float* grad, float* mom, float* teacher, int64_t n,
                            const float* lr, float momentum, float weight_decay, float grad_scale,
                            float ema_keep, float ema_one_minus_keep, int first_step, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "sgd_ema: negative or oversized extent");
  SFOD_REQUIRE(param != nullptr && grad != nullptr && mom != nullptr && lr != nullptr, "sgd_ema: null argument (param, grad, mom, lr)");
  if (n == 0) return 0;
  // 1 - k is the HOST's double subtraction rounded once (the reference: python float (1 - keep_rate) -> float32 scalar);
  // recomputing it from the float32 k would be off by 1.7e-5 relative for k = 0.9996
  const float omk = ema_one_minus_keep;
  hipLaunchKernelGGL(k_sgd_ema, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, param, grad, mom,
                     teacher, n, lr, momentum, weight_decay, grad_scale, ema_keep, omk, first_step);
  return sfod_check_launch("sgd_ema");
}

__global__ void k_ema(float* __restrict__ t, const float* __restrict__ s, int64_t n, float keep, float omk) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    t[i] = ema_mix(s[i], omk, t[i], keep);
}

extern "C" int sfod_ema(float* teacher, const float* student, int64_t n, float keep, float one_minus_keep,
                        void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "ema: negative or oversized extent");
  SFOD_REQUIRE(n == 0 || (teacher != nullptr && student != nullptr), "ema: null argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_ema, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, teacher, student, n, keep,
                     one_minus_keep);
  return sfod_check_launch("ema");
}

// int64 buffers (num_batches_tracked) in the reference's EMA: s * (1 - k) + t * k in fp32, truncated by the copy back
// into the int64 buffer (source_free_adaptive_teacher.py:583-603 + load_state_dict; SURVEY A.17 iv) -- one launch
// instead of torch's cast / mul / mul / add / cast / copy chain
__global__ void k_ema_i64(long long* __restrict__ t, const long long* __restrict__ s, int n, float keep, float omk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    // torch's operation order: two rounded fp32 products, then a rounded sum (no fused multiply-add: with equal
    // counters the result sits exactly on an integer and one contracted rounding decides the truncation)
    t[i] = (long long)ema_mix((float)s[i], omk, (float)t[i], keep);
  }
}
extern "C" int sfod_ema_i64(int64_t* teacher, const int64_t* student, int n, float keep, float one_minus_keep,
                            void* stream) {
  SFOD_REQUIRE_EXTENTS("ema_i64", n);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_ema_i64, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (long long*)teacher,
                     (const long long*)student, n, keep, one_minus_keep);
  return sfod_check_launch("ema_i64");
}

// The three scalars the trainer logs after the teacher pass (source_free_adaptive_teacher.py:411-423,445-452), one
// workgroup: out[0] = mean over images of the mean detection score (0 for an image without detections),
// out[1] = #(RPN proposals with objectness logit > thr) / B, out[2] = mean pseudo-label count.
__global__ void __launch_bounds__(256)
k_teacher_metrics(const float* __restrict__ det_scores, const int* __restrict__ det_count, int D,
                  const float* __restrict__ rpn_logits, const int* __restrict__ rpn_count, int P,
                  const int* __restrict__ gt_count, int B, float thr, float* __restrict__ out) {
  __shared__ float red[3][256];
  float conf = 0.f, npp = 0.f, ngt = 0.f;
  for (int b = 0; b < B; ++b) {
    const int nd = min(det_count[b], D), np = min(rpn_count[b], P);
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nd; i += 256) s += det_scores[(int64_t)b * D + i];
    for (int i = threadIdx.x; i < np; i += 256) c += rpn_logits[(int64_t)b * P + i] > thr ? 1.f : 0.f;
    conf += s / (float)max(nd, 1);
    npp += c;
    if (threadIdx.x == 0) ngt += (float)gt_count[b];
  }
  red[0][threadIdx.x] = conf; red[1][threadIdx.x] = npp; red[2][threadIdx.x] = ngt;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 3) out[threadIdx.x] = red[threadIdx.x][0] / (float)B;
}
extern "C" int sfod_teacher_metrics(const float* det_scores, const int* det_count, int D, const float* rpn_logits,
                                    const int* rpn_count, int P, const int* gt_count, int B, float thr, float* out,
                                    void* stream) {
  SFOD_REQUIRE_EXTENTS("teacher_metrics", D, P, B);
  SFOD_REQUIRE(B >= 1, "teacher_metrics: B < 1");
  SFOD_REQUIRE(det_scores && det_count && rpn_logits && rpn_count && gt_count && out, "teacher_metrics: null argument");
  hipLaunchKernelGGL(k_teacher_metrics, dim3(1), dim3(256), 0, (hipStream_t)stream, det_scores, det_count, D, rpn_logits,
                     rpn_count, P, gt_count, B, thr, out);
  return sfod_check_launch("teacher_metrics");
}

__global__ void k_fill(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}
extern "C" int sfod_fill_f32(float* p, int64_t n, float v, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "fill_f32: negative or oversized extent");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_fill, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, n, v);
  return sfod_check_launch("fill");
}

template <typename S, typename D>
__global__ void k_cast(const S* __restrict__ s, D* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = from_f32<D>(to_f32(s[i]));
}
// fp32 <-> SFOD_BF16X3 pairs, 8 elements per thread (n a multiple of 8)
template <typename S, typename D>
__global__ void k_cast8(const S* __restrict__ s, D* __restrict__ d, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load_n<S, 8>(s + i * 8, v);
    store_n<D, 8>(d + i * 8, v);
  }
}
// fp32 -> SFOD_F16X3 pairs AND SFOD_BF16X3 pairs in one pass ("f16x3" mode: forward operand + weight-gradient operand)
__global__ void k_cast8_both(const float* __restrict__ s, splith_t* __restrict__ dh, split_t* __restrict__ db, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    load_n<float, 8>(s + i * 8, v);
    store_n<splith_t, 8>(dh + i * 8, v);
    store_n<split_t, 8>(db + i * 8, v);
  }
}
extern "C" int sfod_cast_pairs_both(const float* src, void* dst_f16x3, void* dst_bf16x3, int64_t n, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "cast_pairs_both: negative or oversized extent");
  SFOD_REQUIRE(n % 8 == 0, "cast_pairs_both: operand-pair tensors hold whole 8-element groups");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_cast8_both, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, src, (splith_t*)dst_f16x3,
                     (split_t*)dst_bf16x3, n / 8);
  return sfod_check_launch("cast_pairs_both");
}

extern "C" int sfod_cast(const void* src, void* dst, int64_t n, int src_dt, int dst_dt, void* stream) {
  SFOD_REQUIRE(sfod_i64s_ok({n}), "cast: negative or oversized extent");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(ew_grid(n)), b(256);
  if (sfod_is_pairs(src_dt) || sfod_is_pairs(dst_dt)) {
    SFOD_REQUIRE(n % 8 == 0, "cast: operand-pair tensors hold whole 8-element groups");
    const dim3 g8(ew_grid(n / 8));
#define CAST8(S, D) hipLaunchKernelGGL((k_cast8<S, D>), g8, b, 0, s, (const S*)src, (D*)dst, n / 8)
    if (src_dt == SFOD_F32 && dst_dt == SFOD_BF16X3) CAST8(float, split_t);
    else if (src_dt == SFOD_F32 && dst_dt == SFOD_F16X3) CAST8(float, splith_t);
    else if (src_dt == SFOD_BF16X3 && dst_dt == SFOD_F32) CAST8(split_t, float);
    else if (src_dt == SFOD_F16X3 && dst_dt == SFOD_F32) CAST8(splith_t, float);
    else if (src_dt == SFOD_BF16X3 && dst_dt == SFOD_BF16X3) CAST8(split_t, split_t);
    else if (src_dt == SFOD_F16X3 && dst_dt == SFOD_F16X3) CAST8(splith_t, splith_t);
    else if (src_dt == SFOD_F16X3 && dst_dt == SFOD_BF16X3) CAST8(splith_t, split_t);   // re-split of hi + lo
    else if (src_dt == SFOD_BF16X3 && dst_dt == SFOD_F16X3) CAST8(split_t, splith_t);
    else
      SFOD_REQUIRE(false, "cast: operand pairs convert from / to fp32 and each other only");
#undef CAST8
    return sfod_check_launch("cast");
  }
  if (src_dt == SFOD_F32 && dst_dt == SFOD_BF16)
    hipLaunchKernelGGL((k_cast<float, bf16_t>), g, b, 0, s, (const float*)src, (bf16_t*)dst, n);
  else if (src_dt == SFOD_BF16 && dst_dt == SFOD_F32)
    hipLaunchKernelGGL((k_cast<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, (float*)dst, n);
  else if (src_dt == SFOD_F32 && dst_dt == SFOD_F32)
    hipLaunchKernelGGL((k_cast<float, float>), g, b, 0, s, (const float*)src, (float*)dst, n);
  else
    hipLaunchKernelGGL((k_cast<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
  return sfod_check_launch("cast");
}

// ---- SFOD_F16X3 saturation report (common.h: g_f16_sat) -----------------------------------------------------------------
SFOD_DEFINE_F16_POLL(sfod_f16_poll_elementwise)
extern "C" int sfod_f16x3_poll(uint32_t* word, void* stream) {
  SFOD_REQUIRE(word != nullptr, "f16x3_poll: null word");
  hipStream_t s = (hipStream_t)stream;
  sfod_f16_poll_elementwise(word, s);
  sfod_f16_poll_roi(word, s);
  sfod_f16_poll_first(word, s);
  sfod_f16_poll_stem(word, s);
  return sfod_check_launch("f16x3_poll");
}
