// ROIAlign (torchvision roi_align, aligned=True, adaptive sampling grid) on NHWC features.
// Reference call site: daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:117
// (box_pooler = d2 ROIPooler -> ROIAlign(7, 1/stride, 0, aligned=True)); SURVEY A.11.
//
// Forward: one workgroup per ROI, 16-byte channel vectors across the lanes; the feature map of the
// hot config is 37x75x512 (2.8 MB in bf16) and stays L2-resident, so the [R,49,C] output writes are the
// HBM traffic.  Backward: tiled gather (one owner per gradient element, no atomics) for pooled == 7.
#include "conv_internal.h"
#include <cstdlib>

#define ROI_MAXP 8
#define ROI_MAXP_FWD 16
#define ROI_CBLK_DEFAULT 128
#define ROI_NT_DEFAULT 1

struct Sample {
  int y_low, x_low, y_high, x_high;
  float w1, w2, w3, w4;
  bool ok;
};

__device__ __forceinline__ Sample bilinear_setup(float y, float x, int H, int W) {
  Sample s;
  s.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
  const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
  s.y_low = y_low; s.x_low = x_low; s.y_high = y_high; s.x_high = x_high;
  s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
  return s;
}

struct RoiGeom {
  int b;
  float start_h, start_w, bin_h, bin_w;
  int grid_h, grid_w;
  float count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int pooled) {
  RoiGeom g;
  g.b = (int)roi[0];
  const float offset = 0.5f;  // aligned=True
  g.start_w = roi[1] * scale - offset;
  g.start_h = roi[2] * scale - offset;
  const float end_w = roi[3] * scale - offset, end_h = roi[4] * scale - offset;
  const float roi_w = end_w - g.start_w, roi_h = end_h - g.start_h;
  g.bin_h = roi_h / (float)pooled;
  g.bin_w = roi_w / (float)pooled;
  g.grid_h = (int)ceilf(roi_h / (float)pooled);
  g.grid_w = (int)ceilf(roi_w / (float)pooled);
  const int c = g.grid_h * g.grid_w;
  g.count = (float)(c > 1 ? c : 1);
  return g;
}

// 16-byte vector helpers (8 bf16 / 4 fp32 channels per lane)
template <typename T> struct RVec;
template <> struct RVec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* o) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* o) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
  static __device__ __forceinline__ void store_nt(float* p, const float* o) {
    store16<true>(p, make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])));
  }
};
template <> struct RVec<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(w[i] << 16);
      o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
    union { bf16_t h[8]; uint4 v; } u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u.h[i] = (bf16_t)o[i];
    *reinterpret_cast<uint4*>(p) = u.v;
  }
  static __device__ __forceinline__ void store_nt(bf16_t* p, const float* o) {
    union { bf16_t h[8]; uint4 v; } u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u.h[i] = (bf16_t)o[i];
    store16<true>(p, u.v);
  }
};

// SFOD_BF16X3 features / pooled outputs: 8 logical channels per lane (one 32-byte (8 hi | 8 lo) group)
template <> struct RVec<split_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const split_t* p, float* o) { split_load8(p, o); }
  static __device__ __forceinline__ void store(split_t* p, const float* o) { split_store8(p, o); }
  static __device__ __forceinline__ void store_nt(split_t* p, const float* o) { split_store8<true>(p, o); }
};
template <> struct RVec<splith_t> {     // SFOD_F16X3: the same with half pairs
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const splith_t* p, float* o) { split_load8(p, o); }
  static __device__ __forceinline__ void store(splith_t* p, const float* o) { split_store8(p, o); }
  static __device__ __forceinline__ void store_nt(splith_t* p, const float* o) { split_store8<true>(p, o); }
};

// Forward.  One workgroup per ROI; a lane owns one 16-byte channel vector (NHWC: a bilinear corner
// is one contiguous row segment, so a wavefront reads 1 KiB per corner), the 49 bins are spread over
// the remaining thread groups.  The sampling arithmetic is scalar per bin and identical, operation
// for operation, to torchvision's kernel (and to oracle/csrc/roi_align.c).
template <typename T>
__global__ void __launch_bounds__(256)
k_roi_align_fwd(const T* __restrict__ feat, int H, int W, int C, const float* __restrict__ rois,
                int pooled, float scale, T* __restrict__ out) {
  constexpr int V = RVec<T>::N;
  const int r = blockIdx.x;
  const float* roi = rois + (int64_t)r * 5;
  const int nbins = pooled * pooled;
  T* orow = out + (int64_t)r * nbins * C;
  const int cv = C / V;                               // channel vectors per pixel
  const int clanes = min(cv, (int)blockDim.x);
  const int blanes = blockDim.x / clanes;
  const int cl = threadIdx.x % clanes, bl = threadIdx.x / clanes;
  if (bl >= blanes) return;
  if (roi[0] < 0.f) {  // padding row
    float z[V];
#pragma unroll
    for (int i = 0; i < V; ++i) z[i] = 0.f;
    for (int bin = bl; bin < nbins; bin += blanes)
      for (int c = cl; c < cv; c += clanes) RVec<T>::store(orow + (int64_t)bin * C + c * V, z);
    return;
  }
  const RoiGeom g = roi_geom(roi, scale, pooled);
  const T* fb = feat + (int64_t)g.b * H * W * C;
  for (int bin = bl; bin < nbins; bin += blanes) {
    const int ph = bin / pooled, pw = bin % pooled;
    for (int c = cl; c < cv; c += clanes) {
      float acc[V];
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] = 0.f;
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
          const Sample s = bilinear_setup(yy, xx, H, W);
          if (!s.ok) continue;
          float v1[V], v2[V], v3[V], v4[V];
          RVec<T>::load(fb + ((int64_t)s.y_low * W + s.x_low) * C + c * V, v1);
          RVec<T>::load(fb + ((int64_t)s.y_low * W + s.x_high) * C + c * V, v2);
          RVec<T>::load(fb + ((int64_t)s.y_high * W + s.x_low) * C + c * V, v3);
          RVec<T>::load(fb + ((int64_t)s.y_high * W + s.x_high) * C + c * V, v4);
#pragma unroll
          for (int i = 0; i < V; ++i) acc[i] += s.w1 * v1[i] + s.w2 * v2[i] + s.w3 * v3[i] + s.w4 * v4[i];
        }
      }
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] = acc[i] / g.count;
      RVec<T>::store(orow + (int64_t)bin * C + c * V, acc);
    }
  }
}

// Separable forward (same factorisation as the backward below): out[ph][pw] = sum_py sum_px Ay[ph][py] *
// Ax[pw][px] * F[py][px] / count.  Every footprint pixel is read ONCE per channel pair (the sample-by-sample form
// reads 4 corners per sample: ~5x the L2 -> CU traffic, which bounded it), contracted first with Ay (7 partial
// columns in registers) and then with Ax into the 7 x 7 outputs the thread keeps in registers.  fp32 sums in a
// different order than torchvision's loop: equal within rounding (tests: 1e-5 relative in fp32).
template <typename T> struct Pair;
template <> struct Pair<float> {
  static __device__ __forceinline__ void load(const float* p, float& a, float& b) {
    const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y;
  }
  static __device__ __forceinline__ void store(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
};
template <> struct Pair<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float& a, float& b) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
    a = __uint_as_float(w << 16); b = __uint_as_float(w & 0xffff0000u);
  }
  static __device__ __forceinline__ void store(bf16_t* p, float a, float b) {
    union { bf16_t h[2]; uint32_t v; } u;
    u.h[0] = (bf16_t)a; u.h[1] = (bf16_t)b;
    *reinterpret_cast<uint32_t*>(p) = u.v;
  }
};

template <> struct Pair<split_t> {     // only used to zero-fill padding rows: two logical elements = 8 bytes
  static __device__ __forceinline__ void store(split_t* p, float, float) { *reinterpret_cast<uint2*>(p) = make_uint2(0u, 0u); }
};
template <> struct Pair<splith_t> {
  static __device__ __forceinline__ void store(splith_t* p, float, float) { *reinterpret_cast<uint2*>(p) = make_uint2(0u, 0u); }
};

template <typename T, int P, bool NT>
__global__ void __launch_bounds__(256)
k_roi_align_fwd_sep(const T* __restrict__ feat, int H, int W, int C, const float* __restrict__ rois, float scale,
                    T* __restrict__ out, int R, int ncb) {
  extern __shared__ __attribute__((aligned(16))) float sw[];   // Ay [P][H], Ax [P][W], then int sup[2 * P][2]
  // grid = ncb channel blocks x R boxes, channel-block major: at any time the chip works on ONE slice of C / ncb channels of
  // the feature map, which has to stay in an XCD's 4 MB L2 across the boxes that share it -- with all 1024 channels per
  // workgroup the 16 000 boxes of a teacher pass re-read the 93 MB map ~80x past L2.  Round 6: the slice alone does not do
  // it -- the [R,49,C] output STREAM (3.2 GB per teacher call) passes through the same L2 and evicts the slice between two
  // touches of a line.  128-channel slices (1.45 MB per image) + non-temporal output stores (NT): HBM reads per launch
  // 1596 -> 460 MB (mean of the two configs' teacher calls, PMC: profiles/r6_roi_align_fwd_ab.txt), 1.77 -> 1.68 ms at 1024
  // channels.  The kernel is bound by neither its reads nor its gathers' latency (same file).
  const int r = blockIdx.x % R, cb = blockIdx.x / R;
  const float* roi = rois + (int64_t)r * 5;
  const int tid = threadIdx.x;
  T* orow = out + (int64_t)r * P * P * C;
  const int cblk = C / ncb;          // channels of this workgroup: [cb * cblk, (cb + 1) * cblk)
  if (roi[0] < 0.f) {  // padding row
    for (int i = tid; i < P * P * cblk / 2; i += blockDim.x) {
      const int e = 2 * i, bin = e / cblk, c = e - bin * cblk;
      Pair<T>::store(orow + (int64_t)bin * C + cb * cblk + c, 0.f, 0.f);
    }
    return;
  }
  const RoiGeom g = roi_geom(roi, scale, P);
  float* Ay = sw;
  float* Ax = sw + P * H;
  int* sup = reinterpret_cast<int*>(Ax + P * W);            // [ph] (ylo, yhi), then [pw] (xlo, xhi): support of each row
  for (int i = tid; i < P * (H + W); i += blockDim.x) sw[i] = 0.f;
  __syncthreads();
  if (tid < 2 * P) {
    const bool isy = tid < P;
    const int p = isy ? tid : tid - P;
    const int L = isy ? H : W;
    const int gn = isy ? g.grid_h : g.grid_w;
    const float start = isy ? g.start_h : g.start_w, bin = isy ? g.bin_h : g.bin_w;
    float* row = (isy ? Ay : Ax) + p * L;
    int lo = L, hi = -1;
    for (int i = 0; i < gn; ++i) {
      float v = start + (float)p * bin + ((float)i + .5f) * bin / (float)gn;
      if (v < -1.0f || v > (float)L) continue;
      if (v <= 0.f) v = 0.f;
      int l = (int)v, hgh;
      if (l >= L - 1) { hgh = l = L - 1; v = (float)l; } else hgh = l + 1;
      const float lw = v - (float)l, hw = 1.f - lw;
      row[l] += hw;
      row[hgh] += lw;
      lo = min(lo, l);
      hi = max(hi, hgh);
    }
    sup[2 * tid] = lo;          // lo > hi: this bin row / column has no valid sample
    sup[2 * tid + 1] = hi;
  }
  __syncthreads();
  int xlo = W, xhi = -1;
#pragma unroll
  for (int pw = 0; pw < P; ++pw) { xlo = min(xlo, sup[2 * (P + pw)]); xhi = max(xhi, sup[2 * (P + pw) + 1]); }
  const float inv = 1.f / g.count;
  const T* fb = feat + (int64_t)g.b * H * W * C;
  // a lane owns one 16-byte channel vector (a wavefront reads 1 KiB per pixel); the bin rows are spread over the
  // remaining thread groups like the bins of the sample-by-sample kernel
  constexpr int V = RVec<T>::N;
  const int cv = cblk / V;
  const int clanes = min(cv, (int)blockDim.x);
  const int groups = blockDim.x / clanes;
  const int cl = tid % clanes, grp = tid / clanes;
  if (grp >= groups) return;
  for (int cc = cl; cc < cv; cc += clanes) {
    const int c = cb * cv + cc;
    const T* fc = fb + c * V;
    for (int ph = grp; ph < P; ph += groups) {
      // a bin row only sees the few feature rows its samples touch: contract those with Ay, then spread over pw
      const int ylo = sup[2 * ph], yhi = sup[2 * ph + 1];
      float o[P][V];
#pragma unroll
      for (int pw = 0; pw < P; ++pw)
#pragma unroll
        for (int i = 0; i < V; ++i) o[pw][i] = 0.f;
      // (round 6: batching the rows of a footprint column into 2 / 4 loads in flight and requesting the next column ahead
      // was built and measured -- 1.83 / 2.40 ms against this loop's 1.68 ms at 1024 channels, same box: the registers cost
      // occupancy and the kernel is not waiting on its gathers.  profiles/r6_roi_align_fwd_ab.txt)
      for (int px = xlo; px <= xhi; ++px) {
        float col[V];
#pragma unroll
        for (int i = 0; i < V; ++i) col[i] = 0.f;
        for (int py = ylo; py <= yhi; ++py) {
          float f[V];
          RVec<T>::load(fc + ((int64_t)py * W + px) * C, f);
          const float a = Ay[ph * H + py];
#pragma unroll
          for (int i = 0; i < V; ++i) col[i] = __builtin_fmaf(a, f[i], col[i]);
        }
#pragma unroll
        for (int pw = 0; pw < P; ++pw) {
          const float ax = Ax[pw * W + px];
#pragma unroll
          for (int i = 0; i < V; ++i) o[pw][i] = __builtin_fmaf(ax, col[i], o[pw][i]);
        }
      }
#pragma unroll
      for (int pw = 0; pw < P; ++pw) {
#pragma unroll
        for (int i = 0; i < V; ++i) o[pw][i] *= inv;
        if constexpr (NT) RVec<T>::store_nt(orow + (int64_t)(ph * P + pw) * C + c * V, o[pw]);
        else RVec<T>::store(orow + (int64_t)(ph * P + pw) * C + c * V, o[pw]);
      }
    }
  }
}

// Backward.  The bilinear sampling of one ROI is separable: the weight of feature pixel (py, px)
// in pooled bin (ph, pw) is Ay[ph][py] * Ax[pw][px] / count, with Ay / Ax the 1-D interpolation
// weights of the bin's sample rows / columns summed over the adaptive grid (the validity test of a
// sample factorises the same way).  So instead of scattering 4 corners per sample and channel
// (global float atomics run at only ~1.3 TB/s on this chip and were the whole cost), a workgroup
// (= one ROI) builds Ay [P][H] and Ax [P][W] in LDS, every thread keeps the P x P upstream values of
// its channels in registers, contracts them with Ax and Ay and issues ONE atomic add per footprint
// pixel and channel (a wavefront adds 256 contiguous bytes): ~5x fewer atomics than the scatter.

template <typename T>
__global__ void __launch_bounds__(256)
k_roi_align_bwd(const T* __restrict__ dout, int H, int W, int C, const float* __restrict__ rois,
                int pooled, float scale, float* __restrict__ dfeat) {
  extern __shared__ __attribute__((aligned(16))) float sw[];   // Ay [pooled][H], Ax [pooled][W], then int lim[4]
  const int r = blockIdx.x;
  const float* roi = rois + (int64_t)r * 5;
  if (roi[0] < 0.f) return;
  const int nbins = pooled * pooled;
  const RoiGeom g = roi_geom(roi, scale, pooled);
  float* Ay = sw;
  float* Ax = sw + pooled * H;
  int* lim = reinterpret_cast<int*>(Ax + pooled * W);            // ylo, yhi, xlo, xhi
  const int tid = threadIdx.x;
  for (int i = tid; i < pooled * (H + W); i += blockDim.x) sw[i] = 0.f;
  if (tid == 0) { lim[0] = H; lim[1] = -1; lim[2] = W; lim[3] = -1; }
  __syncthreads();
  if (tid < 2 * pooled) {
    // thread ph builds row ph of Ay, thread pooled + pw row pw of Ax (plain read-modify-write: one owner)
    const bool isy = tid < pooled;
    const int p = isy ? tid : tid - pooled;
    const int L = isy ? H : W;
    const int gn = isy ? g.grid_h : g.grid_w;
    const float start = isy ? g.start_h : g.start_w, bin = isy ? g.bin_h : g.bin_w;
    float* row = (isy ? Ay : Ax) + p * L;
    int lo = L, hi = -1;
    for (int i = 0; i < gn; ++i) {
      float v = start + (float)p * bin + ((float)i + .5f) * bin / (float)gn;
      if (v < -1.0f || v > (float)L) continue;
      if (v <= 0.f) v = 0.f;
      int l = (int)v, hgh;
      if (l >= L - 1) { hgh = l = L - 1; v = (float)l; } else hgh = l + 1;
      const float lw = v - (float)l, hw = 1.f - lw;
      row[l] += hw;
      row[hgh] += lw;
      lo = min(lo, l);
      hi = max(hi, hgh);
    }
    if (hi >= 0) {
      atomicMin(&lim[isy ? 0 : 2], lo);
      atomicMax(&lim[isy ? 1 : 3], hi);
    }
  }
  __syncthreads();
  const int ylo = lim[0], yhi = lim[1], xlo = lim[2], xhi = lim[3];
  if (yhi < ylo || xhi < xlo) return;
  const float inv = 1.f / g.count;
  const T* grow = dout + (int64_t)r * nbins * C;
  float* fb = dfeat + (int64_t)g.b * H * W * C;
  for (int c = tid; c < C; c += blockDim.x) {
    float gv[ROI_MAXP][ROI_MAXP];
#pragma unroll
    for (int ph = 0; ph < ROI_MAXP; ++ph)
#pragma unroll
      for (int pw = 0; pw < ROI_MAXP; ++pw)
        gv[ph][pw] = (ph < pooled && pw < pooled) ? to_f32(grow[(int64_t)(ph * pooled + pw) * C + c]) * inv : 0.f;
    for (int px = xlo; px <= xhi; ++px) {
      float tcol[ROI_MAXP];
#pragma unroll
      for (int ph = 0; ph < ROI_MAXP; ++ph) {
        float a = 0.f;
#pragma unroll
        for (int pw = 0; pw < ROI_MAXP; ++pw)
          if (pw < pooled) a += Ax[pw * W + px] * gv[ph][pw];
        tcol[ph] = a;
      }
      for (int py = ylo; py <= yhi; ++py) {
        float v = 0.f;
#pragma unroll
        for (int ph = 0; ph < ROI_MAXP; ++ph)
          if (ph < pooled) v += Ay[ph * H + py] * tcol[ph];
        if (v != 0.f) atomicAdd(fb + ((int64_t)py * W + px) * C + c, v);
      }
    }
  }
}

// Backward, gather form (pooled == 7).  One workgroup owns an 8x8-pixel tile of one image's gradient map
// for a 256-channel slice: it lists the ROIs whose footprint touches the tile (in ROI order: ballot +
// prefix, so the fp32 summation order is fixed), then for each of them forms the tile-restricted
// separable weights Ay [7][8] / Ax [7][8] (14 threads, double-buffered in LDS) and every thread (one
// channel) accumulates  acc[py][px] += sum_ph Ay[ph][py] * (sum_pw Ax[pw][px] * g[ph][pw])  in
// registers.  Each gradient element has exactly one owner: no atomics, one read-modify-write of the
// tile at the end, and the result is bit-reproducible.  Bin rows / columns whose weights are all zero
// inside the tile are skipped (wave-uniform masks).
constexpr int RT = 8;            // tile edge in feature pixels
constexpr int RT_LIST = 4096;    // ROI ids listed per pass

template <typename T>
__global__ void __launch_bounds__(256)
k_roi_align_bwd_tiled(const T* __restrict__ dout, int H, int W, int C, const float* __restrict__ rois, int R,
                      float scale, float* __restrict__ dfeat) {
  constexpr int P = 7;
  __shared__ int list[RT_LIST];
  __shared__ __attribute__((aligned(16))) float wts[2][2][P][RT];   // [buffer][y|x][bin][pixel]
  __shared__ int masks[2][2];
  __shared__ int wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ntx = (W + RT - 1) / RT;
  const int y0 = (blockIdx.x / ntx) * RT, x0 = (blockIdx.x % ntx) * RT;
  const int b = blockIdx.z;
  const int c = blockIdx.y * 256 + tid;
  const bool cok = c < C;
  float acc[RT][RT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < RT; ++j) acc[i][j] = 0.f;

  // weights of ROI r restricted to the tile -> buffer `buf`; run by threads 0 .. 2P-1
  auto build = [&](int r, int buf) {
    const RoiGeom g = roi_geom(rois + (int64_t)r * 5, scale, P);
    const bool isy = tid < P;
    const int p = isy ? tid : tid - P;
    const int L = isy ? H : W, o = isy ? y0 : x0;
    const int gn = isy ? g.grid_h : g.grid_w;
    const float start = isy ? g.start_h : g.start_w, bin = isy ? g.bin_h : g.bin_w;
    float row[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) row[i] = 0.f;
    for (int i = 0; i < gn; ++i) {
      float v = start + (float)p * bin + ((float)i + .5f) * bin / (float)gn;
      if (v < -1.0f || v > (float)L) continue;
      if (v <= 0.f) v = 0.f;
      int l = (int)v, hgh;
      if (l >= L - 1) { hgh = l = L - 1; v = (float)l; } else hgh = l + 1;
      const float lw = v - (float)l, hw = 1.f - lw;
#pragma unroll
      for (int k = 0; k < RT; ++k) {
        if (l - o == k) row[k] += hw;
        if (hgh - o == k) row[k] += lw;
      }
    }
    bool nz = false;
#pragma unroll
    for (int k = 0; k < RT; ++k) {
      wts[buf][isy ? 0 : 1][p][k] = row[k];
      nz |= row[k] != 0.f;
    }
    if (nz) atomicOr(&masks[buf][isy ? 0 : 1], 1 << p);
  };

  for (int r0 = 0; r0 < R; r0 += RT_LIST) {
    // ---- ROIs of image b touching the tile, in ROI order -----------------------------------------
    int n = 0;
    const int rend = min(R, r0 + RT_LIST);
    for (int rb = r0; rb < rend; rb += 256) {
      const int r = rb + tid;
      bool hit = false;
      if (r < rend) {
        const float* roi = rois + (int64_t)r * 5;
        if (roi[0] >= 0.f && (int)roi[0] == b) {
          const RoiGeom g = roi_geom(roi, scale, P);
          if (g.grid_h > 0 && g.grid_w > 0) {
            // footprint bound: samples lie in [start, start + P*bin], a sample touches floor(v), floor(v)+1
            const float eh = g.start_h + (float)P * g.bin_h, ew = g.start_w + (float)P * g.bin_w;
            const int ylo = (int)fmaxf(g.start_h, 0.f), yhi = min((int)fmaxf(eh, 0.f) + 1, H - 1);
            const int xlo = (int)fmaxf(g.start_w, 0.f), xhi = min((int)fmaxf(ew, 0.f) + 1, W - 1);
            hit = eh >= -1.f && ew >= -1.f && g.start_h <= (float)H && g.start_w <= (float)W &&
                  ylo < y0 + RT && yhi >= y0 && xlo < x0 + RT && xhi >= x0;
          }
        }
      }
      const uint64_t bal = __ballot(hit);
      if (lane == 0) wtot[wv] = __popcll(bal);
      __syncthreads();
      int base = n;
      for (int k = 0; k < wv; ++k) base += wtot[k];
      if (hit) list[base + __popcll(bal & ((1ull << lane) - 1ull))] = r;
      n += wtot[0] + wtot[1] + wtot[2] + wtot[3];
      __syncthreads();
    }
    if (n == 0) continue;

    // ---- accumulate ------------------------------------------------------------------------------
    if (tid < 4) masks[tid >> 1][tid & 1] = 0;
    __syncthreads();
    if (tid < 2 * P) build(list[0], 0);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      const int buf = i & 1;
      const int ym = masks[buf][0], xm = masks[buf][1];
      if (i + 1 < n && tid < 2 * P) build(list[i + 1], buf ^ 1);
      if (ym != 0 && xm != 0 && cok) {
        const int r = list[i];
        const float* roi = rois + (int64_t)r * 5;
        const RoiGeom g = roi_geom(roi, scale, P);
        const float inv = 1.f / g.count;
        const T* grow = dout + (int64_t)r * (P * P) * C + c;
        float gv[P][P];
#pragma unroll
        for (int ph = 0; ph < P; ++ph)
#pragma unroll
          for (int pw = 0; pw < P; ++pw)
            gv[ph][pw] = to_f32(grow[(int64_t)(ph * P + pw) * C]);   // unconditional: all 49 in flight
        float ax[P][RT];
#pragma unroll
        for (int pw = 0; pw < P; ++pw) {
          const float4 a0 = *reinterpret_cast<const float4*>(&wts[buf][1][pw][0]);
          const float4 a1 = *reinterpret_cast<const float4*>(&wts[buf][1][pw][4]);
          ax[pw][0] = a0.x * inv; ax[pw][1] = a0.y * inv; ax[pw][2] = a0.z * inv; ax[pw][3] = a0.w * inv;
          ax[pw][4] = a1.x * inv; ax[pw][5] = a1.y * inv; ax[pw][6] = a1.z * inv; ax[pw][7] = a1.w * inv;
        }
#pragma unroll
        for (int ph = 0; ph < P; ++ph) {
          if (!((ym >> ph) & 1)) continue;
          float t[RT];
#pragma unroll
          for (int px = 0; px < RT; ++px) {
            float a = 0.f;
#pragma unroll
            for (int pw = 0; pw < P; ++pw) a = fmaf(ax[pw][px], gv[ph][pw], a);
            t[px] = a;
          }
          const float4 b0 = *reinterpret_cast<const float4*>(&wts[buf][0][ph][0]);
          const float4 b1 = *reinterpret_cast<const float4*>(&wts[buf][0][ph][4]);
          const float ay[RT] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int py = 0; py < RT; ++py)
#pragma unroll
            for (int px = 0; px < RT; ++px) acc[py][px] = fmaf(ay[py], t[px], acc[py][px]);
        }
      }
      __syncthreads();
      if (tid < 2) masks[buf][tid] = 0;      // buffer `buf` is rebuilt for ROI i+2 after the next barrier
    }
    __syncthreads();
  }
  if (!cok) return;
  float* fb = dfeat + (int64_t)b * H * W * C + c;
#pragma unroll
  for (int py = 0; py < RT; ++py)
#pragma unroll
    for (int px = 0; px < RT; ++px) {
      const int y = y0 + py, x = x0 + px;
      if (y < H && x < W && acc[py][px] != 0.f) fb[((int64_t)y * W + x) * C] += acc[py][px];
    }
}

// SFOD_ROI_BWD_ATOMIC=1 selects the scatter (atomic) form for A/B measurements.
static const bool g_roi_bwd_tiled = []() { const char* e = getenv("SFOD_ROI_BWD_ATOMIC"); return !(e && e[0] == '1'); }();

template <typename T>
static int dispatch_roi_bwd(const void* dout, int B, int H, int W, int C, const float* rois, int R, int pooled,
                            float scale, float* dfeat, hipStream_t s) {
  if (pooled == 7 && g_roi_bwd_tiled) {
    const dim3 grid(((H + RT - 1) / RT) * ((W + RT - 1) / RT), (C + 255) / 256, B);
    hipLaunchKernelGGL(k_roi_align_bwd_tiled<T>, grid, dim3(256), 0, s, (const T*)dout, H, W, C, rois, R, scale,
                       dfeat);
    return sfod_check_launch("roi_align_bwd_tiled");
  }
  const size_t lds = (size_t)pooled * (H + W) * 4 + 16;
  SFOD_REQUIRE(lds <= 64 * 1024, "roi_align_bwd: feature map too large");
  hipLaunchKernelGGL(k_roi_align_bwd<T>, dim3(R), dim3(256), lds, s, (const T*)dout, H, W, C, rois, pooled, scale,
                     dfeat);
  return sfod_check_launch("roi_align_bwd");
}

extern "C" int sfod_roi_align_fwd(const void* feat, int B, int H, int W, int C, const float* rois, int R,
                                  int pooled, float scale, void* out, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("roi_align_fwd", B, H, W, C, R, pooled);
  (void)B;
  if (R == 0) return 0;
  SFOD_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 1, "roi_align: empty feature map");
  // (the backward's register tiles bound it at 8; the forward's bin loop has no such limit: 14 is Detectron2's default
  // POOLER_RESOLUTION, the size its own unit tests' golden losses are computed at -- tests/test_gpu_d2_golden.py)
  SFOD_REQUIRE(pooled >= 1 && pooled <= ROI_MAXP_FWD, "roi_align: pooled size must be in [1, 16]");
  SFOD_REQUIRE(dt == SFOD_F32 || dt == SFOD_BF16 || sfod_is_pairs(dt), "roi_align: unknown dt");
  SFOD_REQUIRE(feat != nullptr && rois != nullptr && out != nullptr, "roi_align: null argument (feat, rois, out)");
  SFOD_REQUIRE(sfod_prod_fits({B, H, W, C}, 1LL << 40) && sfod_prod_fits({R, C, pooled, pooled}, 1LL << 40) &&
               sfod_prod_fits({R, C / 8 + 1}), "roi_align: oversized problem");
  SFOD_REQUIRE(C % ((dt == SFOD_F32) ? 4 : 8) == 0, "roi_align: C must be a multiple of the 16-byte vector");
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)pooled * (H + W) * 4 + 4 * pooled * 4;
  if (pooled == 7 && lds <= 48 * 1024) {       // the configs' POOLER_RESOLUTION: separable form
    // channel blocks (one L2-resident slice of the feature map at a time); SFOD_ROI_CBLK: logical channels per block
    // (A/B; 0: one workgroup per box), SFOD_ROI_NT=1: non-temporal output stores
    static const int cblk_env = []() { const char* e = getenv("SFOD_ROI_CBLK"); return e ? atoi(e) : ROI_CBLK_DEFAULT; }();
    static const int nt = []() { const char* e = getenv("SFOD_ROI_NT"); return e ? atoi(e) : ROI_NT_DEFAULT; }();
    const int V = (dt == SFOD_F32) ? 4 : 8;
    int ncb = 1;
    if (cblk_env > 0 && cblk_env % V == 0 && C % cblk_env == 0 && C > cblk_env) ncb = C / cblk_env;
    const int cblk = C / ncb;
    // threads: one lane per 16-byte (pairs: 32-byte) channel vector of the block x 8 bin-row groups (7 active)
    int threads = (cblk / V) * 8;
    threads = threads < 64 ? 64 : (threads > 256 ? 256 : threads);
    const dim3 grid((unsigned)R * ncb), block((unsigned)threads);
#define SFOD_ROI_SEP(T_, NT_) hipLaunchKernelGGL((k_roi_align_fwd_sep<T_, 7, NT_>), grid, block, lds, s, (const T_*)feat, H, W, C, \
                                                 rois, scale, (T_*)out, R, ncb)
    if (dt == SFOD_F32) { if (nt) SFOD_ROI_SEP(float, true); else SFOD_ROI_SEP(float, false); }
    else if (dt == SFOD_BF16X3) { if (nt) SFOD_ROI_SEP(split_t, true); else SFOD_ROI_SEP(split_t, false); }
    else if (dt == SFOD_F16X3) { if (nt) SFOD_ROI_SEP(splith_t, true); else SFOD_ROI_SEP(splith_t, false); }
    else { if (nt) SFOD_ROI_SEP(bf16_t, true); else SFOD_ROI_SEP(bf16_t, false); }
#undef SFOD_ROI_SEP
    return sfod_check_launch("roi_align_fwd_sep");
  }
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_roi_align_fwd<float>, dim3(R), dim3(256), 0, s, (const float*)feat, H, W, C, rois,
                       pooled, scale, (float*)out);
  else if (dt == SFOD_BF16X3)
    hipLaunchKernelGGL(k_roi_align_fwd<split_t>, dim3(R), dim3(256), 0, s, (const split_t*)feat, H, W, C, rois,
                       pooled, scale, (split_t*)out);
  else if (dt == SFOD_F16X3)
    hipLaunchKernelGGL(k_roi_align_fwd<splith_t>, dim3(R), dim3(256), 0, s, (const splith_t*)feat, H, W, C, rois,
                       pooled, scale, (splith_t*)out);
  else
    hipLaunchKernelGGL(k_roi_align_fwd<bf16_t>, dim3(R), dim3(256), 0, s, (const bf16_t*)feat, H, W, C,
                       rois, pooled, scale, (bf16_t*)out);
  return sfod_check_launch("roi_align_fwd");
}

extern "C" int sfod_roi_align_bwd(const void* dout, int B, int H, int W, int C, const float* rois, int R,
                                  int pooled, float scale, float* dfeat, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("roi_align_bwd", B, H, W, C, R, pooled);
  if (R == 0 || B == 0) return 0;
  SFOD_REQUIRE(C % 8 == 0, "roi_align_bwd: C must be a multiple of 8");
  SFOD_REQUIRE(pooled >= 1 && pooled <= ROI_MAXP, "roi_align_bwd: pooled size must be <= 8");
  SFOD_REQUIRE(dout != nullptr && rois != nullptr && dfeat != nullptr, "roi_align_bwd: null argument (dout, rois, dfeat)");
  SFOD_REQUIRE(sfod_prod_fits({B, H, W, C}, 1LL << 40) && sfod_prod_fits({R, C, pooled, pooled}, 1LL << 40),
               "roi_align_bwd: oversized problem");
  hipStream_t s = (hipStream_t)stream;
  SFOD_REQUIRE(!sfod_is_pairs(dt), "roi_align_bwd: the upstream gradient is fp32 in the operand-pair modes");
  if (dt == SFOD_F32) return dispatch_roi_bwd<float>(dout, B, H, W, C, rois, R, pooled, scale, dfeat, s);
  return dispatch_roi_bwd<bf16_t>(dout, B, H, W, C, rois, R, pooled, scale, dfeat, s);
}

SFOD_DEFINE_F16_POLL(sfod_f16_poll_roi)
