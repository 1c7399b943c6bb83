// ROIAlign (torchvision roi_align, aligned=True, adaptive sampling grid) on NHWC features.
// Reference call site: daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:117
// (box_pooler = d2 ROIPooler -> ROIAlign(7, 1/stride, 0, aligned=True)); SURVEY A.11.
//
// Layout: one workgroup per ROI; channels run across the lanes (NHWC makes every corner read a
// contiguous row segment), pooled bins are spread over the remaining threads.  The feature map
// of the hot config is 18x37x512 (<= 1.4 MB): it stays L2-resident, so corner reads never
// reach HBM; output writes ([R,49,C]) are the HBM traffic.
#include "common.h"

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

template <typename T>
__global__ void __launch_bounds__(256)
k_roi_align_fwd(const T* __restrict__ feat, int H, int W, int C, const float* __restrict__ rois,
                int pooled, float scale, T* __restrict__ out) {
  const int r = blockIdx.x;
  const float* roi = rois + (int64_t)r * 5;
  const int nbins = pooled * pooled;
  T* orow = out + (int64_t)r * nbins * C;
  if (roi[0] < 0.f) {  // padding row
    for (int i = threadIdx.x; i < nbins * C; i += blockDim.x) orow[i] = from_f32<T>(0.f);
    return;
  }
  const RoiGeom g = roi_geom(roi, scale, pooled);
  const T* fb = feat + (int64_t)g.b * H * W * C;
  // thread -> (bin lane, channel lane); channels contiguous across lanes
  const int clanes = min(C, (int)blockDim.x);
  const int blanes = blockDim.x / clanes;
  const int cl = threadIdx.x % clanes, bl = threadIdx.x / clanes;
  if (bl >= blanes) return;
  for (int bin = bl; bin < nbins; bin += blanes) {
    const int ph = bin / pooled, pw = bin % pooled;
    for (int c = cl; c < C; c += clanes) {
      float acc = 0.f;
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
          const Sample s = bilinear_setup(yy, xx, H, W);
          if (!s.ok) continue;
          const float v1 = to_f32(fb[((int64_t)s.y_low * W + s.x_low) * C + c]);
          const float v2 = to_f32(fb[((int64_t)s.y_low * W + s.x_high) * C + c]);
          const float v3 = to_f32(fb[((int64_t)s.y_high * W + s.x_low) * C + c]);
          const float v4 = to_f32(fb[((int64_t)s.y_high * W + s.x_high) * C + c]);
          acc += s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4;
        }
      }
      orow[(int64_t)bin * C + c] = from_f32<T>(acc / g.count);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_roi_align_bwd(const T* __restrict__ dout, int H, int W, int C, const float* __restrict__ rois,
                int pooled, float scale, float* __restrict__ dfeat) {
  const int r = blockIdx.x;
  const float* roi = rois + (int64_t)r * 5;
  if (roi[0] < 0.f) return;
  const int nbins = pooled * pooled;
  const T* grow = dout + (int64_t)r * nbins * C;
  const RoiGeom g = roi_geom(roi, scale, pooled);
  float* fb = dfeat + (int64_t)g.b * H * W * C;
  const int clanes = min(C, (int)blockDim.x);
  const int blanes = blockDim.x / clanes;
  const int cl = threadIdx.x % clanes, bl = threadIdx.x / clanes;
  if (bl >= blanes) return;
  for (int bin = bl; bin < nbins; bin += blanes) {
    const int ph = bin / pooled, pw = bin % pooled;
    for (int c = cl; c < C; c += clanes) {
      const float gval = to_f32(grow[(int64_t)bin * C + c]);
      for (int iy = 0; iy < g.grid_h; ++iy) {
        const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
        for (int ix = 0; ix < g.grid_w; ++ix) {
          const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
          const Sample s = bilinear_setup(yy, xx, H, W);
          if (!s.ok) continue;
          // one dword per lane, 256 contiguous bytes per wave-instruction
          atomicAdd(fb + ((int64_t)s.y_low * W + s.x_low) * C + c, gval * s.w1 / g.count);
          atomicAdd(fb + ((int64_t)s.y_low * W + s.x_high) * C + c, gval * s.w2 / g.count);
          atomicAdd(fb + ((int64_t)s.y_high * W + s.x_low) * C + c, gval * s.w3 / g.count);
          atomicAdd(fb + ((int64_t)s.y_high * W + s.x_high) * C + c, gval * s.w4 / g.count);
        }
      }
    }
  }
}

extern "C" int sfod_roi_align_fwd(const void* feat, int B, int H, int W, int C, const float* rois, int R,
                                  int pooled, float scale, void* out, int dt, void* stream) {
  (void)B;
  if (R == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_roi_align_fwd<float>, dim3(R), dim3(256), 0, s, (const float*)feat, H, W, C, rois,
                       pooled, scale, (float*)out);
  else
    hipLaunchKernelGGL(k_roi_align_fwd<bf16_t>, dim3(R), dim3(256), 0, s, (const bf16_t*)feat, H, W, C,
                       rois, pooled, scale, (bf16_t*)out);
  return sfod_check_launch("roi_align_fwd");
}

extern "C" int sfod_roi_align_bwd(const void* dout, int B, int H, int W, int C, const float* rois, int R,
                                  int pooled, float scale, float* dfeat, int dt, void* stream) {
  (void)B;
  if (R == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dt == SFOD_F32)
    hipLaunchKernelGGL(k_roi_align_bwd<float>, dim3(R), dim3(256), 0, s, (const float*)dout, H, W, C, rois,
                       pooled, scale, dfeat);
  else
    hipLaunchKernelGGL(k_roi_align_bwd<bf16_t>, dim3(R), dim3(256), 0, s, (const bf16_t*)dout, H, W, C,
                       rois, pooled, scale, dfeat);
  return sfod_check_launch("roi_align_bwd");
}
