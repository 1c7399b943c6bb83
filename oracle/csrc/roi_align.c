/* Oracle (TEST INFRASTRUCTURE, never linked into the product library).
 *
 * Plain-C restatement of torchvision.ops.roi_align (CPU kernel algorithm) as reached by the
 * reference through detectron2 ROIPooler -> ROIAlign(aligned=True, sampling_ratio=0):
 *   reference call site  daod/modeling/roi_heads/source_free_adaptive_teacher_roi_heads.py:117
 *   semantics            SURVEY.md Appendix A.11 (torchvision roi_align, un-vendored, un-pinned)
 * PARITY UNPINNED: torchvision is not installed in the build container; this follows its
 * published algorithm (pre-computed bilinear samples, adaptive grid = ceil(roi / pooled)).
 *
 * Layout: input NCHW fp32, rois [R,5] = (batch_idx, x1, y1, x2, y2), output [R,C,PH,PW].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int pos1, pos2, pos3, pos4;
  float w1, w2, w3, w4;
} precalc_t;

static void pre_calc(int height, int width, int ph_n, int pw_n, float start_h, float start_w,
                     float bin_h, float bin_w, int grid_h, int grid_w, precalc_t* pc) {
  int idx = 0;
  for (int ph = 0; ph < ph_n; ph++)
    for (int pw = 0; pw < pw_n; pw++)
      for (int iy = 0; iy < grid_h; iy++) {
        const float yy = start_h + ph * bin_h + (iy + .5f) * bin_h / (float)grid_h;
        for (int ix = 0; ix < grid_w; ix++) {
          const float xx = start_w + pw * bin_w + (ix + .5f) * bin_w / (float)grid_w;
          float x = xx, y = yy;
          precalc_t* p = &pc[idx++];
          if (y < -1.0f || y > height || x < -1.0f || x > width) {
            memset(p, 0, sizeof(*p));
            continue;
          }
          if (y <= 0) y = 0;
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; }
          else y_high = y_low + 1;
          if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; }
          else x_high = x_low + 1;
          const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
          p->pos1 = y_low * width + x_low;
          p->pos2 = y_low * width + x_high;
          p->pos3 = y_high * width + x_low;
          p->pos4 = y_high * width + x_high;
          p->w1 = hy * hx; p->w2 = hy * lx; p->w3 = ly * hx; p->w4 = ly * lx;
        }
      }
}

static void roi_geom(const float* roi, float scale, int aligned, int ph_n, int pw_n,
                     int sampling_ratio, float* start_h, float* start_w, float* bin_h,
                     float* bin_w, int* grid_h, int* grid_w) {
  const float offset = aligned ? 0.5f : 0.0f;
  *start_w = roi[1] * scale - offset;
  *start_h = roi[2] * scale - offset;
  const float end_w = roi[3] * scale - offset;
  const float end_h = roi[4] * scale - offset;
  float roi_w = end_w - *start_w, roi_h = end_h - *start_h;
  if (!aligned) { roi_w = fmaxf(roi_w, 1.f); roi_h = fmaxf(roi_h, 1.f); }
  *bin_h = roi_h / (float)ph_n;
  *bin_w = roi_w / (float)pw_n;
  *grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / ph_n);
  *grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / pw_n);
}

/* forward */
void oracle_roi_align_fwd(const float* input, const float* rois, float* output, int R, int C,
                          int H, int W, int PH, int PW, float scale, int sampling_ratio,
                          int aligned) {
  for (int n = 0; n < R; n++) {
    const float* roi = rois + 5 * n;
    const int b = (int)roi[0];
    float sh, sw, bh, bw; int gh, gw;
    roi_geom(roi, scale, aligned, PH, PW, sampling_ratio, &sh, &sw, &bh, &bw, &gh, &gw);
    const int cnt_i = gh * gw;
    const float count = (float)(cnt_i > 1 ? cnt_i : 1);
    const size_t npc = (size_t)(gh > 0 ? gh : 0) * (gw > 0 ? gw : 0) * PH * PW;
    precalc_t* pc = (precalc_t*)malloc(sizeof(precalc_t) * (npc ? npc : 1));
    if (gh > 0 && gw > 0) pre_calc(H, W, PH, PW, sh, sw, bh, bw, gh, gw, pc);
    for (int c = 0; c < C; c++) {
      const float* in = input + ((size_t)b * C + c) * H * W;
      float* out = output + ((size_t)n * C + c) * PH * PW;
      size_t idx = 0;
      for (int ph = 0; ph < PH; ph++)
        for (int pw = 0; pw < PW; pw++) {
          float val = 0.f;
          for (int iy = 0; iy < gh; iy++)
            for (int ix = 0; ix < gw; ix++) {
              const precalc_t* p = &pc[idx++];
              val += p->w1 * in[p->pos1] + p->w2 * in[p->pos2] + p->w3 * in[p->pos3] +
                     p->w4 * in[p->pos4];
            }
          out[ph * PW + pw] = val / count;
        }
    }
    free(pc);
  }
}

/* backward: grad_input must be zero-initialised by the caller */
void oracle_roi_align_bwd(const float* grad_out, const float* rois, float* grad_in, int R, int C,
                          int H, int W, int PH, int PW, float scale, int sampling_ratio,
                          int aligned) {
  for (int n = 0; n < R; n++) {
    const float* roi = rois + 5 * n;
    const int b = (int)roi[0];
    float sh, sw, bh, bw; int gh, gw;
    roi_geom(roi, scale, aligned, PH, PW, sampling_ratio, &sh, &sw, &bh, &bw, &gh, &gw);
    const int cnt_i = gh * gw;
    const float count = (float)(cnt_i > 1 ? cnt_i : 1);
    for (int c = 0; c < C; c++) {
      float* gi = grad_in + ((size_t)b * C + c) * H * W;
      const float* go = grad_out + ((size_t)n * C + c) * PH * PW;
      for (int ph = 0; ph < PH; ph++)
        for (int pw = 0; pw < PW; pw++) {
          const float g = go[ph * PW + pw];
          for (int iy = 0; iy < gh; iy++) {
            const float yy = sh + ph * bh + (iy + .5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ix++) {
              const float xx = sw + pw * bw + (ix + .5f) * bw / (float)gw;
              float x = xx, y = yy;
              if (y < -1.0f || y > H || x < -1.0f || x > W) continue;
              if (y <= 0) y = 0;
              if (x <= 0) x = 0;
              int y_low = (int)y, x_low = (int)x, y_high, x_high;
              if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; }
              else y_high = y_low + 1;
              if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; }
              else x_high = x_low + 1;
              const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
              gi[y_low * W + x_low] += g * (hy * hx) / count;
              gi[y_low * W + x_high] += g * (hy * lx) / count;
              gi[y_high * W + x_low] += g * (ly * hx) / count;
              gi[y_high * W + x_high] += g * (ly * lx) / count;
            }
          }
        }
    }
  }
}
