// Strong augmentation on the device: the torchvision-on-Pillow transforms of
// daod/data/detection_utils.py:7-36 applied by daod/data/mappers/two_crop_augmentation_mapper.py:141-146, on
// planar uint8 [3][H][W] frames.  Bit-exact with Pillow / torch's CPU casts (oracle/augment.py pins every
// formula against Pillow, RGB<->HSV over all 2^24 inputs); built with -ffp-contract=off because Pillow's
// x86-64 build rounds every float operation separately.  All of it is HBM-bound byte work: one read and one
// write of a 2.2 MB frame per launch.
#include "common.h"

namespace {

constexpr int AUG_MAX_OPS = 8;
enum { OP_BRIGHTNESS = 0, OP_CONTRAST = 1, OP_SATURATION = 2, OP_HUE = 3, OP_GRAYSCALE = 4 };

struct AugOps {
  int n;
  int code[AUG_MAX_OPS];
  float factor[AUG_MAX_OPS];
  int hshift[AUG_MAX_OPS];   // hue: uint8(hue_factor * 255), wrapped (computed by the caller in double)
};

struct Px { int r, g, b; };

// Pillow convert("L"): ITU-R 601-2 luma, 16-bit fixed point
__device__ __forceinline__ int luma(Px p) { return (p.r * 19595 + p.g * 38470 + p.b * 7471 + 0x8000) >> 16; }

// Pillow Image.blend on uint8 (float32; truncation inside [0,1], clip + truncation outside)
__device__ __forceinline__ int blend1(int deg, int v, float a, bool inside) {
  const float t = (float)deg + a * ((float)v - (float)deg);
  if (inside) return (int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ Px blend(Px deg, Px v, float a) {
  const bool inside = a >= 0.f && a <= 1.f;
  return Px{blend1(deg.r, v.r, a, inside), blend1(deg.g, v.g, a, inside), blend1(deg.b, v.b, a, inside)};
}

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Pillow Convert.c rgb2hsv_row: float variables, expressions with double literals
__device__ __forceinline__ Px rgb2hsv(Px p) {
  const int maxc = max(p.r, max(p.g, p.b)), minc = min(p.r, min(p.g, p.b));
  if (minc == maxc) return Px{0, 0, maxc};
  const float cr = (float)(maxc - minc);
  const float s = cr / (float)maxc;
  const float rc = (float)(maxc - p.r) / cr, gc = (float)(maxc - p.g) / cr, bc = (float)(maxc - p.b) / cr;
  float h;
  if (p.r == maxc) h = (float)((double)bc - (double)gc);
  else if (p.g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  return Px{clip8((int)((double)h * 255.0)), clip8((int)((double)s * 255.0)), maxc};
}

// Pillow Convert.c hsv2rgb_row; round() = half away from zero (arguments are >= 0)
__device__ __forceinline__ Px hsv2rgb(Px q) {
  const int v = q.b;
  if (q.g == 0) return Px{v, v, v};
  const double hh = (double)(float)q.r * 6.0 / 255.0;
  const int i = (int)floor(hh);
  const float f = (float)(hh - (double)i);
  const float fs = (float)((double)(float)q.g / 255.0);
  const double vd = (double)(float)v, fsd = (double)fs, fd = (double)f;
  const int p = clip8((int)floor(vd * (1.0 - fsd) + 0.5));
  const int qq = clip8((int)floor(vd * (1.0 - fsd * fd) + 0.5));
  const int t = clip8((int)floor(vd * (1.0 - fsd * (1.0 - fd)) + 0.5));
  switch (i % 6) {
    case 0: return Px{v, t, p};
    case 1: return Px{qq, v, p};
    case 2: return Px{p, v, t};
    case 3: return Px{p, qq, v};
    case 4: return Px{t, p, v};
    default: return Px{v, p, qq};
  }
}

// ops [first, last) on one pixel; `mean` is the contrast op's flat level (valid when a contrast op is in range)
__device__ __forceinline__ Px apply_ops(Px p, const AugOps& o, int first, int last, int mean) {
  for (int k = first; k < last; ++k) {
    const float a = o.factor[k];
    switch (o.code[k]) {
      case OP_BRIGHTNESS: p = blend(Px{0, 0, 0}, p, a); break;
      case OP_CONTRAST: p = blend(Px{mean, mean, mean}, p, a); break;
      case OP_SATURATION: { const int l = luma(p); p = blend(Px{l, l, l}, p, a); break; }
      case OP_HUE: { Px h = rgb2hsv(p); h.r = (h.r + o.hshift[k]) & 255; p = hsv2rgb(h); break; }
      default: { const int l = luma(p); p = Px{l, l, l}; break; }
    }
  }
  return p;
}

// sum of the luma of the image after ops [0, upto): the contrast op's ImageStat mean (exact integer sum)
__global__ void __launch_bounds__(256)
k_aug_luma_sum(const uint8_t* __restrict__ in, int64_t n, AugOps o, int upto, unsigned long long* __restrict__ sum) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    Px p{in[i], in[n + i], in[2 * n + i]};
    p = apply_ops(p, o, 0, upto, 0);
    acc += (unsigned long long)luma(p);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);   // one atomic per workgroup
}

__global__ void __launch_bounds__(256)
k_aug_color(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int64_t n, AugOps o,
            const unsigned long long* __restrict__ sum) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int mean = 0;
  if (sum) mean = (int)((double)*sum / (double)n + 0.5);     // int(ImageStat.Stat(L).mean[0] + 0.5)
  Px p{in[i], in[n + i], in[2 * n + i]};
  p = apply_ops(p, o, 0, o.n, mean);
  out[i] = (uint8_t)p.r;
  out[n + i] = (uint8_t)p.g;
  out[2 * n + i] = (uint8_t)p.b;
}

// One extended-box pass (Pillow BoxBlur.c ImagingLineBoxBlur8) along x (sx = 1, line stride W) or along y
// (sx = W): 2*radius+1 inner taps of weight ww, two outer taps of weight fw, 8.24 fixed point, edges repeat the
// border pixel.  radius <= 2 for the configured sigma range, so the direct sum is cheaper than a running one.
__global__ void __launch_bounds__(256)
k_aug_box_pass(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int C, int H, int W, int along_y,
               int radius, unsigned ww, unsigned fw) {
  const int64_t n = (int64_t)C * H * W;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const int pos = along_y ? y : x, len = along_y ? H : W;
  const int64_t step = along_y ? W : 1;
  const uint8_t* line = in + (i - (int64_t)pos * step);
  unsigned acc = 0;
  for (int k = -radius; k <= radius; ++k) acc += line[(int64_t)min(max(pos + k, 0), len - 1) * step];
  const unsigned far = (unsigned)line[(int64_t)max(pos - radius - 1, 0) * step] +
                       (unsigned)line[(int64_t)min(pos + radius + 1, len - 1) * step];
  out[i] = (uint8_t)((acc * ww + far * fw + (1u << 23)) >> 24);
}

// RandomErasing(value="random") + ToPILImage: the N(0,1) fill goes through mul(255).byte()
__global__ void __launch_bounds__(256)
k_aug_erase(uint8_t* __restrict__ img, int C, int H, int W, int i0, int j0, int h, int w,
            const float* __restrict__ noise) {
  const int64_t n = (int64_t)C * h * w;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int x = (int)(t % w), y = (int)((t / w) % h), c = (int)(t / ((int64_t)w * h));
  const float v = noise[t] * 255.f;
  img[((int64_t)c * H + i0 + y) * W + j0 + x] = (uint8_t)((long long)v & 255);
}

// Pillow BoxBlur.c _gaussian_blur_radius (all float)
float gaussian_box_radius(float radius, int passes) {
  float sigma2, L, l, a;
  sigma2 = radius * radius / passes;
  L = sqrt(12.0 * sigma2 + 1.0);
  l = floor((L - 1.0) / 2.0);
  a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
  a /= 6 * (sigma2 - (l + 1) * (l + 1));
  return l + a;
}

}  // namespace

extern "C" int sfod_aug_color(const uint8_t* in, uint8_t* out, int H, int W, int n_ops, const int32_t* codes,
                              const float* factors, void* ws, void* stream) {
  SFOD_REQUIRE_EXTENTS("aug_color", H, W, n_ops);
  SFOD_REQUIRE(n_ops >= 0 && n_ops <= AUG_MAX_OPS, "aug_color: at most 8 ops");
  SFOD_REQUIRE(n_ops == 0 || (codes != nullptr && factors != nullptr), "aug_color: codes / factors are host arrays of n_ops entries");
  SFOD_REQUIRE(sfod_prod_fits({H, W}, 1LL << 40), "aug_color: oversized image");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)H * W;
  if (n == 0) return 0;
  AugOps o{};
  o.n = n_ops;
  int contrast_at = -1;
  for (int k = 0; k < n_ops; ++k) {
    SFOD_REQUIRE(codes[k] >= OP_BRIGHTNESS && codes[k] <= OP_GRAYSCALE, "aug_color: unknown op code");
    o.code[k] = codes[k];
    o.factor[k] = factors[k];
    o.hshift[k] = (int)factors[k] & 255;      // hue: the factor slot carries np.uint8(hue_factor * 255)
    if (codes[k] == OP_CONTRAST) {
      SFOD_REQUIRE(contrast_at < 0, "aug_color: one contrast op per call");
      contrast_at = k;
    }
  }
  unsigned long long* sum = nullptr;
  if (contrast_at >= 0) {
    SFOD_REQUIRE(ws != nullptr, "aug_color: contrast needs the 8-byte workspace");
    sum = (unsigned long long*)ws;
    (void)hipMemsetAsync(sum, 0, sizeof(unsigned long long), s);
    const int blocks = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    hipLaunchKernelGGL(k_aug_luma_sum, dim3(blocks), dim3(256), 0, s, in, n, o, contrast_at, sum);
    int rc = sfod_check_launch("aug_luma_sum");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_aug_color, dim3(cdiv(n, 256)), dim3(256), 0, s, in, out, n, o, sum);
  return sfod_check_launch("aug_color");
}

extern "C" int sfod_aug_gaussian_blur(const uint8_t* in, uint8_t* out, uint8_t* tmp, int C, int H, int W,
                                      float sigma, void* stream) {
  SFOD_REQUIRE_EXTENTS("aug_gaussian_blur", C, H, W);
  SFOD_REQUIRE(sfod_prod_fits({C, H, W}, 1LL << 40), "aug_gaussian_blur: oversized image");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)C * H * W;
  if (n == 0) return 0;
  SFOD_REQUIRE(in != out && in != tmp && out != tmp, "aug_gaussian_blur: in / out / tmp must be distinct");
  SFOD_REQUIRE(sigma > 0.f && sigma <= 1024.f, "aug_gaussian_blur: sigma outside (0, 1024]");      // (NaN fails both)
  const float fr = gaussian_box_radius(sigma, 3);
  const int radius = (int)fr;
  const unsigned ww = (unsigned)((1 << 24) / (fr * 2 + 1));
  const unsigned fw = ((1u << 24) - (unsigned)(radius * 2 + 1) * ww) / 2;
  // 3 passes along x, 3 along y (Pillow transposes in between; the arithmetic per line is the same)
  const uint8_t* src = in;
  uint8_t* bufs[2] = {tmp, out};
  for (int pass = 0; pass < 6; ++pass) {
    uint8_t* dst = bufs[pass & 1];             // passes 0..5 -> tmp, out, tmp, out, tmp, out
    hipLaunchKernelGGL(k_aug_box_pass, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, C, H, W, pass >= 3 ? 1 : 0,
                       radius, ww, fw);
    int rc = sfod_check_launch("aug_box_pass");
    if (rc) return rc;
    src = dst;
  }
  return 0;
}

extern "C" int sfod_aug_erase(uint8_t* img, int C, int H, int W, int i, int j, int h, int w, const float* noise,
                              void* stream) {
  SFOD_REQUIRE_EXTENTS("aug_erase", C, H, W, i, j, h, w);
  SFOD_REQUIRE(i >= 0 && j >= 0 && h >= 0 && w >= 0 && (int64_t)i + h <= H && (int64_t)j + w <= W,
               "aug_erase: rectangle outside the image");
  SFOD_REQUIRE(sfod_prod_fits({C, H, W}, 1LL << 40), "aug_erase: oversized image");
  const int64_t n = (int64_t)C * h * w;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_aug_erase, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, img, C, H, W, i, j, h, w,
                     noise);
  return sfod_check_launch("aug_erase");
}
