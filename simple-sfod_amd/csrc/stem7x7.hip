// ResNet BasicStem (detectron2 build_resnet_backbone: conv 7x7 stride 2 pad 3, 3 -> 64, FrozenBN folded into the weights,
// ReLU) as ONE kernel on operand pairs (SFOD_BF16X3 / SFOD_F16X3), gfx950.
//
// Before (round 4): sfod_im2col_stem wrote the [B*Ho*Wo][160] operand matrix to HBM (921 MB for eight 600x1200 frames,
// 0.57 ms), the generic GEMM read it back (N = 64: one column tile, 0.75 ms), per pass -- 4 % of the config #5 step.  The
// im2col row of a pixel is 147 values of a 7x7x3 window: here a workgroup keeps the input patch of its output tile in
// LDS (4 x 32 output pixels <- 13 x 69 input pixels x 3 channels, 10.5 KiB), every lane GATHERS the eight k-values its MFMA
// operand needs for its pixel straight from that patch, splits them into (hi, lo) in registers and feeds
// v_mfma_f32_32x32x16 -- the operand matrix never exists.  The packed weights (64 x 160 pairs, 40 KiB, sfod_pack_fc_weight's
// layout) sit in LDS for the workgroup's whole walk over its tiles (persistent grid).
//
// Same numbers as the path it replaces, bit for bit: the k order (k = (ky * 7 + kx) * 3 + c, zero columns up to 160), the
// split of every value (split_store8's), the MFMA sequence per accumulator (k-steps of 16 in order; hi*lo, lo*hi, hi*hi)
// and the epilogue (weight scale, bias, ReLU) are the generic kernel's (gemm_conv.hip::k_conv_fwd, SPLIT).
#include "common.h"
#include "conv_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int ST_TH = 4, ST_TW = 32;                 // output tile: 4 rows x 32 columns = one 32-pixel MFMA tile per wave
constexpr int ST_PR = 2 * ST_TH + 5, ST_PC = 2 * ST_TW + 5;      // input patch: 13 x 69 pixels
constexpr int ST_K = 147, ST_KPAD = 160, ST_N = 64;
constexpr int ST_WROW = ST_KPAD * 4 + 16;            // bytes per weight row in LDS (+16: conflict-free b128 reads over 16 rows)
constexpr int ST_PATCH = ST_PR * ST_PC * 3;          // words; slot ST_PATCH holds 0 (padding columns of K)
constexpr int ST_LDS = ST_N * ST_WROW + (ST_PATCH + 4) * 4 + ST_KPAD * 4;

template <int FMT> struct PairOf;
template <> struct PairOf<1> {
  static __device__ __forceinline__ void split(float v, unsigned short& h, unsigned short& l) {
    const bf16_t hh = (bf16_t)v;
    const bf16_t ll = (bf16_t)(v - (float)hh);
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, ll);
  }
};
template <> struct PairOf<2> {
  static __device__ __forceinline__ void split(float v, unsigned short& h, unsigned short& l) {
    f16_t hh, ll;
    f16_pair(v, hh, ll);                               // saturating, reported through g_f16_sat (common.h)
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, ll);
  }
};

template <int FMT>
__global__ void __launch_bounds__(256)
k_stem7x7(const float* __restrict__ x, const unsigned char* __restrict__ wpk, const float* __restrict__ bias,
          const unsigned* __restrict__ wamax, float* __restrict__ y, int B, int H, int W, int Cp, int Ho, int Wo,
          int tiles_x, int tiles_y, int act) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sW = smem;                                            // [64][ST_WROW]
  // patch + the zero slot.  Every input value is split into its (hi, lo) pair ONCE, when the patch is written, and stored as
  // hi | lo << 16: a window element is used by ~12 output pixels, the gathers below only pick the halves apart.
  unsigned* sP = reinterpret_cast<unsigned*>(smem + ST_N * ST_WROW);
  int* sLut = reinterpret_cast<int*>(smem + ST_N * ST_WROW + (ST_PATCH + 4) * 4);   // k -> patch offset of (ky, kx, c)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, px = lane & 31;

  // ---- once per workgroup: the packed weights and the k -> (ky, kx, c) table ---------------------------------------------
  for (int c = tid; c < ST_N * (ST_KPAD * 4 / 16); c += 256) {        // 16-byte chunks: 40 per row
    const int n = c / (ST_KPAD * 4 / 16), q = c - n * (ST_KPAD * 4 / 16);
    *reinterpret_cast<uint4*>(sW + n * ST_WROW + q * 16) =
        *reinterpret_cast<const uint4*>(wpk + (int64_t)n * (ST_KPAD * 4) + q * 16);
  }
  for (int k = tid; k < ST_KPAD; k += 256) {
    int off = ST_PATCH;                                                // the zero slot
    if (k < ST_K) {
      const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
      off = (ky * ST_PC + kx) * 3 + c;
    }
    sLut[k] = off;
  }
  if (tid < 4) sP[ST_PATCH + tid] = 0u;                               // split(0.0f) = (0, 0)
  const float inv = (FMT == 2 && wamax != nullptr) ? winv_from_absmax(*wamax) : 1.f;
  float bcol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bcol[j] = bias != nullptr ? bias[j * 32 + px] : 0.f;

  const int ntiles = B * tiles_y * tiles_x;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_y * tiles_x);
    const int trem = tile - b * (tiles_y * tiles_x);
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const float* xb = x + (int64_t)b * H * W * Cp;
    __syncthreads();                                                   // the previous tile's gathers are done (and sW / sLut are written)
    for (int p = tid; p < ST_PR * ST_PC; p += 256) {
      const int r = p / ST_PC, c = p - r * ST_PC;
      const int iy = iy0 + r, ix = ix0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const float4*>(xb + ((int64_t)iy * W + ix) * Cp);
      const float vv[3] = {v.x, v.y, v.z};
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) {
        unsigned short hh, ll;
        PairOf<FMT>::split(vv[c3], hh, ll);
        sP[p * 3 + c3] = (unsigned)hh | ((unsigned)ll << 16);
      }
    }
    __syncthreads();

    // this wave: output row oy0 + wave, columns ox0 .. ox0 + 31; lane (px, h) feeds pixel px with k = 16 s + 8 h .. + 7
    const int pbase = ((2 * wave) * ST_PC + 2 * px) * 3;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 2
    for (int s = 0; s < ST_KPAD / 16; ++s) {
      union { unsigned u[4]; bf16x8 v; } ah, al;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int o0 = sLut[16 * s + 8 * h + e], o1 = sLut[16 * s + 8 * h + e + 1];
        const unsigned w0 = sP[o0 == ST_PATCH ? ST_PATCH : pbase + o0];
        const unsigned w1 = sP[o1 == ST_PATCH ? ST_PATCH : pbase + o1];
        ah.u[e >> 1] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);     // (lo16 of w0) | (lo16 of w1) << 16: the two hi parts
        al.u[e >> 1] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);     // the two lo parts
      }
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned char* wr = sW + (j * 32 + px) * ST_WROW + (2 * s + h) * 32;
        bh[j] = *reinterpret_cast<const bf16x8*>(wr);
        bl[j] = *reinterpret_cast<const bf16x8*>(wr + 16);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = mfma_pairs<FMT>(ah.v, bl[j], acc[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = mfma_pairs<FMT>(al.v, bh[j], acc[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = mfma_pairs<FMT>(ah.v, bh[j], acc[j]);
    }
    // ---- epilogue: weight scale (f16 pairs), bias, activation; 32 lanes x 4 bytes = 128-byte segments per row ---------------
    const int oy = oy0 + wave;
    if (oy < Ho) {
      float* yrow = y + (((int64_t)b * Ho + oy) * Wo) * ST_N;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = (r & 3) + 8 * (r >> 2) + 4 * h;               // pixel inside the 32-pixel tile (32x32 accumulator layout)
          const int ox = ox0 + ml;
          float v = acc[j][r];
          if (FMT == 2) v *= inv;
          v += bcol[j];
          v = (act == 1) ? fmaxf(v, 0.f) : v;
          if (ox < Wo) yrow[(int64_t)ox * ST_N + j * 32 + px] = v;
        }
    }
  }
}

}  // namespace

SFOD_DEFINE_F16_POLL(sfod_f16_poll_stem)

extern "C" int sfod_stem7x7_supported(int B, int H, int W, int Cp, int dt) {
  if (!sfod_ints_ok({B, H, W, Cp}) || !sfod_prod_fits({B, H, W}) || !sfod_prod_fits({B, H, W, 64}, 1LL << 40)) return 0;
  return (sfod_is_pairs(dt) && Cp >= 4 && Cp % 4 == 0 && B >= 1 && H >= 1 && W >= 1) ? 1 : 0;
}

extern "C" int sfod_stem7x7(const float* x, const void* w_packed, const uint32_t* w_absmax, const float* bias, float* y,
                            int B, int H, int W, int Cp, int act, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("stem7x7", B, H, W, Cp);
  SFOD_REQUIRE(sfod_stem7x7_supported(B, H, W, Cp, dt), "stem7x7: operand pairs (SFOD_BF16X3 / SFOD_F16X3), Cp a multiple of 4, non-empty input");
  SFOD_REQUIRE(x != nullptr && w_packed != nullptr && y != nullptr, "stem7x7: null argument (x, w_packed, y)");
  SFOD_REQUIRE(act == 0 || act == 1, "stem7x7: act is 0 (none) or 1 (ReLU)");
  SFOD_REQUIRE(w_absmax == nullptr || dt == SFOD_F16X3, "stem7x7: scaled weights are an SFOD_F16X3 format");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int tiles_x = (Wo + ST_TW - 1) / ST_TW, tiles_y = (Ho + ST_TH - 1) / ST_TH;
  SFOD_REQUIRE(sfod_prod_fits({B, tiles_x, tiles_y}), "stem7x7: oversized tile grid");
  const int ntiles = B * tiles_x * tiles_y;
  hipStream_t s = (hipStream_t)stream;
  const int grid = ntiles < 256 * 3 ? ntiles : 256 * 3;               // persistent: three workgroups per CU share the LDS
  if (dt == SFOD_F16X3) {
    static const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem7x7<2>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
    if (rc != hipSuccess) { sfod_set_error("hipFuncSetAttribute(stem7x7): %s", hipGetErrorString(rc)); return -(int)rc; }
    hipLaunchKernelGGL(k_stem7x7<2>, dim3(grid), dim3(256), ST_LDS, s, x, (const unsigned char*)w_packed, bias, w_absmax, y, B, H,
                       W, Cp, Ho, Wo, tiles_x, tiles_y, act);
  } else {
    static const hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem7x7<1>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
    if (rc != hipSuccess) { sfod_set_error("hipFuncSetAttribute(stem7x7): %s", hipGetErrorString(rc)); return -(int)rc; }
    hipLaunchKernelGGL(k_stem7x7<1>, dim3(grid), dim3(256), ST_LDS, s, x, (const unsigned char*)w_packed, bias, nullptr, y, B, H,
                       W, Cp, Ho, Wo, tiles_x, tiles_y, act);
  }
  return sfod_check_launch("stem7x7");
}
