// First VGG layer (conv1_1: 3 -> 64 channels, 3x3, pad 1) on bf16 MFMA, gfx950.
//
// The input has 3 real channels padded to one 16-byte chunk (8 bf16), so an im2col row is 9 taps x 8
// channels = 72 values and the layer is bound by writing its 64-channel output, not by arithmetic.  The
// generic implicit GEMM spends 0.48 ms on it (per-lane tap decoding, 128-row tiles, half-empty K tiles).
// Here: workgroup = 8 x 32 output pixels, 4 waves (two image rows each); the 10 x 34 halo patch is staged
// in LDS as one 16-byte chunk per pixel, so an A fragment of k-step s (taps 2s, 2s+1) is ONE conflict-free
// ds_read_b128 per lane; all ten weight fragments live in registers for the whole (grid-stride) tile loop.
// Epilogue as in conv3x3_patch.hip: bias, per-wave LDS staging for 16-byte stores, BatchNorm (sum, M2,
// count) per workgroup.
#include "conv_internal.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int TH = 8, TW = 32, PW = TW + 2, PH = TH + 2;
constexpr int PATCH_BYTES = PH * PW * 16;            // 5440
constexpr int STG_OFF = 5632;                        // staged C tiles: 4 waves x 64 rows x 144 B
constexpr int STG_PITCH = 144;
constexpr int SRED_OFF = STG_OFF + 4 * 64 * STG_PITCH;   // float [4][64][2] + float [4]
constexpr int LDS_BYTES = SRED_OFF + 4 * 64 * 2 * 4 + 16;
// SFOD_BF16X3 variant: 32-byte pixels ((8 hi | 8 lo) pairs), fp32 staging of 32 rows x 64 columns per wave (272-byte pitch)
constexpr int X3_PATCH_BYTES = PH * PW * 32;         // 10880
constexpr int X3_W_OFF = 11008;                      // weight fragments [k-step][j][hi/lo][lane] x 16 B (held in LDS, not
constexpr int X3_W_BYTES = 5 * 2 * 2 * 64 * 16;      // in 80 VGPRs: 3 workgroups per CU instead of 1)
constexpr int X3_STG_OFF = X3_W_OFF + X3_W_BYTES;
constexpr int X3_FP = 272;
constexpr int X3_SRED_OFF = X3_STG_OFF + 4 * 16 * X3_FP;   // 16 rows x 64 columns per wave at a time
constexpr int X3_LDS_BYTES = X3_SRED_OFF + 4 * 64 * 2 * 4 + 16;

struct F1Args {
  const bf16_t* x;     // [B,H,W,8]   (bf16x3: 16 bf16 per pixel = 8 hi | 8 lo)
  const bf16_t* w;     // [64][9][8]  (bf16x3: 16 bf16 per (output channel, tap))
  const float* bias;
  const float* scale;  // optional per-channel affine applied to (conv + bias) before the activation: the BatchNorm
  const float* shift;  //   of a second pass whose statistics came from a first, store-free pass (teacher forward)
  void* y;             // [B,H,W,ldy] bf16 (bf16x3: fp32 without an affine, (hi, lo) pairs with one); nullptr: statistics only
  float* stats;
  int B, H, W, ldy, act;
  int tiles_y, tiles_x, ntiles;
  const unsigned* wamax;   // SFOD_F16X3: bits of max|w| of the scaled packed weights (common.h), or nullptr
};

__global__ void __launch_bounds__(256)
k_conv_first(F1Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l31 = lane & 31;

  // weight fragments: bq[j][s] = w[j*32 + l31][tap 2s+h][0..7], zeros for the padding tap 9
  bf16x8 bq[2][5];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + h;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (tap < 9) v = *reinterpret_cast<const uint4*>(a.w + ((j * 32 + l31) * 9 + tap) * 8);
      bq[j][s] = __builtin_bit_cast(bf16x8, v);
    }
  float bcol[2], scol[2], hcol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bcol[j] = a.bias ? a.bias[j * 32 + l31] : 0.f;
    scol[j] = a.scale ? a.scale[j * 32 + l31] : 1.f;
    hcol[j] = a.shift ? a.shift[j * 32 + l31] : 0.f;
  }
  const float lo = (a.act == 1) ? 0.f : -__builtin_inff();

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int t = tile;
    const int txi = t % a.tiles_x;
    t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int x0 = txi * TW, y0 = tyi * TH;
    const bf16_t* ximg = a.x + (int64_t)b * a.H * a.W * 8;
    __syncthreads();   // previous tile's LDS users are done
    for (int p = threadIdx.x; p < PH * PW; p += 256) {
      const int py = p / PW, px = p - py * PW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = *reinterpret_cast<const uint4*>(ximg + ((int64_t)iy * a.W + ix) * 8);
      *reinterpret_cast<uint4*>(smem + p * 16) = v;
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + h;               // per-lane tap (two taps per k-step)
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (2 * wave + i + ky) * PW + l31 + kx;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (tap < 9) v = *reinterpret_cast<const uint4*>(smem + row * 16);
        const bf16x8 af = __builtin_bit_cast(bf16x8, v);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[j][s], acc[i][j], 0, 0, 0);
      }
    }
    // ---- epilogue: rows of M-fragment i are the 32 pixels of image row y0 + 2*wave + i
    unsigned char* stg = smem + STG_OFF + wave * (64 * STG_PITCH);
    const bool full = (y0 + TH <= a.H) && (x0 + TW <= a.W);   // interior tile: no row masks
    unsigned vmask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned m = 0xffffu;
      if (!full) {
        m = 0;
        const int iy = y0 + 2 * wave + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * h;
          if (iy < a.H && x0 + px < a.W) m |= (1u << r);
        }
      }
      vmask[i] = m;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += bcol[j];
    if (a.y != nullptr) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            // scol / hcol are (1, 0) without an affine and lo is -inf without a ReLU: two VALU ops per element,
            // no per-element select on the (uniform) mode
            const float o = fmaxf(__builtin_fmaf(acc[i][j][r], scol[j], hcol[j]), lo);
            *reinterpret_cast<bf16_t*>(stg + ml * STG_PITCH + (j * 32 + l31) * 2) = (bf16_t)o;
          }
      // same-wave readback (LDS is in order per wave): 8 rows x 128 B per instruction
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const int iy = y0 + 2 * wave + (row >> 5), ix = x0 + (row & 31);
        if (iy < a.H && ix < a.W) {
          const uint4 v = *reinterpret_cast<const uint4*>(stg + row * STG_PITCH + ch * 16);
          *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.y) + (((int64_t)b * a.H + iy) * a.W + ix) * a.ldy + ch * 8) = v;
        }
      }
    }
    if (a.stats != nullptr) {
      float* sred = reinterpret_cast<float*>(smem + SRED_OFF);   // [wave][64][2]
      float* scnt = sred + 4 * 64 * 2;
      int cnt = __builtin_popcount(vmask[0]) + __builtin_popcount(vmask[1]);
      cnt += __shfl_xor(cnt, 32);
      const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float sm = 0.f, q = 0.f, mean;
        if (full) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += acc[i][j][r];
          sm += __shfl_xor(sm, 32);
          mean = sm * inv;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = acc[i][j][r] - mean;
              q = __builtin_fmaf(d, d, q);
            }
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += ((vmask[i] >> r) & 1u) ? acc[i][j][r] : 0.f;
          sm += __shfl_xor(sm, 32);
          mean = sm * inv;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = acc[i][j][r] - mean;
              q += ((vmask[i] >> r) & 1u) ? d * d : 0.f;
            }
        }
        q += __shfl_xor(q, 32);
        if (h == 0) {
          sred[(wave * 64 + j * 32 + l31) * 2 + 0] = sm;
          sred[(wave * 64 + j * 32 + l31) * 2 + 1] = q;
        }
      }
      if (lane == 0) scnt[wave] = (float)cnt;
      __syncthreads();
      if (threadIdx.x < 64) {
        const int col = threadIdx.x;
        double n_tot = 0.0, s_tot = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { n_tot += (double)scnt[k]; s_tot += (double)sred[(k * 64 + col) * 2]; }
        const double mu = n_tot > 0.0 ? s_tot / n_tot : 0.0;
        double m2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double nk = (double)scnt[k];
          if (nk > 0.0) {
            const double d = (double)sred[(k * 64 + col) * 2] / nk - mu;
            m2 += (double)sred[(k * 64 + col) * 2 + 1] + nk * d * d;
          }
        }
        a.stats[((int64_t)tile * 2 + 0) * 64 + col] = (float)s_tot;
        a.stats[((int64_t)tile * 2 + 1) * 64 + col] = (float)m2;
        if (col == 0) a.stats[(int64_t)a.ntiles * 2 * 64 + tile] = (float)n_tot;
      }
    }
  }
}

// ---- SFOD_BF16X3 variant ----------------------------------------------------------------------------------------------
// Same tiling; a k-step is (taps 2s, 2s+1) x 8 logical channels fed as hi*lo + lo*hi + hi*hi.  Output: fp32 y (student
// pass: the BatchNorm backward needs it), or -- with the affine of a second, recomputing pass (teacher) -- the next
// layer's operand pairs z = relu(scale * (conv + bias) + shift) directly; per wave 32 rows at a time are staged in LDS
// and stored as whole 256-byte pixels.
template <int FMT>      // 1: bf16 pairs (SFOD_BF16X3), 2: f16 pairs (SFOD_F16X3): MFMA opcode and the z pairs it writes
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_conv_first_x3(F1Args a) {
  using pair_t = typename std::conditional<FMT == 2, splith_t, split_t>::type;
  // SFOD_F16X3 weights carry a per-tensor power-of-two scale (common.h): undone on the accumulators (exact)
  const float winv = (FMT == 2 && a.wamax != nullptr) ? winv_from_absmax(*a.wamax) : 1.0f;
  __shared__ __attribute__((aligned(16))) unsigned char smem[X3_LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  // weight fragments -> LDS, once per workgroup (the grid is persistent)
  for (int e = threadIdx.x; e < 5 * 2 * 64; e += 256) {
    const int ln = e & 63, j = (e >> 6) & 1, ks = e >> 7;
    const int tap = 2 * ks + (ln >> 5);
    uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
    if (tap < 9) {
      const uint4* p = reinterpret_cast<const uint4*>(a.w + ((j * 32 + (ln & 31)) * 9 + tap) * 16);
      vh = p[0];
      vl = p[1];
    }
    *reinterpret_cast<uint4*>(smem + X3_W_OFF + (((ks * 2 + j) * 2 + 0) * 64 + ln) * 16) = vh;
    *reinterpret_cast<uint4*>(smem + X3_W_OFF + (((ks * 2 + j) * 2 + 1) * 64 + ln) * 16) = vl;
  }
  float bcol[2], scol[2], hcol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bcol[j] = a.bias ? a.bias[j * 32 + l31] : 0.f;
    scol[j] = a.scale ? a.scale[j * 32 + l31] : 1.f;
    hcol[j] = a.shift ? a.shift[j * 32 + l31] : 0.f;
  }
  const float lo = (a.act == 1) ? 0.f : -__builtin_inff();
  const bool zsplit = (a.scale != nullptr);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int t = tile;
    const int txi = t % a.tiles_x;
    t /= a.tiles_x;
    const int tyi = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int x0 = txi * TW, y0 = tyi * TH;
    const bf16_t* ximg = a.x + (int64_t)b * a.H * a.W * 16;
    __syncthreads();   // previous tile's LDS users are done
    for (int p = threadIdx.x; p < PH * PW * 2; p += 256) {      // one 16-byte chunk (hi or lo) per thread and round
      const int pp = p >> 1, half = p & 1;
      const int py = pp / PW, px = pp - py * PW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
        v = *reinterpret_cast<const uint4*>(ximg + ((int64_t)iy * a.W + ix) * 16 + half * 8);
      *reinterpret_cast<uint4*>(smem + p * 16) = v;
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const int tap = 2 * s + h;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (2 * wave + i + ky) * PW + l31 + kx;
        uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
        if (tap < 9) {
          vh = *reinterpret_cast<const uint4*>(smem + row * 32);
          vl = *reinterpret_cast<const uint4*>(smem + row * 32 + 16);
        }
        const bf16x8 ah = __builtin_bit_cast(bf16x8, vh), al = __builtin_bit_cast(bf16x8, vl);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 bh = __builtin_bit_cast(
              bf16x8, *reinterpret_cast<const uint4*>(smem + X3_W_OFF + (((s * 2 + j) * 2 + 0) * 64 + lane) * 16));
          const bf16x8 bl = __builtin_bit_cast(
              bf16x8, *reinterpret_cast<const uint4*>(smem + X3_W_OFF + (((s * 2 + j) * 2 + 1) * 64 + lane) * 16));
          f32x16 c = acc[i][j];
          c = mfma_pairs<FMT>(ah, bl, c);
          c = mfma_pairs<FMT>(al, bh, c);
          c = mfma_pairs<FMT>(ah, bh, c);
          acc[i][j] = c;
        }
      }
    }
    const bool full = (y0 + TH <= a.H) && (x0 + TW <= a.W);
    unsigned vmask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned m = 0xffffu;
      if (!full) {
        m = 0;
        const int iy = y0 + 2 * wave + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * h;
          if (iy < a.H && x0 + px < a.W) m |= (1u << r);
        }
      }
      vmask[i] = m;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * winv + bcol[j];
    if (a.y != nullptr) {
      unsigned char* stg = smem + X3_STG_OFF + wave * (16 * X3_FP);
#pragma unroll
      for (int i = 0; i < 2; ++i) {          // M-fragment i = the 32 pixels of image row y0 + 2 * wave + i
        const int iy = y0 + 2 * wave + i;
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {     // pixels 16 * qh .. 16 * qh + 15 (accumulator registers 8 * qh .. 8 * qh + 7)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const int r = qh * 8 + rr;
              const int ml = (rr & 3) + 8 * (rr >> 2) + 4 * h;
              const float o = fmaxf(__builtin_fmaf(acc[i][j][r], scol[j], hcol[j]), lo);
              *reinterpret_cast<float*>(stg + ml * X3_FP + (j * 32 + l31) * 4) = o;
            }
          if (zsplit) {
            // 8 lanes per pixel: 8 consecutive channels -> one 32-byte (hi | lo) group
#pragma unroll
            for (int it = 0; it < 2; ++it) {
              const int row = it * 8 + (lane >> 3), gch = lane & 7;
              const int ix = x0 + qh * 16 + row;
              if (iy < a.H && ix < a.W) {
                const float4 v0 = *reinterpret_cast<const float4*>(stg + row * X3_FP + gch * 32);
                const float4 v1 = *reinterpret_cast<const float4*>(stg + row * X3_FP + gch * 32 + 16);
                const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                split_store8(reinterpret_cast<pair_t*>(a.y) + (((int64_t)b * a.H + iy) * a.W + ix) * a.ldy + gch * 8, f);
              }
            }
          } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int row = it * 4 + (lane >> 4), ch = lane & 15;
              const int ix = x0 + qh * 16 + row;
              if (iy < a.H && ix < a.W)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (((int64_t)b * a.H + iy) * a.W + ix) * a.ldy + ch * 4) =
                    *reinterpret_cast<const float4*>(stg + row * X3_FP + ch * 16);
            }
          }
        }
      }
    }
    if (a.stats != nullptr) {
      float* sred = reinterpret_cast<float*>(smem + X3_SRED_OFF);   // [wave][64][2]
      float* scnt = sred + 4 * 64 * 2;
      int cnt = __builtin_popcount(vmask[0]) + __builtin_popcount(vmask[1]);
      cnt += __shfl_xor(cnt, 32);
      const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float sm = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) sm += ((vmask[i] >> r) & 1u) ? acc[i][j][r] : 0.f;
        sm += __shfl_xor(sm, 32);
        const float mean = sm * inv;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[i][j][r] - mean;
            q += ((vmask[i] >> r) & 1u) ? d * d : 0.f;
          }
        q += __shfl_xor(q, 32);
        if (h == 0) {
          sred[(wave * 64 + j * 32 + l31) * 2 + 0] = sm;
          sred[(wave * 64 + j * 32 + l31) * 2 + 1] = q;
        }
      }
      if (lane == 0) scnt[wave] = (float)cnt;
      __syncthreads();
      if (threadIdx.x < 64) {
        const int col = threadIdx.x;
        double n_tot = 0.0, s_tot = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { n_tot += (double)scnt[k]; s_tot += (double)sred[(k * 64 + col) * 2]; }
        const double mu = n_tot > 0.0 ? s_tot / n_tot : 0.0;
        double m2 = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double nk = (double)scnt[k];
          if (nk > 0.0) {
            const double d = (double)sred[(k * 64 + col) * 2] / nk - mu;
            m2 += (double)sred[(k * 64 + col) * 2 + 1] + nk * d * d;
          }
        }
        a.stats[((int64_t)tile * 2 + 0) * 64 + col] = (float)s_tot;
        a.stats[((int64_t)tile * 2 + 1) * 64 + col] = (float)m2;
        if (col == 0) a.stats[(int64_t)a.ntiles * 2 * 64 + tile] = (float)n_tot;
      }
    }
  }
}

}  // namespace

int sfod_f1_nblk(int B, int H, int W) { return B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW); }

// split 1 / 2: SFOD_BF16X3 / SFOD_F16X3 operands; y is fp32 (scale == nullptr) or operand pairs (scale / shift given)
int sfod_f1_launch(const void* x, const void* w, const float* bias, void* y, float* stats, int B, int H, int W,
                   int ldy, int act, hipStream_t s, const float* scale, const float* shift, int split, const unsigned* wamax) {
  F1Args a;
  a.wamax = wamax;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = y; a.stats = stats;
  a.scale = scale; a.shift = shift;
  a.B = B; a.H = H; a.W = W; a.ldy = ldy; a.act = act;
  a.tiles_y = (H + TH - 1) / TH; a.tiles_x = (W + TW - 1) / TW;
  a.ntiles = B * a.tiles_y * a.tiles_x;
  const int resident = split ? 256 * 3 : 256 * 8;      // bf16x3: 3 workgroups per CU (LDS), each stages the weights once
  int grid = a.ntiles < resident ? a.ntiles : resident;
  if (split == 2) hipLaunchKernelGGL(k_conv_first_x3<2>, dim3(grid), dim3(256), 0, s, a);
  else if (split) hipLaunchKernelGGL(k_conv_first_x3<1>, dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_conv_first, dim3(grid), dim3(256), 0, s, a);
  return sfod_check_launch("conv_first");
}

SFOD_DEFINE_F16_POLL(sfod_f16_poll_first)
