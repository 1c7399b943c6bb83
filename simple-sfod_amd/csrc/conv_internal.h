// Internal (non-ABI) interface between gemm_conv.hip (the sfod_conv_* entry points) and the
// halo-patch 3x3 kernels in conv3x3_patch.hip / wgrad3x3_rows.hip.
#pragma once
#include "common.h"

// ---- forward / dgrad 3x3, bf16 -----------------------------------------------------------------
// Tile plan of the halo-patch kernel for one layer shape; ok == 0 -> shape not supported, use the
// generic implicit-GEMM kernel.
struct P3Plan {
  int ok;
  int G;                 // 1: 128 output channels / workgroup, 2: 64
  int FM;                // 32-row fragments per wave: 4 (G=1), 2 or 1 (G=2: 512- or 256-pixel tiles)
  int TH, TW;            // output pixels per tile (TH*TW <= 512 | 256, (TH+2)*(TW+2) <= 640 | 384)
  int tiles_y, tiles_x, tiles_n;
  int nblk;              // BatchNorm statistics blocks = B * tiles_y * tiles_x
  int m16;               // G = 1, FM = 2: operand-pair launches with Cin % 64 == 0 run k_conv3x3_m16 (1: 8 waves, 2: 4 waves; same tiles)
};
// pairs: the operands are (hi, lo) pairs (SFOD_BF16X3 / SFOD_F16X3; Cin in physical 16-bit channels)
P3Plan sfod_p3_plan(int B, int H, int W, int Cin, int Cout, int pairs = 0);
// BatchNorm-backward reduction folded into a data-gradient launch (fp32 output [B,H,W,Cout] dense): see P3Args
struct P3BnRed {
  const float* y;        // saved pre-BatchNorm output of the layer whose output gradient is being produced
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float* ws;             // [plan.nblk][2][Cout] partial (dbeta, dgamma) sums
};
// BatchNorm + ReLU of the PRODUCER folded into this launch's operand path (k_conv3x3_m16<4, 8, 1, false, true>): x is the
// producer's pre-BatchNorm fp32 output, the kernel converts each patch slice to operand pairs in LDS.  Forward-only passes.
struct P3BnIn {
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
};
bool sfod_p3_bnin_ok(const P3Plan& p, int Cin_phys, int split);
int sfod_p3_launch(const P3Plan& p, const void* x, const void* w, const float* bias, void* y, float* stats,
                   int B, int H, int W, int Cin, int Cout, int ldy, int act, int out_f32, hipStream_t s, int split = 0,
                   const P3BnRed* red = nullptr, const unsigned* wamax = nullptr, const P3BnIn* bnin = nullptr);

// BatchNorm statistics buffer layout shared by every conv kernel:
//   stats[blk][0][c] = sum over the block's valid rows, stats[blk][1][c] = sum of squared deviations
//   from the block mean, and, behind the nblk*2*C sums, counts[blk] = number of valid rows (float).

// ---- weight gradient 3x3, bf16 -------------------------------------------------------------------
struct W3Plan {
  int ok;
  int CO;                // 32-channel output fragments per workgroup: 4 (128 channels) or 2 (Cout <= 64)
  int TH, TW, tiles_y, tiles_x;
  int co_tiles, ci_tiles;
  int tiles_per_split, nsplit, nslab;
  int64_t ws_bytes;      // fp32 partial slabs [nslab][Cout][9][Cin]
  int w64;               // bf16x3 only: the 64 co x 64 ci block / 128-pixel tile kernel (k_wgrad3x3_w64)
};
W3Plan sfod_w3_plan(int B, int H, int W, int Cin, int Cout, int lddy, int split = 0);
// out_mode 0: dw packed [Cout][9][Cin], accumulated into; 1 / 2: dw OIHW, overwritten / accumulated into
int sfod_w3_launch(const W3Plan& p, const void* x, const void* dy, float* dw, void* ws, int B, int H, int W,
                   int Cin, int Cout, int lddy, int out_mode, hipStream_t s, int split = 0);

// ---- deterministic mode (gemm_conv.hip: sfod_set_deterministic / SFOD_DETERMINISTIC) -----------------------------------
bool sfod_deterministic();

// ---- SFOD_F16X3 saturation words of the translation units that produce half pairs (common.h) -----------------------
void sfod_f16_poll_elementwise(unsigned* out, hipStream_t s);
void sfod_f16_poll_roi(unsigned* out, hipStream_t s);
void sfod_f16_poll_first(unsigned* out, hipStream_t s);
void sfod_f16_poll_stem(unsigned* out, hipStream_t s);

// ---- first layer (Cin = one padded chunk of 8, Cout = 64), bf16 / bf16x3 ------------------------------------------
int sfod_f1_nblk(int B, int H, int W);
// y == nullptr: statistics only (no stores); scale / shift: optional per-channel affine before the activation
int sfod_f1_launch(const void* x, const void* w, const float* bias, void* y, float* stats, int B, int H, int W,
                   int ldy, int act, hipStream_t s, const float* scale = nullptr, const float* shift = nullptr, int split = 0,
                   const unsigned* wamax = nullptr);
