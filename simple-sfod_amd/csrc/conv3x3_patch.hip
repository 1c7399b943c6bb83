// Halo-patch 3x3 convolution (forward and data-gradient), bf16 MFMA, gfx950.
//
// Why not the generic implicit GEMM (gemm_conv.hip): that kernel re-fetches every input pixel once
// per filter tap (9x) from L2 into LDS and is bound by the L2 -> LDS path, not by the matrix cores.
// Here a workgroup owns a TH x TW patch of output pixels (<= 512) of one image and BN output
// channels; per 32-channel slice of the input it DMAs the (TH+2) x (TW+2) halo patch into LDS ONCE
// and feeds all nine taps from it (an A fragment row for tap (ky,kx) is just the patch row shifted
// by ky*(TW+2)+kx).  Only the weights stream per tap.  L2 -> LDS bytes per MAC drop ~3.3x.
//
//   LDS   patch[2]  : 640 rows x 64 B  (32 bf16 channels / pixel), 16-byte chunk c of row r stored
//                     at chunk c ^ ((r >> 2) & 3): any 16 rows distinct mod 16 are conflict free
//                     for ds_read_b128
//         B ring[4] : 8 KiB = G (tap, 32ch) slices x BN weight rows x 64 B, same swizzle
//   waves 8 (512 threads).  G=1: BN=128, 4(M) x 2(N) waves, wave tile 128 px x 64 ch
//                           G=2: BN=64,  8(M) x 1(N) waves, wave tile  64 px x 64 ch
//   K loop "body" = G input slices = 9 stages; stage = G (tap, slice) pairs = 16 MFMA / wave.
//         One s_barrier per stage.  DMA (global_load_lds, 16 B / lane) runs 3 stages ahead for the
//         weights and up to a whole body ahead for the next patch; the queue is never drained in the
//         loop: each stage waits with a counted s_waitcnt vmcnt(N) (tables below, derived by
//         simulating the per-wave issue order: every wave issues the same number of DMAs per stage).
//   epilogue: bias + activation; bf16 tiles are staged per wave through LDS for 16-byte stores;
//         optional BatchNorm partial statistics (sum, M2, count) per workgroup (fp64 combine).
#include "conv_internal.h"
#include <type_traits>
#include <stdlib.h>
#include <atomic>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// (k_conv3x3_m16 has its own set for tools/experiments/m16_knockout.sh: M16_KO_{MFMA,WDMA,PDMA,BAR,EPI}, M16_V_LATEW,
// M16_V_NOCONF: fragment rows forced conflict-free -- what the LDS bank conflicts of ragged tile widths cost;
// M16_V_EXTRAVALU=n: n more VALU instructions per stage -- the slope says what the loop's ~66 non-MFMA VALU instructions cost.)
// Knock-out switches for tools/experiments/conv_knockout.sh (time floors of the bf16x3 loop; results are WRONG by
// construction, the product build defines none): P3_KO_MFMA no MFMAs, P3_KO_READS no LDS fragment reads, P3_KO_PATCHDMA /
// P3_KO_WDMA no patch / weight LDS-DMA inside the K loop, P3_KO_EPI no epilogue (stores, statistics).
namespace {

template <int J> using IC = std::integral_constant<int, J>;

constexpr int BSLOT = 8192;
constexpr int STG_PITCH = 144;                   // staged C row: 64 bf16 + 16 B pad

// LDS layout of one kernel variant.  FM = 32-row fragments per wave along M:
//   G=1: FM 4 (512-pixel tile), G=2: FM 2 (512-pixel tile) or FM 1 (256-pixel tile, ~76 KiB of LDS so that
//   TWO workgroups share a CU: the 64-channel layers are bound by streaming their input / output, and a
//   second resident workgroup overlaps one tile's prologue / epilogue with the other's MFMA loop).
template <int G, int FM> struct Lay {
  static constexpr bool SMALL = (G == 1) ? (FM == 2) : (FM == 1);   // 256-pixel tile, two workgroups per CU
  static constexpr int NPW = SMALL ? 3 : 5;                     // patch DMA pieces (1 KiB) per wave and slice
  static constexpr int PATCH_ROWS = NPW * 8 * 16;               // 640 | 384 pixels incl. halo
  static constexpr int PATCH_BYTES = PATCH_ROWS * 64;
  static constexpr int D = SMALL ? 2 : 3;                       // stages the weight DMA runs ahead
  static constexpr int NBS = D + 1;                             // weight ring slots
  static constexpr int BRING_OFF = 2 * PATCH_BYTES;
  static constexpr int OPER = BRING_OFF + NBS * BSLOT;
  static constexpr int WROWS = FM * 32;
  static constexpr int STG = 8 * WROWS * STG_PITCH;             // staged C tiles overlay the operand buffers
  static constexpr int PIXTAB_OFF = (OPER > STG) ? OPER : STG;  // int32 [512]
  static constexpr int SRED_OFF = PIXTAB_OFF + 2048;            // float [8 waves][64 cols][2] + float [8]
  static constexpr int TOTAL = SRED_OFF + 8 * 64 * 2 * 4 + 32;
};
constexpr int PATCH_ROWS_BIG = 640, PATCH_ROWS_SMALL = 384;

struct P3Args {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  void* y;
  float* stats;
  int B, H, W, Cin, Cout, ldy, act;
  int TH, TW, PW;
  int tiles_x, tiles_y, tiles_n;
  unsigned m_pw, m_tw, m_tn, m_tx, m_ty;   // fdiv magics of PW, TW, tiles_n, tiles_x, tiles_y
  int nbody;    // Cin / (32 * G)
  int ntiles;
  int nblk;
  // data-gradient launches only (fp32 output): the BatchNorm-backward reduction of the layer whose output gradient this
  // kernel produces, folded into the epilogue.  The output tile dz and the saved pre-BatchNorm tensor red_y share one
  // shape; per workgroup (sum g, sum g * xhat) with g = dz * [bn(y) > 0] go to red_ws[mtile][2][Cout] -- what
  // k_bn_bwd_reduce would write, so k_bn_bwd_finalize / k_bn_bwd_apply follow unchanged.  nullptr: off.
  const float* red_y;
  const float* red_mean;
  const float* red_invstd;
  const float* red_gamma;
  const float* red_beta;
  float* red_ws;
  const unsigned* wamax;    // SFOD_F16X3: bits of max|w| of the scaled packed weights (common.h), or nullptr
  // k_conv3x3_m16<.., XF>: x is the PRODUCER's pre-BatchNorm fp32 output [B,H,W,Cin/2] (the same bytes per pixel as operand
  // pairs); the kernel applies relu((x - mean) * (invstd * gamma) + beta) and the (hi, lo) split to each patch slice in LDS
  const float* xf_mean;
  const float* xf_invstd;
  const float* xf_gamma;
  const float* xf_beta;
};

// 16 bytes per lane, global -> LDS (wave-uniform LDS base + lane * 16), through a raw buffer
// resource: a lane whose byte offset is >= num_records (OOB_OFF) gets zeros written -- that is how
// padding pixels, tile overhang and channel tails are produced, without a branch or a zero page.
constexpr unsigned OOB_OFF = 0x80000000u;
__device__ __forceinline__ void bufload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(l), 16, (int)voff, (int)soff, 0, 0);
}

// Stage wait: this wave's DMAs up to the counted point have landed AND its own LDS reads have returned.
// The lgkmcnt(0) matters: the compiler sinks a stage's last fragment reads + MFMAs below the next
// s_barrier; a read that is merely issued when its wave arrives at the barrier can still be in the LDS
// queue when another wave's DMA (issued right after the barrier) overwrites that ring slot / patch
// buffer -- nothing orders a ds_read against an incoming LDS-DMA (seen as rare wrong tiles under load).
// x / d for small operands (x * d < 2^32) with a host-computed magic m = 2^32 / d + 1: the setup of a
// workgroup sits on the critical path before its first DMA, and a runtime integer division is ~40 VALU ops
__device__ __forceinline__ unsigned fdiv(unsigned x, unsigned d, unsigned m) { return d == 1 ? x : __umulhi(x, m); }

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// activation without per-element branches: act 0 identity, 1 ReLU, 2 LeakyReLU(0.2) == max(v, relu ? 0 : v * slope)
// with slope 1 / - / 0.2 (a runtime `if (act == ..)` per element compiled to scalar branches around every value)
struct ActP { float slope; bool relu; };
__device__ __forceinline__ ActP act_params(int act) { return ActP{act == 2 ? 0.2f : 1.f, act == 1}; }
__device__ __forceinline__ float act_f(float v, ActP p) { return fmaxf(v, p.relu ? 0.f : v * p.slope); }

// vmcnt immediates per stage (steady state / last body), see header comment.  Derived by simulating the
// per-wave DMA issue order (patch pieces, then the weights of stage j + D) of each variant.
template <int G, int FM> struct WaitTab;
template <> struct WaitTab<1, 4> {
  static constexpr int N[9] = {2, 3, 4, 4, 4, 4, 3, 2, 2};
  static constexpr int NL[9] = {2, 2, 2, 2, 2, 2, 2, 1, 0};
};
template <> struct WaitTab<2, 2> {
  static constexpr int N[9] = {1, 4, 5, 4, 1, 2, 4, 5, 4};
  static constexpr int NL[9] = {1, 4, 5, 4, 1, 2, 2, 1, 0};
};
template <> struct WaitTab<1, 2> {   // 256-pixel tile: 3 patch pieces per slice (stages 0-2), weights 2 stages ahead
  static constexpr int N[9] = {1, 2, 2, 2, 1, 1, 1, 1, 1};
  static constexpr int NL[9] = {1, 1, 1, 1, 1, 1, 1, 1, 0};
};
template <> struct WaitTab<2, 1> {   // 3 patch pieces per slice (stages 0-2 / 5-7), weights 2 stages ahead
  static constexpr int N[9] = {1, 2, 2, 2, 1, 1, 2, 2, 2};
  static constexpr int NL[9] = {1, 2, 2, 2, 1, 1, 1, 1, 0};
};

// SPLIT (1: SFOD_BF16X3, 2: SFOD_F16X3 -- the MFMA opcode is the only difference): x and w hold (hi, lo) pairs -- per 8
// logical channels 8 hi then 8 lo
// values, so the kernel sees 2 * Cin "physical" bf16 channels and its DMA / LDS side is unchanged; a 64-byte patch
// row is then 16 logical channels as chunks (hi 0-7 | lo 0-7 | hi 8-15 | lo 8-15), ONE MFMA k-step, fed as
// hi*lo + lo*hi + hi*hi: the same four fragment reads as two bf16 k-steps, three MFMAs instead of two.
template <int G, int FM, typename OutT, int SPLIT = 0, bool RED = false>
__global__ void __launch_bounds__(512, (Lay<G, FM>::SMALL ? 4 : 2))   // 2nd arg: waves per SIMD (small tiles: 2 workgroups / CU)
k_conv3x3_patch(P3Args a) {
  using L = Lay<G, FM>;
  constexpr int BN = 128 / G;
  constexpr int FN = 2;                  // 32-col fragments per wave along N
  constexpr int WROWS = FM * 32;         // rows of a wave tile
  constexpr int NPW = L::NPW, PATCH_BYTES = L::PATCH_BYTES, BRING_OFF = L::BRING_OFF, D = L::D, NBS = L::NBS;
  constexpr int PIXTAB_OFF = L::PIXTAB_OFF, SRED_OFF = L::SRED_OFF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (G == 1) ? (wave >> 1) : wave;
  const int wn = (G == 1) ? (wave & 1) : 0;
  const int h = lane >> 5;

  // ---- tile decode (XCD-aware order: workgroups of one XCD walk consecutive tiles) ---------------
  int bid = blockIdx.x;
  {
    const int q = a.ntiles / 8, r = a.ntiles % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  int t = (int)fdiv((unsigned)bid, (unsigned)a.tiles_n, a.m_tn);
  const int tn = bid - t * a.tiles_n;
  int t2 = (int)fdiv((unsigned)t, (unsigned)a.tiles_x, a.m_tx);
  const int txi = t - t2 * a.tiles_x;
  const int b = (int)fdiv((unsigned)t2, (unsigned)a.tiles_y, a.m_ty);
  const int tyi = t2 - b * a.tiles_y;
  const int mtile = (b * a.tiles_y + tyi) * a.tiles_x + txi;
  const int x0 = txi * a.TW, y0 = tyi * a.TH, n0 = tn * BN;
  const int PW = a.PW;
  const int npix = a.TH * a.TW;
  const bf16_t* ximg = a.x + (int64_t)b * a.H * a.W * a.Cin;
  const int Ktot = 9 * a.Cin;

  // ---- DMA descriptors (first: the prologue's loads should leave as early as possible) --------------
  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)ximg, (short)0, a.H * a.W * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.w, (short)0, a.Cout * Ktot * 2, 0x00020000);
  unsigned poff[NPW];  // byte offset inside the image of this lane's 16-byte chunk (slice 0) or OOB_OFF
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int row = (wave * NPW + k) * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ ((row >> 2) & 3);
    const int py = (int)fdiv((unsigned)row, (unsigned)PW, a.m_pw), px = row - py * PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = (py < a.TH + 2) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    poff[k] = ok ? (unsigned)(((iy * a.W + ix) * a.Cin + lc * 8) * 2) : OOB_OFF;
  }
  unsigned boff;  // weight byte offset of this lane's chunk for (tap 0, slice 0) or OOB_OFF
  {
    const int r = wave * 16 + (lane >> 2);
    const int n = r % BN;
    const int lc = (lane & 3) ^ ((n >> 2) & 3);
    boff = (n0 + n < a.Cout) ? (unsigned)(((n0 + n) * Ktot + lc * 8) * 2) : OOB_OFF;
  }
  const int epair = (G == 2) ? (wave >> 2) : 0;  // which (tap, slice) pair of a stage this wave's B DMA feeds

  auto issue_patch = [&](int k, int slice, int buf) {
#ifdef P3_KO_PATCHDMA
    if (slice > 0) return;          // the first slice's patch is reused for every body
#endif
    bufload16(xres, poff[k], (unsigned)slice * 64u, smem + buf * PATCH_BYTES + (wave * NPW + k) * 1024);
  };
  // weights of global stage sg (= body * 9 + j)
  auto issue_b = [&](int sg) {
#ifdef P3_KO_WDMA
    if (sg >= D) return;            // only the prologue's weight stages are ever loaded
#endif
    const int gp = sg * G + epair;  // global pair index
    const int slice = gp / 9, tap = gp - slice * 9;
    bufload16(wres, boff, (unsigned)(tap * a.Cin + slice * 32) * 2u, smem + BRING_OFF + (sg % NBS) * BSLOT + wave * 1024);
  };

  // ---- prologue: first patch + weights of the first D stages (in flight while the tables below are built) --
#pragma unroll
  for (int k = 0; k < NPW; ++k) issue_patch(k, 0, 0);
#pragma unroll
  for (int d = 0; d < D; ++d) issue_b(d);

  // ---- pixel table: tile pixel -> pixel index inside the image, -1 outside (epilogue only) -----------
  int* pixtab = reinterpret_cast<int*>(smem + PIXTAB_OFF);
  {
    const int p = threadIdx.x;
    int v = -1;
    if (p < npix) {
      const int ty = (int)fdiv((unsigned)p, (unsigned)a.TW, a.m_tw), tx = p - ty * a.TW;
      if (y0 + ty < a.H && x0 + tx < a.W) v = (y0 + ty) * a.W + (x0 + tx);
    }
    pixtab[p] = v;
  }

  // ---- fragment addressing -------------------------------------------------------------------------
  int rowA[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int p = wm * WROWS + i * 32 + (lane & 31);
    const int ty = (int)fdiv((unsigned)p, (unsigned)a.TW, a.m_tw);
    rowA[i] = (p < npix) ? (ty * PW + (p - ty * a.TW)) : 0;
  }
  int offB[FN];  // byte offset inside a (pair) slice of the B slot, k-step 0 (k-step 1 = ^32)
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = wn * 64 + j * 32 + (lane & 31);
    offB[j] = n * 64 + (((SPLIT ? 2 * h : h) ^ ((n >> 2) & 3)) << 4);
  }

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one (tap, slice) pair: 2 k-steps of 16 channels
  auto compute_pair = [&](int tap, const unsigned char* patch, const unsigned char* bsl) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int dtap = ky * PW + kx;
    int addrA[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int ra = rowA[i];
      asm volatile("" : "+v"(ra));   // keeps the 9 x FM tap addresses from being hoisted out of the K loop (VGPRs)
      const int row = ra + dtap;
      addrA[i] = row * 64 + (((SPLIT ? 2 * h : h) ^ ((row >> 2) & 3)) << 4);
    }
    if constexpr (SPLIT) {
      bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
#ifdef P3_KO_READS
#pragma unroll
      for (int i = 0; i < FM; ++i) asm volatile("" : "=v"(ah[i]), "=v"(al[i]));     // whatever the registers hold
#pragma unroll
      for (int j = 0; j < FN; ++j) asm volatile("" : "=v"(bh[j]), "=v"(bl[j]));
      (void)addrA; (void)patch; (void)bsl;
#else
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(patch + addrA[i]);
        al[i] = *reinterpret_cast<const bf16x8*>(patch + (addrA[i] ^ 16));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(bsl + offB[j]);
        bl[j] = *reinterpret_cast<const bf16x8*>(bsl + (offB[j] ^ 16));
      }
#endif
#ifdef P3_KO_MFMA
#pragma unroll
      for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
      for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
      acc[0][0][0] += 1.0f;
      return;
#endif
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma_pairs<SPLIT>(ah[i], bl[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma_pairs<SPLIT>(al[i], bh[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = mfma_pairs<SPLIT>(ah[i], bh[j], acc[i][j]);
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(patch + (addrA[i] ^ (s << 5)));
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(bsl + (offB[j] ^ (s << 5)));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };


  const int nstages = a.nbody * 9;
  for (int body = 0; body < a.nbody; ++body) {
    const bool last = (body == a.nbody - 1);
    const int sg0 = body * 9;
    auto stage = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (last) wait_vm<WaitTab<G, FM>::NL[j]>(); else wait_vm<WaitTab<G, FM>::N[j]>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // -- issue: patch pieces first, then the weights of stage j + D
      if constexpr (G == 1) {
        if constexpr (j < NPW) {
          if (!last) issue_patch(j, body + 1, (body + 1) & 1);
        }
      } else if constexpr (FM == 2) {
        // stages 0..3: second slice of this body -> buffer 1; stages 5..8: first slice of the next
        // body -> buffer 0 (its last reader was stage 4)
        if constexpr (j == 0) {
          issue_patch(0, 2 * body + 1, 1);
          issue_patch(1, 2 * body + 1, 1);
        } else if constexpr (j < 4) {
          issue_patch(j + 1, 2 * body + 1, 1);
        } else if constexpr (j == 5) {
          if (!last) { issue_patch(0, 2 * body + 2, 0); issue_patch(1, 2 * body + 2, 0); }
        } else if constexpr (j > 5) {
          if (!last) issue_patch(j - 4, 2 * body + 2, 0);
        }
      } else {
        // 256-pixel tile: 3 pieces per slice, one per stage
        if constexpr (j < 3) issue_patch(j, 2 * body + 1, 1);
        else if constexpr (j >= 5 && j < 8) { if (!last) issue_patch(j - 5, 2 * body + 2, 0); }
      }
      if (sg0 + j + D < nstages) issue_b(sg0 + j + D);
      // -- compute stage j
      const unsigned char* bsl = smem + BRING_OFF + ((sg0 + j) % NBS) * BSLOT;
      // raised priority over the stage's fragment reads + MFMAs (cdna_hip_programming.md T5): +0.7-1 % per layer in a
      // same-box A/B (profiles/r2d_rejected_experiments.txt has the static-priority form, which lost)
      __builtin_amdgcn_s_setprio(1);
      if constexpr (G == 1) {
        compute_pair(j, smem + (body & 1) * PATCH_BYTES, bsl);
      } else {
        compute_pair((2 * j) % 9, smem + ((2 * j) / 9) * PATCH_BYTES, bsl);
        compute_pair((2 * j + 1) % 9, smem + ((2 * j + 1) / 9) * PATCH_BYTES, bsl + BN * 64);
      }
      __builtin_amdgcn_s_setprio(0);
    };
    stage(IC<0>{}); stage(IC<1>{}); stage(IC<2>{}); stage(IC<3>{}); stage(IC<4>{});
    stage(IC<5>{}); stage(IC<6>{}); stage(IC<7>{}); stage(IC<8>{});
  }

  // ---- epilogue -------------------------------------------------------------------------------------
#ifdef P3_KO_EPI
  {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 1.2345e-30f) reinterpret_cast<float*>(a.y)[0] = t;       // keeps the accumulators alive, stores nothing
    return;
  }
#endif
  float bcol[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    bcol[j] = (a.bias != nullptr && n < a.Cout) ? a.bias[n] : 0.f;
  }
  if constexpr (SPLIT == 2) {     // undo the packed weights' power-of-two scale (exact)
    if (a.wamax != nullptr) {
      const float inv = winv_from_absmax(*a.wamax);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
    }
  }
  __syncthreads();  // every wave is done reading operands; LDS is reused below
  const ActP actp = act_params(a.act);
  const int64_t ybase = (int64_t)b * a.H * a.W;
  // interior tile (the common case): every row of every wave is a real pixel -> no row masks
  const bool full = (npix == 8 / (G == 1 ? 2 : 1) * WROWS) && (y0 + a.TH <= a.H) && (x0 + a.TW <= a.W);
  unsigned vmask[FM];  // bit r: row (i, r) of this lane is a real output pixel
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    unsigned m = 0xffffu;
    if (!full) {
      m = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (pixtab[wm * WROWS + ml] >= 0) m |= (1u << r);
      }
    }
    vmask[i] = m;
  }
  if constexpr (sizeof(OutT) == 2) {
    unsigned char* stg = smem + wave * (WROWS * STG_PITCH);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int nl = j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = acc[i][j][r] + bcol[j];
          acc[i][j][r] = v;
          *reinterpret_cast<bf16_t*>(stg + ml * STG_PITCH + nl * 2) = (bf16_t)act_f(v, actp);
        }
      }
    const bool vec_ok = (a.ldy % 8) == 0;
    bf16_t* yo = reinterpret_cast<bf16_t*>(a.y);
#pragma unroll 4
    for (int it = 0; it < WROWS / 8; ++it) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      const int pix = pixtab[wm * WROWS + row];
      const int n = n0 + wn * 64 + ch * 8;
      if (pix < 0 || n >= a.Cout) continue;
      const unsigned char* src = stg + row * STG_PITCH + ch * 16;
      bf16_t* dst = yo + (ybase + pix) * a.ldy + n;
      if (vec_ok && n + 8 <= a.Cout) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
      } else {
        for (int e = 0; e < 8 && n + e < a.Cout; ++e) dst[e] = reinterpret_cast<const bf16_t*>(src)[e];
      }
    }
  } else {
    // fp32 output (bf16x3 mode: every layer).  One 32-row fragment at a time is staged through the wave's LDS region
    // (32 rows x 64 columns fp32, 272-byte pitch) and read back as 16-byte chunks: a store instruction then writes
    // 4 pixels x 256 contiguous bytes instead of 2 x 128 bytes of one dword per lane -- 8 x dwordx4 per lane and
    // fragment instead of 32 x dword (the epilogue of the short-K layers is store-issue bound).
    float* yo = reinterpret_cast<float*>(a.y);
    constexpr int FP = 272;                                   // staged row pitch (bytes)
    static_assert(8 * 32 * FP <= Lay<G, FM>::STG || 8 * 32 * FP <= Lay<G, FM>::OPER, "fp32 staging must fit the operand area");
    unsigned char* stg = smem + wave * (32 * FP);
    const bool vec_ok = (a.ldy % 4) == 0;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int nl = j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = acc[i][j][r] + bcol[j];
          acc[i][j][r] = v;
          *reinterpret_cast<float*>(stg + ml * FP + nl * 4) = act_f(v, actp);
        }
      }
      // same-wave readback (LDS is in order per wave)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4), ch = lane & 15;
        const int pix = pixtab[wm * WROWS + i * 32 + row];
        const int n = n0 + wn * 64 + ch * 4;
        if (pix < 0 || n >= a.Cout) continue;
        const float4 v = *reinterpret_cast<const float4*>(stg + row * FP + ch * 16);
        float* dst = yo + (ybase + pix) * a.ldy + n;
        if (vec_ok && n + 4 <= a.Cout) {
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          const float e[4] = {v.x, v.y, v.z, v.w};
          for (int q = 0; q < 4 && n + q < a.Cout; ++q) dst[q] = e[q];
        }
      }
    }
  }

  if constexpr (RED) {
    // Fused BatchNorm-backward reduction (see P3Args), on the accumulators in their MFMA layout (acc now holds dz): a lane
    // owns one column per N fragment, so the per-channel constants are 4 registers and y is fetched value by value --
    // 32 lanes x 4 B = one 128-byte run per pixel row, like the statistics below.  Kept out of the store loop on
    // purpose: there the constants and sums of 4 channels per lane cost 24 registers that the 128-VGPR variants spill.
    static_assert(sizeof(OutT) == 4, "the reduction epilogue belongs to fp32 data-gradient launches");
    float* sred = reinterpret_cast<float*>(smem + SRED_OFF);     // [wave][64 cols][2]
    const float* __restrict__ yb = a.red_y + ybase * a.ldy;      // this image (wave-uniform base)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * 64 + j * 32 + (lane & 31);
      const int nc = min(n, a.Cout - 1);
      const float mu = a.red_mean[nc], is = a.red_invstd[nc];
      const float sc = is * a.red_gamma[nc], sh = a.red_beta[nc];
      float sb = 0.f, sg = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int pix = pixtab[wm * WROWS + ml];
          const float yv = yb[(unsigned)(max(pix, 0) * a.ldy + nc)];      // 32-bit offset: H * W * C < 2^30 (plan)
          const float d = yv - mu;
          const float g = (pix >= 0 && d * sc + sh > 0.f) ? acc[i][j][r] : 0.f;     // the mask k_bn_bwd_apply recomputes
          sb += g;
          sg = fmaf(g, d * is, sg);
        }
      sb += __shfl_xor(sb, 32);
      sg += __shfl_xor(sg, 32);
      if (h == 0) {
        sred[(wave * 64 + j * 32 + (lane & 31)) * 2 + 0] = sb;
        sred[(wave * 64 + j * 32 + (lane & 31)) * 2 + 1] = sg;
      }
    }
    __syncthreads();
    if (threadIdx.x < BN) {
      const int col = threadIdx.x;
      const int cwn = (G == 1) ? (col >> 6) : 0;
      const int cl = col & 63;
      constexpr int NWM = (G == 1) ? 4 : 8;
      float sb = 0.f, sg = 0.f;
#pragma unroll
      for (int k = 0; k < NWM; ++k) {
        const int wv = (G == 1) ? (k * 2 + cwn) : k;
        sb += sred[(wv * 64 + cl) * 2 + 0];
        sg += sred[(wv * 64 + cl) * 2 + 1];
      }
      const int n = n0 + col;
      if (n < a.Cout) {
        a.red_ws[(int64_t)mtile * 2 * a.Cout + n] = sb;               // k_bn_bwd_finalize: [blk][0..C) = dbeta part,
        a.red_ws[(int64_t)mtile * 2 * a.Cout + a.Cout + n] = sg;      //                     [blk][C..2C) = dgamma part
      }
    }
    if (a.stats != nullptr) __syncthreads();      // sred is reused by the statistics below
  }

  if (a.stats != nullptr) {
    // per-wave (count, sum, M2) of the pre-activation values, combined over the workgroup in fp64
    float* sred = reinterpret_cast<float*>(smem + SRED_OFF);     // [wave][64][2]
    float* scnt = sred + 8 * 64 * 2;                              // [wave]
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < FM; ++i) cnt += __builtin_popcount(vmask[i]);
    cnt += __shfl_xor(cnt, 32);
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float s = 0.f, q = 0.f, mean;
      if (full) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        s += __shfl_xor(s, 32);
        mean = s * inv;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[i][j][r] - mean;
            q = fmaf(d, d, q);
          }
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) s += ((vmask[i] >> r) & 1u) ? acc[i][j][r] : 0.f;
        s += __shfl_xor(s, 32);
        mean = s * inv;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[i][j][r] - mean;
            q += ((vmask[i] >> r) & 1u) ? d * d : 0.f;
          }
      }
      q += __shfl_xor(q, 32);
      if (h == 0) {
        sred[(wave * 64 + j * 32 + (lane & 31)) * 2 + 0] = s;
        sred[(wave * 64 + j * 32 + (lane & 31)) * 2 + 1] = q;
      }
    }
    if (lane == 0) scnt[wave] = (float)cnt;
    __syncthreads();
    if (threadIdx.x < BN) {
      const int col = threadIdx.x;           // column of the workgroup tile
      const int cwn = (G == 1) ? (col >> 6) : 0;
      const int cl = col & 63;
      constexpr int NWM = (G == 1) ? 4 : 8;
      double n_tot = 0.0, s_tot = 0.0;
#pragma unroll
      for (int k = 0; k < NWM; ++k) {
        const int wv = (G == 1) ? (k * 2 + cwn) : k;
        n_tot += (double)scnt[wv];
        s_tot += (double)sred[(wv * 64 + cl) * 2 + 0];
      }
      const double mu = n_tot > 0.0 ? s_tot / n_tot : 0.0;
      double m2 = 0.0;
#pragma unroll
      for (int k = 0; k < NWM; ++k) {
        const int wv = (G == 1) ? (k * 2 + cwn) : k;
        const double nk = (double)scnt[wv];
        if (nk > 0.0) {
          const double d = (double)sred[(wv * 64 + cl) * 2 + 0] / nk - mu;
          m2 += (double)sred[(wv * 64 + cl) * 2 + 1] + nk * d * d;
        }
      }
      const int n = n0 + col;
      if (n < a.Cout) {
        a.stats[((int64_t)mtile * 2 + 0) * a.Cout + n] = (float)s_tot;
        a.stats[((int64_t)mtile * 2 + 1) * a.Cout + n] = (float)m2;
      }
      if (col == 0 && tn == 0) a.stats[(int64_t)a.nblk * 2 * a.Cout + mtile] = (float)n_tot;
    }
  }
}


// =====================================================================================================================
// k_conv3x3_m16 (round 4): the 256-pixel x 128-channel workgroup shape on v_mfma_f32_16x16x32_{bf16,f16}, operand pairs only.
//
// Why: back to back on random operands the halo-patch kernel is bound by the power the chip may draw, not by its
// instruction stream -- the same launch on all-zero operands runs 40 % faster (conv4_2, B = 8: 0.98 -> 0.70 ms,
// profiles/r4_conv_power_bound.txt) -- and per FLOP the 16x16x32 shape moves less accumulator data through the register
// file than 32x32x16 (1 KiB read + written per 16 K MACs at K = 32 instead of 4 KiB per 32 K MACs at K = 16): in a bare
// LDS-read + MFMA loop of this kernel's shape it sustains 1.17x the FLOP/s under the cap
// (tools/experiments/mfma_shape_power.hip).  Measured on this kernel (profiles/r4_m16_*): it holds 1.88 GHz where the
// 32x32x16 form holds 1.65, with the matrix pipe 70 % instead of 75 % busy: +2 ... 6 % per layer back to back, ~1 % inside
// the training step (where BatchNorm passes between the convolutions let the chip clock higher anyway).
//
// K = 32 per instruction = TWO (tap, 16-logical-channel slice) pairs: lane group q = lane >> 4 feeds k = 8q .. 8q+7 from
// pair (q & 1), channels 8 (q >> 1) .. +7, for both operands.  So a stage is two consecutive pairs (u = 2j, 2j+1 of the 18
// pairs of a "super-body" = two 32-physical-channel input slices), 9 stages per super-body, ONE barrier per 48 MFMAs per
// wave (the 32x32x16 form: one per 12).  The patch image, its swizzle and the weight rows are the 32x32x16 kernel's; a
// fragment read is one ds_read_b128 whose lanes address two different taps (pixel operand) / two 8 KiB weight blocks.
//   LDS   patch[2]   2 x 384 rows x 64 B      slice 2*sb -> buffer 0, slice 2*sb+1 -> buffer 1
//         weights[2] 2 x (2 pairs x 128 rows x 64 B): stage g reads slot g & 1 while the DMA fills the other one
//         = 80 KiB exactly = two workgroups per CU; the epilogue's scratch overlays the operands
//   MFMA  A = weight fragment (rows = output channels), B = pixel fragment (columns = pixels): a lane then owns 4
//         consecutive channels of one pixel per tile -> 16-byte stores straight from the accumulators, no LDS staging
//   waves <NW, NIP>: NW = (NW / 2) (pixels) x 2 (channels) waves, each NIP 16-pixel tiles x 64 channels.
//         <8, 4> (shape 5, the default): wave tile 64 x 64, 124 VGPRs, four waves per SIMD, 16 fragment reads per 48 MFMAs;
//         <4, 8> (shape 6): wave tile 128 x 64, 206 VGPRs, two waves per SIMD, 24 reads per 96 MFMAs, pixel fragments
//         double-buffered -- 1-2 % slower, but leaves 96 registers per SIMD lane for a co-resident kernel
//   loop  ONE rolled body per stage: every per-stage constant (tap offsets, ring slot, which patch pieces go out) is a
//         scalar value and the patch DMA offsets are recomputed (~15 VALU beside 48 MFMAs) -- the nine-stage unrolled form
//         of the 32x32x16 kernel made hipcc rotate accumulators through temporaries and spill at this register budget
//   DMA   weights of stage g+1 go out right behind stage g's weight-fragment reads (16 / NW per wave), the <= 2 patch
//         pieces of a stage behind its first MFMA groups; the stage wait allows exactly those patch pieces to be
//         outstanding (slice 1 of this super-body streams in during stages 0-2, slice 0 of the next during 5-7)
//   Tried and dropped (profiles/r4_m16_rejected.txt): a 512-pixel tile with one workgroup per CU, a 4-slot weight ring
//         (three stages ahead) and the two wave halves half a stage apart -- twice the barriers cost more than the stagger
//         and the deeper prefetch returned; in-kernel stamps show no dominant stall in the loop (DMA wait 6 %, barrier 2 %,
//         fragment reads + DMA issue 18 % of a wave's stage, covered by the SIMD's other waves)
template <int FMT>
__device__ __forceinline__ f32x4_t mfma16(bf16x8 a, bf16x8 b, f32x4_t c) {
  if constexpr (FMT == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sfod_f16x8, a), __builtin_bit_cast(sfod_f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// sum over the 16 lanes of a DPP row (every lane of the row ends up with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// counted wait with a run-time (wave-uniform) count: at most n of this wave's DMAs may still be in flight (n > 15: 15)
__device__ __forceinline__ void wait_vm_dyn(int n) {
  switch (n) {
    case 0: wait_vm<0>(); break;   case 1: wait_vm<1>(); break;   case 2: wait_vm<2>(); break;   case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;   case 5: wait_vm<5>(); break;   case 6: wait_vm<6>(); break;   case 7: wait_vm<7>(); break;
    case 8: wait_vm<8>(); break;   case 9: wait_vm<9>(); break;   case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
    case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; default: wait_vm<15>(); break;
  }
}

// NWN = waves along the channel axis: 2 -> 128 output channels per workgroup (the shapes above), 1 -> 64 (round 6: the
// 64-channel tile shapes of the plan -- Cout <= 64 layers, conv5 / RPN at 600x1200 -- on the same loop instead of the
// 32x32x16 kernel's 32 x 64 wave tiles: <4, 4, 1> = 4 waves x (64 px x 64 ch) = 256 px x 64 ch, 64 KiB, two workgroups per CU;
// <8, 4, 1> = 512 px x 64 ch, 96 KiB)
template <int NW, int NIP, int NWN = 2> struct M16Lay {
  static constexpr int PIX = (NW / NWN) * NIP * 16;         // pixels per workgroup: 256 | 512
  static constexpr int PR = (PIX == 256) ? 384 : 640;       // patch rows (pixels incl. halo) per buffer
  static constexpr int NPW = PR / 16 / NW;                  // 1 KiB patch pieces per wave and slice
  static constexpr int PPS = (NPW + 2) / 3;                 // ... issued per stage (stages 0-2 / 5-7)
  static constexpr int PPP = 4 * NWN;                       // 1 KiB weight pieces per (tap, slice) pair: 64 * NWN rows x 64 B
  static constexpr int WPW = 2 * PPP / NW;                  // 1 KiB weight pieces per wave and stage: 4 | 2 | 1
  static constexpr int PATCH_BYTES = PR * 64;
  static constexpr int WR_OFF = 2 * PATCH_BYTES;
  static constexpr int WSLOT = 2 * PPP * 1024;              // 16384 | 8192
  static constexpr int LDS = WR_OFF + 2 * WSLOT;            // 81920 (256 pixels) | 114688; NWN = 1: 65536 | 98304
  static_assert(WPW >= 1 && PPP % WPW == 0 && PPS <= 3, "a wave's weight pieces of a stage lie inside one pair");
};

#ifdef M16_STAMP
// diagnostic build (tools/experiments/m16_knockout.sh, variant STAMP): cycle sums of wave 0 of every workgroup:
// [0] DMA wait, [1] barrier, [2] fragment reads + DMA issue, [3] MFMA phase, [4] stages, [5] kernel cycles, [6] workgroups
__device__ unsigned long long g_m16_stamps[8];
#endif

typedef float __attribute__((address_space(4))) cfloat_k;      // a float in the constant address space

template <int NW, int NIP, int FMT, bool RED, bool XF = false, int NWN = 2>
__global__ void __launch_bounds__(NW * 64, (NW * (160 * 1024 / M16Lay<NW, NIP, NWN>::LDS)) / 4 > 0 ? (NW * (160 * 1024 / M16Lay<NW, NIP, NWN>::LDS)) / 4 : 1)
k_conv3x3_m16(P3Args a) {
  using L = M16Lay<NW, NIP, NWN>;
  static_assert(!XF || NWN == 2, "the in-LDS BatchNorm transform lives in the 128-channel shape");
  static_assert(!XF || (NW == 4 && NIP == 8 && FMT == 1 && !RED), "the in-LDS BatchNorm transform lives in the 4-wave bf16-pair forward kernel");
#ifdef M16_STAMP
  const unsigned long long stamp_t0 = __builtin_amdgcn_s_memtime();
#endif
  constexpr int PR = L::PR, NPW = L::NPW, WPW = L::WPW, PATCH_BYTES = L::PATCH_BYTES, WR_OFF = L::WR_OFF, WSLOT = L::WSLOT;
  constexpr int NWM = NW / NWN;          // waves along the pixel axis
  constexpr int PPP = L::PPP;
  constexpr int WPX = NIP * 16;          // pixels per wave (NIP 16-pixel tiles)
  constexpr int XB = (NIP >= 8) ? 2 : 1; // pixel fragment register sets: the 128-register shape has room for one only
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int q = lane >> 4, r16 = lane & 15;
  const int pb = q & 1, chh = q >> 1;        // which pair of the stage / which 8-channel half this lane group feeds

  // ---- tile decode (as k_conv3x3_patch) ----------------------------------------------------------------------------
  int bid = blockIdx.x;
  {
    const int qq = a.ntiles / 8, r = a.ntiles % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + bid / 8;
  }
  int t = (int)fdiv((unsigned)bid, (unsigned)a.tiles_n, a.m_tn);
  const int tn = bid - t * a.tiles_n;
  int t2 = (int)fdiv((unsigned)t, (unsigned)a.tiles_x, a.m_tx);
  const int txi = t - t2 * a.tiles_x;
  const int b = (int)fdiv((unsigned)t2, (unsigned)a.tiles_y, a.m_ty);
  const int tyi = t2 - b * a.tiles_y;
  const int mtile = (b * a.tiles_y + tyi) * a.tiles_x + txi;
  const int x0 = txi * a.TW, y0 = tyi * a.TH, n0 = tn * (64 * NWN);
  const int PW = a.PW;
  const int npix = a.TH * a.TW;
  const bf16_t* ximg = a.x + (int64_t)b * a.H * a.W * a.Cin;
  const int Ktot = 9 * a.Cin;

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc((void*)ximg, (short)0, a.H * a.W * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, a.Cout * Ktot * 2, 0x00020000);
  // byte offset inside the image of this lane's 16-byte chunk of patch piece k (slice 0), or OOB_OFF.  Recomputed at every
  // use (~15 VALU instructions beside 96 MFMAs) instead of living in NPW registers: the stage loop below is ONE rolled
  // loop body whose patch piece index is a run-time value
  auto patch_off = [&](int k) -> unsigned {
    const int row = (wave * NPW + k) * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ ((row >> 2) & 3);
    const int py = (int)fdiv((unsigned)row, (unsigned)PW, a.m_pw), px = row - py * PW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = (py < a.TH + 2) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    return ok ? (unsigned)(((iy * a.W + ix) * a.Cin + lc * 8) * 2) : OOB_OFF;
  };
  // weight pieces of a stage: 2 PPP x 1 KiB = (pair a: 64 NWN rows, pair b: 64 NWN rows); wave w issues pieces w*WPW .. +WPW-1
  unsigned boff[WPW];
#pragma unroll
  for (int k = 0; k < WPW; ++k) {
    const int n = ((wave * WPW + k) & (PPP - 1)) * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ ((n >> 2) & 3);
    boff[k] = (n0 + n < a.Cout) ? (unsigned)(((n0 + n) * Ktot + lc * 8) * 2) : OOB_OFF;
  }
  const int wpair = (wave * WPW) / PPP;      // which pair of a stage this wave's weight pieces belong to
  auto issue_patch = [&](int k, int slice, int buf) {
    bufload16(xres, patch_off(k), (unsigned)__builtin_amdgcn_readfirstlane(slice * 64),
              smem + buf * PATCH_BYTES + (wave * NPW + k) * 1024);
  };
  // (tap, slice) of pair u (0 .. 17) of super-body sb: u < 9: (u, 2 sb), else (u - 9, 2 sb + 1)
  // this wave's share of the weights of global pair index gu = 18 * sb + u (its pair of the stage) -> stage slot `slot`
  auto issue_w = [&](int sbi, int u0, int slot) {
    const int u = u0 + wpair;
    const int sl = (u >= 9) ? 1 : 0, tap = u - 9 * sl;
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((tap * a.Cin + (2 * sbi + sl) * 32) * 2);
#pragma unroll
    for (int k = 0; k < WPW; ++k)
      bufload16(wres, boff[k], so, smem + WR_OFF + slot * WSLOT + (wave * WPW + k) * 1024);
  };

  // ---- prologue ----------------------------------------------------------------------------------------------------
  const int nsb = a.nbody;          // Cin / 64
  const int nst = nsb * 9;          // stages
#pragma unroll
  for (int k = 0; k < NPW; ++k) issue_patch(k, 0, 0);
  issue_w(0, 0, 0);

  // ---- XF: fp32 y -> relu(bn(y)) -> (hi, lo) pairs, in place in a patch buffer -----------------------------------------
  // A patch row is 64 bytes = 16 fp32 channels of one patch pixel = the 16 logical channels of the slice; its operand image is
  // hi 0-7 | lo 0-7 | hi 8-15 | lo 8-15 (16-byte chunks at the row's swizzled positions).  The unit of work is a HALF row
  // (8 channels: read two chunks, write the same two): 384 rows x 2 halves = 3 units per thread, all lanes busy, nobody else
  // touches a unit's 32 bytes.  Unit k of thread t: k = 0 -> (row t, half 0); k = 1 -> (row t + 256, half 0) in waves 0-1,
  // (row t - 128, half 1) in waves 2-3; k = 2 -> (row t + 128, half 1): the half is wave-uniform, so the BatchNorm
  // coefficients are scalar loads.  Rows outside the image (the DMA wrote zeros: padding, tile overhang) must stay zero --
  // relu(beta - mean * scale) is not -- so each lane keeps a validity bit per unit.  The arithmetic is k_bn_relu_pool_fwd's,
  // operation for operation (the fused and the unfused path agree bit for bit).
  unsigned xf_valid = 0;
  int xf_row[3] = {0, 0, 0};
  const int xf_g1 = __builtin_amdgcn_readfirstlane(wave >= 2 ? 1 : 0);
  if constexpr (XF) {
    const int tid = (int)threadIdx.x;
    xf_row[0] = tid;
    xf_row[1] = (wave >= 2) ? tid - 128 : tid + 256;
    xf_row[2] = tid + 128;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int row = xf_row[k];
      const int py = (int)fdiv((unsigned)row, (unsigned)PW, a.m_pw), px = row - py * PW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      const bool ok = (py < a.TH + 2) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      xf_valid |= ok ? (1u << k) : 0u;
    }
  }
  auto transform = [&](int buf, int slice) {
    if constexpr (XF) {
      const int pbuf = buf * PATCH_BYTES;
      const int c0 = __builtin_amdgcn_readfirstlane(slice * 16);
      float4 v[3][2];
      int pa[3][2];                          // byte offsets in the LDS image (32-bit: pointers would double the registers)
#pragma unroll
      for (int k = 0; k < 3; ++k) {          // all six reads first: one LDS latency, not three
        const int g = (k == 0) ? 0 : (k == 2 ? 1 : xf_g1);
        const int row = xf_row[k], sw = (row >> 2) & 3;
        pa[k][0] = pbuf + row * 64 + (((2 * g) ^ sw) << 4);
        pa[k][1] = pbuf + row * 64 + (((2 * g + 1) ^ sw) << 4);
        v[k][0] = *reinterpret_cast<const float4*>(smem + pa[k][0]);       // channels 8 g .. 8 g + 3
        v[k][1] = *reinterpret_cast<const float4*>(smem + pa[k][1]);       // channels 8 g + 4 .. 8 g + 7
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int g = (k == 0) ? 0 : (k == 2 ? 1 : xf_g1);
        const int cb = c0 + 8 * g;                                  // wave-uniform: scalar loads
        const float yv[8] = {v[k][0].x, v[k][0].y, v[k][0].z, v[k][0].w, v[k][1].x, v[k][1].y, v[k][1].z, v[k][1].w};
        // relu and the validity select in one op: med3(t, 0, +inf) = max(t, 0), med3(t, 0, 0) = 0
        const float top = ((xf_valid >> k) & 1u) ? __builtin_inff() : 0.f;
        union { unsigned u[4]; uint4 q; } hi, lo;
        // constant address space: a wave-uniform load from it is a scalar load (s_load_dwordx8).  As plain global loads
        // these were VECTOR loads -- the kernel stores to global memory, so the compiler will not call the arrays invariant
        // -- and their vmcnt(0) waits drained the LDS-DMA pipeline in every transform.
        const cfloat_k* k_mean = (const cfloat_k*)(uintptr_t)a.xf_mean + cb;
        const cfloat_k* k_invstd = (const cfloat_k*)(uintptr_t)a.xf_invstd + cb;
        const cfloat_k* k_gamma = (const cfloat_k*)(uintptr_t)a.xf_gamma + cb;
        const cfloat_k* k_beta = (const cfloat_k*)(uintptr_t)a.xf_beta + cb;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          float z[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float sc = k_invstd[e + i] * k_gamma[e + i];
            const float t = fmaf(yv[e + i] - k_mean[e + i], sc, k_beta[e + i]);
            z[i] = __builtin_amdgcn_fmed3f(t, 0.f, top);
          }
          // two channels per conversion (v_cvt_pk_bf16_f32); the high parts back as floats are a shift and a mask
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{z[0], z[1]}, bf16x2_t));
          const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
          hi.u[e >> 1] = h;
          lo.u[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{z[0] - h0, z[1] - h1}, bf16x2_t));
        }
        *reinterpret_cast<uint4*>(smem + pa[k][0]) = hi.q;
        *reinterpret_cast<uint4*>(smem + pa[k][1]) = lo.q;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (XF) {
    wait_vm<WPW>();                 // the first patch has landed (this wave's pieces); the first weights may still be in flight
    __builtin_amdgcn_s_barrier();
    transform(0, 0);
  }

  // ---- fragment addressing -----------------------------------------------------------------------------------------
  int rowA[NIP];
#pragma unroll
  for (int i = 0; i < NIP; ++i) {
    const int p = wm * WPX + i * 16 + r16;
    const int ty = (int)fdiv((unsigned)p, (unsigned)a.TW, a.m_tw);
    rowA[i] = (p < npix) ? (ty * PW + (p - ty * a.TW)) : 0;
#ifdef M16_V_NOCONF      // knock-out: the 16 pixels of a lane group on 16 consecutive patch rows (wrong pixels, no bank conflicts)
    rowA[i] = p;
#endif
    asm volatile("" : "+v"(rowA[i]));       // materialised here, not re-derived from spilled 64-bit products inside the K loop
  }
  // weight fragment of channel tile ic: + ic * 1024; lo = ^ 16   ((n >> 2) & 3 == (r16 >> 2) & 3 for n = 16 m + r16)
  const int offW = pb * (PPP * 1024) + (wn * 64 + r16) * 64 + (((2 * chh) ^ ((r16 >> 2) & 3)) << 4);

  f32x4_t acc[4][NIP];     // [channel tile][pixel tile]: channel = wn*64 + ic*16 + q*4 + r, pixel = wm*128 + ip*16 + r16
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NIP; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // patch row offset of this lane group's pair in stage j of a super-body: buffer * PR rows + tap shift (PR % 16 == 0: the
  // swizzle of the buffer-relative row is the swizzle of the global row).  Pairs ua = 2j, ub = 2j + 1.
  auto stage_offq = [&](int j) -> int {
    const int ua = 2 * j, ub = 2 * j + 1;
    const int sa = (ua >= 9) ? 1 : 0, ta = ua - 9 * sa, sbb = (ub >= 9) ? 1 : 0, tb = ub - 9 * sbb;
    const int kya = (ta * 11) >> 5, kyb = (tb * 11) >> 5;            // t / 3 for t < 9
    const int dta = sa * PR + kya * PW + (ta - 3 * kya), dtb = sbb * PR + kyb * PW + (tb - 3 * kyb);
#ifdef M16_V_NOCONF      // ... and both pairs of a stage 0 mod 16 rows apart
    return pb ? (dtb & ~15) : (dta & ~15);
#endif
    return pb ? dtb : dta;
  };

  {
  // ---- K loop: one rolled body per stage (j = stage inside the super-body sb; all stage constants are scalar values)
  int sb = 0, j = 0, par = 0, nprev = 0;      // nprev: patch pieces the previous stage issued behind its weights
#ifdef M16_V_EXTRAVALU
  int extra_valu = (int)threadIdx.x;
#endif
#ifdef M16_STAMP
  unsigned long long stamp_sum[4] = {0, 0, 0, 0};
#endif
#pragma unroll 1
  for (int g = 0; g < nst; ++g) {
    const bool last = (sb == nsb - 1);
    // the DMAs of the previous stage: weights (needed now) first, then its patch pieces (may stay in flight)
#ifdef M16_STAMP
    const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
#ifndef M16_KO_BAR
    if (nprev == 0) wait_vm<0>();
    else if (nprev == 1) wait_vm<1>();
    else if (nprev == 2) wait_vm<2>();
    else wait_vm<3>();
#ifdef M16_STAMP
    const unsigned long long st_a2 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
#ifdef M16_STAMP
    const unsigned long long st_b = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (XF) {             // the slice that landed during the last two stages: convert it in place, one stage before its first use
      if (j == 3) transform(1, 2 * sb + 1);
      else if (j == 8 && sb + 1 < nsb) transform(0, 2 * sb + 2);
      __builtin_amdgcn_sched_barrier(0);
    }

    const unsigned char* wsl = smem + WR_OFF + par * WSLOT;
    bf16x8 wh[4], wl[4];
#pragma unroll
    for (int ic = 0; ic < 4; ++ic) {
      wh[ic] = *reinterpret_cast<const bf16x8*>(wsl + ic * 1024 + offW);
      wl[ic] = *reinterpret_cast<const bf16x8*>(wsl + ic * 1024 + (offW ^ 16));
    }
    const int offq = stage_offq(j);
    bf16x8 xh[XB], xl[XB];
    auto read_x = [&](int ip, int slot) {
      const int row = rowA[ip] + offq;
      const int ad = row * 64 + (((2 * chh) ^ ((row >> 2) & 3)) << 4);
      xh[slot] = *reinterpret_cast<const bf16x8*>(smem + ad);
      xl[slot] = *reinterpret_cast<const bf16x8*>(smem + (ad ^ 16));
    };
    read_x(0, 0);
    // the next stage's weights: behind the fragment reads (their LDS latency covers the issue)
#if !defined(M16_KO_WDMA) && !defined(M16_V_LATEW)
    if (j < 8) issue_w(sb, 2 * j + 2, par ^ 1);
    else if (!last) issue_w(sb + 1, 0, par ^ 1);
#endif
    // patch pieces of this stage (<= PPS): slice 1 of this super-body -> buffer 1 in stages 0-2, slice 0 of the next -> buffer 0 in 5-7
    // (XF: all pieces in TWO stages, 0-1 / 5-6, so that the slice has landed one stage before its first use and is
    // converted in place during that stage: 3 / 8)
    constexpr int PPS = XF ? (NPW + 1) / 2 : L::PPS;
    int np = 0, pslice = 0, pbuf = 0, pk = 0;
    if (j < (XF ? 2 : 3)) { pk = PPS * j; np = min(PPS, NPW - pk); pslice = 2 * sb + 1; pbuf = 1; }
    else if (j >= 5 && j < (XF ? 7 : 8) && !last) { pk = PPS * (j - 5); np = min(PPS, NPW - pk); pslice = 2 * sb + 2; pbuf = 0; }
    __builtin_amdgcn_sched_barrier(0);
#ifdef M16_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long st_c = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ip = 0; ip < NIP; ++ip) {
      if constexpr (XB == 2) { if (ip + 1 < NIP) read_x(ip + 1, (ip + 1) & 1); }
      else { if (ip > 0) read_x(ip, 0); }
      const bf16x8 ch = xh[ip % XB], cl = xl[ip % XB];
#ifdef M16_KO_MFMA
#pragma unroll
      for (int ic = 0; ic < 4; ++ic) asm volatile("" ::"v"(wh[ic]), "v"(wl[ic]));
      asm volatile("" ::"v"(ch), "v"(cl));
      acc[0][ip][0] += 1.0f;
#else
#pragma unroll
      for (int ic = 0; ic < 4; ++ic) acc[ic][ip] = mfma16<FMT>(wh[ic], cl, acc[ic][ip]);
#pragma unroll
      for (int ic = 0; ic < 4; ++ic) acc[ic][ip] = mfma16<FMT>(wl[ic], ch, acc[ic][ip]);
#pragma unroll
      for (int ic = 0; ic < 4; ++ic) acc[ic][ip] = mfma16<FMT>(wh[ic], ch, acc[ic][ip]);
#endif
#ifndef M16_KO_PDMA
      if (ip < PPS) { if (ip < np) issue_patch(pk + ip, pslice, pbuf); }
#endif
#ifdef M16_V_EXTRAVALU      // perturbation: M16_V_EXTRAVALU extra integer VALU instructions per stage (spread over the pixel tiles)
#pragma unroll
      for (int e = 0; e < (M16_V_EXTRAVALU + NIP - 1) / NIP; ++e) asm volatile("v_add_u32 %0, %0, %0" : "+v"(extra_valu));
#endif
#if defined(M16_V_LATEW) && !defined(M16_KO_WDMA)
      if (ip == 2) {
        if (j < 8) issue_w(sb, 2 * j + 2, par ^ 1);
        else if (!last) issue_w(sb + 1, 0, par ^ 1);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
#ifdef M16_STAMP
    {
      // all MFMAs of the stage issued (the last one may still be in the pipe): stamp and accumulate the four phases
      const unsigned long long st_d = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp_sum[0] += st_a2 - st_a;      // own DMA wait (+ own LDS reads)
      stamp_sum[1] += st_b - st_a2;      // barrier
      stamp_sum[2] += st_c - st_b;       // weight + first pixel fragment reads, DMA issue
      stamp_sum[3] += st_d - st_c;       // 96 MFMAs with the streamed fragment reads and the patch DMA issue
    }
#endif
    nprev = np;
    par ^= 1;
    if (++j == 9) { j = 0; ++sb; }
  }
#ifdef M16_STAMP
  if (threadIdx.x == 0) {
    atomicAdd(&g_m16_stamps[0], stamp_sum[0]); atomicAdd(&g_m16_stamps[1], stamp_sum[1]);
    atomicAdd(&g_m16_stamps[2], stamp_sum[2]); atomicAdd(&g_m16_stamps[3], stamp_sum[3]);
    atomicAdd(&g_m16_stamps[4], (unsigned long long)nst);
    atomicAdd(&g_m16_stamps[5], __builtin_amdgcn_s_memtime() - stamp_t0);
    atomicAdd(&g_m16_stamps[6], 1ull);
  }
#endif
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
#ifdef M16_KO_EPI
  {
    float tt = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < NIP; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) tt += acc[i][jj][r];
    if (tt == 1.2345e-30f) reinterpret_cast<float*>(a.y)[0] = tt;       // keeps the accumulators alive, stores nothing
    return;
  }
#endif
  if constexpr (FMT == 2) {       // undo the packed weights' power-of-two scale (exact)
    if (a.wamax != nullptr) {
      const float inv = winv_from_absmax(*a.wamax);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NIP; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] *= inv;
    }
  }
  const ActP actp = act_params(a.act);
  const int64_t ybase = (int64_t)b * a.H * a.W;
  int pix[NIP];          // pixel index inside the image of this lane's pixel of tile ip, -1: not an output pixel
#pragma unroll
  for (int ip = 0; ip < NIP; ++ip) {
    const int p = wm * WPX + ip * 16 + r16;
    int v = -1;
    if (p < npix) {
      const int ty = (int)fdiv((unsigned)p, (unsigned)a.TW, a.m_tw), tx = p - ty * a.TW;
      if (y0 + ty < a.H && x0 + tx < a.W) v = (y0 + ty) * a.W + (x0 + tx);
    }
    pix[ip] = v;
  }
  const int cb = n0 + wn * 64 + q * 4;          // first of this lane's 4 channels in channel tile 0 (+ ic * 16)
  float* yo = reinterpret_cast<float*>(a.y);
  // every channel of this wave's 64 exists and rows are 16-byte aligned (wave-uniform; the common case): 32 x dwordx4
  // per lane, back to back -- a lane group of 16 lanes x 4 rows writes 64 contiguous bytes of each of its 16 pixels
  const bool fast = (a.ldy % 4) == 0 && (n0 + wn * 64 + 64 <= a.Cout);
  float bv[4][4];
#pragma unroll
  for (int ic = 0; ic < 4; ++ic)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = cb + ic * 16 + r;
      bv[ic][r] = (a.bias != nullptr && c < a.Cout) ? a.bias[c] : 0.f;
    }
  int64_t rowoff[NIP];
#pragma unroll
  for (int ip = 0; ip < NIP; ++ip) rowoff[ip] = (ybase + max(pix[ip], 0)) * a.ldy;
#pragma unroll
  for (int ic = 0; ic < 4; ++ic)
#pragma unroll
    for (int ip = 0; ip < NIP; ++ip)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[ic][ip][r] += bv[ic][r];          // pre-activation value: what the statistics are taken of
  if (fast) {
#pragma unroll
    for (int ic = 0; ic < 4; ++ic)
#pragma unroll
      for (int ip = 0; ip < NIP; ++ip) {
        float4 o;
        o.x = act_f(acc[ic][ip][0], actp); o.y = act_f(acc[ic][ip][1], actp);
        o.z = act_f(acc[ic][ip][2], actp); o.w = act_f(acc[ic][ip][3], actp);
        if (pix[ip] >= 0) *reinterpret_cast<float4*>(yo + rowoff[ip] + (cb + ic * 16)) = o;
      }
  } else {
#pragma unroll
    for (int ic = 0; ic < 4; ++ic)
#pragma unroll
      for (int ip = 0; ip < NIP; ++ip)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = cb + ic * 16 + r;
          if (pix[ip] >= 0 && c < a.Cout) yo[rowoff[ip] + c] = act_f(acc[ic][ip][r], actp);
        }
  }

  if constexpr (!RED) { if (a.stats == nullptr) return; }
  __syncthreads();       // every wave is done reading operands: LDS is reused below
  float* sred = reinterpret_cast<float*>(smem);        // [wave][64 channels][2]
  float* scnt = sred + NW * 64 * 2;                    // [wave]

  if constexpr (RED) {
    // BatchNorm-backward partial sums of the tensor this launch writes (see P3Args): g = dz * [bn(y) > 0], (sum g, sum g xhat)
    const float* __restrict__ yb = a.red_y + ybase * a.ldy;
#pragma unroll
    for (int ic = 0; ic < 4; ++ic) {
      const int c = cb + ic * 16;
      float mu[4], is[4], sc[4], sh[4], sbv[4], sgv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nc = min(c + r, a.Cout - 1);
        mu[r] = a.red_mean[nc]; is[r] = a.red_invstd[nc]; sc[r] = is[r] * a.red_gamma[nc]; sh[r] = a.red_beta[nc];
        sbv[r] = 0.f; sgv[r] = 0.f;
      }
      const int cc = min(c, a.Cout - 4);             // dense fp32 output, Cout % 4 == 0 (checked at launch)
#pragma unroll
      for (int ip = 0; ip < NIP; ++ip) {
        const float4 yv4 = *reinterpret_cast<const float4*>(yb + (unsigned)(max(pix[ip], 0) * a.ldy + cc));
        const float* yv = &yv4.x;
        const bool ok = pix[ip] >= 0 && c < a.Cout;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = yv[r] - mu[r];
          const float g = (ok && d * sc[r] + sh[r] > 0.f) ? acc[ic][ip][r] : 0.f;      // the mask k_bn_bwd_apply recomputes
          sbv[r] += g;
          sgv[r] = fmaf(g, d * is[r], sgv[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = row16_sum(sbv[r]), s2 = row16_sum(sgv[r]);
        if (r16 == 0) {
          sred[(wave * 64 + ic * 16 + q * 4 + r) * 2 + 0] = s1;
          sred[(wave * 64 + ic * 16 + q * 4 + r) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < 64 * NWN) {
      const int col = threadIdx.x, cwn = col >> 6, cl = col & 63;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < NWM; ++k) {
        s1 += sred[((k * NWN + cwn) * 64 + cl) * 2 + 0];
        s2 += sred[((k * NWN + cwn) * 64 + cl) * 2 + 1];
      }
      const int n = n0 + col;
      if (n < a.Cout) {
        a.red_ws[(int64_t)mtile * 2 * a.Cout + n] = s1;
        a.red_ws[(int64_t)mtile * 2 * a.Cout + a.Cout + n] = s2;
      }
    }
    if (a.stats == nullptr) return;
    __syncthreads();
  }

  {
    // per-wave (count, sum, M2 about the wave's mean) of the pre-activation values, combined over the workgroup in fp64
    int cnt = 0;
#pragma unroll
    for (int ip = 0; ip < NIP; ++ip) cnt += (pix[ip] >= 0) ? 1 : 0;
    cnt = (int)row16_sum((float)cnt);                       // <= 128: exact
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
    for (int ic = 0; ic < 4; ++ic)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f;
#pragma unroll
        for (int ip = 0; ip < NIP; ++ip) s += (pix[ip] >= 0) ? acc[ic][ip][r] : 0.f;
        s = row16_sum(s);
        const float mean = s * inv;
        float m2 = 0.f;
#pragma unroll
        for (int ip = 0; ip < NIP; ++ip) {
          const float d = acc[ic][ip][r] - mean;
          m2 += (pix[ip] >= 0) ? d * d : 0.f;
        }
        m2 = row16_sum(m2);
        if (r16 == 0) {
          sred[(wave * 64 + ic * 16 + q * 4 + r) * 2 + 0] = s;
          sred[(wave * 64 + ic * 16 + q * 4 + r) * 2 + 1] = m2;
        }
      }
    if (lane == 0) scnt[wave] = (float)cnt;
    __syncthreads();
    if (threadIdx.x < 64 * NWN) {
      const int col = threadIdx.x, cwn = col >> 6, cl = col & 63;
      double n_tot = 0.0, s_tot = 0.0;
#pragma unroll
      for (int k = 0; k < NWM; ++k) {
        const int wv = k * NWN + cwn;
        n_tot += (double)scnt[wv];
        s_tot += (double)sred[(wv * 64 + cl) * 2 + 0];
      }
      const double mu = n_tot > 0.0 ? s_tot / n_tot : 0.0;
      double m2 = 0.0;
#pragma unroll
      for (int k = 0; k < NWM; ++k) {
        const int wv = k * NWN + cwn;
        const double nk = (double)scnt[wv];
        if (nk > 0.0) {
          const double d = (double)sred[(wv * 64 + cl) * 2 + 0] / nk - mu;
          m2 += (double)sred[(wv * 64 + cl) * 2 + 1] + nk * d * d;
        }
      }
      const int n = n0 + col;
      if (n < a.Cout) {
        a.stats[((int64_t)mtile * 2 + 0) * a.Cout + n] = (float)s_tot;
        a.stats[((int64_t)mtile * 2 + 1) * a.Cout + n] = (float)m2;
      }
      if (col == 0 && tn == 0) a.stats[(int64_t)a.nblk * 2 * a.Cout + mtile] = (float)n_tot;
    }
  }
}

}  // namespace

// Workgroup shape + tile shape.  Variants (all 8 waves):
//   1  G=1 FM=4  512 px x 128 ch, wave tile 128 x 64, one workgroup per CU (112 KiB of LDS, 212 VGPRs)
//   2  G=1 FM=2  256 px x 128 ch, wave tile  64 x 64, two workgroups per CU (72 KiB, 114 VGPRs)
//   3  G=2 FM=1  256 px x  64 ch, wave tile  32 x 64, two workgroups per CU (76 KiB,  72 VGPRs)
//   4  G=2 FM=2  512 px x  64 ch, wave tile  64 x 64, one workgroup per CU
//   5  shape 2 on v_mfma_f32_16x16x32 (k_conv3x3_m16<8, 4>: 8 waves, wave tile 64 x 64, two workgroups per CU; operand pairs
//      with Cin % 32 == 0 only, everything else runs as 2).  The automatic choice takes it wherever it picks shape 2
//      (SFOD_P3_M16=0 / sfod_set_conv3x3_m16(0): the 32x32x16 kernel everywhere, for A/B runs)
//   6  the same on 4 waves per workgroup (k_conv3x3_m16<4, 8>, wave tile 128 x 64)
//   7  shape 3 on v_mfma_f32_16x16x32 (round 6: k_conv3x3_m16<4, 4, NWN = 1>: 4 waves x (64 px x 64 ch), 64 KiB, two workgroups
//      per CU; operand pairs with Cin % 32 == 0).  Slower than shape 3 on every layer measured (see sfod_p3_launch): never the
//      automatic choice unless SFOD_P3_M16_N64=1
//   8  shape 4 on it (k_conv3x3_m16<8, 4, NWN = 1>: 512 px x 64 ch, 96 KiB, one workgroup per CU)
//   9  shape 3 on it with 8 waves x (32 px x 64 ch) (k_conv3x3_m16<8, 2, NWN = 1>: 64 KiB, two workgroups = 16 waves per CU)
// Measured per layer (tools/bench_conv.py, interleaved A/B; profiles/r1q_conv_variants.txt): two resident
// workgroups overlap each other's prologue / epilogue / barrier stalls, which beats the larger tiles' lower
// L2 -> LDS traffic on every VGG shape; between the two small shapes the 64 x 64 wave tile needs one LDS
// fragment read per MFMA instead of 1.5 (the 32 x 64 tile keeps the LDS array ~100 % busy at full MFMA rate)
// and wins by 7-18 % wherever its 128-channel tiles still fill the chip (>= 512 workgroups).
// SFOD_P3_VARIANT=1..4 / sfod_set_conv3x3_variant force a shape where the channel counts allow it (A/B, tests).
// process-wide tuning knob for A/B runs and tests (relaxed atomic: a plain word, no ordering needed); -1: not
// initialised (SFOD_P3_VARIANT or 0 = auto).  It selects among kernels that compute the same values.
static std::atomic<int> g_p3_variant{-1};

// 16x16x32 form of the 256 x 128 shape for operand pairs (k_conv3x3_m16): -1 not initialised (SFOD_P3_M16, default on)
static std::atomic<int> g_p3_m16{-1};
extern "C" int sfod_set_conv3x3_m16(int on) {
  g_p3_m16.store((on >= 0 && on <= 2) ? on : 1, std::memory_order_relaxed);
  return 0;
}
static int p3_m16_n64_enabled() {
  static const int on = []() { const char* e = getenv("SFOD_P3_M16_N64"); return e ? atoi(e) : 0; }();
  return on;
}
static int p3_m16_enabled() {       // 0 off, 1 the 8-wave form (default), 2 the 4-wave form (SFOD_P3_M16=2: co-residency experiments)
  int v = g_p3_m16.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* ev = getenv("SFOD_P3_M16");
    int want = ev ? atoi(ev) : 1, expect = -1;
    if (want < 0 || want > 2) want = 1;
    g_p3_m16.compare_exchange_strong(expect, want, std::memory_order_relaxed);
    v = g_p3_m16.load(std::memory_order_relaxed);
  }
  return v;
}

extern "C" int sfod_set_conv3x3_variant(int variant) {
  g_p3_variant.store((variant >= 1 && variant <= 9) ? variant : 0, std::memory_order_relaxed);
  return 0;
}

// tile shape for a BM-pixel workgroup: maximise covered-output efficiency under TH*TW <= BM and
// (TH+2)*(TW+2) <= patch capacity; returns false if nothing fits
static bool p3_best_tile(int H, int W, int BM, int cap, int& TH, int& TW, int& tiles_y, int& tiles_x) {
  double best = -1.0;
  for (int tw = 4; tw <= 128 && tw <= W + 3; ++tw) {
    int th = BM / tw;
    while (th > 1 && (th + 2) * (tw + 2) > cap) --th;
    if (th > H) th = H;
    if (th < 1 || (th + 2) * (tw + 2) > cap) continue;
    const int ty = (H + th - 1) / th, tx = (W + tw - 1) / tw;
    // even out the rows so that the last tile row is not nearly empty
    th = (H + ty - 1) / ty;
    const double eff = (double)H * W / ((double)ty * tx * (double)BM);
    // tie-break towards wide tiles (longer contiguous runs per patch row)
    // tile widths that are a multiple of 32 keep the 32 pixels of an A fragment contiguous in the patch, i.e. its
    // ds_read_b128 lane groups conflict-free (16 rows distinct mod 16); worth ~2 % of pixel efficiency (300x600 maps:
    // 8x32 at 0.974 beats 10x25 at 0.977 by 2-3 %, profiles/r2d_rejected_experiments.txt)
    const double score = eff + 1e-6 * tw + ((tw % 32 == 0) ? 0.02 : 0.0);
    if (score > best) { best = score; TH = th; TW = tw; tiles_y = ty; tiles_x = tx; }
  }
  return best >= 0.0;
}

P3Plan sfod_p3_plan(int B, int H, int W, int Cin, int Cout, int pairs) {
  P3Plan p;
  p.ok = 0;
  p.m16 = 0;
  if (H < 1 || W < 1 || B < 1) return p;
  int variant = g_p3_variant.load(std::memory_order_relaxed);
  if (variant < 0) {
    const char* ev = getenv("SFOD_P3_VARIANT");
    variant = ev ? atoi(ev) : 0;
    int expect = -1;
    g_p3_variant.compare_exchange_strong(expect, variant, std::memory_order_relaxed);   // a concurrent setter wins
    variant = g_p3_variant.load(std::memory_order_relaxed);
  }
  p.m16 = (variant == 5) ? 1 : ((variant == 7 || variant == 8) ? 3 : (variant == 9 ? 4 : (variant == 6 ? 2 : ((variant < 1 || variant > 9) ? p3_m16_enabled() : 0))));
  if (variant == 5 || variant == 6) variant = 2;
  if (variant == 7 || variant == 9) variant = 3;        // the 64-channel tile shapes on the 16x16x32 loop (k_conv3x3_m16<.., NWN = 1>)
  if (variant == 8) variant = 4;
  if (variant < 1 || variant > 4) {
    const int64_t mt = ((int64_t)B * H * W + 255) / 256;          // 256-pixel tiles (lower bound)
    const int64_t wg128 = mt * ((Cout + 127) / 128);
    variant = (Cout > 64 && wg128 >= 512) ? 2 : 3;
    // long-K layers whose 512 x 128 tiles fill whole rounds of the 256 CUs (one workgroup per CU): the big tile's
    // 2x lower L2 -> LDS traffic wins by ~4 % there (1024x2048 frames: conv4_2 / conv5); with a ragged last
    // round or tiles that overhang the map (600x1200 frames) the two-workgroups-per-CU shape stays ahead
    if (variant == 2 && Cin >= 512 && Cout >= 512) {
      int th, tw, ty, tx;
      if (p3_best_tile(H, W, 512, PATCH_ROWS_BIG, th, tw, ty, tx)) {
        const int64_t wgs = (int64_t)B * ty * tx * ((Cout + 127) / 128);
        const int64_t rounds = (wgs + 255) / 256;
        const double px_eff = (double)H * W / ((double)ty * tx * 512.0);     // 75x150 maps: 0.915 -> stays on 256 x 128
        if (wgs >= 512 && (double)wgs / (double)(rounds * 256) >= 0.97 && px_eff >= 0.97) variant = 1;
      }
    }
  }
  if ((variant == 3 || variant == 4) && Cin % 64 != 0) variant = (variant == 3) ? 2 : 1;
  p.G = (variant <= 2) ? 1 : 2;
  p.FM = (variant == 1) ? 4 : (variant == 3 ? 1 : 2);
  if (Cin % (32 * p.G) != 0) return p;
  if ((int64_t)H * W * Cin >= (int64_t)1 << 30 || (int64_t)Cout * 9 * Cin >= (int64_t)1 << 30) return p;  // 32-bit byte offsets
  const int BM = (p.G == 1 ? 4 : 8) * p.FM * 32;
  const int cap = (BM == 256) ? PATCH_ROWS_SMALL : PATCH_ROWS_BIG;
  if (!p3_best_tile(H, W, BM, cap, p.TH, p.TW, p.tiles_y, p.tiles_x)) return p;
  p.tiles_n = (Cout + (128 / p.G) - 1) / (128 / p.G);
  p.nblk = B * p.tiles_y * p.tiles_x;
  p.ok = 1;
  return p;
}

#ifdef M16_STAMP
extern "C" int sfod_debug_m16_stamps(unsigned long long* out8, int reset) {
  SFOD_REQUIRE_EXTENTS("debug_m16_stamps", reset);
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_m16_stamps), 64);
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_m16_stamps), z, 64); }
  return 0;
}
#endif

template <int NW, int NIP, int FMT, bool RED, bool XF = false, int NWN = 2>
static int p3_launch_m16(P3Args a, hipStream_t s) {
  auto kern = k_conv3x3_m16<NW, NIP, FMT, RED, XF, NWN>;
  constexpr int LDS = M16Lay<NW, NIP, NWN>::LDS;
  static const hipError_t attr_rc =
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr_rc != hipSuccess) { sfod_set_error("hipFuncSetAttribute(m16): %s", hipGetErrorString(attr_rc)); return -(int)attr_rc; }
  a.nbody = a.Cin / 64;
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(NW * 64), LDS, s, a);
  return sfod_check_launch("conv3x3_m16");
}

template <int G, int FM, typename OutT, int SPLIT = 0, bool RED = false>
static int p3_launch_one(const P3Args& a, hipStream_t s) {
  auto kern = k_conv3x3_patch<G, FM, OutT, SPLIT, RED>;
  constexpr int LDS = Lay<G, FM>::TOTAL;
  // once per kernel instantiation and process (function-local static: initialised exactly once, thread-safe)
  static const hipError_t attr_rc =
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  if (attr_rc != hipSuccess) { sfod_set_error("hipFuncSetAttribute(p3): %s", hipGetErrorString(attr_rc)); return -(int)attr_rc; }
  hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(512), LDS, s, a);
  return sfod_check_launch("conv3x3_patch");
}

bool sfod_p3_bnin_ok(const P3Plan& p, int Cin_phys, int split) {
  return p.ok && p.m16 != 0 && p.G == 1 && p.FM == 2 && Cin_phys % 64 == 0 && split == 1;
}

// split 1 / 2: SFOD_BF16X3 / SFOD_F16X3 operands; Cin is then the PHYSICAL 16-bit channel count (2 x logical), the output is fp32
int sfod_p3_launch(const P3Plan& p, const void* x, const void* w, const float* bias, void* y, float* stats,
                   int B, int H, int W, int Cin, int Cout, int ldy, int act, int out_f32, hipStream_t s, int split,
                   const P3BnRed* red, const unsigned* wamax, const P3BnIn* bnin) {
  P3Args a;
  a.wamax = wamax;
  a.xf_mean = a.xf_invstd = a.xf_gamma = a.xf_beta = nullptr;
  if (bnin != nullptr) {
    if (!sfod_p3_bnin_ok(p, Cin, split) || red != nullptr || !out_f32) {
      sfod_set_error("conv3x3_patch: the BatchNorm-input form needs bf16x3 operands, the 256 x 128 shape, Cin % 32 == 0");
      return SFOD_EBADARG;
    }
    a.xf_mean = bnin->mean; a.xf_invstd = bnin->invstd; a.xf_gamma = bnin->gamma; a.xf_beta = bnin->beta;
  }
  a.red_y = nullptr; a.red_mean = a.red_invstd = a.red_gamma = a.red_beta = nullptr; a.red_ws = nullptr;
  if (red != nullptr) {
    if (!out_f32 || ldy != Cout || Cout % 4 != 0) { sfod_set_error("conv3x3_patch: BatchNorm-backward epilogue needs a dense fp32 output"); return SFOD_EBADARG; }
    a.red_y = red->y; a.red_mean = red->mean; a.red_invstd = red->invstd; a.red_gamma = red->gamma; a.red_beta = red->beta;
    a.red_ws = red->ws;
  }
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = y; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldy = ldy; a.act = act;
  a.TH = p.TH; a.TW = p.TW; a.PW = p.TW + 2;
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.tiles_n = p.tiles_n;
  auto magic = [](int d) { return (unsigned)((((unsigned long long)1) << 32) / (unsigned)d + 1ull); };
  a.m_pw = magic(a.PW); a.m_tw = magic(a.TW); a.m_tn = magic(a.tiles_n); a.m_tx = magic(a.tiles_x);
  a.m_ty = magic(a.tiles_y);
  a.nbody = Cin / (32 * p.G);
  a.ntiles = B * p.tiles_y * p.tiles_x * p.tiles_n;
  a.nblk = p.nblk;
  if (split) {
    if (!out_f32) { sfod_set_error("conv3x3_patch: operand pairs write fp32"); return SFOD_EBADARG; }
    // the 256 x 128 shape on 16x16x32 MFMAs: same tile plan (statistics blocks, reduction rows), two slices per super-body
    if (bnin != nullptr) return p3_launch_m16<4, 8, 1, false, true>(a, s);
    if (p.m16 && p.G == 1 && p.FM == 2 && Cin % 64 == 0 && (red == nullptr || split == 1)) {
      if (p.m16 == 2) {                 // shape 6: 4 waves x (128 px x 64 ch), 2 x 208 of a SIMD lane's 512 registers
        if (red != nullptr) return p3_launch_m16<4, 8, 1, true>(a, s);
        return split == 2 ? p3_launch_m16<4, 8, 2, false>(a, s) : p3_launch_m16<4, 8, 1, false>(a, s);
      }
      if (red != nullptr) return p3_launch_m16<8, 4, 1, true>(a, s);
      return split == 2 ? p3_launch_m16<8, 4, 2, false>(a, s) : p3_launch_m16<8, 4, 1, false>(a, s);
    }
    // round 6: the 64-channel tile shapes on the same loop (shape 3: 256 px x 64 ch -> <4, 4, NWN = 1>; shape 4: 512 px x 64 ch
    // -> <8, 4, NWN = 1>).  Built, bit-checked (variants 7 / 8 in the kernel tests) and measured SLOWER than the 32x32x16 kernel's
    // 8-wave 256 x 64 shape on every layer that takes these tiles (conv1_2 1.35 vs 1.15 ms, dgrad2_1 0.62 vs 0.55, conv5_1 0.31
    // vs 0.28, RPN 0.14 vs 0.10: profiles/r6_m16_n64_variants.txt) -- half the resident waves (8 per CU) cost more on these
    // short-K, store-heavy shapes than the 4.5x fewer fragment reads return.  Forced variants 7 / 8 and SFOD_P3_M16_N64=1 only.
    if (p.m16 && p.G == 2 && Cin % 64 == 0 && (red == nullptr || split == 1) && (p.m16 >= 3 || p3_m16_n64_enabled())) {
      if (p.FM == 1 && (p.m16 == 4 || p3_m16_n64_enabled() == 2)) {      // variant 9: 8 waves x (32 px x 64 ch), 16 waves per CU
        if (red != nullptr) return p3_launch_m16<8, 2, 1, true, false, 1>(a, s);
        return split == 2 ? p3_launch_m16<8, 2, 2, false, false, 1>(a, s) : p3_launch_m16<8, 2, 1, false, false, 1>(a, s);
      }
      if (p.FM == 1) {
        if (red != nullptr) return p3_launch_m16<4, 4, 1, true, false, 1>(a, s);
        return split == 2 ? p3_launch_m16<4, 4, 2, false, false, 1>(a, s) : p3_launch_m16<4, 4, 1, false, false, 1>(a, s);
      }
      if (red != nullptr) return p3_launch_m16<8, 4, 1, true, false, 1>(a, s);
      return split == 2 ? p3_launch_m16<8, 4, 2, false, false, 1>(a, s) : p3_launch_m16<8, 4, 1, false, false, 1>(a, s);
    }
    if (red != nullptr) {
      if (split != 1) { sfod_set_error("conv3x3_patch: the BatchNorm-backward epilogue runs on bf16x3 operands"); return SFOD_EBADARG; }
      if (p.G == 1 && p.FM == 2) return p3_launch_one<1, 2, float, 1, true>(a, s);
      if (p.G == 1) return p3_launch_one<1, 4, float, 1, true>(a, s);
      if (p.FM == 2) return p3_launch_one<2, 2, float, 1, true>(a, s);
      return p3_launch_one<2, 1, float, 1, true>(a, s);
    }
    if (split == 2) {
      if (p.G == 1 && p.FM == 2) return p3_launch_one<1, 2, float, 2>(a, s);
      if (p.G == 1) return p3_launch_one<1, 4, float, 2>(a, s);
      if (p.FM == 2) return p3_launch_one<2, 2, float, 2>(a, s);
      return p3_launch_one<2, 1, float, 2>(a, s);
    }
    if (p.G == 1 && p.FM == 2) return p3_launch_one<1, 2, float, 1>(a, s);
    if (p.G == 1) return p3_launch_one<1, 4, float, 1>(a, s);
    if (p.FM == 2) return p3_launch_one<2, 2, float, 1>(a, s);
    return p3_launch_one<2, 1, float, 1>(a, s);
  }
  if (p.G == 1 && p.FM == 2) return out_f32 ? p3_launch_one<1, 2, float>(a, s) : p3_launch_one<1, 2, bf16_t>(a, s);
  if (p.G == 1) return out_f32 ? p3_launch_one<1, 4, float>(a, s) : p3_launch_one<1, 4, bf16_t>(a, s);
  if (p.FM == 2) return out_f32 ? p3_launch_one<2, 2, float>(a, s) : p3_launch_one<2, 2, bf16_t>(a, s);
  return out_f32 ? p3_launch_one<2, 1, float>(a, s) : p3_launch_one<2, 1, bf16_t>(a, s);
}
