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

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

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

}  // namespace

// Workgroup shape + tile shape.  Variants (all 8 waves):
//   1  G=1 FM=4  512 px x 128 ch, wave tile 128 x 64, one workgroup per CU (112 KiB of LDS, 212 VGPRs)
//   2  G=1 FM=2  256 px x 128 ch, wave tile  64 x 64, two workgroups per CU (72 KiB, 114 VGPRs)
//   3  G=2 FM=1  256 px x  64 ch, wave tile  32 x 64, two workgroups per CU (76 KiB,  72 VGPRs)
//   4  G=2 FM=2  512 px x  64 ch, wave tile  64 x 64, one workgroup per CU
// Measured per layer (tools/bench_conv.py, interleaved A/B; profiles/r1q_conv_variants.txt): two resident
// workgroups overlap each other's prologue / epilogue / barrier stalls, which beats the larger tiles' lower
// L2 -> LDS traffic on every VGG shape; between the two small shapes the 64 x 64 wave tile needs one LDS
// fragment read per MFMA instead of 1.5 (the 32 x 64 tile keeps the LDS array ~100 % busy at full MFMA rate)
// and wins by 7-18 % wherever its 128-channel tiles still fill the chip (>= 512 workgroups).
// SFOD_P3_VARIANT=1..4 / sfod_set_conv3x3_variant force a shape where the channel counts allow it (A/B, tests).
// process-wide tuning knob for A/B runs and tests (relaxed atomic: a plain word, no ordering needed); -1: not
// initialised (SFOD_P3_VARIANT or 0 = auto).  It selects among kernels that compute the same values.
static std::atomic<int> g_p3_variant{-1};

extern "C" int sfod_set_conv3x3_variant(int variant) {
  g_p3_variant.store((variant >= 1 && variant <= 4) ? variant : 0, std::memory_order_relaxed);
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

P3Plan sfod_p3_plan(int B, int H, int W, int Cin, int Cout) {
  P3Plan p;
  p.ok = 0;
  if (H < 1 || W < 1 || B < 1) return p;
  int variant = g_p3_variant.load(std::memory_order_relaxed);
  if (variant < 0) {
    const char* ev = getenv("SFOD_P3_VARIANT");
    variant = ev ? atoi(ev) : 0;
    int expect = -1;
    g_p3_variant.compare_exchange_strong(expect, variant, std::memory_order_relaxed);   // a concurrent setter wins
    variant = g_p3_variant.load(std::memory_order_relaxed);
  }
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

// split 1 / 2: SFOD_BF16X3 / SFOD_F16X3 operands; Cin is then the PHYSICAL 16-bit channel count (2 x logical), the output is fp32
int sfod_p3_launch(const P3Plan& p, const void* x, const void* w, const float* bias, void* y, float* stats,
                   int B, int H, int W, int Cin, int Cout, int ldy, int act, int out_f32, hipStream_t s, int split,
                   const P3BnRed* red, const unsigned* wamax) {
  P3Args a;
  a.wamax = wamax;
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
