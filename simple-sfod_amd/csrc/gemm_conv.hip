// MFMA implicit-GEMM kernels of the hot path (gfx950 / CDNA4, wave64):
//   k_conv_fwd   y[m][n]  = act(sum_k A[m][k] * Wt[n][k] + bias[n])   conv3x3 / conv1x1 / nn.Linear,
//                also the data-gradient (dgrad) with rotated weights      (K2, K5, K14, K18)
//   k_conv_wgrad dw[n][k'] += sum_m dy[m][n] * A[m][k']                 weight gradient
// A is never materialised: a 16-byte chunk of an im2col row is (pixel shifted by a tap, 4 fp32 /
// 8 bf16 consecutive NHWC channels) and is DMA'd straight into LDS by global_load_lds; padding
// taps and tail rows read a zero page instead.
//
// Tiling: 256 threads = 2x2 waves, each wave a (32*WM) x (32*WN) block of 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16 in throughput mode, v_mfma_f32_32x32x2_f32 in fp32 parity mode);
// every LDS row is 128 bytes (64 bf16 / 32 fp32 of K) and the 16-byte chunk index is XOR-swizzled
// with (row>>1)&7 so a ds_read_b128 of one K-chunk over 32 consecutive rows is conflict-free.
// The LDS image is lane-linear (what the DMA writes), so the swizzle is applied to the per-lane
// SOURCE address and again on the read.  Two LDS stages: stage t+1 is in flight while stage t
// feeds the MFMAs.
#include "conv_internal.h"
#include <algorithm>
#include <atomic>
// Knock-out switches for tools/experiments/gemm_knockout.sh (time floors of k_conv_fwd's pair modes; results wrong by
// construction, the product build defines none): GEMM_KO_MFMA no MFMAs, GEMM_KO_STORE no output stores (fp32 outputs),
// GEMM_KO_STATS no BatchNorm partial statistics.

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];

struct ConvArgs {
  int M;          // rows = B*H*W
  int H, W;       // spatial size (1x1 for linear layers with M = R rows)
  int Cin;        // input channels as stored (row stride of x), multiple of the chunk width
  int Cout;       // output channels
  int ks;         // 1 or 3
  int cpt_shift;  // log2(chunks per tap) when ks == 3
  int kchunks;    // total 16-byte chunks along K = taps * Cin * sizeof(T) / 16
  int ldy;        // row stride of y
  int act;        // 0 none, 1 relu, 2 leaky relu 0.2
  const unsigned* wamax;   // SFOD_F16X3 weights packed with a per-tensor power-of-two scale (common.h): max|w| bits, or nullptr
  int kt_per;     // split-K (blockIdx.y = split): K tiles per split, 0 = the whole K range in one workgroup
  int nsplit;     // split-K: grid.y (host side)
  int64_t slab;   // split-K: elements between the splits' output slabs
};

template <typename T> struct Chunk { static constexpr int E = 16 / sizeof(T); };

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(l), 16, 0, 0);
}

// tap of a flattened K chunk index and the channel-chunk inside it
__device__ __forceinline__ void chunk_to_tap(int gq, int ks, int cpt_shift, int& tap, int& cc) {
  if (ks == 1) { tap = 0; cc = gq; }
  else { tap = gq >> cpt_shift; cc = gq & ((1 << cpt_shift) - 1); }
}

// branch-free: act 0 identity, 1 ReLU, 2 LeakyReLU(0.2) == max(v, relu ? 0 : v * slope), slope 1 / - / 0.2
__device__ __forceinline__ float apply_act(float v, int act) {
  const float slope = (act == 2) ? 0.2f : 1.f;
  return fmaxf(v, (act == 1) ? 0.f : v * slope);
}

// =============================================================================================
// forward / dgrad / linear
// =============================================================================================
// UT ("uniform tap"): every 8-chunk K tile lies inside ONE filter tap (Cin*sizeof(T) a multiple of
// 128 bytes, or a 1x1 / linear layer), so tap decoding is scalar work per K tile and a lane's
// source address is row_base + tile_delta; the general path (first layer, Cin = 3) decodes per lane.
// WR = wave rows (2 -> 128-row tile / 256 threads, 4 -> 256-row tile / 512 threads);
// NST = LDS stages: 2 = load(t+1) || compute(t) with a full drain per K tile; 3 = two K tiles in
// flight across raw s_barriers with a counted s_waitcnt vmcnt (the DMA queue is never drained
// inside the loop).
// SPLIT (1: SFOD_BF16X3, 2: SFOD_F16X3): T = 16-bit words over 2 * Cin physical channels holding (8 hi | 8 lo) groups; a
// 128-byte LDS row is then 32 logical channels = two k-steps, each fed as hi*lo + lo*hi + hi*hi (see sfod_hip.h); the two
// pair formats differ in the MFMA opcode only.
// BKB = bytes per LDS row (K extent of a stage): 128 (8 chunks), or 64 (4 chunks; bf16x3: ONE k-step per stage) for the
// 256 x 256 tile (WN = 4: wave tile 64 x 128) whose three stages of 128-byte rows would not fit LDS.  The wide tile moves
// 1/3 fewer operand bytes L2 -> LDS per MFMA than 256 x 128 -- the bound of the long-K linear layers in bf16x3, whose
// operands are 4 bytes per element.
template <typename T, typename OutT, int WM, int WN, bool UT, int WR, int NST, int SPLIT = 0, int BKB = 128>
__global__ void __launch_bounds__(WR * 128)
k_conv_fwd(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
           OutT* __restrict__ y, float* __restrict__ stats, ConvArgs a, int tiles_n, int ntiles) {
  static_assert(WM == 2, "wave tile is 64 rows");
  constexpr int NT = WR * 128;
  constexpr int BM = WR * 32 * WM, BN = 64 * WN;
  constexpr int E = Chunk<T>::E;
  constexpr int CPR = BKB / 16;            // 16-byte chunks per LDS row
  constexpr int RPI = 1024 / BKB;          // rows per 1-KiB DMA instruction
  constexpr int STAGE = (BM + BN) * BKB;
  static_assert(BKB == 128 || BKB == 64, "LDS rows of 128 or 64 bytes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // chunk c of row r is stored at chunk c ^ swz(r): conflict-free ds_read_b128 lane groups for either row length
  auto swz = [](int r) { return BKB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

  // XCD-aware tile order: workgroups that share an XCD (same blockIdx % 8) walk consecutive tiles
  int bid = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int taps = a.ks * a.ks;
  const int HW = a.H * a.W;

  // ---- per-thread DMA descriptors ---------------------------------------------------------------
  constexpr int AI = BM / (RPI * 2 * WR), BI = BN / (RPI * 2 * WR);  // 1-KiB DMA instructions per wave and stage
  static_assert(BI >= 1, "tile too narrow for this wave count");
  int a_row[AI];      // row inside the tile
  int a_oy[AI], a_ox[AI];
  int64_t a_pix[AI];  // flat pixel index (== m) or -1
  for (int i = 0; i < AI; ++i) {
    const int q = wave * AI + i;
    const int row = q * RPI + lane / CPR;
    a_row[i] = row;
    const int m = m0 + row;
    if (m < a.M) {
      const int rem = m % HW;
      a_oy[i] = rem / a.W;
      a_ox[i] = rem % a.W;
      a_pix[i] = m;
    } else { a_pix[i] = -1; a_oy[i] = 0; a_ox[i] = 0; }
  }
  int b_row[BI];
  for (int i = 0; i < BI; ++i) b_row[i] = (wave * BI + i) * RPI + lane / CPR;
  const int pc = lane % CPR;
  // split-K: this workgroup walks K tiles kt0 .. kt0 + KT - 1 of the KT_all and writes its partial products into slab
  // blockIdx.y (no bias, no activation: the launcher's slab sum applies them); kt_per == 0: everything, as before
  const int KT_all = (a.kchunks + CPR - 1) / CPR;
  const int kt0 = a.kt_per ? (int)blockIdx.y * a.kt_per : 0;
  const int KT = a.kt_per ? min(a.kt_per, KT_all - kt0) : KT_all;
  if (a.kt_per) y += (int64_t)blockIdx.y * a.slab;
  const int64_t wrow_elems = (int64_t)a.kchunks * E;

  // hoisted per-row state of the uniform-tap path
  int64_t a_base[AI];   // element offset of (pixel, logical chunk) in x
  int a_lcs[AI];        // logical chunk of this lane
  unsigned a_mask[AI];  // bit t: tap t reads inside the image (bit 0 only for 1x1)
  int64_t b_base[BI];
  int b_lcs[BI];
  bool b_ok[BI];
  if (UT) {
    for (int i = 0; i < AI; ++i) {
      const int lc = pc ^ swz(a_row[i]);
      a_lcs[i] = lc;
      a_base[i] = (a_pix[i] >= 0 ? a_pix[i] : 0) * a.Cin + (int64_t)lc * E;
      unsigned m = 0;
      if (a_pix[i] >= 0) {
        if (a.ks == 1) m = 1u;
        else
          for (int t = 0; t < 9; ++t) {
            const int iy = a_oy[i] + t / 3 - 1, ix = a_ox[i] + t % 3 - 1;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) m |= (1u << t);
          }
      }
      a_mask[i] = m;
    }
    for (int i = 0; i < BI; ++i) {
      const int lc = pc ^ swz(b_row[i]);
      b_lcs[i] = lc;
      const int n = n0 + b_row[i];
      b_ok[i] = n < a.Cout;
      b_base[i] = (int64_t)(b_ok[i] ? n : 0) * wrow_elems + (int64_t)lc * E;
    }
  }

  auto stage_load = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + BM * BKB;
    kt += kt0;
    if (UT) {
      // scalar (wave-uniform) tap decode for the whole K tile
      const int gq0 = kt * CPR;
      int tap = 0, cc0 = gq0;
      if (a.ks != 1) { tap = gq0 >> a.cpt_shift; cc0 = gq0 & ((1 << a.cpt_shift) - 1); }
      const int ky = tap / 3, kx = tap - ky * 3;
      const int64_t delta = (a.ks == 1) ? (int64_t)cc0 * E
                                        : ((int64_t)(ky - 1) * a.W + (kx - 1)) * a.Cin + (int64_t)cc0 * E;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const bool ok = ((a_mask[i] >> tap) & 1u) && (gq0 + a_lcs[i] < a.kchunks);
        const void* src = ok ? (const void*)(x + a_base[i] + delta) : (const void*)g_zero_page;
        glds16(src, sA + (wave * AI + i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const bool ok = b_ok[i] && (gq0 + b_lcs[i] < a.kchunks);
        const void* src = ok ? (const void*)(w + b_base[i] + (int64_t)gq0 * E) : (const void*)g_zero_page;
        glds16(src, sB + (wave * BI + i) * 1024);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = a_row[i];
      const int lc = pc ^ swz(row);
      const int gq = kt * CPR + lc;
      const void* src = g_zero_page;
      if (a_pix[i] >= 0 && gq < a.kchunks) {
        int tap, cc;
        chunk_to_tap(gq, a.ks, a.cpt_shift, tap, cc);
        if (a.ks == 1) {
          src = x + a_pix[i] * a.Cin + (int64_t)cc * E;
        } else {
          const int ky = tap / 3, kx = tap - ky * 3;
          const int iy = a_oy[i] + ky - 1, ix = a_ox[i] + kx - 1;
          if (tap < taps && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
            src = x + (a_pix[i] + (int64_t)(ky - 1) * a.W + (kx - 1)) * a.Cin + (int64_t)cc * E;
        }
      }
      glds16(src, sA + (wave * AI + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = b_row[i];
      const int lc = pc ^ swz(row);
      const int gq = kt * CPR + lc;
      const int n = n0 + row;
      const void* src = g_zero_page;
      if (n < a.Cout && gq < a.kchunks) src = w + (int64_t)n * wrow_elems + (int64_t)gq * E;
      glds16(src, sB + (wave * BI + i) * 1024);
    }
  };

  // ---- fragment read offsets ------------------------------------------------------------------
  int offA[WM], swzA[WM], offB[WN], swzB[WN];
  for (int i = 0; i < WM; ++i) {
    const int r = wr * 32 * WM + i * 32 + (lane & 31);
    offA[i] = r * BKB;
    swzA[i] = swz(r);
  }
  for (int j = 0; j < WN; ++j) {
    const int r = wc * 32 * WN + j * 32 + (lane & 31);
    offB[j] = r * BKB;
    swzB[j] = swz(r);
  }
  const int h = lane >> 5;

  f32x16 acc[WM][WN];
  for (int i = 0; i < WM; ++i)
    for (int j = 0; j < WN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto stage_compute = [&](int buf) {
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sB = sA + BM * BKB;
    if constexpr (SPLIT) {
#pragma unroll
      for (int t = 0; t < BKB / 64; ++t) {
        bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8*>(sA + offA[i] + (((4 * t + 2 * h) ^ swzA[i]) << 4));
          al[i] = *reinterpret_cast<const bf16x8*>(sA + offA[i] + (((4 * t + 2 * h + 1) ^ swzA[i]) << 4));
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          bh[j] = *reinterpret_cast<const bf16x8*>(sB + offB[j] + (((4 * t + 2 * h) ^ swzB[j]) << 4));
          bl[j] = *reinterpret_cast<const bf16x8*>(sB + offB[j] + (((4 * t + 2 * h + 1) ^ swzB[j]) << 4));
        }
#ifdef GEMM_KO_MFMA
#pragma unroll
        for (int i = 0; i < WM; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
        acc[0][0][0] += 1.0f;
#else
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = mfma_pairs<SPLIT>(ah[i], bl[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = mfma_pairs<SPLIT>(al[i], bh[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = mfma_pairs<SPLIT>(ah[i], bh[j], acc[i][j]);
#endif
      }
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < BKB / 32; ++s) {
        bf16x8 af[WM], bfr[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
          af[i] = *reinterpret_cast<const bf16x8*>(sA + offA[i] + (((2 * s + h) ^ swzA[i]) << 4));
#pragma unroll
        for (int j = 0; j < WN; ++j)
          bfr[j] = *reinterpret_cast<const bf16x8*>(sB + offB[j] + (((2 * s + h) ^ swzB[j]) << 4));
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CPR; ++c) {
        f32x4 af[WM], bfr[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(sA + offA[i] + ((c ^ swzA[i]) << 4));
#pragma unroll
        for (int j = 0; j < WN; ++j)
          bfr[j] = *reinterpret_cast<const f32x4*>(sB + offB[j] + ((c ^ swzB[j]) << 4));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const float av = half ? (h ? af[i][3] : af[i][2]) : (h ? af[i][1] : af[i][0]);
              const float bv = half ? (h ? bfr[j][3] : bfr[j][2]) : (h ? bfr[j][1] : bfr[j][0]);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
            }
        }
      }
    }
  };

  // ---- main loop --------------------------------------------------------------------------------
  if constexpr (NST == 2) {
    stage_load(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < KT - 1; ++kt) {
      stage_load(kt + 1, cur ^ 1);
      stage_compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
    stage_compute(cur);
  } else {
    // NST >= 3 stages: K tiles t+1 .. t+NST-1 are in flight while tile t feeds the MFMAs.  Per wave and stage
    // exactly AI+BI DMA instructions are issued, so "all but the newest (NST-2) x (AI+BI)" == "tile t landed".
    static_assert(NST >= 3 && NST <= 4 && (NST - 2) * (AI + BI) <= 63, "counted vmcnt immediate");
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
      if (p < KT) stage_load(p, p);
    int cur = 0, nxt = NST - 1;
    for (int kt = 0; kt < KT; ++kt) {
      // lgkmcnt(0): this wave's fragment reads of tile kt-1 have returned before anyone's DMA may overwrite
      // that buffer (the compiler sinks the last reads + MFMAs below the barrier otherwise)
      if (NST == 4 && kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (AI + BI)) : "memory");
      else if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // tile kt landed for every wave; everyone is done with tile kt-1
      asm volatile("" ::: "memory");
      if (kt + NST - 1 < KT) stage_load(kt + NST - 1, nxt);
      stage_compute(cur);
      cur = (cur == NST - 1) ? 0 : cur + 1;
      nxt = (nxt == NST - 1) ? 0 : nxt + 1;
    }
  }

  if constexpr (SPLIT == 2) {     // undo the packed weights' power-of-two scale (exact)
    if (a.wamax != nullptr) {
      const float inv = winv_from_absmax(*a.wamax);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
    }
  }
  // ---- epilogue: bias + activation + store; optional per-block BN partial statistics ----------
  float bcol[WN];
  for (int j = 0; j < WN; ++j) {
    const int n = n0 + wc * 32 * WN + j * 32 + (lane & 31);
    bcol[j] = (bias != nullptr && n < a.Cout) ? bias[n] : 0.f;
  }
  constexpr bool STAGED = (sizeof(OutT) == 2);   // 2-byte outputs go through LDS for 16-byte stores
  constexpr int CPITCH = BN * 2 + 16;            // bytes per staged row (+16: spread the banks)
  constexpr int SRED_OFF = STAGED ? BM * CPITCH : 0;   // scratch for the BN partials, behind the staged tile (if any)
  if (STAGED) __syncthreads();                   // every wave is done with the operand stages
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int nl = wc * 32 * WN + j * 32 + (lane & 31);
      const int n = n0 + nl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wr * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int m = m0 + ml;
        const float v = acc[i][j][r] + bcol[j];
        acc[i][j][r] = v;
        if (STAGED) {
          *reinterpret_cast<OutT*>(smem + ml * CPITCH + nl * 2) = from_f32<OutT>(apply_act(v, a.act));
        } else {
#ifdef GEMM_KO_STORE
          if (v == 1.2345e-30f) y[(int64_t)m * a.ldy + n] = from_f32<OutT>(v);     // keeps v alive, stores nothing
#else
          if (m < a.M && n < a.Cout) y[(int64_t)m * a.ldy + n] = from_f32<OutT>(apply_act(v, a.act));
#endif
        }
      }
    }
  if (STAGED) {
    __syncthreads();
    constexpr int CH_PER_ROW = BN / 8;           // 16-byte chunks per staged row
    const bool vec_ok = (a.ldy % 8) == 0;
    for (int c = threadIdx.x; c < BM * CH_PER_ROW; c += NT) {
      const int row = c / CH_PER_ROW, ch = c % CH_PER_ROW;
      const int m = m0 + row, n = n0 + ch * 8;
      if (m >= a.M || n >= a.Cout) continue;
      const unsigned char* src = smem + row * CPITCH + ch * 16;
      OutT* dst = y + (int64_t)m * a.ldy + n;
      if (vec_ok && n + 8 <= a.Cout) {
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
      } else {
        for (int e = 0; e < 8 && n + e < a.Cout; ++e) dst[e] = reinterpret_cast<const OutT*>(src)[e];
      }
    }
  }

#ifdef GEMM_KO_STATS
  stats = nullptr;
#endif
  if (stats != nullptr) {
    // statistics are reported per 128-row half tile: wave rows (2*hf, 2*hf+1) form half hf
    __syncthreads();  // LDS reuse
    float* sred = reinterpret_cast<float*>(smem + SRED_OFF);  // [WR][BN]
    const int hf = wr >> 1;
    const int rows_valid = min(128, a.M - (m0 + hf * 128));
    // pass 1: column sums
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          s += (m < a.M) ? acc[i][j][r] : 0.f;
        }
      s += __shfl_xor(s, 32);
      if (h == 0) sred[wr * BN + wc * 32 * WN + j * 32 + (lane & 31)] = s;
    }
    __syncthreads();
    float csum[WN], cmean[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int nl = wc * 32 * WN + j * 32 + (lane & 31);
      csum[j] = sred[(2 * hf) * BN + nl] + sred[(2 * hf + 1) * BN + nl];
      cmean[j] = csum[j] / (float)max(rows_valid, 1);
    }
    __syncthreads();
    // pass 2: squared deviations from the half-tile mean
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wr * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float d = acc[i][j][r] - cmean[j];
          s += (m < a.M) ? d * d : 0.f;
        }
      s += __shfl_xor(s, 32);
      if (h == 0) sred[wr * BN + wc * 32 * WN + j * 32 + (lane & 31)] = s;
    }
    __syncthreads();
    if ((wr & 1) == 0 && h == 0 && rows_valid > 0) {
      const int sblk = tile_m * (BM / 128) + hf;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int nl = wc * 32 * WN + j * 32 + (lane & 31);
        const int n = n0 + nl;
        if (n < a.Cout) {
          stats[((int64_t)sblk * 2 + 0) * a.Cout + n] = csum[j];
          stats[((int64_t)sblk * 2 + 1) * a.Cout + n] = sred[(2 * hf) * BN + nl] + sred[(2 * hf + 1) * BN + nl];
        }
      }
      // row count of the block, behind the [nblk][2][Cout] sums
      if (tile_n == 0 && wc == 0 && lane == 0)
        stats[(int64_t)((a.M + 127) / 128) * 2 * a.Cout + sblk] = (float)rows_valid;
    }
  }
}

static int ilog2_exact(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

// process-wide kernel-selection knob for A/B runs and tests (0 auto, 1 generic implicit GEMM only, 2 halo-patch kernel
// whenever the shape allows); a relaxed atomic word -- every choice computes the same values
static std::atomic<int> g_conv_algo_v{0};
#define g_conv_algo (g_conv_algo_v.load(std::memory_order_relaxed))
extern "C" int sfod_set_conv_algo(int algo) { g_conv_algo_v.store(algo, std::memory_order_relaxed); return 0; }

// SFOD_BF16X3 tensors are bf16 tensors with twice the channels (8 hi | 8 lo groups) as far as DMA, LDS layout and
// tile plans are concerned: the kernels below are planned / launched on the PHYSICAL channel count.
// SFOD_F16X3: the same with half pairs -- only the MFMA opcode of the forward kernels differs (split_code).
static inline bool is_bf16_storage(int dt) { return dt == SFOD_BF16 || sfod_is_pairs(dt); }
static inline int phys_ch(int dt, int c) { return sfod_is_pairs(dt) ? 2 * c : c; }
static inline int split_code(int dt) { return dt == SFOD_BF16X3 ? 1 : (dt == SFOD_F16X3 ? 2 : 0); }

static bool use_patch_kernel(const P3Plan& p, int B, int H, int W, int ksize, int dt) {
  if (ksize != 3 || !is_bf16_storage(dt) || !p.ok || g_conv_algo == 1) return false;
  if (g_conv_algo == 2) return true;
  // with the 256-pixel shapes the halo-patch kernel also wins on small maps (batch 1: the 18x37 RPN head
  // 0.042 vs 0.071 ms, conv5 0.046 vs 0.071); only degenerate problems stay on the generic kernel
  return (int64_t)B * p.tiles_y * p.tiles_x * p.tiles_n >= 8;
}

// first VGG layer: 3 real channels in one 8-wide chunk, 64 outputs; bf16 in / bf16 out, or bf16x3 pairs in / fp32 out
static bool use_first_kernel(int B, int H, int W, int Cin, int Cout, int ksize, int dt, int ldy, int out_dt) {
  const bool types = (dt == SFOD_BF16 && out_dt == SFOD_BF16) || (sfod_is_pairs(dt) && out_dt == SFOD_F32);
  return ksize == 3 && types && Cin == 8 && Cout == 64 && ldy % 8 == 0 &&
         g_conv_algo != 1 && (int64_t)B * H * W >= 4096;
}

static bool use_wide_gemm(int M, int Cout, int ks);

// host arithmetic of the conv family runs on int pixel counts (M = B * H * W) and int K (Cin * ksize^2): both must fit
static inline bool conv_shape_fits(int B, int H, int W, int Cin, int Cout, int ksize) {
  return sfod_ints_ok({B, H, W, Cin, Cout, ksize}) && ksize <= 7 && sfod_prod_fits({B, H, W}) &&
         sfod_prod_fits({Cin, ksize, ksize}) && sfod_prod_fits({Cout, ksize, ksize}) &&
         sfod_prod_fits({B, H, W, Cin > Cout ? Cin : Cout}, 1LL << 40);
}

extern "C" int sfod_conv_fwd_algo(int B, int H, int W, int Cin, int Cout, int ksize, int dt) {
  if (!conv_shape_fits(B, H, W, Cin, Cout, ksize)) return 0;      // hostile extents: not served / nothing
  if (use_first_kernel(B, H, W, Cin, Cout, ksize, dt, 64, sfod_is_pairs(dt) ? SFOD_F32 : SFOD_BF16)) return 3;
  const P3Plan p = (ksize == 3 && is_bf16_storage(dt)) ? sfod_p3_plan(B, H, W, phys_ch(dt, Cin), Cout, sfod_is_pairs(dt)) : P3Plan{};
  if (use_patch_kernel(p, B, H, W, ksize, dt)) return 2;
  return (sfod_is_pairs(dt) && use_wide_gemm(B * H * W, Cout, ksize)) ? 4 : 1;     // (fp32 output assumed: bf16x3 has no other)
}

extern "C" int sfod_conv_stats_blocks(int B, int H, int W, int Cin, int Cout, int ksize, int dt) {
  if (!conv_shape_fits(B, H, W, Cin, Cout, ksize)) return 0;      // hostile extents: not served / nothing
  if (use_first_kernel(B, H, W, Cin, Cout, ksize, dt, 64, sfod_is_pairs(dt) ? SFOD_F32 : SFOD_BF16)) return sfod_f1_nblk(B, H, W);
  const P3Plan p = (ksize == 3 && is_bf16_storage(dt)) ? sfod_p3_plan(B, H, W, phys_ch(dt, Cin), Cout, sfod_is_pairs(dt)) : P3Plan{};
  if (use_patch_kernel(p, B, H, W, ksize, dt)) return p.nblk;
  return (B * H * W + 127) / 128;
}

template <typename T, typename OutT, int WN, bool UT, int WR, int NST, int SPLIT = 0, int BKB = 128>
static int launch_one(const void* x, const void* w, const float* bias, void* y, float* stats,
                      const ConvArgs& a, hipStream_t s) {
  constexpr int BM = WR * 64, BN = 64 * WN;
  constexpr int OPER = NST * (BM + BN) * BKB;
  // staged C tile (2-byte outputs only) + BN partial scratch
  constexpr int EPI = (sizeof(OutT) == 2 ? BM * (BN * 2 + 16) : 0) + WR * BN * 4;
  constexpr int LDS = OPER > EPI ? OPER : EPI;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  auto kern = k_conv_fwd<T, OutT, 2, WN, UT, WR, NST, SPLIT, BKB>;
  // once per kernel instantiation and process (function-local static: initialised exactly once, thread-safe)
  static const hipError_t attr_rc = (LDS > 64 * 1024)
      ? hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)
      : hipSuccess;
  if (attr_rc != hipSuccess) { sfod_set_error("hipFuncSetAttribute: %s", hipGetErrorString(attr_rc)); return -(int)attr_rc; }
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.Cout + BN - 1) / BN;
  const int nt = tiles_m * tiles_n;
  hipLaunchKernelGGL(kern, dim3(nt, a.kt_per ? a.nsplit : 1), dim3(WR * 128), LDS, s, (const T*)x, (const T*)w, bias, (OutT*)y,
                     stats, a, tiles_n, nt);
  return sfod_check_launch("conv_fwd");
}

// ---- split-K for linear layers with few rows ----------------------------------------------------------------------------
// One frame per GPU (the yaml's literal batch): fc1 of the student is 512 x 25088 -> 1024 = 32 workgroups of 256 x 64 on 256
// CUs, each walking 784 K tiles (0.6 ms for 26 GFLOP).  With the K range cut in S pieces the grid is S x 32, every piece
// writes its fp32 partial tile into slab s of the caller's scratch, and k_splitk_sum adds the slabs in index order (fixed:
// run-to-run identical) together with bias and activation.  Chosen only when the plain grid leaves half the chip idle and
// every piece still has >= 32 of at least 256 K tiles; the scratch is S x M x Cout floats.
static int splitk_plan(int M, int kchunks, int Cout, int ks, bool fp32_out, bool stats) {
  static const int on = []() { const char* e = getenv("SFOD_GEMM_SPLITK"); return e ? atoi(e) : 1; }();
  if (!on || ks != 1 || !fp32_out || stats || M < 1) return 0;
  const int64_t wgs = (int64_t)((M + 255) / 256) * ((Cout + 63) / 64);
  const int KT_all = (kchunks + 7) / 8;
  // long K only (fc1-class layers: >= 256 K tiles = 8192 logical channels in the pair modes): a short-K layer with few rows
  // is a 10 us launch either way, and every split is one more fp32 summation order in the model
  if (wgs > 128 || KT_all < 256) return 0;
  int S = (int)(256 / wgs);
  if (S > 8) S = 8;
  if (S > KT_all / 32) S = KT_all / 32;
  return S >= 2 ? S : 0;
}

__global__ void __launch_bounds__(256) k_splitk_sum(const float* __restrict__ ws, int S, int64_t slab, int M, int Cout,
                                                    const float* __restrict__ bias, int act, float* __restrict__ y, int ldy) {
  const int cv = Cout / 4;
  const int64_t total = (int64_t)M * cv;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(t / cv), n = (int)(t % cv) * 4;
    float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)m * Cout + n);
    for (int sidx = 1; sidx < S; ++sidx) {
      const float4 u = *reinterpret_cast<const float4*>(ws + sidx * slab + (int64_t)m * Cout + n);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (bias != nullptr) { v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3]; }
    v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act);
    float* dst = y + (int64_t)m * ldy + n;
    if (ldy % 4 == 0) *reinterpret_cast<float4*>(dst) = v;
    else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
  }
}

template <typename T, int SPLIT>
static int launch_splitk(const void* x, const void* w, const float* bias, float* y, ConvArgs a, int S, float* ws,
                         hipStream_t s) {
  const int KT_all = (a.kchunks + 7) / 8;
  a.kt_per = (KT_all + S - 1) / S;
  a.nsplit = (KT_all + a.kt_per - 1) / a.kt_per;       // no empty split
  a.slab = (int64_t)a.M * a.Cout;
  const int act = a.act, ldy = a.ldy;
  a.act = 0; a.ldy = a.Cout;
  const int rc = launch_one<T, float, 1, true, 4, 3, SPLIT>(x, w, nullptr, ws, nullptr, a, s);
  if (rc != 0) return rc;
  const int64_t total = (int64_t)a.M * (a.Cout / 4);
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(k_splitk_sum, dim3(grid), dim3(256), 0, s, ws, a.nsplit, a.slab, a.M, a.Cout, bias, act, y, ldy);
  return sfod_check_launch("splitk_sum");
}

// bf16x3, ksize 1, fp32 out: does the 256 x 256 tile serve this shape?  (SFOD_GEMM_WIDE = 0 never / 2 always: A/B runs)
static bool use_wide_gemm(int M, int Cout, int ks) {
  static const int wide = []() { const char* e = getenv("SFOD_GEMM_WIDE"); return e ? atoi(e) : 1; }();
  if (wide <= 0 || ks != 1 || Cout < 256) return false;
  const int64_t t256 = (int64_t)((M + 255) / 256) * ((Cout + 255) / 256);
  const int64_t t128 = (int64_t)((M + 255) / 256) * ((Cout + 127) / 128);
  const double r256 = (double)((t256 + 255) / 256), r128 = (double)((t128 + 255) / 256) * 0.5;   // rounds, in 256-wide tile times
  return wide == 2 || (t256 >= 200 && r256 * 0.80 <= r128);
}

// forced tile shape of the generic kernel (A/B runs, tests): -1 not initialised (SFOD_GEMM_TILE or 0 = the planner's choice)
static std::atomic<int> g_gemm_tile{-1};
extern "C" int sfod_set_gemm_tile(int tile) {
  g_gemm_tile.store((tile >= 0 && tile <= 8) ? tile : 0, std::memory_order_relaxed);
  return 0;
}
static int gemm_tile_forced() {
  int v = g_gemm_tile.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("SFOD_GEMM_TILE");
    int want = e ? atoi(e) : 0, expect = -1;
    if (want < 0 || want > 8) want = 0;
    g_gemm_tile.compare_exchange_strong(expect, want, std::memory_order_relaxed);
    v = g_gemm_tile.load(std::memory_order_relaxed);
  }
  return v;
}

template <typename T, typename OutT, bool UT, int SPLIT = 0>
static int launch_conv_fwd_ut(const void* x, const void* w, const float* bias, void* y, float* stats,
                              const ConvArgs& a, hipStream_t s) {
  // SFOD_GEMM_TILE = 1..5: force a tile shape where it exists (A/B runs: 1 128x64, 2 128x128, 3 256x128, 4 256x64, 5 256x256)
  const int forced = gemm_tile_forced();
  if constexpr (UT) {
    if (forced == 1) return launch_one<T, OutT, 1, UT, 2, 2, SPLIT>(x, w, bias, y, stats, a, s);
    if (forced == 2) return launch_one<T, OutT, 2, UT, 2, 2, SPLIT>(x, w, bias, y, stats, a, s);
    if (forced == 3) return launch_one<T, OutT, 2, UT, 4, 3, SPLIT>(x, w, bias, y, stats, a, s);
    if (forced == 4) return launch_one<T, OutT, 1, UT, 4, 3, SPLIT>(x, w, bias, y, stats, a, s);
    if constexpr (SPLIT && sizeof(OutT) == 4) {
      if (forced == 5) return launch_one<T, OutT, 4, UT, 4, 3, SPLIT, 64>(x, w, bias, y, stats, a, s);
      // round 6: 64-byte K stages so that TWO workgroups share a CU (their prologues / epilogues / barrier stalls overlap):
      // 6: 256 x 128, 3 stages (72 KiB); 7: 128 x 128, 3 stages (48 KiB: three per CU); 8: 256 x 128, 4 stages (96 KiB)
      if (forced == 6) return launch_one<T, OutT, 2, UT, 4, 3, SPLIT, 64>(x, w, bias, y, stats, a, s);
      if (forced == 7) return launch_one<T, OutT, 2, UT, 2, 3, SPLIT, 64>(x, w, bias, y, stats, a, s);
      if (forced == 8) return launch_one<T, OutT, 2, UT, 4, 4, SPLIT, 64>(x, w, bias, y, stats, a, s);
      // (9 / 10, round 6: a PERSISTENT form of this tile -- one workgroup walking several tiles, its DMA pipeline running across
      // tile boundaries, statistics scratch behind the stages, epilogue barriers that leave vmcnt alone -- was built, was
      // bit-identical to 3 / 6 on 20 cases, and ran 15 % / 35 % SLOWER on every ResNet-101-C4 1x1 shape: the cold pipeline
      // per tile is not what these launches wait for.  Removed; profiles/r6_r101_gemm_floor.txt has the table.)
    }
  }
  if (a.Cout <= 64) return launch_one<T, OutT, 1, UT, 2, 2, SPLIT>(x, w, bias, y, stats, a, s);
  // long-K linear layers with few rows (the student's fc1: 4096 x 25088 -> 1024): 256 x 64 tiles, 8 waves, 3-stage
  // pipeline -- one 8-wave workgroup per CU instead of one 4-wave one (0.35 -> 0.30 ms; SFOD_GEMM_TALL=0 disables)
  static const int tall = []() { const char* e = getenv("SFOD_GEMM_TALL"); return e ? atoi(e) : 300; }();
  if constexpr (UT) {
    if (tall > 0 && a.ks == 1 && a.Cin >= 4096 && (int64_t)((a.M + 255) / 256) * ((a.Cout + 63) / 64) <= tall)
      return launch_one<T, OutT, 1, UT, 4, 3, SPLIT>(x, w, bias, y, stats, a, s);
  }
  // bf16x3 linear layers / 1x1 convolutions with wide outputs: 256 x 256 tiles (64-byte K stages) when their rounds of
  // 256 workgroups (one per CU) are filled well enough to beat 256 x 128 (profiles/r2d_bench_gemm.txt)
  if constexpr (UT && SPLIT && sizeof(OutT) == 4) {
    // (a 4-stage form of this pipeline -- three K tiles in flight -- measured the same: profiles/r3_rejected_experiments.txt)
    if (use_wide_gemm(a.M, a.Cout, a.ks)) return launch_one<T, OutT, 4, UT, 4, 3, SPLIT, 64>(x, w, bias, y, stats, a, s);
  }
  // fp32 (MFMA-bound at 1/16 of the bf16 rate: a tile's time is its FLOPs, the operand stream is never the limit): pick
  // the tile shape by whole-chip rounds.  The ResNet-C4 layers at 600x1200 have few tiles per CU (res4: 22 800 rows x
  // 256 channels = 358 tiles of 128 x 128 on 256 CUs: two rounds for 1.4 rounds of work); 128 x 64 tiles pack the same
  // layer into 716 / 768 slots (three per CU) -- cost = ceil(workgroups / 256) x tile area, smaller is better, ties to
  // the larger tile.  SFOD_GEMM_FP32_TILES=0 restores the fixed rule (A/B).
  if constexpr (sizeof(T) == 4 && UT) {
    static const int cost_rule = []() { const char* e = getenv("SFOD_GEMM_FP32_TILES"); return e ? atoi(e) : 1; }();
    if (cost_rule) {
      auto rounds = [&](int bm, int bn) {
        const int64_t wgs = (int64_t)((a.M + bm - 1) / bm) * ((a.Cout + bn - 1) / bn);
        return (double)((wgs + 255) / 256) * bm * bn;
      };
      const double c256 = rounds(256, 128) * 0.98, c128 = rounds(128, 128), c64 = rounds(128, 64) * 1.03;
      if (c64 < c128 && c64 < c256) return launch_one<T, OutT, 1, UT, 2, 2, SPLIT>(x, w, bias, y, stats, a, s);
      if (c256 <= c128) return launch_one<T, OutT, 2, UT, 4, 3, SPLIT>(x, w, bias, y, stats, a, s);
      return launch_one<T, OutT, 2, UT, 2, 2, SPLIT>(x, w, bias, y, stats, a, s);
    }
  }
  // 256 x 128 tiles with a 3-stage DMA pipeline once the grid still fills the chip (>= 2 tiles / CU)
  const int64_t big_tiles = (int64_t)((a.M + 255) / 256) * ((a.Cout + 127) / 128);
  if (UT && big_tiles >= 384) return launch_one<T, OutT, 2, UT, 4, 3, SPLIT>(x, w, bias, y, stats, a, s);
  return launch_one<T, OutT, 2, UT, 2, 2, SPLIT>(x, w, bias, y, stats, a, s);
}

template <typename T, typename OutT, int SPLIT = 0>
static int launch_conv_fwd(const void* x, const void* w, const float* bias, void* y, float* stats,
                           const ConvArgs& a, hipStream_t s) {
  const int cpt = a.Cin / Chunk<T>::E;
  const bool ut = (a.ks == 1) || (cpt % 8 == 0);
  if (ut) return launch_conv_fwd_ut<T, OutT, true, SPLIT>(x, w, bias, y, stats, a, s);
  return launch_conv_fwd_ut<T, OutT, false, SPLIT>(x, w, bias, y, stats, a, s);
}

extern "C" int sfod_conv_first_supported(int B, int H, int W, int Cin, int Cout, int dt, int ldy) {
  if (!(conv_shape_fits(B, H, W, Cin, Cout, 3) && sfod_ints_ok({ldy}))) return 0;      // hostile extents: not served / nothing
  return use_first_kernel(B, H, W, Cin, Cout, 3, dt, ldy, sfod_is_pairs(dt) ? SFOD_F32 : SFOD_BF16) ? 1 : 0;
}

extern "C" int sfod_conv_first_fused(const void* x, const void* w, const float* bias, const float* scale,
                                     const float* shift, void* y, float* stats, int B, int H, int W, int ldy,
                                     int act, int dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("conv_first_fused", B, H, W, ldy);
  return sfod_conv_first_fused_ws(x, w, nullptr, bias, scale, shift, y, stats, B, H, W, ldy, act, dt, stream);
}

extern "C" int sfod_conv_first_fused_ws(const void* x, const void* w, const uint32_t* w_absmax, const float* bias,
                                        const float* scale, const float* shift, void* y, float* stats, int B, int H,
                                        int W, int ldy, int act, int dt, void* stream) {
  SFOD_REQUIRE(conv_shape_fits(B, H, W, 8, 64, 3) && sfod_ints_ok({ldy}), "conv_first_fused: negative or oversized extent");
  SFOD_REQUIRE(x != nullptr && w != nullptr, "conv_first_fused: null operand");
  SFOD_REQUIRE(w_absmax == nullptr || dt == SFOD_F16X3, "conv_first: scaled weights are an SFOD_F16X3 format");
  SFOD_REQUIRE(sfod_conv_first_supported(B, H, W, 8, 64, dt, ldy),
               "conv_first_fused: shape not served by the first-layer kernel (sfod_conv_first_supported)");
  SFOD_REQUIRE(y != nullptr || stats != nullptr, "conv_first_fused: nothing to produce");
  SFOD_REQUIRE((scale == nullptr) == (shift == nullptr), "conv_first_fused: scale and shift come together");
  return sfod_f1_launch(x, w, bias, y, stats, B, H, W, ldy, act, (hipStream_t)stream, scale, shift, split_code(dt), w_absmax);
}

extern "C" int sfod_conv_fwd_scratch(const void* x, const void* w, const uint32_t* w_absmax, const float* bias, void* y,
                                     int B, int H, int W, int Cin, int Cout, int ksize, int ldy, int act, float* stats,
                                     int dt, int out_dt, void* scratch, int64_t scratch_bytes, void* stream);

extern "C" int sfod_conv_fwd(const void* x, const void* w, const float* bias, void* y, int B, int H, int W,
                             int Cin, int Cout, int ksize, int ldy, int act, float* stats, int dt,
                             int out_dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("conv_fwd", B, H, W, Cin, Cout, ksize, ldy);
  return sfod_conv_fwd_ws(x, w, nullptr, bias, y, B, H, W, Cin, Cout, ksize, ldy, act, stats, dt, out_dt, stream);
}

extern "C" int sfod_conv_fwd_ws(const void* x, const void* w, const uint32_t* w_absmax, const float* bias, void* y, int B,
                                int H, int W, int Cin, int Cout, int ksize, int ldy, int act, float* stats, int dt,
                                int out_dt, void* stream) {
  SFOD_REQUIRE_EXTENTS("conv_fwd_ws", B, H, W, Cin, Cout, ksize, ldy);
  return sfod_conv_fwd_scratch(x, w, w_absmax, bias, y, B, H, W, Cin, Cout, ksize, ldy, act, stats, dt, out_dt, nullptr, 0,
                               stream);
}

// Scratch the caller may offer to sfod_conv_fwd_scratch for this shape (0: none wanted): today the slabs of the split-K form
// of linear layers with few rows (splitk_plan above).
extern "C" int64_t sfod_conv_fwd_scratch_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int dt, int out_dt,
                                               int with_stats) {
  if (!conv_shape_fits(B, H, W, Cin, Cout, ksize)) return 0;      // hostile extents: not served / nothing
  if (dt != SFOD_F32 && dt != SFOD_BF16 && !sfod_is_pairs(dt)) return 0;
  const int E = (dt == SFOD_F32) ? 4 : 8;
  if (ksize != 1 || Cout % 4 != 0 || Cin % E != 0 || (int64_t)B * H * W == 0 || (int64_t)B * H * W > (1 << 24)) return 0;
  const int S = splitk_plan(B * H * W, phys_ch(dt, Cin) / E, Cout, ksize, out_dt == SFOD_F32, with_stats != 0);
  return (int64_t)S * B * H * W * Cout * 4;
}

extern "C" int sfod_conv_fwd_scratch(const void* x, const void* w, const uint32_t* w_absmax, const float* bias, void* y,
                                     int B, int H, int W, int Cin, int Cout, int ksize, int ldy, int act, float* stats,
                                     int dt, int out_dt, void* scratch, int64_t scratch_bytes, void* stream) {
  SFOD_REQUIRE(conv_shape_fits(B, H, W, Cin, Cout, ksize) && sfod_ints_ok({ldy}) && sfod_i64s_ok({scratch_bytes}),
               "conv_fwd: negative or oversized extent");
  SFOD_REQUIRE(x != nullptr && w != nullptr && y != nullptr, "conv_fwd: null operand (x, w, y)");
  SFOD_REQUIRE(act >= 0 && act <= 2, "conv_fwd: unknown act");
  SFOD_REQUIRE(out_dt == SFOD_F32 || out_dt == SFOD_BF16 || sfod_is_pairs(out_dt), "conv_fwd: unknown out_dt");
  SFOD_REQUIRE(w_absmax == nullptr || dt == SFOD_F16X3, "conv: scaled weights are an SFOD_F16X3 format");
  SFOD_REQUIRE(ksize == 1 || ksize == 3, "conv: ksize must be 1 or 3");
  SFOD_REQUIRE(ldy >= Cout, "conv: ldy < Cout");
  const int E = (dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(dt == SFOD_F32 || dt == SFOD_BF16 || sfod_is_pairs(dt), "conv: unknown dt");
  SFOD_REQUIRE(Cin % E == 0, "conv: Cin must be a multiple of the 16-byte chunk (operand pairs: of 8)");
  SFOD_REQUIRE(!sfod_is_pairs(dt) || out_dt == SFOD_F32, "conv: operand pairs write fp32");
  if ((int64_t)B * H * W == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (use_first_kernel(B, H, W, Cin, Cout, ksize, dt, ldy, out_dt))
    return sfod_f1_launch(x, w, bias, y, stats, B, H, W, ldy, act, s, nullptr, nullptr, split_code(dt), w_absmax);
  const int split = split_code(dt);
  Cin = phys_ch(dt, Cin);           // from here on: bf16 channels as stored
  if (ksize == 3 && is_bf16_storage(dt)) {
    const P3Plan p = sfod_p3_plan(B, H, W, Cin, Cout, split != 0);
    if (use_patch_kernel(p, B, H, W, ksize, dt))
      return sfod_p3_launch(p, x, w, bias, y, stats, B, H, W, Cin, Cout, ldy, act, out_dt == SFOD_F32, s, split, nullptr,
                            w_absmax);
  }
  ConvArgs a;
  a.wamax = w_absmax;
  a.M = B * H * W; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.ldy = ldy; a.act = act;
  const int cpt = Cin / E;
  a.cpt_shift = 0;
  if (ksize == 3) {
    a.cpt_shift = ilog2_exact(cpt);
    SFOD_REQUIRE(a.cpt_shift >= 0, "conv3x3: Cin/chunk must be a power of two");
  }
  a.kchunks = ksize * ksize * cpt;
  a.kt_per = 0; a.nsplit = 1; a.slab = 0;
  if (scratch != nullptr && Cout % 4 == 0 && (int64_t)a.M <= (1 << 24)) {
    const int S = splitk_plan(a.M, a.kchunks, Cout, ksize, out_dt == SFOD_F32, stats != nullptr);
    if (S >= 2 && scratch_bytes >= (int64_t)S * a.M * Cout * 4) {
      float* ws = (float*)scratch;
      if (dt == SFOD_F32) return launch_splitk<float, 0>(x, w, bias, (float*)y, a, S, ws, s);
      if (split == 2) return launch_splitk<bf16_t, 2>(x, w, bias, (float*)y, a, S, ws, s);
      if (split) return launch_splitk<bf16_t, 1>(x, w, bias, (float*)y, a, S, ws, s);
      return launch_splitk<bf16_t, 0>(x, w, bias, (float*)y, a, S, ws, s);
    }
  }
  if (dt == SFOD_F32) {
    SFOD_REQUIRE(out_dt == SFOD_F32, "conv: fp32 compute writes fp32");
    return launch_conv_fwd<float, float>(x, w, bias, y, stats, a, s);
  }
  if (split == 2) return launch_conv_fwd<bf16_t, float, 2>(x, w, bias, y, stats, a, s);
  if (split) return launch_conv_fwd<bf16_t, float, 1>(x, w, bias, y, stats, a, s);
  if (out_dt == SFOD_F32) return launch_conv_fwd<bf16_t, float>(x, w, bias, y, stats, a, s);
  return launch_conv_fwd<bf16_t, bf16_t>(x, w, bias, y, stats, a, s);
}

// 3x3 convolution whose INPUT is the previous layer's pre-BatchNorm output: relu(bn(x)) and the operand-pair split happen in
// the kernel's LDS patch instead of in a separate elementwise pass (forward-only passes: nobody else needs the activated
// tensor).  _supported: 1 when the shape is served (SFOD_BF16X3 operands, the 256 x 128 halo-patch shape, Cin % 32 == 0).
extern "C" int sfod_conv_fwd_bnin_supported(int B, int H, int W, int Cin, int Cout, int dt) {
  if (!conv_shape_fits(B, H, W, Cin, Cout, 3)) return 0;      // hostile extents: not served / nothing
  if (dt != SFOD_BF16X3 || (int64_t)B * H * W == 0 || g_conv_algo == 1) return 0;
  const int pc = phys_ch(dt, Cin);
  const P3Plan p = sfod_p3_plan(B, H, W, pc, Cout, 1);
  return (use_patch_kernel(p, B, H, W, 3, dt) && sfod_p3_bnin_ok(p, pc, 1)) ? 1 : 0;
}

extern "C" int sfod_conv_fwd_bnin(const float* x_pre, const float* in_mean, const float* in_invstd, const float* in_gamma,
                                  const float* in_beta, const void* w, const float* bias, float* y, int B, int H, int W,
                                  int Cin, int Cout, int ldy, int act, float* stats, int dt, void* stream) {
  SFOD_REQUIRE(conv_shape_fits(B, H, W, Cin, Cout, 3) && sfod_ints_ok({ldy}), "conv_fwd_bnin: negative or oversized extent");
  SFOD_REQUIRE(sfod_conv_fwd_bnin_supported(B, H, W, Cin, Cout, dt), "conv_fwd_bnin: shape not served (sfod_conv_fwd_bnin_supported)");
  SFOD_REQUIRE(x_pre && in_mean && in_invstd && in_gamma && in_beta && w && y, "conv_fwd_bnin: null argument");
  SFOD_REQUIRE(ldy >= Cout, "conv_fwd_bnin: ldy < Cout");
  const int pc = phys_ch(dt, Cin);
  const P3Plan p = sfod_p3_plan(B, H, W, pc, Cout, 1);
  const P3BnIn bnin{in_mean, in_invstd, in_gamma, in_beta};
  return sfod_p3_launch(p, x_pre, w, bias, y, stats, B, H, W, pc, Cout, ldy, act, 1, (hipStream_t)stream, 1, nullptr, nullptr,
                        &bnin);
}

// Data gradient of a 3x3 convolution (x = dy of the layer above as operand, w = its rotated weights) whose epilogue also
// makes the BatchNorm-backward partial sums of the layer BELOW, i.e. of the tensor this launch writes (dz): halo-patch
// kernel with fp32 output only.  sfod_conv_dgrad_bnred_blocks: number of partial rows it writes (0: shape / dtype not
// served -- run sfod_conv_fwd and the separate reduction instead).
extern "C" int sfod_conv_dgrad_bnred_blocks(int B, int H, int W, int Cin, int Cout, int dt) {
  if (!conv_shape_fits(B, H, W, Cin, Cout, 3)) return 0;      // hostile extents: not served / nothing
  if (dt != SFOD_BF16X3 && dt != SFOD_BF16) return 0;
  if (dt == SFOD_BF16) return 0;        // bf16 mode writes bf16 gradients: no fp32 tile to reduce
  if (Cout % 4 != 0) return 0;
  const P3Plan p = sfod_p3_plan(B, H, W, phys_ch(dt, Cin), Cout, sfod_is_pairs(dt));
  return use_patch_kernel(p, B, H, W, 3, dt) ? p.nblk : 0;
}

extern "C" int sfod_conv_dgrad_bnred(const void* x, const void* w, void* dz, int B, int H, int W, int Cin, int Cout,
                                     int dt, const float* y, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, float* red_ws, void* stream) {
  SFOD_REQUIRE(conv_shape_fits(B, H, W, Cin, Cout, 3), "conv_dgrad_bnred: negative or oversized extent");
  SFOD_REQUIRE(x != nullptr && w != nullptr && dz != nullptr, "conv_dgrad_bnred: null operand");
  SFOD_REQUIRE(sfod_conv_dgrad_bnred_blocks(B, H, W, Cin, Cout, dt) > 0,
               "conv_dgrad_bnred: shape not served (sfod_conv_dgrad_bnred_blocks)");
  SFOD_REQUIRE(y && mean && invstd && gamma && beta && red_ws, "conv_dgrad_bnred: null argument");
  const int pc = phys_ch(dt, Cin);
  const P3Plan p = sfod_p3_plan(B, H, W, pc, Cout, sfod_is_pairs(dt));
  const P3BnRed red{y, mean, invstd, gamma, beta, red_ws};
  return sfod_p3_launch(p, x, w, nullptr, dz, nullptr, B, H, W, pc, Cout, Cout, 0, 1, (hipStream_t)stream,
                        dt == SFOD_BF16X3, &red);
}

// =============================================================================================
// weight gradient.  GEMM over the pixel axis: dw[co][nf] += sum_pix dy[pix][co] * xs[pix][nf],
// nf = flattened (tap, ci).  Both operands arrive pixel-major ([pix][channels]), i.e. transposed
// for MFMA; in bf16 the fragments are fetched with ds_read_b64_tr_b16 (hardware transpose read),
// in fp32 a fragment element is a plain 4-byte read.  The pixel range is split across
// workgroups (gridDim.z) and the fp32 partial tiles are combined with float atomics
// (128-byte contiguous segments per wave-instruction).
// =============================================================================================
struct WgradArgs {
  int M, H, W;
  int Cin, Cout, ks, cpt_shift;
  int nchunks;   // taps * Cin / E : chunks along the flattened (tap, ci) axis
  int lddy;
  int Ntot;      // taps * Cin (row stride of dw)
  int pix_per_split;
  int tiles_n, tiles_m;
  // deterministic mode (sfod_set_deterministic): > 0 = every pixel split STORES its partial tile into its own slab
  // dw + split * slab_stride (workspace), summed afterwards in split order by k_wgrad_slab_sum; 0 = float atomics into dw
  int64_t slab_stride;
};
// one element of a pixel split's partial weight-gradient tile
__device__ __forceinline__ void wg_emit(float* dw, const WgradArgs& a, int split, int co, int n, float v) {
  if (a.slab_stride > 0) dw[(int64_t)split * a.slab_stride + (int64_t)co * a.Ntot + n] = v;
  else atomicAdd(dw + (int64_t)co * a.Ntot + n, v);
}

// SPLIT (SFOD_BF16X3 operands; T = bf16, WgradArgs in LOGICAL channels): the workgroup tile is 64 output channels x 64
// flattened (tap, ci) columns.  The DMA de-interleaves the (8 hi | 8 lo) groups -- a 256-byte LDS row is the 64 hi values
// of the tile's channels followed by their 64 lo values -- so that a transposed fragment read again covers 32 channels of
// ONE kind.  Every wave computes the whole 64 x 64 tile (hi*lo + lo*hi + hi*hi: 12 MFMAs and 16 transposed reads per
// k-step) for one of the four 16-pixel k-steps of a stage; the four partial tiles are summed through LDS before the
// float atomics (which go straight into the logical gradient: no quadrant temporary, no combine pass).
template <typename T, int WM, int WN, bool SPLIT = false>
__global__ void __launch_bounds__(256)
k_conv_wgrad(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw, WgradArgs a) {
  constexpr int BM = SPLIT ? 64 : 64 * WM, BN = SPLIT ? 64 : 64 * WN;      // logical channels / columns per workgroup
  constexpr int E = Chunk<T>::E;
  constexpr int BKP = (sizeof(T) == 2) ? 64 : 32;  // pixels per stage
  constexpr int RA = SPLIT ? 256 : BM * sizeof(T), RB = SPLIT ? 256 : BN * sizeof(T);  // LDS row bytes
  constexpr int TA = BKP * RA, TB = BKP * RB;
  constexpr int STAGE = TA + TB;
  constexpr int AI = TA / 1024 / 4, BI = TB / 1024 / 4;  // DMA instructions per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware order: the workgroups of one XCD (blockIdx % 8) walk consecutive work items, and
  // work items are split-major, so all (n, m) tiles of one pixel slice share one L2.
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt / 8, r = nt % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int tiles = a.tiles_n * a.tiles_m;
  const int split = bid / tiles, trem = bid % tiles;
  // output-channel tiles fastest: the workgroups that share a column block of x (the big operand; dy is small) run
  // back to back on one XCD and hit its L2
  const int n0 = (trem / a.tiles_m) * BN, co0 = (trem % a.tiles_m) * BM;
  const int p_begin = split * a.pix_per_split;
  const int p_end = min(a.M, p_begin + a.pix_per_split);
  if (p_begin >= p_end) return;
  const int taps = a.ks * a.ks;

  // DMA descriptors.  A chunk at tile byte offset o: row o / R, physical chunk (o % R) / 16.
  int a_prow[AI], a_lc[AI];
  for (int i = 0; i < AI; ++i) {
    const int o = (wave * AI + i) * 1024 + lane * 16;
    const int prow = o / RA, pcol = (o % RA) >> 4;
    a_prow[i] = prow;
    a_lc[i] = (sizeof(T) == 2) ? (pcol ^ ((prow & 3) << 2)) : pcol;     // SPLIT: position p -> kind p >> 3, 8-channel group p & 7
  }
  int b_prow[BI], b_tap[BI], b_cc[BI], b_oy[BI], b_ox[BI];
  bool b_ok[BI];
  for (int i = 0; i < BI; ++i) {
    const int o = (wave * BI + i) * 1024 + lane * 16;
    const int prow = o / RB, pcol = (o % RB) >> 4;
    b_prow[i] = prow;
    const int lc = (sizeof(T) == 2) ? (pcol ^ ((prow & 3) << 2)) : pcol;
    const int nq = n0 / E + (SPLIT ? (lc & 7) : lc);      // 8-channel group along the flattened (tap, ci) axis
    int tap, cc;
    chunk_to_tap(nq, a.ks, a.cpt_shift, tap, cc);
    b_tap[i] = tap;
    b_cc[i] = SPLIT ? (cc * 2 + (lc >> 3)) : cc;          // SPLIT: 16-byte chunk inside the pixel = 2 * group + kind
    b_ok[i] = (nq < a.nchunks) && (tap < taps);
    const int pix = p_begin + prow;
    const int rem = pix % (a.H * a.W);
    b_oy[i] = rem / a.W;
    b_ox[i] = rem % a.W;
  }

  auto stage_load = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + TA;
    const int pbase = p_begin + kt * BKP;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int pix = pbase + a_prow[i];
      const int col = co0 + (SPLIT ? (a_lc[i] & 7) : a_lc[i]) * E;
      const void* src = g_zero_page;
      if (pix < p_end && col + E <= a.lddy && col < a.Cout) {
        if constexpr (SPLIT) src = dy + ((int64_t)pix * a.lddy + col) * 2 + (a_lc[i] >> 3) * 8;   // bf16 units: 4 B per channel
        else src = dy + (int64_t)pix * a.lddy + col;
      }
      glds16(src, sA + (wave * AI + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int pix = pbase + b_prow[i];
      const void* src = g_zero_page;
      if (pix < p_end && b_ok[i]) {
        constexpr int PS = SPLIT ? 2 : 1;       // bf16 elements per (logical) channel in global memory
        if (a.ks == 1) {
          src = x + (int64_t)pix * a.Cin * PS + (int64_t)b_cc[i] * E;
        } else {
          const int ky = b_tap[i] / 3, kx = b_tap[i] - ky * 3;
          const int iy = b_oy[i] + ky - 1, ix = b_ox[i] + kx - 1;
          if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
            src = x + ((int64_t)pix + (int64_t)(ky - 1) * a.W + (kx - 1)) * a.Cin * PS + (int64_t)b_cc[i] * E;
        }
      }
      glds16(src, sB + (wave * BI + i) * 1024);
      // advance this row's (oy, ox) by BKP pixels for the next stage
      if (a.ks != 1) {
        b_ox[i] += BKP;
        while (b_ox[i] >= a.W) {
          b_ox[i] -= a.W;
          if (++b_oy[i] == a.H) b_oy[i] = 0;
        }
      }
    }
  };

  f32x16 acc[WM][WN];
  for (int i = 0; i < WM; ++i)
    for (int j = 0; j < WN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int h = lane >> 5;
  // bf16 transpose-read lane roles: 16-lane group g, within-group t -> row q = t/4, cols 4*(t%4)
  const int g16 = lane >> 4, t16 = lane & 15;
  const int tq = t16 >> 2, tp = t16 & 3;

  auto stage_compute = [&](int buf) {
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sB = sA + TA;
    if constexpr (SPLIT) {
      const int s = wave;                       // this wave's k-step (16 pixels) of the stage
      s16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int prow = 16 * s + 8 * h + 4 * e + tq;
        const int sw = (prow & 3) << 6;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int cb = (f * 32 + (g16 & 1) * 16 + 4 * tp) * 2;      // byte column of the hi value; lo = + 128
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(sA + prow * 256 + (cb ^ sw)));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(sA + prow * 256 + ((cb + 128) ^ sw)));
          const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(sB + prow * 256 + (cb ^ sw)));
          const s16x4 v3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(sB + prow * 256 + ((cb + 128) ^ sw)));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ah[f][4 * e + k] = v0[k]; al[f][4 * e + k] = v1[k];
            bh[f][4 * e + k] = v2[k]; bl[f][4 * e + k] = v3[k];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), c, 0, 0, 0);
          acc[i][j] = c;
        }
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < BKP / 16; ++s) {
        s16x8 af[WM], bfr[WN];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int prow = 16 * s + 8 * h + 4 * e + tq;
          const int sw = (prow & 3) << 6;
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            const int col = wr * 32 * WM + i * 32 + (g16 & 1) * 16 + 4 * tp;
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)LDS_PTR(sA + prow * RA + ((col * 2) ^ sw)));
            af[i][4 * e + 0] = v[0]; af[i][4 * e + 1] = v[1]; af[i][4 * e + 2] = v[2]; af[i][4 * e + 3] = v[3];
          }
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const int col = wc * 32 * WN + j * 32 + (g16 & 1) * 16 + 4 * tp;
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)LDS_PTR(sB + prow * RB + ((col * 2) ^ sw)));
            bfr[j][4 * e + 0] = v[0]; bfr[j][4 * e + 1] = v[1]; bfr[j][4 * e + 2] = v[2]; bfr[j][4 * e + 3] = v[3];
          }
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bfr[j]), acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int s = 0; s < BKP / 2; ++s) {
        const int prow = 2 * s + h;
        float af[WM], bfr[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
          af[i] = *reinterpret_cast<const float*>(sA + prow * RA + (wr * 32 * WM + i * 32 + (lane & 31)) * 4);
#pragma unroll
        for (int j = 0; j < WN; ++j)
          bfr[j] = *reinterpret_cast<const float*>(sB + prow * RB + (wc * 32 * WN + j * 32 + (lane & 31)) * 4);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  const int KT = (p_end - p_begin + BKP - 1) / BKP;
  stage_load(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < KT - 1; ++kt) {
    stage_load(kt + 1, cur ^ 1);
    stage_compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  stage_compute(cur);

  if constexpr (SPLIT) {
    // the four waves hold partial 64 x 64 tiles over different k-steps: sum them through LDS (the operand stages are
    // dead), then one float atomic per element and pixel split
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [wave][tile (i, j)][r][lane]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 4096; e += 256) {
      const float v = (red[e] + red[4096 + e]) + (red[8192 + e] + red[12288 + e]);
      const int ln = e & 63, r = (e >> 6) & 15, t = e >> 10;
      const int co = co0 + (t >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
      const int n = n0 + (t & 1) * 32 + (ln & 31);
      if (co < a.Cout && n < a.Ntot) wg_emit(dw, a, split, co, n, v);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + wc * 32 * WN + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wr * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < a.Cout && n < a.Ntot) wg_emit(dw, a, split, co, n, acc[i][j][r]);
      }
    }
}


// SFOD_BF16X3 weight gradient of the wide layers (Cout >= 128 and >= 256 flattened (tap, ci) columns: fc1 / fc2, the
// 1x1 convolutions of the bottlenecks).  The 64 x 64 kernel above re-reads both operands once per 64 outputs and drains
// its DMA queue every 12 MFMAs; here a workgroup owns 128 output channels x 256 columns (8 waves = 2 x 4 of 64 x 64,
// every wave over ALL pixels: no LDS reduction), a stage is 32 pixels = two k-steps (24 MFMAs per wave between
// barriers) and three stages are in flight with a counted vmcnt, as in k_conv_fwd.  Same de-interleaving DMA: an LDS row
// (one pixel) holds the tile's hi values followed by its lo values.
__global__ void __launch_bounds__(512)
k_conv_wgrad_x3w(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ dw, WgradArgs a) {
  constexpr int BM = 128, BN = 256, BKP = 32;
  constexpr int RA = BM * 4, RB = BN * 4;                  // LDS row bytes (hi half | lo half)
  constexpr int TA = BKP * RA, TB = BKP * RB, STAGE = TA + TB;
  constexpr int AI = TA / 1024 / 8, BI = TB / 1024 / 8;    // DMA instructions per wave and stage (2 + 4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt / 8, r = nt % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int tiles = a.tiles_n * a.tiles_m;
  const int split = bid / tiles, trem = bid % tiles;
  // output-channel tiles fastest: the workgroups that share a column block of x (the big operand; dy is small) run
  // back to back on one XCD and hit its L2
  const int n0 = (trem / a.tiles_m) * BN, co0 = (trem % a.tiles_m) * BM;
  const int p_begin = split * a.pix_per_split;
  const int p_end = min(a.M, p_begin + a.pix_per_split);
  if (p_begin >= p_end) return;
  const int taps = a.ks * a.ks;

  int a_prow[AI], a_goff[AI];
  bool a_ok[AI];
  for (int i = 0; i < AI; ++i) {
    const int o = (wave * AI + i) * 1024 + lane * 16;
    const int prow = o / RA, pcol = (o % RA) >> 4;
    const int lc = pcol ^ ((prow & 3) << 2);               // position -> kind lc >> 4, 8-channel group lc & 15
    const int col = co0 + (lc & 15) * 8;
    a_prow[i] = prow;
    a_goff[i] = col * 2 + (lc >> 4) * 8;                   // bf16 units inside the pixel: 4 B per channel
    a_ok[i] = (col + 8 <= a.lddy) && (col < a.Cout);
  }
  int b_prow[BI], b_tap[BI], b_cc[BI], b_oy[BI], b_ox[BI];
  bool b_ok[BI];
  for (int i = 0; i < BI; ++i) {
    const int o = (wave * BI + i) * 1024 + lane * 16;
    const int prow = o / RB, pcol = (o % RB) >> 4;
    const int lc = pcol ^ ((prow & 3) << 2);               // kind lc >> 5, group lc & 31
    const int nq = n0 / 8 + (lc & 31);
    int tap, cc;
    chunk_to_tap(nq, a.ks, a.cpt_shift, tap, cc);
    b_prow[i] = prow;
    b_tap[i] = tap;
    b_cc[i] = cc * 2 + (lc >> 5);
    b_ok[i] = (nq < a.nchunks) && (tap < taps);
    const int rem = (p_begin + prow) % (a.H * a.W);
    b_oy[i] = rem / a.W;
    b_ox[i] = rem % a.W;
  }

  auto stage_load = [&](int kt, int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + TA;
    const int pbase = p_begin + kt * BKP;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int pix = pbase + a_prow[i];
      const void* src = (pix < p_end && a_ok[i]) ? (const void*)(dy + (int64_t)pix * a.lddy * 2 + a_goff[i])
                                                 : (const void*)g_zero_page;
      glds16(src, sA + (wave * AI + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int pix = pbase + b_prow[i];
      const void* src = g_zero_page;
      if (pix < p_end && b_ok[i]) {
        if (a.ks == 1) {
          src = x + (int64_t)pix * a.Cin * 2 + (int64_t)b_cc[i] * 8;
        } else {
          const int ky = b_tap[i] / 3, kx = b_tap[i] - ky * 3;
          const int iy = b_oy[i] + ky - 1, ix = b_ox[i] + kx - 1;
          if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
            src = x + ((int64_t)pix + (int64_t)(ky - 1) * a.W + (kx - 1)) * a.Cin * 2 + (int64_t)b_cc[i] * 8;
        }
      }
      glds16(src, sB + (wave * BI + i) * 1024);
      if (a.ks != 1) {
        b_ox[i] += BKP;
        while (b_ox[i] >= a.W) {
          b_ox[i] -= a.W;
          if (++b_oy[i] == a.H) b_oy[i] = 0;
        }
      }
    }
  };

  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int h = lane >> 5;
  const int g16 = lane >> 4, t16 = lane & 15;
  const int tq = t16 >> 2, tp = t16 & 3;

  auto stage_compute = [&](int buf) {
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sB = sA + TA;
#pragma unroll
    for (int s = 0; s < BKP / 16; ++s) {
      s16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int prow = 16 * s + 8 * h + 4 * e + tq;
        const int sw = (prow & 3) << 6;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int ca = (wr * 64 + f * 32 + (g16 & 1) * 16 + 4 * tp) * 2;      // byte column of the hi value
          const int cb = (wc * 64 + f * 32 + (g16 & 1) * 16 + 4 * tp) * 2;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(sA + prow * RA + (ca ^ sw)));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(sA + prow * RA + ((ca + RA / 2) ^ sw)));
          const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(sB + prow * RB + (cb ^ sw)));
          const s16x4 v3 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(sB + prow * RB + ((cb + RB / 2) ^ sw)));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            ah[f][4 * e + k] = v0[k]; al[f][4 * e + k] = v1[k];
            bh[f][4 * e + k] = v2[k]; bl[f][4 * e + k] = v3[k];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
    }
  };

  const int KT = (p_end - p_begin + BKP - 1) / BKP;
  stage_load(0, 0);
  if (KT > 1) stage_load(1, 1);
  int cur = 0, nxt = 2;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AI + BI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // stage kt landed for every wave; everyone is done with stage kt-1
    asm volatile("" ::: "memory");
    if (kt + 2 < KT) stage_load(kt + 2, nxt);
    stage_compute(cur);
    cur = (cur == 2) ? 0 : cur + 1;
    nxt = (nxt == 2) ? 0 : nxt + 1;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < a.Cout && n < a.Ntot) wg_emit(dw, a, split, co, n, acc[i][j][r]);
      }
    }
}

static bool use_patch_wgrad(const W3Plan& p, int ksize, int dt) {
  if (ksize != 3 || !is_bf16_storage(dt) || dt == SFOD_F16X3 || !p.ok || g_conv_algo == 1) return false;
  if (g_conv_algo == 2) return true;
  // tiny problems: not enough pixel tiles to give every (co, ci) block a few tiles per split
  return p.nsplit * p.tiles_per_split >= 3;
}

// The halo-patch weight-gradient kernel addresses x / dy with 32-bit byte offsets, so a batch whose tensors exceed
// 2 GiB (B = 8 frames of 1024x2048 at 64 / 128 channels) is processed in sub-batches of `nb` images: each sub-launch
// writes its slabs and the slab reduction of every sub-batch but the first accumulates.  -> plan of the first (largest)
// sub-batch; nb = images per sub-batch (0: the shape is not served by the kernel at all).
static W3Plan w3_plan_chunked(int B, int H, int W, int Cin, int Cout, int ksize, int lddy, int dt, int& nb) {
  nb = 0;
  W3Plan p;
  p.ok = 0;
  if (ksize != 3 || !is_bf16_storage(dt) || dt == SFOD_F16X3 || (int64_t)B * H * W == 0) return p;   // half pairs: forward only
  const int split = (dt == SFOD_BF16X3);
  for (int n = B; n >= 1; n = (n == 1) ? 0 : (n + 1) / 2) {
    p = sfod_w3_plan(n, H, W, Cin, Cout, lddy, split);
    if (p.ok) { nb = n; break; }
  }
  if (!p.ok || !use_patch_wgrad(p, ksize, dt)) { nb = 0; p.ok = 0; }
  return p;
}

static int w3_launch_chunked(const void* x, const void* dy, float* dw, void* ws, int64_t ws_bytes, int B, int H, int W,
                             int Cin, int Cout, int lddy, int dt, int out_mode, int nb, hipStream_t s) {
  const int split = (dt == SFOD_BF16X3);
  const int64_t eb = split ? 4 : 2;
  for (int b0 = 0; b0 < B; b0 += nb) {
    const int n = (B - b0 < nb) ? (B - b0) : nb;
    const W3Plan p = sfod_w3_plan(n, H, W, Cin, Cout, lddy, split);
    if (!p.ok || ws == nullptr || ws_bytes < p.ws_bytes) {
      sfod_set_error("wgrad: workspace too small (sfod_conv_wgrad_ws_bytes)");
      return SFOD_EBADARG;
    }
    const char* xb = (const char*)x + (int64_t)b0 * H * W * Cin * eb;
    const char* dyb = (const char*)dy + (int64_t)b0 * H * W * lddy * eb;
    // out_mode 0 (packed, += ) always accumulates; 1 (OIHW overwrite) becomes 2 (OIHW +=) after the first sub-batch
    const int mode = (b0 == 0) ? out_mode : (out_mode == 1 ? 2 : out_mode);
    const int rc = sfod_w3_launch(p, xb, dyb, dw, ws, n, H, W, Cin, Cout, lddy, mode, s, split);
    if (rc) return rc;
  }
  return 0;
}

// ---- deterministic mode (SFOD.DETERMINISTIC / SFOD_DETERMINISTIC=1): no float atomics in any weight / bias gradient -----
// The halo-patch 3x3 weight gradient always sums its pixel splits through slabs in a fixed order; the generic kernels (1x1,
// linear, first layer, fp32) combine theirs with float atomics, whose arrival order differs from run to run.  In this mode
// they store each split's partial tile into a slab of the workspace instead and k_wgrad_slab_sum adds the slabs in split
// order: run-to-run bit-identical gradients (tests/test_gpu_trajectory.py), for one more pass over splits x |dw| floats.
static std::atomic<int> g_deterministic{-1};
extern "C" int sfod_set_deterministic(int on) {
  g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed);
  return 0;
}
bool sfod_deterministic() {
  int v = g_deterministic.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* ev = getenv("SFOD_DETERMINISTIC");
    int want = (ev && atoi(ev) != 0) ? 1 : 0, expect = -1;
    g_deterministic.compare_exchange_strong(expect, want, std::memory_order_relaxed);
    v = g_deterministic.load(std::memory_order_relaxed);
  }
  return v != 0;
}
extern "C" int sfod_get_deterministic(void) { return sfod_deterministic() ? 1 : 0; }

__global__ void __launch_bounds__(256) k_wgrad_slab_sum(const float* __restrict__ ws, float* __restrict__ dw, int64_t n,
                                                       int nslab, int64_t stride) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 4 <= n && (stride & 3) == 0) {
      float4 acc = *reinterpret_cast<const float4*>(dw + i);
      for (int sl = 0; sl < nslab; ++sl) {
        const float4 v = *reinterpret_cast<const float4*>(ws + sl * stride + i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      *reinterpret_cast<float4*>(dw + i) = acc;
    } else {
      for (int64_t e = i; e < n && e < i + 4; ++e) {
        float acc = dw[e];
        for (int sl = 0; sl < nslab; ++sl) acc += ws[sl * stride + e];
        dw[e] = acc;
      }
    }
  }
}

// pixel splits + tile counts of the generic weight-gradient launch (the one rule for the launch and the workspace query)
struct GenWgradPlan { int wide, tiles_n, tiles_m, splits, pps; };
static GenWgradPlan gen_wgrad_plan(int M, int Cout, int Ntot, int dt) {
  GenWgradPlan g;
  const int split = (dt == SFOD_BF16X3);
  // wide tiles pay off when the OUTPUT is big (fc1: 1024 x 25088 -> 784 tiles; 1.29 -> 1.02 ms); a small output with a
  // long pixel axis (1x1 bottleneck convolutions, fc2) needs many pixel splits either way and the 64 x 64 kernel's
  // shorter prologue wins there (profiles/r2d_bench_wgrad.txt).  algo 3: A/B hook, always the 64 x 64 kernel
  g.wide = split && Cout >= 128 && Ntot >= 256 && g_conv_algo != 3 &&
           (g_conv_algo == 4 || (int64_t)((Ntot + 255) / 256) * ((Cout + 127) / 128) >= 256);
  if (g.wide) {
    // wide layers: 128 x 256 tiles, one 8-wave workgroup per CU.  Pixel splits: the fewest that fill >= 85 % of the
    // last round of 256 workgroups (more splits = more float-atomic traffic), each at least 8 stages long
    constexpr int BKP = 32;
    g.tiles_n = (Ntot + 255) / 256; g.tiles_m = (Cout + 127) / 128;
    const int tiles = g.tiles_n * g.tiles_m;
    const int max_splits = std::max(1, M / (8 * BKP));
    int best = 1;
    double best_eff = 0.0;
    for (int sp = 1; sp <= std::min(max_splits, 512); ++sp) {
      const int64_t wgs = (int64_t)tiles * sp;
      const double eff = (double)wgs / (double)((wgs + 255) / 256 * 256);
      if (eff > best_eff + 1e-9) { best_eff = eff; best = sp; }
      if (eff >= 0.85 && wgs >= 512) { best = sp; break; }
      if (wgs >= 4096) break;
    }
    int pps = (M + best - 1) / best;
    pps = (pps + BKP - 1) / BKP * BKP;
    g.pps = pps;
    g.splits = (M + pps - 1) / pps;
    return g;
  }
  const int BKP = (dt == SFOD_F32) ? 32 : 64;
  const int TILE = split ? 64 : 128;        // bf16x3: 64 logical channels x 64 logical columns per workgroup
  g.tiles_n = (Ntot + TILE - 1) / TILE; g.tiles_m = (Cout + TILE - 1) / TILE;
  // split the pixel axis so that the grid has ~4 workgroups per CU
  int splits = (1024 + g.tiles_n * g.tiles_m - 1) / (g.tiles_n * g.tiles_m);
  const int max_splits = (M + BKP - 1) / BKP;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int pps = (M + splits - 1) / splits;
  pps = (pps + BKP - 1) / BKP * BKP;
  g.pps = pps;
  g.splits = (M + pps - 1) / pps;
  return g;
}

extern "C" int64_t sfod_conv_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int ksize, int lddy, int dt) {
  if (!(conv_shape_fits(B, H, W, Cin, Cout, ksize) && sfod_ints_ok({lddy}))) return 0;      // hostile extents: not served / nothing
  int nb = 0;
  if (dt != SFOD_F32) {
    const W3Plan p = w3_plan_chunked(B, H, W, Cin, Cout, ksize, lddy, dt, nb);
    if (nb > 0) return p.ws_bytes;      // the first sub-batch is the largest
  }
  // the generic kernel accumulates with float atomics straight into dw -- unless the deterministic mode wants slabs
  if (!sfod_deterministic() || (int64_t)B * H * W == 0) return 0;
  const int Ntot = ksize * ksize * Cin;
  const GenWgradPlan g = gen_wgrad_plan(B * H * W, Cout, Ntot, dt);
  return g.splits > 1 ? (int64_t)g.splits * Cout * Ntot * 4 : 0;
}

extern "C" int sfod_conv_wgrad_oihw_supported(int B, int H, int W, int Cin, int Cout, int ksize, int lddy, int dt) {
  if (!(conv_shape_fits(B, H, W, Cin, Cout, ksize) && sfod_ints_ok({lddy}))) return 0;      // hostile extents: not served / nothing
  int nb;
  w3_plan_chunked(B, H, W, Cin, Cout, ksize, lddy, dt, nb);
  return nb > 0 ? 1 : 0;
}

extern "C" int sfod_conv_wgrad_oihw(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin,
                                    int Cout, int ksize, int lddy, int dt, int accumulate, void* ws,
                                    int64_t ws_bytes, void* stream) {
  SFOD_REQUIRE(conv_shape_fits(B, H, W, Cin, Cout, ksize) && sfod_ints_ok({lddy}) && sfod_i64s_ok({ws_bytes}),
               "conv_wgrad_oihw: negative or oversized extent");
  SFOD_REQUIRE(x != nullptr && dy != nullptr && dw_oihw != nullptr, "conv_wgrad_oihw: null operand");
  SFOD_REQUIRE(lddy >= Cout, "conv_wgrad_oihw: lddy < Cout");
  int nb;
  w3_plan_chunked(B, H, W, Cin, Cout, ksize, lddy, dt, nb);
  SFOD_REQUIRE(nb > 0, "wgrad_oihw: shape not served by the halo-patch kernel (query sfod_conv_wgrad_oihw_supported)");
  return w3_launch_chunked(x, dy, dw_oihw, ws, ws_bytes, B, H, W, Cin, Cout, lddy, dt, accumulate ? 2 : 1, nb,
                           (hipStream_t)stream);
}

extern "C" int sfod_conv_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin,
                               int Cout, int ksize, int lddy, int dt, void* ws, int64_t ws_bytes,
                               void* stream) {
  SFOD_REQUIRE(conv_shape_fits(B, H, W, Cin, Cout, ksize) && sfod_ints_ok({lddy}) && sfod_i64s_ok({ws_bytes}),
               "conv_wgrad: negative or oversized extent");
  SFOD_REQUIRE(x != nullptr && dy != nullptr && dw != nullptr, "conv_wgrad: null operand");
  SFOD_REQUIRE(lddy >= Cout, "conv_wgrad: lddy < Cout");
  SFOD_REQUIRE(ksize == 1 || ksize == 3, "wgrad: ksize must be 1 or 3");
  SFOD_REQUIRE(dt == SFOD_F32 || dt == SFOD_BF16 || dt == SFOD_BF16X3, "wgrad: unknown dt");
  const int E = (dt == SFOD_F32) ? 4 : 8;
  SFOD_REQUIRE(Cin % E == 0 && lddy % E == 0, "wgrad: Cin / lddy must be multiples of the 16-byte chunk (bf16x3: of 8)");
  if ((int64_t)B * H * W == 0) return 0;
  const int split = (dt == SFOD_BF16X3);
  SFOD_REQUIRE(!split || Cout % 8 == 0, "wgrad: bf16x3 needs Cout % 8 == 0");
  {
    int nb;
    w3_plan_chunked(B, H, W, Cin, Cout, ksize, lddy, dt, nb);
    if (nb > 0) return w3_launch_chunked(x, dy, dw, ws, ws_bytes, B, H, W, Cin, Cout, lddy, dt, 0, nb, (hipStream_t)stream);
  }
  hipStream_t s = (hipStream_t)stream;
  float* dw_out = dw;
  WgradArgs a;
  a.M = B * H * W; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.lddy = lddy;
  const int cpt = Cin / E;
  a.cpt_shift = 0;
  if (ksize == 3) {
    a.cpt_shift = ilog2_exact(cpt);
    SFOD_REQUIRE(a.cpt_shift >= 0, "wgrad3x3: Cin/chunk must be a power of two");
  }
  a.nchunks = ksize * ksize * cpt;
  a.Ntot = ksize * ksize * Cin;
  if (a.M == 0) return 0;
  const GenWgradPlan g = gen_wgrad_plan(a.M, Cout, a.Ntot, dt);
  a.pix_per_split = g.pps;
  a.tiles_n = g.tiles_n;
  a.tiles_m = g.tiles_m;
  // deterministic mode: every split but a lone one stores into its slab; the slabs are summed into dw in split order
  a.slab_stride = 0;
  const bool slabs = sfod_deterministic() && g.splits > 1;
  if (slabs) {
    const int64_t need = (int64_t)g.splits * Cout * a.Ntot * 4;
    if (ws == nullptr || ws_bytes < need) {
      sfod_set_error("wgrad: deterministic mode needs the workspace of sfod_conv_wgrad_ws_bytes");
      return SFOD_EBADARG;
    }
    a.slab_stride = (int64_t)Cout * a.Ntot;
    dw_out = (float*)ws;
  }
  const int tiles = g.tiles_n * g.tiles_m;
  if (g.wide) {
    constexpr int LDS = 3 * 32 * (128 + 256) * 4;
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wgrad_x3w),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    SFOD_REQUIRE(attr_rc == hipSuccess, "wgrad: cannot raise the dynamic LDS limit");
    hipLaunchKernelGGL(k_conv_wgrad_x3w, dim3(tiles * g.splits), dim3(512), LDS, s, (const bf16_t*)x, (const bf16_t*)dy,
                       dw_out, a);
  } else {
    dim3 grid(tiles * g.splits);
    if (dt == SFOD_F32)
      hipLaunchKernelGGL((k_conv_wgrad<float, 2, 2>), grid, dim3(256), 2 * 2 * 32 * 512, s, (const float*)x,
                         (const float*)dy, dw_out, a);
    else if (split)
      hipLaunchKernelGGL((k_conv_wgrad<bf16_t, 2, 2, true>), grid, dim3(256), 2 * 2 * 64 * 256, s, (const bf16_t*)x,
                         (const bf16_t*)dy, dw_out, a);
    else
      hipLaunchKernelGGL((k_conv_wgrad<bf16_t, 2, 2>), grid, dim3(256), 2 * 2 * 64 * 256, s, (const bf16_t*)x,
                         (const bf16_t*)dy, dw_out, a);
  }
  int rc = sfod_check_launch(g.wide ? "conv_wgrad_x3w" : "conv_wgrad");
  if (rc == 0 && slabs) {
    const int64_t n = (int64_t)Cout * a.Ntot;
    hipLaunchKernelGGL(k_wgrad_slab_sum, dim3(cdiv(n, 1024) > 4096 ? 4096 : cdiv(n, 1024)), dim3(256), 0, s,
                       (const float*)ws, dw, n, g.splits, a.slab_stride);
    rc = sfod_check_launch("wgrad_slab_sum");
  }
  return rc;
}
