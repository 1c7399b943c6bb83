// Halo-patch 3x3 weight gradient, bf16 MFMA, gfx950.
//
//   dw[co][tap][ci] = sum_pixels dy[pix][co] * x[pix + tap][ci]          (GEMM K axis = pixels)
//
// The generic kernel (gemm_conv.hip) streams an im2col slice of x per (tap, ci) tile, i.e. every
// input pixel crosses L2 -> LDS nine times per output tile, and resolves its split-K with float
// atomics.  Here a workgroup owns a (CO x 32 output channels) x (64 input channels) x 9 taps block
// of dw and walks TH x TW pixel tiles (<= 256 pixels) of the images: per tile the (TH+2) x (TW+2)
// halo patch of x is DMA'd into LDS once and all nine taps read it at shifted rows; dy streams in
// 64-pixel chunks.  Partial results of the pixel splits go to fp32 slabs with plain stores and a
// second kernel sums the slabs in a fixed order (deterministic, no atomics).
//
//   LDS   both operands are pixel-major and stored as 32-channel "planes" with 64-byte rows: a
//         ds_read_b64_tr_b16 (hardware transpose) of one wave half touches 4 consecutive rows x 64 B
//         = all 64 banks once -> conflict free without a swizzle, for any tap shift, and the kx
//         shift is an immediate offset.
//           x patch[2] : 2 planes x 384 rows x 64 B = 48 KiB each
//           dy ring[3] : CO planes x 64 rows x 64 B (16 KiB for CO = 4)
//   waves 8.  CO=4: wave = (co fragment 0..3, ci fragment 0..1), nine 32x32 accumulators (one per tap)
//             CO=2: (Cout <= 64) waves 0-3 / 4-7 take the even / odd half of the k-steps of every
//                   chunk and write separate slabs
//   K loop body = one pixel tile = 4 stages (64-pixel chunks of dy); one s_barrier per stage; DMA
//         (buffer_load ... lds, out-of-range lanes read zeros) runs two chunks / one whole patch
//         ahead behind counted s_waitcnt vmcnt(N).
#include "conv_internal.h"
#include <type_traits>
#include <atomic>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// Knock-out switches for tools/experiments/wgrad_knockout.sh (time floors of the pipelined bf16x3 loop; results are
// WRONG by construction, the product build defines none of them):
//   W3_KO_MFMA   no MFMAs (the fragment registers are still consumed)      W3_KO_XREADS  x fragment rows read once per
//   W3_KO_DMA    no LDS-DMA inside the tile loop (prologue data is reused)                tile instead of per chunk / row
//   W3_KO_BARRIER no s_barrier inside the tile loop
namespace {

template <int J> using IC = std::integral_constant<int, J>;

constexpr int XROWS = 384;                       // patch capacity (pixels incl. halo)
constexpr int XPLANE = XROWS * 64;               // 24576
constexpr int XBUF = 2 * XPLANE;                 // 49152
constexpr int DY_OFF = 2 * XBUF;                 // 98304
constexpr int DYSLOT = 4 * 64 * 64;              // 16384 (CO = 4 planes)
constexpr int TAB_OFF = DY_OFF + 3 * DYSLOT;     // 147456
constexpr int TABP_OFF = TAB_OFF;                // u16 [256]  tile pixel -> patch row (tap 0,0)
constexpr int TABT_OFF = TAB_OFF + 512;          // u16 [256]  tile pixel -> ty<<8|tx, 0xffff outside the tile
constexpr int TABR_OFF = TAB_OFF + 1024;         // u16 [384]  patch row -> py<<8|px, 0xffff unused
constexpr int TABX_OFF = TAB_OFF + 1024 + 768;   // PIPE: i32 [384]  patch row -> byte offset relative to the tile's first pixel (interior tiles), REL_NONE unused
constexpr int TABD_OFF = TABX_OFF + 1536;        // PIPE: i32 [256]  tile pixel -> byte offset of its dy row relative to the tile's first pixel, REL_NONE outside
constexpr int LDS_TOTAL = TABD_OFF + 1024;
constexpr unsigned OOB_OFF = 0x80000000u;

struct W3Args {
  const bf16_t* x;
  const bf16_t* dy;
  float* slab;
  int B, H, W, Cin, Cout, lddy;
  int TH, TW, PW;
  int tiles_y, tiles_x;      // per image
  int ntiles;                // B * tiles_y * tiles_x
  int tiles_per_split;
  int co_tiles, ci_tiles, nsplit;
};

__device__ __forceinline__ void bufload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(l), 16, (int)voff, (int)soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// ds_read_b64_tr_b16 through inline asm: hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of the
// builtin form whenever an LDS-DMA is in flight, which would drain the prefetch queue every stage.
// The asm form is invisible to the waitcnt pass, so the consumer waits with wait_lgkm<N>() below
// (LDS returns in order) followed by a sched_barrier so that no MFMA is hoisted above the wait.
template <int OFF> __device__ __forceinline__ s16x4 tr_read(unsigned addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 cat8(s16x4 a, s16x4 b) {
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// SPLIT (SFOD_BF16X3 operands, CO = 4): x / dy hold (8 hi | 8 lo) bf16 groups, 4 bytes per logical channel.  The DMA
// de-interleaves them: a lane fetches every OTHER 16-byte chunk of its pixel, so that an LDS plane is again 32 channels
// x 64-byte rows of ONE kind -- x planes = (hi, lo) of 32 logical input channels, dy planes = (hi, lo) x 2 fragments of
// 32 logical output channels -- and the conflict-free transposed reads of the bf16 kernel apply unchanged.  A wave owns
// one 32 (co) x 32 (ci) logical block for all nine taps and one of the four k-steps of every 64-pixel chunk (the four
// k-step groups write separate slabs); per tap it issues hi*lo + lo*hi + hi*hi.  W3Args sizes are LOGICAL channels.
//
// PIPE (SPLIT only, round 3): the same arithmetic with the per-chunk critical path shortened --
//   * the tile-pixel -> patch-row lookups (tabP) of the wave's four chunks are read ONCE before the loop (they do not
//     depend on the tile), instead of a dependent LDS read in front of every chunk's fragment reads;
//   * the first filter row of the NEXT chunk's x fragments is requested during the last filter row's MFMAs of this
//     chunk (the patch of the current tile is complete and nobody writes it before the next tile), so that only the
//     four dy reads wait behind the stage barrier;
//   * the stage's LDS-DMA instructions are issued between the three filter rows' MFMA groups instead of in front of
//     the fragment reads (both waves of a SIMD leave the barrier together: DMA issue in front kept the matrix pipe idle);
//   * raised wave priority over the MFMA groups.
template <int CO, bool SPLIT = false, bool PIPE = false>
__global__ void __launch_bounds__(512)
k_wgrad3x3_patch(W3Args a) {
  static_assert(!SPLIT || CO == 4, "bf16x3 operands use the four-plane dy ring");
  static_assert(!PIPE || SPLIT, "the pipelined chunk loop exists for the bf16x3 variant");
  constexpr int ND = CO / 2;                 // dy DMA instructions per wave and chunk
  constexpr int KG = SPLIT ? 4 : ((CO == 4) ? 1 : 2);      // k-step groups (wave groups that split a chunk's k-steps)
  constexpr int KS = 4 / KG;                 // k-steps (16 pixels) per wave and chunk
  constexpr int EB = SPLIT ? 4 : 2;          // bytes per (logical) channel in global memory
  constexpr int CO_TILE = SPLIT ? 64 : CO * 32, CI_TILE = SPLIT ? 32 : 64;   // logical channels per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wco = SPLIT ? (wave & 1) : ((CO == 4) ? (wave >> 1) : ((wave >> 1) & 1));
  const int wci = SPLIT ? 0 : (wave & 1);
  const int kg = SPLIT ? (wave >> 1) : ((CO == 4) ? 0 : (wave >> 2));
  const int h = lane >> 5, g1 = (lane >> 4) & 1, t16 = lane & 15;
  const int tq = t16 >> 2, tp = t16 & 3;

  // ---- work decode: blockIdx -> (split, co tile, ci tile); XCD-aware so that the workgroups of one
  // XCD share the pixel range (x / dy lines are then L2 hits for all (co, ci) tiles of the split)
  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt / 8, r = nt % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int pairs = a.co_tiles * a.ci_tiles;
  const int split = bid / pairs;
  const int pr = bid - split * pairs;
  const int co0 = (pr / a.ci_tiles) * CO_TILE, ci0 = (pr % a.ci_tiles) * CI_TILE;
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
  const int PW = a.PW;
  const int npix = a.TH * a.TW;

  // ---- lookup tables ---------------------------------------------------------------------------------
  unsigned short* tabP = reinterpret_cast<unsigned short*>(smem + TABP_OFF);
  unsigned short* tabT = reinterpret_cast<unsigned short*>(smem + TABT_OFF);
  unsigned short* tabR = reinterpret_cast<unsigned short*>(smem + TABR_OFF);
  if (threadIdx.x < 256) {
    const int p = threadIdx.x;
    if (p < npix) {
      const int ty = p / a.TW, tx = p - ty * a.TW;
      tabP[p] = (unsigned short)(ty * PW + tx);
      tabT[p] = (unsigned short)((ty << 8) | tx);
    } else {
      tabP[p] = 0;
      tabT[p] = 0xffff;
    }
  }
  if (threadIdx.x < XROWS) {
    const int r = threadIdx.x;
    const int py = r / PW, px = r - py * PW;
    tabR[r] = (py < a.TH + 2) ? (unsigned short)((py << 8) | px) : (unsigned short)0xffff;
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.x, (short)0, (int)min((int64_t)0x7fffffff, (int64_t)a.B * a.H * a.W * a.Cin * EB), 0x00020000);
  const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.dy, (short)0, (int)min((int64_t)0x7fffffff, (int64_t)a.B * a.H * a.W * a.lddy * EB), 0x00020000);

  // per-lane constant parts of the DMA descriptors
  // patch piece k (0..5): instruction q = wave*6+k -> plane q/24, rows 16*(q%24) + lane/4, chunk lane%4
  // (the patch row of piece k is re-read from tabR where it is needed: six registers less to carry through the loop)
  // plane = q / 24 = wave / 4 and the chunk (lane & 3) do not depend on k: ONE byte offset inside a pixel per lane
  int pr_col;       // BYTE offset inside a pixel of this lane's 16-byte chunk, -1 if beyond Cin
  {
    const int plane = wave >> 2;
    if constexpr (SPLIT) {     // plane 0 / 1 = hi / lo of the 32 logical channels ci0 ..: group (lane & 3), every other chunk
      const int c = ci0 + (lane & 3) * 8;
      pr_col = (c < a.Cin) ? (c * 4 + plane * 16) : -1;
    } else {
      const int c = ci0 + plane * 32 + (lane & 3) * 8;
      pr_col = (c < a.Cin) ? c * 2 : -1;
    }
  }
  auto patch_row = [&](int k) { return (int)tabR[((wave * 6 + k) % 24) * 16 + (lane >> 2)]; };
  // dy piece i (0..ND-1): instruction q = wave*ND+i -> plane q/4, rows 16*(q%4) + lane/4.  ND = 2: both pieces of a
  // wave lie in one plane ((2 wave + i) >> 2 does not depend on i), ND = 1 trivially: ONE byte offset per lane
  int dy_col;       // BYTE offset inside a pixel, -1: no load
  {
    const int q = wave * ND;
    if constexpr (SPLIT) {     // plane p: fragment p >> 1 (32 logical output channels), p & 1 = hi / lo
      const int pl = q >> 2;
      const int c = co0 + (pl >> 1) * 32 + (lane & 3) * 8;
      dy_col = (c + 8 <= a.lddy && c < a.Cout) ? (c * 4 + (pl & 1) * 16) : -1;
    } else {
      const int c = co0 + (q >> 2) * 32 + (lane & 3) * 8;
      dy_col = (c + 8 <= a.lddy && c < a.Cout) ? c * 2 : -1;
    }
  }

  // Interior tiles (no image border inside the halo / the tile): every DMA offset is a per-tile scalar base plus a
  // per-lane constant; lanes that never load (table padding, channel tails) carry 0x80000000, which lands
  // beyond num_records for any base (< 2^31) -> zeros.  Border tiles take the general path below.
  // PIPE: these per-lane constants live in two small LDS tables instead (14 registers the pipelined loop does not have:
  // with them it spilled, and a scratch reload in front of a DMA drains vmcnt); the look-up goes out behind an MFMA group.
  unsigned pr_rel[PIPE ? 1 : 6];
  unsigned dy_rel[PIPE ? 1 : 4][ND];
  unsigned* tabX = reinterpret_cast<unsigned*>(smem + TABX_OFF);
  unsigned* tabD = reinterpret_cast<unsigned*>(smem + TABD_OFF);
  // table entries are SIGNED byte offsets relative to the tile's first pixel (the halo's first row / column lies before
  // it); REL_NONE marks rows / lanes that never load (no real offset comes near it: |offset| < (TH + 2) * W * C * 4)
  constexpr unsigned REL_NONE = 0x40000000u;
  const unsigned colx = pr_col >= 0 ? (unsigned)pr_col : REL_NONE, cold = dy_col >= 0 ? (unsigned)dy_col : REL_NONE;
  if constexpr (PIPE) {
    if (threadIdx.x < XROWS) {
      const int pk = tabR[threadIdx.x];
      const int py = pk >> 8, px = pk & 255;
      tabX[threadIdx.x] = (pk != 0xffff) ? (unsigned)(((py - 1) * a.W + (px - 1)) * a.Cin * EB) : REL_NONE;
    }
    if (threadIdx.x < 256) {
      const int tt = tabT[threadIdx.x];
      tabD[threadIdx.x] = (tt != 0xffff) ? (unsigned)(((tt >> 8) * a.W + (tt & 255)) * a.lddy * EB) : REL_NONE;
    }
    __syncthreads();
    pr_rel[0] = 0u;
    dy_rel[0][0] = 0u;
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int pk = patch_row(k);
      const int py = pk >> 8, px = pk & 255;
      pr_rel[PIPE ? 0 : k] = (pk != 0xffff && pr_col >= 0)
                      ? (unsigned)(((py - 1) * a.W + (px - 1)) * a.Cin * EB + pr_col) : OOB_OFF;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int q = wave * ND + i;
        const int row = (q & 3) * 16 + (lane >> 2);
        const int tt = tabT[c * 64 + row];
        dy_rel[PIPE ? 0 : c][i] = (tt != 0xffff && dy_col >= 0)
                           ? (unsigned)(((tt >> 8) * a.W + (tt & 255)) * a.lddy * EB + dy_col) : OOB_OFF;
      }
  }
  // interior-tile offset from a table entry and the lane's column part (either may be "never loads")
  auto rel_off = [&](unsigned base, unsigned t, unsigned col) {
    return (t == REL_NONE || col == REL_NONE) ? OOB_OFF : base + t + col;
  };

  // Tile origins (image, y0, x0) of the current and the next tile are carried in scalar registers and advanced
  // once per tile: decoding them with two runtime integer divisions at every DMA issue (~10 per tile) cost
  // about as many VALU cycles per stage as the stage's MFMAs.
  struct Org { int b, y0, x0; unsigned base_x, base_dy; bool in_x, in_dy; };
  auto finish = [&](Org o) {
    const int pix = (o.b * a.H + o.y0) * a.W + o.x0;
    o.base_dy = (unsigned)(pix * a.lddy * EB);
    o.base_x = (unsigned)(pix * a.Cin * EB);       // offset of the tile's first pixel; the halo starts one row / column before
    o.in_dy = (o.y0 + a.TH <= a.H) && (o.x0 + a.TW <= a.W);
    o.in_x = o.in_dy && o.y0 >= 1 && o.x0 >= 1 && (o.y0 + a.TH + 1 <= a.H) && (o.x0 + a.TW + 1 <= a.W);
    return o;
  };
  auto tile_origin = [&](int t) {
    const int per = a.tiles_y * a.tiles_x;
    Org o;
    o.b = t / per;
    const int r = t - o.b * per;
    const int tyi = r / a.tiles_x;
    o.y0 = tyi * a.TH;
    o.x0 = (r - tyi * a.tiles_x) * a.TW;
    return finish(o);
  };
  auto advance = [&](Org o) {
    o.x0 += a.TW;
    if (o.x0 >= a.tiles_x * a.TW) {
      o.x0 = 0;
      o.y0 += a.TH;
      if (o.y0 >= a.tiles_y * a.TH) { o.y0 = 0; ++o.b; }
    }
    return finish(o);
  };
  // byte offset of patch piece k / dy piece i of chunk c for tile o (OOB_OFF: the lane reads zeros)
  auto patch_off = [&](int k, Org o) -> unsigned {
    if (o.in_x) {
      if constexpr (PIPE) return rel_off(o.base_x, tabX[((wave * 6 + k) % 24) * 16 + (lane >> 2)], colx);
      else return o.base_x + pr_rel[PIPE ? 0 : k];
    }
    const int b = o.b, y0 = o.y0, x0 = o.x0;
    const int pk = patch_row(k);             // border tiles only: an LDS read instead of a carried register
    const int py = pk >> 8, px = pk & 255;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = pk != 0xffff && pr_col >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    return ok ? (unsigned)(((b * a.H + iy) * a.W + ix) * a.Cin * EB + pr_col) : OOB_OFF;
  };
  auto issue_patch = [&](int k, Org o, int buf) {
    bufload16(xres, patch_off(k, o), 0u, smem + buf * XBUF + (wave * 6 + k) * 1024);
  };
  // two pieces: both look-ups first (one LDS round trip in the table-driven PIPE path), then both DMAs
  auto issue_patch2 = [&](int k0, int k1, Org o, int buf) {
    const unsigned o0 = patch_off(k0, o), o1 = patch_off(k1, o);
    bufload16(xres, o0, 0u, smem + buf * XBUF + (wave * 6 + k0) * 1024);
    bufload16(xres, o1, 0u, smem + buf * XBUF + (wave * 6 + k1) * 1024);
  };
  // dy chunk c of tile t into ring slot
  auto issue_dy = [&](Org o, auto cc, int slot) {
    constexpr int c = decltype(cc)::value;
    const int b = o.b, y0 = o.y0, x0 = o.x0;
    unsigned offs[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int q = wave * ND + i;
      if (o.in_dy) {
        if constexpr (PIPE) offs[i] = rel_off(o.base_dy, tabD[c * 64 + (q & 3) * 16 + (lane >> 2)], cold);
        else offs[i] = o.base_dy + dy_rel[PIPE ? 0 : c][i];
      } else {
        const int row = (q & 3) * 16 + (lane >> 2);
        const int tt = tabT[c * 64 + row];
        const int ty = tt >> 8, tx = tt & 255;
        const bool ok = tt != 0xffff && dy_col >= 0 && y0 + ty < a.H && x0 + tx < a.W;
        offs[i] = ok ? (unsigned)(((b * a.H + y0 + ty) * a.W + x0 + tx) * a.lddy * EB + dy_col) : OOB_OFF;
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) bufload16(dres, offs[i], 0u, smem + DY_OFF + slot * DYSLOT + (wave * ND + i) * 1024);
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // lane parts of the fragment addresses
  const int cbb = (g1 * 16 + 4 * tp) * 2;                       // byte column inside a 64-byte plane row
  const int a_lane = (SPLIT ? 2 * wco : wco) * 4096 + (8 * h + tq) * 64 + cbb;   // dy: + slot base + (16 s + 4 e) * 64 (SPLIT: hi plane; lo = + 4096)
  const int b_lane = wci * XPLANE + cbb;                         // x : + buffer base + row * 64          (SPLIT: hi plane; lo = + XPLANE)

  // LDS byte addresses are 32-bit offsets from the start of the dynamic LDS segment (the kernel has
  // no static LDS, so the segment starts at LDS address 0).
  auto compute_chunk = [&](int c, int xb_off, int dyb_off) {
#pragma unroll
    for (int sl = 0; sl < KS; ++sl) {
      const int s = kg * KS + sl;   // k-step inside the chunk (kg is wave-uniform)
      // pixels of this lane's two transposed reads: p = 64 c + 16 s + 8 h + 4 e + tq
      const int p0 = c * 64 + s * 16 + 8 * h + tq;
      const unsigned rx0 = (unsigned)(xb_off + tabP[p0] * 64 + b_lane);
      const unsigned rx1 = (unsigned)(xb_off + tabP[p0 + 4] * 64 + b_lane);
      const unsigned ra = (unsigned)(dyb_off + a_lane + s * 1024);
      if constexpr (SPLIT) {
        const s16x4 a0h = tr_read<0>(ra), a1h = tr_read<256>(ra), a0l = tr_read<4096>(ra), a1l = tr_read<4096 + 256>(ra);
        s16x4 q[2][12];   // per filter row: (rx0, rx1) x kx 0..2 of the hi plane, then of the lo plane
        auto issue_row = [&](s16x4* d, unsigned r0, unsigned r1) {
          d[0] = tr_read<0>(r0);   d[1] = tr_read<0>(r1);
          d[2] = tr_read<64>(r0);  d[3] = tr_read<64>(r1);
          d[4] = tr_read<128>(r0); d[5] = tr_read<128>(r1);
          d[6] = tr_read<XPLANE>(r0);       d[7] = tr_read<XPLANE>(r1);
          d[8] = tr_read<XPLANE + 64>(r0);  d[9] = tr_read<XPLANE + 64>(r1);
          d[10] = tr_read<XPLANE + 128>(r0); d[11] = tr_read<XPLANE + 128>(r1);
        };
        issue_row(q[0], rx0, rx1);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          if (ky < 2) {
            issue_row(q[(ky + 1) & 1], rx0 + (ky + 1) * PW * 64, rx1 + (ky + 1) * PW * 64);
            wait_lgkm<12>();
          } else {
            wait_lgkm<0>();
          }
          const bf16x8 ah = cat8(a0h, a1h), al = cat8(a0l, a1l);
          const s16x4* r = q[ky & 1];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const bf16x8 bh = cat8(r[2 * kx], r[2 * kx + 1]), bl = cat8(r[6 + 2 * kx], r[6 + 2 * kx + 1]);
            f32x16 c = acc[ky * 3 + kx];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
            acc[ky * 3 + kx] = c;
          }
        }
        continue;
      }
      const s16x4 a0 = tr_read<0>(ra), a1 = tr_read<256>(ra);
      s16x4 q[2][6];
      // group ky = 0
      q[0][0] = tr_read<0>(rx0);   q[0][1] = tr_read<0>(rx1);
      q[0][2] = tr_read<64>(rx0);  q[0][3] = tr_read<64>(rx1);
      q[0][4] = tr_read<128>(rx0); q[0][5] = tr_read<128>(rx1);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        if (ky < 2) {
          const unsigned r0 = rx0 + (ky + 1) * PW * 64, r1 = rx1 + (ky + 1) * PW * 64;
          q[(ky + 1) & 1][0] = tr_read<0>(r0);   q[(ky + 1) & 1][1] = tr_read<0>(r1);
          q[(ky + 1) & 1][2] = tr_read<64>(r0);  q[(ky + 1) & 1][3] = tr_read<64>(r1);
          q[(ky + 1) & 1][4] = tr_read<128>(r0); q[(ky + 1) & 1][5] = tr_read<128>(r1);
          wait_lgkm<6>();
        } else {
          wait_lgkm<0>();
        }
        const bf16x8 af = cat8(a0, a1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              af, cat8(q[ky & 1][2 * kx], q[ky & 1][2 * kx + 1]), acc[ky * 3 + kx], 0, 0, 0);
      }
    }
  };

  // ---- PIPE: per-wave constants and the pipelined chunk -------------------------------------------------
  // rows of this lane's two transposed reads (tile pixels p0, p0 + 4) for the wave's k-step of chunks 0..3, as byte
  // offsets inside a patch plane, two per register
  unsigned prow[4] = {0u, 0u, 0u, 0u};
  if constexpr (PIPE) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int p0 = c * 64 + kg * 16 + 8 * h + tq;
      prow[c] = (unsigned)tabP[p0] | ((unsigned)tabP[p0 + 4] << 16);
    }
  }
  s16x4 qq[2][12];   // PIPE: fragment rows, alternating; the NEXT chunk's first row is requested a stage early
  auto issue_row_p = [&](s16x4* d, unsigned r0, unsigned r1) {
#ifdef W3_KO_XREADS
    asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]));
    asm volatile("" : "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]));
    (void)r0; (void)r1;
    return;
#endif
    d[0] = tr_read<0>(r0);   d[1] = tr_read<0>(r1);
    d[2] = tr_read<64>(r0);  d[3] = tr_read<64>(r1);
    d[4] = tr_read<128>(r0); d[5] = tr_read<128>(r1);
    d[6] = tr_read<XPLANE>(r0);       d[7] = tr_read<XPLANE>(r1);
    d[8] = tr_read<XPLANE + 64>(r0);  d[9] = tr_read<XPLANE + 64>(r1);
    d[10] = tr_read<XPLANE + 128>(r0); d[11] = tr_read<XPLANE + 128>(r1);
  };
  // chunk c of the current tile.  have0: its first filter row is already in flight in qq[c & 1] (requested by chunk
  // c - 1); dma0 / dma1 / dma2: the stage's DMA issue, split over the three filter rows
  auto compute_chunk_pipe = [&](auto cc, int xb_off, int dyb_off, auto&& dma0, auto&& dma1) {
    constexpr int c = decltype(cc)::value;
    constexpr int P = c & 1;
    constexpr bool have0 = (c > 0), pref = (c < 3);
    unsigned pw = prow[c];
    asm volatile("" : "+v"(pw));     // keeps the per-chunk fragment addresses from being hoisted out of the tile loop (VGPRs)
    const unsigned rx0 = (unsigned)(xb_off + (int)(pw & 0xffffu) * 64 + b_lane);
    const unsigned rx1 = (unsigned)(xb_off + (int)(pw >> 16) * 64 + b_lane);
    const unsigned ra = (unsigned)(dyb_off + a_lane + kg * 1024);
    const s16x4 a0h = tr_read<0>(ra), a1h = tr_read<256>(ra), a0l = tr_read<4096>(ra), a1l = tr_read<4096 + 256>(ra);
    if constexpr (!have0) issue_row_p(qq[P], rx0, rx1);
    auto mfma_row = [&](const s16x4* r, int ky) {
      const bf16x8 ah = cat8(a0h, a1h), al = cat8(a0l, a1l);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const bf16x8 bh = cat8(r[2 * kx], r[2 * kx + 1]), bl = cat8(r[6 + 2 * kx], r[6 + 2 * kx + 1]);
        f32x16 v = acc[ky * 3 + kx];
#ifdef W3_KO_MFMA
        asm volatile("" ::"v"(ah), "v"(al), "v"(bh), "v"(bl));      // operands stay live, no matrix work
        v[kx] += 1.0f;
#else
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, v, 0, 0, 0);
#endif
        acc[ky * 3 + kx] = v;
      }
      __builtin_amdgcn_s_setprio(0);
      // the row's MFMAs stay in front of the next row's reads: sunk below them (the scheduler likes to) they would keep
      // a third fragment row alive (24 VGPRs -> spills at the 256-register budget of two waves per SIMD)
      __builtin_amdgcn_sched_barrier(0);
    };
    // filter row 0.  The stage's DMA instructions go BEHIND an MFMA group, where one fragment row (not two) is alive:
    // their address arithmetic (border tiles: table lookups + bounds tests) otherwise pushes the wave over its 256
    // registers; the matrix pipe works on the group just issued while they go out.
#ifdef W3_KO_XREADS
#define W3_WAIT12() wait_lgkm<0>()
#else
#define W3_WAIT12() wait_lgkm<12>()
#endif
    issue_row_p(qq[P ^ 1], rx0 + PW * 64, rx1 + PW * 64);
    W3_WAIT12();
    mfma_row(qq[P], 0);
    dma0();
    // filter row 1
    issue_row_p(qq[P], rx0 + 2 * PW * 64, rx1 + 2 * PW * 64);
    W3_WAIT12();
    mfma_row(qq[P ^ 1], 1);
    dma1();
    // filter row 2 (+ the next chunk's first row)
    if constexpr (pref) {
      unsigned pn = prow[(c + 1) & 3];
      asm volatile("" : "+v"(pn));
      const unsigned nx0 = (unsigned)(xb_off + (int)(pn & 0xffffu) * 64 + b_lane);
      const unsigned nx1 = (unsigned)(xb_off + (int)(pn >> 16) * 64 + b_lane);
      issue_row_p(qq[P ^ 1], nx0, nx1);
      W3_WAIT12();
    } else {
      wait_lgkm<0>();
    }
    mfma_row(qq[P], 2);
  };

  if (t_begin < t_end) {
    // ---- prologue: patch of the first tile, dy chunks 0 and 1 ---------------------------------------
    Org cur = tile_origin(t_begin);
#pragma unroll
    for (int k = 0; k < 6; ++k) issue_patch(k, cur, 0);
    issue_dy(cur, IC<0>{}, 0);
    issue_dy(cur, IC<1>{}, 1);
    int gc = 0;  // global chunk counter (ring slot = gc % 3)
    for (int t = t_begin; t < t_end; ++t, cur = advance(cur)) {
      const Org nxt = advance(cur);
      const bool last = (t == t_end - 1);
      const int xbuf = (t - t_begin) & 1;
      auto stage = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int N = (c == 0) ? ND : (c == 3 ? 1 + ND : 2 + ND);
        constexpr int NL = (c == 3) ? 0 : ND;
#ifdef W3_KO_DMA
        wait_vm<0>();
#else
        if (last) wait_vm<NL>(); else wait_vm<N>();
#endif
#ifndef W3_KO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        auto dma_patch = [&]() {
#ifdef W3_KO_DMA
          return;
#endif
          if (!last) {
            if constexpr (c == 0) issue_patch2(0, 1, nxt, xbuf ^ 1);
            else if constexpr (c == 1) issue_patch2(2, 3, nxt, xbuf ^ 1);
            else if constexpr (c == 2) issue_patch(4, nxt, xbuf ^ 1);
            else issue_patch(5, nxt, xbuf ^ 1);
          }
        };
        auto dma_dy = [&]() {
#ifdef W3_KO_DMA
          return;
#endif
          // dy chunk two stages ahead
          constexpr int c2 = (c + 2) & 3;
          const int t2 = (c >= 2) ? t + 1 : t;
          int slot = gc + 2;
          slot -= (slot >= 3) ? 3 : 0;
          slot -= (slot >= 3) ? 3 : 0;
          if (t2 < t_end) issue_dy((c >= 2) ? nxt : cur, IC<c2>{}, slot);
        };
        if constexpr (PIPE) {
          // per-wave DMA order unchanged (patch pieces, then dy): the counted vmcnt waits above still hold
          compute_chunk_pipe(cc, xbuf * XBUF, DY_OFF + gc * DYSLOT, dma_patch, dma_dy);
        } else {
          dma_patch();
          dma_dy();
          compute_chunk(c, xbuf * XBUF, DY_OFF + gc * DYSLOT);
        }
        gc = (gc == 2) ? 0 : gc + 1;
      };
      stage(IC<0>{}); stage(IC<1>{}); stage(IC<2>{}); stage(IC<3>{});
    }
  }

  // ---- write the partial block into this split's slab ----------------------------------------------------
  const int slab_idx = split * KG + kg;
  float* out = a.slab + (int64_t)slab_idx * a.Cout * 9 * a.Cin;
  const int ci = ci0 + wci * 32 + (lane & 31);      // logical channels in both modes
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < a.Cout && ci < a.Cin) out[((int64_t)co * 9 + t) * a.Cin + ci] = acc[t][r];
    }
}

// =================================================================================================================
// W64 (bf16x3 operands, round 3): a 64 co x 64 ci block per workgroup on 128-pixel tiles.
//
// The knock-out runs of the 64 x 32 kernel above (profiles/r3_wgrad_knockout.txt) say what bounds it: not the matrix
// pipe, not the barriers -- the operand stream L2 -> LDS (5.7 GB per conv4_2 launch), because the split format's LDS
// footprint leaves a workgroup only a 64 x 32 block of dw: 84 FLOP per DMA'd byte.  Same LDS budget, different cut:
//   x patch[2] : 4 planes (ci fragment 0 / 1 x hi / lo) x 192 rows x 64 B = 48 KiB each   (128-pixel tile + halo)
//   dy ring[3] : 4 planes (co fragment 0 / 1 x hi / lo) x  64 rows x 64 B = 16 KiB each
//   waves 8 = (co fragment) x (ci fragment) x (k-step pair): wave (wco, wci, kg) owns one 32 x 32 block x 9 taps and the
//   k-steps 2 kg, 2 kg + 1 of every 64-pixel chunk; two k-step groups -> two slabs per pixel split (was four).
// Per 128 pixels a workgroup now DMAs 48 + 32 KiB for 9.4 MFLOP: 118 FLOP/B.  The chunk loop is the pipelined one
// (look-ups hoisted, next k-step's first fragment row requested early, DMA behind the MFMA groups, LDS offset tables);
// a stage = one chunk = two k-steps = 54 MFMAs per wave between barriers (was 27).  Same MFMA sequence per accumulator
// element?  No: the pixel tiles differ (128 instead of 256 pixels, other tile shapes), so partial sums are grouped
// differently -- results agree with the other loops to fp32 summation order, not bit for bit.
// =================================================================================================================
constexpr int X2_ROWS = 192;
constexpr int X2_PLANE = X2_ROWS * 64;          // 12288
constexpr int X2_BUF = 4 * X2_PLANE;            // 49152
constexpr int D2_OFF = 2 * X2_BUF;              // 98304
constexpr int D2_SLOT = 4 * 4096;               // 16384
constexpr int T2_OFF = D2_OFF + 3 * D2_SLOT;    // 147456
constexpr int T2P_OFF = T2_OFF;                 // u16 [128]  tile pixel -> patch row (tap 0,0)
constexpr int T2T_OFF = T2_OFF + 256;           // u16 [128]  tile pixel -> ty<<8|tx, 0xffff outside the tile
constexpr int T2R_OFF = T2_OFF + 512;           // u16 [192]  patch row -> py<<8|px, 0xffff unused
constexpr int T2X_OFF = T2_OFF + 896;           // i32 [192]  patch row -> byte offset relative to the tile's first pixel
constexpr int T2D_OFF = T2X_OFF + 768;          // i32 [128]  tile pixel -> byte offset of its dy row, relative likewise
constexpr int LDS2_TOTAL = T2D_OFF + 512;       // 149632

__global__ void __launch_bounds__(512)
k_wgrad3x3_w64(W3Args a) {
  constexpr int EB = 4;
  constexpr unsigned REL_NONE = 0x40000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wco = wave & 1, wci = (wave >> 1) & 1, kg = wave >> 2;
  const int h = lane >> 5, g1 = (lane >> 4) & 1, t16 = lane & 15;
  const int tq = t16 >> 2, tp = t16 & 3;

  int bid = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt / 8, r = nt % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int pairs = a.co_tiles * a.ci_tiles;
  const int split = bid / pairs;
  const int pr = bid - split * pairs;
  const int co0 = (pr / a.ci_tiles) * 64, ci0 = (pr % a.ci_tiles) * 64;
  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
  const int PW = a.PW;
  const int npix = a.TH * a.TW;

  // ---- lookup tables -------------------------------------------------------------------------------------------
  unsigned short* tabP = reinterpret_cast<unsigned short*>(smem + T2P_OFF);
  unsigned short* tabT = reinterpret_cast<unsigned short*>(smem + T2T_OFF);
  unsigned short* tabR = reinterpret_cast<unsigned short*>(smem + T2R_OFF);
  unsigned* tabX = reinterpret_cast<unsigned*>(smem + T2X_OFF);
  unsigned* tabD = reinterpret_cast<unsigned*>(smem + T2D_OFF);
  if (threadIdx.x < 128) {
    const int p = threadIdx.x;
    if (p < npix) {
      const int ty = p / a.TW, tx = p - ty * a.TW;
      tabP[p] = (unsigned short)(ty * PW + tx);
      tabT[p] = (unsigned short)((ty << 8) | tx);
      tabD[p] = (unsigned)((ty * a.W + tx) * a.lddy * EB);
    } else {
      tabP[p] = 0;
      tabT[p] = 0xffff;
      tabD[p] = REL_NONE;
    }
  }
  if (threadIdx.x < X2_ROWS) {
    const int r = threadIdx.x;
    const int py = r / PW, px = r - py * PW;
    const bool ok = py < a.TH + 2;
    tabR[r] = ok ? (unsigned short)((py << 8) | px) : (unsigned short)0xffff;
    tabX[r] = ok ? (unsigned)(((py - 1) * a.W + (px - 1)) * a.Cin * EB) : REL_NONE;      // signed, see REL_NONE above
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.x, (short)0, (int)min((int64_t)0x7fffffff, (int64_t)a.B * a.H * a.W * a.Cin * EB), 0x00020000);
  const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc(
      (void*)a.dy, (short)0, (int)min((int64_t)0x7fffffff, (int64_t)a.B * a.H * a.W * a.lddy * EB), 0x00020000);

  // DMA pieces (1 KiB = 16 rows x 64 B of ONE plane).  x: 48 per patch, piece q = wave * 6 + k -> plane q / 12 = wave >> 1,
  // row group q % 12 = (wave & 1) * 6 + k;  dy: 16 per chunk, piece q = wave * 2 + i -> plane wave >> 1, row group
  // 2 * (wave & 1) + i.  Plane p = (32-channel fragment p >> 1, hi / lo p & 1): a lane takes the hi or the lo 16-byte
  // chunk of the 8-channel group (lane & 3) of its pixel -- the de-interleaving DMA of the 64 x 32 kernel.
  const int plane = wave >> 1;
  int pr_col, dy_col;        // byte offset inside a pixel, -1: the lane never loads (channel tails)
  {
    const int cx = ci0 + (plane >> 1) * 32 + (lane & 3) * 8;
    pr_col = (cx < a.Cin) ? (cx * 4 + (plane & 1) * 16) : -1;
    const int cd = co0 + (plane >> 1) * 32 + (lane & 3) * 8;
    dy_col = (cd + 8 <= a.lddy && cd < a.Cout) ? (cd * 4 + (plane & 1) * 16) : -1;
  }
  const unsigned colx = pr_col >= 0 ? (unsigned)pr_col : REL_NONE, cold = dy_col >= 0 ? (unsigned)dy_col : REL_NONE;
  auto rel_off = [&](unsigned base, unsigned t, unsigned col) {
    return (t == REL_NONE || col == REL_NONE) ? OOB_OFF : base + t + col;
  };

  struct Org { int b, y0, x0; unsigned base_x, base_dy; bool in_x, in_dy; };
  auto finish = [&](Org o) {
    const int pix = (o.b * a.H + o.y0) * a.W + o.x0;
    o.base_dy = (unsigned)(pix * a.lddy * EB);
    o.base_x = (unsigned)(pix * a.Cin * EB);
    o.in_dy = (o.y0 + a.TH <= a.H) && (o.x0 + a.TW <= a.W);
    o.in_x = o.in_dy && o.y0 >= 1 && o.x0 >= 1 && (o.y0 + a.TH + 1 <= a.H) && (o.x0 + a.TW + 1 <= a.W);
    return o;
  };
  auto tile_origin = [&](int t) {
    const int per = a.tiles_y * a.tiles_x;
    Org o;
    o.b = t / per;
    const int r = t - o.b * per;
    const int tyi = r / a.tiles_x;
    o.y0 = tyi * a.TH;
    o.x0 = (r - tyi * a.tiles_x) * a.TW;
    return finish(o);
  };
  auto advance = [&](Org o) {
    o.x0 += a.TW;
    if (o.x0 >= a.tiles_x * a.TW) {
      o.x0 = 0;
      o.y0 += a.TH;
      if (o.y0 >= a.tiles_y * a.TH) { o.y0 = 0; ++o.b; }
    }
    return finish(o);
  };
  auto patch_off = [&](int k, Org o) -> unsigned {
    const int row = ((wave & 1) * 6 + k) * 16 + (lane >> 2);
    if (o.in_x) return rel_off(o.base_x, tabX[row], colx);
    const int pk = tabR[row];
    const int py = pk >> 8, px = pk & 255;
    const int iy = o.y0 - 1 + py, ix = o.x0 - 1 + px;
    const bool ok = pk != 0xffff && pr_col >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    return ok ? (unsigned)(((o.b * a.H + iy) * a.W + ix) * a.Cin * EB + pr_col) : OOB_OFF;
  };
  auto issue_patch = [&](int k, Org o, int buf) {
    bufload16(xres, patch_off(k, o), 0u, smem + buf * X2_BUF + (wave * 6 + k) * 1024);
  };
  // dy chunk c (0 / 1) of tile o into ring slot
  auto issue_dy = [&](Org o, int c, int slot) {
    unsigned offs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = c * 64 + (2 * (wave & 1) + i) * 16 + (lane >> 2);
      if (o.in_dy) {
        offs[i] = rel_off(o.base_dy, tabD[row], cold);
      } else {
        const int tt = tabT[row];
        const int ty = tt >> 8, tx = tt & 255;
        const bool ok = tt != 0xffff && dy_col >= 0 && o.y0 + ty < a.H && o.x0 + tx < a.W;
        offs[i] = ok ? (unsigned)(((o.b * a.H + o.y0 + ty) * a.W + o.x0 + tx) * a.lddy * EB + dy_col) : OOB_OFF;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) bufload16(dres, offs[i], 0u, smem + D2_OFF + slot * D2_SLOT + (wave * 2 + i) * 1024);
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int cbb = (g1 * 16 + 4 * tp) * 2;                          // byte column inside a 64-byte plane row
  const int a_lane = (2 * wco) * 4096 + (8 * h + tq) * 64 + cbb;   // dy: hi plane of fragment wco (lo = + 4096), + 1024 per k-step
  const int b_lane = (2 * wci) * X2_PLANE + cbb;                   // x : hi plane of fragment wci (lo = + X2_PLANE), + row * 64

  // patch rows of this lane's two transposed reads (tile pixels p0, p0 + 4) for (chunk c, k-step 2 kg + sl), two per register
  unsigned prow[4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int p0 = c * 64 + (2 * kg + sl) * 16 + 8 * h + tq;
      prow[c * 2 + sl] = (unsigned)tabP[p0] | ((unsigned)tabP[p0 + 4] << 16);
    }

  s16x4 qq[2][12];
  auto issue_row = [&](s16x4* d, unsigned r0, unsigned r1) {
    d[0] = tr_read<0>(r0);   d[1] = tr_read<0>(r1);
    d[2] = tr_read<64>(r0);  d[3] = tr_read<64>(r1);
    d[4] = tr_read<128>(r0); d[5] = tr_read<128>(r1);
    d[6] = tr_read<X2_PLANE>(r0);       d[7] = tr_read<X2_PLANE>(r1);
    d[8] = tr_read<X2_PLANE + 64>(r0);  d[9] = tr_read<X2_PLANE + 64>(r1);
    d[10] = tr_read<X2_PLANE + 128>(r0); d[11] = tr_read<X2_PLANE + 128>(r1);
  };
  // One k-step (16 pixels x 9 taps) of chunk c.  Fragment rows alternate between qq[X] and qq[X ^ 1], X = sl: a k-step
  // reads its filter rows 1 and 2 and the NEXT k-step's row 0 while the previous row's MFMAs run.  first: nothing was
  // requested for this k-step yet (first k-step of a tile); more: request the next k-step's row 0 (not after the last
  // k-step of a tile: the next patch may not be complete).  hook0 / hook1: DMA issue behind the first two MFMA groups.
  auto kstep = [&](auto cc, auto slc, int xb_off, int dyb_off, auto&& hook0, auto&& hook1) {
    constexpr int c = decltype(cc)::value, sl = decltype(slc)::value;
    constexpr int X = sl;
    constexpr bool first = (c == 0 && sl == 0), more = !(c == 1 && sl == 1);
    unsigned pw = prow[c * 2 + sl];
    asm volatile("" : "+v"(pw));       // keeps the fragment addresses of all four k-steps from being hoisted (VGPRs)
    const unsigned rx0 = (unsigned)(xb_off + (int)(pw & 0xffffu) * 64 + b_lane);
    const unsigned rx1 = (unsigned)(xb_off + (int)(pw >> 16) * 64 + b_lane);
    const unsigned ra = (unsigned)(dyb_off + a_lane + (2 * kg + sl) * 1024);
    const s16x4 a0h = tr_read<0>(ra), a1h = tr_read<256>(ra), a0l = tr_read<4096>(ra), a1l = tr_read<4096 + 256>(ra);
    if constexpr (first) issue_row(qq[X], rx0, rx1);
    auto mfma_row = [&](const s16x4* r, int ky) {
      const bf16x8 ah = cat8(a0h, a1h), al = cat8(a0l, a1l);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const bf16x8 bh = cat8(r[2 * kx], r[2 * kx + 1]), bl = cat8(r[6 + 2 * kx], r[6 + 2 * kx + 1]);
        f32x16 v = acc[ky * 3 + kx];
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, v, 0, 0, 0);
        acc[ky * 3 + kx] = v;
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);     // see the 64 x 32 kernel: a sunk MFMA group keeps a third fragment row alive
    };
    issue_row(qq[X ^ 1], rx0 + PW * 64, rx1 + PW * 64);
    wait_lgkm<12>();
    mfma_row(qq[X], 0);
    hook0();
    issue_row(qq[X], rx0 + 2 * PW * 64, rx1 + 2 * PW * 64);
    wait_lgkm<12>();
    mfma_row(qq[X ^ 1], 1);
    hook1();
    if constexpr (more) {
      constexpr int nidx = (sl == 0) ? c * 2 + 1 : (c + 1) * 2;     // next k-step: (c, 1) or (c + 1, 0), same patch
      unsigned pn = prow[nidx & 3];
      asm volatile("" : "+v"(pn));
      const unsigned nx0 = (unsigned)(xb_off + (int)(pn & 0xffffu) * 64 + b_lane);
      const unsigned nx1 = (unsigned)(xb_off + (int)(pn >> 16) * 64 + b_lane);
      issue_row(qq[X ^ 1], nx0, nx1);
      wait_lgkm<12>();
    } else {
      wait_lgkm<0>();
    }
    mfma_row(qq[X], 2);
  };

  if (t_begin < t_end) {
    // ---- prologue: patch of the first tile, both dy chunks ---------------------------------------------------
    Org cur = tile_origin(t_begin);
#pragma unroll
    for (int k = 0; k < 6; ++k) issue_patch(k, cur, 0);
    issue_dy(cur, 0, 0);
    issue_dy(cur, 1, 1);
    int gc = 0;  // global chunk counter (ring slot = gc % 3)
    for (int t = t_begin; t < t_end; ++t, cur = advance(cur)) {
      const Org nxt = advance(cur);
      const bool last = (t == t_end - 1);
      const int xbuf = (t - t_begin) & 1;
      auto stage = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        // Per wave and stage the DMA order is patch x 3, dy x 2 (nothing in the last tile).  Stage 0 needs the patch and
        // dy chunk 0: everything but the 2 dy pieces of chunk 1 issued last; stage 1 needs dy chunk 1: everything but
        // stage 0's five (none in the last tile).
        if constexpr (c == 0) wait_vm<2>();
        else { if (last) wait_vm<0>(); else wait_vm<5>(); }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int slot = gc + 2;
        slot -= (slot >= 3) ? 3 : 0;
        const int xo = xbuf * X2_BUF, dyo = D2_OFF + gc * D2_SLOT;
        kstep(cc, IC<0>{}, xo, dyo,
              [&]() { if (!last) issue_patch(3 * c + 0, nxt, xbuf ^ 1); },
              [&]() { if (!last) issue_patch(3 * c + 1, nxt, xbuf ^ 1); });
        kstep(cc, IC<1>{}, xo, dyo,
              [&]() { if (!last) issue_patch(3 * c + 2, nxt, xbuf ^ 1); },
              [&]() { if (!last) issue_dy(nxt, c, slot); });
        gc = (gc == 2) ? 0 : gc + 1;
      };
      stage(IC<0>{}); stage(IC<1>{});
    }
  }

  // ---- write the partial block into this split's slab (two k-step groups -> two slabs per split) ---------------
  float* out = a.slab + (int64_t)(split * 2 + kg) * a.Cout * 9 * a.Cin;
  const int ci = ci0 + wci * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (co < a.Cout && ci < a.Cin) out[((int64_t)co * 9 + t) * a.Cin + ci] = acc[t][r];
    }
}

// dw[i] (+)= sum_s slab[s][i], fixed order.  MODE 0: dw is the packed [Cout][9][Cin] gradient (accumulated into);
// MODE 1 / 2: dw is the state-dict layout OIHW [Cout][Cin][3][3], overwritten (1) or accumulated into (2) -- the
// unpack pass and the zero fill of a packed temporary are folded into this reduction.
// SG slab groups per workgroup: 256 threads = (256 / SG) float4 elements x SG groups, each group sums every
// SG-th slab and the groups are combined through LDS in a fixed order (deterministic).  Small gradients
// (conv1_2: 9216 float4 x 512 slabs = 75 MB) would otherwise be read by 36 workgroups.
template <int MODE, int SG>
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ dw, int64_t n4, int nslab, int64_t stride4, int Cin) {
  constexpr int EPB = 256 / SG;                    // elements (float4) per workgroup
  __shared__ float4 part[SG > 1 ? 256 : 1];
  const int el = threadIdx.x % EPB, grp = threadIdx.x / EPB;
  for (int64_t base = (int64_t)blockIdx.x * EPB; base < n4; base += (int64_t)gridDim.x * EPB) {
    const int64_t i = base + el;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      for (int k = grp; k < nslab; k += SG) {
        const float4 v = reinterpret_cast<const float4*>(slab)[k * stride4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    if constexpr (SG > 1) {
      __syncthreads();                             // previous iteration's readers are done
      part[threadIdx.x] = s;
      __syncthreads();
      if (grp != 0) continue;
#pragma unroll
      for (int g2 = 1; g2 < SG; ++g2) {
        const float4 v = part[g2 * EPB + el];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    if (i >= n4) continue;
    if constexpr (MODE == 0) {
      float4 o = reinterpret_cast<const float4*>(dw)[i];
      o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
      reinterpret_cast<float4*>(dw)[i] = o;
    } else {
      const int64_t e = i * 4;                       // packed element (co, tap, ci .. ci+3)
      const int ci = (int)(e % Cin);
      const int64_t ct = e / Cin;
      const int tap = (int)(ct % 9);
      const int64_t co = ct / 9;
      const int64_t o = (co * Cin + ci) * 9 + tap;
      if constexpr (MODE == 2) { s.x += dw[o]; s.y += dw[o + 9]; s.z += dw[o + 18]; s.w += dw[o + 27]; }
      dw[o] = s.x; dw[o + 9] = s.y; dw[o + 18] = s.z; dw[o + 27] = s.w;
    }
  }
}

}  // namespace

// A/B knob of the bf16x3 weight gradient (environment SFOD_W3_PIPE): 2 (default) the 64 x 64-block kernel on 128-pixel
// tiles where Cin >= 64 (else 1), 1 the pipelined 64 x 32-block loop, 0 the round-2 loop.  1 and 0 give bit-identical
// results; 2 sums the same products in a different grouping (other pixel tiles).
static std::atomic<int> g_w3_pipe{-1};   // -1: not initialised (SFOD_W3_PIPE or 2)
extern "C" int sfod_set_wgrad3x3_pipe(int on) {
  g_w3_pipe.store((on < 0 || on > 2) ? 2 : on, std::memory_order_relaxed);
  return 0;
}
static int w3_mode() {
  int m = g_w3_pipe.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* ev = getenv("SFOD_W3_PIPE");
    m = ev ? atoi(ev) : 2;
    if (m < 0 || m > 2) m = 2;
    int expect = -1;
    g_w3_pipe.compare_exchange_strong(expect, m, std::memory_order_relaxed);
    m = g_w3_pipe.load(std::memory_order_relaxed);
  }
  return m;
}

// split != 0: SFOD_BF16X3 operands; Cin / Cout / lddy are LOGICAL channel counts in either case
W3Plan sfod_w3_plan(int B, int H, int W, int Cin, int Cout, int lddy, int split) {
  W3Plan p;
  p.ok = 0;
  p.w64 = 0;
  if (B < 1 || H < 1 || W < 1) return p;
  if (split && w3_mode() == 2 && Cin >= 64 && Cin % 8 == 0 && Cout % 8 == 0 && lddy % 8 == 0 &&
      (int64_t)B * H * W * Cin < ((int64_t)1 << 29) && (int64_t)B * H * W * lddy < ((int64_t)1 << 29)) {
    // 128-pixel tiles, patch (TH + 2) x (TW + 2) <= 192 rows
    double best = -1.0;
    for (int tw = 4; tw <= 64 && tw <= W + 3; ++tw) {
      int th = 128 / tw;
      while (th > 1 && (th + 2) * (tw + 2) > X2_ROWS) --th;
      if (th > H) th = H;
      if (th < 1 || (th + 2) * (tw + 2) > X2_ROWS) continue;
      const int ty = (H + th - 1) / th, tx = (W + tw - 1) / tw;
      th = (H + ty - 1) / ty;
      const double eff = (double)H * W / ((double)ty * tx * 128.0);
      const double score = eff + 1e-6 * tw;
      if (score > best) { best = score; p.TH = th; p.TW = tw; p.tiles_y = ty; p.tiles_x = tx; }
    }
    if (best >= 0.0) {
      p.w64 = 1;
      p.CO = 4;
      p.co_tiles = (Cout + 63) / 64;
      p.ci_tiles = (Cin + 63) / 64;
      const int pairs = p.co_tiles * p.ci_tiles;
      const int ntiles = B * p.tiles_y * p.tiles_x;
      int ns = 256 / pairs;
      if (ns < 1) ns = 1;
      const int64_t slab_bytes = (int64_t)Cout * 9 * Cin * 4 * 2;       // two k-step groups
      while (ns > 1 && ns * slab_bytes > ((int64_t)128 << 20)) --ns;
      if (ns > ntiles) ns = ntiles;
      p.tiles_per_split = (ntiles + ns - 1) / ns;
      p.nsplit = (ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
      p.nslab = p.nsplit * 2;
      p.ws_bytes = (int64_t)p.nslab * Cout * 9 * Cin * 4;
      p.ok = 1;
      return p;
    }
  }
  if (split) {
    if (Cin % 8 != 0 || Cout % 8 != 0 || lddy % 8 != 0) return p;
    if ((int64_t)B * H * W * Cin >= ((int64_t)1 << 29) || (int64_t)B * H * W * lddy >= ((int64_t)1 << 29)) return p;   // 32-bit byte offsets
  } else {
    if (Cin % 32 != 0 || Cout % 32 != 0 || lddy % 8 != 0) return p;
    if ((int64_t)B * H * W * Cin >= ((int64_t)1 << 30) || (int64_t)B * H * W * lddy >= ((int64_t)1 << 30)) return p;
  }
  double best = -1.0;
  for (int tw = 4; tw <= 128 && tw <= W + 3; ++tw) {
    int th = 256 / tw;
    while (th > 1 && (th + 2) * (tw + 2) > XROWS) --th;
    if (th > H) th = H;
    if (th < 1 || (th + 2) * (tw + 2) > XROWS || th + 2 > 255 || tw + 2 > 255) continue;
    const int ty = (H + th - 1) / th, tx = (W + tw - 1) / tw;
    th = (H + ty - 1) / ty;
    const double eff = (double)H * W / ((double)ty * tx * 256.0);
    const double score = eff + 1e-6 * tw;
    if (score > best) { best = score; p.TH = th; p.TW = tw; p.tiles_y = ty; p.tiles_x = tx; }
  }
  if (best < 0.0) return p;
  const int kg = split ? 4 : (Cout <= 64 ? 2 : 1);      // k-step groups = slabs per pixel split
  p.CO = split ? 4 : ((Cout <= 64) ? 2 : 4);
  p.co_tiles = split ? (Cout + 63) / 64 : (Cout + p.CO * 32 - 1) / (p.CO * 32);
  p.ci_tiles = split ? (Cin + 31) / 32 : (Cin + 63) / 64;
  const int pairs = p.co_tiles * p.ci_tiles;
  const int ntiles = B * p.tiles_y * p.tiles_x;
  // one workgroup per CU (LDS-bound occupancy): aim at 256 workgroups; the slabs the splits write
  // (and the reduction reads back) are kept below 128 MiB
  int ns = 256 / pairs;
  if (ns < 1) ns = 1;
  const int64_t slab_bytes = (int64_t)Cout * 9 * Cin * 4 * kg;
  while (ns > 1 && ns * slab_bytes > ((int64_t)128 << 20)) --ns;
  if (ns > ntiles) ns = ntiles;
  p.tiles_per_split = (ntiles + ns - 1) / ns;
  p.nsplit = (ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
  p.nslab = p.nsplit * kg;
  p.ws_bytes = (int64_t)p.nslab * Cout * 9 * Cin * 4;
  p.ok = 1;
  return p;
}

// split != 0 (SFOD_BF16X3): x / dy hold (hi, lo) pairs; Cin / Cout / lddy are LOGICAL channel counts in either case
int sfod_w3_launch(const W3Plan& p, const void* x, const void* dy, float* dw, void* ws, int B, int H, int W,
                   int Cin, int Cout, int lddy, int out_mode, hipStream_t s, int split) {
  W3Args a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.slab = (float*)ws;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.lddy = lddy;
  a.TH = p.TH; a.TW = p.TW; a.PW = p.TW + 2;
  a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x;
  a.ntiles = B * p.tiles_y * p.tiles_x;
  a.tiles_per_split = p.tiles_per_split;
  a.co_tiles = p.co_tiles; a.ci_tiles = p.ci_tiles; a.nsplit = p.nsplit;
  static const hipError_t attr_rc = []() {     // once per process (function-local static: thread-safe)
    const void* ks[4] = {(const void*)k_wgrad3x3_patch<4>, (const void*)k_wgrad3x3_patch<2>,
                         (const void*)k_wgrad3x3_patch<4, true>, (const void*)k_wgrad3x3_patch<4, true, true>};
    for (int i = 0; i < 4; ++i) {
      hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
      if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute((const void*)k_wgrad3x3_w64, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_TOTAL);
  }();
  if (attr_rc != hipSuccess) { sfod_set_error("hipFuncSetAttribute(w3): %s", hipGetErrorString(attr_rc)); return -(int)attr_rc; }
  dim3 grid(p.co_tiles * p.ci_tiles * p.nsplit), blk(512);
  const int pipe = w3_mode();
  if (split && p.w64) hipLaunchKernelGGL(k_wgrad3x3_w64, grid, blk, LDS2_TOTAL, s, a);
  else if (split && pipe) hipLaunchKernelGGL((k_wgrad3x3_patch<4, true, true>), grid, blk, LDS_TOTAL, s, a);
  else if (split) hipLaunchKernelGGL((k_wgrad3x3_patch<4, true>), grid, blk, LDS_TOTAL, s, a);
  else if (p.CO == 4) hipLaunchKernelGGL(k_wgrad3x3_patch<4>, grid, blk, LDS_TOTAL, s, a);
  else hipLaunchKernelGGL(k_wgrad3x3_patch<2>, grid, blk, LDS_TOTAL, s, a);
  int rc = sfod_check_launch("wgrad3x3_patch");
  if (rc) return rc;
  const int64_t n = (int64_t)Cout * 9 * Cin;  // multiple of 4 (Cin % 32 == 0; bf16x3: Cin % 8 == 0)
  const int64_t n4 = n / 4;
  const bool wide = true;                           // slabs spread over 8 groups per workgroup (more loads in flight)
  const int epb = wide ? 32 : 256;
  int g = (int)((n4 + epb - 1) / epb);
  if (g > 8192) g = 8192;
#define RED_LAUNCH(MODE_)                                                                                         \
  do {                                                                                                            \
    if (wide) hipLaunchKernelGGL((k_wgrad_reduce<MODE_, 8>), dim3(g), dim3(256), 0, s, (const float*)ws, dw, n4,  \
                                 p.nslab, n4, Cin);                                                               \
    else hipLaunchKernelGGL((k_wgrad_reduce<MODE_, 1>), dim3(g), dim3(256), 0, s, (const float*)ws, dw, n4,       \
                            p.nslab, n4, Cin);                                                                    \
  } while (0)
  if (out_mode == 0) RED_LAUNCH(0);
  else if (out_mode == 1) RED_LAUNCH(1);
  else RED_LAUNCH(2);
#undef RED_LAUNCH
  return sfod_check_launch("wgrad_reduce");
}
