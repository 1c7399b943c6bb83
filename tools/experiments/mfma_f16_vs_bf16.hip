// Experiment (round 3): the split-precision product on f16 pairs instead of bf16 pairs.
//  (1) rate under the power cap: the bf16x3 conv inner-loop shape (2+2+2+2 ds_read_b128, 12 x v_mfma_f32_32x32x16_{bf16,f16}
//      per K step, 8 waves, 2 workgroups per CU) on operands that ARE (hi, lo) pairs of N(0,1) values in the respective format
//      (the multiplier array's switching activity depends on the mantissa width that toggles);
//  (2) subnormal f16 inputs: does v_mfma_f32_32x32x16_f16 keep them (lo = f16(v - hi) of |v| < 2^-3 is subnormal)?
// build: hipcc --offload-arch=gfx950 -O3 mfma_f16_vs_bf16.hip -o mfma_f16_vs_bf16
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cstring>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int F16>
__global__ void __launch_bounds__(512, 4) k(const uint4* src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 73728 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = src[(blockIdx.x * 977 + i) % 65536];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sum = 0.f;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int h = lane >> 5, l = lane & 31;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int base = ((wave * 64 + l + s * 7) & 511) * 64 + ((it + s) & 7) * 4096;
      uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {      // 32-byte groups: 16 B of hi, 16 B of lo
        ah[i] = *reinterpret_cast<const uint4*>(smem + ((base + i * 2048 + h * 32) % 73728 & ~31));
        al[i] = *reinterpret_cast<const uint4*>(smem + (((base + i * 2048 + h * 32) % 73728 & ~31) + 16));
        bh[i] = *reinterpret_cast<const uint4*>(smem + ((base + 32768 + i * 2048 + h * 32) % 73728 & ~31));
        bl[i] = *reinterpret_cast<const uint4*>(smem + (((base + 32768 + i * 2048 + h * 32) % 73728 & ~31) + 16));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (F16) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bl[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[i]), __builtin_bit_cast(f16x8, bh[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[i]), __builtin_bit_cast(f16x8, bh[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bl[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[i]), __builtin_bit_cast(bf16x8, bh[j]), acc[i][j], 0, 0, 0);
          }
        }
    }
    __builtin_amdgcn_s_barrier();
  }
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

// one wave: D = A x B with A[m][k] = a_val for all m, k; B[k][n] = b_val -> every D element = 16 * a_val * b_val
__global__ void k_denorm(float* out, unsigned short a_bits, unsigned short b_bits) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, a_bits); b[i] = __builtin_bit_cast(_Float16, b_bits); }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf16_to(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  std::mt19937 g(1);
  const size_t n = 65536 * 8;                    // 16-bit words
  std::vector<unsigned short> hb(n), hf(n);
  for (size_t grp = 0; grp < n / 16; ++grp)
    for (int i = 0; i < 8; ++i) {
      float v = std::normal_distribution<float>(0.f, 1.f)(g);
      unsigned short h = bf16_of(v);
      hb[grp * 16 + i] = h;
      hb[grp * 16 + 8 + i] = bf16_of(v - bf16_to(h));
      __half hh = __float2half(v);
      __half hl = __float2half(v - __half2float(hh));
      memcpy(&hf[grp * 16 + i], &hh, 2);
      memcpy(&hf[grp * 16 + 8 + i], &hl, 2);
    }
  uint4 *db, *df; float* o;
  hipMalloc(&db, n * 2); hipMalloc(&df, n * 2); hipMalloc(&o, 512 * 512 * 4);
  hipMemcpy(db, hb.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(df, hf.data(), n * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  const int iters = 400, grid = 512;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      for (int w = 0; w < 30; ++w) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 73728, 0, db, o, iters); else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 73728, 0, df, o, iters); }
      hipEventRecord(a);
      for (int w = 0; w < 100; ++w) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 73728, 0, db, o, iters); else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 73728, 0, df, o, iters); }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 100;
      const double fl = 9.0 * 12 * 32768 * iters * 8.0 * grid;
      printf("%s pairs  %.3f ms  %.1f TFLOP/s issued (3 MFMAs per product: /3 algorithmic)\n", mode == 0 ? "bf16" : "f16 ", ms, fl / ms / 1e9);
    }
  // subnormal inputs: a = 2^-20 (subnormal f16: 0x0010), b = 1.0 -> 16 * 2^-20 = 1.5259e-5 if kept, 0 if flushed
  float r;
  hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, o, (unsigned short)0x0010, (unsigned short)0x3c00);
  hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
  printf("subnormal A (2^-20) x 1.0, K=16: %.6e  (kept: %.6e)\n", r, 16.0 * ldexp(1.0, -20));
  hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, o, (unsigned short)0x0001, (unsigned short)0x0001);
  hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
  printf("subnormal x subnormal (2^-24 x 2^-24), K=16: %.6e  (kept: %.6e)\n", r, 16.0 * ldexp(1.0, -48));
  return 0;
}
