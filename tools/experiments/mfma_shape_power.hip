// Experiment (round 2): do 16x16x32 bf16 MFMAs deliver more FLOP/s than 32x32x16 under the chip's power cap in a loop
// shaped like the bf16x3 conv inner loop (operands re-read from LDS with ds_read_b128, three MFMAs per operand pair)?
//   A: per K-step of 16: 2 A + 2 A' + 2 B + 2 B' fragment reads (b128), 12 x v_mfma_f32_32x32x16_bf16   (64x64 wave tile)
//   B: per K-step of 32: 4+4+4+4 reads, 48 x v_mfma_f32_16x16x32_bf16                                   (64x64 wave tile)
// 8 waves per workgroup, 2 workgroups per CU (80 KB LDS each), random operand bits.  Prints TFLOP/s of issued MFMA work.
// build: hipcc --offload-arch=gfx950 -O3 mfma_shape_power.hip -o mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ void __launch_bounds__(512, 4) k(const uint4* src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 73728 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = src[(blockIdx.x * 977 + i) % 65536];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sum = 0.f;
  if (MODE == 0) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int h = lane >> 5, l = lane & 31;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const int base = ((wave * 64 + l + s * 7) & 511) * 64 + ((it + s) & 7) * 4096;
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + i * 2048 + (2 * h) * 16) % 73728 & ~15));
          al[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + i * 2048 + (2 * h + 1) * 16) % 73728 & ~15));
          bh[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + 32768 + i * 2048 + (2 * h) * 16) % 73728 & ~15));
          bl[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + 32768 + i * 2048 + (2 * h + 1) * 16) % 73728 & ~15));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      __builtin_amdgcn_s_barrier();
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  } else {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    const int q = lane >> 4, l = lane & 15;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 9; s += 2) {      // two taps per K = 32 step (the 9th pairs with the next slice: count 4.5 steps)
        const int base = ((wave * 64 + l + s * 7) & 511) * 64 + ((it + s) & 7) * 4096 + (q >> 1) * 448;
        bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + i * 1024 + (2 * (q & 1)) * 16) % 73728 & ~15));
          al[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + i * 1024 + (2 * (q & 1) + 1) * 16) % 73728 & ~15));
          bh[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + 32768 + i * 1024 + (2 * (q & 1)) * 16) % 73728 & ~15));
          bl[i] = *reinterpret_cast<const bf16x8*>(smem + ((base + 32768 + i * 1024 + (2 * (q & 1) + 1) * 16) % 73728 & ~15));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      __builtin_amdgcn_s_barrier();
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  }
  out[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main() {
  std::mt19937 g(1);
  std::vector<unsigned short> h(65536 * 8);
  for (auto& v : h) { float f = std::normal_distribution<float>(0.f, 1.f)(g); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  uint4* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 512 * 512 * 4);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  const int iters = 400, grid = 512;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      for (int w = 0; w < 3; ++w) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 73728, 0, d, o, iters); else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 73728, 0, d, o, iters); }
      hipEventRecord(a);
      for (int w = 0; w < 10; ++w) { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 73728, 0, d, o, iters); else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 73728, 0, d, o, iters); }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
      // issued MFMA flops per wave and iteration: mode 0: 9 steps x 12 x (32*32*16*2); mode 1: 5 steps x 48 x (16*16*32*2)
      const double fl = (mode == 0 ? 9.0 * 12 * 32768 : 5.0 * 48 * 16384) * iters * 8.0 * grid;
      printf("%s  %.3f ms  %.1f TFLOP/s issued\n", mode == 0 ? "32x32x16" : "16x16x32", ms, fl / ms / 1e9);
    }
  return 0;
}
