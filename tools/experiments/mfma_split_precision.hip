// Experiment (round 2): accuracy of fp32-emulating split-operand MFMA products on gfx950.
//   one wave computes a 32x32 tile  C = X[32,K] * W[32,K]^T  in several arithmetic modes and compares
//   with a float64 reference on the host.  Modes: f32 MFMA (32x32x2), bf16 x1, bf16 x3 (hi/lo, 3 products),
//   bf16 x6 (3 terms), f16 x3 (hi/lo unscaled), f16 x3 with the operands pre-scaled by powers of two.
// build: hipcc --offload-arch=gfx950 -O3 mfma_split_precision.hip -o mfma_split_precision
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ void store_tile(float* C, const f32x16& acc) {
  const int lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + n] = acc[r];
}

// mode 0: f32 mfma, 1: bf16x1, 2: bf16x3, 3: bf16x6, 4: f16x3, 5: f16x3 scaled (sx, sw powers of two), 6: f16 x1
__global__ void k(const float* X, const float* W, int K, float* C, int mode, float sx, float sw) {
  const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
  f32x16 acc = {0};
  if (mode == 0) {
    // v_mfma_f32_32x32x2_f32: lane holds A[row r][k = h], B[k = h][col r]
    for (int k0 = 0; k0 < K; k0 += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X[r * K + k0 + h], W[r * K + k0 + h], acc, 0, 0, 0);
    store_tile(C, acc);
    return;
  }
  for (int k0 = 0; k0 < K; k0 += 16) {
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = X[r * K + k0 + 8 * h + j]; b[j] = W[r * K + k0 + 8 * h + j]; }
    if (mode == 1 || mode == 2 || mode == 3) {
      bf16x8 ah, al, al2, bh, bl, bl2;
      for (int j = 0; j < 8; ++j) {
        ah[j] = (__bf16)a[j]; float ra = a[j] - (float)ah[j]; al[j] = (__bf16)ra; al2[j] = (__bf16)(ra - (float)al[j]);
        bh[j] = (__bf16)b[j]; float rb = b[j] - (float)bh[j]; bl[j] = (__bf16)rb; bl2[j] = (__bf16)(rb - (float)bl[j]);
      }
      if (mode == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al2, bh, acc, 0, 0, 0);
      }
      if (mode >= 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    } else {
      f16x8 ah, al, bh, bl;
      const float fa = (mode == 5) ? sx : 1.f, fb = (mode == 5) ? sw : 1.f;
      for (int j = 0; j < 8; ++j) {
        const float av = a[j] * fa, bv = b[j] * fb;
        ah[j] = (_Float16)av; al[j] = (_Float16)(av - (float)ah[j]);
        bh[j] = (_Float16)bv; bl[j] = (_Float16)(bv - (float)bh[j]);
      }
      if (mode != 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  if (mode == 5) for (int i = 0; i < 16; ++i) acc[i] *= 1.f / (sx * sw);
  store_tile(C, acc);
}

int main() {
  const char* names[7] = {"f32 mfma", "bf16 x1", "bf16 x3", "bf16 x6", "f16 x3", "f16 x3 scaled", "f16 x1"};
  struct Case { const char* name; int K; int relu; float xs, ws; float sx, sw; };
  const Case cases[] = {
      {"conv act x weights (K=4608, relu x ~1, w ~0.02)", 4608, 1, 1.f, 0.02f, 1.f, 256.f},
      {"fc1 (K=25088, relu x, w 0.01)", 25088, 1, 1.f, 0.01f, 1.f, 512.f},
      {"dgrad (K=4608, dy ~1e-6, w 0.02)", 4608, 0, 1e-6f, 0.02f, 1048576.f * 4096.f, 256.f},
      {"wgrad-like (K=8192, dy 1e-6 x relu x)", 8192, 1, 1.f, 1e-6f, 1.f, 1048576.f * 4096.f},
  };
  for (const Case& cs : cases) {
    const int K = cs.K;
    std::mt19937 g(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> X(32 * K), W(32 * K);
    for (auto& v : X) { v = nd(g) * cs.xs; if (cs.relu && v < 0) v = 0; }
    for (auto& v : W) v = nd(g) * cs.ws;
    std::vector<double> ref(1024);
    std::vector<float> seq(1024);
    double rms = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double s = 0; float f = 0.f;
        for (int k = 0; k < K; ++k) { s += (double)X[m * K + k] * (double)W[n * K + k]; f = fmaf(X[m * K + k], W[n * K + k], f); }
        ref[m * 32 + n] = s; seq[m * 32 + n] = f; rms += s * s;
      }
    rms = sqrt(rms / 1024);
    printf("== %s  (rms of outputs %.3e)\n", cs.name, rms);
    {
      double mx = 0, sq = 0, mean = 0;
      for (int i = 0; i < 1024; ++i) { double e = ((double)seq[i] - ref[i]) / rms; mx = fmax(mx, fabs(e)); sq += e * e; mean += e; }
      printf("  %-16s max %.3e  rms %.3e  mean %+.3e   (error / rms(out))\n", "cpu f32 fma", mx, sqrt(sq / 1024), mean / 1024);
    }
    float *dX, *dW, *dC;
    hipMalloc(&dX, X.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 7; ++mode) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dX, dW, K, dC, mode, cs.sx, cs.sw);
      std::vector<float> C(1024);
      hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
      double mx = 0, sq = 0, mean = 0;
      for (int i = 0; i < 1024; ++i) { double e = ((double)C[i] - ref[i]) / rms; mx = fmax(mx, fabs(e)); sq += e * e; mean += e; }
      printf("  %-16s max %.3e  rms %.3e  mean %+.3e\n", names[mode], mx, sqrt(sq / 1024), mean / 1024);
    }
    hipFree(dX); hipFree(dW); hipFree(dC);
  }
  return 0;
}
