// FETCH_SIZE calibration on known byte counts, in the access patterns of this repo's kernels (gfx950).
//
// MI355X_MICROARCH.md (HBM): "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The halo-patch
// convolution (csrc/conv3x3_patch.hip) does NOT stream: a wave-instruction of `buffer_load ... lds` fetches 16 pixel
// rows x 64 B (4 lanes x 16 B per pixel = one 32-channel slice), consecutive pixels P bytes apart (P = bytes per pixel:
// 256 B for 64 logical bf16x3 channels ... 2048 B for 512), and the other slices of the same pixels follow a few
// microseconds later from the same workgroup.  Every kernel below reads its whole buffer EXACTLY ONCE, so the true
// number of bytes leaving HBM (or the Infinity Cache: the counter sits at the L2's memory side) is the buffer size:
//
//   stream        1 KiB contiguous per wave-instruction (the guide's calibrated case; expect 0.5)
//   seg64_P<P>    the patch kernel's pattern: 16 rows x 64 B, row pitch P, slices of a 16-row group back to back
//   seg64far_P<P> the same, but a workgroup finishes slice s of ALL its rows before slice s+1 (a whole K-loop body
//                 apart, as in the real kernel)
//
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o f --output-format csv -- ./fetch_size_calibration
// then   python3 tools/experiments/fetch_size_summary.py out   (FETCH_SIZE [KB] * 1024 / bytes per launch)
//
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_size_calibration fetch_size_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void bufload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(l), 16, (int)voff, (int)soff, 0, 0);
}

// 256 threads; every wave-instruction = 1 KiB contiguous
__global__ void __launch_bounds__(256) k_stream(const char* x, size_t bytes, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_wg = bytes / gridDim.x;
  const char* base = x + (size_t)blockIdx.x * per_wg;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)per_wg, 0x00020000);
  for (unsigned off = wave * 1024u; off < per_wg; off += 4096u) bufload16(r, off + lane * 16u, 0, smem + wave * 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[5] == 77 && smem[1029] == 78 && smem[2222] == 1) atomicAdd(sink, 1u);
}

// the patch kernel's pattern.  rows = pixels of P bytes; a wave-instruction takes the 64-byte slice s of 16
// consecutive rows.  FAR = false: for each 16-row group all P/64 slices back to back; FAR = true: slice-major over the
// workgroup's whole row range (rows_per_wg rows), so the two 64-byte halves of a 128-byte line are fetched far apart
template <bool FAR, int P>
__global__ void __launch_bounds__(256) k_seg64(const char* x, size_t bytes, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t per_wg = bytes / gridDim.x;
  const char* base = x + (size_t)blockIdx.x * per_wg;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)per_wg, 0x00020000);
  const int rows = (int)(per_wg / P), nslice = P / 64;
  const unsigned lane_off = (unsigned)(lane >> 2) * (unsigned)P + (unsigned)(lane & 3) * 16u;
  if (FAR) {
    for (int s = 0; s < nslice; ++s)
      for (int g = wave; g * 16 < rows; g += 4) bufload16(r, (unsigned)g * 16u * P + lane_off, (unsigned)s * 64u, smem + wave * 1024);
  } else {
    for (int g = wave; g * 16 < rows; g += 4)
      for (int s = 0; s < nslice; ++s) bufload16(r, (unsigned)g * 16u * P + lane_off, (unsigned)s * 64u, smem + wave * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[5] == 77 && smem[1029] == 78 && smem[2222] == 1) atomicAdd(sink, 1u);
}

// plain register loads, 16 B per lane, contiguous (global_load_dwordx4): the guide's other calibrated form
__global__ void __launch_bounds__(256) k_stream_reg(const uint4* x, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    uint4 v = x[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) atomicAdd(sink, 1u);
}

int main() {
  const size_t bytes = (size_t)768 << 20;            // 768 MiB: 3 x the Infinity Cache, 24 x the L2s
  const int grid = 2048;                             // 384 KiB per workgroup (rows of every P divide it)
  char* x;
  unsigned* sink;
  CK(hipMalloc(&x, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(sink, 0, 4));
  // random-ish content (not zeros: DVFS and any zero-compression path out of the picture)
  unsigned* h = (unsigned*)malloc(bytes);
  unsigned v = 12345u;
  for (size_t i = 0; i < bytes / 4; ++i) { v = v * 1664525u + 1013904223u; h[i] = v; }
  CK(hipMemcpy(x, h, bytes, hipMemcpyHostToDevice));
  free(h);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, x, bytes, sink);
    hipLaunchKernelGGL(k_stream_reg, dim3(grid), dim3(256), 0, 0, (const uint4*)x, bytes / 16, sink);
#define SEG(P) hipLaunchKernelGGL((k_seg64<false, P>), dim3(grid), dim3(256), 0, 0, x, bytes, sink); \
               hipLaunchKernelGGL((k_seg64<true, P>), dim3(grid), dim3(256), 0, 0, x, bytes, sink);
    SEG(128) SEG(256) SEG(512) SEG(1024) SEG(2048)
  }
  CK(hipDeviceSynchronize());
  unsigned s;
  CK(hipMemcpy(&s, sink, 4, hipMemcpyDeviceToHost));
  printf("bytes_per_launch %zu sink %u\n", bytes, s);
  return 0;
}
