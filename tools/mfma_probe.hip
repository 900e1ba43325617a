// MFMA throughput ceiling on this part (hardware probe, not part of the product path): register-only loops of
// v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16 with 128 accumulator registers per wave, one and two waves per
// SIMD, every CU busy.   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  if (s == 12345.f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 256 * 512 * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int iters = 20000;  // x 32 MFMAs of 16x16x32 (or 16 of 32x32x16) = 524288 flop per lane-independent wave iteration
  for (int threads : {256, 512}) {
    for (int shape : {16, 32}) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        if (shape == 16) mfma_loop<16><<<256, threads>>>(out, iters); else mfma_loop<32><<<256, threads>>>(out, iters);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      const double flops = 256.0 * (threads / 64) * iters * 32.0 * 16 * 16 * 32 * 2;
      printf("%s  %d waves per SIMD: %8.2f ms  %7.1f TFLOP/s\n", shape == 16 ? "16x16x32" : "32x32x16", threads / 256, best,
             flops / best / 1e9);
    }
  }
  return 0;
}
