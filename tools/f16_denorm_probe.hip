// Hardware probe (round 6): does v_mfma_f32_16x16x32_f16 on gfx950 keep fp16 DENORMAL inputs, or flush them to zero?
// Decides how the lo planes of an fp16 hi + lo split may be stored.
//   hipcc --offload-arch=gfx950 -O2 tools/f16_denorm_probe.hip -o tools/_probe/f16_denorm_probe && tools/_probe/f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* a_vals, float* out, int n) {
    // A = value in every element, B = 1: D[i][j] = 32 * value
    for (int t = 0; t < n; ++t) {
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = a_vals[t]; b[i] = (_Float16)1.0f; }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        if (threadIdx.x == 0) out[t] = acc[0];
    }
}

int main() {
    const int n = 8;
    float want[n];
    _Float16 h[n];
    const float vals[n] = {1.0f, 6.1035156e-05f /*2^-14 min normal*/, 3.0517578e-05f /*2^-15 denormal*/, 9.5367432e-07f /*2^-20*/,
                           5.9604645e-08f /*2^-24 smallest denormal*/, 0.000123f, 3.3e-05f, 1e-06f};
    for (int i = 0; i < n; ++i) { h[i] = (_Float16)vals[i]; want[i] = 32.f * (float)h[i]; }
    _Float16* d_a; float* d_o;
    hipMalloc(&d_a, sizeof(h)); hipMalloc(&d_o, n * sizeof(float));
    hipMemcpy(d_a, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_a, d_o, n);
    float got[n];
    hipMemcpy(got, d_o, sizeof(got), hipMemcpyDeviceToHost);
    int flushed = 0;
    for (int i = 0; i < n; ++i) {
        printf("a = %.9g (fp16 %.9g): mfma sum %.9g, exact %.9g%s\n", vals[i], (float)h[i], got[i], want[i],
               got[i] == want[i] ? "" : "   <-- differs");
        if (want[i] != 0.f && got[i] == 0.f) ++flushed;
    }
    printf("%s\n", flushed ? "fp16 denormal inputs are FLUSHED by v_mfma_f32_16x16x32_f16" : "fp16 denormal inputs are KEPT by v_mfma_f32_16x16x32_f16");
    return 0;
}
