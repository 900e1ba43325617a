// hw_probe.hip -- prints the gfx950 semantics this repo's kernels rely on (run on the GPU box; output is evidence).
//   1. ds_read_b64_tr_b16: which element each lane receives from a 4x16 bf16 block (wgrad_tn_bf16.hip)
//   2. global_load_lds_dwordx4: LDS image = wave base + lane*16 (conv_nt_bf16.hip / wgrad_tn_bf16.hip staging)
//   3. v_mfma_f32_16x16x32_bf16 C/D layout with asymmetric operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS __attribute__((address_space(3)))
#define GLB __attribute__((address_space(1)))

__device__ unsigned short f2bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }

__global__ void tr_probe(float* out) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[32 * 128];  // [32 rows][128 cols] bf16, 256 B rows
    for (int i = threadIdx.x; i < 32 * 128; i += 64) tile[i] = f2bf((float)((i / 128) * 16 + (i % 128) % 16));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i16 = lane & 15;
    // group g reads block rows g*8 .. g*8+3, columns 16..31 (col block 1)
    const char* p = (const char*)tile + (g * 8 + (i16 >> 2)) * 256 + 1 * 32 + (i16 & 3) * 8;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS s16x4*)p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = __uint_as_float(((unsigned)(unsigned short)v[j]) << 16);
}

__global__ void glds_probe(const unsigned* src, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned buf[256];
    // lane i copies 16 B from src + (63 - i) * 4 dwords  (reversed source, linear destination expected)
    __builtin_amdgcn_global_load_lds((const GLB void*)(src + (63 - threadIdx.x) * 4), (LDS void*)buf, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = buf[i];
}

__global__ void mfma_probe(float* out) {
    const int lane = threadIdx.x;
    // A[i][k] = (i == k) ? 1 : 0 for k < 16 (identity in the first 16 k), B[k][j] = k * 16 + j  (asymmetric)
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = (lane >> 4) * 8 + e;
        a[e] = (short)f2bf((lane & 15) == k ? 1.f : 0.f);
        b[e] = (short)f2bf(k < 16 ? (float)(k * 16 + (lane & 15)) : 0.f);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

int main() {
    float* d;
    hipMalloc(&d, 64 * 4 * 4);
    std::vector<float> h(256);
    tr_probe<<<1, 64>>>(d);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("== ds_read_b64_tr_b16: lane -> 4 values (value = row_in_tile*16 + col%%16; expect lane i of group g: (g*8+j)*16 + i)\n");
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d: %5.0f %5.0f %5.0f %5.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        for (int j = 0; j < 4; ++j) bad += h[l * 4 + j] != (float)(((l >> 4) * 8 + j) * 16 + (l & 15));
    }
    printf("tr_probe mismatches vs expectation: %d\n", bad);

    unsigned *src, *dst;
    hipMalloc(&src, 1024);
    hipMalloc(&dst, 1024);
    std::vector<unsigned> hs(256), hd(256);
    for (int i = 0; i < 256; ++i) hs[i] = i;
    hipMemcpy(src, hs.data(), 1024, hipMemcpyHostToDevice);
    glds_probe<<<1, 64>>>(src, dst);
    hipMemcpy(hd.data(), dst, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int i = 0; i < 256; ++i) bad += hd[i] != (unsigned)((63 - i / 4) * 4 + i % 4);
    printf("== global_load_lds_dwordx4: lds[0..7] = %u %u %u %u %u %u %u %u ; mismatches vs 'base + lane*16': %d\n", hd[0],
           hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7], bad);

    mfma_probe<<<1, 64>>>(d);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) bad += h[l * 4 + r] != (float)(((l >> 4) * 4 + r) * 16 + (l & 15));
    printf("== mfma 16x16x32 bf16: D[row=(lane>>4)*4+r][col=lane&15] mismatches: %d (lane0: %g %g %g %g)\n", bad, h[0], h[1],
           h[2], h[3]);
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
