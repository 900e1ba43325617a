// comm_probe.hip -- stand-in for the local HBM side of a gradient all-reduce (tools/comm_interference_probe.py):
// dst[i] += src[i] over n floats by a FIXED number of work-groups (grid-stride), so that its rate can be throttled to what
// RCCL's few communication work-groups would move while the MFMA kernels of backward own the rest of the chip.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void reduce_like_kernel(float* __restrict__ dst, const float* __restrict__ src, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = ((const float4*)src)[i];
        float4 b = ((float4*)dst)[i];
        b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
        ((float4*)dst)[i] = b;
    }
}

// The same with `lds_bytes` of dynamic LDS per work-group: with 96 KB a work-group cannot share a CU with any of the step's MFMA
// work-groups (each holds > 64 KB of the CU's 160 KB) -- the pessimistic model of a communication kernel that OWNS its CUs, as
// the CTC lattice waves were measured to do (profiles/r04_trace_*): every launch sized to exactly 256 work-groups then needs a
// second round.
__global__ __launch_bounds__(256) void reduce_like_excl_kernel(float* __restrict__ dst, const float* __restrict__ src, long n4) {
    extern __shared__ float hold[];
    hold[threadIdx.x] = (float)blockIdx.x;
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = ((const float4*)src)[i];
        float4 b = ((float4*)dst)[i];
        b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
        ((float4*)dst)[i] = b;
    }
    if (hold[(threadIdx.x + 1) & 255] < 0.f) dst[0] = 0.f;  // (keeps the allocation alive)
}

extern "C" int probe_reduce_like_excl(float* dst, const float* src, long n, int blocks, int lds_bytes, void* stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)reduce_like_excl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(reduce_like_excl_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, dst, src, n / 4);
    return (int)hipGetLastError();
}

extern "C" int probe_reduce_like(float* dst, const float* src, long n, int blocks, void* stream) {
    hipLaunchKernelGGL(reduce_like_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, src, n / 4);
    return (int)hipGetLastError();
}
