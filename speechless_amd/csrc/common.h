// common.h -- shared device/host helpers for the gfx950 kernels of the speechless hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/speechless_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Element type of the 16-bit MFMA operands of conv_nt_bf16.hip / wgrad_tn_bf16.hip: bf16 -- or, in the translation units that
// build.py compiles a second time with -DSL_ELEM_F16 (conv_nt_f16.o, wgrad_tn_f16.o), fp16: the "f16x3" parity path
// (DESIGN.md section 1).  Tiles move as 16-bit words either way; only the matrix instruction and the plane conversions differ.
typedef _Float16 sl_half8 __attribute__((ext_vector_type(8)));
#if defined(SL_ELEM_F16)
#define SL_MFMA16(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sl_half8, (a)), __builtin_bit_cast(sl_half8, (b)), (c), 0, 0, 0)
#define SL_MFMA32(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sl_half8, (a)), __builtin_bit_cast(sl_half8, (b)), (c), 0, 0, 0)
#else
#define SL_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define SL_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif

#define SL_LDS __attribute__((address_space(3)))
#define SL_GLOBAL __attribute__((address_space(1)))

// thread-local error string, set by sl_set_error (capi.hip)
void sl_set_error(const char* fmt, ...);

#define SL_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            sl_set_error(__VA_ARGS__);          \
            return SL_ERR_INVALID_ARGUMENT;     \
        }                                       \
    } while (0)

// measurement hook (sl_profile_next_kernel, capi.hip): HIP events attached to the MAIN kernel of the next sl_conv1d_nt /
// sl_conv1d_wgrad[_grouped] call (not to its split-K / reduction tail)
void sl_prof_take(hipEvent_t* start, hipEvent_t* stop);
// Launch of a call's MAIN kernel.  When events are armed they are attached to the dispatch itself (hipExtLaunchKernelGGL): they
// carry the kernel's own begin / end timestamps -- the duration rocprofv3 --kernel-trace reports for it.  (Until round 4 the
// events were recorded on the stream in front of and behind the launch, which adds the dispatch gap: 10 % on a 0.2 ms kernel.)
#define SL_LAUNCH_MAIN(kernel, grid, block, lds, s, ...)                                         \
    do {                                                                                         \
        hipEvent_t sl_e0_, sl_e1_;                                                               \
        sl_prof_take(&sl_e0_, &sl_e1_);                                                          \
        if (sl_e0_ != nullptr || sl_e1_ != nullptr)                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, s, sl_e0_, sl_e1_, 0, __VA_ARGS__);  \
        else                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__);                        \
    } while (0)

static inline int sl_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        sl_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return SL_ERR_LAUNCH_FAILED;
    }
    return SL_OK;
}

// fp32 -> bf16 round-to-nearest-even (bit pattern), matches oracle round_to_bf16
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    return (unsigned int)f32_to_bf16_bits(lo) | ((unsigned int)f32_to_bf16_bits(hi) << 16);
}
// the same rounding in one instruction (v_cvt_pk_bf16_f32, gfx950); differs from the software form only on NaN payloads
typedef float sl_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 sl_bf16x2_t __attribute__((ext_vector_type(2)));
typedef short sl_s16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short sl_u16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int pack_bf16x2_hw(float lo, float hi) {
    const sl_f32x2_t v = {lo, hi};
    const sl_bf16x2_t h = __builtin_convertvector(v, sl_bf16x2_t);
    return __builtin_bit_cast(unsigned int, h);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

// fp32 <-> fp16 (round to nearest even, v_cvt_f16_f32; clamped to the largest finite fp16 so that an out-of-range value of the
// f16x3 planes stays a number -- the scales of that path keep everything far inside)
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f) {
    const _Float16 h = (_Float16)fminf(fmaxf(f, -65504.f), 65504.f);
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float f16_bits_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ unsigned int pack_f16x2(float lo, float hi) {
    return (unsigned int)f32_to_f16_bits(lo) | ((unsigned int)f32_to_f16_bits(hi) << 16);
}
__device__ __forceinline__ float f16_lo_to_f32(unsigned int u) { return f16_bits_to_f32((unsigned short)(u & 0xFFFFu)); }
__device__ __forceinline__ float f16_hi_to_f32(unsigned int u) { return f16_bits_to_f32((unsigned short)(u >> 16)); }

// dropout: counter-based generator, one 64-bit mix (splitmix64 finaliser) of (seed, element index) -> 32 uniform bits.  The
// mask of an element depends only on (seed, index), so a step is reproducible from its seed and no state is kept.
__device__ __forceinline__ unsigned int dropout_bits(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned int)(z >> 32);
}

// XCD-aware work-group remap: the dispatcher places block b on XCD b % 8 (speed only, never correctness).
// Gives each XCD a contiguous range of logical ids so that neighbouring tiles share the XCD's private L2.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    // launch grids are padded to a multiple of 8 (xcd_grid); XCD k = bid % 8 takes the contiguous id range
    // [k * per, (k + 1) * per), ids >= total have no work (the caller returns)
    const int per = (total + 7) >> 3;
    return (bid & 7) * per + (bid >> 3);
}
static inline int xcd_grid(int total) { return ((total + 7) >> 3) << 3; }

// How many CUs the launch choosers may count on (sl_set_available_cus, capi.hip; default: all 256).  The MFMA kernels take a
// whole CU per work-group and their grids are sized to fill the chip in whole rounds; a concurrent kernel that OWNS CUs (the
// communication kernels of a multi-GPU run) turns every launch of exactly 256 work-groups into two rounds.
int sl_cus();

// kernels' C++ entry points (called from capi.hip)
int conv_nt_bf16(const void* x, const void* w, const float* bias, const void* mask, void* y, const sl_conv_geom* g,
                 int epilogue, int out_f32, int cfg, void* workspace, size_t workspace_bytes, hipStream_t s);
size_t conv_nt_bf16_workspace_bytes(const sl_conv_geom* g, int cfg);
int conv_nt_f32(const void* x, const void* w, const float* bias, const void* mask, void* y, const sl_conv_geom* g,
                int epilogue, int cfg, hipStream_t s);
bool conv_chain_bf16_supported(const sl_conv_geom* g, int n_layers);
int conv_chain_bf16(const void* x, void* const* ys, const void* const* ws, const float* const* biases,
                    const void* const* masks, const sl_conv_geom* g, int n_layers, int epilogue, hipStream_t s);
int wgrad_split_count(const sl_conv_geom* g, int tile);
int wgrad_tn_bf16(const void* x, const void* gr, float* dw, const sl_conv_geom* g, int cfg, int groups, long x_gs,
                  long g_gs, long dw_gs, float* ws, size_t ws_bytes, hipStream_t s);
size_t wgrad_tn_bf16_workspace_bytes(const sl_conv_geom* g, int cfg, int groups);
int wgrad_tn_f32(const void* x, const void* gr, float* dw, const sl_conv_geom* g, float* ws, int splits, int cfg,
                 hipStream_t s);
int wgrad_f32_tile(const sl_conv_geom* g, int cfg);
bool conv1x1_bwd_bf16_supported(const sl_conv_geom* g, int k_real);
size_t conv1x1_bwd_bf16_workspace_bytes(const sl_conv_geom* g, int cfg);
int conv1x1_bwd_bf16(const void* x, const void* gr, const void* w_dgrad, void* dx, float* dw, const sl_conv_geom* g,
                     int epilogue, int cfg, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
size_t wgrad_multi_bf16_workspace_bytes(const sl_wgrad_job* jobs, int n_jobs);
int wgrad_multi_bf16(const sl_wgrad_job* jobs, int n_jobs, void* ws, size_t ws_bytes, hipStream_t s);
// the same entry points of the -DSL_ELEM_F16 translation units (fp16 operands: SL_F16)
int conv_nt_f16(const void* x, const void* w, const float* bias, const void* mask, void* y, const sl_conv_geom* g,
                int epilogue, int out_f32, int cfg, void* workspace, size_t workspace_bytes, hipStream_t s);
size_t conv_nt_f16_workspace_bytes(const sl_conv_geom* g, int cfg);
int wgrad_tn_f16(const void* x, const void* gr, float* dw, const sl_conv_geom* g, int cfg, int groups, long x_gs,
                 long g_gs, long dw_gs, float* ws, size_t ws_bytes, hipStream_t s);
size_t wgrad_tn_f16_workspace_bytes(const sl_conv_geom* g, int cfg, int groups);
size_t wgrad_multi_f16_workspace_bytes(const sl_wgrad_job* jobs, int n_jobs);
int wgrad_multi_f16(const sl_wgrad_job* jobs, int n_jobs, void* ws, size_t ws_bytes, hipStream_t s);
