// misc.hip -- HBM-bound helpers of the hot path: weight packing, input packing, bias gradient, Keras-2.0 Adam.
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ T cvt_out(float v);
template <>
__device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <>
__device__ __forceinline__ unsigned short cvt_out<unsigned short>(float v) { return f32_to_bf16_bits(v); }

// master [k][cin][cout] fp32 -> w_fwd [cout][k][cin] (32x32 LDS transpose per tap) and w_dgrad [cin][k-1-tap][cout]
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ wm, T* __restrict__ wf,
                                                           T* __restrict__ wd, int k, int cin, int cout) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int ci0 = blockIdx.y * 32;
    const int co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + ty + r * 8;
        const float v = wm[((long)tap * cin + ci) * cout + co0 + tx];
        tile[ty + r * 8][tx] = v;
        if (wd) wd[((long)ci * k + (k - 1 - tap)) * cout + co0 + tx] = cvt_out<T>(v);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + ty + r * 8;
        wf[((long)co * k + tap) * cin + ci0 + tx] = cvt_out<T>(tile[tx][ty + r * 8]);
    }
}

// one wave per input row (b, t): 64 consecutive bins per load (coalesced for any bin count -- 257 as well as 128), all
// loads of two rows in flight before the first store; 32-bit index arithmetic, one division per row
template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ src, T* __restrict__ dst, int t_in,
                                                         int f, int dst_row0, int dst_rs, long dst_bs, long rows,
                                                         int ones_ch) {
    constexpr int ROWS = 2, COLS = 8;  // per wave and pass; bins beyond 64 * COLS go through the tail loop
    const int lane = threadIdx.x & 63;
    const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    float v[ROWS][COLS];
    T* dp[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const long r = r0 + q < rows ? r0 + q : rows - 1;  // (a duplicate of the last row: same values, same place)
        const long b = r / t_in;
        const int t = (int)(r - b * t_in);
        const float* sp = src + r * f;
        dp[q] = dst + b * dst_bs + (long)(dst_row0 + t) * dst_rs;
#pragma unroll
        for (int k = 0; k < COLS; ++k) v[q][k] = lane + 64 * k < f ? sp[lane + 64 * k] : 0.f;
        for (int c = lane + 64 * COLS; c < f; c += 64) dp[q][c] = cvt_out<T>(sp[c]);
        if (ones_ch >= 0 && lane == 0) dp[q][ones_ch] = cvt_out<T>(1.f);  // the input's ones channel (sl_pack_input_ones)
    }
#pragma unroll
    for (int q = 0; q < ROWS; ++q)
#pragma unroll
        for (int k = 0; k < COLS; ++k)
            if (lane + 64 * k < f) dp[q][lane + 64 * k] = cvt_out<T>(v[q][k]);
}

// Raw-wave front layer (reference net.py:310-312: Conv1D(250 filters, 250 taps, stride 160, SAME) over the samples): the sample
// windows of the output frames, one row of k * cin columns per frame, so that the layer is a 1 x 1 GEMM over them for the
// ordinary kernels (forward: sl_conv1d_nt, weight gradient: sl_conv1d_wgrad).  3 GFLOP per 32 x 8 s of audio: the window matrix
// (1.6 x the samples at stride 160 / 250 taps) is the cheap part of a layer that is nothing next to the stack behind it.
//   dst[b][t][j * cin + c] = audio[b][t * stride + j - pad_left][c]  (0 outside [0, t_in)),  columns >= k * cin: 0
template <typename T>
__global__ __launch_bounds__(256) void wave_frames_kernel(const float* __restrict__ audio, T* __restrict__ dst, int t_in, int cin,
                                                          int k, int stride, int pad_left, int t_out, int dst_rs, long dst_bs) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)t_out * dst_rs) return;
    const int t = (int)(i / dst_rs), col = (int)(i - (long)t * dst_rs);
    float v = 0.f;
    if (col < k * cin) {
        const int j = col / cin, c = col - j * cin;
        const long s = (long)t * stride + j - pad_left;
        if (s >= 0 && s < t_in) v = audio[((long)b * t_in + s) * cin + c];
    }
    dst[(long)b * dst_bs + i] = cvt_out<T>(v);
}

// stage 1: partial[b][co] = sum_t g[b][row0+t][co]; block = 256 threads = 16 column-groups(8 ch) x 16 row lanes
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const T* __restrict__ g, float* __restrict__ partial,
                                                                int t_out, int cout, int g_row0, int g_rs, long g_bs) {
    __shared__ float red[16][129];
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 128;
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    const T* base = g + (long)b * g_bs + (long)g_row0 * g_rs + c0 + cg * 8;
    for (int t = rl; t < t_out; t += 16) {
        const T* p = base + (long)t * g_rs;
        if (sizeof(T) == 2) {
            // four independent 16-B loads in flight per lane (rows t, t+16, t+32, t+48) -> HBM latency is overlapped
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = (t + 16 * u < t_out) ? *(const u32x4*)(p + (long)(16 * u) * g_rs) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] += __uint_as_float(v[u][i] << 16);
                    acc[2 * i + 1] += __uint_as_float(v[u][i] & 0xFFFF0000u);
                }
            t += 48;
        } else {
            const f32x4 v0 = *(const f32x4*)p;
            const f32x4 v1 = *(const f32x4*)((const float*)p + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] += v0[i];
                acc[4 + i] += v1[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rl][cg * 8 + i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 128) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
        partial[(long)b * cout + c0 + threadIdx.x] = s;
    }
}

__global__ void bias_grad_final_kernel(const float* __restrict__ partial, float* __restrict__ db, int batch, int cout) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cout) return;
    // the same sum in the same order, eight loads in flight at a time: one dependent load per addition made this launch a chain of
    // `batch` cache round trips (9.0 us at batch 32 in the step's kernel trace, profiles/r05_trace_config3.txt)
    float s = 0.f;
    for (int b0 = 0; b0 < batch; b0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (b0 + j < batch) ? partial[(long)(b0 + j) * cout + c] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (b0 + j < batch) s += v[j];
    }
    db[c] = s;
}

// ONE definition of the element update for every Adam kernel of this file, with the contraction pinned: left to the
// compiler, `b1 * m + (1 - b1) * g` becomes fma(b1, m, (1 - b1) * g) in one kernel and fma(1 - b1, g, b1 * m) in another,
// and the plain elementwise kernel (the sharded optimizer's) would differ from the fused Adam + repack kernel in the
// last bit -- data-parallel runs with and without the sharded optimizer are held to bit-identical weights.
__device__ __forceinline__ void adam_element(float& p, const float g, float& m, float& v, const float lr_t, const float b1,
                                             const float b2, const float eps) {
#pragma clang fp contract(off)
    const float gm = (1.f - b1) * g;
    const float gv = (1.f - b2) * g * g;
    m = __builtin_fmaf(b1, m, gm);
    v = __builtin_fmaf(b2, v, gv);
    const float step = lr_t * m / (sqrtf(v) + eps);
    p = p - step;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n4, float lr_t, float b1, float b2, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 gv = ((const f32x4*)g)[i];
    f32x4 mv = ((f32x4*)m)[i];
    f32x4 vv = ((f32x4*)v)[i];
    f32x4 pv = ((f32x4*)p)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float pj = pv[j], mj = mv[j], vj = vv[j];
        adam_element(pj, gv[j], mj, vj, lr_t, b1, b2, eps);
        pv[j] = pj, mv[j] = mj, vv[j] = vj;
    }
    ((f32x4*)m)[i] = mv;
    ((f32x4*)v)[i] = vv;
    ((f32x4*)p)[i] = pv;
}

// Fused Adam + operand repack for ONE layer: a single pass over the layer's fp32 master weights / gradient / moments
// [k][cin][cout] (+ the bias block that follows them) that also emits both device operand layouts
//   w_fwd [cout][k][cin]   (32x64 tile transposed through LDS)   and   w_dgrad[cin][k-1-tap][cout]
// 4 fp32 reads + 3 fp32 writes + 2 narrow writes per parameter instead of Adam (4r+3w) followed by pack (1r+2w).
// grid (cout/64, cin/32, k + 1): z == k is the bias block (only y == 0 works there).
// PLANES = 3 (bf16x3, T = unsigned short): the operand rows are [w_hi | w_hi | w_lo], w_hi = bf16(w), w_lo = bf16(w - w_hi).
// FMT (PLANES = 3 only): 0 = bf16 planes, 1 = fp16 planes of wscale * w (f16x3: wscale a power of two, see split3.hip)
template <int FMT>
__device__ __forceinline__ u32x2 pack_hi4(float a0, float a1, float a2, float a3) {
    if (FMT == 1) return (u32x2){pack_f16x2(a0, a1), pack_f16x2(a2, a3)};
    return (u32x2){pack_bf16x2(a0, a1), pack_bf16x2(a2, a3)};
}
template <int FMT>
__device__ __forceinline__ u32x2 pack_lo4(float a0, float a1, float a2, float a3) {
    if (FMT == 1) {
        auto lo = [](float v) { return v - f16_bits_to_f32(f32_to_f16_bits(v)); };
        return (u32x2){pack_f16x2(lo(a0), lo(a1)), pack_f16x2(lo(a2), lo(a3))};
    }
    auto lo = [](float v) { return v - bf16_bits_to_f32(f32_to_bf16_bits(v)); };
    return (u32x2){pack_bf16x2(lo(a0), lo(a1)), pack_bf16x2(lo(a2), lo(a3))};
}
template <typename T, bool ADAM = true, int PLANES = 1, int FMT = 0>
__device__ __forceinline__ void adam_pack_block(float (&tile)[32][65], float* __restrict__ p, const float* __restrict__ g,
                                                float* __restrict__ m, float* __restrict__ v, T* __restrict__ wf,
                                                T* __restrict__ wd, int k, int cin, int cout, float lr_t, float b1,
                                                float b2, float eps, int bx, int by, int bz, float wscale = 1.f) {
    const int tap = bz;
    const int co0 = bx * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    auto adam4 = [&](long idx) {
        if (!ADAM) return *(const f32x4*)(p + idx);  // pack only: the masters are already up to date
        const f32x4 gv = *(const f32x4*)(g + idx);
        f32x4 mv = *(f32x4*)(m + idx), vv = *(f32x4*)(v + idx), pv = *(f32x4*)(p + idx);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = pv[j], mj = mv[j], vj = vv[j];
            adam_element(pj, gv[j], mj, vj, lr_t, b1, b2, eps);
            pv[j] = pj, mv[j] = mj, vv[j] = vj;
        }
        *(f32x4*)(m + idx) = mv;
        *(f32x4*)(v + idx) = vv;
        *(f32x4*)(p + idx) = pv;
        return pv;
    };
    if (tap == k) {  // bias block: cout floats right behind the weights
        if (by == 0 && ty == 0) adam4((long)k * cin * cout + co0 + tx * 4);
        return;
    }
    const int ci0 = by * 32;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int cil = ty + r * 16;
        const long idx = ((long)tap * cin + ci0 + cil) * cout + co0 + tx * 4;
        f32x4 pv = adam4(idx);
        if (FMT == 1) pv = pv * wscale;  // (the operand copies hold wscale * w; the masters were written by adam4)
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[cil][tx * 4 + j] = pv[j];
        if (wd) {
            T* o = wd + ((long)(ci0 + cil) * k + (k - 1 - tap)) * (PLANES * cout) + co0 + tx * 4;
            if (PLANES == 3) {
                const u32x2 h = pack_hi4<FMT>(pv[0], pv[1], pv[2], pv[3]);
                *(u32x2*)o = h;
                *(u32x2*)(o + cout) = h;
                *(u32x2*)(o + 2 * cout) = pack_lo4<FMT>(pv[0], pv[1], pv[2], pv[3]);
            } else if (sizeof(T) == 2) {
                *(u32x2*)o = (u32x2){pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
            } else {
                *(f32x4*)o = pv;
            }
        }
    }
    __syncthreads();
    // transposed store: 64 co rows x 32 ci; thread -> (co = threadIdx/8 + 32*r, ci4 = (threadIdx%8)*4)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int col = (threadIdx.x >> 3) + r * 32;
        const int ci4 = (threadIdx.x & 7) * 4;
        T* o = wf + ((long)(co0 + col) * k + tap) * (PLANES * cin) + ci0 + ci4;
        const float a0 = tile[ci4][col], a1 = tile[ci4 + 1][col], a2 = tile[ci4 + 2][col], a3 = tile[ci4 + 3][col];
        if (PLANES == 3) {
            const u32x2 h = pack_hi4<FMT>(a0, a1, a2, a3);
            *(u32x2*)o = h;
            *(u32x2*)(o + cin) = h;
            *(u32x2*)(o + 2 * cin) = pack_lo4<FMT>(a0, a1, a2, a3);
        } else if (sizeof(T) == 2) {
            *(u32x2*)o = (u32x2){pack_bf16x2(a0, a1), pack_bf16x2(a2, a3)};
        } else {
            *(f32x4*)o = (f32x4){a0, a1, a2, a3};
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adam_pack_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, T* __restrict__ wf,
                                                        T* __restrict__ wd, int k, int cin, int cout, float lr_t, float b1,
                                                        float b2, float eps) {
    __shared__ float tile[32][65];
    adam_pack_block<T>(tile, p, g, m, v, wf, wd, k, cin, cout, lr_t, b1, b2, eps, blockIdx.x, blockIdx.y, blockIdx.z);
}

// every trainable layer in ONE launch: the small layers' updates (0.46 M parameters each, 8 us per launch of pure
// latency) ride along with the big ones.  blockIdx.x is a linear block id; the table maps it to (layer, x, y, z).
struct AdamTable {
    int n;
    int block_begin[SL_ADAM_MAX_LAYERS + 1];
    long offset[SL_ADAM_MAX_LAYERS];  // first weight of the layer in the flat fp32 buffers
    void* wf[SL_ADAM_MAX_LAYERS];
    void* wd[SL_ADAM_MAX_LAYERS];
    int k[SL_ADAM_MAX_LAYERS], cin[SL_ADAM_MAX_LAYERS], cout[SL_ADAM_MAX_LAYERS];
};

template <typename T, bool ADAM = true, int PLANES = 1, int FMT = 0>
__global__ __launch_bounds__(256) void adam_pack_multi_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ m, float* __restrict__ v, AdamTable t,
                                                              float lr_t, float b1, float b2, float eps, float wscale = 1.f) {
    __shared__ float tile[32][65];
    int layer = 0;
    while (layer + 1 < t.n && (int)blockIdx.x >= t.block_begin[layer + 1]) ++layer;
    const int local = blockIdx.x - t.block_begin[layer];
    const int k = t.k[layer], cin = t.cin[layer], cout = t.cout[layer];
    const int nx = cout / 64, ny = cin / 32;
    const int bx = local % nx, by = (local / nx) % ny, bz = local / (nx * ny);
    const long off = t.offset[layer];
    if (!ADAM && bz == k) return;  // (the bias block has no operand copy)
    adam_pack_block<T, ADAM, PLANES, FMT>(tile, p + off, ADAM ? g + off : nullptr, ADAM ? m + off : nullptr,
                                          ADAM ? v + off : nullptr, (T*)t.wf[layer], (T*)t.wd[layer], k, cin, cout, lr_t, b1, b2,
                                          eps, bx, by, bz, wscale);
}

}  // namespace

static int adam_table_from(const sl_adam_layer* layers, int n_layers, const char* who, AdamTable* t, int* blocks_out) {
    t->n = n_layers;
    int blocks = 0;
    for (int i = 0; i < n_layers; ++i) {
        const sl_adam_layer& L = layers[i];
        SL_CHECK_ARG(L.w_fwd && L.k > 0 && L.cin_pad > 0 && L.cout_pad > 0 && L.cin_pad % 32 == 0 && L.cout_pad % 64 == 0 &&
                         L.offset >= 0 && L.offset % 4 == 0,
                     "%s: layer %d: need w_fwd, cin_pad %% 32 == 0, cout_pad %% 64 == 0, offset %% 4 == 0", who, i);
        t->block_begin[i] = blocks;
        t->offset[i] = L.offset;
        t->wf[i] = L.w_fwd;
        t->wd[i] = L.w_dgrad;
        t->k[i] = L.k;
        t->cin[i] = L.cin_pad;
        t->cout[i] = L.cout_pad;
        blocks += (L.cout_pad / 64) * (L.cin_pad / 32) * (L.k + 1);
    }
    t->block_begin[n_layers] = blocks;
    *blocks_out = blocks;
    return SL_OK;
}

extern "C" int sl_adam_pack_layers(float* param, const float* grad, float* m, float* v, const sl_adam_layer* layers,
                                   int n_layers, int dtype, int step, float lr, float beta1, float beta2, float eps,
                                   void* stream) {
    SL_CHECK_ARG(param && grad && m && v && layers, "sl_adam_pack_layers: null pointer");
    SL_CHECK_ARG(n_layers >= 1 && n_layers <= SL_ADAM_MAX_LAYERS, "sl_adam_pack_layers: 1..%d layers per call",
                 SL_ADAM_MAX_LAYERS);
    SL_CHECK_ARG(step >= 1 && (dtype == SL_BF16 || dtype == SL_F32), "sl_adam_pack_layers: bad step or dtype");
    AdamTable t;
    int blocks = 0;
    const int rc = adam_table_from(layers, n_layers, "sl_adam_pack_layers", &t, &blocks);
    if (rc != SL_OK) return rc;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    if (dtype == SL_BF16)
        hipLaunchKernelGGL((adam_pack_multi_kernel<unsigned short>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, param,
                           grad, m, v, t, (float)lr_t, beta1, beta2, eps);
    else
        hipLaunchKernelGGL((adam_pack_multi_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, m,
                           v, t, (float)lr_t, beta1, beta2, eps);
    return sl_check_launch("sl_adam_pack_layers");
}

extern "C" int sl_split3_adam_pack_layers(float* param, const float* grad, float* m, float* v, const sl_adam_layer* layers,
                                          int n_layers, int step, float lr, float beta1, float beta2, float eps, void* stream) {
    SL_CHECK_ARG(param && grad && m && v && layers, "sl_split3_adam_pack_layers: null pointer");
    SL_CHECK_ARG(n_layers >= 1 && n_layers <= SL_ADAM_MAX_LAYERS && step >= 1, "sl_split3_adam_pack_layers: 1..%d layers per call",
                 SL_ADAM_MAX_LAYERS);
    AdamTable t;
    int blocks = 0;
    const int rc = adam_table_from(layers, n_layers, "sl_split3_adam_pack_layers", &t, &blocks);
    if (rc != SL_OK) return rc;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    hipLaunchKernelGGL((adam_pack_multi_kernel<unsigned short, true, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, param,
                       grad, m, v, t, (float)lr_t, beta1, beta2, eps, 1.f);
    return sl_check_launch("sl_split3_adam_pack_layers");
}

extern "C" int sl_splitf16_adam_pack_layers(float* param, const float* grad, float* m, float* v, const sl_adam_layer* layers,
                                            int n_layers, int step, float lr, float beta1, float beta2, float eps,
                                            float w_scale, void* stream) {
    SL_CHECK_ARG(param && grad && m && v && layers && w_scale > 0.f, "sl_splitf16_adam_pack_layers: null pointer or bad scale");
    SL_CHECK_ARG(n_layers >= 1 && n_layers <= SL_ADAM_MAX_LAYERS && step >= 1, "sl_splitf16_adam_pack_layers: 1..%d layers per call",
                 SL_ADAM_MAX_LAYERS);
    AdamTable t;
    int blocks = 0;
    const int rc = adam_table_from(layers, n_layers, "sl_splitf16_adam_pack_layers", &t, &blocks);
    if (rc != SL_OK) return rc;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    hipLaunchKernelGGL((adam_pack_multi_kernel<unsigned short, true, 3, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       param, grad, m, v, t, (float)lr_t, beta1, beta2, eps, w_scale);
    return sl_check_launch("sl_splitf16_adam_pack_layers");
}

extern "C" int sl_pack_layers(const float* param, const sl_adam_layer* layers, int n_layers, int dtype, void* stream) {
    SL_CHECK_ARG(param && layers, "sl_pack_layers: null pointer");
    SL_CHECK_ARG(n_layers >= 1 && n_layers <= SL_ADAM_MAX_LAYERS, "sl_pack_layers: 1..%d layers per call",
                 SL_ADAM_MAX_LAYERS);
    SL_CHECK_ARG(dtype == SL_BF16 || dtype == SL_F32, "sl_pack_layers: bad dtype");
    AdamTable t;
    int blocks = 0;
    const int rc = adam_table_from(layers, n_layers, "sl_pack_layers", &t, &blocks);
    if (rc != SL_OK) return rc;
    float* p = const_cast<float*>(param);  // (read only in the pack-only instantiation)
    if (dtype == SL_BF16)
        hipLaunchKernelGGL((adam_pack_multi_kernel<unsigned short, false>), dim3(blocks), dim3(256), 0,
                           (hipStream_t)stream, p, nullptr, nullptr, nullptr, t, 0.f, 0.f, 0.f, 0.f);
    else
        hipLaunchKernelGGL((adam_pack_multi_kernel<float, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p,
                           nullptr, nullptr, nullptr, t, 0.f, 0.f, 0.f, 0.f);
    return sl_check_launch("sl_pack_layers");
}

namespace {
struct BgwTable {
    sl_bgw_layer l[SL_BGW_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void bias_grad_from_wgrad_kernel(float* __restrict__ grads, BgwTable t, int copy) {
    const sl_bgw_layer L = t.l[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= L.cout_pad) return;
    float* row = grads + L.w_off + (long)(L.cin_pad - 1) * L.cout_pad + c;  // tap 0; a tap further = cin_pad * cout_pad
    const long tap_stride = (long)L.cin_pad * L.cout_pad;
    if (copy) grads[L.b_off + c] = row[(long)L.tap * tap_stride];
    for (int k = 0; k < L.k; ++k) row[(long)k * tap_stride] = 0.f;
}
}  // namespace

extern "C" int sl_bias_grad_from_wgrad(float* grads, const sl_bgw_layer* layers, int n_layers, int copy, void* stream) {
    SL_CHECK_ARG(grads != nullptr && layers != nullptr, "sl_bias_grad_from_wgrad: null pointer");
    SL_CHECK_ARG(n_layers >= 1 && n_layers <= SL_BGW_MAX_LAYERS, "sl_bias_grad_from_wgrad: 1..%d layers per call",
                 SL_BGW_MAX_LAYERS);
    BgwTable t;
    int widest = 0;
    for (int i = 0; i < n_layers; ++i) {
        SL_CHECK_ARG(layers[i].k >= 1 && layers[i].tap >= 0 && layers[i].tap < layers[i].k && layers[i].cin_pad >= 1 &&
                         layers[i].cout_pad >= 1,
                     "sl_bias_grad_from_wgrad: bad layer %d", i);
        t.l[i] = layers[i];
        if (layers[i].cout_pad > widest) widest = layers[i].cout_pad;
    }
    hipLaunchKernelGGL(bias_grad_from_wgrad_kernel, dim3((widest + 255) / 256, n_layers), dim3(256), 0,
                       (hipStream_t)stream, grads, t, copy);
    return sl_check_launch("sl_bias_grad_from_wgrad");
}

extern "C" int sl_adam_pack_layer(float* param, const float* grad, float* m, float* v, void* w_fwd, void* w_dgrad, int k,
                                  int cin_pad, int cout_pad, int dtype, int step, float lr, float beta1, float beta2,
                                  float eps, void* stream) {
    SL_CHECK_ARG(param && grad && m && v && w_fwd, "sl_adam_pack_layer: null pointer");
    SL_CHECK_ARG(k > 0 && cin_pad > 0 && cout_pad > 0 && cin_pad % 32 == 0 && cout_pad % 64 == 0 && step >= 1,
                 "sl_adam_pack_layer: need cin_pad %% 32 == 0, cout_pad %% 64 == 0, step >= 1");
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    dim3 grid(cout_pad / 64, cin_pad / 32, k + 1);
    if (dtype == SL_BF16)
        hipLaunchKernelGGL((adam_pack_kernel<unsigned short>), grid, dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                           (unsigned short*)w_fwd, (unsigned short*)w_dgrad, k, cin_pad, cout_pad, (float)lr_t, beta1,
                           beta2, eps);
    else
        hipLaunchKernelGGL((adam_pack_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, param, grad, m, v,
                           (float*)w_fwd, (float*)w_dgrad, k, cin_pad, cout_pad, (float)lr_t, beta1, beta2, eps);
    return sl_check_launch("sl_adam_pack_layer");
}

extern "C" int sl_pack_weights(const float* w_master, void* w_fwd, void* w_dgrad, int k, int cin_pad, int cout_pad,
                               int dtype, void* stream) {
    SL_CHECK_ARG(k > 0 && cin_pad > 0 && cout_pad > 0 && cin_pad % 32 == 0 && cout_pad % 32 == 0,
                 "sl_pack_weights: channel counts must be multiples of 32");
    SL_CHECK_ARG(w_master && w_fwd, "sl_pack_weights: null pointer");
    dim3 grid(cout_pad / 32, cin_pad / 32, k);
    if (dtype == SL_BF16)
        hipLaunchKernelGGL((pack_weights_kernel<unsigned short>), grid, dim3(256), 0, (hipStream_t)stream, w_master,
                           (unsigned short*)w_fwd, (unsigned short*)w_dgrad, k, cin_pad, cout_pad);
    else
        hipLaunchKernelGGL((pack_weights_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, w_master,
                           (float*)w_fwd, (float*)w_dgrad, k, cin_pad, cout_pad);
    return sl_check_launch("sl_pack_weights");
}

extern "C" int sl_wave_frames(const float* audio, void* dst, int batch, int t_in, int cin, int k, int stride, int pad_left,
                              int t_out, int dst_row_stride, int64_t dst_batch_stride, int dtype, void* stream) {
    SL_CHECK_ARG(audio && dst && batch > 0 && t_in > 0 && cin > 0 && k > 0 && stride > 0 && pad_left >= 0 && t_out > 0,
                 "sl_wave_frames: bad sizes");
    SL_CHECK_ARG(dst_row_stride >= k * cin && dst_batch_stride >= (int64_t)t_out * dst_row_stride,
                 "sl_wave_frames: a row holds k * cin = %d columns", k * cin);
    SL_CHECK_ARG(dtype == SL_BF16 || dtype == SL_F32, "sl_wave_frames: unknown dtype %d", dtype);
    const long n = (long)t_out * dst_row_stride;
    const dim3 grid((unsigned)((n + 255) / 256), batch);
    if (dtype == SL_BF16)
        hipLaunchKernelGGL((wave_frames_kernel<unsigned short>), grid, dim3(256), 0, (hipStream_t)stream, audio,
                           (unsigned short*)dst, t_in, cin, k, stride, pad_left, t_out, dst_row_stride, (long)dst_batch_stride);
    else
        hipLaunchKernelGGL((wave_frames_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, audio, (float*)dst, t_in, cin,
                           k, stride, pad_left, t_out, dst_row_stride, (long)dst_batch_stride);
    return sl_check_launch("sl_wave_frames");
}

extern "C" int sl_pack_input(const float* src, void* dst, int batch, int t_in, int f, int dst_row0, int dst_row_stride,
                             int64_t dst_batch_stride, int dtype, void* stream) {
    return sl_pack_input_ones(src, dst, batch, t_in, f, dst_row0, dst_row_stride, dst_batch_stride, -1, dtype, stream);
}

extern "C" int sl_pack_input_ones(const float* src, void* dst, int batch, int t_in, int f, int dst_row0, int dst_row_stride,
                                  int64_t dst_batch_stride, int ones_channel, int dtype, void* stream) {
    SL_CHECK_ARG(batch > 0 && t_in > 0 && f > 0 && dst_row_stride >= f, "sl_pack_input: bad sizes");
    SL_CHECK_ARG(ones_channel < 0 || (ones_channel >= f && ones_channel < dst_row_stride),
                 "sl_pack_input_ones: the ones channel must be a padding channel (f <= c < dst_row_stride)");
    const long total = (long)batch * t_in;  // rows; 8 per work-group
    const unsigned grid = (unsigned)((total + 7) / 8);
    if (dtype == SL_BF16)
        hipLaunchKernelGGL((pack_input_kernel<unsigned short>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src,
                           (unsigned short*)dst, t_in, f, dst_row0, dst_row_stride, (long)dst_batch_stride, total, ones_channel);
    else
        hipLaunchKernelGGL((pack_input_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst,
                           t_in, f, dst_row0, dst_row_stride, (long)dst_batch_stride, total, ones_channel);
    return sl_check_launch("sl_pack_input");
}

extern "C" size_t sl_bias_grad_workspace_bytes(const sl_conv_geom* g) {
    if (!g) return 0;
    return (size_t)g->batch * g->cout * sizeof(float);
}

extern "C" int sl_bias_grad(const void* g, float* db, const sl_conv_geom* geom, int dtype, void* workspace,
                            size_t workspace_bytes, void* stream) {
    SL_CHECK_ARG(g && db && geom && workspace, "sl_bias_grad: null pointer");
    SL_CHECK_ARG(geom->cout % 128 == 0, "sl_bias_grad: cout must be a multiple of 128");
    if (workspace_bytes < sl_bias_grad_workspace_bytes(geom)) {
        sl_set_error("sl_bias_grad: workspace too small");
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(geom->cout / 128, geom->batch);
    if (dtype == SL_BF16)
        hipLaunchKernelGGL((bias_grad_partial_kernel<unsigned short>), grid, dim3(256), 0, s, (const unsigned short*)g,
                           (float*)workspace, geom->t_out, geom->cout, geom->y_row0, geom->y_row_stride,
                           (long)geom->y_batch_stride);
    else
        hipLaunchKernelGGL((bias_grad_partial_kernel<float>), grid, dim3(256), 0, s, (const float*)g,
                           (float*)workspace, geom->t_out, geom->cout, geom->y_row0, geom->y_row_stride,
                           (long)geom->y_batch_stride);
    int rc = sl_check_launch("sl_bias_grad(partial)");
    if (rc != SL_OK) return rc;
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3((geom->cout + 255) / 256), dim3(256), 0, s,
                       (const float*)workspace, db, geom->batch, geom->cout);
    return sl_check_launch("sl_bias_grad(final)");
}

namespace {

// y = keep ? x * 1/(1-rate) : 0   (Keras inverted dropout, training phase).  4 elements per thread.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ src, T* __restrict__ dst, long n,
                                                      unsigned int threshold, float scale, unsigned long long seed) {
    const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long i = i0 + j;
        if (i < n) {
            const bool keep = dropout_bits(seed, (unsigned long long)i) >= threshold;
            float v;
            if (sizeof(T) == 2)
                v = bf16_bits_to_f32((unsigned short)src[i]);
            else
                v = (float)src[i];
            dst[i] = cvt_out<T>(keep ? v * scale : 0.f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* __restrict__ x, long n, float scale) {
    const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long i = i0 + j;
        if (i < n) {
            float v;
            if (sizeof(T) == 2)
                v = bf16_bits_to_f32((unsigned short)x[i]);
            else
                v = (float)x[i];
            x[i] = cvt_out<T>(v * scale);
        }
    }
}

// backward of Dropout behind an ELU layer: g holds dL/d(dropped activation) (dgrad with SL_EPI_NONE), y the stored
// post-dropout activation m' = keep ? elu(z) / (1 - rate) : 0.  The keep decision is recomputed from (seed, index) --
// a stored zero cannot tell "dropped" from elu(z) == 0 -- and elu'(z) = z > 0 ? 1 : elu(z) + 1 = m' * (1 - rate) + 1.
template <typename T>
__global__ __launch_bounds__(256) void elu_dropout_backward_kernel(T* __restrict__ g, const T* __restrict__ y, long n,
                                                                   unsigned int threshold, float scale, float keep_prob,
                                                                   unsigned long long seed) {
    const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long i = i0 + j;
        if (i < n) {
            const bool keep = dropout_bits(seed, (unsigned long long)i) >= threshold;
            float gv, m;
            if (sizeof(T) == 2) {
                gv = bf16_bits_to_f32((unsigned short)g[i]);
                m = bf16_bits_to_f32((unsigned short)y[i]);
            } else {
                gv = (float)g[i];
                m = (float)y[i];
            }
            const float d = m > 0.f ? 1.f : m * keep_prob + 1.f;
            g[i] = cvt_out<T>(keep ? gv * scale * d : 0.f);
        }
    }
}

}  // namespace

extern "C" int sl_elu_dropout_backward(void* g, const void* y, size_t n, int dtype, float rate, uint64_t seed,
                                       void* stream) {
    SL_CHECK_ARG(g && y, "sl_elu_dropout_backward: null pointer");
    SL_CHECK_ARG(rate >= 0.f && rate < 1.f, "sl_elu_dropout_backward: rate %f outside [0, 1)", (double)rate);
    SL_CHECK_ARG(dtype == SL_BF16 || dtype == SL_F32, "sl_elu_dropout_backward: unknown dtype %d", dtype);
    if (n == 0) return SL_OK;
    const unsigned int threshold = (unsigned int)((double)rate * 4294967296.0);
    const float scale = 1.f / (1.f - rate);
    const unsigned blocks = (unsigned)((n + 1023) / 1024);
    if (dtype == SL_BF16)
        hipLaunchKernelGGL(elu_dropout_backward_kernel<unsigned short>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (unsigned short*)g, (const unsigned short*)y, (long)n, threshold, scale, 1.f - rate,
                           (unsigned long long)seed);
    else
        hipLaunchKernelGGL(elu_dropout_backward_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (float*)g, (const float*)y, (long)n, threshold, scale, 1.f - rate, (unsigned long long)seed);
    return sl_check_launch("sl_elu_dropout_backward");
}

extern "C" int sl_dropout(const void* src, void* dst, size_t n, int dtype, float rate, uint64_t seed, void* stream) {
    SL_CHECK_ARG(src && dst, "sl_dropout: null pointer");
    SL_CHECK_ARG(rate >= 0.f && rate < 1.f, "sl_dropout: rate %f outside [0, 1)", (double)rate);
    SL_CHECK_ARG(dtype == SL_BF16 || dtype == SL_F32, "sl_dropout: unknown dtype %d", dtype);
    if (n == 0) return SL_OK;
    const unsigned int threshold = (unsigned int)((double)rate * 4294967296.0);
    const float scale = 1.f / (1.f - rate);
    const unsigned blocks = (unsigned)((n + 1023) / 1024);
    if (dtype == SL_BF16)
        hipLaunchKernelGGL(dropout_kernel<unsigned short>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)src, (unsigned short*)dst, (long)n, threshold, scale,
                           (unsigned long long)seed);
    else
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)src,
                           (float*)dst, (long)n, threshold, scale, (unsigned long long)seed);
    return sl_check_launch("sl_dropout");
}

extern "C" int sl_scale(void* x, size_t n, int dtype, float scale, void* stream) {
    SL_CHECK_ARG(x, "sl_scale: null pointer");
    SL_CHECK_ARG(dtype == SL_BF16 || dtype == SL_F32, "sl_scale: unknown dtype %d", dtype);
    if (n == 0) return SL_OK;
    const unsigned blocks = (unsigned)((n + 1023) / 1024);
    if (dtype == SL_BF16)
        hipLaunchKernelGGL(scale_kernel<unsigned short>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (unsigned short*)x, (long)n, scale);
    else
        hipLaunchKernelGGL(scale_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)x, (long)n,
                           scale);
    return sl_check_launch("sl_scale");
}

extern "C" int sl_adam_step(float* param, const float* grad, float* m, float* v, size_t n, int step, float lr,
                            float beta1, float beta2, float eps, void* stream) {
    SL_CHECK_ARG(param && grad && m && v, "sl_adam_step: null pointer");
    SL_CHECK_ARG(n % 4 == 0 && step >= 1, "sl_adam_step: n must be a multiple of 4 and step >= 1");
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    const long n4 = (long)(n / 4);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       m, v, n4, (float)lr_t, beta1, beta2, eps);
    return sl_check_launch("sl_adam_step");
}
