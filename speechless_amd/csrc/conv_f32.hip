// conv_f32.hip -- fp32 realisation of the same two row-shifted GEMMs (sl_conv1d_nt / sl_conv1d_wgrad, dtype SL_F32).
//
// This is the PARITY path: BASELINE config 2 asks for bit-exact greedy-decoded label indices against the fp32 CPU
// path, which bf16 storage cannot promise on a near-flat random-init softmax.  Plain LDS-tiled fp32 FMA kernels with
// a fixed, sequential contraction order (tap-major, then channel) -> deterministic, fp32-roundoff-class error.
// Same HBM layout, same geometry struct, same epilogues as the bf16 MFMA kernels.
#include "common.h"

namespace {

constexpr int FT = 64;   // tile edge
constexpr int FK = 16;   // contraction depth per step
constexpr int FP = FT + 4;

struct F32NtArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* mask;
    float* y;
    int batch, t_out, t_tiles, n_tiles;
    int x_row0, x_rs;
    long x_bs;
    int y_row0, y_rs;
    long y_bs;
    int w_rs, chunks, nsteps;
};

template <int EPI>
__global__ __launch_bounds__(256) void conv_nt_f32_kernel(F32NtArgs a) {
    __shared__ float xs[FK][FP];
    __shared__ float ws[FK][FP];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int m_tiles = a.batch * a.t_tiles;
    const int wg = blockIdx.x;
    const int n_tile = wg / m_tiles;
    const int m_tile = wg - n_tile * m_tiles;
    const int b = m_tile / a.t_tiles;
    const int t0 = (m_tile - b * a.t_tiles) * FT;
    const int co0 = n_tile * FT;

    const int lrow = tid >> 2, lc4 = (tid & 3) * 4;
    const float* xsrc = a.x + (long)b * a.x_bs + (long)(a.x_row0 + t0 + lrow) * a.x_rs + lc4;
    const float* wsrc = a.w + (long)(co0 + lrow) * a.w_rs + lc4;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int s = 0; s < a.nsteps; ++s) {
        const int tap = s / a.chunks;
        const int cc = s - tap * a.chunks;
        const f32x4 xv = *(const f32x4*)(xsrc + (long)tap * a.x_rs + cc * FK);
        const f32x4 wv = *(const f32x4*)(wsrc + (long)s * FK);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xs[lc4 + e][lrow] = xv[e];
            ws[lc4 + e][lrow] = wv[e];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            const f32x4 av = *(const f32x4*)&xs[k][ty * 4];
            const f32x4 bv = *(const f32x4*)&ws[k][tx * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
    }

    const int co = co0 + tx * 4;
    f32x4 bias_v = {0.f, 0.f, 0.f, 0.f};
    if (EPI == SL_EPI_BIAS || EPI == SL_EPI_BIAS_RELU || EPI == SL_EPI_BIAS_ELU) bias_v = *(const f32x4*)(a.bias + co);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty * 4 + i;
        if (t >= a.t_out) continue;
        const long yidx = (long)b * a.y_bs + (long)(a.y_row0 + t) * a.y_rs + co;
        f32x4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        if (EPI == SL_EPI_BIAS || EPI == SL_EPI_BIAS_RELU || EPI == SL_EPI_BIAS_ELU) v += bias_v;
        if (EPI == SL_EPI_BIAS_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (EPI == SL_EPI_BIAS_ELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : expm1f(v[j]);
        }
        if (EPI == SL_EPI_ELU_MASK) {
            const f32x4 mv = *(const f32x4*)(a.mask + yidx);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= mv[j] > 0.f ? 1.f : mv[j] + 1.f;
        }
        if (EPI == SL_EPI_RELU_MASK) {
            const f32x4 mv = *(const f32x4*)(a.mask + yidx);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
        }
        *(f32x4*)(a.y + yidx) = v;
    }
}

template <int EPI>
int launch_nt(const F32NtArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((conv_nt_f32_kernel<EPI>), dim3(a.batch * a.t_tiles * a.n_tiles), dim3(256), 0, s, a);
    return sl_check_launch("sl_conv1d_nt(f32)");
}

struct F32TnArgs {
    const float* x;
    const float* g;
    float* out;
    int batch, taps, cin, cout, t_chunks;
    int x_row0, x_rs;
    long x_bs;
    int g_row0, g_rs;
    long g_bs;
    int ci_tiles, co_tiles, tiles, splits, b_per_split;
    long split_stride;
};

__global__ __launch_bounds__(256) void wgrad_tn_f32_kernel(F32TnArgs a) {
    __shared__ float xs[FK][FP];
    __shared__ float gs[FK][FP];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    int wg = blockIdx.x;
    const int tap = wg % a.taps;
    wg /= a.taps;
    const int ci_tile = wg % a.ci_tiles;
    wg /= a.ci_tiles;
    const int co_tile = wg % a.co_tiles;
    const int split = wg / a.co_tiles;
    const int b_begin = split * a.b_per_split;
    int b_end = b_begin + a.b_per_split;
    if (b_end > a.batch) b_end = a.batch;

    const int lrow = tid >> 4, lc4 = (tid & 15) * 4;
    const float* xbase = a.x + (long)(a.x_row0 + tap + lrow) * a.x_rs + ci_tile * FT + lc4;
    const float* gbase = a.g + (long)(a.g_row0 + lrow) * a.g_rs + co_tile * FT + lc4;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int b = b_begin; b < b_end; ++b) {
        for (int tc = 0; tc < a.t_chunks; ++tc) {
            const f32x4 xv = *(const f32x4*)(xbase + (long)b * a.x_bs + (long)(tc * FK) * a.x_rs);
            const f32x4 gv = *(const f32x4*)(gbase + (long)b * a.g_bs + (long)(tc * FK) * a.g_rs);
            __syncthreads();
            *(f32x4*)&xs[lrow][lc4] = xv;
            *(f32x4*)&gs[lrow][lc4] = gv;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < FK; ++k) {
                const f32x4 av = *(const f32x4*)&xs[k][ty * 4];
                const f32x4 bv = *(const f32x4*)&gs[k][tx * 4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
            }
        }
    }
    float* out = a.out + (long)split * a.split_stride;
    const int ci = ci_tile * FT + ty * 4;
    const int co = co_tile * FT + tx * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *(f32x4*)(out + ((long)tap * a.cin + ci + i) * a.cout + co) = (f32x4){acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
}

}  // namespace

int wgrad_reduce(const float* ws, float* dw, long n, int splits, hipStream_t s);

int conv_nt_f32(const void* x, const void* w, const float* bias, const void* mask, void* y, const sl_conv_geom* g,
                int epilogue, hipStream_t s) {
    F32NtArgs a;
    a.x = (const float*)x;
    a.w = (const float*)w;
    a.bias = bias;
    a.mask = (const float*)mask;
    a.y = (float*)y;
    a.batch = g->batch;
    a.t_out = g->t_out;
    a.t_tiles = (g->t_out + FT - 1) / FT;
    a.n_tiles = g->cout / FT;
    a.x_row0 = g->x_row0;
    a.x_rs = g->x_row_stride;
    a.x_bs = g->x_batch_stride;
    a.y_row0 = g->y_row0;
    a.y_rs = g->y_row_stride;
    a.y_bs = g->y_batch_stride;
    a.w_rs = g->taps * g->cin;
    a.chunks = g->cin / FK;
    a.nsteps = g->taps * a.chunks;
    switch (epilogue) {
        case SL_EPI_NONE: return launch_nt<SL_EPI_NONE>(a, s);
        case SL_EPI_BIAS: return launch_nt<SL_EPI_BIAS>(a, s);
        case SL_EPI_BIAS_RELU: return launch_nt<SL_EPI_BIAS_RELU>(a, s);
        case SL_EPI_RELU_MASK: return launch_nt<SL_EPI_RELU_MASK>(a, s);
        case SL_EPI_BIAS_ELU: return launch_nt<SL_EPI_BIAS_ELU>(a, s);
        case SL_EPI_ELU_MASK: return launch_nt<SL_EPI_ELU_MASK>(a, s);
    }
    sl_set_error("sl_conv1d_nt: unknown epilogue %d", epilogue);
    return SL_ERR_INVALID_ARGUMENT;
}

int wgrad_tn_f32(const void* x, const void* gr, float* dw, const sl_conv_geom* g, float* ws, int splits, hipStream_t s) {
    F32TnArgs a;
    a.x = (const float*)x;
    a.g = (const float*)gr;
    a.batch = g->batch;
    a.taps = g->taps;
    a.cin = g->cin;
    a.cout = g->cout;
    a.t_chunks = (g->t_out + FK - 1) / FK;
    a.x_row0 = g->x_row0;
    a.x_rs = g->x_row_stride;
    a.x_bs = g->x_batch_stride;
    a.g_row0 = g->y_row0;
    a.g_rs = g->y_row_stride;
    a.g_bs = g->y_batch_stride;
    a.ci_tiles = g->cin / FT;
    a.co_tiles = g->cout / FT;
    a.tiles = a.taps * a.ci_tiles * a.co_tiles;
    a.splits = splits;
    a.b_per_split = (g->batch + splits - 1) / splits;
    a.split_stride = (long)g->taps * g->cin * g->cout;
    a.out = splits > 1 ? ws : dw;
    hipLaunchKernelGGL(wgrad_tn_f32_kernel, dim3(a.tiles * splits), dim3(256), 0, s, a);
    int rc = sl_check_launch("sl_conv1d_wgrad(f32)");
    if (rc != SL_OK) return rc;
    if (splits > 1) return wgrad_reduce(ws, dw, a.split_stride, splits, s);
    return SL_OK;
}
