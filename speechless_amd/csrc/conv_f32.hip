// conv_f32.hip -- fp32 realisation of the same two row-shifted GEMMs (sl_conv1d_nt / sl_conv1d_wgrad, dtype SL_F32).
//
// This is the PARITY path: BASELINE config 2 asks for bit-exact greedy-decoded label indices against the fp32 CPU
// path, which bf16 storage cannot promise on a near-flat random-init softmax.  Same HBM layout, same geometry struct,
// same epilogues as the bf16 MFMA kernels; fixed contraction order -> deterministic, fp32-roundoff-class error.
//
// Two realisations of each GEMM:
//   * conv_nt_f32_mfma_kernel / wgrad_tn_f32_mfma_kernel (default): the exact-fp32 matrix instruction of gfx950,
//     v_mfma_f32_32x32x2_f32 -- one f32 per lane for A and B, result bitwise a k-ordered fmaf chain, 64 FLOP/clk/SIMD
//     = 157 TFLOP/s, the fp32 vector peak, from one wave per SIMD (MI355X_MICROARCH.md) -- 128 x 128 tiles, four waves of
//     2 x 2 MFMA tiles, 16-deep steps double-buffered in LDS;
//   * conv_nt_f32_kernel / wgrad_tn_f32_kernel (cfg bit 0; the round-1 kernels, kept as an independent cross-check):
//     plain LDS-tiled VALU FMA loops.
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

// ---- exact-fp32 MFMA kernel ------------------------------------------------------------------------------------------
// D^T orientation: MFMA rows (A operand) = output channels, columns (B operand) = time rows, so a lane ends with four
// consecutive channels of one time row per accumulator quad -> 16-byte stores.  v_mfma_f32_32x32x2_f32 lane l feeds
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; which two contraction indices form a "pair" is free as long as
// A and B agree, so lanes < 32 take k = 4 * (2c) .. + 3 and lanes >= 32 take k = 4 * (2c + 1) .. + 3 of an 8-deep chunk c:
// one ds_read_b128 per operand row and chunk feeds four MFMAs.  LDS rows are 16 floats (four 16-byte slots); the slot
// index is XORed with (row >> 2) & 3 so that the 16 lanes of a ds_read_b128 group (rows r .. r+15, same slot) cover all
// 64 banks.
constexpr int MT = 128;  // tile edge (time rows x output channels)
constexpr int MK = 16;   // contraction depth per step
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int mfma_slot(int row, int slot) { return (row * 4 + (slot ^ ((row >> 2) & 3))) * 4; }

template <int EPI>
__global__ __launch_bounds__(256) void conv_nt_f32_mfma_kernel(F32NtArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[2][MT * MK];
    __shared__ __attribute__((aligned(16))) float ws[2][MT * MK];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wt = (wave & 1) * 64, wc = (wave >> 1) * 64;  // this wave's 64 time rows x 64 channels of the tile
    const int m_tiles = a.batch * a.t_tiles;
    const int wg = blockIdx.x;
    const int n_tile = wg / m_tiles;
    const int m_tile = wg - n_tile * m_tiles;
    const int b = m_tile / a.t_tiles;
    const int t0 = (m_tile - b * a.t_tiles) * MT;
    const int co0 = n_tile * MT;

    const int lrow = tid >> 2, lslot = tid & 3;  // global -> LDS: rows lrow and lrow + 64, one 16-byte slot each
    const float* xsrc = a.x + (long)b * a.x_bs + (long)(a.x_row0 + t0 + lrow) * a.x_rs + lslot * 4;
    const float* wsrc = a.w + (long)(co0 + lrow) * a.w_rs + lslot * 4;
    const long x64 = 64L * a.x_rs, w64 = 64L * a.w_rs;
    const int st0 = mfma_slot(lrow, lslot), st1 = mfma_slot(lrow + 64, lslot);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 xv0, xv1, wv0, wv1;
    auto fetch = [&](int s) {
        const int tap = s / a.chunks;
        const int cc = s - tap * a.chunks;
        const float* xp = xsrc + (long)tap * a.x_rs + cc * MK;
        const float* wp = wsrc + (long)s * MK;
        xv0 = *(const f32x4*)xp;
        xv1 = *(const f32x4*)(xp + x64);
        wv0 = *(const f32x4*)wp;
        wv1 = *(const f32x4*)(wp + w64);
    };
    auto stage = [&](int buf) {
        *(f32x4*)&xs[buf][st0] = xv0;
        *(f32x4*)&xs[buf][st1] = xv1;
        *(f32x4*)&ws[buf][st0] = wv0;
        *(f32x4*)&ws[buf][st1] = wv1;
    };
    fetch(0);
    stage(0);
    __syncthreads();
    const int r32 = lane & 31, half = lane >> 5;
    for (int s = 0; s < a.nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < a.nsteps) fetch(s + 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *(const f32x4*)&ws[buf][mfma_slot(wc + i * 32 + r32, c * 2 + half)];
                bf[i] = *(const f32x4*)&xs[buf][mfma_slot(wt + i * 32 + r32, c * 2 + half)];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < a.nsteps) stage(buf ^ 1);
        __syncthreads();
    }

    // C/D map of the 32x32 shapes: column = lane & 31 (time row), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (channel)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int t = t0 + wt + j * 32 + r32;
        if (t >= a.t_out) continue;
        const long yrow = (long)b * a.y_bs + (long)(a.y_row0 + t) * a.y_rs;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = co0 + wc + i * 32 + q * 8 + half * 4;
                const long yidx = yrow + co;
                f32x4 v = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                if (EPI == SL_EPI_BIAS || EPI == SL_EPI_BIAS_RELU || EPI == SL_EPI_BIAS_ELU)
                    v += *(const f32x4*)(a.bias + co);
                if (EPI == SL_EPI_BIAS_RELU) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
                }
                if (EPI == SL_EPI_BIAS_ELU) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = v[u] > 0.f ? v[u] : expm1f(v[u]);
                }
                if (EPI == SL_EPI_ELU_MASK) {
                    const f32x4 mv = *(const f32x4*)(a.mask + yidx);
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] *= mv[u] > 0.f ? 1.f : mv[u] + 1.f;
                }
                if (EPI == SL_EPI_RELU_MASK) {
                    const f32x4 mv = *(const f32x4*)(a.mask + yidx);
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = mv[u] > 0.f ? v[u] : 0.f;
                }
                *(f32x4*)(a.y + yidx) = v;
            }
        }
    }
}

template <int EPI>
int launch_nt_mfma(F32NtArgs a, int t_out, hipStream_t s) {
    a.t_tiles = (t_out + MT - 1) / MT;
    a.n_tiles = a.n_tiles * FT / MT;
    a.chunks = a.chunks * FK / MK;
    a.nsteps = a.nsteps * FK / MK;
    SL_LAUNCH_MAIN((conv_nt_f32_mfma_kernel<EPI>), dim3(a.batch * a.t_tiles * a.n_tiles), dim3(256), 0, s, a);
    return sl_check_launch("sl_conv1d_nt(f32, mfma)");
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

// ---- exact-fp32 MFMA weight gradient: dW[tap][ci][co] = sum_{b,t} x[b][t + tap][ci] * g[b][t][co] -----------------------
// Both operands are time-major, which is what v_mfma_f32_32x32x2_f32 wants from LDS rows: lane l reads
// A = x[t = 2p + (l >> 5)][ci = l & 31], B = g[t = 2p + (l >> 5)][co = l & 31] with plain ds_read_b32, lanes along the
// channel axis.  LDS rows are padded to 160 floats so that the two half-waves (rows 2p and 2p + 1) hit disjoint banks.
constexpr int WP = MT + 32;

__global__ __launch_bounds__(256) void wgrad_tn_f32_mfma_kernel(F32TnArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[2][MK * WP];
    __shared__ __attribute__((aligned(16))) float gs[2][MK * WP];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;  // this wave's 64 input x 64 output channels of the tile
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

    const int lrow = tid >> 5, lc4 = (tid & 31) * 4;  // global -> LDS: rows lrow and lrow + 8, four channels each
    const float* xbase = a.x + (long)(a.x_row0 + tap + lrow) * a.x_rs + ci_tile * MT + lc4;
    const float* gbase = a.g + (long)(a.g_row0 + lrow) * a.g_rs + co_tile * MT + lc4;
    const long x8 = 8L * a.x_rs, g8 = 8L * a.g_rs;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int per_b = a.t_chunks;
    const int nsteps = (b_end - b_begin) * per_b;
    f32x4 xv0, xv1, gv0, gv1;
    auto fetch = [&](int s) {
        const int b = b_begin + s / per_b;
        const int tc = s % per_b;
        const float* xp = xbase + (long)b * a.x_bs + (long)(tc * MK) * a.x_rs;
        const float* gp = gbase + (long)b * a.g_bs + (long)(tc * MK) * a.g_rs;
        xv0 = *(const f32x4*)xp;
        xv1 = *(const f32x4*)(xp + x8);
        gv0 = *(const f32x4*)gp;
        gv1 = *(const f32x4*)(gp + g8);
    };
    auto stage = [&](int buf) {
        *(f32x4*)&xs[buf][lrow * WP + lc4] = xv0;
        *(f32x4*)&xs[buf][(lrow + 8) * WP + lc4] = xv1;
        *(f32x4*)&gs[buf][lrow * WP + lc4] = gv0;
        *(f32x4*)&gs[buf][(lrow + 8) * WP + lc4] = gv1;
    };
    const int r32 = lane & 31, half = lane >> 5;
    if (nsteps > 0) {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) fetch(s + 1);
#pragma unroll
        for (int p = 0; p < MK / 2; ++p) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = xs[buf][(2 * p + half) * WP + wi + i * 32 + r32];
                bf[i] = gs[buf][(2 * p + half) * WP + wj + i * 32 + r32];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) stage(buf ^ 1);
        __syncthreads();
    }
    // C/D map: column = lane & 31 (output channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (input channel)
    float* out = a.out + (long)split * a.split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci_tile * MT + wi + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int co = co_tile * MT + wj + j * 32 + r32;
                out[((long)tap * a.cin + ci) * a.cout + co] = acc[i][j][r];
            }
}

}  // namespace

int wgrad_reduce(const float* ws, float* dw, long n, int splits, hipStream_t s);

// tile edge of the fp32 weight-gradient kernel for this geometry: 128 (MFMA) or 64 (VALU: cfg bit 0, or channel counts
// that are not multiples of 128)
int wgrad_f32_tile(const sl_conv_geom* g, int cfg) { return (!(cfg & 1) && g->cin % MT == 0 && g->cout % MT == 0) ? MT : FT; }

int conv_nt_f32(const void* x, const void* w, const float* bias, const void* mask, void* y, const sl_conv_geom* g,
                int epilogue, int cfg, hipStream_t s) {
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
    if (!(cfg & 1)) {  // default: the exact-fp32 MFMA kernel (cout is a multiple of 128, cin of 64 by the layout rules)
        switch (epilogue) {
            case SL_EPI_NONE: return launch_nt_mfma<SL_EPI_NONE>(a, g->t_out, s);
            case SL_EPI_BIAS: return launch_nt_mfma<SL_EPI_BIAS>(a, g->t_out, s);
            case SL_EPI_BIAS_RELU: return launch_nt_mfma<SL_EPI_BIAS_RELU>(a, g->t_out, s);
            case SL_EPI_RELU_MASK: return launch_nt_mfma<SL_EPI_RELU_MASK>(a, g->t_out, s);
            case SL_EPI_BIAS_ELU: return launch_nt_mfma<SL_EPI_BIAS_ELU>(a, g->t_out, s);
            case SL_EPI_ELU_MASK: return launch_nt_mfma<SL_EPI_ELU_MASK>(a, g->t_out, s);
        }
    }
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

int wgrad_tn_f32(const void* x, const void* gr, float* dw, const sl_conv_geom* g, float* ws, int splits, int cfg,
                 hipStream_t s) {
    const int tile = wgrad_f32_tile(g, cfg);
    F32TnArgs a;
    a.x = (const float*)x;
    a.g = (const float*)gr;
    a.batch = g->batch;
    a.taps = g->taps;
    a.cin = g->cin;
    a.cout = g->cout;
    a.t_chunks = (g->t_out + FK - 1) / FK;  // (FK == MK)
    a.x_row0 = g->x_row0;
    a.x_rs = g->x_row_stride;
    a.x_bs = g->x_batch_stride;
    a.g_row0 = g->y_row0;
    a.g_rs = g->y_row_stride;
    a.g_bs = g->y_batch_stride;
    a.ci_tiles = g->cin / tile;
    a.co_tiles = g->cout / tile;
    a.tiles = a.taps * a.ci_tiles * a.co_tiles;
    a.splits = splits;
    a.b_per_split = (g->batch + splits - 1) / splits;
    a.split_stride = (long)g->taps * g->cin * g->cout;
    a.out = splits > 1 ? ws : dw;
    if (tile == MT)
        SL_LAUNCH_MAIN(wgrad_tn_f32_mfma_kernel, dim3(a.tiles * splits), dim3(256), 0, s, a);
    else
        SL_LAUNCH_MAIN(wgrad_tn_f32_kernel, dim3(a.tiles * splits), dim3(256), 0, s, a);
    int rc = sl_check_launch("sl_conv1d_wgrad(f32)");
    if (rc != SL_OK) return rc;
    if (splits > 1) return wgrad_reduce(ws, dw, a.split_stride, splits, s);
    return SL_OK;
}
