// wgrad_tn_bf16.hip -- conv weight gradient on gfx950 (bf16 in, fp32 accumulate, fp32 out).
//
//     dw[tap][c][co] = sum_b sum_t x[b][x_row0 + t + tap][c] * g[b][g_row0 + t][co]
// (TF Conv2DBackpropFilter reached by autodiff from reference speechless/net.py:389,550.)
//
// Both operands are time-major in HBM (channels-last), i.e. the contraction index t is the STRIDED one: a "TN" GEMM.
// Mapping to CDNA4:
//   * work-group = one (tap, 128 input channels, 128 output channels) tile of dw, 4 waves (2x2) of 64x64 patches,
//     v_mfma_f32_16x16x32_bf16; the contraction walks (utterance, 64-row time chunk); the tap is only a row shift
//     of the activation tile's source address.
//   * tiles are DMA-copied to LDS in their natural [t][channel] shape (global_load_lds_dwordx4, double buffered) and
//     the K-major MFMA fragments are produced by the gfx950 transpose read ds_read_b64_tr_b16 (one 4(t)x16(channel)
//     block per 16-lane group).  LDS rows are 256 B; the 32-B slot index is XOR-swizzled (on the DMA source address)
//     so that the 8 rows a 32-lane group touches cover all 64 banks.
//   * reduction over the batch is split across work-groups when the tile count cannot fill 256 CUs; partials go to a
//     workspace and are summed in a fixed order by wgrad_reduce_kernel (deterministic, no float atomics).
#include "common.h"

namespace {

constexpr int TCI = 128;
constexpr int TCO = 128;
constexpr int TK = 64;                      // time rows per step
constexpr int TILE_BYTES = TK * 128 * 2;    // 16 KiB
constexpr int LDS_BYTES = 4 * TILE_BYTES;   // {X,G} x 2 buffers

struct TnArgs {
    const __bf16* x;
    const __bf16* g;
    float* out;
    int batch, taps, cin, cout, t_chunks;
    int x_row0, x_rs;
    long x_bs;
    int g_row0, g_rs;
    long g_bs;
    int ci_tiles, co_tiles, tiles;
    int splits, b_per_split;
    long split_stride;
};

__device__ __forceinline__ void glds16(const __bf16* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const SL_GLOBAL void*)gsrc, (SL_LDS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 tr_read8(const char* p0) {
    // two 4x16 transpose reads: rows t..t+3 and t+4..t+7 (the +4 rows sit 4*256 B further)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SL_LDS s16x4*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SL_LDS s16x4*)(p0 + 4 * 256));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void wgrad_tn_bf16_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1;  // ci half
    const int wn = wave & 1;   // co half
    const int g = lane >> 4;

    // logical id = ((split * co_tiles + co_tile) * ci_tiles + ci_tile) * taps + tap
    int wg = xcd_remap(blockIdx.x, a.tiles * a.splits);
    const int tap = wg % a.taps;
    wg /= a.taps;
    const int ci_tile = wg % a.ci_tiles;
    wg /= a.ci_tiles;
    const int co_tile = wg % a.co_tiles;
    const int split = wg / a.co_tiles;
    const int b_begin = split * a.b_per_split;
    int b_end = b_begin + a.b_per_split;
    if (b_end > a.batch) b_end = a.batch;
    const int nsteps = (b_end - b_begin) * a.t_chunks;

    // ---- staging: wave copies rows [wave*16, wave*16+16) of both tiles, 4 rows (1 KiB) per instruction
    const int srow = wave * 16 + (lane >> 4);  // + q*4
    int schunk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int key = (lane >> 4) | ((q >> 1) << 2);
        const int slot16 = lane & 15;
        schunk[q] = ((((slot16 >> 1) ^ key) << 1) | (slot16 & 1)) * 8;
    }
    const __bf16* xbase = a.x + (long)(a.x_row0 + tap + srow) * a.x_rs + ci_tile * TCI;
    const __bf16* gbase = a.g + (long)(a.g_row0 + srow) * a.g_rs + co_tile * TCO;

    auto stage = [&](int step, int buf) {
        const int bb = step / a.t_chunks;
        const int tc = step - bb * a.t_chunks;
        const int b = b_begin + bb;
        const __bf16* xs = xbase + (long)b * a.x_bs + (long)(tc * TK) * a.x_rs;
        const __bf16* gs = gbase + (long)b * a.g_bs + (long)(tc * TK) * a.g_rs;
        char* xl = smem + buf * (2 * TILE_BYTES) + wave * (16 * 256);
        char* gl = xl + TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(xs + (long)(q * 4) * a.x_rs + schunk[q], xl + q * 1024);
            glds16(gs + (long)(q * 4) * a.g_rs + schunk[q], gl + q * 1024);
        }
    };

    // ---- transpose-read addresses.  lane (g, i): row = kk*32 + g*8 + h*4 + (i>>2); 8-B piece (i&3) of 16-col block c16
    const int i16 = lane & 15;
    const int rkey = (i16 >> 2) | ((g & 1) << 2);
    const int rrow = g * 8 + (i16 >> 2);
    // byte offset inside a tile for column block c16: rrow*256 + ((c16 ^ rkey) * 32) + (i16 & 3) * 8
    int xoff[4], goff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xoff[j] = rrow * 256 + (((wm * 4 + j) ^ rkey) * 32) + (i16 & 3) * 8;
        goff[j] = rrow * 256 + (((wn * 4 + j) ^ rkey) * 32) + (i16 & 3) * 8;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) stage(0, 0);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nsteps) stage(s + 1, (s + 1) & 1);
        const char* xl = smem + (s & 1) * (2 * TILE_BYTES);
        const char* gl = xl + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) af[jn] = tr_read8(gl + goff[jn] + kk * (32 * 256));
#pragma unroll
            for (int it = 0; it < 4; ++it) bfr[it] = tr_read8(xl + xoff[it] + kk * (32 * 256));
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    acc[jn][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[jn], bfr[it], acc[jn][it], 0, 0, 0);
        }
    }

    // ---- store: lane holds co = co_base + jn*16 + g*4 + {0..3} for ci = ci_base + it*16 + (lane & 15)
    float* out = a.out + (long)split * a.split_stride;
    const int ci_base = ci_tile * TCI + wm * 64 + (lane & 15);
    const int co_base = co_tile * TCO + wn * 64 + g * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long row = ((long)tap * a.cin + ci_base + it * 16) * a.cout;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) *(f32x4*)(out + row + co_base + jn * 16) = acc[jn][it];
    }
}

}  // namespace

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long n4, int splits,
                                    long split_stride4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4* p = (const f32x4*)ws + i;
    f32x4 s = p[0];
    for (int k = 1; k < splits; ++k) s += p[k * split_stride4];
    ((f32x4*)out)[i] = s;
}

int wgrad_split_count(const sl_conv_geom* g, int tile) {
    const long tiles = (long)g->taps * (g->cin / tile) * (g->cout / tile);
    long want = (512 + tiles - 1) / tiles;
    if (want < 1) want = 1;
    if (want > g->batch) want = g->batch;
    const int bps = (int)((g->batch + want - 1) / want);
    return (g->batch + bps - 1) / bps;
}

int wgrad_reduce(const float* ws, float* dw, long n, int splits, hipStream_t s) {
    const long n4 = n / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ws, dw, n4, splits,
                       n4);
    return sl_check_launch("wgrad_reduce");
}

int wgrad_tn_bf16(const void* x, const void* gr, float* dw, const sl_conv_geom* g, float* ws, int splits,
                  hipStream_t s) {
    TnArgs a;
    a.x = (const __bf16*)x;
    a.g = (const __bf16*)gr;
    a.batch = g->batch;
    a.taps = g->taps;
    a.cin = g->cin;
    a.cout = g->cout;
    a.t_chunks = (g->t_out + TK - 1) / TK;
    a.x_row0 = g->x_row0;
    a.x_rs = g->x_row_stride;
    a.x_bs = g->x_batch_stride;
    a.g_row0 = g->y_row0;
    a.g_rs = g->y_row_stride;
    a.g_bs = g->y_batch_stride;
    a.ci_tiles = g->cin / TCI;
    a.co_tiles = g->cout / TCO;
    a.tiles = a.taps * a.ci_tiles * a.co_tiles;
    a.splits = splits;
    a.b_per_split = (g->batch + splits - 1) / splits;
    a.split_stride = (long)g->taps * g->cin * g->cout;
    a.out = splits > 1 ? ws : dw;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_tn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_tn_bf16_kernel, dim3(a.tiles * splits), dim3(256), LDS_BYTES, s, a);
    int rc = sl_check_launch("sl_conv1d_wgrad(bf16)");
    if (rc != SL_OK) return rc;
    if (splits > 1) return wgrad_reduce(ws, dw, a.split_stride, splits, s);
    return SL_OK;
}
