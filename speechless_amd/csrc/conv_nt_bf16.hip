// conv_nt_bf16.hip -- im2col-free row-shifted NT GEMM for the Wav2Letter conv stack on gfx950 (bf16 in, fp32 accumulate).
//
// Computes (see include/speechless_hip.h, sl_conv1d_nt):
//     y[b][y_row0+t][co] = epi( sum_tap sum_c x[b][x_row0+t+tap][c] * w[co][tap][c] )
// which is the forward of a Keras Conv1D (reference speechless/net.py:304-305) and, with flipped/transposed packed
// weights, its input gradient (autodiff from net.py:389,550).
//
// Mapping to CDNA4:
//   * work-group = 128 time rows x 128 output channels, 4 waves (2x2), each wave a 64x64 patch = 4x4 MFMA
//     v_mfma_f32_16x16x32_bf16 tiles; fp32 accumulators (64 VGPR/lane).
//   * the contraction runs over (tap, 64-channel chunk): one step = one 128x64 activation tile (a plain row-shifted
//     2-D tile of the halo'd channels-last tensor -> no im2col, no bounds checks) and one 128x64 weight tile.
//   * both tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered, one barrier
//     per step.  LDS rows are 128 B; the 16-B slot index is XOR-swizzled (on the SOURCE address, LDS image stays
//     lane-linear as the DMA requires) so that every ds_read_b128 lane group hits 16 distinct bank slots.
//   * D^T orientation: MFMA "A" = weights (rows = co), "B" = activations (cols = t), so each lane ends up holding
//     16 CONSECUTIVE output channels of one time row -> the epilogue (bias + ReLU / ReLU-mask, bf16 convert) stores
//     32 contiguous bytes per lane straight into the channels-last tensor.
//   * blockIdx -> tile mapping is XCD-aware: an XCD's work-groups share weight tiles in its private L2.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;   // 16 KiB per operand tile
constexpr int LDS_BYTES = 4 * TILE_BYTES;  // {X,W} x 2 buffers = 64 KiB -> 2 work-groups per CU

struct NtArgs {
    const __bf16* x;
    const __bf16* w;
    const float* bias;
    const __bf16* mask;
    void* y;
    int batch, t_out, t_tiles, n_tiles;
    int x_row0, x_rs;
    long x_bs;
    int y_row0, y_rs;
    long y_bs;
    int w_rs;    // taps * cin
    int chunks;  // cin / 64
    int nsteps;  // taps * chunks
};

__device__ __forceinline__ void glds16(const __bf16* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const SL_GLOBAL void*)gsrc, (SL_LDS void*)lds_wave_base, 16, 0, 0);
}

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void conv_nt_bf16_kernel(NtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1;  // which 64-row half (time)
    const int wn = wave & 1;   // which 64-channel half (co)
    const int g = lane >> 4;

    const int m_tiles = a.batch * a.t_tiles;
    const int wg = xcd_remap(blockIdx.x, m_tiles * a.n_tiles);
    const int n_tile = wg / m_tiles;
    const int m_tile = wg - n_tile * m_tiles;
    const int b = m_tile / a.t_tiles;
    const int t0 = (m_tile - b * a.t_tiles) * BM;
    const int co0 = n_tile * BN;

    // ---- staging addresses: wave `wave` DMA-copies rows [wave*32, wave*32+32) of both tiles, 8 rows per instruction
    const int srow = wave * 32 + (lane >> 3);                // + q*8
    const int xchunk = (lane & 7) ^ (lane >> 3);             // slot ^ (row & 7)
    const __bf16* xsrc = a.x + (long)b * a.x_bs + (long)(a.x_row0 + t0 + srow) * a.x_rs + xchunk * 8;
    const __bf16* wsrc = a.w + (long)(co0 + srow) * a.w_rs;
    int wchunk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 32 + q * 8 + (lane >> 3);
        const int key = ((row >> 1) & 1) | (((row >> 4) & 3) << 1);
        wchunk[q] = ((lane & 7) ^ key) * 8;
    }
    const long x_q_stride = 8L * a.x_rs;
    const long w_q_stride = 8L * a.w_rs;

    auto stage = [&](int step, int buf) {
        const int tap = step / a.chunks;
        const int cc = step - tap * a.chunks;
        const __bf16* xs = xsrc + (long)tap * a.x_rs + cc * BK;
        const __bf16* ws = wsrc + (long)step * BK;
        char* xl = smem + buf * (2 * TILE_BYTES) + wave * (32 * 128);
        char* wl = xl + TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(xs + q * x_q_stride, xl + q * 1024);
            glds16(ws + q * w_q_stride + wchunk[q], wl + q * 1024);
        }
    };

    // ---- fragment read addresses
    const int brow = wm * 64 + (lane & 15);                                    // + it*16
    const int arow = wn * 64 + ((lane & 15) >> 2) * 16 + (lane & 3);           // + jn*4
    const int bkey = lane & 7;
    const int akey = ((lane >> 1) & 1) | (((lane >> 2) & 3) << 1);
    const int boff = brow * 128 + ((g ^ bkey) << 4);
    const int aoff = arow * 128 + ((g ^ akey) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    for (int s = 0; s < a.nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < a.nsteps) stage(s + 1, (s + 1) & 1);
        const char* xl = smem + (s & 1) * (2 * TILE_BYTES);
        const char* wl = xl + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) af[jn] = *(const bf16x8*)(wl + ((aoff + jn * 512) ^ (kk << 6)));
#pragma unroll
            for (int it = 0; it < 4; ++it) bfr[it] = *(const bf16x8*)(xl + ((boff + it * 2048) ^ (kk << 6)));
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    acc[jn][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[jn], bfr[it], acc[jn][it], 0, 0, 0);
        }
    }

    // ---- epilogue: lane holds, for each it, 16 consecutive channels co_base..co_base+15 of time row t
    const int co_base = co0 + wn * 64 + g * 16;
    float bias_v[16];
    if (EPI == SL_EPI_BIAS || EPI == SL_EPI_BIAS_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 bv = *(const f32x4*)(a.bias + co_base + i * 4);
            bias_v[i * 4 + 0] = bv[0];
            bias_v[i * 4 + 1] = bv[1];
            bias_v[i * 4 + 2] = bv[2];
            bias_v[i * 4 + 3] = bv[3];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int t = t0 + wm * 64 + it * 16 + (lane & 15);
        if (t >= a.t_out) continue;
        const long yidx = (long)b * a.y_bs + (long)(a.y_row0 + t) * a.y_rs + co_base;
        float v[16];
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[jn * 4 + r] = acc[jn][it][r];
        if (EPI == SL_EPI_BIAS || EPI == SL_EPI_BIAS_RELU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += bias_v[i];
        }
        if (EPI == SL_EPI_BIAS_RELU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (EPI == SL_EPI_RELU_MASK) {
            const u32x4 m0 = *(const u32x4*)(a.mask + yidx);
            const u32x4 m1 = *(const u32x4*)(a.mask + yidx + 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // bf16 > 0  <=>  sign bit clear and magnitude non-zero
                const unsigned int lo0 = m0[i] & 0xFFFFu, hi0 = m0[i] >> 16;
                const unsigned int lo1 = m1[i] & 0xFFFFu, hi1 = m1[i] >> 16;
                if (!(lo0 != 0 && lo0 < 0x8000u)) v[i * 2] = 0.f;
                if (!(hi0 != 0 && hi0 < 0x8000u)) v[i * 2 + 1] = 0.f;
                if (!(lo1 != 0 && lo1 < 0x8000u)) v[8 + i * 2] = 0.f;
                if (!(hi1 != 0 && hi1 < 0x8000u)) v[8 + i * 2 + 1] = 0.f;
            }
        }
        if (OUT_F32) {
            float* yo = (float*)a.y + yidx;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(f32x4*)(yo + i * 4) = (f32x4){v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
        } else {
            __bf16* yo = (__bf16*)a.y + yidx;
            u32x4 p0, p1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p0[i] = pack_bf16x2(v[i * 2], v[i * 2 + 1]);
                p1[i] = pack_bf16x2(v[8 + i * 2], v[8 + i * 2 + 1]);
            }
            *(u32x4*)(yo) = p0;
            *(u32x4*)(yo + 8) = p1;
        }
    }
}

template <int EPI, bool OUT_F32>
int launch_one(const NtArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_nt_bf16_kernel<EPI, OUT_F32>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int grid = a.batch * a.t_tiles * a.n_tiles;
    hipLaunchKernelGGL((conv_nt_bf16_kernel<EPI, OUT_F32>), dim3(grid), dim3(256), LDS_BYTES, s, a);
    return sl_check_launch("sl_conv1d_nt(bf16)");
}

template <int EPI>
int launch(const NtArgs& a, int out_f32, hipStream_t s) {
    return out_f32 ? launch_one<EPI, true>(a, s) : launch_one<EPI, false>(a, s);
}

}  // namespace

int conv_nt_bf16(const void* x, const void* w, const float* bias, const void* mask, void* y, const sl_conv_geom* g,
                 int epilogue, int out_f32, hipStream_t s) {
    NtArgs a;
    a.x = (const __bf16*)x;
    a.w = (const __bf16*)w;
    a.bias = bias;
    a.mask = (const __bf16*)mask;
    a.y = y;
    a.batch = g->batch;
    a.t_out = g->t_out;
    a.t_tiles = (g->t_out + BM - 1) / BM;
    a.n_tiles = g->cout / BN;
    a.x_row0 = g->x_row0;
    a.x_rs = g->x_row_stride;
    a.x_bs = g->x_batch_stride;
    a.y_row0 = g->y_row0;
    a.y_rs = g->y_row_stride;
    a.y_bs = g->y_batch_stride;
    a.w_rs = g->taps * g->cin;
    a.chunks = g->cin / BK;
    a.nsteps = g->taps * a.chunks;
    switch (epilogue) {
        case SL_EPI_NONE: return launch<SL_EPI_NONE>(a, out_f32, s);
        case SL_EPI_BIAS: return launch<SL_EPI_BIAS>(a, out_f32, s);
        case SL_EPI_BIAS_RELU: return launch<SL_EPI_BIAS_RELU>(a, out_f32, s);
        case SL_EPI_RELU_MASK: return launch<SL_EPI_RELU_MASK>(a, out_f32, s);
    }
    sl_set_error("sl_conv1d_nt: unknown epilogue %d", epilogue);
    return SL_ERR_INVALID_ARGUMENT;
}
