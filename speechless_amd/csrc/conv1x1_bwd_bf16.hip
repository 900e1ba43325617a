// conv1x1_bwd_bf16.hip -- backward of a 1x1 convolution onto k <= 32 real output channels in ONE launch (bf16, gfx950).
//
// output_conv (reference speechless/net.py:326-330: Conv1D(grapheme_set_size, 1)) maps 2000 channels onto 29 classes.  Its
// backward pass is two GEMMs with almost no arithmetic:
//     dx[b][t][ci] = (sum_co g[b][t][co] * w[ci][co]) * act'(x[b][t][ci])          (Conv2DBackpropInput + ReLU / ELU mask)
//     dw[ci][co]   =  sum_{b,t} x[b][t][ci] * g[b][t][co]                            (Conv2DBackpropFilter)
// As separate launches (sl_conv1d_nt + sl_conv1d_wgrad, 128-wide tiles of which three quarters are channel padding) they
// read the layer's input x -- 65 MB at configuration 3 -- twice and run at 3.3 / 1.4 TB/s.  Here a work-group owns 128
// input channels x one contiguous range of 64-frame chunks: the x tile of a chunk goes to LDS ONCE (global_load_lds, 3-slot
// ring, counted vmcnt) and serves as the MFMA operand of dw (transposed reads, ds_read_b64_tr_b16), as the activation mask
// of dx and, through the ones channel (row cin_pad - 1 of dw, see sl_bias_grad_from_wgrad), as the bias gradient.
//   * dx: D^T orientation (MFMA A = the 128 x 32 block of w, held in registers for the whole kernel; B = the chunk's g rows)
//     with the weight rows in a permuted order, so that a lane ends with 16 consecutive channels of one time row: 32-byte
//     contiguous stores, every store instruction writes whole 128-byte lines.
//   * dw: per work-group partial sums over its frames (128 x 32 fp32 = 16 KB slab), combined in a FIXED order by a small
//     second launch (conv1x1_bwd_reduce_kernel: 8 MB of slabs, ~3 us) -> deterministic, no float atomics.
// HBM traffic: x once + g once + dx once (134 MB at configuration 3) against x twice + g twice + dx + partial tiles.
#include "common.h"

namespace {

constexpr int TK = 64;    // frames per step
constexpr int CBW = 128;  // input channels per work-group
constexpr int KC = 32;    // output channels computed (k <= 32 real ones; the padding beyond stays zero)
constexpr int SLOTS = 3;
constexpr int X_BYTES = TK * CBW * 2;  // 16 KB: [64 rows][128 ch] bf16, 32-B slots XOR-swizzled per row
constexpr int G_BYTES = TK * KC * 2;   // 4 KB:  [64 rows][32 co] bf16, natural layout
constexpr int SLOT_BYTES = X_BYTES + G_BYTES;
constexpr int NI = 5;  // LDS-DMA instructions per wave and tile (4 x, 1 g)
#if defined(SL_BWD1_PROBE_NO_DX)  // timing probes (wrong results by construction): see tools/README.md
constexpr int NS = 0;
#else
constexpr int NS = 4;  // dx store instructions per lane and step
#endif

struct BwdArgs {
    const __bf16* x;
    const __bf16* g;
    const __bf16* wd;  // [cin][wd_rs] bf16, columns [0, 32) used
    __bf16* dx;
    float* dw;        // [cin][cout_pad] fp32
    float* slabs;     // [splits][cin][32] fp32
    int batch, t_chunks, cin, cout_pad, wd_rs;
    int x_row0, x_rs;
    long x_bs;
    int g_row0, g_rs;
    long g_bs;
    int n_cb, splits, chunks_per_split, total_chunks;
};

__device__ __forceinline__ void glds16(const __bf16* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const SL_GLOBAL void*)gsrc, (SL_LDS void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// LDS reads go through inline asm with one hand-placed wait: through plain loads the compiler cannot tell that a read does
// not alias the LDS-DMA requests in flight and drains them (s_waitcnt vmcnt(0)) in front of the first read of every step
template <int OFF>
__device__ __forceinline__ void ds_read128(u32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void ds_tr(s16x4& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ bf16x8 frag8(s16x4 lo, s16x4 hi) {
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
struct StepFrags {
    u32x4 gb;         // dx: B fragment, g[row][co 8q .. 8q+7]
    u32x4 mask[4];    // dx: x[row][16 consecutive channels] for both 64-channel groups (2 x 32 B)
    s16x4 gl[2][2], gh[2][2];  // dw: g^T fragments [k-half][co tile]
    s16x4 xl[2][2], xh[2][2];  // dw: x^T fragments [k-half][ci tile]
};
__device__ __forceinline__ void wait_step_frags(StepFrags& f) {  // lgkmcnt(0) tied to every register the reads fill
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.gb), "+v"(f.mask[0]), "+v"(f.mask[1]), "+v"(f.mask[2]), "+v"(f.mask[3]), "+v"(f.gl[0][0]),
                   "+v"(f.gh[0][0]), "+v"(f.gl[0][1]), "+v"(f.gh[0][1]), "+v"(f.gl[1][0]), "+v"(f.gh[1][0]), "+v"(f.gl[1][1]),
                   "+v"(f.gh[1][1]), "+v"(f.xl[0][0]), "+v"(f.xh[0][0]), "+v"(f.xl[0][1]), "+v"(f.xh[0][1]), "+v"(f.xl[1][0]),
                   "+v"(f.xh[1][0]), "+v"(f.xl[1][1]), "+v"(f.xh[1][1]));
}

// chain rule through the activation, from the STORED activation y (as sl_conv1d_nt's SL_EPI_RELU_MASK / SL_EPI_ELU_MASK)
template <int MODE>
__device__ __forceinline__ float through_act(float d, float y) {
    if (MODE == SL_EPI_ELU_MASK) return y > 0.f ? d : d * (y + 1.f);
    return y > 0.f ? d : 0.f;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv1x1_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, i16 = lane & 15;

    // logical id = split * n_cb + channel block: the channel blocks of one frame range sit on one XCD and share its g rows
    const int id = xcd_remap(blockIdx.x, a.splits * a.n_cb);
    if (id >= a.splits * a.n_cb) return;  // grid padding (xcd_grid)
    const int split = id / a.n_cb;
    const int cb = id - split * a.n_cb;
    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > a.total_chunks) c_end = a.total_chunks;
    const int n = c_end - c_begin;  // >= 1 (the host leaves no split empty)

    // ---- the 128 x 32 block of w as MFMA A fragments, in registers for the whole kernel.  Tile j = 4 * grp + jj holds the
    // rows m <-> channel grp * 64 + (m / 4) * 16 + jj * 4 + m % 4, so that lane (q, i) ends with D rows 4q .. 4q + 3 of the four
    // tiles of a group = the 16 consecutive channels grp * 64 + 16 q .. + 15 of time row i.
    bf16x8 wfrag[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = (j >> 2) * 64 + (i16 >> 2) * 16 + (j & 3) * 4 + (i16 & 3);
        wfrag[j] = *(const bf16x8*)(a.wd + (long)(cb * CBW + ch) * a.wd_rs + q * 8);
    }
    // (a use in front of the loop: the compiler's wait for these loads must not land inside it)
    asm volatile("" : "+v"(wfrag[0]), "+v"(wfrag[1]), "+v"(wfrag[2]), "+v"(wfrag[3]), "+v"(wfrag[4]), "+v"(wfrag[5]),
                 "+v"(wfrag[6]), "+v"(wfrag[7]));

    // ---- request stream (sources advance by increments; past the end the last tile is requested again into a free slot,
    // which keeps the number of requests per step -- and with it every counted wait -- uniform)
    const int b0 = c_begin / a.t_chunks, tc0 = c_begin - b0 * a.t_chunks;
    const long x_step = (long)TK * a.x_rs, g_step = (long)TK * a.g_rs;
    const long x_wrap = a.x_bs - (long)(a.t_chunks - 1) * x_step, g_wrap = a.g_bs - (long)(a.t_chunks - 1) * g_step;
    const __bf16* xs = a.x + (long)b0 * a.x_bs + (long)(a.x_row0 + tc0 * TK) * a.x_rs + cb * CBW;
    const __bf16* gs = a.g + (long)b0 * a.g_bs + (long)(a.g_row0 + tc0 * TK) * a.g_rs;
    int tc_r = tc0, left = n;
    int xoff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {  // 1-KiB instruction = 4 rows x 256 B; 32-B slot XOR key(row) (as wgrad_tn_bf16.hip)
        const int row = (wave * 4 + p) * 4 + (lane >> 4);
        const int key = (row & 3) | (((row >> 3) & 1) << 2);
        const int chunk16 = ((((lane & 15) >> 1) ^ key) << 1) | (lane & 1);
        xoff[p] = row * a.x_rs + chunk16 * 8;
    }
    const int goff = (wave * 16 + (lane >> 2)) * a.g_rs + (lane & 3) * 8;
    auto request = [&](int slot) {
        char* xl = smem + slot * SLOT_BYTES + wave * 4096;
#pragma unroll
        for (int p = 0; p < 4; ++p) glds16(xs + xoff[p], xl + p * 1024);
        glds16(gs + goff, smem + slot * SLOT_BYTES + X_BYTES + wave * 1024);
        const bool more = left > 1;
        const bool wrap = tc_r + 1 == a.t_chunks;
        xs += more ? (wrap ? x_wrap : x_step) : 0L;
        gs += more ? (wrap ? g_wrap : g_step) : 0L;
        tc_r = more ? (wrap ? 0 : tc_r + 1) : tc_r;
        left = more ? left - 1 : left;
    };

    // ---- LDS read addresses inside a slot
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int drow = wave * 16 + i16;  // dx: this lane's time row of the chunk
    const int dkey = (drow & 3) | (((drow >> 3) & 1) << 2);
    const unsigned gb_off = X_BYTES + drow * (KC * 2) + q * 16;
    unsigned mask_off[2];
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) mask_off[grp] = drow * 256 + (((grp * 4 + q) ^ dkey) * 32);
    const int rrow = q * 8 + (i16 >> 2);                 // dw: transposed reads, row inside a k-half
    const int rkey = (i16 >> 2) | ((q & 1) << 2);        // = key(rrow) = key(rrow + 4) = key(rrow + 32)
    unsigned gt_off[2], xt_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        gt_off[j] = X_BYTES + rrow * (KC * 2) + j * 32 + (i16 & 3) * 8;
        xt_off[j] = rrow * 256 + (((wave * 2 + j) ^ rkey) * 32) + (i16 & 3) * 8;
    }

    f32x4 wacc[2][2];  // dw: [co tile][ci tile]: co = jn * 16 + 4 q + r, ci = wave * 32 + it * 16 + i16
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int it = 0; it < 2; ++it) wacc[jn][it] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // dx store stream: row (x_row0 + chunk * 64 + drow) of the chunk's utterance, channels cb * 128 + grp * 64 + 16 q
    __bf16* dxp = a.dx + (long)b0 * a.x_bs + (long)(a.x_row0 + tc0 * TK + drow) * a.x_rs + cb * CBW + q * 16;
    int tc_c = tc0;

    request(0);
    request(1);
    int slot = 0;
    for (int i = 0; i < n; ++i) {
        // tile i has landed: everything issued after its requests may stay in flight -- the requests of tile i + 1 and
        // (from the second step on) the dx stores of the steps since (vmcnt counts loads and stores in issue order)
        if (i == 0)
            wait_vmcnt<NI>();
        else if (i == 1)
            wait_vmcnt<NI + NS>();
        else
            wait_vmcnt<NI + NS + NS>();
        __builtin_amdgcn_s_barrier();  // ... everybody's share of it; and everybody is done reading tile i - 1
        asm volatile("" ::: "memory");
        request(slot == 0 ? SLOTS - 1 : slot - 1);  // tile i + 2 into the slot of tile i - 1
        const unsigned sl = lds0 + slot * SLOT_BYTES;
        StepFrags f;
        ds_read128<0>(f.gb, sl + gb_off);
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            ds_read128<0>(f.mask[grp * 2], sl + mask_off[grp]);
            ds_read128<16>(f.mask[grp * 2 + 1], sl + mask_off[grp]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ds_tr<0>(f.gl[0][j], sl + gt_off[j]);
            ds_tr<4 * KC * 2>(f.gh[0][j], sl + gt_off[j]);
            ds_tr<32 * KC * 2>(f.gl[1][j], sl + gt_off[j]);
            ds_tr<36 * KC * 2>(f.gh[1][j], sl + gt_off[j]);
            ds_tr<0>(f.xl[0][j], sl + xt_off[j]);
            ds_tr<4 * 256>(f.xh[0][j], sl + xt_off[j]);
            ds_tr<32 * 256>(f.xl[1][j], sl + xt_off[j]);
            ds_tr<36 * 256>(f.xh[1][j], sl + xt_off[j]);
        }
        wait_step_frags(f);
        // ---- dx: 8 tiles of 16 channels x this wave's 16 rows, one 32-deep MFMA each
        const bf16x8 gfrag = __builtin_bit_cast(bf16x8, f.gb);
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            f32x4 d[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                d[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[grp * 4 + jj], gfrag, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            unsigned int out[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4 m = f.mask[grp * 2 + h];
#pragma unroll
                for (int p = 0; p < 4; ++p) {  // dword p of half h: channels 8 h + 2 p, + 1 = tile jj = 2 h + p / 2, rows 2 (p % 2), + 1
                    const int jj = 2 * h + (p >> 1), r = 2 * (p & 1);
                    const float y0 = __uint_as_float(m[p] << 16), y1 = __uint_as_float(m[p] & 0xFFFF0000u);
                    out[h * 4 + p] = pack_bf16x2_hw(through_act<MODE>(d[jj][r], y0), through_act<MODE>(d[jj][r + 1], y1));
                }
            }
            // rows beyond the valid time hold exact zeros here (their g rows are zero by the layout invariant), so the
            // stores are unconditional: the invariant survives and the number of stores per step stays uniform
#if defined(SL_BWD1_PROBE_NO_DX)
            asm volatile("" ::"v"(out[0]), "v"(out[1]), "v"(out[2]), "v"(out[3]), "v"(out[4]), "v"(out[5]), "v"(out[6]), "v"(out[7]));
#else
            *(u32x4*)(dxp + grp * 64) = (u32x4){out[0], out[1], out[2], out[3]};
            *(u32x4*)(dxp + grp * 64 + 8) = (u32x4){out[4], out[5], out[6], out[7]};
#endif
        }
        // ---- dw: 32 input channels of this wave x 32 output channels, two 32-frame k-halves
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const bf16x8 af = frag8(f.gl[kk][jn], f.gh[kk][jn]);
#pragma unroll
                for (int it = 0; it < 2; ++it)
                    wacc[jn][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, frag8(f.xl[kk][it], f.xh[kk][it]), wacc[jn][it], 0, 0, 0);
            }
        const bool wrap = tc_c + 1 == a.t_chunks;
        dxp += wrap ? x_wrap : x_step;
        tc_c = wrap ? 0 : tc_c + 1;
        slot = (slot + 1 == SLOTS) ? 0 : slot + 1;
    }

    wait_vmcnt<0>();  // the surplus requests of the branch-free stream must not outlive the work-group's LDS
    // ---- this work-group's partial dw: slab [split][ci][32]
    float* slab = a.slabs + ((long)split * a.cin + cb * CBW + wave * 32 + i16) * KC + q * 4;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) *(f32x4*)(slab + (long)it * 16 * KC + jn * 16) = wacc[jn][it];
}

// dw = sum over the splits' slabs in split order (fixed: deterministic), one float4 per thread, eight loads in flight at a
// time -- all unconditional (the index is clamped, the surplus values are not added): a load behind a runtime condition
// would be waited for on its own, one dependent round trip per slab.  Also writes the zeros of the padded output channels.
// A launch of its own behind the main kernel: the slabs of a channel block are 32 x 16 KB -- far beyond what a last-arriving
// work-group combines for free (measured: release fence + ticket + one combining work-group per channel block 29 us, more
// than the main kernel's 25), so the combine happens at the launch boundary (cdna_hip_programming.md, in-launch split-K).
__global__ __launch_bounds__(256) void conv1x1_bwd_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                 int cin, int cout_pad, int splits, int accumulate) {
    const int e = blockIdx.x * 256 + threadIdx.x;  // float4 index: ci = e / 8, co = (e % 8) * 4
    if (e >= cin * (KC / 4)) return;
    const int ci = e >> 3, co = (e & 7) * 4;
    const float* src = slabs + (long)ci * KC + co;
    const long stride = (long)cin * KC;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < splits; k0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j < splits ? k0 + j : splits - 1;
            v[j] = *(const f32x4*)(src + k * stride);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 t = s + v[j];
            s = k0 + j == 0 ? v[j] : (k0 + j < splits ? t : s);
        }
    }
    float* out = dw + (long)ci * cout_pad;
    if (accumulate) {  // a later part of the batch (sl_conv1d_backward_1x1_part): dw = dw_of_the_parts_before + this part's sum
        *(f32x4*)(out + co) = *(const f32x4*)(out + co) + s;
        return;
    }
    *(f32x4*)(out + co) = s;
    for (int z = KC + co; z < cout_pad; z += KC) *(f32x4*)(out + z) = (f32x4){0.f, 0.f, 0.f, 0.f};
}

int pick_splits(const sl_conv_geom* g, int cfg, int* chunks_per_split) {
    const int t_chunks = (g->t_out + TK - 1) / TK;
    const int total = g->batch * t_chunks;
    const int n_cb = g->cin / CBW;
    const int target = cfg > 0 ? cfg : sl_cus();  // work-groups: one per CU measured best (26.5 us; 512: 29.0, 768: 32.5)
    int want = (target + n_cb - 1) / n_cb;
    if (want > total) want = total;
    if (want < 1) want = 1;
    int cps = (total + want - 1) / want;
    if (cps < 2 && total >= 2) cps = 2;  // (a one-step work-group is all prologue)
    *chunks_per_split = cps;
    return (total + cps - 1) / cps;  // no empty split
}

}  // namespace

bool conv1x1_bwd_bf16_supported(const sl_conv_geom* g, int k_real) {
    // g: the layer's weight-gradient geometry (x_*: its input x, y_*: the gradient of its output)
    return g->taps == 1 && k_real >= 1 && k_real <= KC && g->cout >= KC && g->cout % KC == 0 && g->cin % CBW == 0 &&
           g->x_row_stride % 8 == 0 && g->y_row_stride % 8 == 0 && g->batch > 0 && g->t_out > 0;
}

size_t conv1x1_bwd_bf16_workspace_bytes(const sl_conv_geom* g, int cfg) {
    int cps;
    const int splits = pick_splits(g, cfg, &cps);
    return (size_t)splits * g->cin * KC * sizeof(float);  // the work-groups' partial dw slabs
}

int conv1x1_bwd_bf16(const void* x, const void* gr, const void* w_dgrad, void* dx, float* dw, const sl_conv_geom* g,
                     int epilogue, int cfg, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    BwdArgs a;
    a.x = (const __bf16*)x;
    a.g = (const __bf16*)gr;
    a.wd = (const __bf16*)w_dgrad;
    a.dx = (__bf16*)dx;
    a.dw = dw;
    a.batch = g->batch;
    a.t_chunks = (g->t_out + TK - 1) / TK;
    a.cin = g->cin;
    a.cout_pad = g->cout;
    a.wd_rs = g->cout;
    a.x_row0 = g->x_row0;
    a.x_rs = g->x_row_stride;
    a.x_bs = g->x_batch_stride;
    a.g_row0 = g->y_row0;
    a.g_rs = g->y_row_stride;
    a.g_bs = g->y_batch_stride;
    a.n_cb = g->cin / CBW;
    a.total_chunks = a.batch * a.t_chunks;
    a.splits = pick_splits(g, cfg, &a.chunks_per_split);
    if (ws == nullptr || ws_bytes < conv1x1_bwd_bf16_workspace_bytes(g, cfg)) {
        sl_set_error("sl_conv1d_backward_1x1: workspace too small");
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    a.slabs = (float*)ws;
    const int grid = xcd_grid(a.splits * a.n_cb);
    const int lds = SLOTS * SLOT_BYTES;
    if (epilogue == SL_EPI_ELU_MASK)
        SL_LAUNCH_MAIN((conv1x1_bwd_kernel<SL_EPI_ELU_MASK>), dim3(grid), dim3(256), lds, s, a);
    else
        SL_LAUNCH_MAIN((conv1x1_bwd_kernel<SL_EPI_RELU_MASK>), dim3(grid), dim3(256), lds, s, a);
    int rc = sl_check_launch("sl_conv1d_backward_1x1");
    if (rc != SL_OK) return rc;
    const int n4 = g->cin * (KC / 4);
    hipLaunchKernelGGL(conv1x1_bwd_reduce_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, a.slabs, dw, g->cin, g->cout,
                       a.splits, accumulate);
    return sl_check_launch("sl_conv1d_backward_1x1(reduce)");
}
