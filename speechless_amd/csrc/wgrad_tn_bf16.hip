// wgrad_tn_bf16.hip -- conv weight gradient on gfx950 (bf16 in, fp32 accumulate, fp32 out).
//
//     dw[tap][c][co] = sum_b sum_t x[b][x_row0 + t + tap][c] * g[b][g_row0 + t][co]
// (TF Conv2DBackpropFilter reached by autodiff from reference speechless/net.py:389,550.)
//
// Both operands are time-major in HBM (channels-last), i.e. the contraction index t is the STRIDED one: a "TN" GEMM.
// Mapping to CDNA4:
//   * work-group = one (tap, 64*WM input channels, 64*WN output channels) tile of dw, WM*WN waves of 64x64 patches
//     (v_mfma_f32_16x16x32_bf16, fp32 accumulators); 128x128 (4 waves) ... 256x256 (16 waves, one work-group per CU).
//     The contraction walks (utterance, 64-row time chunk); the tap is only a row shift of the activation tile's
//     source address (im2col-free).
//   * tiles are DMA-copied to LDS in their natural [t][channel] shape (global_load_lds_dwordx4) through a STAGES-deep
//     ring with counted s_waitcnt vmcnt(N) + raw s_barrier, and the K-major MFMA fragments are produced by the gfx950
//     transpose read ds_read_b64_tr_b16 (one 4(t) x 16(channel) block per 16-lane group).  The 32-B slot index of each
//     LDS row is XOR-swizzled (on the DMA source address) so that the 8 rows a 32-lane group touches cover all 64 banks.
//   * reduction over the batch is split across work-groups when the tile count cannot fill 256 CUs; partials go to a
//     workspace and are summed in a fixed order by wgrad_reduce_kernel (deterministic, no float atomics).
#include "common.h"

#include <type_traits>

namespace {

constexpr int TK = 64;  // time rows per step

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
    // grouped launch: `groups` independent layers of identical geometry (the seven inner_conv_i) in one grid;
    // layer q uses x + q*x_gs, g + q*g_gs and writes dw + q*dw_gs (or workspace slice q*splits + split)
    int groups;
    long x_gs, g_gs, dw_gs;
};

__device__ __forceinline__ void glds16(const __bf16* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const SL_GLOBAL void*)gsrc, (SL_LDS void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Transpose reads are issued through inline asm with a hand-counted s_waitcnt.  Through the builtin the compiler cannot
// tell that the read does not alias the LDS-DMA loads in flight and puts s_waitcnt vmcnt(0) in front of the first
// read of every step -- AFTER the DMA of the next tile has been issued -- which serialises the HBM/L2 latency of every
// tile with its MFMAs (measured: 1.9 us per 64-row step where the MFMAs need 1.0).
template <int OFF>
__device__ __forceinline__ void ds_tr(s16x4& d, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// one 8(t) x 16(channel) fragment = two 4x16 transpose reads: rows t..t+3 and t+4..t+7 (4 LDS rows further)
template <int OFF, int RB>
__device__ __forceinline__ void tr_read8(s16x4& lo, s16x4& hi, unsigned addr) {
    ds_tr<OFF>(lo, addr);
    ds_tr<OFF + 4 * RB>(hi, addr);
}
__device__ __forceinline__ bf16x8 frag8(s16x4 lo, s16x4 hi) {
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// s_waitcnt lgkmcnt(CNT) the MFMAs consuming these half-fragments cannot be hoisted above
template <int CNT>
__device__ __forceinline__ void wait_ba(s16x4 (&bl)[4], s16x4 (&bh)[4], s16x4 (&al)[2], s16x4 (&ah)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]), "+v"(bh[3]),
                   "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1])
                 : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void wait_a(s16x4 (&al)[2], s16x4 (&ah)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]) : "n"(CNT));
}

// per-lane source offset (elements) of DMA instruction j for a [64][W*64] bf16 tile whose LDS rows are RB = 128*W bytes:
// lane -> row j*RPI + lane/LPR, 16-B slot lane%LPR; the 32-B slot index is XORed with key(row) = (row&3)|((row>>3)&1)<<2
template <int W>
__device__ __forceinline__ int dma_src_offset(int j, int lane, int row_stride) {
    constexpr int LPR = 8 * W;      // lanes (16-B slots) per row
    constexpr int RPI = 64 / LPR;   // rows per 1-KiB DMA instruction
    const int row = j * RPI + lane / LPR;
    const int slot16 = lane % LPR;
    const int key = (row & 3) | (((row >> 3) & 1) << 2);
    const int chunk16 = (((slot16 >> 1) ^ key) << 1) | (slot16 & 1);
    return row * row_stride + chunk16 * 8;
}

template <int WM, int WN, int STAGES>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16 ? 4 : 2)) void wgrad_tn_bf16_kernel(TnArgs a) {
    constexpr int NW = WM * WN;
    constexpr int TCI = 64 * WM;
    constexpr int TCO = 64 * WN;
    constexpr int XRB = 128 * WM;  // LDS row bytes of the activation tile
    constexpr int GRB = 128 * WN;
    constexpr int X_BYTES = TK * XRB;
    constexpr int STAGE_BYTES = TK * (XRB + GRB);
    constexpr int XPW = (8 * WM) / NW;  // DMA instructions per wave per stage (activation tile = 8*WM KiB)
    constexpr int GPW = (8 * WN) / NW;
    static_assert((8 * WM) % NW == 0 && (8 * WN) % NW == 0, "tiles must split evenly over the waves");
    constexpr int NI = XPW + GPW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;  // ci block
    const int wn = wave % WN;  // co block
    const int g = lane >> 4;

    // logical id = ((split * co_tiles + co_tile) * ci_tiles + ci_tile) * taps + tap
    int wg = xcd_remap(blockIdx.x, a.tiles * a.splits * a.groups);
    if (wg >= a.tiles * a.splits * a.groups) return;  // grid padding (xcd_grid)
    const int group = wg / (a.tiles * a.splits);
    wg -= group * (a.tiles * a.splits);
    const int tap = wg % a.taps;
    wg /= a.taps;
    const int ci_tile = wg % a.ci_tiles;
    wg /= a.ci_tiles;
    const int co_tile = wg % a.co_tiles;
    const int split = wg / a.co_tiles;
    const int b_begin = split * a.b_per_split;
    int b_end = b_begin + a.b_per_split;
    if (b_end > a.batch) b_end = a.batch;
    const int n = (b_end - b_begin) * a.t_chunks;

    int xoff[XPW], goff_src[GPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) xoff[q] = dma_src_offset<WM>(wave * XPW + q, lane, a.x_rs);
#pragma unroll
    for (int q = 0; q < GPW; ++q) goff_src[q] = dma_src_offset<WN>(wave * GPW + q, lane, a.g_rs);
    const __bf16* xbase = a.x + group * a.x_gs + (long)(a.x_row0 + tap) * a.x_rs + ci_tile * TCI;
    const __bf16* gbase = a.g + group * a.g_gs + (long)a.g_row0 * a.g_rs + co_tile * TCO;

    // the tiles are requested in step order: the sources advance by pointer increments (one 64-row chunk down, or on to
    // the next utterance) instead of a per-step "step -> (utterance, chunk)" division and 64-bit multiplies -- with one
    // wave per SIMD (the 4-wave tile) that scalar clump sat between the barrier and the MFMAs with the matrix pipe empty
    const long x_step = (long)TK * a.x_rs, g_step = (long)TK * a.g_rs;
    const long x_wrap = (long)a.x_bs - (long)(a.t_chunks - 1) * x_step;
    const long g_wrap = (long)a.g_bs - (long)(a.t_chunks - 1) * g_step;
    const __bf16* xs_r = xbase + (long)b_begin * a.x_bs;
    const __bf16* gs_r = gbase + (long)b_begin * a.g_bs;
    int tc_r = 0;
    auto stage = [&](int buf) {
        char* xl = smem + buf * STAGE_BYTES + (wave * XPW) * 1024;
        char* gl = smem + buf * STAGE_BYTES + X_BYTES + (wave * GPW) * 1024;
#pragma unroll
        for (int q = 0; q < XPW; ++q) glds16(xs_r + xoff[q], xl + q * 1024);
#pragma unroll
        for (int q = 0; q < GPW; ++q) glds16(gs_r + goff_src[q], gl + q * 1024);
        const bool wrap = tc_r + 1 == a.t_chunks;
        xs_r += wrap ? x_wrap : x_step;
        gs_r += wrap ? g_wrap : g_step;
        tc_r = wrap ? 0 : tc_r + 1;
    };

    // ---- transpose-read addresses.  lane (g, i): row = kk*32 + g*8 + h*4 + (i>>2); 8-B piece (i&3) of 16-col block c16
    const int i16 = lane & 15;
    const int rkey = (i16 >> 2) | ((g & 1) << 2);
    const int rrow = g * 8 + (i16 >> 2);
    int xoffr[4], goffr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xoffr[j] = rrow * XRB + (((wm * 4 + j) ^ rkey) * 32) + (i16 & 3) * 8;
        goffr[j] = X_BYTES + rrow * GRB + (((wn * 4 + j) ^ rkey) * 32) + (i16 & 3) * 8;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(size_t)smem;
    // One 64-row step = 2 k-halves x 2 channel-halves = 4 quarters of 8 MFMAs; the fragments of the next quarter are
    // requested before the MFMAs of the current one (hand-counted waits), within the 128-VGPR budget of 16 waves:
    //   x fragments (all 4 of a k-half): xl/xh[k-half]; gradient fragments: two at a time in two alternating pairs
    auto read_x = [&](unsigned sl, auto kk_c, s16x4 (&bl)[4], s16x4 (&bh)[4]) {
        constexpr int KK = decltype(kk_c)::value;
#pragma unroll
        for (int it = 0; it < 4; ++it) tr_read8<KK * 32 * XRB, XRB>(bl[it], bh[it], sl + xoffr[it]);
    };
    auto read_g = [&](unsigned sl, auto kk_c, int jh, s16x4 (&al)[2], s16x4 (&ah)[2]) {
        constexpr int KK = decltype(kk_c)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) tr_read8<KK * 32 * GRB, GRB>(al[j], ah[j], sl + goffr[jh * 2 + j]);
    };
    auto mma_q = [&](int jh, s16x4 (&al)[2], s16x4 (&ah)[2], s16x4 (&bl)[4], s16x4 (&bh)[4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8 af = frag8(al[j], ah[j]);
#pragma unroll
            for (int it = 0; it < 4; ++it)
                acc[jh * 2 + j][it] =
                    SL_MFMA16(af, frag8(bl[it], bh[it]), acc[jh * 2 + j][it]);
        }
    };
    const std::integral_constant<int, 0> K0{};
    const std::integral_constant<int, 1> K1{};
    auto compute_step = [&](unsigned sl) {
        s16x4 xl[4], xh[4], pl[2], ph[2], ql[2], qh[2];
        read_x(sl, K0, xl, xh);
        read_g(sl, K0, 0, pl, ph);  // 12 reads
        read_g(sl, K0, 1, ql, qh);  // +4
        wait_ba<4>(xl, xh, pl, ph);
        mma_q(0, pl, ph, xl, xh);
        __builtin_amdgcn_sched_barrier(0);
        read_g(sl, K1, 0, pl, ph);  // p is free: its MFMAs have been issued
        wait_a<4>(ql, qh);
        mma_q(1, ql, qh, xl, xh);
        __builtin_amdgcn_sched_barrier(0);
        read_x(sl, K1, xl, xh);     // x fragments are single-buffered (128-VGPR budget): this read is the exposed one
        read_g(sl, K1, 1, ql, qh);
        wait_ba<4>(xl, xh, pl, ph);
        mma_q(0, pl, ph, xl, xh);
        __builtin_amdgcn_sched_barrier(0);
        wait_a<0>(ql, qh);
        mma_q(1, ql, qh, xl, xh);
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int i = 0; i < STAGES - 1; ++i)
        if (i < n) stage(i);
    int cur = 0, nxt = STAGES - 1;
    for (int i = 0; i < n; ++i) {
        if (i + STAGES - 1 <= n)
            wait_vmcnt<NI*(STAGES - 2)>();
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + STAGES - 1 < n) stage(nxt);
        const unsigned sl = lds0 + cur * STAGE_BYTES;
        compute_step(sl);
        cur = (cur + 1 == STAGES) ? 0 : cur + 1;
        nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
    }

    // ---- store: lane holds co = co_base + jn*16 + g*4 + {0..3} for ci = ci_base + it*16 + (lane & 15)
    float* out = a.splits > 1 ? a.out + ((long)group * a.splits + split) * a.split_stride : a.out + group * a.dw_gs;
    const int ci_base = ci_tile * TCI + wm * 64 + (lane & 15);
    const int co_base = co_tile * TCO + wn * 64 + g * 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long row = ((long)tap * a.cin + ci_base + it * 16) * a.cout;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) *(f32x4*)(out + row + co_base + jn * 16) = acc[jn][it];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Interleaved 256x256 variant: 8 waves, each a 128 (input channels) x 64 (output channels) patch = 32 accumulator tiles,
// two waves per SIMD with 256 registers each.  A 64-row step is two k-halves; the fragments of the other half (12
// transpose-read pairs) are requested one pair per two MFMAs, and after the barrier one LDS-DMA request of the next
// tile rides behind each of the first eight MFMA pairs -- the issue slots of the LDS / DMA instructions sit in the
// shadow of the matrix pipe (same scheme as IlvPhase in conv_nt_bf16.hip).  Fewer LDS bytes per MFMA than the 16-wave
// kernel (24 fragment reads per 64 MFMAs instead of 16 per 32).
struct TrFrags {
    s16x4 gl[4], gh[4];  // gradient fragments (MFMA A operand): 4 x 16 output channels
    s16x4 xl[8], xh[8];  // activation fragments (MFMA B operand): 8 x 16 input channels
};
__device__ __forceinline__ void wait_trfrags(TrFrags& f) {  // lgkmcnt(0) tied to all 24 half-fragments
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.gl[0]), "+v"(f.gh[0]), "+v"(f.gl[1]), "+v"(f.gh[1]), "+v"(f.gl[2]), "+v"(f.gh[2]), "+v"(f.gl[3]),
                   "+v"(f.gh[3]), "+v"(f.xl[0]), "+v"(f.xh[0]), "+v"(f.xl[1]), "+v"(f.xh[1]), "+v"(f.xl[2]), "+v"(f.xh[2]),
                   "+v"(f.xl[3]), "+v"(f.xh[3]), "+v"(f.xl[4]), "+v"(f.xh[4]), "+v"(f.xl[5]), "+v"(f.xh[5]), "+v"(f.xl[6]),
                   "+v"(f.xh[6]), "+v"(f.xl[7]), "+v"(f.xh[7]));
}

template <int KK, int Q, int NQ>
struct TrPhase {
    // KK = k-half the READS address (the MFMAs consume `cur`); group Q = MFMAs 2Q, 2Q+1 (jn = m / 8, it = m % 8)
    template <typename Hook>
    static __device__ __forceinline__ void run(f32x4 (&acc)[4][8], TrFrags& cur, TrFrags& nxt, const unsigned (&gaddr)[4],
                                               const unsigned (&xaddr)[8], const Hook& hook) {
        constexpr int GRB = 512, XRB = 512;
#if !defined(SL_WGRAD_PROBE_NO_MFMA)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = 2 * Q + j;
            const int jn = m / 8, it = m % 8;
            acc[jn][it] = SL_MFMA16(frag8(cur.gl[jn], cur.gh[jn]), frag8(cur.xl[it], cur.xh[it]),
                                                                  acc[jn][it]);
        }
#else
        asm volatile("" : "+v"(cur.gl[Q % 4]), "+v"(cur.xl[Q % 8]));
#endif
#if !defined(SL_WGRAD_PROBE_NO_READS)  // timing probes (wrong results by construction): tools/README.md, DESIGN.md section 4
        if constexpr (Q < 4)
            tr_read8<KK * 32 * GRB, GRB>(nxt.gl[Q], nxt.gh[Q], gaddr[Q]);
        else if constexpr (Q < 12)
            tr_read8<KK * 32 * XRB, XRB>(nxt.xl[Q - 4], nxt.xh[Q - 4], xaddr[Q - 4]);
#endif
#if defined(SL_WGRAD_PROBE_NO_MFMA)
        (void)0;
#endif
        hook(std::integral_constant<int, Q>{});
        __builtin_amdgcn_sched_barrier(0);
        TrPhase<KK, Q + 1, NQ>::template run<Hook>(acc, cur, nxt, gaddr, xaddr, hook);
    }
};
template <int KK, int NQ>
struct TrPhase<KK, NQ, NQ> {
    template <typename Hook>
    static __device__ __forceinline__ void run(f32x4 (&)[4][8], TrFrags&, TrFrags&, const unsigned (&)[4],
                                               const unsigned (&)[8], const Hook&) {}
};

__global__ __launch_bounds__(512, 2) void wgrad_tn_ilv_kernel(TnArgs a) {
    constexpr int NW = 8, STAGES = 2;
    constexpr int TCI = 256, TCO = 256;
    constexpr int XRB = 512, GRB = 512;
    constexpr int X_BYTES = TK * XRB;
    constexpr int STAGE_BYTES = TK * (XRB + GRB);
    constexpr int XPW = 32 / NW, GPW = 32 / NW;  // 1-KiB DMA instructions per wave per stage and operand

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if !defined(SL_NO_PRIO_YOUNG)
    // static priority for the second-dispatched half of the waves (MI355X_MICROARCH.md, "two waves per SIMD", item 4: the younger
    // wave of a SIMD loses every issue arbitration): same-box A/B of the config-3 step 2.2475 -> 2.2374 ms, three alternations
    // of the two builds, every one in favour (profiles/r05_prio_young_ab.txt)
    if (wave >= (int)(blockDim.x >> 7)) __builtin_amdgcn_s_setprio(1);
#endif
    const int wm = wave >> 2;  // 128-channel block of ci
    const int wn = wave & 3;   // 64-channel block of co
    const int g = lane >> 4;

    int wg = xcd_remap(blockIdx.x, a.tiles * a.splits * a.groups);
    if (wg >= a.tiles * a.splits * a.groups) return;  // grid padding (xcd_grid)
    const int group = wg / (a.tiles * a.splits);
    wg -= group * (a.tiles * a.splits);
    const int tap = wg % a.taps;
    wg /= a.taps;
    const int ci_tile = wg % a.ci_tiles;
    wg /= a.ci_tiles;
    const int co_tile = wg % a.co_tiles;
    const int split = wg / a.co_tiles;
    const int b_begin = split * a.b_per_split;
    int b_end = b_begin + a.b_per_split;
    if (b_end > a.batch) b_end = a.batch;
    const int n = (b_end - b_begin) * a.t_chunks;

    int xoff[XPW], goff_src[GPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) xoff[q] = dma_src_offset<4>(wave * XPW + q, lane, a.x_rs);
#pragma unroll
    for (int q = 0; q < GPW; ++q) goff_src[q] = dma_src_offset<4>(wave * GPW + q, lane, a.g_rs);
    const __bf16* xbase = a.x + group * a.x_gs + (long)(a.x_row0 + tap) * a.x_rs + ci_tile * TCI;
    const __bf16* gbase = a.g + group * a.g_gs + (long)a.g_row0 * a.g_rs + co_tile * TCO;

    // Request stream: sources of the NEXT tile, advanced by pointer increments (one 64-row chunk down, or on to the
    // next utterance); past the end the last tile is requested again into a dead slot.  The former per-step
    // "step -> (utterance, chunk)" division plus 64-bit multiplies was a ~55-instruction scalar clump between phase A
    // and the barrier, executed by all eight waves at the same moment with the matrix pipe empty.
    const __bf16* xs_n = nullptr;
    const __bf16* gs_n = nullptr;
    unsigned xl_n = 0, gl_n = 0;  // LDS byte offsets of this wave's part of the slot being refilled
    const long x_step = (long)TK * a.x_rs, g_step = (long)TK * a.g_rs;
    const long x_wrap = (long)a.x_bs - (long)(a.t_chunks - 1) * x_step;
    const long g_wrap = (long)a.g_bs - (long)(a.t_chunks - 1) * g_step;
    const __bf16* xs_r = xbase + (long)b_begin * a.x_bs;
    const __bf16* gs_r = gbase + (long)b_begin * a.g_bs;
    int tc_r = 0, left = n;
    const unsigned wave_x = (wave * XPW) * 1024, wave_g = X_BYTES + (wave * GPW) * 1024;
    auto begin_stage = [&](int buf) {
        xs_n = xs_r;
        gs_n = gs_r;
        xl_n = buf * STAGE_BYTES + wave_x;
        gl_n = buf * STAGE_BYTES + wave_g;
        const bool more = left > 1;
        const bool wrap = tc_r + 1 == a.t_chunks;
        long dx = wrap ? x_wrap : x_step;
        long dg = wrap ? g_wrap : g_step;
        dx = more ? dx : 0L;
        dg = more ? dg : 0L;
        xs_r += dx;
        gs_r += dg;
        tc_r = wrap ? 0 : tc_r + 1;
        left = more ? left - 1 : left;
    };
    auto dma_piece = [&](auto q_c) {
        constexpr int Q = decltype(q_c)::value;
#if !defined(SL_WGRAD_PROBE_NO_DMA)
        if constexpr (Q < XPW)
            glds16(xs_n + xoff[Q], smem + xl_n + Q * 1024);
        else if constexpr (Q < XPW + GPW)
            glds16(gs_n + goff_src[Q - XPW], smem + gl_n + (Q - XPW) * 1024);
#endif
    };
    const int i16 = lane & 15;
    const int rkey = (i16 >> 2) | ((g & 1) << 2);
    const int rrow = g * 8 + (i16 >> 2);
    const unsigned lds0 = (unsigned)(size_t)smem;
    unsigned xrel[8], grel[4];  // read offsets inside a stage (k-half 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) xrel[j] = rrow * XRB + (((wm * 8 + j) ^ rkey) * 32) + (i16 & 3) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) grel[j] = X_BYTES + rrow * GRB + (((wn * 4 + j) ^ rkey) * 32) + (i16 & 3) * 8;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (n > 0) {
#pragma unroll
        for (int i = 0; i < STAGES; ++i) {
            begin_stage(i);
#pragma unroll
            for (int q = 0; q < XPW; ++q) glds16(xs_n + xoff[q], smem + xl_n + q * 1024);
#pragma unroll
            for (int q = 0; q < GPW; ++q) glds16(gs_n + goff_src[q], smem + gl_n + q * 1024);
        }
        wait_vmcnt<(XPW + GPW) * (STAGES - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TrFrags f0, f1;
        unsigned ga[4], xa[8];  // fragment read addresses in the slot being read (k-half 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[j] = lds0 + grel[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[j] = lds0 + xrel[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_read8<0, GRB>(f0.gl[j], f0.gh[j], ga[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) tr_read8<0, XRB>(f0.xl[j], f0.xh[j], xa[j]);
        int cur = 0;
        int slot_delta = 0;  // byte distance from the slot phase A reads to the slot phase B (and the next phase A) reads
        // Phase A carries, in the shadow of its MFMA groups: the request-stream arithmetic (group 1) and, once its own
        // twelve fragment reads are out (groups 12-14), the move of the read addresses to the other slot.
        auto pin_stage = [&]() {
            begin_stage(cur);
            asm volatile("" : "+s"(xs_n), "+s"(gs_n), "+s"(xl_n), "+s"(gl_n), "+s"(xs_r), "+s"(gs_r));
        };
        auto move_g = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) ga[j] += slot_delta;
            asm volatile("" : "+v"(ga[0]), "+v"(ga[1]), "+v"(ga[2]), "+v"(ga[3]));
        };
        auto move_x = [&](int j0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[j0 + j] += slot_delta;
            asm volatile("" : "+v"(xa[j0]), "+v"(xa[j0 + 1]), "+v"(xa[j0 + 2]), "+v"(xa[j0 + 3]));
        };
        auto hook_a = [&](auto q_c) {
            constexpr int Q = decltype(q_c)::value;
            if constexpr (Q == 1) pin_stage();
            if constexpr (Q == 12) move_g();
            if constexpr (Q == 13) move_x(0);
            if constexpr (Q == 14) move_x(4);
        };
#if defined(SL_PROBE_SKEW)
        // TIMING PROBE (wrong results by construction): the second-dispatched half of
        // the waves runs ONE PHASE behind the first -- a barrier per phase instead of one per step, so that in every phase
        // one wave of a SIMD issues MFMAs on fragments it holds while the other issues its fragment reads / requests.  The
        // data hazards a real version would have to solve (requests into a slot the other half still reads: the request
        // stream split by k-half) are ignored here: same instruction stream, same bytes, the timing of the skewed loop.
        if (wave >= 4) __builtin_amdgcn_s_barrier();
#endif
        for (int i = 0; i < n; ++i) {
            const int nxt = cur ^ 1;
            slot_delta = (nxt - cur) * STAGE_BYTES;
            wait_trfrags(f0);
            TrPhase<1, 0, 16>::run(acc, f0, f1, ga, xa, hook_a);  // k-half 0 multiplies, k-half 1 is read
            wait_trfrags(f1);  // my reads of slot cur are complete
#if defined(SL_PROBE_SKEW)
            __builtin_amdgcn_s_barrier();
            wait_vmcnt<0>();
#else
            wait_vmcnt<0>();   // tile i+1 has landed (2-slot ring)
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            TrPhase<0, 0, 16>::run(acc, f1, f0, ga, xa, dma_piece);  // k-half 1 multiplies, next tile's k-half 0 is read
#if defined(SL_PROBE_SKEW)
            __builtin_amdgcn_s_barrier();
#endif
            cur = nxt;
        }
#if defined(SL_PROBE_SKEW)
        if (wave < 4) __builtin_amdgcn_s_barrier();
#endif
        wait_vmcnt<0>();   // the surplus requests still target this work-group's LDS
        wait_trfrags(f0);  // ... and the surplus fragment reads these registers
    }

    // ---- store: lane holds co = co_base + jn*16 + g*4 + {0..3} for ci = ci_base + it*16 + (lane & 15)
    float* out = a.splits > 1 ? a.out + ((long)group * a.splits + split) * a.split_stride : a.out + group * a.dw_gs;
    const int ci_base = ci_tile * TCI + wm * 128 + (lane & 15);
    const int co_base = co_tile * TCO + wn * 64 + g * 4;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const long row = ((long)tap * a.cin + ci_base + it * 16) * a.cout;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) *(f32x4*)(out + row + co_base + jn * 16) = acc[jn][it];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Deep-ring variant of the interleaved kernel (cfg stages 11): 32-frame stages, FOUR LDS slots of 32 KB, one barrier per
// stage.  Same tile (256 x 256, eight waves of 128 x 64), same fragment reads, same MFMA order inside a stage; what changes
// is the distance between a tile's request and its first read: three stages (~2 us) instead of one 64-frame step (~1.3 us).
// The launches whose operand tiles are new data at every step (big_conv_2, the inner layers: each tile is shared by only
// 4-8 work-groups of an XCD, so a step's tiles are compulsory L2 misses) wait for exactly that with the 2-slot ring:
// PMC (profiles/r03_pmc_wgrad_*.json) MFMA pipe busy 59 % / 53 % against 71 % for big_conv_1, whose operands stay in L2.
//   phase(j): MFMAs of stage j from registers | fragment reads of stage j + 1 | requests of stage j + 4 into slot j % 4
//   barrier(j + 1): stage j + 2 landed (vmcnt: two younger stages may stay in flight), everybody done reading slot (j+1) % 4
__global__ __launch_bounds__(512, 2) void wgrad_tn_ilv32_kernel(TnArgs a) {
    constexpr int NW = 8, SLOTS = 4, TS = 32;
    constexpr int TCI = 256, TCO = 256;
    constexpr int XRB = 512, GRB = 512;
    constexpr int X_BYTES = TS * XRB;               // 16 KB
    constexpr int STAGE_BYTES = TS * (XRB + GRB);   // 32 KB
    constexpr int XPW = 16 / NW, GPW = 16 / NW;     // 1-KiB DMA instructions per wave, stage and operand
    constexpr int NI = XPW + GPW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if !defined(SL_NO_PRIO_YOUNG)
    // static priority for the second-dispatched half of the waves (MI355X_MICROARCH.md, "two waves per SIMD", item 4: the younger
    // wave of a SIMD loses every issue arbitration): same-box A/B of the config-3 step 2.2475 -> 2.2374 ms, three alternations
    // of the two builds, every one in favour (profiles/r05_prio_young_ab.txt)
    if (wave >= (int)(blockDim.x >> 7)) __builtin_amdgcn_s_setprio(1);
#endif
    const int wm = wave >> 2;
    const int wn = wave & 3;
    const int g = lane >> 4;

    int wg = xcd_remap(blockIdx.x, a.tiles * a.splits * a.groups);
    if (wg >= a.tiles * a.splits * a.groups) return;
    const int group = wg / (a.tiles * a.splits);
    wg -= group * (a.tiles * a.splits);
    const int tap = wg % a.taps;
    wg /= a.taps;
    const int ci_tile = wg % a.ci_tiles;
    wg /= a.ci_tiles;
    const int co_tile = wg % a.co_tiles;
    const int split = wg / a.co_tiles;
    const int b_begin = split * a.b_per_split;
    int b_end = b_begin + a.b_per_split;
    if (b_end > a.batch) b_end = a.batch;
    const int tc32 = 2 * a.t_chunks;             // 32-frame stages per utterance
    const int n = (b_end - b_begin) * tc32;      // stages of this work-group (even)

    int xoff[XPW], goff_src[GPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) xoff[q] = dma_src_offset<4>(wave * XPW + q, lane, a.x_rs);
#pragma unroll
    for (int q = 0; q < GPW; ++q) goff_src[q] = dma_src_offset<4>(wave * GPW + q, lane, a.g_rs);
    const __bf16* xbase = a.x + group * a.x_gs + (long)(a.x_row0 + tap) * a.x_rs + ci_tile * TCI;
    const __bf16* gbase = a.g + group * a.g_gs + (long)a.g_row0 * a.g_rs + co_tile * TCO;

    const __bf16* xs_n = nullptr;
    const __bf16* gs_n = nullptr;
    unsigned xl_n = 0, gl_n = 0;
    const long x_step = (long)TS * a.x_rs, g_step = (long)TS * a.g_rs;
    const long x_wrap = (long)a.x_bs - (long)(tc32 - 1) * x_step;
    const long g_wrap = (long)a.g_bs - (long)(tc32 - 1) * g_step;
    const __bf16* xs_r = xbase + (long)b_begin * a.x_bs;
    const __bf16* gs_r = gbase + (long)b_begin * a.g_bs;
    int tc_r = 0, left = n;
    const unsigned wave_x = (wave * XPW) * 1024, wave_g = X_BYTES + (wave * GPW) * 1024;
    auto begin_stage = [&](int slot) {
        xs_n = xs_r;
        gs_n = gs_r;
        xl_n = slot * STAGE_BYTES + wave_x;
        gl_n = slot * STAGE_BYTES + wave_g;
        const bool more = left > 1;
        const bool wrap = tc_r + 1 == tc32;
        long dx = wrap ? x_wrap : x_step;
        long dg = wrap ? g_wrap : g_step;
        dx = more ? dx : 0L;
        dg = more ? dg : 0L;
        xs_r += dx;
        gs_r += dg;
        tc_r = wrap ? 0 : tc_r + 1;
        left = more ? left - 1 : left;
    };
    auto dma_piece = [&](auto q_c) {
        constexpr int Q = decltype(q_c)::value;
        if constexpr (Q < XPW)
            glds16(xs_n + xoff[Q], smem + xl_n + Q * 1024);
        else if constexpr (Q < XPW + GPW)
            glds16(gs_n + goff_src[Q - XPW], smem + gl_n + (Q - XPW) * 1024);
    };
    const int i16 = lane & 15;
    const int rkey = (i16 >> 2) | ((g & 1) << 2);
    const int rrow = g * 8 + (i16 >> 2);
    const unsigned lds0 = (unsigned)(size_t)smem;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (n > 0) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            begin_stage(i);
#pragma unroll
            for (int q = 0; q < XPW; ++q) glds16(xs_n + xoff[q], smem + xl_n + q * 1024);
#pragma unroll
            for (int q = 0; q < GPW; ++q) glds16(gs_n + goff_src[q], smem + gl_n + q * 1024);
        }
        wait_vmcnt<NI*(SLOTS - 1)>();  // stage 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TrFrags f0, f1;
        unsigned ga[4], xa[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[j] = lds0 + X_BYTES + rrow * GRB + (((wn * 4 + j) ^ rkey) * 32) + (i16 & 3) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[j] = lds0 + rrow * XRB + (((wm * 8 + j) ^ rkey) * 32) + (i16 & 3) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_read8<0, GRB>(f0.gl[j], f0.gh[j], ga[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) tr_read8<0, XRB>(f0.xl[j], f0.xh[j], xa[j]);
        // the read addresses point at the slot read NEXT (stage 1); every phase moves them one slot on behind its reads
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[j] += STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[j] += STAGE_BYTES;
        int rd = 1;           // slot the next phase reads
        int slot_delta = 0;   // set per phase: distance to the slot after that
        int rq = 0;           // slot the next phase refills (= the slot of the stage it multiplies)
        auto pin_stage = [&]() {
            begin_stage(rq);
            asm volatile("" : "+s"(xs_n), "+s"(gs_n), "+s"(xl_n), "+s"(gl_n), "+s"(xs_r), "+s"(gs_r));
        };
        auto move_g = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) ga[j] += slot_delta;
            asm volatile("" : "+v"(ga[0]), "+v"(ga[1]), "+v"(ga[2]), "+v"(ga[3]));
        };
        auto move_x = [&](int j0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[j0 + j] += slot_delta;
            asm volatile("" : "+v"(xa[j0]), "+v"(xa[j0 + 1]), "+v"(xa[j0 + 2]), "+v"(xa[j0 + 3]));
        };
        // groups 0 .. 3 of a phase: one request each (the stage's four instructions of this wave); group 4: the request
        // stream's arithmetic for the NEXT phase; groups 12 .. 14: the read addresses move on (their twelve reads are out)
        auto hook = [&](auto q_c) {
            constexpr int Q = decltype(q_c)::value;
            dma_piece(q_c);
            if constexpr (Q == 12) move_g();
            if constexpr (Q == 13) move_x(0);
            if constexpr (Q == 14) move_x(4);
        };
        auto phase = [&](TrFrags& cur, TrFrags& nxt) {
            wait_trfrags(cur);
            wait_vmcnt<NI * 2>();  // the stage the reads below address has landed; the two younger ones stay in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            slot_delta = rd == SLOTS - 1 ? -(SLOTS - 1) * STAGE_BYTES : STAGE_BYTES;
            pin_stage();  // requests of this phase go to slot rq
            TrPhase<0, 0, 16>::run(acc, cur, nxt, ga, xa, hook);
            rd = rd == SLOTS - 1 ? 0 : rd + 1;
            rq = rq == SLOTS - 1 ? 0 : rq + 1;
        };
        for (int i = 0; i < n; i += 2) {
            phase(f0, f1);
            phase(f1, f0);
        }
        wait_vmcnt<0>();   // the surplus requests still target this work-group's LDS
        wait_trfrags(f0);  // ... and the surplus fragment reads these registers
    }

    float* out = a.splits > 1 ? a.out + ((long)group * a.splits + split) * a.split_stride : a.out + group * a.dw_gs;
    const int ci_base = ci_tile * TCI + wm * 128 + (lane & 15);
    const int co_base = co_tile * TCO + wn * 64 + g * 4;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const long row = ((long)tap * a.cin + ci_base + it * 16) * a.cout;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) *(f32x4*)(out + row + co_base + jn * 16) = acc[jn][it];
    }
}

int launch_ilv32(const TnArgs& a, hipStream_t s) {
    constexpr int LDS_BYTES = 4 * 32 * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_tn_ilv32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    SL_LAUNCH_MAIN(wgrad_tn_ilv32_kernel, dim3(xcd_grid(a.tiles * a.splits * a.groups)), dim3(512), LDS_BYTES, s, a);
    return sl_check_launch("sl_conv1d_wgrad(bf16, interleaved, 4-slot ring)");
}

int launch_ilv(const TnArgs& a, hipStream_t s) {
    constexpr int LDS_BYTES = 2 * TK * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_tn_ilv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    SL_LAUNCH_MAIN(wgrad_tn_ilv_kernel, dim3(xcd_grid(a.tiles * a.splits * a.groups)), dim3(512), LDS_BYTES, s, a);
    return sl_check_launch("sl_conv1d_wgrad(bf16, interleaved)");
}

// ---------------------------------------------------------------------------------------------------------------------
// Several layers' weight gradients in ONE balanced launch (sl_conv1d_wgrad_multi).  The layers at the bottom of the stack
// have few 256 x 256 tiles (inner_conv_1..7: 49, striding_conv: 24): as launches of their own they need a batch split,
// whose granularity is an utterance -- 196 work-groups on 256 CUs for the grouped inner layers, 128 x 128 tiles at 30 % of
// the matrix peak for striding_conv, each followed by a reduction launch.  Here the tiles of ALL the jobs share one grid
// and the step sequence of a tile (utterance-major, 64-frame chunks) is cut into P = floor(2 * 256 / tiles) SEGMENTS whose
// boundaries are the same for every tile; a work-group takes two segments -- two consecutive ones of one tile (one item,
// one partial tile) or, when P is odd, the last segment of two neighbouring tiles (two items) -- so that every CU gets the
// same number of steps.  Work-groups that run side by side work on the same frames of neighbouring taps and share their
// operand tiles in the XCD's L2.  (A first version cut the concatenated (tile, step) sequence into equal ranges,
// "stream-K": balanced too, but neighbouring work-groups then sit at unrelated frames, nothing is shared and the launch
// became bandwidth-bound: 2.1 us per step instead of 1.4.)  Every item's partial tile goes to its own slot and one
// table-free reduction launch adds a tile's slots in a fixed order (deterministic).  The step loop is the interleaved
// 8-wave loop of wgrad_tn_ilv_kernel, unchanged.
constexpr int MULTI_MAX_JOBS = 16;
struct MultiJob {
    const __bf16* x;
    const __bf16* g;
    float* dw;
    int taps, cin, cout, ci_tiles, co_tiles;
    int x_row0, x_rs, g_row0, g_rs;
    long x_bs, g_bs;
    int tile_begin;  // first tile of this job in the common tile sequence
};
struct MultiArgs {
    MultiJob job[MULTI_MAX_JOBS];
    int n_jobs, total_tiles;
    int batch, t_chunks, spt;  // steps per tile = batch * t_chunks (the same for every job)
    int segs;                  // P: segments per tile
    int pair_wgs;              // total_tiles * (P / 2) work-groups take a pair of segments of one tile ...
    int workers;               // ... the others (P odd) the last segment of two neighbouring tiles
    float* slots;              // [workers][2][256 * 256] partial tiles
};
__device__ __forceinline__ int multi_seg_begin(const MultiArgs& a, int p) { return (int)((long)a.spt * p / a.segs); }

__global__ __launch_bounds__(512, 2) void wgrad_tn_ilv_multi_kernel(MultiArgs a) {
    constexpr int NW = 8, STAGES = 2;
    constexpr int XRB = 512, GRB = 512;
    constexpr int X_BYTES = TK * XRB;
    constexpr int STAGE_BYTES = TK * (XRB + GRB);
    constexpr int XPW = 32 / NW, GPW = 32 / NW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wm = wave >> 2;
    const int wn = wave & 3;

    if (xcd_remap(blockIdx.x, a.workers) >= a.workers) return;  // grid padding (xcd_grid)
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned wave_x = (wave * XPW) * 1024, wave_g = X_BYTES + (wave * GPW) * 1024;

    for (int item = 0; item < 2; ++item) {
        // ---- this work-group's item: one or two per work-group, derived from the block id again for the second one (all
        // wave-uniform and pinned into scalar registers; kept live across the step loop they would not fit: this kernel
        // fills the scalar register file as well as the vector one)
        const int ws = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, a.workers));
        const bool pair = ws < a.pair_wgs;
        const int tile = __builtin_amdgcn_readfirstlane(pair ? ws % a.total_tiles : 2 * (ws - a.pair_wgs) + item);
        if (item == 1 && (pair || tile >= a.total_tiles)) break;
        const int k0 = __builtin_amdgcn_readfirstlane(pair ? ws / a.total_tiles : 0);
        const int off = __builtin_amdgcn_readfirstlane(multi_seg_begin(a, pair ? 2 * k0 : a.segs - 1));
        const int n = __builtin_amdgcn_readfirstlane(multi_seg_begin(a, pair ? 2 * k0 + 2 : a.segs)) - off;
        // Lane-derived values are rebuilt per item (and once more for the epilogue) from an opaque copy of the thread id:
        // hoisted out of the item loop they would stay live across the step loop, which has two registers to spare.
        int zero_i = 0;  // (the lane id from the hardware's mbcnt, not from threadIdx.x: v0 need not stay alive either)
        asm volatile("" : "+v"(zero_i));
        const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero_i));
        const int g = lane >> 4;
        const int i16 = lane & 15;
        const int rkey = (i16 >> 2) | ((g & 1) << 2);
        const int rrow = g * 8 + (i16 >> 2);
        int ji = 0;
        while (ji + 1 < a.n_jobs && tile >= a.job[ji + 1].tile_begin) ++ji;
        const MultiJob& J = a.job[ji];
        int lt = tile - J.tile_begin;
        const int tap = __builtin_amdgcn_readfirstlane(lt % J.taps);
        lt = __builtin_amdgcn_readfirstlane(lt / J.taps);
        const int ci_tile = __builtin_amdgcn_readfirstlane(lt % J.ci_tiles);
        const int co_tile = __builtin_amdgcn_readfirstlane(lt / J.ci_tiles);
        const int b_begin = __builtin_amdgcn_readfirstlane(off / a.t_chunks), tc_begin = off - b_begin * a.t_chunks;
        const int x_rs = J.x_rs, g_rs = J.g_rs;

        int xoff[XPW], goff_src[GPW];
#pragma unroll
        for (int q = 0; q < XPW; ++q) xoff[q] = dma_src_offset<4>(wave * XPW + q, lane, x_rs);
#pragma unroll
        for (int q = 0; q < GPW; ++q) goff_src[q] = dma_src_offset<4>(wave * GPW + q, lane, g_rs);
        // (32-bit strides: the host checks that an utterance's tensor has fewer than 2^31 elements; the scalar register
        // file is as full as the vector one in this kernel)
        const int x_step = TK * x_rs, g_step = TK * g_rs;
        const int x_wrap = (int)J.x_bs - (a.t_chunks - 1) * x_step;
        const int g_wrap = (int)J.g_bs - (a.t_chunks - 1) * g_step;
        // (a channel count that is not a multiple of 256 -- the 640-wide pair view of 257 bins -- ends with a tile that starts
        // at cin - 256 and overlaps its neighbour; the reduction writes the overlapped rows from the neighbour only)
        const int ci_start = __builtin_amdgcn_readfirstlane(min(ci_tile * 256, J.cin - 256));
        const __bf16* xs_r = J.x + (long)(J.x_row0 + tap) * x_rs + ci_start + (long)b_begin * J.x_bs + (long)tc_begin * x_step;
        const __bf16* gs_r = J.g + (long)J.g_row0 * g_rs + co_tile * 256 + (long)b_begin * J.g_bs + (long)tc_begin * g_step;
        const __bf16* xs_n = nullptr;
        const __bf16* gs_n = nullptr;
        unsigned xl_n = 0, gl_n = 0;
        int tc_r = tc_begin, left = n;
        auto begin_stage = [&](int buf) {
            xs_n = xs_r;
            gs_n = gs_r;
            xl_n = buf * STAGE_BYTES + wave_x;
            gl_n = buf * STAGE_BYTES + wave_g;
            const bool more = left > 1;
            const bool wrap = tc_r + 1 == a.t_chunks;
            int dx = wrap ? x_wrap : x_step;
            int dg = wrap ? g_wrap : g_step;
            dx = more ? dx : 0;
            dg = more ? dg : 0;
            xs_r += dx;
            gs_r += dg;
            tc_r = wrap ? 0 : tc_r + 1;
            left = more ? left - 1 : left;
        };
        auto dma_piece = [&](auto q_c) {
            constexpr int Q = decltype(q_c)::value;
            if constexpr (Q < XPW)
                glds16(xs_n + xoff[Q], smem + xl_n + Q * 1024);
            else if constexpr (Q < XPW + GPW)
                glds16(gs_n + goff_src[Q - XPW], smem + gl_n + (Q - XPW) * 1024);
        };

        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if (item > 0) {  // the previous item's last fragment reads must be over before its slots are refilled
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < STAGES; ++i) {
            begin_stage(i);
#pragma unroll
            for (int q = 0; q < XPW; ++q) glds16(xs_n + xoff[q], smem + xl_n + q * 1024);
#pragma unroll
            for (int q = 0; q < GPW; ++q) glds16(gs_n + goff_src[q], smem + gl_n + q * 1024);
        }
        wait_vmcnt<(XPW + GPW) * (STAGES - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TrFrags f0, f1;
        unsigned ga[4], xa[8];  // (rebuilt per item: kept live across items they cost the twelve registers that spill)
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[j] = lds0 + X_BYTES + rrow * GRB + (((wn * 4 + j) ^ rkey) * 32) + (i16 & 3) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[j] = lds0 + rrow * XRB + (((wm * 8 + j) ^ rkey) * 32) + (i16 & 3) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_read8<0, GRB>(f0.gl[j], f0.gh[j], ga[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) tr_read8<0, XRB>(f0.xl[j], f0.xh[j], xa[j]);
        int cur = 0;
        int slot_delta = 0;
        auto pin_stage = [&]() {
            begin_stage(cur);
            asm volatile("" : "+s"(xs_n), "+s"(gs_n), "+s"(xl_n), "+s"(gl_n), "+s"(xs_r), "+s"(gs_r));
        };
        auto move_g = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) ga[j] += slot_delta;
            asm volatile("" : "+v"(ga[0]), "+v"(ga[1]), "+v"(ga[2]), "+v"(ga[3]));
        };
        auto move_x = [&](int j0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[j0 + j] += slot_delta;
            asm volatile("" : "+v"(xa[j0]), "+v"(xa[j0 + 1]), "+v"(xa[j0 + 2]), "+v"(xa[j0 + 3]));
        };
        auto hook_a = [&](auto q_c) {
            constexpr int Q = decltype(q_c)::value;
            if constexpr (Q == 1) pin_stage();
            if constexpr (Q == 12) move_g();
            if constexpr (Q == 13) move_x(0);
            if constexpr (Q == 14) move_x(4);
        };
        for (int i = 0; i < n; ++i) {
            const int nxt = cur ^ 1;
            slot_delta = (nxt - cur) * STAGE_BYTES;
            wait_trfrags(f0);
            TrPhase<1, 0, 16>::run(acc, f0, f1, ga, xa, hook_a);
            wait_trfrags(f1);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            TrPhase<0, 0, 16>::run(acc, f1, f0, ga, xa, dma_piece);
            cur = nxt;
        }
        wait_vmcnt<0>();
        wait_trfrags(f0);

        // ---- this item's partial tile: dense [256 ci][256 co] floats in slot (w, item)
        float* out = a.slots + ((long)__builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, a.workers)) * 2 + item) * (256 * 256);
        int zero_e = 0;
        asm volatile("" : "+v"(zero_e));
        const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero_e));
        const int ci_l = wm * 128 + (lane_e & 15);
        const int co_l = wn * 64 + (lane_e >> 4) * 4;
#pragma unroll
        for (int it = 0; it < 8; ++it)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) *(f32x4*)(out + (ci_l + it * 16) * 256 + co_l + jn * 16) = acc[jn][it];
    }
}

// dw tile = sum of its items' slots: the pair items in segment order, then the single last segment (P odd).
// grid (64, total_tiles), 256 threads x one float4
__global__ __launch_bounds__(256) void wgrad_multi_reduce_kernel(MultiArgs a) {
    const int tile = blockIdx.y;
    int ji = 0;
    while (ji + 1 < a.n_jobs && tile >= a.job[ji + 1].tile_begin) ++ji;
    const MultiJob& J = a.job[ji];
    int lt = tile - J.tile_begin;
    const int tap = lt % J.taps;
    lt /= J.taps;
    const int ci_tile = lt % J.ci_tiles, co_tile = lt / J.ci_tiles;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;  // float index inside the 256 x 256 tile
    const int ci = e >> 8, co = e & 255;
    const int pairs = a.segs / 2;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < pairs; ++k) {
        const f32x4 v = *(const f32x4*)(a.slots + ((long)(k * a.total_tiles + tile) * 2) * (256 * 256) + e);
        s = k == 0 ? v : s + v;
    }
    if (a.segs & 1) {
        const f32x4 v = *(const f32x4*)(a.slots + ((long)(a.pair_wgs + tile / 2) * 2 + (tile & 1)) * (256 * 256) + e);
        s = pairs == 0 ? v : s + v;
    }
    const int ci_start = min(ci_tile * 256, J.cin - 256);  // last tile of a channel count that is not a multiple of 256
    if (ci_start + ci < ci_tile * 256) return;               // ... its overlap is the neighbouring tile's to write
    *(f32x4*)(J.dw + ((long)tap * J.cin + ci_start + ci) * J.cout + co_tile * 256 + co) = s;
}

template <int WM, int WN, int STAGES>
int launch(const TnArgs& a, hipStream_t s) {
    constexpr int LDS_BYTES = STAGES * TK * 128 * (WM + WN);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS ring exceeds the 160 KiB of a CU");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_tn_bf16_kernel<WM, WN, STAGES>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    SL_LAUNCH_MAIN((wgrad_tn_bf16_kernel<WM, WN, STAGES>), dim3(xcd_grid(a.tiles * a.splits * a.groups)), dim3(64 * WM * WN),
                       LDS_BYTES, s, a);
    return sl_check_launch("sl_conv1d_wgrad(bf16)");
}

struct WCfg {
    int wm, wn, stages, splits;  // splits = 0: choose
};

WCfg decode_wcfg(int cfg) { return WCfg{cfg & 15, (cfg >> 4) & 15, (cfg >> 8) & 15, (cfg >> 12) & 255}; }

int choose_splits(const sl_conv_geom* g, int tci, int tco, int target_wgs, int groups = 1) {
    const long tiles = (long)g->taps * (g->cin / tci) * (g->cout / tco) * groups;
    target_wgs = target_wgs * sl_cus() / 256;  // (the callers' targets are for the whole chip: sl_set_available_cus)
    long want = (target_wgs + tiles - 1) / tiles;
    if (want < 1) want = 1;
    if (want > g->batch) want = g->batch;
    // Split counts that divide the batch give equal work per work-group (measured: 4 x 8 utterances 0.144 ms vs 6 uneven
    // splits 0.179 ms on the grouped inner-layer launch).  Among the divisors up to 2 * want take the one with the least
    // (rounds of `target_wgs` resident work-groups) x (utterances per split); ties go to the smaller split count (less
    // partial-sum traffic).  B = 48, 49 tiles: 4 splits (196 work-groups, one round), not 6 (294 = two rounds).
    int best = 1;
    double best_cost = 1e30;
    for (int d = 1; d <= g->batch && d <= 2 * want; ++d) {
        if (g->batch % d) continue;
        const double rounds = (double)((tiles * d + target_wgs - 1) / target_wgs);
        const double cost = rounds * (g->batch / d);
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = d;
        }
    }
    return best;
}

WCfg auto_wcfg(const sl_conv_geom* g, int groups) {
    // measured on MI355X with tools/tune_kernels.py (latest copy: profiles/r01j_tune_kernels.json), see HISTORY.md section 3
    if (g->cin % 256 == 0 && g->cout % 256 == 0) {
        const long tiles256 = (long)g->taps * (g->cin / 256) * (g->cout / 256) * groups;
        // 256x256 tile, 8 waves, interleaved stream: big_conv_1 0.329 ms = 1.55 PFLOP/s (16-wave kernel 0.374, 128x128
        // tiles 0.45); the grouped inner-layer launch 0.105 ms (16-wave 0.123); big_conv_2 (64 such tiles, batch split
        // 4) 0.106 ms against 0.108 for 128x128 tiles.  striding_conv (128 input channels) keeps the 128x128 tiles.
        if (tiles256 >= 64 || (groups > 1 && tiles256 >= 48)) return WCfg{4, 4, 10, choose_splits(g, 256, 256, 256, groups)};
    }
    // short layers: 128x128 tiles, batch split so that ~2 work-groups land on every CU (deeper rings measured no gain)
    const int splits = choose_splits(g, 128, 128, 512, groups);
    // one column of tiles streaming a wide input (output_conv: 16 tiles, HBM-bound): a 3-slot ring and half the splits
    // (fewer partial sums to reduce) measured 26.0 vs 30.2 us
    if (groups == 1 && g->taps == 1 && g->cout == 128 && splits >= 2 && splits % 2 == 0 && g->cin >= 1024)
        return WCfg{2, 2, 3, splits / 2};
    return WCfg{2, 2, 2, splits};
}

bool valid_wcfg(const WCfg& c, const sl_conv_geom* g) {
    const bool shape = (c.wm == 2 && c.wn == 2 && c.stages >= 2 && c.stages <= 4) ||
                       (c.wm == 4 && c.wn == 2 && (c.stages == 2 || c.stages == 3)) ||
                       (c.wm == 2 && c.wn == 4 && (c.stages == 2 || c.stages == 3)) ||
                       (c.wm == 4 && c.wn == 4 && (c.stages == 2 || c.stages == 10 || c.stages == 11));  // 10: 8-wave interleaved kernel, 11: its 4-slot ring variant
    return shape && g->cin % (64 * c.wm) == 0 && g->cout % (64 * c.wn) == 0 && c.splits >= 1 && c.splits <= g->batch;
}

WCfg resolve_wcfg(const sl_conv_geom* g, int cfg, int groups) {
    if (cfg == 0) return auto_wcfg(g, groups);
    WCfg c = decode_wcfg(cfg);
    if (c.splits == 0 && c.wm > 0 && c.wn > 0 && g->cin % (64 * c.wm) == 0 && g->cout % (64 * c.wn) == 0)
        c.splits = choose_splits(g, 64 * c.wm, 64 * c.wn, c.wm * c.wn >= 16 ? 256 : 512, groups);
    return c;
}

}  // namespace

#if defined(SL_ELEM_F16)  // the second translation unit of this file: the same kernels on v_mfma_f32_16x16x32_f16
#define wgrad_split_count wgrad_split_count_f16
#define wgrad_reduce wgrad_reduce_f16
#define wgrad_tn_bf16_workspace_bytes wgrad_tn_f16_workspace_bytes
#define wgrad_tn_bf16 wgrad_tn_f16
#define wgrad_multi_bf16_workspace_bytes wgrad_multi_f16_workspace_bytes
#define wgrad_multi_bf16 wgrad_multi_f16
#define wgrad_reduce_kernel wgrad_reduce_kernel_f16
#define wgrad_reduce_grouped_kernel wgrad_reduce_grouped_kernel_f16
#endif

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
    sl_conv_geom gg = *g;
    return choose_splits(&gg, tile, tile, 512);
}

int wgrad_reduce(const float* ws, float* dw, long n, int splits, hipStream_t s) {
    const long n4 = n / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ws, dw, n4, splits,
                       n4);
    return sl_check_launch("wgrad_reduce");
}

// grouped tail: grid.y = group; sums that group's split partials in a fixed order into dw + group*dw_gs
__global__ void wgrad_reduce_grouped_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int splits,
                                            long dw_gs) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4* p = (const f32x4*)ws + (long)blockIdx.y * splits * n4 + i;
    f32x4 s = p[0];
    for (int k = 1; k < splits; ++k) s += p[k * n4];
    ((f32x4*)(dw + blockIdx.y * dw_gs))[i] = s;
}

size_t wgrad_tn_bf16_workspace_bytes(const sl_conv_geom* g, int cfg, int groups) {
    const WCfg c = resolve_wcfg(g, cfg, groups);
    if (!valid_wcfg(c, g)) return 0;
    const int bps = (g->batch + c.splits - 1) / c.splits;
    const int splits = (g->batch + bps - 1) / bps;
    return splits > 1 ? (size_t)groups * splits * g->taps * g->cin * g->cout * sizeof(float) : 0;
}

int wgrad_tn_bf16(const void* x, const void* gr, float* dw, const sl_conv_geom* g, int cfg, int groups, long x_gs,
                  long g_gs, long dw_gs, float* ws, size_t ws_bytes, hipStream_t s) {
    const WCfg c = resolve_wcfg(g, cfg, groups);
    if (!valid_wcfg(c, g) || groups < 1) {
        sl_set_error("sl_conv1d_wgrad(bf16): invalid tile configuration wm=%d wn=%d stages=%d splits=%d for cin=%d cout=%d",
                     c.wm, c.wn, c.stages, c.splits, g->cin, g->cout);
        return SL_ERR_INVALID_ARGUMENT;
    }
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
    a.ci_tiles = g->cin / (64 * c.wm);
    a.co_tiles = g->cout / (64 * c.wn);
    a.tiles = a.taps * a.ci_tiles * a.co_tiles;
    a.b_per_split = (g->batch + c.splits - 1) / c.splits;
    a.splits = (g->batch + a.b_per_split - 1) / a.b_per_split;
    a.split_stride = (long)g->taps * g->cin * g->cout;
    a.groups = groups;
    a.x_gs = x_gs;
    a.g_gs = g_gs;
    a.dw_gs = dw_gs;
    if (a.splits > 1 && (ws == nullptr || ws_bytes < (size_t)groups * a.splits * a.split_stride * sizeof(float))) {
        sl_set_error("sl_conv1d_wgrad(bf16): workspace too small");
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    a.out = a.splits > 1 ? ws : dw;
    int rc = SL_ERR_UNSUPPORTED;
#define SL_TN_CASE(WM_, WN_, ST_) \
    if (c.wm == WM_ && c.wn == WN_ && c.stages == ST_) rc = launch<WM_, WN_, ST_>(a, s);
    SL_TN_CASE(2, 2, 2)
    SL_TN_CASE(2, 2, 3)
    SL_TN_CASE(2, 2, 4)
    SL_TN_CASE(4, 2, 2)
    SL_TN_CASE(4, 2, 3)
    SL_TN_CASE(2, 4, 2)
    SL_TN_CASE(2, 4, 3)
    SL_TN_CASE(4, 4, 2)
#undef SL_TN_CASE
    if (c.wm == 4 && c.wn == 4 && c.stages == 10) rc = launch_ilv(a, s);
    if (c.wm == 4 && c.wn == 4 && c.stages == 11) rc = launch_ilv32(a, s);
    if (rc != SL_OK) return rc;
    if (a.splits > 1) {
        const long n4 = a.split_stride / 4;
        hipLaunchKernelGGL(wgrad_reduce_grouped_kernel, dim3((unsigned)((n4 + 255) / 256), groups), dim3(256), 0, s, ws,
                           dw, n4, a.splits, dw_gs);
        return sl_check_launch("wgrad_reduce");
    }
    return SL_OK;
}

// ---- sl_conv1d_wgrad_multi: see wgrad_tn_ilv_multi_kernel
static int multi_fill(const sl_wgrad_job* jobs, int n_jobs, MultiArgs* a) {
    if (n_jobs < 1 || n_jobs > MULTI_MAX_JOBS) {
        sl_set_error("sl_conv1d_wgrad_multi: 1..%d jobs per call", MULTI_MAX_JOBS);
        return SL_ERR_INVALID_ARGUMENT;
    }
    a->n_jobs = n_jobs;
    a->batch = jobs[0].geom.batch;
    a->t_chunks = (jobs[0].geom.t_out + TK - 1) / TK;
    a->spt = a->batch * a->t_chunks;
    int tiles = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const sl_conv_geom& g = jobs[i].geom;
        if (g.batch != a->batch || g.t_out != jobs[0].geom.t_out || g.batch <= 0 || g.t_out <= 0 || g.taps <= 0 ||
            g.cin < 256 || g.cout <= 0 || g.cin % 64 || g.cout % 256 || g.x_row_stride % 8 || g.y_row_stride % 8 ||
            g.x_row_stride < g.cin || g.y_row_stride < g.cout || g.x_batch_stride >= (1LL << 31) ||
            g.y_batch_stride >= (1LL << 31)) {
            sl_set_error("sl_conv1d_wgrad_multi: job %d: all jobs need the same batch and t_out, output channels in multiples of "
                         "256, at least 256 input channels in multiples of 64 and row strides that are multiples of 8", i);
            return SL_ERR_UNSUPPORTED;
        }
        MultiJob& J = a->job[i];
        J.x = (const __bf16*)jobs[i].x;
        J.g = (const __bf16*)jobs[i].g;
        J.dw = jobs[i].dw;
        J.taps = g.taps;
        J.cin = g.cin;
        J.cout = g.cout;
        J.ci_tiles = (g.cin + 255) / 256;
        J.co_tiles = g.cout / 256;
        J.x_row0 = g.x_row0;
        J.x_rs = g.x_row_stride;
        J.x_bs = g.x_batch_stride;
        J.g_row0 = g.y_row0;
        J.g_rs = g.y_row_stride;
        J.g_bs = g.y_batch_stride;
        J.tile_begin = tiles;
        tiles += g.taps * J.ci_tiles * J.co_tiles;
    }
    a->total_tiles = tiles;
    // P aligned segments per tile, two per work-group, about one work-group per CU (the interleaved kernel takes a whole CU)
    int segs = 2 * sl_cus() / tiles;
    if (segs > a->spt) segs = a->spt;
    if (segs < 1) segs = 1;
    a->segs = segs;
    a->pair_wgs = tiles * (segs / 2);
    a->workers = a->pair_wgs + ((segs & 1) ? (tiles + 1) / 2 : 0);
    return SL_OK;
}

size_t wgrad_multi_bf16_workspace_bytes(const sl_wgrad_job* jobs, int n_jobs) {
    MultiArgs a;
    if (multi_fill(jobs, n_jobs, &a) != SL_OK) return 0;
    // an upper bound that does not depend on the frame count (the segment count -- and with it the number of work-groups --
    // changes with it for very short batches): tiles * floor(P / 2) + ceil(tiles / 2) <= 256 + tiles work-groups
    return (size_t)(256 + a.total_tiles) * 2 * 256 * 256 * sizeof(float);
}

int wgrad_multi_bf16(const sl_wgrad_job* jobs, int n_jobs, void* ws, size_t ws_bytes, hipStream_t s) {
    MultiArgs a;
    int rc = multi_fill(jobs, n_jobs, &a);
    if (rc != SL_OK) return rc;
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs[i].x || !jobs[i].g || !jobs[i].dw) {
            sl_set_error("sl_conv1d_wgrad_multi: null pointer in job %d", i);
            return SL_ERR_INVALID_ARGUMENT;
        }
    if (ws == nullptr || ws_bytes < (size_t)a.workers * 2 * 256 * 256 * sizeof(float)) {
        sl_set_error("sl_conv1d_wgrad_multi: workspace too small");
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    a.slots = (float*)ws;
    constexpr int LDS_BYTES = 2 * TK * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wgrad_tn_ilv_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    SL_LAUNCH_MAIN(wgrad_tn_ilv_multi_kernel, dim3(xcd_grid(a.workers)), dim3(512), LDS_BYTES, s, a);
    rc = sl_check_launch("sl_conv1d_wgrad_multi");
    if (rc != SL_OK) return rc;
    hipLaunchKernelGGL(wgrad_multi_reduce_kernel, dim3(64, a.total_tiles), dim3(256), 0, s, a);
    return sl_check_launch("sl_conv1d_wgrad_multi(reduce)");
}
