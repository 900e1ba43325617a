// conv_chain_bf16.hip -- a RUN of identical stride-1 convolutions (the seven 250 -> 250 channel, 7-tap inner_conv_i of
// the Wav2Letter stack, reference speechless/net.py:321-323) in ONE launch, forward (bias + ReLU) or input-gradient
// (ReLU mask) direction, with the activations of a time tile held in LDS from layer to layer.
//
// Why: as single launches these layers are 14 GFLOP each -- 5.6 us at the bf16 MFMA peak -- and pay a fixed 6-8 us per
// launch (launch, first-touch of weights and activations that the previous kernel just wrote, pipeline fill and drain)
// plus an L2 -> LDS stream of 32 KB per 64-deep step that is as long as the step's MFMAs: 23.8 us forward / 26 us
// input gradient inside the step = 22-24 % of the peak, 0.37 ms of a 2.28 ms step for 8.5 % of its FLOPs
// (VERDICT r1 item 3).  Here a work-group owns 64 consecutive output frames of one utterance through the WHOLE run:
//   * it loads the input rows it needs ONCE (64 + 2 * pad * layers rows x 256 channels -> LDS), computes every layer
//     for its rows plus the halo the following layers still need (the halo is recomputed, not exchanged: +39 % MFMA
//     work at seven layers, no inter-work-group synchronisation), and hands a layer's output to the next one through
//     LDS (bf16, exactly the value that is also stored to HBM for the backward pass);
//   * per 64-deep step it streams only the 256 x 64 weight tile (32 KB) through a 3-slot LDS-DMA ring
//     (global_load_lds, counted vmcnt), two steps ahead and across layer boundaries;
//   * eight waves = two per SIMD; wave w owns output channels [32w, 32w + 32) for all rows: per step 2 weight fragments
//     + up to 7 activation fragments per k-half feed up to 28 MFMAs (v_mfma_f32_16x16x32_bf16), D^T orientation (MFMA rows
//     = channels) so that a lane ends with 4 consecutive channels of one time row (8-byte LDS stores; the HBM side is
//     row-contiguous from the LDS copy, 16 bytes per lane).
//   * a wave requests exactly the weight rows it reads itself, so the ring needs NO barrier: the waves drift apart inside a
//     layer and one wave's request / read phase runs under its SIMD partner's MFMAs; barriers only at the layer ends.
// Measured (config 3, tools/chain_time.py, tools/chain_stamps.py; history in HISTORY.md section 3.1a): forward 124 us against
// 7 x 23.6 = 165 us of single launches, input gradients 126 against 185 us.  A step is 930 cycles against 713 of MFMA issue.
// LDS: activations 4 chunk-slabs x 120 rows x 128 B = 60 KB (row r, 16-byte slot s stored at slot s ^ (r & 7): a fragment
// read at any tap offset stays conflict-free, as in the slab kernel of conv_nt_bf16.hip) + 3 x 32 KB weight slots.
// Rows outside [0, T') are forced to zero after every layer (SAME padding: relu(bias) is not zero).
#include "common.h"

#include <type_traits>

namespace {

constexpr int CH = 256;          // channels in = out (padded)
constexpr int TM = 64;           // output frames per work-group (TM_SMALL where that fills the chip's rounds better: chain_tile_rows)
constexpr int TM_SMALL = 48;
constexpr int MAX_LAYERS = 8;
constexpr int ACT_ROWS = 120;    // >= 16 * 7 + taps - 1
constexpr int ACT_CHUNK = ACT_ROWS * 128;
constexpr int ACT_BYTES = 4 * ACT_CHUNK;  // 61 440
constexpr int WSLOT = CH * 128;           // 32 768
constexpr int NSLOT = 3;
constexpr int CHAIN_LDS = ACT_BYTES + NSLOT * WSLOT;  // 159 744 <= 160 KiB
constexpr int MT_MAX = 7;

struct ChainArgs {
    const __bf16* x;                  // input of the first layer of the run
    __bf16* y[MAX_LAYERS];            // outputs (forward: activations; dgrad: gradients), same geometry as x
    const __bf16* w[MAX_LAYERS];      // packed weights [256][taps][256] (forward: w_fwd; dgrad: w_dgrad, taps flipped)
    const float* bias[MAX_LAYERS];    // forward
    const __bf16* mask[MAX_LAYERS];   // dgrad: the stored activation whose sign is the ReLU mask, same geometry
    int n_layers, taps, pad;          // pad = rows of left context per layer (forward: pad_left; dgrad: pad_right)
    int batch, t_out, t_tiles;
    int tm;                           // output frames per work-group: TM or TM_SMALL
    int row0, rs;                     // halo rows in front of frame 0, elements per row
    long bs;                          // elements per utterance
};

__device__ __forceinline__ void chain_glds16(const __bf16* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const SL_GLOBAL void*)gsrc, (SL_LDS void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void chain_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int OFF>
__device__ __forceinline__ void chain_ds_read128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int I, int N>
struct ChainReadRun {  // N reads, 16 rows (2048 bytes) apart
    static __device__ __forceinline__ void go(bf16x8 (&f)[N], unsigned addr) {
        chain_ds_read128<I * 2048>(f[I], addr);
        ChainReadRun<I + 1, N>::go(f, addr);
    }
};
template <int N>
struct ChainReadRun<N, N> {
    static __device__ __forceinline__ void go(bf16x8 (&)[N], unsigned) {}
};
// s_waitcnt lgkmcnt(CNT) that the MFMAs consuming these fragments cannot be hoisted above
template <int MT>
struct ChainWait;
template <>
struct ChainWait<3> {
    template <int CNT>
    static __device__ __forceinline__ void frags(bf16x8 (&w)[2], bf16x8 (&x)[3]) {
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]) : "n"(CNT));
    }
};
template <>
struct ChainWait<4> {
    template <int CNT>
    static __device__ __forceinline__ void frags(bf16x8 (&w)[2], bf16x8 (&x)[4]) {
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(CNT));
    }
};
template <>
struct ChainWait<5> {
    template <int CNT>
    static __device__ __forceinline__ void frags(bf16x8 (&w)[2], bf16x8 (&x)[5]) {
        asm volatile("s_waitcnt lgkmcnt(%7)"
                     : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4])
                     : "n"(CNT));
    }
};
template <>
struct ChainWait<6> {
    template <int CNT>
    static __device__ __forceinline__ void frags(bf16x8 (&w)[2], bf16x8 (&x)[6]) {
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5])
                     : "n"(CNT));
    }
};
template <>
struct ChainWait<7> {
    template <int CNT>
    static __device__ __forceinline__ void frags(bf16x8 (&w)[2], bf16x8 (&x)[7]) {
        asm volatile("s_waitcnt lgkmcnt(%9)"
                     : "+v"(w[0]), "+v"(w[1]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]),
                       "+v"(x[6])
                     : "n"(CNT));
    }
};

#if defined(SL_CHAIN_PROBE_TIMES)  // s_memtime stamps per wave: [work-group][wave][40]: begin, after the input rows, per layer
                                   // {steps done, past barrier 1, epilogue done, past barrier 2}, end (tools/chain_stamps.py)
__device__ unsigned long long chain_probe_times[1024 * 8 * 40];
#define SL_CHAIN_STAMP(i)                                                        \
    do {                                                                         \
        unsigned long long t_;                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        if (lane == 0 && blockIdx.x < 1024) chain_probe_times[(blockIdx.x * 8 + wave) * 40 + (i)] = t_; \
    } while (0)
#else
#define SL_CHAIN_STAMP(i)
#endif

template <bool DGRAD>
__global__ __launch_bounds__(512, 1) void conv_chain_bf16_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* act = smem;
    char* wring = smem + ACT_BYTES;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int b = tile / a.t_tiles;
    const int t0 = (tile - b * a.t_tiles) * a.tm;
    const int n = a.n_layers;
    const int taps = a.taps;
    const int pad = a.pad;
    const int halo = taps - 1;          // rows a layer consumes beyond its output rows
    const int steps_per_layer = taps * 4;
    const int total_steps = n * steps_per_layer;
    const int w_rs = taps * CH;         // elements per output channel in the packed weights
    SL_CHAIN_STAMP(0);

    // ---- weight stream: global step g = (layer, tap, 64-channel chunk); DMA instruction j copies weight rows
    // [8j, 8j + 8): lane -> row 8j + lane / 8, physical 16-byte slot lane % 8 = logical slot ^ (row & 7).  The request
    // stream advances by increments (a uniform base pointer + four loop-invariant per-lane offsets, ring slot rotated): the
    // first version recomputed (layer, step) and the slot from g with integer divisions -- 73 scalar instructions per step
    // and wave (PMC: SQ_INSTS_SALU), more issue slots than the step's 22 MFMAs + 32 LDS instructions together.
    int voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        voff[q] = (8 * (wave * 4 + q) + (lane >> 3)) * w_rs + (((lane & 7) ^ (lane >> 3)) << 3);
    // Branch-free: past the last tile the last one is requested again (into a slot nobody reads any more), which also keeps
    // the vmcnt bookkeeping uniform -- one younger request of four instructions is outstanding at every wait.
    const __bf16* st_src = a.w[0];  // (uniform) source of the next tile to request
    int st_layer = 0, st_step = 0, st_slot = 0, st_left = total_steps;
    auto stage_next = [&]() {
        char* dst = wring + st_slot * WSLOT + wave * 4096;
#if !defined(SL_CHAIN_PROBE_NO_DMA)  // SL_CHAIN_PROBE_*: timing probes, wrong results by construction
#pragma unroll
        for (int q = 0; q < 4; ++q) chain_glds16(st_src + voff[q], dst + q * 1024);
#else
        (void)dst;
#endif
        st_slot = st_slot == NSLOT - 1 ? 0 : st_slot + 1;
        const bool more = st_left > 1;
        const bool wrap = more && st_step + 1 == steps_per_layer;
        st_left -= more ? 1 : 0;
        st_layer += wrap ? 1 : 0;
        st_step = wrap ? 0 : st_step + (more ? 1 : 0);
        const __bf16* layer_base = a.w[st_layer];
        st_src = wrap ? layer_base : st_src + (more ? 64 : 0);  // (tap, chunk) pieces of a layer are consecutive
    };
    stage_next();
    stage_next();

    // ---- input rows of the first layer: frames t0 - pad * n .. (64 + halo * n rows), through LDS-DMA like the weights
    // (piece = 8 rows of one 64-channel chunk = 1 KB; lane -> row 8 * group + lane / 8, physical slot lane % 8).  Frames
    // outside the utterance come from a zero row of the tensor itself: [-row0, 0) and [T', T' + 1) are halo / padding rows.
    {
        const int groups = (a.tm + halo * n + 7) >> 3;
        const __bf16* xb = a.x + (long)b * a.bs;
        for (int pc = wave; pc < 4 * groups; pc += 8) {
            const int c = pc & 3, rg = pc >> 2;
            const int r = rg * 8 + (lane >> 3);
            const int t = min(max(t0 - pad * n + r, -a.row0), a.t_out);
            chain_glds16(xb + (long)(a.row0 + t) * a.rs + c * 64 + (((lane & 7) ^ (r & 7)) << 3),
                         act + c * ACT_CHUNK + rg * 1024);
        }
        chain_wait_vmcnt<0>();
    }
    __syncthreads();
    SL_CHAIN_STAMP(1);

    f32x4 acc[MT_MAX][2];
#pragma unroll
    for (int m = 0; m < MT_MAX; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, lq = lane >> 4;
    // weight fragment rows of this lane: channel 32 * wave + 16 * nt + lrow; byte offsets inside a ring slot per k-half
    const int a_off0 = (32 * wave + lrow) * 128, a_key0 = (32 * wave + lrow) & 7;
    const unsigned w_frag[2] = {(unsigned)(a_off0 + ((lq ^ a_key0) << 4)), (unsigned)(a_off0 + (((4 + lq) ^ a_key0) << 4))};

    // epilogue: LDS byte offset of (row lrow, channels cb .. cb + 3): chunk cb / 64, 16-byte slot (cb % 64) / 8 ^ (row & 7)
    unsigned lds_w[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int cb = 32 * wave + 16 * nt + 4 * lq;
        lds_w[nt] = (cb >> 6) * ACT_CHUNK + lrow * 128 + ((((cb & 63) >> 3) ^ (lrow & 7)) << 4) + ((cb & 7) << 1);
    }

    // one layer's steps with a compile-time count of 16-row tiles: straight-line code per step (a wave-uniform
    // `if (m < mt)` around every tile made each tile its own basic block -- LDS read, wait, two MFMAs -- and the kernel
    // 3.4x slower than its MFMAs: 1.26 us per step)
    int cur_slot = 0;
    auto run_steps = [&](auto mt_c) {
        constexpr int MT = decltype(mt_c)::value;
        for (int s = 0; s < steps_per_layer; ++s) {
            // A wave requests exactly the weight rows it reads itself (rows [32 wave, 32 wave + 32) of every tile), so the
            // ring needs no barrier: the wave's own counted vmcnt orders its reads behind its requests, and the slot that is
            // re-requested here was last read two steps ago (retired by that step's lgkmcnt(0)).  The eight waves drift
            // apart inside a layer -- one wave's request / read phase runs under its SIMD partner's MFMAs.
            chain_wait_vmcnt<4>();  // tile g has landed (the younger request stays in flight)
#if defined(SL_CHAIN_PROBE_BARRIER)
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            stage_next();  // tile g + 2 into the slot tile g - 1 has left
            const int tap = s >> 2, chunk = s & 3;
            // fragment reads go through inline asm with hand-counted waits: through plain loads the compiler drains
            // vmcnt(0) before the first read of every step (it cannot prove that a read does not alias the LDS-DMA requests
            // in flight).  The 16-row tiles of a fragment family differ by a constant (16 rows = 2048 bytes: the swizzle key
            // (row & 7) does not change), which rides in the instruction's offset field.  LDS returns in order:
            // lgkmcnt(2 + MT) = the first k-half's fragments are there.
            const unsigned wslot = lds0 + ACT_BYTES + cur_slot * WSLOT;
            cur_slot = cur_slot == NSLOT - 1 ? 0 : cur_slot + 1;
            const int rr = lrow + tap;
            const unsigned arow = lds0 + chunk * ACT_CHUNK + rr * 128;
            const int akey = rr & 7;
            bf16x8 wa[2][2], xb[2][MT];
#if defined(SL_CHAIN_PROBE_NO_READS)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                wa[kh][0] = wa[kh][1] = (bf16x8){};
#pragma unroll
                for (int m = 0; m < MT; ++m) xb[kh][m] = (bf16x8){};
            }
            (void)wslot; (void)arow; (void)akey;
#else
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const unsigned waddr = wslot + w_frag[kh];
                chain_ds_read128<0>(wa[kh][0], waddr);
                chain_ds_read128<16 * 128>(wa[kh][1], waddr);
                ChainReadRun<0, MT>::go(xb[kh], arow + (((kh * 4 + lq) ^ akey) << 4));
            }
#endif
            ChainWait<MT>::template frags<2 + MT>(wa[0], xb[0]);
#if !defined(SL_CHAIN_PROBE_NO_MFMA)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][0], xb[0][m], acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][1], xb[0][m], acc[m][1], 0, 0, 0);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);  // (keeps the second wait behind the first k-half's MFMAs)
            ChainWait<MT>::template frags<0>(wa[1], xb[1]);
#if !defined(SL_CHAIN_PROBE_NO_MFMA)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][0], xb[1][m], acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][1], xb[1][m], acc[m][1], 0, 0, 0);
            }
#endif
        }
    };
    for (int l = 0; l < n; ++l) {
        const int rows_out = a.tm + halo * (n - 1 - l);
        const int mt = (rows_out + 15) >> 4;
        // this wave's bias values, requested before the layer's steps (a load in the epilogue is a full L2 round trip
        // with every MFMA pipe idle)
        f32x4 bv[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        if (!DGRAD) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bv[nt] = *(const f32x4*)(a.bias[l] + 32 * wave + 16 * nt + 4 * lq);
        }
#if defined(SL_CHAIN_PROBE_NO_EPILOGUE)
        if (l > 0 && a.batch > 0) {
            run_steps(std::integral_constant<int, 5>{});
            continue;
        }
#endif
        switch (mt) {
            case 3: run_steps(std::integral_constant<int, 3>{}); break;
            case 4: run_steps(std::integral_constant<int, 4>{}); break;
            case 5: run_steps(std::integral_constant<int, 5>{}); break;
            case 6: run_steps(std::integral_constant<int, 6>{}); break;
            default: run_steps(std::integral_constant<int, 7>{}); break;
        }
        // ---- layer epilogue.  D^T tile: lane holds channels cb + 0..3 (cb = 32 * wave + 16 * nt + 4 * (lane >> 4)) of output
        // row j = 16 * m + (lane & 15), i.e. frame t = t_first + j.  The epilogue is VALU work with every MFMA pipe idle (two
        // waves per SIMD share the VALU): its first version -- per-tile 64-bit addresses, software bf16 rounding, per-element
        // liveness selects -- was 45 instructions per tile = 2.7 us per layer.  Now: addresses = per-lane bases + immediate
        // offsets, v_cvt_pk_bf16_f32, and the zeroing of out-of-utterance rows only in work-groups that have such rows.
        SL_CHAIN_STAMP(2 + 4 * l);
        // ---- layer epilogue.  D^T tile: lane holds channels cb + 0..3 (cb = 32 * wave + 16 * nt + 4 * (lane >> 4)) of output
        // row j = 16 * m + (lane & 15), i.e. frame t = t_first + j.  Everything here runs with the MFMA pipes idle, and the
        // first version was 20 % of the kernel: (a) 45 VALU instructions per tile (64-bit addresses, software bf16 rounding,
        // per-element selects; two waves share a SIMD's VALU) and (b) 8-byte-per-lane HBM stores and ReLU-mask loads in the
        // tile layout: 16 different 128-byte lines per instruction, 1800 line transactions per layer and work-group, and
        // their issue stalls (s_memtime stamps: 5000 cycles per layer forward, 11 000 backward).  Now the tiles only go to
        // LDS (per-lane base addresses + immediate offsets, v_cvt_pk_bf16_f32), and the HBM side is row-contiguous from the
        // LDS copy: 16 bytes per lane, every line written / read once.
        const int t_first = t0 - pad * (n - 1 - l);
        __bf16* yb = a.y[l] + (long)b * a.bs;
        const int jc0 = t0 - t_first, jc_n = min(a.tm, a.t_out - t0);                 // core rows: j - jc0 in [0, jc_n)
        const int jl0 = max(0, -t_first), jl_n = min(16 * mt, a.t_out - t_first) - jl0;  // rows inside the utterance
        const bool edge = t_first < 0 || t_first + 16 * mt > a.t_out;
        // piece i of the row-contiguous view: row 16 i + 2 wave + lane / 32, 16-byte slot lane % 32 of the row's 512 bytes
        const int prow = 2 * wave + (lane >> 5), pslot = lane & 31;
        const unsigned plds = lds0 + (pslot >> 3) * ACT_CHUNK + prow * 128 + (((pslot & 7) ^ (prow & 7)) << 4);  // + 2048 i
        u32x4 mk[MT_MAX];
        if (DGRAD) {  // ReLU masks (the stored forward activations) of all rows, requested before the barrier; rows outside
                      // the utterance read a zero halo / padding row, which also zeroes their gradient
            const __bf16* mb = a.mask[l] + (long)b * a.bs;  // (uniform base + 32-bit lane offset: one VGPR per address)
#pragma unroll
            for (int i = 0; i < MT_MAX; ++i) {  // (all seven pieces whatever mt is: straight-line code, rows stay in bounds)
                const int t = min(max(t_first + 16 * i + prow, -a.row0), a.t_out);
                mk[i] = *(const u32x4*)(mb + (unsigned)((a.row0 + t) * CH + pslot * 8));
            }
        }
        __builtin_amdgcn_s_barrier();  // every wave has finished reading this layer's input rows
        asm volatile("" ::: "memory");
        SL_CHAIN_STAMP(3 + 4 * l);
        auto epilogue = [&](auto edge_c) {
            constexpr bool EDGE = decltype(edge_c)::value;
#pragma unroll
            for (int m = 0; m < MT_MAX; ++m) {
                if (m < mt) {
                    const bool live = !EDGE || (unsigned)(16 * m + lrow - jl0) < (unsigned)jl_n;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        f32x4 v = acc[m][nt];
                        acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (!DGRAD) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] + bv[nt][i], 0.f);
                        }
                        u32x2 pk;
                        pk[0] = pack_bf16x2_hw(v[0], v[1]);
                        pk[1] = pack_bf16x2_hw(v[2], v[3]);
                        if (EDGE && !DGRAD) {  // SAME padding: relu(bias) is not zero
                            pk[0] = live ? pk[0] : 0u;
                            pk[1] = live ? pk[1] : 0u;
                        }
                        // next layer's input row j, channels cb .. cb + 3 (16 rows on = 2048 bytes, same swizzle key)
                        *(u32x2*)(act + lds_w[nt] + m * 2048) = pk;
                    }
                }
            }
        };
        if (edge && !DGRAD)
            epilogue(std::true_type{});
        else
            epilogue(std::false_type{});
        SL_CHAIN_STAMP(4 + 4 * l);
        // the rewritten rows are visible (not __syncthreads(): its fence would drain the weight requests in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (DGRAD) {
            // masking pass over ALL rows (the next layer consumes the halo rows too), HBM store of the core rows
            u32x4 gq[MT_MAX];
#pragma unroll
            for (int i = 0; i < MT_MAX; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(gq[i]) : "v"(plds), "n"(i * 2048));
            // (the registers are operands of the wait: no use of them can be scheduled above it)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(gq[0]), "+v"(gq[1]), "+v"(gq[2]), "+v"(gq[3]), "+v"(gq[4]), "+v"(gq[5]), "+v"(gq[6])
                         :
                         : "memory");
            static_assert(MT_MAX == 7, "the wait above names seven registers");
#pragma unroll
            for (int i = 0; i < MT_MAX; ++i) {
                {
                    // per bf16: forward activation > 0 ? g : 0  ==  g * min(max(bits as int16, 0), 1)
                    const sl_s16x8_t k = __builtin_bit_cast(sl_s16x8_t, mk[i]);
                    const sl_s16x8_t keep = __builtin_elementwise_min(__builtin_elementwise_max(k, (sl_s16x8_t)(0)), (sl_s16x8_t)(1));
                    const sl_u16x8_t prod = __builtin_bit_cast(sl_u16x8_t, gq[i]) * __builtin_bit_cast(sl_u16x8_t, keep);
                    const u32x4 o = __builtin_bit_cast(u32x4, prod);
                    const int j = 16 * i + prow;
                    *(u32x4*)(act + (plds - lds0) + i * 2048) = o;
                    if ((unsigned)(j - jc0) < (unsigned)jc_n)
                        *(u32x4*)(yb + (unsigned)((a.row0 + t_first + j) * CH + pslot * 8)) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            // HBM store of the core rows from the LDS copy (4 pieces per wave)
            const int crow = jc0 + prow;
            const unsigned clds = lds0 + (pslot >> 3) * ACT_CHUNK + crow * 128 + (((pslot & 7) ^ (crow & 7)) << 4);
            u32x4 cq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(cq[i]) : "v"(clds), "n"(i * 2048));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3]) : : "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 o = cq[i];
                const int j = 16 * i + prow;  // (relative to the first core row)
                if (j < jc_n) *(u32x4*)(yb + (unsigned)((a.row0 + t0 + j) * CH + pslot * 8)) = o;
            }
        }
        SL_CHAIN_STAMP(5 + 4 * l);
    }
    SL_CHAIN_STAMP(38);
    chain_wait_vmcnt<0>();  // (the surplus requests of the branch-free stream must not outlive the work-group's LDS)
}

}  // namespace

// the run must fit the kernel's fixed shape: 256 padded channels in and out, odd taps <= 9, 2..8 layers, every tensor in
// the same halo'd geometry
bool conv_chain_bf16_supported(const sl_conv_geom* g, int n_layers) {
    return g->cin == CH && g->cout == CH && g->taps >= 3 && g->taps <= 9 && (g->taps & 1) == 1 && n_layers >= 2 &&
           n_layers <= MAX_LAYERS && 16 * MT_MAX >= TM + (g->taps - 1) * (n_layers - 1) &&
           16 * MT_MAX + g->taps - 1 <= ACT_ROWS && g->x_row_stride == CH && g->y_row_stride == CH &&
           g->x_batch_stride == g->y_batch_stride && g->x_row0 + g->taps / 2 == g->y_row0;
}

#if defined(SL_CHAIN_PROBE_TIMES)
extern "C" int sl_chain_probe_read(void* host_dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(chain_probe_times), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

static int g_chain_tile_rows = 0;  // sl_conv1d_chain_select: 0 = automatic, 48 / 64 = forced
int conv_chain_select(int rows) {
    g_chain_tile_rows = rows;
    return SL_OK;
}

// Output frames per work-group.  A work-group fills a CU (160 KB of LDS), so the launch runs in rounds of 256; 64-frame
// tiles recompute less halo (39 sixteen-row MFMA tiles per 64 frames at seven 7-tap layers) than 48-frame tiles (32 per
// 48 frames = +9 %), but 8 utterances x 2500 frames (config 5) are 320 work-groups = two rounds, the second a quarter full:
// 424 shorter work-groups make 1.66 rounds.  Minimise rounds x MFMA tiles per work-group.
static int chain_tile_rows(int batch, int t_out, int taps, int n_layers) {
    if (g_chain_tile_rows == TM || g_chain_tile_rows == TM_SMALL) return g_chain_tile_rows;
    double best_cost = 1e30;
    int best = TM;
    for (int tm : {TM, TM_SMALL}) {
        long tiles = 0;
        for (int l = 0; l < n_layers; ++l) tiles += (tm + (taps - 1) * (n_layers - 1 - l) + 15) / 16;
        const long wgs = (long)batch * ((t_out + tm - 1) / tm);
        const double cost = (double)((wgs + sl_cus() - 1) / sl_cus()) * (double)tiles;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = tm;
        }
    }
    return best;
}

int conv_chain_bf16(const void* x, void* const* ys, const void* const* ws, const float* const* biases,
                    const void* const* masks, const sl_conv_geom* g, int n_layers, int epilogue, hipStream_t s) {
    ChainArgs a;
    a.x = (const __bf16*)x;
    for (int i = 0; i < n_layers; ++i) {
        a.y[i] = (__bf16*)ys[i];
        a.w[i] = (const __bf16*)ws[i];
        a.bias[i] = biases ? biases[i] : nullptr;
        a.mask[i] = masks ? (const __bf16*)masks[i] : nullptr;
    }
    a.n_layers = n_layers;
    a.taps = g->taps;
    a.pad = g->taps / 2;
    a.batch = g->batch;
    a.t_out = g->t_out;
    a.tm = chain_tile_rows(g->batch, g->t_out, g->taps, n_layers);
    a.t_tiles = (g->t_out + a.tm - 1) / a.tm;
    a.row0 = g->y_row0;
    a.rs = CH;
    a.bs = g->y_batch_stride;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_chain_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  CHAIN_LDS);
        (void)hipFuncSetAttribute((const void*)conv_chain_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  CHAIN_LDS);
        attr_set = true;
    }
    const dim3 grid(a.batch * a.t_tiles);
    if (epilogue == SL_EPI_RELU_MASK)
        SL_LAUNCH_MAIN(conv_chain_bf16_kernel<true>, grid, dim3(512), CHAIN_LDS, s, a);
    else
        SL_LAUNCH_MAIN(conv_chain_bf16_kernel<false>, grid, dim3(512), CHAIN_LDS, s, a);
    return sl_check_launch("sl_conv1d_chain");
}
