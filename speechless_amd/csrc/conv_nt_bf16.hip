// conv_nt_bf16.hip -- im2col-free row-shifted NT GEMM for the Wav2Letter conv stack on gfx950 (bf16 in, fp32 accumulate).
//
// Computes (see include/speechless_hip.h, sl_conv1d_nt):
//     y[b][y_row0+t][co] = epi( sum_tap sum_c x[b][x_row0+t+tap][c] * w[co][tap][c] )
// which is the forward of a Keras Conv1D (reference speechless/net.py:304-305) and, with flipped/transposed packed
// weights, its input gradient (autodiff from net.py:389,550).
//
// Mapping to CDNA4:
//   * work-group tile = BM time rows x (64*WN) output channels, WM*WN waves.  Two wave-level MFMA shapes:
//       M32 = false: v_mfma_f32_16x16x32_bf16, wave patch (16*IT) x 64 (IT = 4: 64x64; IT = 2: 32x64, twice the waves)
//       M32 = true : v_mfma_f32_32x32x16_bf16, wave patch 64 x 64 = 2x2 tiles (higher MFMA ceiling: 2.38 vs 2.07 PF/s
//                    micro-benchmark, half the MFMA instructions)
//     Tile shapes 128x128 (4 or 8 waves) ... 256x256 (16 waves, 1 work-group/CU): bigger tiles halve the L2->LDS
//     operand traffic per flop, which is what bounds the 128x128 tile (~16 TB/s at 1 PFLOP/s against a ~34 TB/s L2).
//   * the contraction runs over (tap, 64-channel chunk): one step = one BMx64 activation tile (a plain row-shifted
//     2-D tile of the halo'd channels-last tensor -> no im2col, no bounds checks) and one BNx64 weight tile.
//   * tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip) through a STAGES-deep ring; the wait
//     is a COUNTED s_waitcnt vmcnt(N) (never 0 in steady state) and the barrier a raw s_barrier, so STAGES-1 tiles
//     stay in flight across barriers.
//   * LDS rows are 128 B; the 16-B slot index is XOR-swizzled on the DMA *source* address (the LDS image must stay
//     lane-linear) with a key chosen per operand / MFMA shape so that every ds_read_b128 lane group (the hardware
//     serves {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... together) hits 16 distinct bank slots.
//   * D^T orientation: MFMA "A" = weights (rows = co), "B" = activations (cols = t), with the weight rows fed to the
//     MFMA in a permuted order so that each lane ends up with 16 CONSECUTIVE output channels of one time row: the
//     epilogue (bias + ReLU / ReLU-mask, bf16 convert) stores 32 contiguous bytes per lane.
//   * split-K (over taps x chunks) when the tile count cannot fill the chip (dgrad of big_conv_1: K = 65536, 64 tiles):
//     fp32 partial tiles go to a workspace and nt_splitk_epilogue_kernel reduces them in a fixed order and applies the
//     epilogue (deterministic).
//   * blockIdx -> tile mapping is XCD-aware (xcd_remap + a 2-D raster): an XCD's work-groups share weight panels and
//     activation rows in its private L2; grids are padded to a multiple of 8 so that this holds for any tile count.
//   * three loop flavours per tile shape (stages field of the cfg word): plain ring (compiler-scheduled LDS reads);
//     register-pipelined (+8: fragments read through inline asm with hand-counted lgkmcnt, barrier in the middle of a
//     step); interleaved (cfg bit 30: IlvPhase, one LDS read / DMA request behind every one or two MFMAs).
//   * conv_nt_slab_bf16_kernel (cfg bit 29) is the chunk-major variant that keeps the activation rows of all taps in LDS.
//   * conv_nt_ks2_bf16_kernel (cfg it = 5) is the 128x128 tile with eight waves, the two waves of a SIMD splitting the k-halves.
//   The library's measured table (auto_cfg) picks shape, flavour and kernel per launch; tools/tune_kernels.py sweeps them.
#include "common.h"

#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int BK = 64;

struct NtArgs {
    const __bf16* x;
    const __bf16* w;
    const float* bias;
    const __bf16* mask;
    void* y;
    float* partial;  // split-K workspace [ksplit][batch][t_tiles*BM][cout]
    int batch, t_out, t_tiles, n_tiles;
    int x_row0, x_rs;
    long x_bs;
    int y_row0, y_rs;
    long y_bs;
    int cout;
    float acc_scale;  // SL_ELEM_F16 build only: the accumulator is multiplied by this before bias / mask (sl_conv_geom.acc_scale)
    int out_planes;  // fp32-output kernels only: 0 = fp32 store; bf16x3 planes [hi | lo | hi] instead: 1 = after ReLU, 2 = as is, 3 = after the ReLU mask
    int w_rs;    // taps * cin
    int taps, cin;
    int nsteps;  // taps * cin / 64
    int ksplit, steps_per_split;
    int gm;  // tile raster: blocks of (all n_tiles) x gm m-tiles are numbered consecutively
};

__device__ __forceinline__ void glds16(const __bf16* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const SL_GLOBAL void*)gsrc, (SL_LDS void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS fragment reads the compiler does not track (it answers every LDS dependency in these kernels with lgkmcnt(0),
// because the LDS-DMA loads leave a "flat access pending" mark): the hand-counted wait below releases the registers.
template <int OFF>
__device__ __forceinline__ void ds_read128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int I, int N, int STRIDE>
struct DsReadRun {
    static __device__ __forceinline__ void go(bf16x8 (&f)[N], unsigned addr) {
        ds_read128<I * STRIDE>(f[I], addr);
        DsReadRun<I + 1, N, STRIDE>::go(f, addr);
    }
};
template <int N, int STRIDE>
struct DsReadRun<N, N, STRIDE> {
    static __device__ __forceinline__ void go(bf16x8 (&)[N], unsigned) {}
};
// s_waitcnt lgkmcnt(CNT) that the MFMAs consuming these fragments cannot be hoisted above
template <int CNT>
__device__ __forceinline__ void wait_frags(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void wait_frags(bf16x8 (&a)[4], bf16x8 (&b)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                   "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                 : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void wait_frags(bf16x8 (&a)[4], bf16x8 (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1])
                 : "n"(CNT));
}

// fp32 -> bf16 in the main epilogue: v_cvt_pk_bf16_f32 (round to nearest even like the software form, which costs five VALU
// instructions per value: 640 of a wave's epilogue at the 256 x 256 tile)
#if defined(SL_NT_SOFT_PACK)
#define SL_NT_PACK pack_bf16x2
#else
#define SL_NT_PACK pack_bf16x2_hw
#endif
enum { MODE_PARTIAL = 100 };  // besides the SL_EPI_* values: raw fp32 accumulators to the split-K workspace

__device__ __forceinline__ float bf16_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int u) { return __uint_as_float(u & 0xFFFF0000u); }
// ELU: y = z > 0 ? z : exp(z) - 1;   dy/dz expressed through the stored output: y > 0 ? 1 : y + 1
__device__ __forceinline__ float elu_f(float z) { return z > 0.f ? z : expm1f(z); }
__device__ __forceinline__ float elu_grad_from_y(float y) { return y > 0.f ? 1.f : y + 1.f; }

// ---- hi + lo planes of the fp32-output kernels (out_planes): bf16 pairs ("bf16x3"), or -- the translation unit built with
// -DSL_ELEM_F16 ("f16x3") -- fp16 pairs: hi = fp16(v), lo = fp16(v - hi), 22 significand bits where |v| >= 2^-3 and an
// absolute 2^-25 below (the lo plane runs into fp16's denormals, which v_mfma_f32_16x16x32_f16 keeps: tools/f16_denorm_probe.hip)
#if defined(SL_ELEM_F16)
__device__ __forceinline__ unsigned int plane_pack_hi(float a0, float a1) { return pack_f16x2(a0, a1); }
__device__ __forceinline__ unsigned int plane_pack_lo(float a0, float a1, unsigned int h) {
    return pack_f16x2(a0 - f16_lo_to_f32(h), a1 - f16_hi_to_f32(h));
}
#define SL_NT_ACC_SCALE(v, a) ((v) * (a).acc_scale)
#define SL_NT_ACC_SCALE_BIAS(v, b, a) fmaf((v), (a).acc_scale, (b))
#else
__device__ __forceinline__ unsigned int plane_pack_hi(float a0, float a1) { return pack_bf16x2(a0, a1); }
__device__ __forceinline__ unsigned int plane_pack_lo(float a0, float a1, unsigned int h) {
    return pack_bf16x2(a0 - __uint_as_float(h << 16), a1 - __uint_as_float(h & 0xFFFF0000u));
}
#define SL_NT_ACC_SCALE(v, a) (v)
#define SL_NT_ACC_SCALE_BIAS(v, b, a) ((v) + (b))
#endif

// epilogue of one run of 16 consecutive output channels of one time row
template <int MODE, bool OUT_F32>
__device__ __forceinline__ void store_run16(const NtArgs& a, float (&v)[16], const float (&bias_v)[16], long yidx) {
    if (MODE == SL_EPI_BIAS || MODE == SL_EPI_BIAS_RELU || MODE == SL_EPI_BIAS_ELU) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = SL_NT_ACC_SCALE_BIAS(v[i], bias_v[i], a);
    } else if (MODE != MODE_PARTIAL) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = SL_NT_ACC_SCALE(v[i], a);
    }
    if (MODE == SL_EPI_BIAS_RELU) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (MODE == SL_EPI_BIAS_ELU) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = elu_f(v[i]);
    }
    if (MODE == SL_EPI_ELU_MASK) {
        const u32x4 m0 = *(const u32x4*)(a.mask + yidx);
        const u32x4 m1 = *(const u32x4*)(a.mask + yidx + 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i * 2] *= elu_grad_from_y(bf16_lo(m0[i]));
            v[i * 2 + 1] *= elu_grad_from_y(bf16_hi(m0[i]));
            v[8 + i * 2] *= elu_grad_from_y(bf16_lo(m1[i]));
            v[8 + i * 2 + 1] *= elu_grad_from_y(bf16_hi(m1[i]));
        }
    }
    if (MODE == SL_EPI_RELU_MASK) {
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
        if (a.out_planes) {
            // bf16x3 (split3.hip): what sl_split3 would do to the fp32 tile in a second pass over HBM -- activation, then
            // v = hi + lo in two bf16 planes, rows [hi | lo | hi] of 3 x cout channels.  A wave-uniform run-time branch
            // inside the two fp32-output instantiations (BIAS, NONE): the bf16 kernels of the benchmarked path do not see it.
            if (a.out_planes == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            if (a.out_planes == 3) {  // mask = the stored activation's hi plane, same geometry as y
                const u32x4 m0 = *(const u32x4*)(a.mask + yidx);
                const u32x4 m1 = *(const u32x4*)(a.mask + yidx + 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned int lo0 = m0[i] & 0xFFFFu, hi0 = m0[i] >> 16;
                    const unsigned int lo1 = m1[i] & 0xFFFFu, hi1 = m1[i] >> 16;
                    if (!(lo0 != 0 && lo0 < 0x8000u)) v[i * 2] = 0.f;
                    if (!(hi0 != 0 && hi0 < 0x8000u)) v[i * 2 + 1] = 0.f;
                    if (!(lo1 != 0 && lo1 < 0x8000u)) v[8 + i * 2] = 0.f;
                    if (!(hi1 != 0 && hi1 < 0x8000u)) v[8 + i * 2 + 1] = 0.f;
                }
            }
            u32x4 h0, h1, l0, l1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a0 = v[i * 2], a1 = v[i * 2 + 1], b0 = v[8 + i * 2], b1 = v[8 + i * 2 + 1];
                h0[i] = plane_pack_hi(a0, a1);
                h1[i] = plane_pack_hi(b0, b1);
                l0[i] = plane_pack_lo(a0, a1, h0[i]);
                l1[i] = plane_pack_lo(b0, b1, h1[i]);
            }
            __bf16* yo = (__bf16*)a.y + yidx;
            *(u32x4*)(yo) = h0;
            *(u32x4*)(yo + 8) = h1;
            *(u32x4*)(yo + a.cout) = l0;
            *(u32x4*)(yo + a.cout + 8) = l1;
            *(u32x4*)(yo + 2 * a.cout) = h0;
            *(u32x4*)(yo + 2 * a.cout + 8) = h1;
            return;
        }
        float* yo = (float*)a.y + yidx;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(yo + i * 4) = (f32x4){v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
    } else {
        __bf16* yo = (__bf16*)a.y + yidx;
        u32x4 p0, p1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p0[i] = SL_NT_PACK(v[i * 2], v[i * 2 + 1]);
            p1[i] = SL_NT_PACK(v[8 + i * 2], v[8 + i * 2 + 1]);
        }
        *(u32x4*)(yo) = p0;  // (nontemporal stores measured: step 2.35 vs 2.30 ms)
        *(u32x4*)(yo + 8) = p1;
    }
}

__device__ __forceinline__ void load_bias16(const float* bias, int co, float (&bias_v)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 bv = *(const f32x4*)(bias + co + i * 4);
        bias_v[i * 4 + 0] = bv[0];
        bias_v[i * 4 + 1] = bv[1];
        bias_v[i * 4 + 2] = bv[2];
        bias_v[i * 4 + 3] = bv[3];
    }
}

// Hand-interleaved half step: the 4*IT MFMAs of one k-half in groups of G, with ONE fragment read of the other k-half
// issued after each of the first 4+IT groups and the hook (e.g. one LDS-DMA request) after every group.  Every group is
// fenced (sched_barrier), so the LDS-read / DMA issue slots sit in the shadow of the matrix pipe instead of in a clump
// between two MFMA blocks; PMC on the un-interleaved loops showed ~110 non-MFMA instructions per step and wave issued
// while the pipe drained (pipe busy 65 % on the 256x256 tile).
template <int IT, int G, int Q, int NQ, int RPG = 1>
struct IlvPhase {
    // RPG = fragment reads per group: 1 spreads the 4+IT reads over the first 4+IT groups (two waves per SIMD cover
    // each other's read latency); 2 packs them into the first (4+IT)/2 groups, so that the last read has half a phase of
    // MFMAs behind it before the next phase waits for it (one wave per SIMD)
    template <int R>
    static __device__ __forceinline__ void read_one(bf16x8 (&na)[4], bf16x8 (&nb)[IT], unsigned a_addr, unsigned b_addr) {
        if constexpr (R < 4)
            ds_read128<R * 512>(na[R], a_addr);
        else if constexpr (R < 4 + IT)
            ds_read128<(R - 4) * 2048>(nb[R - 4], b_addr);
    }
    template <typename Hook>
    static __device__ __forceinline__ void run(f32x4 (&acc)[4 * IT], const bf16x8 (&ca)[4], const bf16x8 (&cb)[IT],
                                               bf16x8 (&na)[4], bf16x8 (&nb)[IT], unsigned a_addr, unsigned b_addr,
                                               const Hook& hook) {
        static_assert(NQ * G == 4 * IT && NQ * RPG >= 4 + IT, "groups must cover the MFMAs and offer a slot per fragment read");
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int m = G * Q + j;
            acc[m] = SL_MFMA16(ca[m / IT], cb[m % IT], acc[m]);
        }
        // (a fence here, forcing the group's MFMAs to issue before its read / request, measured slower: 2.291 vs 2.283 ms)
        read_one<RPG * Q>(na, nb, a_addr, b_addr);
        if constexpr (RPG == 2) read_one<RPG * Q + 1>(na, nb, a_addr, b_addr);
        hook(std::integral_constant<int, Q>{});
        __builtin_amdgcn_sched_barrier(0);
        IlvPhase<IT, G, Q + 1, NQ, RPG>::template run<Hook>(acc, ca, cb, na, nb, a_addr, b_addr, hook);
    }
};
template <int IT, int G, int NQ, int RPG>
struct IlvPhase<IT, G, NQ, NQ, RPG> {
    template <typename Hook>
    static __device__ __forceinline__ void run(f32x4 (&)[4 * IT], const bf16x8 (&)[4], const bf16x8 (&)[IT], bf16x8 (&)[4],
                                               bf16x8 (&)[IT], unsigned, unsigned, const Hook&) {}
};

// STAGES_P: low 3 bits = ring slots, bit 3 = register-pipelined main loop (fragments of the next tile's first half are
// read from LDS while the MFMAs of the current tile's second half run, see the loop)
template <bool M32, int IT, int WM, int WN, int STAGES_P, int MODE, bool OUT_F32>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16 ? 4 : 2)) void conv_nt_bf16_kernel(NtArgs a) {
    constexpr int STAGES = STAGES_P & 7;
    constexpr bool PIPE = (STAGES_P & 8) != 0;
    constexpr bool ILV = (STAGES_P & 16) != 0;  // hand-interleaved variant of the pipelined loop
    static_assert(STAGES >= 2, "ring too shallow");
    static_assert(!ILV || (PIPE && !M32), "the interleaved schedule is a variant of the register-pipelined 16x16 loop");
    constexpr int NW = WM * WN;
    constexpr int WROWS = M32 ? 64 : 16 * IT;  // time rows per wave
    constexpr int BM = WROWS * WM;
    constexpr int BN = 64 * WN;
    constexpr int X_BYTES = BM * 128;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int XPW = (BM / 8) / NW;  // DMA instructions per wave per stage for the activation tile
    constexpr int WPW = (BN / 8) / NW;  // ... and for the weight tile
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must split evenly over the waves");
    constexpr int NI = XPW + WPW;

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
    const int wm = wave / WN;  // which block of time rows
    const int wn = wave % WN;  // which 64-channel block (co)

    const int m_tiles = a.batch * a.t_tiles;
    const int tiles = m_tiles * a.n_tiles;
    const int id = xcd_remap(blockIdx.x, tiles * a.ksplit);
    if (id >= tiles * a.ksplit) return;  // grid padding (xcd_grid)
    const int split = id / tiles;
    const int tile = id - split * tiles;
    // raster: consecutive ids (= one XCD's concurrently resident work-groups, see xcd_remap) cover a 2-D block of
    // n_tiles x gm tiles, so that the activation rows AND the weight panels an XCD streams both stay within its 4 MiB
    // L2.  With gm = m_tiles (1-D, weight-panel-major) big_conv_1's 32 resident activation tiles per XCD (4.7 MB) are
    // evicted between taps: 1.25 GB per launch re-fetched from the Infinity Cache (profiles/r01c_pmc_*.json).
    const int span = a.n_tiles * a.gm;
    const int blk = tile / span;
    const int rem = tile - blk * span;
    int cnt = m_tiles - blk * a.gm;
    if (cnt > a.gm) cnt = a.gm;
    const int n_tile = rem / cnt;
    const int m_tile = blk * a.gm + (rem - n_tile * cnt);
    const int b = m_tile / a.t_tiles;
    const int t0 = (m_tile - b * a.t_tiles) * BM;
    const int co0 = n_tile * BN;
    const int s_begin = split * a.steps_per_split;
    int n = a.nsteps - s_begin;
    if (n > a.steps_per_split) n = a.steps_per_split;

    // ---- staging: DMA instruction j copies rows [8j, 8j+8) of a tile (lane -> row 8j + lane/8, 16-B slot lane%8).
    // swizzle keys (function of the tile row r):
    //   16x16 shape, activations: r & 7            weights: ((r>>1)&1) | (((r>>4)&3)<<1)   (rows reach the MFMA permuted)
    //   32x32 shape, both:        ((r>>1)&3) | (((r>>4)&1)<<2)
    const __bf16* xbase = a.x + (long)b * a.x_bs + (long)(a.x_row0 + t0) * a.x_rs;
    const __bf16* wbase = a.w + (long)co0 * a.w_rs;
    int xoff[XPW], woff[WPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
        const int j = wave * XPW + q;
        const int row = j * 8 + (lane >> 3);
        const int key = M32 ? (((lane >> 4) & 3) | (((j >> 1) & 1) << 2)) : (lane >> 3);
        xoff[q] = row * a.x_rs + (((lane & 7) ^ key) << 3);
    }
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
        const int j = wave * WPW + q;
        const int row = j * 8 + (lane >> 3);
        const int key = M32 ? (((lane >> 4) & 3) | (((j >> 1) & 1) << 2)) : (((lane >> 4) & 1) | (((j >> 1) & 3) << 1));
        woff[q] = row * a.w_rs + (((lane & 7) ^ key) << 3);
    }

    // contraction order: tap OUTER, 64-channel chunk INNER (the weight rows are then walked contiguously).  The
    // alternative (chunk outer, tap inner: consecutive steps re-read the same activation rows shifted by one, L2 reuse
    // distance of one step) was measured SLOWER on MI355X (big_conv_1 fwd 0.424 vs 0.378 ms, dgrad 0.434 vs 0.404 ms):
    // the activation re-reads that miss the 4 MiB L2 (FETCH_SIZE 0.6-1.2 GB per launch) are served by the 256 MiB
    // Infinity Cache and are not the bottleneck, while the strided weight walk costs more.
    auto stage = [&](int step, int buf) {
        const int chunks = a.cin / BK;
        const int tap = step / chunks;
        const int cc = step - tap * chunks;
        const __bf16* xs = xbase + (long)tap * a.x_rs + cc * BK;
        const __bf16* ws = wbase + (long)step * BK;
        char* xl = smem + buf * STAGE_BYTES + (wave * XPW) * 1024;
        char* wl = smem + buf * STAGE_BYTES + X_BYTES + (wave * WPW) * 1024;
#pragma unroll
        for (int q = 0; q < XPW; ++q) glds16(xs + xoff[q], xl + q * 1024);
#pragma unroll
        for (int q = 0; q < WPW; ++q) glds16(ws + woff[q], wl + q * 1024);
    };

    // ---- fragment read addresses (byte offsets inside a ring slot, for k-chunk 0)
    int boff, aoff;
    if (M32) {
        // B operand: lane -> time row lane&31 (+32*it), 16-B chunk (lane>>5) + 2*kstep
        // A operand: lane -> MFMA row i = lane&31 fed from weight row perm(i) = ((i>>2)&1)*16 + (i&3) + 4*(i>>3), so that
        //            D rows (r&3) + 8*(r>>2) + 4*(lane>>5), r = 0..15, are the 16 consecutive channels (lane>>5)*16 + r
        const int i = lane & 31, h = lane >> 5;
        const int brow = wm * 64 + i;
        const int arow = wn * 64 + ((i >> 2) & 1) * 16 + (i & 3) + 4 * (i >> 3);
        const int bkey = ((i >> 1) & 3) | (((i >> 4) & 1) << 2);
        const int akey = ((i >> 1) & 1) | (((i >> 3) & 1) << 1) | (((i >> 2) & 1) << 2);
        boff = brow * 128 + ((h ^ bkey) << 4);
        aoff = X_BYTES + arow * 128 + ((h ^ akey) << 4);
    } else {
        const int g = lane >> 4;
        const int brow = wm * WROWS + (lane & 15);                        // + it*16
        const int arow = wn * 64 + ((lane & 15) >> 2) * 16 + (lane & 3);  // + jn*4
        const int bkey = lane & 7;
        const int akey = ((lane >> 1) & 1) | (((lane >> 2) & 3) << 1);
        boff = brow * 128 + ((g ^ bkey) << 4);
        aoff = X_BYTES + arow * 128 + ((g ^ akey) << 4);
    }

    constexpr int NACC16 = M32 ? 1 : 4 * IT;
    constexpr int NACC32 = M32 ? 4 : 1;
    f32x4 acc[NACC16];    // [jn][it] 16x16 tiles
    f32x16 acc32[NACC32]; // [jn][it] 32x32 tiles
#pragma unroll
    for (int i = 0; i < NACC16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NACC32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;

    // one 64-channel step = two halves of 32 channels; a half's fragments: NA weight vectors + NB activation vectors
    constexpr int NA = M32 ? 4 : 4;
    constexpr int NB = M32 ? 4 : IT;
    auto load_half = [&](int slot, int h, bf16x8 (&af)[NA], bf16x8 (&bfr)[NB]) {
        const char* sl = smem + slot * STAGE_BYTES;
        if (M32) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int ks = h * 2 + k2;
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) af[k2 * 2 + jn] = *(const bf16x8*)(sl + ((aoff + jn * 4096) ^ (ks << 5)));
#pragma unroll
                for (int it = 0; it < 2; ++it) bfr[k2 * 2 + it] = *(const bf16x8*)(sl + ((boff + it * 4096) ^ (ks << 5)));
            }
        } else {
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) af[jn] = *(const bf16x8*)(sl + ((aoff + jn * 512) ^ (h << 6)));
#pragma unroll
            for (int it = 0; it < IT; ++it) bfr[it] = *(const bf16x8*)(sl + ((boff + it * 2048) ^ (h << 6)));
        }
    };
    auto mma_half = [&](const bf16x8 (&af)[NA], const bf16x8 (&bfr)[NB]) {
        if (M32) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int it = 0; it < 2; ++it)
                        acc32[jn * 2 + it] = SL_MFMA32(af[k2 * 2 + jn], bfr[k2 * 2 + it],
                                                                                     acc32[jn * 2 + it]);
        } else {
#pragma unroll
            for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int it = 0; it < IT; ++it)
                    acc[jn * IT + it] = SL_MFMA16(af[jn], bfr[it], acc[jn * IT + it]);
        }
    };

    // Everything the epilogue needs from the kernel arguments is pulled into SGPRs HERE.  Left alone, the compiler parks
    // those scalar loads in the loop pre-header without waiting for them, and a scalar load pending at the loop head
    // forces it to emit lgkmcnt(0) instead of counted waits in front of every MFMA group inside the loop (LDS and
    // scalar loads share the counter and return out of order with respect to each other).
    NtArgs e = a;
    asm volatile("" : "+s"(e.y), "+s"(e.mask), "+s"(e.bias), "+s"(e.partial), "+s"(e.y_bs));
    asm volatile("" : "+s"(e.y_row0), "+s"(e.y_rs), "+s"(e.t_out), "+s"(e.cout), "+s"(e.batch), "+s"(e.t_tiles));

    if constexpr (ILV) {
        // ---- interleaved schedule (see IlvPhase).  Per step and wave:
        //   A: [a0,b0 ready] MFMAs of k-half 0, one fragment read of k-half 1 per group
        //   lgkmcnt(0), counted vmcnt, barrier
        //   B: MFMAs of k-half 1, per group one fragment read of the NEXT tile's k-half 0 and one LDS-DMA request of the
        //      tile that re-uses this slot.  The request stream is branch-free: past the end of the split the last tile
        //      is requested again into a slot nobody reads, so the vmcnt bookkeeping stays exact to the last step.
        constexpr int G = (IT >= 4) ? 2 : 1;  // MFMAs per group: 4*IT / G groups must offer 4+IT read slots (G = 1 for IT = 4: measured slower)
        constexpr int NQ = 4 * IT / G;
        static_assert(NI <= NQ, "one DMA request per group");
        constexpr int RPG = 1;  // (2 for the 4-wave tile -- all reads in the first half of a phase -- measured: no change)
        const __bf16* xs_n = nullptr;
        const __bf16* ws_n = nullptr;
        unsigned xl_n = 0, wl_n = 0;  // LDS byte offsets of this wave's part of the slot being refilled
        // Request stream state: sources of the NEXT activation / weight tile to request, advanced by pointer increments
        // (tap-major order: the weight tiles are consecutive, the activation tile moves one chunk right or wraps to the
        // next tap's row).  The former per-step "step -> (tap, chunk)" division was a ~45-instruction scalar clump
        // between the two MFMA phases; with one wave per SIMD (the 4-wave tiles) nobody covers it.
        const int chunks = a.cin / BK;
        const int tap_first = s_begin / chunks;
        int cc_r = s_begin - tap_first * chunks;
        int rx_left = n, rw_left = n;
        const __bf16* xs_r = xbase + (long)tap_first * a.x_rs + cc_r * BK;
        const __bf16* ws_r = wbase + (long)s_begin * BK;
        const int x_wrap = a.x_rs - (chunks - 1) * BK;
        const unsigned wave_x = (wave * XPW) * 1024, wave_w = X_BYTES + (wave * WPW) * 1024;
        auto next_x = [&](int slot) {  // past the end the last tile is requested again
            xs_n = xs_r;
            xl_n = slot * STAGE_BYTES + wave_x;
            const bool more = rx_left > 1;
            const bool wrap = cc_r + 1 == chunks;
            int dx = wrap ? x_wrap : BK;
            dx = more ? dx : 0;
            xs_r += dx;
            const int cn = wrap ? 0 : cc_r + 1;
            cc_r = more ? cn : cc_r;
            rx_left -= more ? 1 : 0;
        };
        auto next_w = [&](int slot) {
            ws_n = ws_r;
            wl_n = slot * STAGE_BYTES + wave_w;
            const bool more = rw_left > 1;
            ws_r += more ? BK : 0;
            rw_left -= more ? 1 : 0;
        };
        int cur = 0;
        // the address arithmetic rides in phase A, in the shadow of MFMA groups; the asm pins it there (otherwise it sinks
        // below the barrier)
        auto pin_w = [&]() {
            next_w(cur);
            asm volatile("" : "+s"(ws_n), "+s"(wl_n), "+s"(ws_r), "+s"(rw_left));
        };
        auto pin_x = [&]() {
            next_x(cur);
            asm volatile("" : "+s"(xs_n), "+s"(xl_n), "+s"(xs_r), "+s"(cc_r), "+s"(rx_left));
        };
        auto hook_a = [&](auto q_c) {
            constexpr int Q = decltype(q_c)::value;
            if constexpr (Q == 1) pin_w();
            if constexpr (Q == 3) pin_x();
        };
        // (issuing the weight half of a tile's requests in phase A and the activation half in phase B -- half the
        // texture-path load per phase -- was measured: no gain, 0.46 vs 0.45 us per step on the 128x128 tile)
        auto hook_b = [&](auto q_c) {
            constexpr int Q = decltype(q_c)::value;
#if defined(SL_PROBE_NO_DMA)  // timing probe only (wrong results)
            (void)q_c;
#else
            if constexpr (Q < XPW)
                glds16(xs_n + xoff[Q], smem + xl_n + Q * 1024);
            else if constexpr (Q < NI)
                glds16(ws_n + woff[Q - XPW], smem + wl_n + (Q - XPW) * 1024);
#endif
        };
#pragma unroll
        for (int i = 0; i < STAGES; ++i) {
            next_x(i);
            next_w(i);
#pragma unroll
            for (int q = 0; q < XPW; ++q) glds16(xs_n + xoff[q], smem + xl_n + q * 1024);
#pragma unroll
            for (int q = 0; q < WPW; ++q) glds16(ws_n + woff[q], smem + wl_n + q * 1024);
        }
        wait_vmcnt<NI*(STAGES - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8 a0[NA], b0[NB], a1[NA], b1[NB];
        const unsigned lds0 = (unsigned)(size_t)smem;
        DsReadRun<0, NA, 512>::go(a0, lds0 + aoff);
        DsReadRun<0, NB, 2048>::go(b0, lds0 + boff);
#if defined(SL_PROBE_TIMES)  // s_memtime stamps of one steady-state step of wave 0 of work-group 0 -> a.partial (see below)
        unsigned long long tp1 = 0, tp3 = 0, tp4 = 0, tp5 = 0, t1 = 0, t3 = 0, t4 = 0, t5 = 0;
#define SL_STAMP(v) asm volatile("s_memtime %0" : "=s"(v)::"memory")
#else
#define SL_STAMP(v)
#endif
        for (int i = 0; i < n; ++i) {
            const int nxt = (cur + 1 == STAGES) ? 0 : cur + 1;
            wait_frags<0>(a0, b0);
#if defined(SL_PROBE_TIMES)
            tp1 = t1; tp3 = t3; tp4 = t4; tp5 = t5;
#endif
            SL_STAMP(t1);
            IlvPhase<IT, G, 0, NQ, RPG>::run(acc, a0, b0, a1, b1, lds0 + cur * STAGE_BYTES + (aoff ^ 64),
                                             lds0 + cur * STAGE_BYTES + (boff ^ 64), hook_a);
            wait_frags<0>(a1, b1);  // my reads of slot cur are complete
            SL_STAMP(t3);
            wait_vmcnt<NI*(STAGES - 2)>();  // tile i+1 has landed; the younger ones stay in flight
            SL_STAMP(t4);
#if !defined(SL_PROBE_NO_BARRIER)  // timing probe only (wrong results)
            __builtin_amdgcn_s_barrier();
#endif
            SL_STAMP(t5);
            asm volatile("" ::: "memory");
            IlvPhase<IT, G, 0, NQ, RPG>::run(acc, a1, b1, a0, b0, lds0 + nxt * STAGE_BYTES + aoff,
                                             lds0 + nxt * STAGE_BYTES + boff, hook_b);
            cur = nxt;
        }
        wait_vmcnt<0>();        // the surplus requests still target this work-group's LDS
        wait_frags<0>(a0, b0);  // ... and the surplus fragment reads these registers
#if defined(SL_PROBE_TIMES)
        if (id == 0 && tid == 0 && a.partial != nullptr) {
            unsigned long long* o = (unsigned long long*)a.partial;
            o[0] = tp1; o[1] = tp3; o[2] = tp4; o[3] = tp5; o[4] = t1; o[5] = (unsigned long long)n;
        }
#endif
    } else if constexpr (PIPE && !M32) {
        // Register-pipelined ring.  All STAGES slots are filled up front; tile i's slot is refilled with tile
        // i+STAGES at the barrier in the MIDDLE of iteration i, by which time every wave has its whole tile i in
        // registers.  That same barrier publishes tile i+1, whose first-half fragments are then read from LDS while
        // the MFMAs of tile i's second half run; the second-half reads overlap the first-half MFMAs.  The LDS read
        // latency, which the plain loop exposes after every barrier (all waves read, then all waves multiply), is
        // hidden behind MFMA work: measured step time on the 250-channel layers 0.61 us -> see HISTORY.md section 3.1.
#pragma unroll
        for (int i = 0; i < STAGES; ++i)
            if (i < n) stage(s_begin + i, i);
        if (n >= STAGES)
            wait_vmcnt<NI*(STAGES - 1)>();
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8 a0[NA], b0[NB], a1[NA], b1[NB];
        constexpr int NF = NA + NB;  // LDS reads per half
        // per-half read addresses (the 64-B half flips bit 6 of the swizzled slot); + slot * STAGE_BYTES per tile
        const unsigned a_addr0 = (unsigned)(size_t)smem + aoff, a_addr1 = a_addr0 ^ 64u;
        const unsigned b_addr0 = (unsigned)(size_t)smem + boff, b_addr1 = b_addr0 ^ 64u;
        auto read_half = [&](int slot, int h, bf16x8 (&af)[NA], bf16x8 (&bfr)[NB]) {
            const unsigned so = slot * STAGE_BYTES;
            DsReadRun<0, NA, 512>::go(af, (h ? a_addr1 : a_addr0) + so);
            DsReadRun<0, NB, 2048>::go(bfr, (h ? b_addr1 : b_addr0) + so);
        };
        read_half(0, 0, a0, b0);
        int cur = 0;
        // the last tile is peeled off so that the loop body is branch-free around the LDS reads
        for (int i = 0; i + 1 < n; ++i) {
            const int nxt = (cur + 1 == STAGES) ? 0 : cur + 1;
            read_half(cur, 1, a1, b1);
            wait_frags<NF>(a0, b0);  // first half in registers, second half in flight
            mma_half(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            wait_frags<0>(a1, b1);  // my reads of slot cur are complete
            if (i + STAGES <= n)
                wait_vmcnt<NI*(STAGES - 2)>();  // tile i+1 has landed; the younger ones stay in flight
            else
                wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (i + STAGES < n) stage(s_begin + i + STAGES, cur);
            read_half(nxt, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);  // all reads are issued before the MFMAs they hide behind
            mma_half(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        read_half(cur, 1, a1, b1);
        wait_frags<NF>(a0, b0);
        mma_half(a0, b0);
        wait_frags<0>(a1, b1);
        mma_half(a1, b1);
    } else {
        // plain ring: wait for tile i, barrier, refill the slot consumed one iteration earlier, consume tile i
#pragma unroll
        for (int i = 0; i < STAGES - 1; ++i)
            if (i < n) stage(s_begin + i, i);
        int cur = 0;           // ring slot of tile i
        int nxt = STAGES - 1;  // ring slot tile i+STAGES-1 goes to
        for (int i = 0; i < n; ++i) {
            if (i + STAGES - 1 <= n)
                wait_vmcnt<NI*(STAGES - 2)>();  // tile i has landed; the younger ones stay in flight
            else
                wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // everyone's share landed; everyone finished reading the slot refilled below
            asm volatile("" ::: "memory");
            if (i + STAGES - 1 < n) stage(s_begin + i + STAGES - 1, nxt);
            bf16x8 a0[NA], b0[NB], a1[NA], b1[NB];
            load_half(cur, 0, a0, b0);
            load_half(cur, 1, a1, b1);
            mma_half(a0, b0);
            mma_half(a1, b1);
            cur = (cur + 1 == STAGES) ? 0 : cur + 1;
            nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
        }
    }

    // ---- epilogue: the lane holds runs of 16 consecutive channels of a time row
    //   16x16 shape: per it one run   at co0 + wn*64 + (lane>>4)*16, row t0 + wm*WROWS + it*16 + (lane&15)
    //   32x32 shape: per (it, jn) one at co0 + wn*64 + jn*32 + (lane>>5)*16, row t0 + wm*64 + it*32 + (lane&31)
    constexpr int NRUN_T = M32 ? 2 : IT;
    constexpr int NRUN_C = M32 ? 2 : 1;
    const int lane_t = M32 ? (lane & 31) : (lane & 15);
    const int lane_c = M32 ? (lane >> 5) * 16 : (lane >> 4) * 16;
#pragma unroll
    for (int jc = 0; jc < NRUN_C; ++jc) {
        const int co_base = co0 + wn * 64 + jc * 32 + lane_c;
        float bias_v[16];
        if (MODE == SL_EPI_BIAS || MODE == SL_EPI_BIAS_RELU || MODE == SL_EPI_BIAS_ELU)
            load_bias16(e.bias, co_base, bias_v);
#pragma unroll
        for (int it = 0; it < NRUN_T; ++it) {
            const int trow = wm * WROWS + it * (M32 ? 32 : 16) + lane_t;
            float v[16];
            if (M32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc32[jc * 2 + it][r];
            } else {
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[jn * 4 + r] = acc[jn * IT + it][r];
            }
            if (MODE == MODE_PARTIAL) {
                float* out = e.partial + ((long)(split * e.batch + b) * (e.t_tiles * BM) + t0 + trow) * e.cout + co_base;
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(out + i * 4) = (f32x4){v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
            } else {
                const int t = t0 + trow;
                if (t < e.t_out) {
                    const long yidx = (long)b * e.y_bs + (long)(e.y_row0 + t) * e.y_rs + co_base;
                    store_run16<MODE, OUT_F32>(e, v, bias_v, yidx);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Slab variant (16x16x32 shape only).  Contraction order chunk OUTER, tap INNER: for one 64-channel chunk the BM + taps - 1
// activation rows a tile needs for ALL taps are brought into LDS ONCE (the "slab", double buffered) and every tap reads its
// B fragments from the slab at a row offset, so that a step only streams the BN x 64 weight tile.  L2 -> LDS bytes per
// step drop from (BM + BN) * 128 to BN * 128 + BM * 128 / taps (inner layers, taps = 7: 32 -> 18.3 KB; big_conv_1, taps =
// 32, 256x256 tile: 64 -> 33 KB) and the activation re-reads that used to miss the L2 between taps (big_conv_1 dgrad:
// ~2 GB per launch out of the Infinity Cache) disappear.  The XOR swizzle key is the SLAB row & 7, so a fragment read at
// row offset `tap` stays bank-conflict free for every tap (checked exhaustively, HISTORY.md section 3.1).
//   LDS: [slab 0][slab 1][weight ring: STAGES slots];   slab rows = BM + 32  (taps <= 33)
//   vmcnt bookkeeping: the slab of chunk c+1 is issued in step (c, tap 0) BEFORE that step's weight tile, so it is older
//   than every weight tile of chunk c+1 (needs taps >= STAGES) and only the steps with tap in [1, STAGES-2] see it among
//   the instructions that may stay in flight.
template <int IT, int WM, int WN, int STAGES_P, int MODE, bool OUT_F32, bool ILV>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 16 ? 4 : 2)) void conv_nt_slab_bf16_kernel(NtArgs a) {
    constexpr int STAGES = STAGES_P & 7;
    constexpr bool PIPE = (STAGES_P & 8) != 0;
    static_assert(!ILV || PIPE, "the interleaved schedule is a variant of the register-pipelined loop");
    static_assert(STAGES >= 2, "ring too shallow");
    constexpr int NW = WM * WN;
    constexpr int WROWS = 16 * IT;
    constexpr int BM = WROWS * WM;
    constexpr int BN = 64 * WN;
    constexpr int SLAB_GROUPS = BM / 8 + 4;  // 8-row DMA groups per slab
    constexpr int SLAB_BYTES = SLAB_GROUPS * 1024;
    constexpr int W_BYTES = BN * 128;
    constexpr int XPW = (SLAB_GROUPS + NW - 1) / NW;  // slab DMA instructions per wave per chunk
    constexpr int WPW = (BN / 8) / NW;                // weight-tile DMA instructions per wave per step
    static_assert((BN / 8) % NW == 0, "weight tile rows must split evenly over the waves");
    constexpr int RING0 = 2 * SLAB_BYTES;

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
    const int wm = wave / WN;
    const int wn = wave % WN;

    const int m_tiles = a.batch * a.t_tiles;
    const int tiles = m_tiles * a.n_tiles;
    const int id = xcd_remap(blockIdx.x, tiles * a.ksplit);
    if (id >= tiles * a.ksplit) return;  // grid padding (xcd_grid)
    const int split = id / tiles;
    const int tile = id - split * tiles;
    const int span = a.n_tiles * a.gm;  // 2-D raster, see conv_nt_bf16_kernel
    const int blk = tile / span;
    const int rem = tile - blk * span;
    int cnt = m_tiles - blk * a.gm;
    if (cnt > a.gm) cnt = a.gm;
    const int n_tile = rem / cnt;
    const int m_tile = blk * a.gm + (rem - n_tile * cnt);
    const int b = m_tile / a.t_tiles;
    const int t0 = (m_tile - b * a.t_tiles) * BM;
    const int co0 = n_tile * BN;
    const int s_begin = split * a.steps_per_split;  // a multiple of taps (whole chunks per split)
    int n = a.nsteps - s_begin;
    if (n > a.steps_per_split) n = a.steps_per_split;
    const int taps = a.taps;
    const int c0 = s_begin / taps;
    const int nchunks = n / taps;

    const __bf16* xbase = a.x + (long)b * a.x_bs + (long)(a.x_row0 + t0) * a.x_rs;
    const __bf16* wbase = a.w + (long)co0 * a.w_rs;
    int xoff[XPW], xdst[XPW], woff[WPW];
    const int last_row = BM + taps - 2;  // last slab row any tap reads; rows behind it are clamped (never read from LDS)
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
        int j = wave * XPW + q;
        if (j > SLAB_GROUPS - 1) j = SLAB_GROUPS - 1;
        int row = j * 8 + (lane >> 3);
        if (row > last_row) row = last_row;
        xoff[q] = row * a.x_rs + (((lane & 7) ^ (lane >> 3)) << 3);
        xdst[q] = j * 1024;
    }
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
        const int j = wave * WPW + q;
        const int row = j * 8 + (lane >> 3);
        const int key = ((lane >> 4) & 1) | (((j >> 1) & 3) << 1);
        woff[q] = row * a.w_rs + (((lane & 7) ^ key) << 3);
    }
    auto issue_slab = [&](int cc, int par) {
        const __bf16* xs = xbase + cc * BK;
#pragma unroll
        for (int q = 0; q < XPW; ++q) glds16(xs + xoff[q], smem + par * SLAB_BYTES + xdst[q]);
    };
    auto issue_w = [&](int tap, int cc, int slot) {
        const __bf16* ws = wbase + (long)tap * a.cin + cc * BK;
        char* wl = smem + RING0 + slot * W_BYTES + (wave * WPW) * 1024;
#pragma unroll
        for (int q = 0; q < WPW; ++q) glds16(ws + woff[q], wl + q * 1024);
    };

    const int g = lane >> 4;
    const int brow = wm * WROWS + (lane & 15);                        // + it*16 + tap
    const int arow = wn * 64 + ((lane & 15) >> 2) * 16 + (lane & 3);  // + jn*4
    const int akey = ((lane >> 1) & 1) | (((lane >> 2) & 3) << 1);
    const int aoff = RING0 + arow * 128 + ((g ^ akey) << 4);
    auto b_offset = [&](int par, int tap) {  // byte offset of this lane's B fragment (it = 0, first half) in LDS
        const int r = brow + tap;
        return par * SLAB_BYTES + (r << 7) + (((g ^ r) & 7) << 4);
    };

    f32x4 acc[4 * IT];
#pragma unroll
    for (int i = 0; i < 4 * IT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma_half = [&](const bf16x8 (&af)[4], const bf16x8 (&bfr)[IT]) {
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int it = 0; it < IT; ++it)
                acc[jn * IT + it] = SL_MFMA16(af[jn], bfr[it], acc[jn * IT + it]);
    };

    NtArgs e = a;  // epilogue arguments pinned in SGPRs before the loop, see conv_nt_bf16_kernel
    asm volatile("" : "+s"(e.y), "+s"(e.mask), "+s"(e.bias), "+s"(e.partial), "+s"(e.y_bs));
    asm volatile("" : "+s"(e.y_row0), "+s"(e.y_rs), "+s"(e.t_out), "+s"(e.cout), "+s"(e.batch), "+s"(e.t_tiles));

    // issue-side counters (the next weight tile to request) and compute-side counters (the step being multiplied)
    int tap_i = 0, cc_i = c0;
    auto issue_next_w = [&](int slot) {
        issue_w(tap_i, cc_i, slot);
        if (++tap_i == taps) {
            tap_i = 0;
            ++cc_i;
        }
    };
    int tap_c = 0, par_c = 0, chunk_c = 0;  // chunk_c counts chunks of this split

    if constexpr (ILV) {
        // ---- interleaved schedule.  Per step and wave:
        //   A: [a0,b0 ready] 4*IT MFMAs of k-half 0, one read of k-half 1 after every second MFMA
        //   lgkmcnt(0), vmcnt, barrier
        //   B: 4*IT MFMAs of k-half 1, one read of the NEXT tile's k-half 0 after every second MFMA, then the request
        //      of the weight tile that re-uses this slot (branch-free: past the end the last tile is requested again
        //      into a slot nobody reads) somewhere among the remaining MFMAs
        // All per-step scalar / address work (next weight tile's source and slot, the compute-side counters, the LDS
        // address of the next step's activation fragments) rides in phase A in the shadow of MFMA groups, pinned there by
        // asm: left between the barrier and phase B it is a ~40-instruction clump that every wave of the work-group
        // executes at the same moment with the matrix pipe empty (the waves are in lockstep, so the second wave of a
        // SIMD does not cover it).
        int left = n;                     // requests that still advance the stream (past the end: the last tile again)
        const __bf16* ws_r = wbase + (long)c0 * BK;  // source of the next weight tile to request (tap 0 of chunk c0)
        int tap_r = 0;
        const int w_wrap = BK - (taps - 1) * a.cin;  // element step from the last tap of a chunk to tap 0 of the next
        const __bf16* ws_next = nullptr;  // source / LDS offset of the weight tile phase B requests piecewise
        unsigned wl_next = 0;
        const unsigned wave_w = RING0 + (wave * WPW) * 1024;
        auto begin_w = [&](int slot) {
            ws_next = ws_r;
            wl_next = slot * W_BYTES + wave_w;
            const bool more = left > 1;
            const bool wrap = tap_r + 1 == taps;
            int dw = wrap ? w_wrap : a.cin;
            dw = more ? dw : 0;
            ws_r += dw;
            tap_r = wrap ? 0 : tap_r + 1;  // (keeps cycling past the end: dw is zero there)
            left = more ? left - 1 : left;
        };
        auto dma_hook = [&](auto q_c) {  // groups 0..WPW-1 of phase B each carry one request of the tile
            constexpr int Q = decltype(q_c)::value;
#if defined(SL_PROBE_NO_DMA)  // timing probes only (wrong results): how much of the step is waiting for the weight stream
            (void)q_c;
#elif defined(SL_PROBE_HALF_DMA)
            if constexpr (Q < WPW / 2) glds16(ws_next + woff[Q], smem + wl_next + Q * 1024);
#else
            if constexpr (Q < WPW) glds16(ws_next + woff[Q], smem + wl_next + Q * 1024);
#endif
        };
        constexpr int GS = 2;  // MFMAs per interleave group, see conv_nt_bf16_kernel
        issue_slab(c0, 0);
#pragma unroll
        for (int i = 0; i < STAGES; ++i) {
            begin_w(i);
#pragma unroll
            for (int q = 0; q < WPW; ++q) glds16(ws_next + woff[q], smem + wl_next + q * 1024);
        }
        wait_vmcnt<WPW*(STAGES - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8 a0[4], b0[IT], a1[4], b1[IT];
        const unsigned lds0 = (unsigned)(size_t)smem;
        DsReadRun<0, 4, 512>::go(a0, lds0 + aoff);
        DsReadRun<0, IT, 2048>::go(b0, lds0 + b_offset(0, 0));
        int cur = 0;
        // what phase A's hooks prepare for the barrier and phase B of the same step
        int slab_chunk = -1, slab_par = 0;  // slab to request right after the barrier (-1: none)
        int wait_slab = 0;                  // STAGES > 2: tap of the step if a slab request may be among the loads in flight
        unsigned b_next = 0;                // LDS address of the next step's first-half activation fragments
        auto pin_w = [&]() {
            begin_w(cur);
            asm volatile("" : "+s"(ws_next), "+s"(wl_next), "+s"(ws_r));
        };
        auto pin_counters = [&]() {
            const bool has_next = chunk_c + 1 < nchunks;
            slab_chunk = (tap_c == 0 && has_next) ? c0 + chunk_c + 1 : -1;
            slab_par = par_c ^ 1;
            wait_slab = has_next ? tap_c : 0;  // in [1, STAGES-2]: the slab request may stay in flight at the wait
            const bool wrap = tap_c + 1 == taps;
            chunk_c = wrap ? chunk_c + 1 : chunk_c;
            par_c = chunk_c & 1;
            tap_c = wrap ? 0 : tap_c + 1;
            asm volatile("" : "+s"(slab_chunk), "+s"(slab_par), "+s"(chunk_c), "+s"(tap_c));
        };
        auto pin_b_next = [&]() {
            b_next = lds0 + b_offset(par_c, tap_c);
            asm volatile("" : "+v"(b_next));
        };
        auto hook_a = [&](auto q_c) {
            constexpr int Q = decltype(q_c)::value;
            if constexpr (Q == 1) pin_w();
            if constexpr (Q == 3) pin_counters();
            if constexpr (Q == 5) pin_b_next();
        };
        static_assert(4 * IT / GS > 5, "phase A needs six groups for its hooks");
        for (int i = 0; i < n; ++i) {
            const int nxt = (cur + 1 == STAGES) ? 0 : cur + 1;
            wait_frags<0>(a0, b0);
            IlvPhase<IT, GS, 0, 4 * IT / GS>::run(acc, a0, b0, a1, b1, lds0 + ((aoff + cur * W_BYTES) ^ 64),
                                          lds0 + (b_offset(par_c, tap_c) ^ 64), hook_a);
            wait_frags<0>(a1, b1);  // my reads of weight slot cur (and, on a chunk's last tap, of its slab) are complete
            if constexpr (STAGES == 2) {
                wait_vmcnt<0>();
            } else {  // weight tile i+1 has landed; the STAGES-2 younger ones (and a slab requested among them) fly on
                if (wait_slab >= 1 && wait_slab <= STAGES - 2)
                    wait_vmcnt<WPW*(STAGES - 2) + XPW>();
                else
                    wait_vmcnt<WPW*(STAGES - 2)>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (slab_chunk >= 0) issue_slab(slab_chunk, slab_par);  // once per chunk, in front of the straight-line phase
            IlvPhase<IT, GS, 0, 4 * IT / GS>::run(acc, a1, b1, a0, b0, lds0 + aoff + nxt * W_BYTES, b_next, dma_hook);
            cur = nxt;
        }
        wait_vmcnt<0>();       // the surplus requests still target this work-group's LDS
        wait_frags<0>(a0, b0);  // ... and the surplus fragment reads these registers
    } else if constexpr (PIPE) {
        issue_slab(c0, 0);
#pragma unroll
        for (int i = 0; i < STAGES; ++i)
            if (i < n) issue_next_w(i);
        if (n >= STAGES)
            wait_vmcnt<WPW*(STAGES - 1)>();
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8 a0[4], b0[IT], a1[4], b1[IT];
        constexpr int NF = 4 + IT;
        const unsigned lds0 = (unsigned)(size_t)smem;
        auto read_half = [&](int slot, int par, int tap, int h, bf16x8 (&af)[4], bf16x8 (&bfr)[IT]) {
            DsReadRun<0, 4, 512>::go(af, lds0 + ((aoff + slot * W_BYTES) ^ (h << 6)));
            DsReadRun<0, IT, 2048>::go(bfr, lds0 + (b_offset(par, tap) ^ (h << 6)));
        };
        read_half(0, 0, 0, 0, a0, b0);
        int cur = 0;
        for (int i = 0; i + 1 < n; ++i) {
            const int nxt = (cur + 1 == STAGES) ? 0 : cur + 1;
            read_half(cur, par_c, tap_c, 1, a1, b1);
            wait_frags<NF>(a0, b0);
            mma_half(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            wait_frags<0>(a1, b1);  // my reads of weight slot cur (and, on a chunk's last tap, of its slab) are complete
            const bool has_next_chunk = chunk_c + 1 < nchunks;
            if (i + STAGES <= n) {
                if (STAGES > 2 && has_next_chunk && tap_c >= 1 && tap_c <= STAGES - 2)
                    wait_vmcnt<WPW*(STAGES - 2) + XPW>();  // the next chunk's slab may stay in flight as well
                else
                    wait_vmcnt<WPW*(STAGES - 2)>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tap_c == 0 && has_next_chunk) issue_slab(c0 + chunk_c + 1, par_c ^ 1);
            if (i + STAGES < n) issue_next_w(cur);
            if (++tap_c == taps) {
                tap_c = 0;
                par_c ^= 1;
                ++chunk_c;
            }
            read_half(nxt, par_c, tap_c, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma_half(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        read_half(cur, par_c, tap_c, 1, a1, b1);
        wait_frags<NF>(a0, b0);
        mma_half(a0, b0);
        wait_frags<0>(a1, b1);
        mma_half(a1, b1);
    } else {
        issue_slab(c0, 0);
#pragma unroll
        for (int i = 0; i < STAGES - 1; ++i)
            if (i < n) issue_next_w(i);
        int cur = 0;
        int nxt = STAGES - 1;
        for (int i = 0; i < n; ++i) {
            const bool has_next_chunk = chunk_c + 1 < nchunks;
            if (i + STAGES - 1 <= n) {
                if (STAGES > 2 && has_next_chunk && tap_c >= 1 && tap_c <= STAGES - 2)
                    wait_vmcnt<WPW*(STAGES - 2) + XPW>();
                else
                    wait_vmcnt<WPW*(STAGES - 2)>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tap_c == 0 && has_next_chunk) issue_slab(c0 + chunk_c + 1, par_c ^ 1);
            if (i + STAGES - 1 < n) issue_next_w(nxt);
            const char* wl = smem + cur * W_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[4], bfr[IT];
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) af[jn] = *(const bf16x8*)(wl + ((aoff + jn * 512) ^ (kk << 6)));
#pragma unroll
                for (int it = 0; it < IT; ++it)
                    bfr[it] = *(const bf16x8*)(smem + ((b_offset(par_c, tap_c) + it * 2048) ^ (kk << 6)));
                mma_half(af, bfr);
            }
            if (++tap_c == taps) {
                tap_c = 0;
                par_c ^= 1;
                ++chunk_c;
            }
            cur = (cur + 1 == STAGES) ? 0 : cur + 1;
            nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
        }
    }

    // ---- epilogue (identical to conv_nt_bf16_kernel's 16x16 path)
    const int co_base = co0 + wn * 64 + (lane >> 4) * 16;
    float bias_v[16];
    if (MODE == SL_EPI_BIAS || MODE == SL_EPI_BIAS_RELU || MODE == SL_EPI_BIAS_ELU) load_bias16(e.bias, co_base, bias_v);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int trow = wm * WROWS + it * 16 + (lane & 15);
        float v[16];
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[jn * 4 + r] = acc[jn * IT + it][r];
        if (MODE == MODE_PARTIAL) {
            float* out = e.partial + ((long)(split * e.batch + b) * (e.t_tiles * BM) + t0 + trow) * e.cout + co_base;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(out + i * 4) = (f32x4){v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
        } else {
            const int t = t0 + trow;
            if (t < e.t_out) {
                const long yidx = (long)b * e.y_bs + (long)(e.y_row0 + t) * e.y_rs + co_base;
                store_run16<MODE, OUT_F32>(e, v, bias_v, yidx);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 128x128 tile, EIGHT waves: the four 64x64 patches of the 4-wave interleaved kernel, each owned by the two waves that
// share a SIMD (waves p and p + 4); wave p multiplies k-half 0 of every 64-deep step, wave p + 4 k-half 1, into private
// accumulators that are added through LDS in the epilogue.  Same LDS bytes and MFMAs per step as four waves, but two
// instruction streams per SIMD: a single wave issues in order, so every wait, request and address instruction it cannot
// hide behind its own MFMAs is exposed (HISTORY.md section 3.1: 0.45 us per step against 0.244 us of MFMAs).
// One phase and one barrier per step: [frags(i) in registers] wait for tile i+1, barrier, 16 MFMAs on tile i's
// fragments with the 8 fragment reads of tile i+1 and the 4 requests of tile i+STAGES (into tile i's slot, which every
// wave has finished reading before the barrier) interleaved.  Tap-major contraction, branch-free request stream.
template <int STAGES, int MODE, bool OUT_F32>
__global__ __launch_bounds__(512, 2) void conv_nt_ks2_bf16_kernel(NtArgs a) {
    constexpr int IT = 4, WN = 2, NW = 8;
    constexpr int WROWS = 64, BM = 128, BN = 128;
    constexpr int X_BYTES = BM * 128;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int XPW = (BM / 8) / NW, WPW = (BN / 8) / NW, NI = XPW + WPW;  // 2 + 2 requests per wave and tile
    static_assert(STAGES >= 3 && STAGES * STAGE_BYTES <= 160 * 1024, "ring depth");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2;  // the k-half this wave multiplies
    const int pw = wave & 3;   // the patch it shares with wave pw + 4 * (1 - kh)
    const int wm = pw / WN;
    const int wn = pw % WN;

    const int m_tiles = a.batch * a.t_tiles;
    const int tiles = m_tiles * a.n_tiles;
    const int id = xcd_remap(blockIdx.x, tiles * a.ksplit);
    if (id >= tiles * a.ksplit) return;  // grid padding (xcd_grid)
    const int split = id / tiles;
    const int tile = id - split * tiles;
    const int span = a.n_tiles * a.gm;  // 2-D raster, see conv_nt_bf16_kernel
    const int blk = tile / span;
    const int rem = tile - blk * span;
    int cnt = m_tiles - blk * a.gm;
    if (cnt > a.gm) cnt = a.gm;
    const int n_tile = rem / cnt;
    const int m_tile = blk * a.gm + (rem - n_tile * cnt);
    const int b = m_tile / a.t_tiles;
    const int t0 = (m_tile - b * a.t_tiles) * BM;
    const int co0 = n_tile * BN;
    const int s_begin = split * a.steps_per_split;
    int n = a.nsteps - s_begin;
    if (n > a.steps_per_split) n = a.steps_per_split;

    const __bf16* xbase = a.x + (long)b * a.x_bs + (long)(a.x_row0 + t0) * a.x_rs;
    const __bf16* wbase = a.w + (long)co0 * a.w_rs;
    int xoff[XPW], woff[WPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
        const int j = wave * XPW + q;
        const int row = j * 8 + (lane >> 3);
        xoff[q] = row * a.x_rs + (((lane & 7) ^ (lane >> 3)) << 3);
    }
#pragma unroll
    for (int q = 0; q < WPW; ++q) {
        const int j = wave * WPW + q;
        const int row = j * 8 + (lane >> 3);
        const int key = ((lane >> 4) & 1) | (((j >> 1) & 3) << 1);
        woff[q] = row * a.w_rs + (((lane & 7) ^ key) << 3);
    }
    const int g = lane >> 4;
    const int brow = wm * WROWS + (lane & 15);
    const int arow = wn * 64 + ((lane & 15) >> 2) * 16 + (lane & 3);
    const int akey = ((lane >> 1) & 1) | (((lane >> 2) & 3) << 1);
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned boff = lds0 + ((brow * 128 + ((g ^ (lane & 7)) << 4)) ^ (kh << 6));
    const unsigned aoff = lds0 + ((X_BYTES + arow * 128 + ((g ^ akey) << 4)) ^ (kh << 6));

    f32x4 acc[4 * IT];
#pragma unroll
    for (int i = 0; i < 4 * IT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    NtArgs e = a;  // epilogue arguments pinned in SGPRs before the loop, see conv_nt_bf16_kernel
    asm volatile("" : "+s"(e.y), "+s"(e.mask), "+s"(e.bias), "+s"(e.partial), "+s"(e.y_bs));
    asm volatile("" : "+s"(e.y_row0), "+s"(e.y_rs), "+s"(e.t_out), "+s"(e.cout), "+s"(e.batch), "+s"(e.t_tiles));

    // request stream (see conv_nt_bf16_kernel)
    const __bf16* xs_n = nullptr;
    const __bf16* ws_n = nullptr;
    unsigned xl_n = 0, wl_n = 0;
    const int chunks = a.cin / BK;
    const int tap_first = s_begin / chunks;
    int cc_r = s_begin - tap_first * chunks;
    int r_left = n;
    const __bf16* xs_r = xbase + (long)tap_first * a.x_rs + cc_r * BK;
    const __bf16* ws_r = wbase + (long)s_begin * BK;
    const int x_wrap = a.x_rs - (chunks - 1) * BK;
    const unsigned wave_x = (wave * XPW) * 1024, wave_w = X_BYTES + (wave * WPW) * 1024;
    auto next_tile = [&](int slot) {  // past the end the last tile is requested again (into a slot nobody reads)
        xs_n = xs_r;
        ws_n = ws_r;
        xl_n = slot * STAGE_BYTES + wave_x;
        wl_n = slot * STAGE_BYTES + wave_w;
        const bool more = r_left > 1;
        const bool wrap = cc_r + 1 == chunks;
        int dx = wrap ? x_wrap : BK;
        dx = more ? dx : 0;
        xs_r += dx;
        ws_r += more ? BK : 0;
        const int cn = wrap ? 0 : cc_r + 1;
        cc_r = more ? cn : cc_r;
        r_left = more ? r_left - 1 : r_left;
    };
    int cur = 0;
    auto pin_next = [&]() {
        next_tile(cur);
        asm volatile("" : "+s"(xs_n), "+s"(ws_n), "+s"(xl_n), "+s"(wl_n), "+s"(xs_r), "+s"(ws_r));
    };
    auto hook = [&](auto q_c) {  // group 0: addresses of the tile that re-uses slot cur; groups 1..4: its requests
        constexpr int Q = decltype(q_c)::value;
        if constexpr (Q == 0) pin_next();
        if constexpr (Q >= 1 && Q <= XPW) glds16(xs_n + xoff[Q - 1], smem + xl_n + (Q - 1) * 1024);
        if constexpr (Q > XPW && Q <= NI) glds16(ws_n + woff[Q - 1 - XPW], smem + wl_n + (Q - 1 - XPW) * 1024);
    };
#pragma unroll
    for (int i = 0; i < STAGES; ++i) {
        next_tile(i);
#pragma unroll
        for (int q = 0; q < XPW; ++q) glds16(xs_n + xoff[q], smem + xl_n + q * 1024);
#pragma unroll
        for (int q = 0; q < WPW; ++q) glds16(ws_n + woff[q], smem + wl_n + q * 1024);
    }
    wait_vmcnt<NI*(STAGES - 1)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 a0[4], b0[IT], a1[4], b1[IT];
    DsReadRun<0, 4, 512>::go(a0, aoff);
    DsReadRun<0, IT, 2048>::go(b0, boff);
    auto step = [&](bf16x8 (&ca)[4], bf16x8 (&cb)[IT], bf16x8 (&na)[4], bf16x8 (&nb)[IT]) {
        const int nxt = (cur + 1 == STAGES) ? 0 : cur + 1;
        wait_frags<0>(ca, cb);          // tile i's fragments are in registers: nobody reads slot cur any more ...
        wait_vmcnt<NI*(STAGES - 2)>();  // ... my part of tile i+1 has landed; the younger tiles stay in flight
        __builtin_amdgcn_s_barrier();   // ... and so has everybody else's
        asm volatile("" ::: "memory");
        IlvPhase<IT, 2, 0, 2 * IT>::run(acc, ca, cb, na, nb, aoff + nxt * STAGE_BYTES, boff + nxt * STAGE_BYTES, hook);
        cur = nxt;
    };
    for (int i = 0; i + 1 < n; i += 2) {
        step(a0, b0, a1, b1);
        step(a1, b1, a0, b0);
    }
    if (n & 1) {
        step(a0, b0, a1, b1);
        wait_frags<0>(a1, b1);  // the surplus fragment reads target these registers
    } else {
        wait_frags<0>(a0, b0);
    }
    wait_vmcnt<0>();  // the surplus requests still target this work-group's LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- the k-half-1 waves hand their accumulators over through LDS (16 KiB per patch, lane-linear 16-B slots)
    f32x4* xch = (f32x4*)smem + (pw * 16) * 64 + lane;
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 4 * IT; ++i) xch[i * 64] = acc[i];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < 4 * IT; ++i) acc[i] += xch[i * 64];

    // ---- epilogue (identical to conv_nt_bf16_kernel's 16x16 path)
    const int co_base = co0 + wn * 64 + (lane >> 4) * 16;
    float bias_v[16];
    if (MODE == SL_EPI_BIAS || MODE == SL_EPI_BIAS_RELU || MODE == SL_EPI_BIAS_ELU) load_bias16(e.bias, co_base, bias_v);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int trow = wm * WROWS + it * 16 + (lane & 15);
        float v[16];
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[jn * 4 + r] = acc[jn * IT + it][r];
        if (MODE == MODE_PARTIAL) {
            float* out = e.partial + ((long)(split * e.batch + b) * (e.t_tiles * BM) + t0 + trow) * e.cout + co_base;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(f32x4*)(out + i * 4) = (f32x4){v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
        } else {
            const int t = t0 + trow;
            if (t < e.t_out) {
                const long yidx = (long)b * e.y_bs + (long)(e.y_row0 + t) * e.y_rs + co_base;
                store_run16<MODE, OUT_F32>(e, v, bias_v, yidx);
            }
        }
    }
}

// Softmax tail shared by the two fused output-layer kernels.  Lane (g, i) holds, for time row t0 + wave * 16 + i, the logits
// of classes tile * 16 + 4 g + r in acc[tile][r] (bias not yet added).
__device__ __forceinline__ void output_softmax_finish(const f32x4 (&acc)[2], const float* __restrict__ bias,
                                                      float* __restrict__ probs, float* __restrict__ logq,
                                                      float* __restrict__ logits, int b, int t0, int wave, int lane, int g,
                                                      int t_out, int k, int logit_stride, long logit_batch_stride,
                                                      float eps) {
    // ---- lane (g, i): time row t0 + wave*16 + i, classes tile*16 + 4g + r.  Softmax over the k valid classes of the row:
    // 8 values here, the rest in the three lanes that differ in g (lane ^ 16, lane ^ 32).
    const int t = t0 + wave * 16 + (lane & 15);
    float z[8];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cls = j * 16 + 4 * g + r;
            z[j * 4 + r] = cls < k ? acc[j][r] + bias[cls] : -INFINITY;
            m = fmaxf(m, z[j * 4 + r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float e[8], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = z[i] == -INFINITY ? 0.f : expf(z[i] - m);
        sum += e[i];
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // q = (p + eps) / sum_j (p_j + eps), computed the way TF does: log-softmax of u = log(p + eps)   (sl_softmax_logq)
    float u[8], um = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = e[i] / sum;
        u[i] = z[i] == -INFINITY ? -INFINITY : logf(e[i] + eps);
        um = fmaxf(um, u[i]);
    }
    um = fmaxf(um, __shfl_xor(um, 16));
    um = fmaxf(um, __shfl_xor(um, 32));
    float usum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) usum += u[i] == -INFINITY ? 0.f : expf(u[i] - um);
    usum += __shfl_xor(usum, 16);
    usum += __shfl_xor(usum, 32);
    const float lz = um + logf(usum);
    if (t < t_out) {
        const long f = (long)b * t_out + t;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cls = j * 16 + 4 * g + r;
                if (cls < k) {
                    probs[f * k + cls] = e[j * 4 + r];
                    logq[f * k + cls] = u[j * 4 + r] - lz;
                    if (logits) logits[(long)b * logit_batch_stride + (long)t * logit_stride + cls] = z[j * 4 + r];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Output layer: 1x1 convolution onto k <= 32 classes fused with the softmax and the log(p + eps) re-normalisation that
// sl_softmax_logq computes.  As an NT launch this layer is a 128x128 tile of which three quarters is channel padding
// (34 us, and another launch for the softmax); what it really needs is to stream its input once.  Here a work-group owns
// 64 time rows and ALL classes: the whole weight matrix (32 x cin bf16, 128 KiB at cin = 2048, rows padded by 16 B
// against bank conflicts) sits in LDS, the activation rows come through a 3-slot LDS-DMA ring (8 KiB per 64-channel
// step), each of the four waves multiplies its 16 rows by the 32 classes (2 MFMA tiles x 2 k-halves per step) and ends
// with the 32 logits of a time row in four lanes -> softmax by two butterfly steps, no logits round trip through HBM.
// Summation order over the input channels is the NT kernels' (64-channel steps, two 32-deep MFMAs each).
template <int SLOTS>
__global__ __launch_bounds__(256, 1) void output_softmax_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ probs,
                                                                float* __restrict__ logq, float* __restrict__ logits,
                                                                int batch, int t_out, int t_tiles, int cin, int w_rs,
                                                                int x_row0, int x_rs, long x_bs, int k, int logit_stride,
                                                                long logit_batch_stride, float eps) {
    constexpr int BM = 64, SLOT_BYTES = BM * 128, XPW = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = cin * 2 + 16;  // LDS bytes per weight row
    const int wrows = k < 32 ? k : 32;  // weight rows kept in LDS (the MFMA rows behind them read ring bytes: classes >= k are masked)
    const int ring_off = (wrows * wrow + 127) & ~127;  // the k-half XOR of the fragment addresses needs a 128-byte base
    char* ring = smem + ring_off;
    const int b = blockIdx.x / t_tiles;
    const int t0 = (blockIdx.x - b * t_tiles) * BM;
    const int nsteps = cin / BK;

    const __bf16* xbase = x + (long)b * x_bs + (long)(x_row0 + t0) * x_rs;
    int xoff[XPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
        const int row = (wave * XPW + q) * 8 + (lane >> 3);
        xoff[q] = row * x_rs + (((lane & 7) ^ (lane >> 3)) << 3);
    }
    auto request = [&](int step, int slot) {
#pragma unroll
        for (int q = 0; q < XPW; ++q)
            glds16(xbase + step * BK + xoff[q], ring + slot * SLOT_BYTES + (wave * XPW + q) * 1024);
    };
    const int g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned boff = lds0 + ring_off + (wave * 16 + (lane & 15)) * 128 + ((g ^ (lane & 7)) << 4);
    const unsigned aoff = lds0 + (lane & 15) * wrow + g * 16;  // + tile * 16 * wrow + step * 128 + k-half * 64

    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < SLOTS - 1; ++i)
        if (i < nsteps) request(i, i);
    // ---- the weight matrix: 32 rows of cin bf16.  Rows of whole KiB go by LDS-DMA (one KiB per wave instruction, all
    // in flight at once); other widths through registers, 16 bytes per thread and trip.
    if ((cin & 511) == 0) {
        const int chunks = cin / 512;  // KiB per row
        for (int i = wave; i < wrows * chunks; i += 4) {
            const int r = i / chunks, c = i - r * chunks;
            glds16(w + (long)r * w_rs + c * 512 + lane * 8, smem + r * wrow + c * 1024);
        }
    } else {
        const int per_row = cin / 8;  // 16-byte pieces per row
        for (int i = tid; i < wrows * per_row; i += 256) {
            const int r = i / per_row, c = i - r * per_row;
            *(u32x4*)(smem + r * wrow + c * 16) = *(const u32x4*)(w + (long)r * w_rs + c * 8);
        }
    }
    wait_vmcnt<0>();  // weights (and the first activation tiles): the counted waits below start from zero in flight
    __syncthreads();
    int slot = 0;
    for (int c = 0; c < nsteps; ++c) {
        if (c + SLOTS - 2 < nsteps)
            wait_vmcnt<XPW*(SLOTS - 2)>();  // tile c has landed, the SLOTS-2 younger ones may still be in flight
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // ... everybody's part of it, and everybody is done reading tile c-1
        asm volatile("" ::: "memory");
        if (c + SLOTS - 1 < nsteps) request(c + SLOTS - 1, slot == 0 ? SLOTS - 1 : slot - 1);
        bf16x8 af[4], bfr[2];
        const unsigned a_addr = aoff + c * 128;
        ds_read128<0>(af[0], a_addr);
        ds_read128<64>(af[1], a_addr);
        ds_read128<0>(af[2], a_addr + 16 * wrow);
        ds_read128<64>(af[3], a_addr + 16 * wrow);
        const unsigned b_addr = boff + slot * SLOT_BYTES;
        ds_read128<0>(bfr[0], b_addr);
        ds_read128<0>(bfr[1], b_addr ^ 64);
        wait_frags<0>(af, bfr);
        acc[0] = SL_MFMA16(af[0], bfr[0], acc[0]);
        acc[1] = SL_MFMA16(af[2], bfr[0], acc[1]);
        acc[0] = SL_MFMA16(af[1], bfr[1], acc[0]);
        acc[1] = SL_MFMA16(af[3], bfr[1], acc[1]);
        slot = (slot + 1 == SLOTS) ? 0 : slot + 1;
    }

    output_softmax_finish(acc, bias, probs, logq, logits, b, t0, wave, lane, g, t_out, k, logit_stride, logit_batch_stride, eps);
}

// The same layer with the WEIGHTS IN REGISTERS (round 3).  With the weight matrix in LDS (119 KiB of 160) the ring above is
// five 8 KiB slots -- 32 KiB in flight per CU, one barrier per 64-channel step: 22.7 us for the 67 MB of config 3 = 3 TB/s.
// Here the contraction is split over the four waves of the work-group: wave w owns input channels [w cin/4, (w+1) cin/4) of
// all 64 time rows, its slice of the weights (32 classes x cin/4 bf16 = 32 KiB at cin = 2048) sits in 128 VGPRs as MFMA
// fragments for the whole launch, and the wave streams exactly the activation bytes it consumes itself through its own
// ring of four 8 KiB slots (LDS-DMA, counted vmcnt): no barrier in the loop, 128 KiB in flight per CU.  The four partial
// 64 x 32 logit tiles meet once, in LDS, summed in wave order (deterministic); wave w then finishes rows 16 w .. 16 w + 15.
// NSTEP = cin / 256 (64-channel steps per wave).
template <int NSTEP>
__global__ __launch_bounds__(256, 1) void output_softmax_regw_kernel(const __bf16* __restrict__ x,
                                                                     const __bf16* __restrict__ w,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ probs, float* __restrict__ logq,
                                                                     float* __restrict__ logits, int batch, int t_out,
                                                                     int t_tiles, int w_rs, int x_row0, int x_rs, long x_bs,
                                                                     int k, int logit_stride, long logit_batch_stride,
                                                                     float eps) {
    constexpr int BM = 64, SLOT_BYTES = BM * 128, SLOTS = NSTEP < 4 ? NSTEP : 4, DPS = 8;  // DMA instructions per step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int b = blockIdx.x / t_tiles;
    const int t0 = (blockIdx.x - b * t_tiles) * BM;
    const int ch0 = wave * (NSTEP * 64);  // this wave's channel slice
    char* ring = smem + wave * (SLOTS * SLOT_BYTES);
    const __bf16* xbase = x + (long)b * x_bs + (long)(x_row0 + t0) * x_rs + ch0;
    // DMA: instruction q of a step moves rows 8 q .. 8 q + 7 (lane >> 3 = row, lane & 7 = 16-byte slot holding logical
    // piece (lane & 7) ^ row: the fragment reads below find piece p of row r at slot p ^ (r & 7))
    const int xoff0 = (lane >> 3) * x_rs + (((lane & 7) ^ (lane >> 3)) << 3);
    auto request = [&](int step, int slot) {
#pragma unroll
        for (int q = 0; q < DPS; ++q)
            glds16(xbase + step * 64 + q * 8 * x_rs + xoff0, ring + slot * SLOT_BYTES + q * 1024);
    };
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) request(i, i);
    // weight fragments: class tile j, 32-channel piece q of the slice: lane holds w[16 j + (lane & 15)][ch0 + 32 q + 8 g ..+8]
    bf16x8 wf[2][2 * NSTEP];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cls = j * 16 + (lane & 15);
        const __bf16* wr = w + (long)(cls < k ? cls : 0) * w_rs + ch0 + g * 8;
#pragma unroll
        for (int q = 0; q < 2 * NSTEP; ++q) {
            const u32x4 v = *(const u32x4*)(wr + q * 32);
            const u32x4 z = (u32x4){0u, 0u, 0u, 0u};
            wf[j][q] = __builtin_bit_cast(bf16x8, cls < k ? v : z);
        }
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = (unsigned)(size_t)ring;
    const unsigned boff = lds0 + (lane & 15) * 128 + ((g ^ (lane & 7)) << 4);  // + row tile * 2048, ^ 64 for the second k-half
    // the weight loads are ordinary vector-memory loads in front of the younger DMA requests: vmcnt counts both in order,
    // so waiting for a step's DMA also waits for them
#if defined(SL_PROBE_SM_NOLOOP)  // timing probe (wrong results): prologue + exchange + softmax only
    wait_vmcnt<0>();
#else
#pragma unroll
    for (int c = 0; c < NSTEP; ++c) {
        const int slot = c % SLOTS;
        // requests issued so far: min(NSTEP, c + SLOTS) steps; step c has landed once at most (issued - c - 1) steps remain
        const int after = (c + SLOTS < NSTEP ? c + SLOTS : NSTEP) - c - 1;
        if (after == 3) wait_vmcnt<3 * DPS>();
        else if (after == 2) wait_vmcnt<2 * DPS>();
        else if (after == 1) wait_vmcnt<1 * DPS>();
        else wait_vmcnt<0>();
        bf16x8 bfr[8];
        const unsigned b_addr = boff + slot * SLOT_BYTES;
        ds_read128<0>(bfr[0], b_addr);
        ds_read128<2048>(bfr[1], b_addr);
        ds_read128<4096>(bfr[2], b_addr);
        ds_read128<6144>(bfr[3], b_addr);
        ds_read128<0>(bfr[4], b_addr ^ 64);
        ds_read128<2048>(bfr[5], b_addr ^ 64);
        ds_read128<4096>(bfr[6], b_addr ^ 64);
        ds_read128<6144>(bfr[7], b_addr ^ 64);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3]), "+v"(bfr[4]), "+v"(bfr[5]), "+v"(bfr[6]),
                       "+v"(bfr[7]));
        if (c + SLOTS < NSTEP) request(c + SLOTS, slot);  // the slot's bytes are in registers now
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = SL_MFMA16(wf[j][2 * c + h], bfr[4 * h + i], acc[i][j]);
    }
#endif
    // ---- the four waves' partial tiles meet in LDS (the rings are drained): part[wave][row tile][class tile][lane] f32x4
    __syncthreads();
    f32x4* part = (f32x4*)smem;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) part[((wave * 4 + i) * 2 + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    f32x4 sum[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sum[j] = part[((0 * 4 + wave) * 2 + j) * 64 + lane];
#pragma unroll
        for (int p = 1; p < 4; ++p) sum[j] += part[((p * 4 + wave) * 2 + j) * 64 + lane];
    }
    output_softmax_finish(sum, bias, probs, logq, logits, b, t0, wave, lane, g, t_out, k, logit_stride, logit_batch_stride, eps);
}

// split-K tail: out = epi(sum_split partial), 8 channels per thread, fixed summation order
template <int MODE, bool OUT_F32>
__global__ __launch_bounds__(256) void nt_splitk_epilogue_kernel(NtArgs a, int rows_per_batch) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = a.cout / 8;
    const long total = (long)a.batch * a.t_out * c8;
    if (i >= total) return;
    const int co = (int)(i % c8) * 8;
    const long r = i / c8;
    const int t = (int)(r % a.t_out);
    const int b = (int)(r / a.t_out);
    const long split_stride = (long)a.batch * rows_per_batch * a.cout;
    const float* p = a.partial + ((long)b * rows_per_batch + t) * a.cout + co;
    float v[8];
    {
        const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = v0[j];
            v[4 + j] = v1[j];
        }
    }
    for (int s = 1; s < a.ksplit; ++s) {
        const f32x4 v0 = *(const f32x4*)(p + s * split_stride), v1 = *(const f32x4*)(p + s * split_stride + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += v0[j];
            v[4 + j] += v1[j];
        }
    }
    const long yidx = (long)b * a.y_bs + (long)(a.y_row0 + t) * a.y_rs + co;
    if (MODE == SL_EPI_BIAS || MODE == SL_EPI_BIAS_RELU || MODE == SL_EPI_BIAS_ELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = SL_NT_ACC_SCALE_BIAS(v[j], a.bias[co + j], a);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = SL_NT_ACC_SCALE(v[j], a);
    }
    if (MODE == SL_EPI_BIAS_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    if (MODE == SL_EPI_BIAS_ELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = elu_f(v[j]);
    }
    if (MODE == SL_EPI_ELU_MASK) {
        const u32x4 m = *(const u32x4*)(a.mask + yidx);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] *= elu_grad_from_y(bf16_lo(m[j]));
            v[2 * j + 1] *= elu_grad_from_y(bf16_hi(m[j]));
        }
    }
    if (MODE == SL_EPI_RELU_MASK) {
        const u32x4 m = *(const u32x4*)(a.mask + yidx);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int lo = m[j] & 0xFFFFu, hi = m[j] >> 16;
            if (!(lo != 0 && lo < 0x8000u)) v[2 * j] = 0.f;
            if (!(hi != 0 && hi < 0x8000u)) v[2 * j + 1] = 0.f;
        }
    }
    if (OUT_F32) {
        if (a.out_planes) {  // bf16x3 planes (see store_run16)
            if (a.out_planes == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            if (a.out_planes == 3) {
                const u32x4 m = *(const u32x4*)(a.mask + yidx);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned int lo = m[j] & 0xFFFFu, hi = m[j] >> 16;
                    if (!(lo != 0 && lo < 0x8000u)) v[2 * j] = 0.f;
                    if (!(hi != 0 && hi < 0x8000u)) v[2 * j + 1] = 0.f;
                }
            }
            u32x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = plane_pack_hi(v[2 * j], v[2 * j + 1]);
                l[j] = plane_pack_lo(v[2 * j], v[2 * j + 1], h[j]);
            }
            __bf16* yo = (__bf16*)a.y + yidx;
            *(u32x4*)yo = h;
            *(u32x4*)(yo + a.cout) = l;
            *(u32x4*)(yo + 2 * a.cout) = h;
            return;
        }
        float* yo = (float*)a.y + yidx;
        *(f32x4*)yo = (f32x4){v[0], v[1], v[2], v[3]};
        *(f32x4*)(yo + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
        u32x4 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
        *(u32x4*)((__bf16*)a.y + yidx) = pk;
    }
}

template <bool M32, int IT, int WM, int WN, int STAGES, int MODE, bool OUT_F32>
int launch_main(const NtArgs& a, hipStream_t s) {
    const int grid = xcd_grid(a.batch * a.t_tiles * a.n_tiles * a.ksplit);
    if constexpr (!M32 && IT == 50) {  // 128x128 tile, eight waves in k-half pairs (conv_nt_ks2_bf16_kernel)
        constexpr int LDS_BYTES = (STAGES & 7) * 256 * 128;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)conv_nt_ks2_bf16_kernel<(STAGES & 7), MODE, OUT_F32>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            attr_set = true;
        }
        SL_LAUNCH_MAIN((conv_nt_ks2_bf16_kernel<(STAGES & 7), MODE, OUT_F32>), dim3(grid), dim3(512), LDS_BYTES, s, a);
        return sl_check_launch("sl_conv1d_nt(bf16, k-half pairs)");
    } else if constexpr (!M32 && IT >= 100) {  // slab variant: IT - 100 is the real IT (IT - 200: interleaved schedule)
        constexpr bool ILV = IT >= 200;
        constexpr int RIT = ILV ? IT - 200 : IT - 100;
        constexpr int LDS_BYTES = 2 * (16 * RIT * WM / 8 + 4) * 1024 + (STAGES & 7) * 64 * WN * 128;
        static_assert(LDS_BYTES <= 160 * 1024, "slabs + weight ring exceed the 160 KiB of a CU");
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)conv_nt_slab_bf16_kernel<RIT, WM, WN, STAGES, MODE, OUT_F32, ILV>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            attr_set = true;
        }
        SL_LAUNCH_MAIN((conv_nt_slab_bf16_kernel<RIT, WM, WN, STAGES, MODE, OUT_F32, ILV>), dim3(grid),
                           dim3(64 * WM * WN), LDS_BYTES, s, a);
        return sl_check_launch("sl_conv1d_nt(bf16, slab)");
    } else {
        constexpr int LDS_BYTES = (STAGES & 7) * ((M32 ? 64 : 16 * IT) * WM + 64 * WN) * 128;
        static_assert(LDS_BYTES <= 160 * 1024, "LDS ring exceeds the 160 KiB of a CU");
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)conv_nt_bf16_kernel<M32, IT, WM, WN, STAGES, MODE, OUT_F32>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            attr_set = true;
        }
        SL_LAUNCH_MAIN((conv_nt_bf16_kernel<M32, IT, WM, WN, STAGES, MODE, OUT_F32>), dim3(grid), dim3(64 * WM * WN),
                           LDS_BYTES, s, a);
        return sl_check_launch("sl_conv1d_nt(bf16)");
    }
}

template <int MODE, bool OUT_F32>
int launch_tail(const NtArgs& a, int rows_per_batch, hipStream_t s) {
    const long total = (long)a.batch * a.t_out * (a.cout / 8);
    hipLaunchKernelGGL((nt_splitk_epilogue_kernel<MODE, OUT_F32>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       a, rows_per_batch);
    return sl_check_launch("sl_conv1d_nt(bf16 split-K epilogue)");
}

template <bool M32, int IT, int WM, int WN, int STAGES>
int launch_cfg(NtArgs& a, int epilogue, int out_f32, hipStream_t s) {
    constexpr int BM = (M32 ? 64 : 16 * (IT >= 200 ? IT - 200 : IT >= 100 ? IT - 100 : IT == 50 ? 4 : IT)) * WM;
    a.t_tiles = (a.t_out + BM - 1) / BM;
    a.n_tiles = a.cout / (64 * WN);
    if (a.ksplit > 1) {
        int rc = launch_main<M32, IT, WM, WN, STAGES, MODE_PARTIAL, true>(a, s);
        if (rc != SL_OK) return rc;
        const int rows = a.t_tiles * BM;
        if (out_f32) {
            if (epilogue == SL_EPI_BIAS) return launch_tail<SL_EPI_BIAS, true>(a, rows, s);
            if (epilogue == SL_EPI_NONE) return launch_tail<SL_EPI_NONE, true>(a, rows, s);
        } else {
#if !defined(SL_ELEM_F16)  // (the fp16 build serves the plane path only: fp32 or plane outputs)
            switch (epilogue) {
                case SL_EPI_NONE: return launch_tail<SL_EPI_NONE, false>(a, rows, s);
                case SL_EPI_BIAS: return launch_tail<SL_EPI_BIAS, false>(a, rows, s);
                case SL_EPI_BIAS_RELU: return launch_tail<SL_EPI_BIAS_RELU, false>(a, rows, s);
                case SL_EPI_RELU_MASK: return launch_tail<SL_EPI_RELU_MASK, false>(a, rows, s);
                case SL_EPI_BIAS_ELU: return launch_tail<SL_EPI_BIAS_ELU, false>(a, rows, s);
                case SL_EPI_ELU_MASK: return launch_tail<SL_EPI_ELU_MASK, false>(a, rows, s);
            }
#endif
        }
    } else if (out_f32) {
        if (epilogue == SL_EPI_BIAS) return launch_main<M32, IT, WM, WN, STAGES, SL_EPI_BIAS, true>(a, s);
        if (epilogue == SL_EPI_NONE) return launch_main<M32, IT, WM, WN, STAGES, SL_EPI_NONE, true>(a, s);
    } else {
#if !defined(SL_ELEM_F16)
        switch (epilogue) {
            case SL_EPI_NONE: return launch_main<M32, IT, WM, WN, STAGES, SL_EPI_NONE, false>(a, s);
            case SL_EPI_BIAS: return launch_main<M32, IT, WM, WN, STAGES, SL_EPI_BIAS, false>(a, s);
            case SL_EPI_BIAS_RELU: return launch_main<M32, IT, WM, WN, STAGES, SL_EPI_BIAS_RELU, false>(a, s);
            case SL_EPI_RELU_MASK: return launch_main<M32, IT, WM, WN, STAGES, SL_EPI_RELU_MASK, false>(a, s);
            case SL_EPI_BIAS_ELU: return launch_main<M32, IT, WM, WN, STAGES, SL_EPI_BIAS_ELU, false>(a, s);
            case SL_EPI_ELU_MASK: return launch_main<M32, IT, WM, WN, STAGES, SL_EPI_ELU_MASK, false>(a, s);
        }
#endif
    }
    sl_set_error("sl_conv1d_nt(bf16): unsupported epilogue %d with out_f32=%d", epilogue, out_f32);
    return SL_ERR_UNSUPPORTED;
}

struct Cfg {
    int wm, wn, stages, ksplit, it, m32;
    int gm = 0;  // m-tiles per raster block (0 = automatic)
    int slab = 0;  // chunk-major contraction with the activation slab kept in LDS (conv_nt_slab_bf16_kernel)
    int ilv = 0;   // slab kernel with the hand-interleaved MFMA / LDS-read / DMA-request schedule
    int ks2 = 0;   // 128x128 tile with eight waves, the two waves of a SIMD splitting the k-halves (cfg it = 5)
    int bm() const { return (m32 ? 64 : 16 * it) * wm; }
};

// cfg word: wm | wn << 4 | stages << 8 | ksplit << 12 | it << 20 (it = 0 means 4; 5 = 4 with k-half wave pairs) | m32 << 24 | (1 + log2 gm) << 25 (0 = auto)
// | slab << 29 | interleaved << 30;  0 = choose everything automatically
Cfg decode_cfg(int cfg) {
    Cfg c{cfg & 15, (cfg >> 4) & 15, (cfg >> 8) & 15, (cfg >> 12) & 255, (cfg >> 20) & 15, (cfg >> 24) & 1};
    const int gml = (cfg >> 25) & 15;
    c.gm = gml ? 1 << (gml - 1) : 0;
    c.slab = (cfg >> 29) & 1;
    c.ilv = (cfg >> 30) & 1;
    if (c.it == 0) c.it = 4;
    if (c.m32) c.it = 4;
    if (c.it == 5) {
        c.ks2 = 1;
        c.it = 4;
    }
    return c;
}

Cfg auto_cfg(const sl_conv_geom* g) {
    // Table measured on MI355X with tools/tune_kernels.py (latest copy: profiles/r01j_tune_kernels.json), see HISTORY.md section 3.1.
    const long nsteps = (long)g->taps * (g->cin / BK);
    if (g->cout % 256 == 0) {
        const long tiles256 = (long)g->batch * ((g->t_out + 255) / 256) * (g->cout / 256);
        const bool slab_ok = g->taps >= 2 && g->taps <= 33;
        // 256x256 tile, one work-group per CU.  With taps the slab variant (8 waves of 128x64, register-pipelined,
        // hand-interleaved MFMA / LDS-read / DMA-request stream): big_conv_1 forward 0.327 ms = 1.57 PFLOP/s (0.345
        // un-interleaved, 0.372 for the 16-wave tap-major kernel); 1x1 layers: the same 8-wave tile, tap-major,
        // interleaved (big_conv_2 forward 0.104 ms = 1.24 PFLOP/s, 0.107 for the 16-wave kernel)
        if (tiles256 >= 192) return slab_ok ? Cfg{2, 4, 10, 1, 8, 0, 0, 1, 1} : Cfg{2, 4, 10, 1, 8, 0, 0, 0, 1};
        if (nsteps >= 192) {  // long contraction but few tiles (dgrad of big_conv_1: 64 tiles, K = 65536): split K
            // among the split counts that divide the chunks (whole chunks per split, the slab kernel's requirement) take
            // the one with the least (rounds of 256 work-groups) x (steps per split + 35); ties go to the smaller split
            // (less partial-tile traffic).  B = 48: 96 tiles -> 2 splits (192 work-groups), not 3 (288 = two rounds).
            // The 35 steps stand for a work-group's prologue, its fp32 partial tile and the epilogue pass over it --
            // fitted to config 5's shapes, where the counts differ most: striding_conv on 640 pair channels (240 steps;
            // 48 / 80 / 120 tiles -> 5 / 2 / 2 splits: 0.100 / 0.173 / 0.199 ms against 0.103 / 0.229 / 0.260 ms on 128 x
            // 128 tiles, and 0.154 / 0.185 / 0.263 ms for the other count), big_conv_1's input gradient (1024 steps; 80 /
            // 120 tiles -> 8 / 2 splits: 0.570 / 0.688 ms against 0.603 / 0.792 ms).
            const long chunks = g->cin / BK;
            int best = 1;
            double best_cost = 1e30;
            for (int ks = 1; ks <= 8; ++ks) {
                if (chunks % ks) continue;
                const double rounds = (double)((tiles256 * ks + sl_cus() - 1) / sl_cus());
                const double cost = rounds * ((double)nsteps / ks + 35.0);
                if (cost < best_cost - 1e-9) {
                    best_cost = cost;
                    best = ks;
                }
            }
            if (slab_ok) return Cfg{2, 4, 10, best, 8, 0, 0, 1, 1};
            return Cfg{4, 4, 2, best, 4, 0};
        }
    }
    // short layers (one 128x128 tile per CU at most): 4 waves of 64x64, 3-slot ring, register-pipelined and
    // hand-interleaved: 18.4 us per 250-channel layer back to back (8 waves of 32x64 un-interleaved 18.8, plain loop 20.7,
    // eight waves in k-half pairs 18.4, interleaved slab 19.5-20.4: profiles/r01j_tune_kernels.json, r01j_nt_scaling.json);
    // striding_conv 48.5 us (54.0)
    Cfg c{2, 2, 11, 1, 4, 0, 0, 0, 1};
#if !defined(SL_NO_WIDE_ROW_SLAB)
    // rows of >= 512 input channels under several taps (the inner layers on the bf16x3 path: 7 taps x 768 plane channels, 84
    // steps): the interleaved SLAB variant of the same tile -- a chunk's activation rows brought into LDS once for all taps --
    // 50.1 against 55.0 us back to back (54.5 for the k-half pairs below; at 256 channels the order is the other way round:
    // 19.8 against 18.5); tools/nt_wide_cost.py --taps 7 --cout 256 --cins 256,768, round 5
    if (g->taps >= 3 && g->taps <= 33 && g->cin >= 512 && nsteps >= 64) return Cfg{2, 2, 11, 1, 4, 0, 0, 1, 1};
#endif
    // long contractions on this tile (striding_conv: 96 steps): the eight-wave k-half-pair variant's shorter step
    // outweighs its dearer epilogue (49.5 vs 51.7 us right behind the producer of the input)
    if (nsteps >= 64 && g->cout % 128 == 0) c.ks2 = 1;
    return c;
}

bool valid_cfg(const Cfg& full, const sl_conv_geom* g) {
    Cfg c = full;
    c.stages = full.stages & 7;  // bit 3 selects the register-pipelined loop, instantiated for the shapes listed below
    if (c.ks2) {  // instantiated: 2x2 patches, 3 or 4 slots, interleaved tap-major only
        if (c.m32 || c.slab || !c.ilv || c.wm != 2 || c.wn != 2 || (full.stages != 11 && full.stages != 12)) return false;
        if (g->cout % 128 || c.ksplit < 1) return false;
        return c.ksplit <= (long)g->taps * (g->cin / BK);
    }
    if ((full.stages & 8) && (c.m32 || !((c.it == 4 && c.wm * c.wn >= 4 && c.wm * c.wn <= 8) || c.it == 8 ||
                                         (c.it == 2 && c.wm == 4 && c.wn == 2))))
        return false;
    bool shape;
    if (c.m32)
        shape = (c.wm == 2 && c.wn == 2 && c.stages >= 2 && c.stages <= 4) || (c.wm == 4 && c.wn == 4 && c.stages == 2) ||
                (c.wm == 4 && c.wn == 2 && (c.stages == 2 || c.stages == 3)) ||
                (c.wm == 2 && c.wn == 4 && (c.stages == 2 || c.stages == 3));
    else
        shape = (c.it == 2 && c.wm == 2 && c.wn == 2 && c.stages >= 2 && c.stages <= 4) ||
                (c.it == 2 && c.wm == 2 && c.wn == 4 && (c.stages == 2 || c.stages == 3)) ||
                (c.it == 2 && c.wm == 4 && c.wn == 2 && c.stages >= 2 && c.stages <= 4) ||
                (c.it == 2 && c.wm == 8 && c.wn == 2 && (c.stages == 2 || c.stages == 3)) ||
                (c.it == 4 && c.wm == 2 && c.wn == 2 && (c.stages >= 2 && c.stages <= 4)) ||
                (c.it == 4 && c.wm == 4 && c.wn == 2 && (c.stages == 2 || c.stages == 3)) ||
                (c.it == 4 && c.wm == 2 && c.wn == 4 && (c.stages == 2 || c.stages == 3)) ||
                (c.it == 4 && c.wm == 4 && c.wn == 4 && c.stages == 2) ||
                (c.it == 8 && c.wm == 2 && c.wn == 4 && c.stages == 2) ||  // 256x256 tile, 8 waves of 128x64
                (c.it == 8 && c.wm == 2 && c.wn == 2 && (full.stages == 10 || full.stages == 11));  // 256x128, pipelined only
    if (!shape || c.ksplit < 1) return false;
    if (g->cout % (64 * c.wn)) return false;
    const long nsteps = (long)g->taps * (g->cin / BK);
    if (c.ilv) {  // instantiated interleaved shapes: the slab 256x256 kernel and three tap-major pipelined tiles
        const bool slab_ilv = c.slab && !c.m32 && ((c.it == 8 && c.wm == 2 && c.wn == 4 && full.stages == 10) ||
                                                   (c.it == 4 && c.wm == 2 && c.wn == 2 && full.stages >= 10 && full.stages <= 12));
        const bool tap_ilv = !c.slab && !c.m32 && (full.stages & 8) &&
                             ((c.it == 2 && c.wm == 4 && c.wn == 2) || (c.it == 4 && c.wm == 2 && c.wn == 2) ||
                              (c.it == 8 && c.wm == 2 && c.wn == 4 && full.stages == 10));
        if (!slab_ilv && !tap_ilv) return false;
    }
    if (c.slab) {
        // instantiated slab shapes; the slab holds BM + 32 rows and its DMA must be older than the next chunk's tiles
        const bool inst = !c.m32 && ((c.it == 4 && c.wm == 4 && c.wn == 4 && full.stages == 2) ||
                                     (c.it == 8 && c.wm == 2 && c.wn == 4 && full.stages == 10) ||
                                     (c.it == 2 && c.wm == 4 && c.wn == 2 && (full.stages == 11 || full.stages == 3)) ||
                                     (c.it == 4 && c.wm == 2 && c.wn == 2 && (full.stages == 11 || full.stages == 12)) ||
                                     (c.ilv && c.it == 4 && c.wm == 2 && c.wn == 2 && full.stages == 10));
        if (!inst || g->taps > 33 || g->taps < c.stages) return false;
        return c.ksplit <= g->cin / BK;  // whole chunks per split
    }
    return c.ksplit <= nsteps;
}

}  // namespace

#if defined(SL_ELEM_F16)  // the second translation unit of this file: same kernels on v_mfma_*_f16, exported under these names
#define conv_nt_bf16_workspace_bytes conv_nt_f16_workspace_bytes
#define conv_nt_bf16 conv_nt_f16
#endif

size_t conv_nt_bf16_workspace_bytes(const sl_conv_geom* g, int cfg) {
    Cfg c = cfg ? decode_cfg(cfg) : auto_cfg(g);
    if (!valid_cfg(c, g) || c.ksplit <= 1) return 0;
    const long rows = (long)((g->t_out + c.bm() - 1) / c.bm()) * c.bm();
    return (size_t)c.ksplit * g->batch * rows * g->cout * sizeof(float);
}

int conv_nt_bf16(const void* x, const void* w, const float* bias, const void* mask, void* y, const sl_conv_geom* g,
                 int epilogue, int out_f32, int cfg, void* workspace, size_t workspace_bytes, hipStream_t s) {
    Cfg c = cfg ? decode_cfg(cfg) : auto_cfg(g);
    if (!valid_cfg(c, g)) {
        sl_set_error(
            "sl_conv1d_nt(bf16): invalid tile configuration m32=%d it=%d wm=%d wn=%d stages=%d ksplit=%d for cout=%d",
            c.m32, c.it, c.wm, c.wn, c.stages, c.ksplit, g->cout);
        return SL_ERR_INVALID_ARGUMENT;
    }
    NtArgs a;
    a.out_planes = 0;
    a.acc_scale = g->acc_scale != 0.f ? g->acc_scale : 1.f;
    if (out_f32 == 2) {  // bf16x3 planes out of the fp32-output kernels: the activation moves from the template to a.out_planes
        if (epilogue == SL_EPI_BIAS_RELU) {
            a.out_planes = 1;
            epilogue = SL_EPI_BIAS;
        } else if (epilogue == SL_EPI_BIAS || epilogue == SL_EPI_NONE) {
            a.out_planes = 2;
        } else if (epilogue == SL_EPI_RELU_MASK) {
            a.out_planes = 3;
            epilogue = SL_EPI_NONE;
        } else {
            sl_set_error("sl_conv1d_nt(bf16): out_f32 = 2 (bf16x3 planes) goes with the epilogues NONE, BIAS, BIAS_RELU, RELU_MASK");
            return SL_ERR_UNSUPPORTED;
        }
        out_f32 = 1;
    }
    a.x = (const __bf16*)x;
    a.w = (const __bf16*)w;
    a.bias = bias;
    a.mask = (const __bf16*)mask;
    a.y = y;
    a.partial = (float*)workspace;
    a.batch = g->batch;
    a.t_out = g->t_out;
    a.x_row0 = g->x_row0;
    a.x_rs = g->x_row_stride;
    a.x_bs = g->x_batch_stride;
    a.y_row0 = g->y_row0;
    a.y_rs = g->y_row_stride;
    a.y_bs = g->y_batch_stride;
    a.cout = g->cout;
    a.w_rs = g->taps * g->cin;
    a.taps = g->taps;
    a.cin = g->cin;
    a.nsteps = g->taps * (g->cin / BK);
    a.ksplit = c.ksplit;
    {
        const int bm = c.bm();
        const int m_tiles = g->batch * ((g->t_out + bm - 1) / bm);
        const int n_tiles = g->cout / (64 * c.wn);
        int gm = c.gm;
        if (gm == 0) gm = n_tiles >= 4 ? 16 : m_tiles;
        a.gm = gm < m_tiles ? gm : m_tiles;
    }
    a.steps_per_split = (a.nsteps + c.ksplit - 1) / c.ksplit;
    if (c.slab) a.steps_per_split = ((g->cin / BK + c.ksplit - 1) / c.ksplit) * g->taps;  // whole chunks per split
    a.ksplit = (a.nsteps + a.steps_per_split - 1) / a.steps_per_split;  // no empty splits
    if (a.ksplit > 1) {
        const size_t need = conv_nt_bf16_workspace_bytes(g, cfg);
        if (workspace == nullptr || workspace_bytes < need) {
            sl_set_error("sl_conv1d_nt(bf16): split-K workspace too small (%zu < %zu)", workspace_bytes, need);
            return SL_ERR_WORKSPACE_TOO_SMALL;
        }
    }
    if (c.ks2 && c.stages == 11) return launch_cfg<false, 50, 2, 2, 11>(a, epilogue, out_f32, s);
    if (c.ks2 && c.stages == 12) return launch_cfg<false, 50, 2, 2, 12>(a, epilogue, out_f32, s);
#define SL_NT_SLAB_CASE(IT_, WM_, WN_, ST_)                                      \
    if (c.slab && c.it == IT_ && c.wm == WM_ && c.wn == WN_ && c.stages == ST_) \
        return launch_cfg<false, 100 + IT_, WM_, WN_, ST_>(a, epilogue, out_f32, s);
    if (c.slab && c.ilv && c.it == 8) return launch_cfg<false, 208, 2, 4, 10>(a, epilogue, out_f32, s);
    if (c.slab && c.ilv && c.stages == 10) return launch_cfg<false, 204, 2, 2, 10>(a, epilogue, out_f32, s);
    if (c.slab && c.ilv && c.stages == 11) return launch_cfg<false, 204, 2, 2, 11>(a, epilogue, out_f32, s);
    if (c.slab && c.ilv && c.stages == 12) return launch_cfg<false, 204, 2, 2, 12>(a, epilogue, out_f32, s);
    SL_NT_SLAB_CASE(4, 4, 4, 2)
    SL_NT_SLAB_CASE(8, 2, 4, 10)
    SL_NT_SLAB_CASE(2, 4, 2, 11)
    SL_NT_SLAB_CASE(2, 4, 2, 3)
    SL_NT_SLAB_CASE(4, 2, 2, 11)
    SL_NT_SLAB_CASE(4, 2, 2, 12)
#undef SL_NT_SLAB_CASE
    if (c.slab) {
        sl_set_error("sl_conv1d_nt(bf16): slab configuration not instantiated");
        return SL_ERR_UNSUPPORTED;
    }
#define SL_NT_ILV_CASE(IT_, WM_, WN_, ST_)                                          \
    if (!c.slab && c.ilv && c.it == IT_ && c.wm == WM_ && c.wn == WN_ && c.stages == ST_) \
        return launch_cfg<false, IT_, WM_, WN_, (ST_ | 16)>(a, epilogue, out_f32, s);
    SL_NT_ILV_CASE(2, 4, 2, 10)
    SL_NT_ILV_CASE(2, 4, 2, 11)
    SL_NT_ILV_CASE(2, 4, 2, 12)
    SL_NT_ILV_CASE(4, 2, 2, 10)
    SL_NT_ILV_CASE(4, 2, 2, 11)
    SL_NT_ILV_CASE(4, 2, 2, 12)
    SL_NT_ILV_CASE(8, 2, 4, 10)
#undef SL_NT_ILV_CASE
#define SL_NT_CASE(M32_, IT_, WM_, WN_, ST_)                                                \
    if (c.m32 == M32_ && c.it == IT_ && c.wm == WM_ && c.wn == WN_ && c.stages == ST_) \
        return launch_cfg<(M32_ != 0), IT_, WM_, WN_, ST_>(a, epilogue, out_f32, s);
    SL_NT_CASE(0, 4, 2, 2, 10)
    SL_NT_CASE(0, 4, 2, 2, 11)
    SL_NT_CASE(0, 4, 2, 2, 12)
    SL_NT_CASE(0, 2, 4, 2, 10)
    SL_NT_CASE(0, 2, 4, 2, 11)
    SL_NT_CASE(0, 2, 4, 2, 12)
    SL_NT_CASE(0, 4, 4, 2, 10)
    SL_NT_CASE(0, 4, 4, 2, 11)
    SL_NT_CASE(0, 4, 2, 4, 10)
    SL_NT_CASE(0, 4, 2, 4, 11)
    SL_NT_CASE(0, 8, 2, 4, 10)
    SL_NT_CASE(0, 8, 2, 4, 2)
    SL_NT_CASE(0, 8, 2, 2, 10)
    SL_NT_CASE(0, 8, 2, 2, 11)
    SL_NT_CASE(0, 4, 2, 2, 2)
    SL_NT_CASE(0, 4, 2, 2, 3)
    SL_NT_CASE(0, 4, 2, 2, 4)
    SL_NT_CASE(0, 4, 4, 2, 2)
    SL_NT_CASE(0, 4, 4, 2, 3)
    SL_NT_CASE(0, 4, 2, 4, 2)
    SL_NT_CASE(0, 4, 2, 4, 3)
    SL_NT_CASE(0, 4, 4, 4, 2)
    SL_NT_CASE(0, 2, 2, 2, 2)
    SL_NT_CASE(0, 2, 2, 2, 3)
    SL_NT_CASE(0, 2, 2, 2, 4)
    SL_NT_CASE(0, 2, 2, 4, 2)
    SL_NT_CASE(0, 2, 2, 4, 3)
    SL_NT_CASE(0, 2, 4, 2, 2)
    SL_NT_CASE(0, 2, 4, 2, 3)
    SL_NT_CASE(0, 2, 4, 2, 4)
    SL_NT_CASE(0, 2, 8, 2, 2)
    SL_NT_CASE(0, 2, 8, 2, 3)
    SL_NT_CASE(1, 4, 2, 2, 2)
    SL_NT_CASE(1, 4, 2, 2, 3)
    SL_NT_CASE(1, 4, 2, 2, 4)
    SL_NT_CASE(1, 4, 4, 2, 2)
    SL_NT_CASE(1, 4, 4, 2, 3)
    SL_NT_CASE(1, 4, 2, 4, 2)
    SL_NT_CASE(1, 4, 2, 4, 3)
    SL_NT_CASE(1, 4, 4, 4, 2)
#undef SL_NT_CASE
    sl_set_error("sl_conv1d_nt(bf16): configuration not instantiated");
    return SL_ERR_UNSUPPORTED;
}

#if !defined(SL_ELEM_F16)  // (the fused output layer belongs to the bf16 path: one definition, in the bf16 translation unit)
// ---- fused output layer (declared in capi.hip's dispatch: sl_output_softmax)
bool output_softmax_supported(const sl_conv_geom* g, int k) {
    if (g->taps != 1 || k < 1 || k > 32 || g->cout < 32 || g->cin % BK) return false;
    return ((k * (g->cin * 2 + 16) + 127) & ~127) + 3 * 64 * 128 <= 160 * 1024;
}

static int g_output_softmax_variant = 0;  // sl_output_softmax_select: 0 = automatic, 1 = weights in LDS, 2 = weights in registers
int output_softmax_select(int variant) {
    g_output_softmax_variant = variant;
    return SL_OK;
}

template <int NSTEP>
static int launch_output_softmax_regw(const void* x, const void* w, const float* bias, float* probs, float* logq,
                                      float* logits, const sl_conv_geom* g, int k, int logit_stride,
                                      long logit_batch_stride, float eps, hipStream_t s) {
    constexpr int SLOTS = NSTEP < 4 ? NSTEP : 4;
    constexpr int LDS = (4 * SLOTS * 64 * 128) > 32768 ? (4 * SLOTS * 64 * 128) : 32768;  // rings; at least the 32 KiB exchange
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)output_softmax_regw_kernel<NSTEP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int t_tiles = (g->t_out + 63) / 64;
    hipLaunchKernelGGL(output_softmax_regw_kernel<NSTEP>, dim3(g->batch * t_tiles), dim3(256), LDS, s, (const __bf16*)x,
                       (const __bf16*)w, bias, probs, logq, logits, g->batch, g->t_out, t_tiles, g->taps * g->cin, g->x_row0,
                       g->x_row_stride, (long)g->x_batch_stride, k, logit_stride, logit_batch_stride, eps);
    return sl_check_launch("sl_output_softmax(register weights)");
}

int output_softmax_bf16(const void* x, const void* w, const float* bias, float* probs, float* logq, float* logits,
                        const sl_conv_geom* g, int k, int logit_stride, long logit_batch_stride, float eps, hipStream_t s) {
    // weights in registers (one wave per quarter of the input channels, its own ring): cin = 256, 512, 1024 or 2048
    if (g_output_softmax_variant != 1 && g->taps == 1) {
        if (g->cin == 2048) return launch_output_softmax_regw<8>(x, w, bias, probs, logq, logits, g, k, logit_stride, logit_batch_stride, eps, s);
        if (g->cin == 1024) return launch_output_softmax_regw<4>(x, w, bias, probs, logq, logits, g, k, logit_stride, logit_batch_stride, eps, s);
        if (g->cin == 512) return launch_output_softmax_regw<2>(x, w, bias, probs, logq, logits, g, k, logit_stride, logit_batch_stride, eps, s);
        if (g->cin == 256) return launch_output_softmax_regw<1>(x, w, bias, probs, logq, logits, g, k, logit_stride, logit_batch_stride, eps, s);
    }
    const int wbytes = (k * (g->cin * 2 + 16) + 127) & ~127;
    const int slots = wbytes + 5 * 64 * 128 <= 160 * 1024 ? 5 : 3;  // ring depth the rest of the LDS allows
    const int lds = wbytes + slots * 64 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)output_softmax_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)output_softmax_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int t_tiles = (g->t_out + 63) / 64;
    if (slots == 5)
        hipLaunchKernelGGL(output_softmax_kernel<5>, dim3(g->batch * t_tiles), dim3(256), lds, s, (const __bf16*)x,
                           (const __bf16*)w, bias, probs, logq, logits, g->batch, g->t_out, t_tiles, g->cin,
                           g->taps * g->cin, g->x_row0, g->x_row_stride, (long)g->x_batch_stride, k, logit_stride,
                           logit_batch_stride, eps);
    else
        hipLaunchKernelGGL(output_softmax_kernel<3>, dim3(g->batch * t_tiles), dim3(256), lds, s, (const __bf16*)x,
                           (const __bf16*)w, bias, probs, logq, logits, g->batch, g->t_out, t_tiles, g->cin,
                           g->taps * g->cin, g->x_row0, g->x_row_stride, (long)g->x_batch_stride, k, logit_stride,
                           logit_batch_stride, eps);
    return sl_check_launch("sl_output_softmax");
}
#endif  // !SL_ELEM_F16
