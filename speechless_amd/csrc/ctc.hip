// ctc.hip -- output softmax, Keras/TF CTC loss + gradient, greedy decode for gfx950.
//
// Reference call sites (speechless/net.py): output softmax :131,:328-330; _ctc_lambda -> keras ctc_batch_cost ->
// tf.nn.ctc_loss :402-406 (the op sees log(p + 1e-8) and applies its own softmax, so the effective distribution is
// q = (p + eps) / sum(p + eps)); greedy decode :417-436,:452-454 and grapheme_enconding.py:34-57.
//
// Kernels:
//   softmax_logq_kernel : one wave per (b,t) frame, lane = class: p = softmax(logits), logq = log q.   (HBM-bound, tiny)
//   ctc_lattice_kernel  : grid (B, 2).  Work-group (b,0) runs the alpha recursion t = 0..T-1, work-group (b,1) the
//                         beta recursion t = T-1..0, concurrently.  One thread per extended-label state
//                         (S = 2L+1 <= 1024), log-space fp32, previous row double-buffered in LDS (one barrier per
//                         frame), emission log-probs prefetched 8 frames ahead into registers.  The recursion is
//                         sequential in t -> latency-bound by construction; running alpha and beta in separate
//                         work-groups halves the critical path.
//   ctc_lattice_wave_kernel (default for labels of up to 255 graphemes): the same two recursions in the probability domain,
//                         ONE WAVE per utterance and direction, doubles with an exponent per lane (see further down).
//   ctc_grad_kernel     : one wave per frame.  occupancy gamma_t(k) = sum_{s: l'_s = k} exp(alpha+beta-logq-logP) is
//                         reduced in a FIXED order (blank: wave butterfly; graphemes: per-class position lists built
//                         once per work-group) -> deterministic.  Chains through log(p+eps) and the output softmax and
//                         writes d(loss)/d(logits) straight into the halo'd gradient tensor of output_conv.
//   greedy_decode_kernel: one work-group per utterance: argmax (first max wins), merge repeats, drop blank, compact.
#include "common.h"

#include <type_traits>

namespace {

// The lattice lives in LOG2 units: v_exp_f32 / v_log_f32 are base-2 natively, so a 3-way log-sum-exp is 3 + 1 raw
// transcendentals instead of 4 range-reduced libm calls on the 500-step sequential critical path.  The arguments are
// <= 0 (exp2) and in [1,3] (log2), so the raw instructions need no denormal / range fix-ups.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ float lse3_2(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) +
                                     __builtin_amdgcn_exp2f(c - m));
}

// the same in doubles (repair pass, round 6: log values of magnitude 2^12 carry 2.4e-4 in fp32 -- a repaired utterance came
// back at 6e-4 ... 2e-3 of the float64 oracle; in doubles at 1e-6.  Rare path: the library exp2 / log2 cost does not matter)
__device__ __forceinline__ double lse3_2d(double a, double b, double c) {
    const double m = fmax(a, fmax(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + log2(exp2(a - m) + exp2(b - m) + exp2(c - m));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// one wave per frame, lane = class (k <= 64): a frame's logits are one coalesced 4*k-byte read; the four reductions
// are wave butterflies.  (One THREAD per frame with strided rows took 31 us for 16000 frames; this takes ~4.)
__global__ __launch_bounds__(256) void softmax_logq_kernel(const float* __restrict__ logits, float* __restrict__ probs,
                                                           float* __restrict__ logq, long frames, int t_out, int k,
                                                           int logit_stride, long logit_batch_stride, float eps) {
    const long f = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (f >= frames) return;
    const int lane = threadIdx.x & 63;
    const long b = f / t_out;
    const bool on = lane < k;
    const float z = on ? logits[b * logit_batch_stride + (f - b * t_out) * logit_stride + lane] : -INFINITY;
    const float m = wave_max(z);
    const float e = on ? expf(z - m) : 0.f;
    const float sum = wave_sum(e);
    const float pi = e / sum;
    // q = (p + eps) / sum_j (p_j + eps); computed the way TF does: log-softmax of u = log(p + eps)
    const float u = on ? logf(pi + eps) : -INFINITY;
    const float um = wave_max(u);
    const float usum = wave_sum(on ? expf(u - um) : 0.f);
    const float lz = um + logf(usum);
    if (on) {
        probs[f * k + lane] = pi;
        logq[f * k + lane] = u - lz;
    }
}

// workspace layout: alpha [B][T][SP], beta [B][T][SP]  (SP = S rounded up to 64)
__global__ __launch_bounds__(1024) void ctc_lattice_kernel(const float* __restrict__ logq,
                                                           const int32_t* __restrict__ labels,
                                                           const int32_t* __restrict__ label_len,
                                                           const int32_t* __restrict__ input_len,
                                                           float* __restrict__ alpha, float* __restrict__ beta,
                                                           float* __restrict__ loss, int32_t* __restrict__ cls,
                                                           int t_out, int k, int l_max, int sp, int blank) {
    extern __shared__ float rowbuf[];  // 2 x (blockDim + 4)   (list builder: l_max + k + 1 ints)
    const int b = blockIdx.x;
    const int dir = blockIdx.y;
    const int s = threadIdx.x;
    const int L = label_len[b];
    if (dir == 2) {
        // Third role, off the recursion's critical path: per-class position lists of this utterance's label for
        // ctc_grad_kernel (counting sort; inside a class the positions stay in label order, which fixes the summation
        // order of the per-class occupancies -> deterministic).  cls[b] = pos[l_max] | start[k + 1].
        int* s_lab = (int*)rowbuf;
        int* s_start = s_lab + l_max;
        int32_t* pos_out = cls + (long)b * (l_max + k + 1);
        int32_t* start_out = pos_out + l_max;
        for (int i = s; i < L; i += blockDim.x) s_lab[i] = labels[(long)b * l_max + i];
        for (int i = s; i <= k; i += blockDim.x) s_start[i] = 0;
        __syncthreads();
        // rank of position i inside its class = number of earlier positions with the same grapheme
        int ranks[16];  // blockDim >= 64, l_max <= 511 -> at most 8 positions per thread
        int nmine = 0;
        for (int i = s; i < L; i += blockDim.x) {
            const int c = s_lab[i];
            int r = 0;
            for (int j = 0; j < i; ++j) r += (s_lab[j] == c);
            ranks[nmine++] = r;
            atomicAdd(&s_start[c + 1], 1);  // integer count: order-independent
        }
        __syncthreads();
        if (s == 0)
            for (int c = 0; c < k; ++c) s_start[c + 1] += s_start[c];
        __syncthreads();
        nmine = 0;
        for (int i = s; i < L; i += blockDim.x) pos_out[s_start[s_lab[i]] + ranks[nmine++]] = i;
        for (int i = s; i <= k; i += blockDim.x) start_out[i] = s_start[i];
        return;
    }
    const int S = 2 * L + 1;
    int T = input_len[b];
    if (T > t_out) T = t_out;
    const int stride = blockDim.x + 4;
    float* buf0 = rowbuf;
    float* buf1 = rowbuf + stride;
    const int32_t* lab = labels + (long)b * l_max;

    // extended label of this state and whether the skip transition INTO (alpha) / OUT OF (beta) it is allowed
    const bool live = s < S;
    int my = blank;
    bool skip = false;
    if (live && (s & 1)) {
        my = lab[s >> 1];
        if (dir == 0) {
            skip = (s >= 3) && (lab[(s >> 1) - 1] != my);
        } else {
            skip = (s + 2 < S) && (lab[(s >> 1) + 1] != my);
        }
    }
    if (T <= 0) {
        if (dir == 0 && s == 0) loss[b] = INFINITY;
        return;
    }
    float* out = (dir == 0 ? alpha : beta) + (long)b * t_out * sp;
    const float* lq = logq + (long)b * t_out * k + my;

    // pads: alpha reads s-1, s-2 (index s+1, s in a buffer shifted by 2); beta reads s+1, s+2
    if (s < 2) {
        buf0[s] = -INFINITY;
        buf1[s] = -INFINITY;
        buf0[blockDim.x + 2 + s] = -INFINITY;
        buf1[blockDim.x + 2 + s] = -INFINITY;
    }
    const int tstart = dir == 0 ? 0 : T - 1;
    const int tstep = dir == 0 ? 1 : -1;

    // Emissions: RAW log q values are fetched one 8-frame chunk AHEAD, unconditionally (every lane has a valid address:
    // dead lanes point at the blank column, steps past the end are clamped), and only scaled to log2 units when consumed.
    // Lessons from the ISA (hipcc 7.2): (1) any arithmetic or exec-masked branch at load time puts s_waitcnt vmcnt(0)
    // right behind the load; (2) a value loaded in one loop iteration and consumed in the next makes the waitcnt pass
    // fall back to vmcnt(1) (it loses the count across the back-edge); (3) an exec-masked store between load and use
    // forces vmcnt(0).  So: loads at the top of the iteration, all 8 frames' stores unconditional (blockDim == sp, dead
    // lanes own a padding column), and the chunk is pinned (empty asm) at the BOTTOM of the same iteration, where the
    // pass can count the 8 younger stores exactly -> the HBM/L2 latency hides under 8 frames of recursion.
    auto load_e = [&](int step) {
        const int st = step < T ? step : T - 1;
        return lq[(long)(tstart + tstep * st) * k];
    };
    float ec[8], en[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ec[j] = load_e(j);
    float* prev = buf0;
    float* cur = buf1;
    const int nb = dir == 0 ? -1 : 1;  // neighbour direction
    for (int base = 0; base < T; base += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) en[j] = load_e(base + 8 + j);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int step = base + j;  // distance from tstart
            if (step < T) {
                const int t = tstart + tstep * step;
                float v;
                if (step == 0) {
                    const bool init = dir == 0 ? (s <= 1) : (s >= S - 2);
                    v = (live && init) ? ec[0] * LOG2E : -INFINITY;
                } else {
                    // LDS-only barrier: __syncthreads() would also drain vmcnt(0) every frame
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    const float a0 = prev[s + 2];
                    const float a1 = prev[s + 2 + nb];
                    const float a2 = skip ? prev[s + 2 + 2 * nb] : -INFINITY;
                    v = fmaf(ec[j], LOG2E, lse3_2(a0, a1, a2));
                    if (!live) v = -INFINITY;
                }
                cur[s + 2] = v;
                out[(long)t * sp + s] = v;
                float* tmp = prev;
                prev = cur;
                cur = tmp;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("" : "+v"(en[j]));  // materialise the prefetched chunk here, inside the iteration
            ec[j] = en[j];
        }
    }
    if (dir == 0) {
        __syncthreads();
        if (s == 0) {
            const float last = prev[S - 1 + 2];
            const float last2 = S >= 2 ? prev[S - 2 + 2] : -INFINITY;
            const float m = fmaxf(last, last2);
            const float lp2 = (m == -INFINITY) ? -INFINITY : m + log2f(exp2f(last - m) + exp2f(last2 - m));
            loss[b] = -lp2 * LN2;  // back to natural-log units: -log p(label | x)
        }
    }
}

// log2 of a non-negative double, as float (the gradient kernel's lattice units); 0 -> -inf
__device__ __forceinline__ float log2_of_double(double x) {
    if (x == 0.0) return -INFINITY;
    int e;
    const double m = frexp(x, &e);
    return (float)e + __builtin_amdgcn_logf((float)m);
}

// one wave per frame; work-group = 4 waves x FRAMES_PER_WAVE frames.  NJ = SP / 64 lattice columns per lane: all of a
// frame's alpha / beta loads are issued before the first one is used (one HBM/L2 round trip per frame instead of NJ).
// LIN: the lattice comes from ctc_lattice_wave_kernel (doubles in linear units with one exponent per frame, emissions
// u = p + eps instead of q) and is brought to log2 units on the fly; the sum of a frame's state posteriors must then be
// ---- repair of one utterance inside the gradient kernel (see ctc_grad_kernel) ------------------------------------------
// Log-domain alpha and beta lattices of utterance b by ONE work-group of 256 threads (S <= 512): the
// recursion of ctc_lattice_kernel without its tuning, IN DOUBLES since round 6 (the fp32 log values of a 500-frame utterance
// carry 2.4e-4 each; a repaired utterance is now as accurate as an unrepaired one) -- this runs only for an utterance whose
// linear lattice lost mass, and then on one work-group while the rest of the chip goes on.
// Both directions at once: threads 0..127 run alpha, threads 128..255 beta (up to four states per thread, one barrier per
// frame for both).
__device__ void repair_lattices(const float* __restrict__ lq_b, const int* s_lab, int L, int S, int T, int k, int blank,
                                int sp, double* __restrict__ out_alpha, double* __restrict__ out_beta, double* rows,
                                float* loss_b, float* logz2_b, int32_t* zint_b) {
    constexpr int RS = 512 + 4;  // row stride in LDS: index s + 2, two pads either side; rows: [direction][2][RS]
    const int tid = threadIdx.x;
    const int dir = tid >> 7;    // wave-uniform
    const int ht = tid & 127;
    int my[4];
    bool skip[4], live[4], mine[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int st = ht + 128 * i;
        mine[i] = st < sp;
        live[i] = st < S;
        my[i] = blank;
        skip[i] = false;
        if (live[i] && (st & 1)) {
            my[i] = s_lab[st >> 1];
            if (dir == 0)
                skip[i] = (st >= 3) && (s_lab[(st >> 1) - 1] != my[i]);
            else
                skip[i] = (st + 2 < S) && (s_lab[(st >> 1) + 1] != my[i]);
        }
    }
    for (int i = tid; i < 4 * RS; i += 256) rows[i] = -(double)INFINITY;
    __syncthreads();
    double* prev = rows + dir * 2 * RS;
    double* cur = prev + RS;
    double* out = dir == 0 ? out_alpha : out_beta;
    const int tstart = dir == 0 ? 0 : T - 1;
    const int tstep = dir == 0 ? 1 : -1;
    const int nb = dir == 0 ? -1 : 1;
    float e[4], en[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = lq_b[(long)tstart * k + my[i]];
    for (int step = 0; step < T; ++step) {
        const int t = tstart + tstep * step;
        const int tn = tstart + tstep * (step + 1 < T ? step + 1 : step);
#pragma unroll
        for (int i = 0; i < 4; ++i) en[i] = lq_b[(long)tn * k + my[i]];  // next frame's emission, a frame ahead
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int st = ht + 128 * i;
            if (mine[i]) {
                double v;
                if (step == 0) {
                    const bool init = dir == 0 ? (st <= 1) : (st >= S - 2);
                    v = (live[i] && init) ? (double)e[i] * 1.4426950408889634 : -(double)INFINITY;
                } else {
                    const double a0 = prev[st + 2];
                    const double a1 = prev[st + 2 + nb];
                    const double a2 = skip[i] ? prev[st + 2 + 2 * nb] : -(double)INFINITY;
                    v = fma((double)e[i], 1.4426950408889634, lse3_2d(a0, a1, a2));
                    if (!live[i]) v = -(double)INFINITY;
                }
                cur[st + 2] = v;
                out[(long)t * sp + st] = v;
            }
            e[i] = en[i];
        }
        __syncthreads();
        double* tmp = prev;
        prev = cur;
        cur = tmp;
    }
    if (tid == 0) {  // (dir 0: prev is alpha's last row)
        const double last = prev[S - 1 + 2];
        const double last2 = S >= 2 ? prev[S - 2 + 2] : -(double)INFINITY;
        const double m = fmax(last, last2);
        const double lp2 = (m == -INFINITY) ? -(double)INFINITY : m + log2(exp2(last - m) + exp2(last2 - m));
        *loss_b = (float)(-lp2 * 0.6931471805599453);
        // log2 Z for the gradient pass: integer part and fraction apart, as the linear lattice hands it over
        const double fl = (m == -INFINITY) ? 0.0 : floor(lp2);
        *zint_b = (int32_t)fl;
        *logz2_b = (m == -INFINITY) ? -INFINITY : (float)(lp2 - fl);
    }
    __syncthreads();
}

// The frames [t_begin, t_begin + frames) of utterance b, one wave per frame (the body of ctc_grad_kernel; the repair of an
// utterance runs it a second time over ALL its frames on the log-domain lattices).
// one wave per frame; work-group = 4 waves x FRAMES_PER_WAVE frames.  NJ = SP / 64 lattice columns per lane: all of a
// frame's alpha / beta loads are issued before the first one is used (one HBM/L2 round trip per frame instead of NJ).
// LIN: the lattice comes from ctc_lattice_wave_kernel (doubles in linear units with one exponent per frame, emissions
// u = p + eps instead of q) and is brought to log2 units on the fly; the sum of a frame's state posteriors must then be
// 1 -- if the linear lattice lost mass to underflow it is not, and the utterance is flagged for the log-domain repair.
template <int NJ, int LIN>  // LIN: 0 = log-domain rows; 1 = linear rows in doubles (exponent blocks of 16); 2 = in floats (of 8);
                            // 3 = doubles from the wave-PAIR lattice (one exponent per FOUR states: 128 per block of 16)
__device__ __forceinline__ void ctc_grad_frames(
    const float* __restrict__ probs, const float* __restrict__ logq, const void* __restrict__ alpha_v,
    const void* __restrict__ beta_v, const int32_t* __restrict__ ea, const int32_t* __restrict__ eb,
    const float* __restrict__ logz2, const int32_t* __restrict__ zint, const float nll, void* __restrict__ dlogits,
    const int* s_lab, const int* s_pos, const int* s_start, float* s_lq, float* s_gam, int b, int S, int T, int t_out,
    int k, int l_max, int sp, int blank, int t_begin, int frames_per_wg, int g_row0, int g_rs, long g_bs, int out_f32,
    float eps, float grad_scale, int32_t* __restrict__ flags) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // LIN: log2 of the partition sum in u units = zint[b] + logz2[b], integer part and fraction kept apart
    // (loss[b] = -ln Z_u - sum_t ln c_t was written by the lattice wave)
    const bool feasible = nll < INFINITY;
    // lattice units are log2 (LIN == 4: the fraction of the REPAIRED lattice's log2 Z, just written by this work-group)
    const float log_p = LIN == 4 ? *(const volatile float*)&logz2[b] : (LIN ? logz2[b] : -nll * LOG2E);
    float* gam = s_gam + wave * l_max;
    float* wlq = s_lq + wave * 64;
    for (int tt = wave; tt < frames_per_wg; tt += 4) {
        const int t = t_begin + tt;
        if (t >= t_out) break;
        const long fidx = (long)b * t_out + t;
        float dz = 0.f;
#if defined(SL_PROBE_CTC_TOTALS)
        float probe_total = 0.f, probe_na = 0.f, probe_nb = 0.f, probe_nab = 0.f;
#endif
        if (t < T) {
            const float lqv = lane < k ? logq[fidx * k + lane] : 0.f;
            const float pk = lane < k ? probs[fidx * k + lane] : 0.f;
            float occ = 0.f;
            if (feasible) {
                float av[NJ], bv[NJ];  // LIN: av = fractional part (log2 of the two mantissas), bv unused, ai = integer part
                int ai[NJ];
                if (LIN == 4) {
                    // log-domain rows in DOUBLES (the repair pass): alpha + beta - floor(log2 Z) in doubles, the small rest in fp32
                    const double* al = (const double*)alpha_v + fidx * sp;
                    const double* be = (const double*)beta_v + fidx * sp;
                    const double zi = (double)*(const volatile int32_t*)&zint[b];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const bool in = 64 * j < sp;
                        const double ad = in ? al[lane + 64 * j] : -(double)INFINITY;
                        const double bd = in ? be[lane + 64 * j] : -(double)INFINITY;
                        av[j] = (ad == -INFINITY || bd == -INFINITY) ? -INFINITY : (float)((ad + bd) - zi);
                        bv[j] = 0.f;
                        ai[j] = 0;
                    }
                } else if (LIN) {
                    typedef typename std::conditional<LIN == 2, float, double>::type RT;
                    RT ad[NJ], bd[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {  // only the live part of the row was written: s < S rounded up to 8
                        const bool in = lane + 64 * j < ((S + 7) & ~7);
                        if (LIN == 1 || LIN == 3) {  // the high words of the doubles (wave_lattice_run / pair_lattice_run)
                            ad[j] = in ? (RT)__hiloint2double(((const int*)alpha_v)[fidx * sp + lane + 64 * j], 0) : (RT)0;
                            bd[j] = in ? (RT)__hiloint2double(((const int*)beta_v)[fidx * sp + lane + 64 * j], 0) : (RT)0;
                        } else {
                            ad[j] = in ? (RT)((const float*)alpha_v)[fidx * sp + lane + 64 * j] : (RT)0;
                            bd[j] = in ? (RT)((const float*)beta_v)[fidx * sp + lane + 64 * j] : (RT)0;
                        }
                    }
                    // exponents: one per block of 16 steps of the respective direction (alpha: step = t, beta: T-1-t)
                    // Exponents add up to tens of thousands over a long utterance while the posterior needs the FRACTION of
                    // the log2 to 1e-3: integer parts (block exponents, frexp exponents, the integer part of log2 Z) are
                    // summed exactly, only the mantissa logarithms go through fp32.
                    // (one exponent per lattice lane = 8 states and block of 16 steps of the respective direction)
                    constexpr int RBS = LIN == 2 ? 3 : 4;  // log2 of the frames per exponent block
                    constexpr int ELANES = LIN == 3 ? 128 : 64;  // exponents per block
                    constexpr int WSH = LIN == 3 ? 2 : 3;        // log2 of the states that share one
                    const int32_t* eap = ea + ((long)b * ((t_out >> RBS) + 1) + (t >> RBS)) * ELANES;
                    const int32_t* ebp = eb + ((long)b * ((t_out >> RBS) + 1) + ((T - 1 - t) >> RBS)) * ELANES;
                    const int zi = zint[b];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int wl = (lane + 64 * j) >> WSH;
                        const int eab = eap[wl] + ebp[wl] - zi;
                        // a state the other direction cannot reach has posterior 0 whatever this direction holds
                        const bool dead = !(ad[j] > (RT)0 && bd[j] > (RT)0 && ad[j] < (RT)INFINITY && bd[j] < (RT)INFINITY);
#if defined(SL_PROBE_CTC_TOTALS)
                        probe_na += (lane + 64 * j < S && ad[j] > (RT)0) ? 1.f : 0.f;
                        probe_nb += (lane + 64 * j < S && bd[j] > (RT)0) ? 1.f : 0.f;
                        probe_nab += (lane + 64 * j < S && !dead) ? 1.f : 0.f;
#endif
                        int xa, xb;
                        const RT ma = frexp(dead ? (RT)1 : ad[j], &xa), mb = frexp(dead ? (RT)1 : bd[j], &xb);
                        av[j] = dead ? -INFINITY : __builtin_amdgcn_logf((float)ma) + __builtin_amdgcn_logf((float)mb);
                        ai[j] = eab + xa + xb;
                        bv[j] = 0.f;
                    }
                } else {
                    const float* al = (const float*)alpha_v + fidx * sp;
                    const float* be = (const float*)beta_v + fidx * sp;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {  // rows are sp (a multiple of 64) wide: a wave-uniform bound
                        const bool in = 64 * j < sp;
                        av[j] = in ? al[lane + 64 * j] : -INFINITY;
                        bv[j] = in ? be[lane + 64 * j] : -INFINITY;
                        ai[j] = 0;
                    }
                }
                // emission in lattice units: log2 q (log-domain lattice) or log2 (p + eps) (linear lattice)
                wlq[lane] = (LIN && LIN != 4) ? __builtin_amdgcn_logf(pk + eps) : lqv * LOG2E;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const float lq_blank = wlq[blank];
                // blank states (even s) -> butterfly sum; grapheme states (odd s) -> gamma[] in LDS
                float blank_part = 0.f, label_part = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int s = lane + 64 * j;
                    if (s < S) {
                        const float ab = av[j] + bv[j];
                        const float add = LIN ? (float)ai[j] : 0.f;  // (LIN: log_p is the fraction of log2 Z only)
                        if (s & 1) {
                            const int pos = s >> 1;
                            const float lg = (ab - wlq[s_lab[pos]] - log_p) + add;
                            const float gv = (ab == -INFINITY) ? 0.f : exp2f(lg);
                            gam[pos] = gv;
                            label_part += gv;
                        } else {
                            const float lg = (ab - lq_blank - log_p) + add;
                            blank_part += (ab == -INFINITY) ? 0.f : exp2f(lg);
                        }
                    }
                }
                blank_part = wave_sum(blank_part);
#if defined(SL_PROBE_CTC_TOTALS)  // diagnosis: the frame's posterior total - 1 instead of class 0's gradient (wrong results)
                probe_total = blank_part + wave_sum(label_part) - 1.f;
#endif
                if (LIN && flags != nullptr) {
                    const float total = blank_part + wave_sum(label_part);
                    if (lane == 0 && !(fabsf(total - 1.f) < 4e-3f))
                        __hip_atomic_fetch_or(&flags[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (lane < k) {
                    if (lane == blank) {
                        occ = blank_part;
                    } else {
                        const int e0 = s_start[lane], e1 = s_start[lane + 1];
                        for (int i = e0; i < e1; ++i) occ += gam[s_pos[i]];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            // du_k = q_k - occ_k ; dp_k = du_k / (p_k + eps) ; dz_k = p_k * (dp_k - sum_j p_j dp_j)
            float dp = 0.f;
            if (lane < k) dp = (expf(lqv) - occ) / (pk + eps);
            const float inner = wave_sum(pk * dp);
            dz = pk * (dp - inner) * grad_scale;
#if defined(SL_PROBE_CTC_TOTALS)
            {
                const float na = wave_sum(probe_na), nb = wave_sum(probe_nb), nab = wave_sum(probe_nab);
                if (lane == 0) dz = probe_total;
                if (lane == 1) dz = na;
                if (lane == 2) dz = nb;
                if (lane == 3) dz = nab;
            }
#endif
        }
        if (lane < k) {
            const long gi = (long)b * g_bs + (long)(g_row0 + t) * g_rs + lane;
            // device-scope stores (written through this XCD's L2): the repair pass at the tail of ctc_grad_kernel may
            // rewrite these rows from a work-group on another XCD, and a line left dirty here would land on top of it
            if (out_f32)
                __hip_atomic_store((float*)dlogits + gi, dz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                __hip_atomic_store((unsigned short*)dlogits + gi, f32_to_bf16_bits(dz), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// One wave per frame; work-group = 4 waves x frames_per_wg / 4 frames of one utterance.
// tickets (with LIN): THE REPAIR PASS LIVES IN THIS KERNEL.  Every work-group of utterance b raises its done slot when its
// frames are written; the work-group with the utterance's highest index waits for the others' slots, reads the utterance's flag
// -- set by the lattice wave (no alignment / overflow) or by any work-group of this launch (a frame's posteriors did not
// sum to 1) -- and, if it is set, redoes the whole utterance by itself: log-domain alpha and beta lattices into rep_alpha /
// rep_beta (repair_lattices), the loss, then the gradient of every frame from them.  Normally no flag is set and it just
// leaves: the two launches that used to follow (an empty log-domain lattice and an empty gradient pass, 5.5 + 5.3 us
// every step) are gone.
template <int NJ, int LIN>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ probs, const float* __restrict__ logq,
                                                       const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_len,
                                                       const int32_t* __restrict__ input_len,
                                                       const void* __restrict__ alpha_v, const void* __restrict__ beta_v,
                                                       const int32_t* __restrict__ ea, const int32_t* __restrict__ eb,
                                                       const float* __restrict__ logz2, const int32_t* __restrict__ zint,
                                                       float* __restrict__ loss,
                                                       const int32_t* __restrict__ cls, void* __restrict__ dlogits,
                                                       int t_out, int k, int l_max, int sp, int blank, int frames_per_wg,
                                                       int g_row0, int g_rs, long g_bs, int out_f32, float eps,
                                                       float grad_scale, int32_t* __restrict__ flags,
                                                       int32_t* __restrict__ tickets, double* __restrict__ rep_alpha,
                                                       double* __restrict__ rep_beta, int rep_sp) {
    // LDS: labels[l_max] | class_pos[l_max] | class_start[k+1] | lq[4][64] | gamma[4][l_max]
    extern __shared__ int lds_i[];
    int* s_lab = lds_i;
    int* s_pos = s_lab + l_max;
    int* s_start = s_pos + l_max;
    float* s_lq = (float*)(s_start + (k + 1));
    float* s_gam = s_lq + 4 * 64;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int L = label_len[b];
    const int S = 2 * L + 1;
    int T = input_len[b];
    if (T > t_out) T = t_out;
    const int32_t* lab = labels + (long)b * l_max;
    const int32_t* cpos = cls + (long)b * (l_max + k + 1);
    for (int i = tid; i < L; i += 256) {
        s_lab[i] = lab[i];
        s_pos[i] = cpos[i];
    }
    if (tid <= k) s_start[tid] = cpos[l_max + tid];
    __syncthreads();
    ctc_grad_frames<NJ, LIN>(probs, logq, alpha_v, beta_v, ea, eb, logz2, zint, loss[b], dlogits, s_lab, s_pos, s_start, s_lq,
                             s_gam, b, S, T, t_out, k, l_max, sp, blank, blockIdx.x * frames_per_wg, frames_per_wg, g_row0,
                             g_rs, g_bs, out_f32, eps, grad_scale, flags);
    if constexpr (LIN != 0) {
        if (tickets == nullptr) return;
        __shared__ int s_flag;
        __shared__ double s_rows[4 * (512 + 4)];
        // No fence, no shared counter and nobody waiting in the common path.  The gradient rows are device-scope stores and
        // the flag updates device-scope atomics, complete once acknowledged: every wave waits for its own, then one
        // thread of the work-group raises the work-group's OWN done slot (a device-scope release here writes back the
        // whole L2 -- 2000 of them cost 60 us; tickets added to one counter per utterance serialise at 0.2-0.4 us each
        // across the XCDs -- 11 us with returning adds, what the two launches had cost, 25 us polled).
        // (gfx9-family ISA, gfx950 included: vmcnt counts loads AND stores, so vmcnt(0) means this wave's gradient stores
        // were acknowledged by L2; gfx10+ moved stores to vscnt -- this file is built for gfx950 only.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int siblings = (int)gridDim.x - 1;
        int32_t* slots = tickets + (long)b * gridDim.x;
        if ((int)blockIdx.x != siblings) {
            if (tid == 0) __hip_atomic_store(&slots[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // The work-group with the HIGHEST index of the utterance is dispatched after all its siblings (work-groups start in
        // index order), so they are running or done when it gets here: it polls their slots -- once or twice -- and then
        // reads the flag.  (Should the assumption ever fail the wait gives up and the loss says so; it cannot hang.)
        int polls = 0;
        for (;;) {
            int done = 1;
            for (int i = tid; i < siblings; i += 256)
                done &= __hip_atomic_load(&slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__syncthreads_and(done) || ++polls >= (1 << 20)) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_flag = polls >= (1 << 20) ? -1 : (__hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
        }
        __syncthreads();
        if (s_flag == 0 || T <= 0) return;
        if (s_flag < 0) {
            // the host-visible error: the utterance's loss is NaN (and so is the batch mean Wav2Letter.train logs and the
            // value train_on_batch returns), never a plausible number next to a stale gradient
            if (tid == 0) loss[b] = NAN;
            return;
        }
        __threadfence();
        const float* lq_b = logq + (long)b * t_out * k;
        // (log2 Z of the repaired lattice replaces the linear lattice's in logz2 / zint: written through a non-const alias,
        // read back by ctc_grad_frames<., 4> with volatile loads -- the kernel's own pointers are const __restrict__)
        repair_lattices(lq_b, s_lab, L, S, T, k, blank, rep_sp, rep_alpha + (long)b * t_out * rep_sp,
                        rep_beta + (long)b * t_out * rep_sp, s_rows, &loss[b], const_cast<float*>(logz2) + b,
                        const_cast<int32_t*>(zint) + b);
        __threadfence();  // the lattice rows, the loss and log2 Z, written by other threads of this work-group
        __syncthreads();
        const float nll = *(volatile float*)&loss[b];
        ctc_grad_frames<8, 4>(probs, logq, rep_alpha, rep_beta, nullptr, nullptr, logz2, zint, nll, dlogits, s_lab,
                              s_pos, s_start, s_lq, s_gam, b, S, T, t_out, k, l_max, rep_sp, blank, 0, t_out, g_row0, g_rs,
                              g_bs, out_f32, eps, grad_scale, nullptr);
    }
}

// ---- linear-domain lattice: ONE WAVE per (utterance, direction), eight extended-label states per lane ----------------------
// The log-domain kernel above pays, on every one of the T' sequential frames, three v_exp + one v_log, an LDS round trip
// and a 7-wave barrier (0.23 us per frame).  In the probability domain a frame is two additions and one multiplication
// per state; with the whole row of S <= 512 states in ONE wave (lane l owns states 8l .. 8l+7: even j blanks, odd j
// label position 4l + j/2) the only cross-lane traffic is one value (alpha: state 8l-1; beta: 8l+8 and 8l+9) through a
// DPP wave shift -- no LDS, no barrier.  Range: doubles, the emissions are u = p + eps (the per-frame constant
// 1 / sum_j (p_j + eps) that turns u into TF's q is common to a row and goes into the loss as sum_t ln c_t), and every
// 16 frames the row is rescaled by a power of two so that the largest RELEVANT state (alpha: one that can still reach
// the end; beta: one the start can reach) sits at 2^TARGET, the exponent travelling per frame; irrelevant states are
// zeroed at the same time (they never feed relevant ones).  A state more than ~2^1500 below the relevant maximum
// underflows; whether that lost anything is checked by the gradient kernel (sum of a frame's posteriors = 1), and a
// flagged utterance is redone in the log domain (repair pass: the tail of ctc_grad_kernel, nothing to launch).
constexpr int WNS = 8;         // states per lane

// Number type of the lattice.  double (the default): 2^+-1022 of range, rescale every 16 frames to 2^500.
// float (sl_ctc_select 5..7, measurement): a frame's 20 operations run at twice the rate and the row stores are half as
// wide -- 83 instead of 106 us per call at 32 x 500 frames -- but 2^-134 .. 2^119 around a lane's maximum is not enough
// everywhere: (a) a lane that holds a few early paths when the bulk of the mass arrives within a block grows by more
// than 2^119 (blank logit + 5 .. 10 over the labels: overflow at the lattice front); a floor on the exponent from the
// source lane's maximum (E >= E_source - 100) cures that, but (b) with every label at the 1e-8 floor a lane's own states
// are 2^-106 per lane below the source lane's and ARE what reaches the end of the utterance -- the floor flushes them.
// No per-lane exponent serves both without knowing the emission costs in between; a per-frame overflow check would
// cost what the float arithmetic saves.  Either failure is caught (sum of posteriors / non-finite Z) and repaired by the
// log-domain pass, at three times the price -- hence doubles by default.
template <typename R>
struct WaveReal;
template <>
struct WaveReal<double> {
    static constexpr int TARGET = 0, RESCALE = 16, SHIFT_MAX = 400, FLOOR = 180;
};
template <>
struct WaveReal<float> {
    static constexpr int TARGET = 8, RESCALE = 8, SHIFT_MAX = 126, FLOOR = 1 << 20;  // (no floor: see above)
};
constexpr int WRESCALE = 16;   // frames of one straight-line block (two prefetch chunks)

__device__ __forceinline__ double dpp_from_lower_lane(double v) {  // lane l <- lane l-1, lane 0 <- 0
    int lo = __double2loint(v), hi = __double2hiint(v);
    // bound_ctrl: the lane without a source reads 0 -- no initialisation of the destination (two v_mov per shifted double and
    // frame of a lone wave that issues one instruction every ~6 cycles; round 6)
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_upper_lane(double v) {  // lane l <- lane l+1, lane 63 <- 0
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);  // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float dpp_from_lower_lane(float v) {
#if defined(SL_PROBE_CTC_NODPP)  // timing probe (wrong results): no cross-lane shift
    return v * 0.5f;
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ float dpp_from_upper_lane(float v) {
#if defined(SL_PROBE_CTC_NODPP)
    return v * 0.5f;
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
#endif
}
__device__ __forceinline__ int wave_frexp_exp(double v) { return __builtin_amdgcn_frexp_exp(v); }
__device__ __forceinline__ int wave_frexp_exp(float v) { return __builtin_amdgcn_frexp_expf(v); }
__device__ __forceinline__ double wave_ldexp(double v, int e) { return ldexp(v, e); }
__device__ __forceinline__ float wave_ldexp(float v, int e) { return ldexpf(v, e); }
__device__ __forceinline__ double wave_max2(double x, double y) { return fmax(x, y); }
__device__ __forceinline__ float wave_max2(float x, float y) { return fmaxf(x, y); }

__device__ __forceinline__ int dpp_int_from_lower_lane(int v, int lane0_value) {
    return __builtin_amdgcn_update_dpp(lane0_value, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_int_from_upper_lane(int v, int lane63_value) {
    return __builtin_amdgcn_update_dpp(lane63_value, v, 0x130, 0xf, 0xf, false);
}

// {high word of x, high word of y} in ONE instruction (v_pk_mov_b32 with op_sel:[1,1]: tools/pk_mov_probe.hip): a row store packs
// the high words of a lane's doubles, and four v_mov per int4 were 8 of a beta frame's 41 instructions (round 6)
__device__ __forceinline__ unsigned long long pk_high_words(double x, double y) {
    unsigned long long r;
    asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// The recursion of one direction (compile-time DIR: 0 = alpha, forwards; 1 = beta, backwards).  Every store is
// unconditional (dead lanes write their row slice to a dump row with stride 0, every lane writes the frame's exponent to
// the same word): an exec-masked store between a prefetch load and its use would force s_waitcnt vmcnt(0) per frame.
// HELP (round 6): a second wave of the work-group (lattice_helper) fetches the probabilities and GATHERS, per frame and lattice
// lane, the lane's four label emissions and the blank's into 48 contiguous bytes of LDS -- the lattice wave then reads a frame's
// emissions with three ds_read_b128 instead of five ds_read_b64 and does no staging of its own (4 loads, 4 conversions, 4 LDS
// writes and their address arithmetic per 8 frames).  A lone wave's frame time is its instruction count (section 6 of DESIGN.md).
struct HelpShared {
    double* gath;    // [2][8 frames][64 lanes][6]: {uq0, uq1, uq2, uq3, ub, -}
    int* progress;   // chunks (of 8 frames) the helper has published
    int* consumed;   // chunks the lattice wave has finished reading
};
template <int DIR, typename R, bool HELP = false>
__device__ __forceinline__ void wave_lattice_run(const float* __restrict__ pr, const int32_t* __restrict__ lab,
                                                 uint32_t* __restrict__ rows, uint32_t* __restrict__ dump,
                                                 int32_t* __restrict__ eout, int lane, int L, int S, int T, int k,
                                                 int blank, float eps, R* a, int* e_final,
                                                 const HelpShared hs = HelpShared()) {
    constexpr int RB = WaveReal<R>::RESCALE;  // frames between rescales = frames per stored exponent
    constexpr int ZERO_SLOT = 63;             // emission slot of the dead label positions (k <= 63)
    // label slots of this lane: position 4 * lane + i sits in state 8 * lane + 2 * i + 1
    int col[4];
    R sk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos = 4 * lane + i;
        const bool slot_live = pos < L;
        const int me = slot_live ? lab[pos] : blank;
        col[i] = slot_live ? me : ZERO_SLOT;  // dead label slots: u = 0
        bool skip;
        if (DIR == 0)
            skip = slot_live && pos >= 1 && lab[pos - 1] != me;
        else
            skip = pos + 1 < L && lab[pos + 1] != me;
        sk[i] = skip ? (R)1 : (R)0;
    }
    const int tstart = DIR == 0 ? 0 : T - 1;
    const int tstep = DIR == 0 ? 1 : -1;
    const bool lane_live = WNS * lane < S;
    uint32_t* rowp = lane_live ? rows + (long)tstart * (64 * WNS) + WNS * lane : dump + WNS * lane;
    const long row_inc = lane_live ? (long)tstep * (64 * WNS) : 0;

    // Emissions: the raw probabilities of an 8-frame chunk are ONE contiguous span of 8 * k floats; it is fetched one
    // chunk ahead with four coalesced loads per lane and handed out through LDS (a lone wave executes its LDS operations
    // in order, so the read after the write needs no barrier).  Per-lane gathers straight from global memory would put
    // ten vector-memory operations per frame behind the 6-bit vmcnt counter, whose 63 slots divided by the store round
    // trip (~1.5 us) is what paced the first version of this kernel (0.235 us per frame).  The LDS copy holds u = p + eps
    // already in the lattice's number type (the conversions and the five adds per frame were a third of a frame's time:
    // s_memtime probe, 473 -> 324 ticks without them) and a zero in slot 63 for the dead label positions.
    __shared__ R emis[HELP ? 1 : 2][HELP ? 1 : 8 * 64 + 64];  // (+ 64: where the lanes beyond 8 * k of a chunk put their value)
    if (!HELP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            emis[0][j * 64 + lane] = (R)0;
            emis[HELP ? 0 : 1][j * 64 + lane] = (R)0;
        }
    }
    // per-lane constants of the chunk transfer (k is not a compile-time constant: no division inside the loop); frames
    // past the end of the utterance re-read its last frame (their results are never used)
    int st_idx[4], ld_jf[4], ld_c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int idx = lane + 64 * r;
        const int jf = idx / k;
        ld_c[r] = idx - jf * k;
        ld_jf[r] = jf < 8 ? jf : 7;
        st_idx[r] = jf < 8 ? jf * 64 + ld_c[r] : 8 * 64 + lane;
    }
    const float* chunk0 = pr + (long)tstart * k;
    const int kstep = tstep * k;
    auto fetch_chunk = [&](int base, float* e4) {
        if (HELP) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) e4[r] = chunk0[min(base + ld_jf[r], T - 1) * kstep + ld_c[r]];
    };
    auto stage_chunk = [&](int buf, const float* e4) {
        if (HELP) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) emis[HELP ? 0 : buf][st_idx[r]] = (R)(e4[r] + eps);
    };
    float e4[4] = {0.f, 0.f, 0.f, 0.f};
    fetch_chunk(0, e4);
    stage_chunk(0, e4);
    // HELP: hand-over with the helper wave per chunk of 8 frames (plain LDS accesses in program order, relaxed work-group
    // atomics on the counters: see pair_lattice_run)
    int chunks_done = 0;
    auto await_chunk = [&]() {  // the helper has published chunk `chunks_done`
        if (HELP) {
            for (int spins = 0; spins < (1 << 22); ++spins) {
                if (__hip_atomic_load(hs.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > chunks_done) break;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        }
    };
    auto release_chunk = [&]() {  // (every lane stores the same value to the same word: no exec mask)
        if (HELP) {
            ++chunks_done;
            asm volatile("" ::: "memory");
            __hip_atomic_store(hs.consumed, chunks_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
#pragma unroll
    for (int j = 0; j < WNS; ++j) a[j] = (R)0;
    int E = 0;              // this lane's exponent: true value = stored value * 2^E
    bool lane_zero = true;  // nothing has reached this lane's states yet (as of the last rescale)
    R fscale = (R)1;        // 2^(neighbour's exponent - E), fixed between rescales

    // Block floating point PER LANE: lane l's eight states share the exponent E (true value = stored * 2^E), brought
    // back to 2^TARGET every RB steps with no cross-lane reduction; the one value (beta: two) that crosses a lane
    // boundary per frame is rescaled by the exponent difference on the way (frame()).  A single exponent per ROW is not
    // enough: early in training the net says "blank" with p ~ 1 and every label with p ~ eps, so each label a state has
    // consumed costs 2^-26 and the states of one row span thousands of binades -- per lane (four labels) they span a
    // few hundred at most.  Also zeroes the states that cannot matter any more (they never feed ones that do).
    // The row in registers is that of step base - 1.  Branch-free.
    auto rescale = [&](int base) {
        const int tp = tstart + tstep * (base - 1);
        const int lo_edge = DIR == 0 ? S - 2 * (T - tp) : 0;  // alpha: states below cannot reach the end any more
        const int hi_edge = DIR == 0 ? S : 2 * tp + 1;        // beta: states above are unreachable from the start
        // the exponent the block that ends here was stored under (a lane may have taken over its neighbour's exponent
        // in the middle of the block; its rows before that are zeros, which any exponent describes)
        eout[max(base / RB - 1, 0) * 64 + lane] = E;
        R m = (R)0;
#pragma unroll
        for (int j = 0; j < WNS; ++j) {
            const int st = WNS * lane + j;
            a[j] = (st < lo_edge || st > hi_edge) ? (R)0 : a[j];
            m = wave_max2(m, a[j]);
        }
        lane_zero = !(m > (R)0);
        const int shift = lane_zero ? 0 : WaveReal<R>::TARGET - wave_frexp_exp(m);
#pragma unroll
        for (int j = 0; j < WNS; ++j) a[j] = wave_ldexp(a[j], shift);
        E -= shift;
        // A lane that holds nothing yet takes over the exponent of the neighbour its first mass will come from -- once
        // per block, not per frame: in RB frames mass moves at most 2 RB states = RB / 4 lanes (skip transitions), so
        // RB / 4 + 1 rounds of "empty lane <- neighbour" cover every lane that can be reached before the next rescale.
        // ... and a lane that holds SOMETHING, but more than 2^FLOOR below that neighbour's level, is lifted to within
        // 2^FLOOR of it (its own values shrink accordingly, at worst to zero).  Behind a net that has learnt its labels
        // the mass ahead of the alignment's front pays eps at every frame the front does not -- lanes ahead fall
        // thousands of binades below the lane the front is in, and when the front crossed into them the exponent
        // difference exceeded what a double holds (TARGET 500 / no floor: overflow four lanes on, NaN loss, every utterance
        // through the repair pass at 0.9 ms per call from the first epoch that fits anything).  Legitimate spreads are
        // bounded by the eps floor on the emissions: four labels per lane = 2^106 between neighbours; the front crosses
        // at most RB / 4 + 1 lanes per block: 5 x 180 + 25 binades of growth stay inside a double above TARGET = 2^0.
        // (Only a neighbour that holds mass sets a level: the lanes the recursion has left behind for good are zero and
        // keep whatever exponent they had.)
        const int e_own = E;
        const int zsrc = DIR == 0 ? dpp_int_from_lower_lane(lane_zero ? 1 : 0, 1) : dpp_int_from_upper_lane(lane_zero ? 1 : 0, 1);
#pragma unroll
        for (int round = 0; round < RB / 4 + 1; ++round) {
            const int en = DIR == 0 ? dpp_int_from_lower_lane(E, E) : dpp_int_from_upper_lane(E, E);
            E = lane_zero ? en : (zsrc ? E : max(E, en - WaveReal<R>::FLOOR));
        }
        if (WaveReal<R>::FLOOR < (1 << 19)) {
            // Round 6 (tools/fuzz_ctc.py seed 64, case 136: 250 labels in 299 frames, a learnt alignment followed by a blank
            // collapse).  The chain of lifts above reaches RB / 4 + 1 lanes per rescale; a lane one further on kept its own
            // stale mass (thousands of binades below the front), and at the NEXT rescale its source neighbour -- flushed to zero
            // by its own lift, holding the front's exponent -- no longer "held mass", so the lane was not lifted then either:
            // the front's mass crossed into it under the SHIFT_MAX clamp of fscale, i.e. was crushed (beta of the first four
            // frames lost, posteriors 0, the utterance through the repair pass).  A lane that holds something but sits more
            // than 2^SHIFT_MAX below a source neighbour that is still in play (zero or not: a zero lane carries the exponent it
            // adopted from ITS source) is lifted like the others; what it held is > 2^(SHIFT_MAX - FLOOR) below what is about
            // to arrive.  Neighbours the recursion has left behind for good (beyond lo_edge / hi_edge) keep stale exponents
            // from long ago and set no level, as before.
            const int en_r = DIR == 0 ? dpp_int_from_lower_lane(E, E) : dpp_int_from_upper_lane(E, E);
            const bool src_gone = DIR == 0 ? (WNS * lane - 1 < lo_edge) : (WNS * (lane + 1) > min(hi_edge, S - 1));  // (lanes beyond S never held anything)
            E = (!lane_zero && !src_gone && en_r - E > WaveReal<R>::SHIFT_MAX) ? en_r - WaveReal<R>::FLOOR : E;
            const int lift = lane_zero ? 0 : E - e_own;  // >= 0
#pragma unroll
            for (int j = 0; j < WNS; ++j) a[j] = wave_ldexp(a[j], -lift);
        }
        const int en = DIR == 0 ? dpp_int_from_lower_lane(E, E) : dpp_int_from_upper_lane(E, E);
        fscale = wave_ldexp((R)1, max(min(en - E, WaveReal<R>::SHIFT_MAX), -4 * WaveReal<R>::SHIFT_MAX));
    };
    // one frame: emissions from LDS, three phases of mutually independent operations (a lone wave hides no latency by
    // itself: the eight two-term sums, the four skip terms, the eight products; the empty asm statements pin the phase
    // order), store of the row.
    auto frame_core = [&](bool first, const R ub, const R (&uq)[4]) {
        R n[WNS];
        if (DIR == 0) {
            // state 8l - 1 (a label state) of the lane below, brought to this lane's exponent
            const R below = dpp_from_lower_lane(a[7]) * fscale;
            n[0] = a[0] + below;
#pragma unroll
            for (int i = 1; i < WNS; ++i) n[i] = a[i] + a[i - 1];
            asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]));
            n[1] = fma(sk[0], below, n[1]);
            n[3] = fma(sk[1], a[1], n[3]);
            n[5] = fma(sk[2], a[3], n[5]);
            n[7] = fma(sk[3], a[5], n[7]);
        } else {
            const R up0 = dpp_from_upper_lane(a[0]) * fscale;  // state 8l + 8 (blank)
            const R up1 = dpp_from_upper_lane(a[1]) * fscale;  // state 8l + 9 (label)
            n[7] = a[7] + up0;
#pragma unroll
            for (int i = 0; i < WNS - 1; ++i) n[i] = a[i] + a[i + 1];
            asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]));
            n[7] = fma(sk[3], up1, n[7]);
            n[5] = fma(sk[2], a[7], n[5]);
            n[3] = fma(sk[1], a[5], n[3]);
            n[1] = fma(sk[0], a[3], n[1]);
        }
        asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            n[2 * i] *= ub;
            n[2 * i + 1] *= uq[i];
        }
        if (first) {  // step 0: the recursion above ran on zeros; only the entry states are set
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bool blank_entry, label_entry;
                if (DIR == 0) {
                    blank_entry = lane == 0 && i == 0;                       // state 0
                    label_entry = lane == 0 && i == 0;                       // state 1 (u = 0 for an empty label)
                } else {
                    blank_entry = WNS * lane + 2 * i == S - 1;               // state S - 1
                    label_entry = L > 0 && WNS * lane + 2 * i + 1 == S - 2;  // state S - 2
                }
                n[2 * i] = blank_entry ? ub : (R)0;
                n[2 * i + 1] = label_entry ? uq[i] : (R)0;
            }
        }
        // dead blank states (s >= S, even) only ever see zeros: their label neighbours have u = 0
#pragma unroll
        for (int i = 0; i < WNS; ++i) a[i] = n[i];
#if !defined(SL_PROBE_CTC_NOSTORE)  // (timing probe: no lattice stores)
        if (sizeof(R) == 8) {
            // THE HIGH WORDS ONLY: sign, the full 11-bit exponent and 20 mantissa bits of every state (the recursion keeps
            // the doubles).  The row stores were 136 of a frame's 333 cycles (clock probe: four 1 KB store instructions
            // through a 64 B/clk port); half the bytes, no conversion instruction -- the high word IS a register.  The
            // gradient kernel reads a state as (high word, 0): truncated by < 2^-20, a posterior by < 1e-6 relative.
#pragma unroll
            for (int i = 0; i < WNS; i += 4)
                *(ulonglong2*)(rowp + i) = make_ulonglong2(pk_high_words((double)a[i], (double)a[i + 1]),
                                                           pk_high_words((double)a[i + 2], (double)a[i + 3]));
        } else {
#pragma unroll
            for (int i = 0; i < WNS; i += 4)
                *(float4*)((float*)rowp + i) = make_float4((float)a[i], (float)a[i + 1], (float)a[i + 2], (float)a[i + 3]);
        }
#endif
        rowp += row_inc;
    };
    // (Issuing a frame's LDS reads one frame ahead -- inline-asm reads, hand-placed wait -- was measured and changed
    // nothing: the reads are not what a frame waits for.  A frame is 47 instructions of a lone wave, ~7 cycles each.)
    auto frame = [&](bool first, const R* erow) {
        R uq[4];
        if constexpr (HELP) {  // erow: this lane's {uq0, uq1, uq2, uq3, ub, -} of the frame
            const double2 q01 = ((const double2*)erow)[0], q23 = ((const double2*)erow)[1], qb = ((const double2*)erow)[2];
            uq[0] = (R)q01.x;
            uq[1] = (R)q01.y;
            uq[2] = (R)q23.x;
            uq[3] = (R)q23.y;
            frame_core(first, (R)qb.x, uq);
            return;
        }
#if defined(SL_PROBE_CTC_NOEMIS)  // timing probe (wrong results): no emission reads
        const R ub = (R)0.03;
#pragma unroll
        for (int i = 0; i < 4; ++i) uq[i] = (R)0.03;
#else
        const R ub = erow[blank];
#pragma unroll
        for (int i = 0; i < 4; ++i) uq[i] = erow[col[i]];
#endif
        frame_core(first, ub, uq);
    };

    // a frame's emission source: row j of staging buffer `buf` (by class), or -- HELP -- this lane's gathered 48 bytes
    auto erow_of = [&](int buf, int j) -> const R* {
        if constexpr (HELP)
            return (const R*)(hs.gath + ((long)(buf * 8 + j) * 64 + lane) * 6);
        else
            return &emis[buf][j * 64];
    };
    // Full blocks of 16 steps run as straight-line code: the waitcnt pass can then count the stores that are
    // younger than a chunk's prefetch loads exactly (with a branch per frame it assumed none and waited for the stores'
    // round trip once per chunk).
    int base = 0;
    if (T >= WRESCALE) {
        rescale(0);
        await_chunk();
        fetch_chunk(8, e4);
        frame(true, erow_of(0, 0));  // (peeled: the only frame with the entry-state special case)
#pragma unroll
        for (int j = 1; j < 8; ++j) frame(false, erow_of(0, j));
        stage_chunk(1, e4);
        release_chunk();
        await_chunk();
        fetch_chunk(16, e4);
        if (RB == 8) rescale(8);
#pragma unroll
        for (int j = 0; j < 8; ++j) frame(false, erow_of(1, j));
        stage_chunk(0, e4);
        release_chunk();
        base = WRESCALE;
    }
    for (; base + WRESCALE <= T; base += WRESCALE) {
        rescale(base);
        await_chunk();
        fetch_chunk(base + 8, e4);
#pragma unroll
        for (int j = 0; j < 8; ++j) frame(false, erow_of(0, j));
        stage_chunk(1, e4);
        release_chunk();
        await_chunk();
        fetch_chunk(base + 16, e4);
        if (RB == 8) rescale(base + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) frame(false, erow_of(1, j));
        stage_chunk(0, e4);
        release_chunk();
    }
    // tail: fewer than 16 steps left (the next chunk is already staged in buffer 0)
    if (base < T) {
        rescale(base);
        await_chunk();
        fetch_chunk(base + 8, e4);
        for (int j = 0; j < 8 && base + j < T; ++j) frame(base + j == 0, erow_of(0, j));
        stage_chunk(1, e4);
        release_chunk();
        if (base + 8 < T) await_chunk();
        if (RB == 8 && base + 8 < T) rescale(base + 8);
        for (int j = 0; j < 8 && base + 8 + j < T; ++j) frame(false, erow_of(1, j));
    }
    eout[((T - 1) / RB) * 64 + lane] = E;  // the last (possibly partial) block
    *e_final = E;
}

// ---- the same lattice on a PAIR of waves per (utterance, direction) (round 6) -----------------------------------------------
// The lone wave is latency-bound: 265 cycles per frame, of which the 20 dependent double operations of a lane's eight states
// are 125 and the two 1 KB row stores 68 (0.10 ms of the config-3 step, 0.39 ms at config 5, nothing
// beside it).  Here a lane owns FOUR states (lane gl of 128 owns states 4 gl .. 4 gl + 3, label positions 2 gl, 2 gl + 1): half
// the arithmetic, one row store and three emission reads per wave and frame.  The recursion moves mass in ONE direction across
// the wave boundary (alpha: from wave 0's lane 63 up into wave 1's lane 0; beta: from wave 1's lane 0 down into wave 0's lane
// 63), so the upstream wave never waits: it writes its boundary state(s) of every row into an LDS mailbox (one slot per row)
// and raises a progress counter three times per block of 16 frames -- after the block's rescale (with the boundary lane's
// new exponent and its empty flag) and after each half of 8 frames -- and the downstream wave, which needs row t - 1 for row t,
// runs half a block behind: it polls the counter once per half block, fetches the half block's eight boundary values in one
// go and otherwise executes exactly the single-wave recursion.  A wave's LDS operations execute in order and LDS has no
// caches, so "data written before the counter, counter read before the data" needs no fence.  Block floating point as in
// wave_lattice_run; the boundary lane of the downstream wave sees the upstream boundary lane's FINAL exponent of the block in
// every adoption round (a valid assignment like any other: exponents only re-express the values).
constexpr int PNS = 4;            // states per lane
constexpr int PLANES = 128;       // lanes per direction
constexpr int PAIR_EMIS = 8 * 64 + 64;

struct PairShared {
    double* mbox;     // [T + 1][2]: boundary state(s) of row r at slot r + 1 (slot 0 = the zeros in front of row 0)
    double* mdump;    // [128][2]: where the lanes that are not at the wave boundary put their (unused) copy -- unconditional stores
    int* ebox;        // [blocks + 1][2]: exponent and empty flag of the upstream boundary lane after the block's rescale
    int* progress;    // events the upstream wave has completed (3 per block)
    double* emis;     // [2 waves][2][PAIR_EMIS]
};

// The hand-over is built from PLAIN LDS accesses in program order, a compiler barrier and relaxed work-group atomics on the
// counter: `volatile` (or acquire / release) accesses make the memory legaliser put s_waitcnt lgkmcnt(0) -- or vmcnt(0): the
// row stores' round trip -- behind every one of them, per frame (the first version of this kernel: 730 cycles per frame).
template <int DIR>
__device__ __forceinline__ void pair_lattice_run(const float* __restrict__ pr, const int32_t* __restrict__ lab,
                                                 uint32_t* __restrict__ rows, uint32_t* __restrict__ dump,
                                                 int32_t* __restrict__ eout, const PairShared sh, int wv, int lane, int L,
                                                 int S, int T, int k, int blank, float eps, double* a, int* e_final) {
    typedef double R;
    constexpr int RB = 16;
    constexpr int ZERO_SLOT = 63;
    constexpr int FLOOR = WaveReal<double>::FLOOR, SHIFT_MAX = WaveReal<double>::SHIFT_MAX, TARGET = WaveReal<double>::TARGET;
    const int gl = wv * 64 + lane;                 // lane of the direction: states 4 gl .. 4 gl + 3
    const bool upstream = DIR == 0 ? wv == 0 : wv == 1;
    // the lane whose neighbour lives in the other wave (downstream side) / the lane the other wave reads (upstream side)
    const bool edge_lane = DIR == 0 ? (upstream ? lane == 63 : lane == 0) : (upstream ? lane == 0 : lane == 63);
    const bool from_other = !upstream && edge_lane;
    int col[2];
    R sk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = 2 * gl + i;
        const bool slot_live = pos < L;
        const int me = slot_live ? lab[pos] : blank;
        col[i] = slot_live ? me : ZERO_SLOT;
        bool skip;
        if (DIR == 0)
            skip = slot_live && pos >= 1 && lab[pos - 1] != me;
        else
            skip = pos + 1 < L && lab[pos + 1] != me;
        sk[i] = skip ? (R)1 : (R)0;
    }
    const int tstart = DIR == 0 ? 0 : T - 1;
    const int tstep = DIR == 0 ? 1 : -1;
    const bool lane_live = PNS * gl < S;
    uint32_t* rowp = lane_live ? rows + (long)tstart * 512 + PNS * gl : dump + PNS * gl;
    const long row_inc = lane_live ? (long)tstep * 512 : 0;
    // mailbox store address of this lane: the real slot for the upstream boundary lane (advancing by a row per frame), a
    // private dump slot (stride 0) for everybody else -- every lane stores, no exec mask, no branch
    const bool poster = upstream && edge_lane;
    double* mslot = poster ? sh.mbox : sh.mdump + 2 * gl;   // slot of row -1
    const int mslot_inc = poster ? 2 : 0;

    R* emis0 = sh.emis + (wv * 2 + 0) * PAIR_EMIS;
    R* emis1 = sh.emis + (wv * 2 + 1) * PAIR_EMIS;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        emis0[j * 64 + lane] = (R)0;
        emis1[j * 64 + lane] = (R)0;
    }
    int st_idx[4], ld_jf[4], ld_c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int idx = lane + 64 * r;
        const int jf = idx / k;
        ld_c[r] = idx - jf * k;
        ld_jf[r] = jf < 8 ? jf : 7;
        st_idx[r] = jf < 8 ? jf * 64 + ld_c[r] : 8 * 64 + lane;
    }
    const float* chunk0 = pr + (long)tstart * k;
    const int kstep = tstep * k;
    auto fetch_chunk = [&](int base, float* e4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) e4[r] = chunk0[min(base + ld_jf[r], T - 1) * kstep + ld_c[r]];
    };
    auto stage_chunk = [&](R* buf, const float* e4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[st_idx[r]] = (R)(e4[r] + eps);
    };
    float e4[4];
    fetch_chunk(0, e4);
    stage_chunk(emis0, e4);
#pragma unroll
    for (int j = 0; j < PNS; ++j) a[j] = (R)0;
    int E = 0;
    bool lane_zero = true;
    R fscale = (R)1;
    int events = 0;  // upstream: events published; downstream: events consumed
    int mbase = 0;   // (partial last block: first mailbox slot of the half being run)

    auto publish = [&]() {  // upstream: everything this wave wrote to the mailbox so far precedes this store in LDS order
        if (upstream) {
            ++events;
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(sh.progress, events, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto await = [&]() {    // downstream: the upstream wave has completed the event of the same number
        if (!upstream) {
            ++events;
            // (bounded: the upstream wave never waits, so this ends within a block's time; should that ever fail the wave goes
            // on with what the mailbox holds instead of hanging -- the gradient kernel's sum check then flags the utterance)
            for (int spins = 0; spins < (1 << 22); ++spins) {
                if (__hip_atomic_load(sh.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= events) break;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        }
    };
    // boundary values of the row this wave's registers hold, into the slot `mslot` points at (then on to the next row's)
    auto post_row = [&]() {
        mslot += mslot_inc;
        if (DIR == 0) {
            mslot[0] = a[PNS - 1];
        } else {
            *(double2*)mslot = make_double2(a[0], a[1]);
        }
    };

    auto rescale = [&](int base) {
        const int tp = tstart + tstep * (base - 1);
        const int lo_edge = DIR == 0 ? S - 2 * (T - tp) : 0;
        const int hi_edge = DIR == 0 ? S : 2 * tp + 1;
        eout[max(base / RB - 1, 0) * PLANES + gl] = E;
        R m = (R)0;
#pragma unroll
        for (int j = 0; j < PNS; ++j) {
            const int st = PNS * gl + j;
            a[j] = (st < lo_edge || st > hi_edge) ? (R)0 : a[j];
            m = wave_max2(m, a[j]);
        }
        lane_zero = !(m > (R)0);
        const int shift = lane_zero ? 0 : TARGET - wave_frexp_exp(m);
#pragma unroll
        for (int j = 0; j < PNS; ++j) a[j] = wave_ldexp(a[j], shift);
        E -= shift;
        // the neighbour's values of the lane at the wave boundary: from the other wave (downstream side)
        int x_e = E, x_zero = 1;
        if (!upstream) {
            x_e = sh.ebox[2 * (base / RB)];
            x_zero = sh.ebox[2 * (base / RB) + 1];
        }
        const int e_own = E;
        int zsrc = DIR == 0 ? dpp_int_from_lower_lane(lane_zero ? 1 : 0, 1) : dpp_int_from_upper_lane(lane_zero ? 1 : 0, 1);
        zsrc = from_other ? x_zero : zsrc;
        // in 16 frames mass moves at most 32 states = 8 lanes of four: nine rounds of "empty lane <- neighbour"
#pragma unroll
        for (int round = 0; round < RB / 2 + 1; ++round) {
            int en = DIR == 0 ? dpp_int_from_lower_lane(E, E) : dpp_int_from_upper_lane(E, E);
            en = from_other ? x_e : en;
            E = lane_zero ? en : (zsrc ? E : max(E, en - FLOOR));
        }
        {   // (the rule of wave_lattice_run's rescale: a lane more than 2^SHIFT_MAX below a source neighbour still in play)
            int en_r = DIR == 0 ? dpp_int_from_lower_lane(E, E) : dpp_int_from_upper_lane(E, E);
            en_r = from_other ? x_e : en_r;
            const bool src_gone = DIR == 0 ? (PNS * gl - 1 < lo_edge) : (PNS * (gl + 1) > min(hi_edge, S - 1));
            E = (!lane_zero && !src_gone && en_r - E > SHIFT_MAX) ? en_r - FLOOR : E;
        }
        const int lift = lane_zero ? 0 : E - e_own;
#pragma unroll
        for (int j = 0; j < PNS; ++j) a[j] = wave_ldexp(a[j], -lift);
        int en = DIR == 0 ? dpp_int_from_lower_lane(E, E) : dpp_int_from_upper_lane(E, E);
        en = from_other ? x_e : en;
        fscale = wave_ldexp((R)1, max(min(en - E, SHIFT_MAX), -4 * SHIFT_MAX));
        // upstream: the boundary lane's exponent and empty flag of this block, and row base - 1 AS RESCALED (the row the
        // downstream wave's first frame of the block reads): the slot written last is written again
        if (poster) {
            sh.ebox[2 * (base / RB)] = E;
            sh.ebox[2 * (base / RB) + 1] = lane_zero ? 1 : 0;
        }
        mslot -= mslot_inc;
        post_row();
    };
    // x0 / x1: the boundary value(s) of the previous row from the other wave (used by the downstream edge lane only)
    auto frame_core = [&](const bool first, const R ub, const R (&uq)[2], const R x0, const R x1) {
        R n[PNS];
        if (DIR == 0) {
            R below = dpp_from_lower_lane(a[PNS - 1]);
            below = (from_other ? x0 : below) * fscale;
            n[0] = a[0] + below;
#pragma unroll
            for (int i = 1; i < PNS; ++i) n[i] = a[i] + a[i - 1];
            asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]));
            n[1] = fma(sk[0], below, n[1]);
            n[3] = fma(sk[1], a[1], n[3]);
        } else {
            R up0 = dpp_from_upper_lane(a[0]);
            R up1 = dpp_from_upper_lane(a[1]);
            up0 = (from_other ? x0 : up0) * fscale;
            up1 = (from_other ? x1 : up1) * fscale;
            n[PNS - 1] = a[PNS - 1] + up0;
#pragma unroll
            for (int i = 0; i < PNS - 1; ++i) n[i] = a[i] + a[i + 1];
            asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]));
            n[3] = fma(sk[1], up1, n[3]);
            n[1] = fma(sk[0], a[3], n[1]);
        }
        asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            n[2 * i] *= ub;
            n[2 * i + 1] *= uq[i];
        }
        // step 0 (branch-free: selects; `first` is false at compile time for every frame but the first of a half block): the
        // recursion above ran on zeros, only the entry states are set
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bool blank_entry, label_entry;
            if (DIR == 0) {
                blank_entry = gl == 0 && i == 0;
                label_entry = gl == 0 && i == 0;
            } else {
                blank_entry = PNS * gl + 2 * i == S - 1;
                label_entry = L > 0 && PNS * gl + 2 * i + 1 == S - 2;
            }
            n[2 * i] = first ? (blank_entry ? ub : (R)0) : n[2 * i];
            n[2 * i + 1] = first ? (label_entry ? uq[i] : (R)0) : n[2 * i + 1];
        }
#pragma unroll
        for (int i = 0; i < PNS; ++i) a[i] = n[i];
        *(ulonglong2*)rowp = make_ulonglong2(pk_high_words(a[0], a[1]), pk_high_words(a[2], a[3]));
        rowp += row_inc;
    };
    // the boundary values of rows base - 1 .. base + 6 for a half block (downstream; everything else gets zeros it never uses)
    R x0[8], x1[8];
    auto preload = [&](int base, int count) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x0[j] = (R)0;
            x1[j] = (R)0;
        }
        if (!upstream) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < count) {
                    if (DIR == 0) {
                        x0[j] = sh.mbox[2 * (base + j)];  // slot of row base + j - 1
                    } else {
                        const double2 v = *(const double2*)(sh.mbox + 2 * (base + j));
                        x0[j] = v.x;
                        x1[j] = v.y;
                    }
                }
            }
        }
    };
    // eight frames, straight-line (no branch between a chunk's prefetch loads and their use: the waitcnt pass counts the
    // row stores in between exactly -- see wave_lattice_run)
    auto frames8 = [&](const bool entry, const R* erow0) {
        // a frame's three emission reads are issued ONE FRAME AHEAD: a frame of four states per lane is ~35 instructions, and
        // an LDS read issued inside it was what it waited for (lgkmcnt in front of the products: 106 us per call at 32 x 500)
        R ub = erow0[blank];
        R uq[2] = {erow0[col[0]], erow0[col[1]]};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            R nub = ub, nuq[2] = {uq[0], uq[1]};
            if (j < 7) {
                const R* erow = erow0 + (j + 1) * 64;
                nub = erow[blank];
                nuq[0] = erow[col[0]];
                nuq[1] = erow[col[1]];
            }
            frame_core(j == 0 && entry, ub, uq, x0[j], x1[j]);
            post_row();
            ub = nub;
            uq[0] = nuq[0];
            uq[1] = nuq[1];
        }
    };
    auto frames_some = [&](int count, const bool entry, const R* erow0) {  // (the last, partial block)
        for (int j = 0; j < count; ++j) {
            const R* erow = erow0 + j * 64;
            const R ub = erow[blank];
            R uq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) uq[i] = erow[col[i]];
            R xa = (R)0, xb = (R)0;
            if (!upstream) {
                xa = sh.mbox[2 * (mbase + j)];
                xb = sh.mbox[2 * (mbase + j) + 1];
            }
            frame_core(j == 0 && entry, ub, uq, xa, xb);
            post_row();
        }
    };

    int base = 0;
    for (; base + RB <= T; base += RB) {
        await();  // (downstream: the upstream rescale of this block)
        rescale(base);
        publish();
        await();  // (downstream: the upstream wave's first half of this block)
        preload(base, 8);
        fetch_chunk(base + 8, e4);
        frames8(base == 0, emis0);
        stage_chunk(emis1, e4);
        publish();
        await();
        preload(base + 8, 8);
        fetch_chunk(base + 16, e4);
        frames8(false, emis1);
        stage_chunk(emis0, e4);
        publish();
    }
    if (base < T) {  // fewer than 16 steps left: the same three events
        await();
        rescale(base);
        publish();
        await();
        fetch_chunk(base + 8, e4);
        mbase = base;
        frames_some(min(8, T - base), base == 0, emis0);
        stage_chunk(emis1, e4);
        publish();
        await();
        mbase = base + 8;
        frames_some(max(0, min(8, T - base - 8)), false, emis1);
        publish();
    }
    eout[((T - 1) / RB) * PLANES + gl] = E;
    *e_final = E;
}

// workspace: alpha, beta R[B][T][512] (+ one dump row per utterance and direction); ea, eb int32[B][T/RB+1][64], RB =
// WaveReal<R>::RESCALE; logz2 float[B]; cls as for the log-domain kernel
template <typename R>
__global__ __launch_bounds__(64) void ctc_lattice_wave_kernel(const float* __restrict__ probs,
                                                              const float* __restrict__ logq,
                                                              const int32_t* __restrict__ labels,
                                                              const int32_t* __restrict__ label_len,
                                                              const int32_t* __restrict__ input_len,
                                                              uint32_t* __restrict__ alpha, uint32_t* __restrict__ beta,
                                                              uint32_t* __restrict__ dump, int32_t* __restrict__ ea,
                                                              int32_t* __restrict__ eb, float* __restrict__ logz2,
                                                              int32_t* __restrict__ zint, float* __restrict__ loss,
                                                              int32_t* __restrict__ cls,
                                                              int32_t* __restrict__ flags,
                                                              int32_t* __restrict__ tickets, int grad_wgs, int t_out,
                                                              int k, int l_max, int blank, float eps) {
    extern __shared__ int wl_lds[];  // list builder: l_max + k + 1 ints; alpha wave: 2 doubles
    const int b = blockIdx.x;
    const int dir = blockIdx.y;
    const int lane = threadIdx.x;
    const int L = label_len[b];
    if (dir == 2) {
        // per-class position lists for the gradient kernel (same as in ctc_lattice_kernel, 64 threads)
        // the done slots of this utterance's gradient work-groups (repair pass inside ctc_grad_kernel)
        for (int i = lane; i < grad_wgs; i += 64) tickets[(long)b * grad_wgs + i] = 0;
        int* s_lab = wl_lds;
        int* s_start = s_lab + l_max;
        int32_t* pos_out = cls + (long)b * (l_max + k + 1);
        int32_t* start_out = pos_out + l_max;
        for (int i = lane; i < L; i += 64) s_lab[i] = labels[(long)b * l_max + i];
        for (int i = lane; i <= k; i += 64) s_start[i] = 0;
        __syncthreads();
        int ranks[8];  // l_max <= 255 -> at most 4 positions per thread
        int nmine = 0;
        for (int i = lane; i < L; i += 64) {
            const int c = s_lab[i];
            int r = 0;
            for (int j = 0; j < i; ++j) r += (s_lab[j] == c);
            ranks[nmine++] = r;
            atomicAdd(&s_start[c + 1], 1);
        }
        __syncthreads();
        if (lane == 0)
            for (int c = 0; c < k; ++c) s_start[c + 1] += s_start[c];
        __syncthreads();
        nmine = 0;
        for (int i = lane; i < L; i += 64) pos_out[s_start[s_lab[i]] + ranks[nmine++]] = i;
        for (int i = lane; i <= k; i += 64) start_out[i] = s_start[i];
        return;
    }
    const int S = 2 * L + 1;
    int T = input_len[b];
    if (T > t_out) T = t_out;
    if (T <= 0) {
        if (dir == 0 && lane == 0) {
            loss[b] = INFINITY;
            logz2[b] = 0.f;
            zint[b] = 0;
            flags[b] = 0;
        }
        return;
    }
    const int32_t* lab = labels + (long)b * l_max;
    const float* pr = probs + (long)b * t_out * k;
    constexpr int RB = WaveReal<R>::RESCALE;
    R a[WNS];
    int E;
    if (dir == 0) {
        // sum over the scored frames of ln c_t, c_t = 1 / sum_j (p_j + eps) = q_blank / (p_blank + eps): what separates
        // the u lattice from TF's (q) one
        float csum = 0.f;
        for (int t = lane; t < T; t += 64)
            csum += logq[((long)b * t_out + t) * k + blank] - logf(pr[(long)t * k + blank] + eps);
        csum = wave_sum(csum);
        float repeats_f = 0.f;  // equal neighbours in the label: each needs a blank in between
        for (int i = lane + 1; i < L; i += 64) repeats_f += lab[i] == lab[i - 1] ? 1.f : 0.f;
        const int repeats = (int)wave_sum(repeats_f);
#if defined(SL_PROBE_CTC_CLOCK)
        const long long probe_t0 = clock64();
#endif
        wave_lattice_run<0, R>(pr, lab, alpha + (long)b * t_out * (64 * WNS), dump + (long)(2 * b) * (64 * WNS),
                               ea + (long)b * (t_out / RB + 1) * 64, lane, L, S, T, k, blank, eps, a, &E);
        // Z_u = alpha_{T-1}(S-1) + alpha_{T-1}(S-2)
        double* fin = (double*)wl_lds;
        int* fin_e = (int*)(fin + 2);
#pragma unroll
        for (int i = 0; i < WNS; ++i) {
            if (WNS * lane + i == S - 1) {
                fin[0] = (double)a[i];
                fin_e[0] = E;
            }
            if (WNS * lane + i == S - 2) {
                fin[1] = (double)a[i];
                fin_e[1] = E;
            }
        }
        __syncthreads();
        if (lane == 0) {
            // the two end states may sit in different lanes, i.e. under different exponents
            const double z1 = fin[0], z2 = S >= 2 ? fin[1] : 0.0;
            const int e1 = fin_e[0], e2 = S >= 2 ? fin_e[1] : e1;
            const int ez = (z2 > 0.0 && (z1 == 0.0 || e2 > e1)) ? e2 : e1;
            const double z = ldexp(z1, max(e1 - ez, -2000)) + ldexp(z2, max(e2 - ez, -2000));
            int xz = 0;
            const double mz = frexp(z > 0.0 ? z : 1.0, &xz);
            const float frac = __builtin_amdgcn_logf((float)mz);  // log2 of the mantissa, in [-1, 0)
            logz2[b] = frac;
            zint[b] = xz + ez;
            // -ln Z_u in double: the integer part is in the tens of thousands for a long utterance
            loss[b] = z > 0.0 ? (float)(-((double)(xz + ez) + (double)frac) * 0.6931471805599453 - (double)csum) : INFINITY;
            // Z = 0: no alignment at all (more labels, plus a blank between equal neighbours, than frames: known without
            // any lattice, the loss is +inf and the gradient kernel's convention for it needs no lattice either), or every
            // alignment underflowed: that one goes to the log-domain repair pass.  NaN / inf: the float lattice overflowed.
            flags[b] = (z > 0.0 && z < INFINITY) ? 0 : (z == 0.0 && L + repeats > T ? 0 : 1);
#if defined(SL_PROBE_CTC_CLOCK)  // timing probe: s_memtime ticks per frame of the alpha recursion instead of the loss
            loss[b] = (float)(clock64() - probe_t0) / (float)T;
#endif
        }
    } else {
        wave_lattice_run<1, R>(pr, lab, beta + (long)b * t_out * (64 * WNS), dump + (long)(2 * b + 1) * (64 * WNS),
                               eb + (long)b * (t_out / RB + 1) * 64, lane, L, S, T, k, blank, eps, a, &E);
    }
}

// ---- the helper wave of wave_lattice_run<DIR, double, true> ----------------------------------------------------------------
// Fetches the probabilities of a chunk of 8 frames with four coalesced loads per lane (one chunk ahead), stages u = p + eps by
// class in its own LDS area, gathers for every lattice lane the emissions of its four label slots and of the blank, and writes
// them lane-major (48 bytes per lane and frame: ds_write_b128 x 3, bank-conflict-free at that stride) into the two-deep ring the
// lattice wave reads.  It runs a chunk ahead of the lattice wave and waits only when both ring slots are still unread.
template <int DIR>
__device__ __forceinline__ void lattice_helper(const float* __restrict__ pr, const int32_t* __restrict__ lab, const HelpShared hs,
                                               double* __restrict__ raw, int lane, int L, int T, int k, int blank, float eps) {
    constexpr int ZERO_SLOT = 63;
    int col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pos = 4 * lane + i;
        col[i] = pos < L ? lab[pos] : ZERO_SLOT;  // dead label slots: u = 0
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[j * 64 + lane] = 0.0;  // (slot 63 of every frame stays zero: k <= 63 classes)
    int st_idx[4], ld_jf[4], ld_c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int idx = lane + 64 * r;
        const int jf = idx / k;
        ld_c[r] = idx - jf * k;
        ld_jf[r] = jf < 8 ? jf : 7;
        st_idx[r] = jf < 8 ? jf * 64 + ld_c[r] : 8 * 64 + lane;
    }
    const int tstart = DIR == 0 ? 0 : T - 1;
    const int tstep = DIR == 0 ? 1 : -1;
    const float* chunk0 = pr + (long)tstart * k;
    const int kstep = tstep * k;
    float e4[4];
    auto fetch = [&](int base) {
#pragma unroll
        for (int r = 0; r < 4; ++r) e4[r] = chunk0[min(base + ld_jf[r], T - 1) * kstep + ld_c[r]];
    };
    const int nchunks = (T + 7) / 8;
    fetch(0);
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) raw[st_idx[r]] = (double)(e4[r] + eps);
        fetch(8 * (c + 1));  // (the next chunk's loads fly during this chunk's gather; past the end: the last frame again)
        // ring slot c & 1 was last used by chunk c - 2: free once the lattice wave has released it
        for (int spins = 0; spins < (1 << 22); ++spins) {
            if (__hip_atomic_load(hs.consumed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= c - 1) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        double* dst = hs.gath + ((long)((c & 1) * 8) * 64 + lane) * 6;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double* row = raw + j * 64;
            const double2 q01 = make_double2(row[col[0]], row[col[1]]);
            const double2 q23 = make_double2(row[col[2]], row[col[3]]);
            const double2 qb = make_double2(row[blank], 0.0);
            double2* d = (double2*)(dst + (long)j * 64 * 6);
            d[0] = q01;
            d[1] = q23;
            d[2] = qb;
        }
        asm volatile("" ::: "memory");
        __hip_atomic_store(hs.progress, c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (all lanes, same value)
    }
}

// per-class position lists for the gradient kernel, by ONE wave (the y = 2 work-groups of the pair / helped lattice kernels)
__device__ __forceinline__ void build_class_lists_wave(int* s_lab, int lane, const int32_t* __restrict__ labels,
                                                       int32_t* __restrict__ cls, int32_t* __restrict__ tickets, int b, int L,
                                                       int grad_wgs, int k, int l_max) {
    for (int i = lane; i < grad_wgs; i += 64) tickets[(long)b * grad_wgs + i] = 0;
    int* s_start = s_lab + l_max;
    int32_t* pos_out = cls + (long)b * (l_max + k + 1);
    int32_t* start_out = pos_out + l_max;
    for (int i = lane; i < L; i += 64) s_lab[i] = labels[(long)b * l_max + i];
    for (int i = lane; i <= k; i += 64) s_start[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int ranks[8];
    int nmine = 0;
    for (int i = lane; i < L; i += 64) {
        const int c = s_lab[i];
        int r = 0;
        for (int j = 0; j < i; ++j) r += (s_lab[j] == c);
        ranks[nmine++] = r;
        atomicAdd(&s_start[c + 1], 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0)
        for (int c = 0; c < k; ++c) s_start[c + 1] += s_start[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    nmine = 0;
    for (int i = lane; i < L; i += 64) pos_out[s_start[s_lab[i]] + ranks[nmine++]] = i;
    for (int i = lane; i <= k; i += 64) start_out[i] = s_start[i];
}

// dynamic LDS of ctc_lattice_helped_kernel: fin 32 | csum 16 | counters 16 | raw (8 * 64 + 64) doubles | gath 2 * 8 * 64 * 6 doubles
constexpr int HELP_RAW = 8 * 64 + 64;
constexpr size_t HELP_LDS_BYTES = 64 + (size_t)HELP_RAW * sizeof(double) + (size_t)2 * 8 * 64 * 6 * sizeof(double);
__host__ inline size_t helped_lds_bytes(int l_max, int k) {
    const size_t lists = (size_t)(l_max + k + 1) * sizeof(int);
    return HELP_LDS_BYTES > lists ? HELP_LDS_BYTES : lists;
}

// grid (B, 3), 128 threads: y = 0 alpha, 1 beta -- wave 0 the lattice (wave_lattice_run<.., true>), wave 1 its helper; y = 2 the
// class lists (wave 0)
__global__ __launch_bounds__(128) void ctc_lattice_helped_kernel(const float* __restrict__ probs, const float* __restrict__ logq,
                                                                 const int32_t* __restrict__ labels,
                                                                 const int32_t* __restrict__ label_len,
                                                                 const int32_t* __restrict__ input_len,
                                                                 uint32_t* __restrict__ alpha, uint32_t* __restrict__ beta,
                                                                 uint32_t* __restrict__ dump, int32_t* __restrict__ ea,
                                                                 int32_t* __restrict__ eb, float* __restrict__ logz2,
                                                                 int32_t* __restrict__ zint, float* __restrict__ loss,
                                                                 int32_t* __restrict__ cls, int32_t* __restrict__ flags,
                                                                 int32_t* __restrict__ tickets, int grad_wgs, int t_out, int k,
                                                                 int l_max, int blank, float eps) {
    extern __shared__ __attribute__((aligned(16))) char hl_lds[];
    const int b = blockIdx.x;
    const int dir = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = label_len[b];
    if (dir == 2) {
        if (wv == 0) build_class_lists_wave((int*)hl_lds, lane, labels, cls, tickets, b, L, grad_wgs, k, l_max);
        return;
    }
    const int S = 2 * L + 1;
    int T = input_len[b];
    if (T > t_out) T = t_out;
    if (T <= 0) {
        if (dir == 0 && threadIdx.x == 0) {
            loss[b] = INFINITY;
            logz2[b] = 0.f;
            zint[b] = 0;
            flags[b] = 0;
        }
        return;
    }
    double* fin = (double*)hl_lds;
    int* fin_e = (int*)(fin + 2);
    float* csum_s = (float*)(hl_lds + 32);
    int* repeats_s = (int*)(hl_lds + 36);
    HelpShared hs;
    hs.progress = (int*)(hl_lds + 48);
    hs.consumed = (int*)(hl_lds + 52);
    double* raw = (double*)(hl_lds + 64);
    hs.gath = raw + HELP_RAW;
    if (threadIdx.x == 0) {
        *hs.progress = 0;
        *hs.consumed = 0;
    }
    __syncthreads();
    const int32_t* lab = labels + (long)b * l_max;
    const float* pr = probs + (long)b * t_out * k;
    if (wv == 1) {
        if (dir == 0)
            lattice_helper<0>(pr, lab, hs, raw, lane, L, T, k, blank, eps);
        else
            lattice_helper<1>(pr, lab, hs, raw, lane, L, T, k, blank, eps);
        if (dir == 0) {
            // sum over the scored frames of ln c_t and the repeat count (see ctc_lattice_wave_kernel): the helper has the slack
            float csum = 0.f;
            for (int t = lane; t < T; t += 64)
                csum += logq[((long)b * t_out + t) * k + blank] - logf(pr[(long)t * k + blank] + eps);
            csum = wave_sum(csum);
            float repeats_f = 0.f;
            for (int i = lane + 1; i < L; i += 64) repeats_f += lab[i] == lab[i - 1] ? 1.f : 0.f;
            const int repeats = (int)wave_sum(repeats_f);
            if (lane == 0) {
                *csum_s = csum;
                *repeats_s = repeats;
            }
        }
        __syncthreads();
        return;
    }
    double a[WNS];
    int E;
    if (dir == 0) {
        wave_lattice_run<0, double, true>(pr, lab, alpha + (long)b * t_out * (64 * WNS), dump + (long)(2 * b) * (64 * WNS),
                                          ea + (long)b * (t_out / 16 + 1) * 64, lane, L, S, T, k, blank, eps, a, &E, hs);
#pragma unroll
        for (int i = 0; i < WNS; ++i) {
            if (WNS * lane + i == S - 1) {
                fin[0] = a[i];
                fin_e[0] = E;
            }
            if (WNS * lane + i == S - 2) {
                fin[1] = a[i];
                fin_e[1] = E;
            }
        }
        __syncthreads();
        if (lane == 0) {
            const float csum = *csum_s;
            const int repeats = *repeats_s;
            const double z1 = fin[0], z2 = S >= 2 ? fin[1] : 0.0;
            const int e1 = fin_e[0], e2 = S >= 2 ? fin_e[1] : e1;
            const int ez = (z2 > 0.0 && (z1 == 0.0 || e2 > e1)) ? e2 : e1;
            const double z = ldexp(z1, max(e1 - ez, -2000)) + ldexp(z2, max(e2 - ez, -2000));
            int xz = 0;
            const double mz = frexp(z > 0.0 ? z : 1.0, &xz);
            const float frac = __builtin_amdgcn_logf((float)mz);
            logz2[b] = frac;
            zint[b] = xz + ez;
            loss[b] = z > 0.0 ? (float)(-((double)(xz + ez) + (double)frac) * 0.6931471805599453 - (double)csum) : INFINITY;
            flags[b] = (z > 0.0 && z < INFINITY) ? 0 : (z == 0.0 && L + repeats > T ? 0 : 1);
        }
    } else {
        wave_lattice_run<1, double, true>(pr, lab, beta + (long)b * t_out * (64 * WNS), dump + (long)(2 * b + 1) * (64 * WNS),
                                          eb + (long)b * (t_out / 16 + 1) * 64, lane, L, S, T, k, blank, eps, a, &E, hs);
        __syncthreads();
    }
}

// dynamic LDS of ctc_lattice_pair_kernel (bytes): fin 32 | progress 32 | ebox | mbox | emis   (dir 2: the list builder's ints)
__host__ __device__ inline size_t pair_lds_ebox(int t_out) { return (size_t)2 * (t_out / 16 + 2) * sizeof(int); }
__host__ __device__ inline size_t pair_lds_mbox(int t_out) { return (size_t)2 * (t_out + 2) * sizeof(double); }
__host__ inline size_t pair_lds_bytes(int t_out, int l_max, int k) {
    size_t lattice = 64 + ((pair_lds_ebox(t_out) + 15) / 16) * 16 + pair_lds_mbox(t_out) + (size_t)2 * PLANES * sizeof(double) +
                     (size_t)4 * PAIR_EMIS * sizeof(double);
    const size_t lists = (size_t)(l_max + k + 1) * sizeof(int);
    return lattice > lists ? lattice : lists;
}

// grid (B, 3), 128 threads: y = 0 alpha, 1 beta (two waves each, pair_lattice_run), 2 the per-class position lists (wave 0)
__global__ __launch_bounds__(128) void ctc_lattice_pair_kernel(const float* __restrict__ probs, const float* __restrict__ logq,
                                                               const int32_t* __restrict__ labels,
                                                               const int32_t* __restrict__ label_len,
                                                               const int32_t* __restrict__ input_len,
                                                               uint32_t* __restrict__ alpha, uint32_t* __restrict__ beta,
                                                               uint32_t* __restrict__ dump, int32_t* __restrict__ ea,
                                                               int32_t* __restrict__ eb, float* __restrict__ logz2,
                                                               int32_t* __restrict__ zint, float* __restrict__ loss,
                                                               int32_t* __restrict__ cls, int32_t* __restrict__ flags,
                                                               int32_t* __restrict__ tickets, int grad_wgs, int t_out, int k,
                                                               int l_max, int blank, float eps) {
    extern __shared__ __attribute__((aligned(16))) char pl_lds[];
    const int b = blockIdx.x;
    const int dir = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = label_len[b];
    if (dir == 2) {
        if (wv == 0) build_class_lists_wave((int*)pl_lds, lane, labels, cls, tickets, b, L, grad_wgs, k, l_max);
        return;
    }
    const int S = 2 * L + 1;
    int T = input_len[b];
    if (T > t_out) T = t_out;
    if (T <= 0) {
        if (dir == 0 && threadIdx.x == 0) {
            loss[b] = INFINITY;
            logz2[b] = 0.f;
            zint[b] = 0;
            flags[b] = 0;
        }
        return;
    }
    double* fin = (double*)pl_lds;
    int* fin_e = (int*)(fin + 2);
    PairShared sh;
    sh.progress = (int*)(pl_lds + 32);
    sh.ebox = (int*)(pl_lds + 64);
    sh.mbox = (double*)(pl_lds + 64 + ((pair_lds_ebox(t_out) + 15) / 16) * 16);
    sh.mdump = (double*)((char*)sh.mbox + pair_lds_mbox(t_out));
    sh.emis = sh.mdump + 2 * PLANES;
    if (threadIdx.x == 0) *sh.progress = 0;
    __syncthreads();
    const int32_t* lab = labels + (long)b * l_max;
    const float* pr = probs + (long)b * t_out * k;
    double a[PNS];
    int E;
    const int gl = wv * 64 + lane;
    if (dir == 0) {
        // sum over the scored frames of ln c_t and the repeat count (see ctc_lattice_wave_kernel): by the DOWNSTREAM wave, which
        // has half a block to wait for its first boundary values anyway
        float csum = 0.f;
        int repeats = 0;
        if (wv == 1) {
            for (int t = lane; t < T; t += 64)
                csum += logq[((long)b * t_out + t) * k + blank] - logf(pr[(long)t * k + blank] + eps);
            csum = wave_sum(csum);
            float repeats_f = 0.f;
            for (int i = lane + 1; i < L; i += 64) repeats_f += lab[i] == lab[i - 1] ? 1.f : 0.f;
            repeats = (int)wave_sum(repeats_f);
        }
        pair_lattice_run<0>(pr, lab, alpha + (long)b * t_out * 512, dump + (long)(2 * b) * 512,
                            ea + (long)b * (t_out / 16 + 1) * PLANES, sh, wv, lane, L, S, T, k, blank, eps, a, &E);
#pragma unroll
        for (int i = 0; i < PNS; ++i) {
            if (PNS * gl + i == S - 1) {
                fin[0] = a[i];
                fin_e[0] = E;
            }
            if (PNS * gl + i == S - 2) {
                fin[1] = a[i];
                fin_e[1] = E;
            }
        }
        __syncthreads();
        if (wv == 1 && lane == 0) {
            const double z1 = fin[0], z2 = S >= 2 ? fin[1] : 0.0;
            const int e1 = fin_e[0], e2 = S >= 2 ? fin_e[1] : e1;
            const int ez = (z2 > 0.0 && (z1 == 0.0 || e2 > e1)) ? e2 : e1;
            const double z = ldexp(z1, max(e1 - ez, -2000)) + ldexp(z2, max(e2 - ez, -2000));
            int xz = 0;
            const double mz = frexp(z > 0.0 ? z : 1.0, &xz);
            const float frac = __builtin_amdgcn_logf((float)mz);
            logz2[b] = frac;
            zint[b] = xz + ez;
            loss[b] = z > 0.0 ? (float)(-((double)(xz + ez) + (double)frac) * 0.6931471805599453 - (double)csum) : INFINITY;
            flags[b] = (z > 0.0 && z < INFINITY) ? 0 : (z == 0.0 && L + repeats > T ? 0 : 1);
        }
    } else {
        pair_lattice_run<1>(pr, lab, beta + (long)b * t_out * 512, dump + (long)(2 * b + 1) * 512,
                            eb + (long)b * (t_out / 16 + 1) * PLANES, sh, wv, lane, L, S, T, k, blank, eps, a, &E);
    }
}

__global__ __launch_bounds__(256) void greedy_decode_kernel(const float* __restrict__ probs,
                                                            const int32_t* __restrict__ input_len,
                                                            int32_t* __restrict__ out, int32_t* __restrict__ out_len,
                                                            int32_t* __restrict__ frame_argmax, int t_out, int k,
                                                            int blank) {
    extern __shared__ int s_idx[];  // t_out ints + 256 scan slots
    int* s_scan = s_idx + t_out;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    int T = input_len[b];
    if (T > t_out) T = t_out;
    if (T < 0) T = 0;
    for (int t = tid; t < t_out; t += 256) {
        int best = -1;
        if (t < T) {
            const float* p = probs + ((long)b * t_out + t) * k;
            float bv = p[0];
            best = 0;
            for (int i = 1; i < k; ++i) {
                const float v = p[i];
                if (v > bv) {
                    bv = v;
                    best = i;
                }
            }
        }
        s_idx[t] = best;
        if (frame_argmax) frame_argmax[(long)b * t_out + t] = best;
    }
    __syncthreads();
    // each thread owns a contiguous slice of frames
    const int per = (T + 255) / 256;
    const int t0 = tid * per;
    int t1 = t0 + per;
    if (t1 > T) t1 = T;
    int cnt = 0;
    for (int t = t0; t < t1; ++t) {
        const int c = s_idx[t];
        const int pv = t > 0 ? s_idx[t - 1] : -1;
        cnt += (c != blank && c != pv);
    }
    s_scan[tid] = cnt;
    __syncthreads();
    // inclusive Hillis-Steele scan over 256 entries
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int pos = s_scan[tid] - cnt;
    const int total = s_scan[255];
    int32_t* o = out + (long)b * t_out;
    for (int t = t0; t < t1; ++t) {
        const int c = s_idx[t];
        const int pv = t > 0 ? s_idx[t - 1] : -1;
        if (c != blank && c != pv) o[pos++] = c;
    }
    for (int t = total + tid; t < t_out; t += 256) o[t] = -1;
    if (tid == 0) out_len[b] = total;
}

__host__ int lattice_sp(int l_max) { return ((2 * l_max + 1) + 63) / 64 * 64; }

// which lattice sl_ctc_loss_grad runs (sl_ctc_select): 0 = automatic -- when the labels fit (2 * l_max + 1 <= 512, k <= 63)
// the probability-domain wave lattice in doubles with the log-domain repair pass behind it, otherwise the log-domain
// lattice (per call at 32 x 500 frames: 108 vs 132 us; at 8 x 4000 frames 556 vs 912); 1 = log-domain lattice only; 2 =
// double wave lattice without the repair pass (measurement); 3 = double wave lattice, then every utterance redone by
// the repair pass (tests); 4 = double wave lattice + repair; 5 / 6 / 7 = the FLOAT wave lattice with repair / without /
// with forced repair (measurement: 87 / 463 us, but see WaveReal)
int g_ctc_variant = 0;
#ifndef SL_CTC_DEFAULT_WAVE
// sl_ctc_select(0): 10 = the lattice wave with its helper wave (round 6: 92.4 -> 88.0 us per call at 32 x 500, 502 -> 471 us at
// 8 x 4000, bit-identical results), 4 = the lone wave, 8 = the wave pair (A/B builds: -DSL_CTC_DEFAULT_WAVE=4)
#define SL_CTC_DEFAULT_WAVE 10
#endif

struct CtcLayout {
    size_t log_alpha, log_beta, cls, lin_alpha, lin_beta, dump, ea, eb, logz2, zint, flags, tickets, total;
};
__host__ CtcLayout ctc_layout(int batch, int t_out, int l_max) {
    CtcLayout w;
    const size_t rows = (size_t)batch * t_out;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) / 256 * 256;
        return at;
    };
    // log-domain rows: floats for the labels the wave lattice does not take (variant 1), DOUBLES for the repair pass (round 6)
    w.log_alpha = take(rows * lattice_sp(l_max) * sizeof(double));
    w.log_beta = take(rows * lattice_sp(l_max) * sizeof(double));
    w.cls = take((size_t)batch * (l_max + 65) * sizeof(int32_t));
    const bool wave = 2 * l_max + 1 <= 64 * WNS;
    w.lin_alpha = take(wave ? rows * 64 * WNS * sizeof(uint32_t) : 0);  // a float, or the high word of a double, per state
    w.lin_beta = take(wave ? rows * 64 * WNS * sizeof(uint32_t) : 0);
    w.dump = take(wave ? (size_t)2 * batch * 64 * WNS * sizeof(uint32_t) : 0);
    // one exponent per lane and block of 16 (double) / 8 (float) steps; the pair lattice: 128 per block of 16
    const size_t eblocks = (size_t)batch * (t_out / 8 + 2) * 64;
    w.ea = take(wave ? eblocks * sizeof(int32_t) : 0);
    w.eb = take(wave ? eblocks * sizeof(int32_t) : 0);
    w.logz2 = take((size_t)batch * sizeof(float));
    w.zint = take((size_t)batch * sizeof(int32_t));
    w.flags = take((size_t)batch * sizeof(int32_t));
    w.tickets = take((size_t)batch * ((t_out + 7) / 8) * sizeof(int32_t));  // one done slot per gradient work-group
    w.total = off;
    return w;
}

}  // namespace

extern "C" int sl_ctc_select(int variant) {
    SL_CHECK_ARG(variant >= 0 && variant <= 11, "sl_ctc_select: variant %d outside 0..11", variant);
    g_ctc_variant = variant;
    return SL_OK;
}

extern "C" int sl_softmax_logq(const float* logits, float* probs, float* logq, int batch, int t_out, int k,
                               int logit_stride, int64_t logit_batch_stride, float eps, void* stream) {
    SL_CHECK_ARG(batch > 0 && t_out > 0 && k > 0 && logit_stride >= k, "sl_softmax_logq: bad sizes");
    const long frames = (long)batch * t_out;
    SL_CHECK_ARG(k <= 64, "sl_softmax_logq: at most 64 classes (one lane per class)");
    hipLaunchKernelGGL(softmax_logq_kernel, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       logits, probs, logq, frames, t_out, k, logit_stride, (long)logit_batch_stride, eps);
    return sl_check_launch("sl_softmax_logq");
}

extern "C" size_t sl_ctc_workspace_bytes(int batch, int t_out, int l_max) {
    if (batch <= 0 || t_out <= 0 || l_max < 0) return 0;
    // log-domain alpha + beta lattices and the class position lists (pos[l_max] | start[k + 1], k <= 64) per utterance;
    // when the labels fit the wave lattice: its double lattices, per-frame exponents, log2 Z and repair flags as well
    // MONOTONIC in l_max: labels beyond the wave lattice's 255 graphemes drop its double lattices from the layout, so the
    // largest need below l_max is that of the longest label the wave lattice still takes -- a workspace sized for l_max
    // serves every batch with shorter labels (a buffer set sees both in ordinary variable-length training)
    size_t total = ctc_layout(batch, t_out, l_max).total;
    const int wave_max = (64 * WNS - 1) / 2;
    if (l_max > wave_max) {
        const size_t t2 = ctc_layout(batch, t_out, wave_max).total;
        if (t2 > total) total = t2;
    }
    return total;
}

extern "C" int sl_ctc_loss_grad(const float* probs, const float* logq, const int32_t* labels, const int32_t* label_len,
                                const int32_t* input_len, float* loss, void* dlogits, int batch, int t_out, int k,
                                int l_max, int g_row0, int g_row_stride, int64_t g_batch_stride, int dtype, float eps,
                                float grad_scale, void* workspace, size_t workspace_bytes, void* stream) {
    SL_CHECK_ARG(batch > 0 && t_out > 0 && k > 1 && k <= 64, "sl_ctc_loss_grad: need batch,t_out > 0 and 1 < k <= 64");
    SL_CHECK_ARG(l_max >= 1, "sl_ctc_loss_grad: l_max must be >= 1 (pad the label batch to width 1 for empty labels)");
    const int sp = lattice_sp(l_max);
    if (2 * l_max + 1 > 1024) {
        sl_set_error("sl_ctc_loss_grad: label length %d > 511 unsupported (one lattice state per thread)", l_max);
        return SL_ERR_UNSUPPORTED;
    }
    const CtcLayout w = ctc_layout(batch, t_out, l_max);
    if (workspace_bytes < w.total) {
        sl_set_error("sl_ctc_loss_grad: workspace too small");
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)workspace;
    float* alpha = (float*)(base + w.log_alpha);
    float* beta = (float*)(base + w.log_beta);
    int32_t* cls = (int32_t*)(base + w.cls);
    int32_t* flags = (int32_t*)(base + w.flags);
    // which lattice: see sl_ctc_select
    const bool fits = 2 * l_max + 1 <= 64 * WNS && k <= 63;
    int v = g_ctc_variant;
    // 8 (round 6): the double lattice on a PAIR of waves per direction (pair_lattice_run); its LDS mailbox holds a slot per
    // frame, so very long utterances (T' > 8000) stay on the single wave
    const bool pair_fits = fits && pair_lds_bytes(t_out, l_max, k) <= 150 * 1024;
    if (v == 0) v = fits ? ((SL_CTC_DEFAULT_WAVE != 8 || pair_fits) ? SL_CTC_DEFAULT_WAVE : 4) : 1;
    if (v != 1 && !fits) v = 1;
    if (v == 8 && !pair_fits) v = 4;
    if (v == 9 && !pair_fits) v = 2;
    if ((v == 10 || v == 11) && !fits) v = 1;
    const bool wave = v != 1;
    const bool pair = v == 8 || v == 9;
    const bool helped = v == 10 || v == 11;  // 10 / 11 (round 6): the lone lattice wave + a helper wave that gathers its emissions
    const bool wave_f32 = v >= 5 && v <= 7;
    const bool repair = wave && v != 2 && v != 6 && v != 9 && v != 11;
    const bool force_repair = v == 3 || v == 7;
    const int frames_per_wg = 8;  // two frames per wave: 16000 frames -> 8000 waves in flight
    const size_t lds2 = (size_t)(2 * l_max + (k + 1)) * sizeof(int) + (size_t)(4 * 64 + 4 * l_max) * sizeof(float);
    const dim3 grid((t_out + frames_per_wg - 1) / frames_per_wg, batch);
    const int out_f32 = dtype == SL_F32 ? 1 : 0;
    int rc;
    if (wave) {
        void* la = base + w.lin_alpha;
        void* lb = base + w.lin_beta;
        int32_t* ea = (int32_t*)(base + w.ea);
        int32_t* eb = (int32_t*)(base + w.eb);
        float* logz2 = (float*)(base + w.logz2);
        int32_t* zint = (int32_t*)(base + w.zint);
        int32_t* tickets = (int32_t*)(base + w.tickets);
        size_t lds = (size_t)(l_max + k + 1) * sizeof(int);
        if (lds < 2 * sizeof(double) + 2 * sizeof(int)) lds = 2 * sizeof(double) + 2 * sizeof(int);
        if (helped) {
            const size_t hlds = helped_lds_bytes(l_max, k);
            static bool attr_set_h = false;
            if (!attr_set_h) {
                (void)hipFuncSetAttribute((const void*)ctc_lattice_helped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024);
                attr_set_h = true;
            }
            hipLaunchKernelGGL(ctc_lattice_helped_kernel, dim3(batch, 3), dim3(128), hlds, s, probs, logq, labels, label_len,
                               input_len, (uint32_t*)la, (uint32_t*)lb, (uint32_t*)(base + w.dump), ea, eb, logz2, zint, loss,
                               cls, flags, tickets, (int)grid.x, t_out, k, l_max, k - 1, eps);
        } else if (pair) {
            const size_t plds = pair_lds_bytes(t_out, l_max, k);
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)ctc_lattice_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024);
                attr_set = true;
            }
            hipLaunchKernelGGL(ctc_lattice_pair_kernel, dim3(batch, 3), dim3(128), plds, s, probs, logq, labels, label_len,
                               input_len, (uint32_t*)la, (uint32_t*)lb, (uint32_t*)(base + w.dump), ea, eb, logz2, zint, loss,
                               cls, flags, tickets, (int)grid.x, t_out, k, l_max, k - 1, eps);
        } else if (wave_f32)
            hipLaunchKernelGGL(ctc_lattice_wave_kernel<float>, dim3(batch, 3), dim3(64), lds, s, probs, logq, labels,
                               label_len, input_len, (uint32_t*)la, (uint32_t*)lb, (uint32_t*)(base + w.dump), ea, eb, logz2, zint,
                               loss, cls, flags, tickets, (int)grid.x, t_out, k, l_max, k - 1, eps);
        else
            hipLaunchKernelGGL(ctc_lattice_wave_kernel<double>, dim3(batch, 3), dim3(64), lds, s, probs, logq, labels,
                               label_len, input_len, (uint32_t*)la, (uint32_t*)lb, (uint32_t*)(base + w.dump), ea, eb, logz2, zint,
                               loss, cls, flags, tickets, (int)grid.x, t_out, k, l_max, k - 1, eps);
        rc = sl_check_launch("sl_ctc_loss_grad(wave lattice)");
        if (rc != SL_OK) return rc;
        if (force_repair) {  // tests: every utterance is redone by the repair pass inside the gradient kernel
            rc = (int)hipMemsetAsync(flags, 1, (size_t)batch * sizeof(int32_t), s);
            if (rc != 0) return SL_ERR_LAUNCH_FAILED;
        }
        // rows are 512 values wide; the kernel reads 8 columns per lane.  The repair pass is the tail of this launch
        // (ctc_grad_kernel: the work-group that finishes an utterance last redoes it if it was flagged).
        int32_t* rep_tickets = repair ? tickets : nullptr;
        if (pair)
            hipLaunchKernelGGL((ctc_grad_kernel<8, 3>), grid, dim3(256), lds2, s, probs, logq, labels, label_len,
                               input_len, (const void*)la, (const void*)lb, ea, eb, logz2, zint, loss, cls, dlogits, t_out, k,
                               l_max, 64 * WNS, k - 1, frames_per_wg, g_row0, g_row_stride, (long)g_batch_stride, out_f32,
                               eps, grad_scale, flags, rep_tickets, (double*)alpha, (double*)beta, sp);
        else if (wave_f32)
            hipLaunchKernelGGL((ctc_grad_kernel<8, 2>), grid, dim3(256), lds2, s, probs, logq, labels, label_len,
                               input_len, (const void*)la, (const void*)lb, ea, eb, logz2, zint, loss, cls, dlogits, t_out, k,
                               l_max, 64 * WNS, k - 1, frames_per_wg, g_row0, g_row_stride, (long)g_batch_stride, out_f32,
                               eps, grad_scale, flags, rep_tickets, (double*)alpha, (double*)beta, sp);
        else
            hipLaunchKernelGGL((ctc_grad_kernel<8, 1>), grid, dim3(256), lds2, s, probs, logq, labels, label_len,
                               input_len, (const void*)la, (const void*)lb, ea, eb, logz2, zint, loss, cls, dlogits, t_out, k,
                               l_max, 64 * WNS, k - 1, frames_per_wg, g_row0, g_row_stride, (long)g_batch_stride, out_f32,
                               eps, grad_scale, flags, rep_tickets, (double*)alpha, (double*)beta, sp);
        return sl_check_launch("sl_ctc_loss_grad(grad)");
    }
    // labels beyond the wave lattice: the log-domain lattice for every utterance
    const int threads = sp;  // multiple of 64, >= S
    size_t lds = 2 * (threads + 4) * sizeof(float);
    const size_t lds_lists = (size_t)(l_max + k + 1) * sizeof(int);
    if (lds < lds_lists) lds = lds_lists;
    hipLaunchKernelGGL(ctc_lattice_kernel, dim3(batch, 3), dim3(threads), lds, s, logq, labels, label_len, input_len,
                       alpha, beta, loss, cls, t_out, k, l_max, sp, k - 1);
    rc = sl_check_launch("sl_ctc_loss_grad(lattice)");
    if (rc != SL_OK) return rc;
#define SL_CTC_GRAD(NJ_)                                                                                              \
    hipLaunchKernelGGL((ctc_grad_kernel<NJ_, 0>), grid, dim3(256), lds2, s, probs, logq, labels, label_len,        \
                       input_len, (const void*)alpha, (const void*)beta, nullptr, nullptr, nullptr, nullptr, loss, cls,   \
                       dlogits,                                                                                        \
                       t_out, k, l_max, sp, k - 1, frames_per_wg, g_row0, g_row_stride, (long)g_batch_stride, out_f32,  \
                       eps, grad_scale, nullptr, nullptr, nullptr, nullptr, 0)
    if (sp <= 256) {
        SL_CTC_GRAD(4);
    } else if (sp <= 512) {
        SL_CTC_GRAD(8);
    } else {
        SL_CTC_GRAD(16);
    }
#undef SL_CTC_GRAD
    return sl_check_launch("sl_ctc_loss_grad(grad)");
}

extern "C" int sl_greedy_decode(const float* probs, const int32_t* input_len, int32_t* out, int32_t* out_len,
                                int32_t* frame_argmax, int batch, int t_out, int k, int blank, void* stream) {
    SL_CHECK_ARG(batch > 0 && t_out > 0 && k > 0, "sl_greedy_decode: bad sizes");
    const size_t lds = (size_t)(t_out + 256) * sizeof(int);
    if (lds > 150 * 1024) {
        sl_set_error("sl_greedy_decode: t_out %d too large for the LDS-resident decoder", t_out);
        return SL_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)greedy_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(greedy_decode_kernel, dim3(batch), dim3(256), lds, (hipStream_t)stream, probs, input_len, out,
                       out_len, frame_argmax, t_out, k, blank);
    return sl_check_launch("sl_greedy_decode");
}
