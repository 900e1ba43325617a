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
//   ctc_grad_kernel     : one wave per frame.  occupancy gamma_t(k) = sum_{s: l'_s = k} exp(alpha+beta-logq-logP) is
//                         reduced in a FIXED order (blank: wave butterfly; graphemes: per-class position lists built
//                         once per work-group) -> deterministic.  Chains through log(p+eps) and the output softmax and
//                         writes d(loss)/d(logits) straight into the halo'd gradient tensor of output_conv.
//   greedy_decode_kernel: one work-group per utterance: argmax (first max wins), merge repeats, drop blank, compact.
#include "common.h"

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

// one wave per frame; work-group = 4 waves x FRAMES_PER_WAVE frames.  NJ = SP / 64 lattice columns per lane: all of a
// frame's alpha / beta loads are issued before the first one is used (one HBM/L2 round trip per frame instead of NJ).
template <int NJ>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ probs, const float* __restrict__ logq,
                                                       const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_len,
                                                       const int32_t* __restrict__ input_len,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const float* __restrict__ loss, const int32_t* __restrict__ cls,
                                                       void* __restrict__ dlogits, int t_out, int k, int l_max, int sp,
                                                       int blank, int frames_per_wg, int g_row0, int g_rs, long g_bs,
                                                       int out_f32, float eps, float grad_scale) {
    // LDS: labels[l_max] | class_pos[l_max] | class_start[k+1] | lq[4][64] | gamma[4][l_max]
    extern __shared__ int lds_i[];
    int* s_lab = lds_i;
    int* s_pos = s_lab + l_max;
    int* s_start = s_pos + l_max;
    float* s_lq = (float*)(s_start + (k + 1));
    float* s_gam = s_lq + 4 * 64;
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
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

    const float nll = loss[b];
    const bool feasible = nll < INFINITY;
    const float log_p = -nll * LOG2E;  // lattice units are log2
    float* gam = s_gam + wave * l_max;
    float* wlq = s_lq + wave * 64;
    const int t_begin = blockIdx.x * frames_per_wg;
    for (int tt = wave; tt < frames_per_wg; tt += 4) {
        const int t = t_begin + tt;
        if (t >= t_out) break;
        const long fidx = (long)b * t_out + t;
        float dz = 0.f;
        if (t < T) {
            const float lqv = lane < k ? logq[fidx * k + lane] : 0.f;
            const float pk = lane < k ? probs[fidx * k + lane] : 0.f;
            float occ = 0.f;
            if (feasible) {
                const float* al = alpha + fidx * sp;
                const float* be = beta + fidx * sp;
                float av[NJ], bv[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {  // rows are sp (a multiple of 64) wide: a wave-uniform bound
                    const bool in = 64 * j < sp;
                    av[j] = in ? al[lane + 64 * j] : -INFINITY;
                    bv[j] = in ? be[lane + 64 * j] : -INFINITY;
                }
                wlq[lane] = lqv * LOG2E;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const float lq_blank = wlq[blank];
                // blank states (even s) -> butterfly sum; grapheme states (odd s) -> gamma[] in LDS
                float blank_part = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int s = lane + 64 * j;
                    if (s < S) {
                        const float ab = av[j] + bv[j];
                        if (s & 1) {
                            const int pos = s >> 1;
                            const float lg = ab - wlq[s_lab[pos]] - log_p;
                            gam[pos] = (ab == -INFINITY) ? 0.f : exp2f(lg);
                        } else {
                            const float lg = ab - lq_blank - log_p;
                            blank_part += (ab == -INFINITY) ? 0.f : exp2f(lg);
                        }
                    }
                }
                blank_part = wave_sum(blank_part);
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
        }
        if (lane < k) {
            const long gi = (long)b * g_bs + (long)(g_row0 + t) * g_rs + lane;
            if (out_f32)
                ((float*)dlogits)[gi] = dz;
            else
                ((unsigned short*)dlogits)[gi] = f32_to_bf16_bits(dz);
        }
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

}  // namespace

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
    // alpha + beta lattices, then per utterance the class position lists: pos[l_max] | start[k + 1], k <= 64
    return (size_t)2 * batch * t_out * lattice_sp(l_max) * sizeof(float) + (size_t)batch * (l_max + 65) * sizeof(int32_t);
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
    if (workspace_bytes < sl_ctc_workspace_bytes(batch, t_out, l_max)) {
        sl_set_error("sl_ctc_loss_grad: workspace too small");
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t s = (hipStream_t)stream;
    float* alpha = (float*)workspace;
    float* beta = alpha + (size_t)batch * t_out * sp;
    int32_t* cls = (int32_t*)(beta + (size_t)batch * t_out * sp);
    const int threads = sp;  // multiple of 64, >= S
    size_t lds = 2 * (threads + 4) * sizeof(float);
    const size_t lds_lists = (size_t)(l_max + k + 1) * sizeof(int);
    if (lds < lds_lists) lds = lds_lists;
    hipLaunchKernelGGL(ctc_lattice_kernel, dim3(batch, 3), dim3(threads), lds, s, logq, labels, label_len, input_len,
                       alpha, beta, loss, cls, t_out, k, l_max, sp, k - 1);
    int rc = sl_check_launch("sl_ctc_loss_grad(lattice)");
    if (rc != SL_OK) return rc;
    const int frames_per_wg = 8;  // two frames per wave: 16000 frames -> 8000 waves in flight
    const size_t lds2 = (size_t)(2 * l_max + (k + 1)) * sizeof(int) + (size_t)(4 * 64 + 4 * l_max) * sizeof(float);
    const dim3 grid((t_out + frames_per_wg - 1) / frames_per_wg, batch);
#define SL_CTC_GRAD(NJ_)                                                                                              \
    hipLaunchKernelGGL(ctc_grad_kernel<NJ_>, grid, dim3(256), lds2, s, probs, logq, labels, label_len, input_len, alpha, \
                       beta, loss, cls, dlogits, t_out, k, l_max, sp, k - 1, frames_per_wg, g_row0, g_row_stride,     \
                       (long)g_batch_stride, dtype == SL_F32 ? 1 : 0, eps, grad_scale)
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
