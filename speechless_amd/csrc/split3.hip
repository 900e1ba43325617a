// split3.hip -- "bf16x3": fp32-class accuracy on the bf16 matrix cores (the fast parity path).
//
// north_star asks for greedy-decoded indices bit-exact against the reference's fp32 CPU path and gradients within 1e-3.  bf16
// storage cannot give that (2^-9 per element); the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, conv_f32.hip) can, at 157 TFLOP/s.
// In between: every fp32 value v is carried as TWO bf16 numbers, hi = bf16(v) and lo = bf16(v - hi) (|v - hi - lo| <= 2^-17 |v|),
// and a product x * w is taken as  x_hi w_hi + x_lo w_hi + x_hi w_lo  with fp32 accumulation (the dropped lo * lo term is
// 2^-18 relative) -- three bf16 MFMAs instead of one, 2.5 PFLOP/s / 3 of peak instead of 157 TFLOP/s.
//
// The MFMA kernels do not change at all.  A tensor row holds three planes of C channels, [P0 = hi | P1 = lo | P2 = hi], and
// the packed weight rows hold [w_hi | w_hi | w_lo]: the ordinary row-shifted NT GEMM (sl_conv1d_nt) over 3 C contraction
// channels IS the three-term product.  Its fp32 result goes to a staging buffer (out_f32) and the kernels of this file
// apply the activation / mask and split it back into planes.  The weight gradient runs sl_conv1d_wgrad twice -- the
// [hi | lo] prefix of x against g_hi (the hh and lh terms in one 2C x C' product) and x_hi against g_lo (hl) -- and
// split3_wgrad_combine adds the three blocks.  All kernels here are elementwise / HBM-bound.
#include "common.h"

namespace {

// The two plane formats.  bf16 pairs ("bf16x3"): 16-17 significand bits at fp32's range.  fp16 pairs ("f16x3", round 6): hi =
// fp16(v), lo = fp16(v - hi) -- 22 bits where |v| >= 2^-3, an absolute 2^-25 below that (lo runs into fp16's denormals, which
// v_mfma_f32_16x16x32_f16 keeps: tools/f16_denorm_probe.hip), range 65504: the engine keeps weights and back-propagated
// gradients scaled by powers of two so that typical magnitudes sit near 1 (speechless_amd/engine_x3.py).
struct PlaneBf16 {
    static __device__ __forceinline__ unsigned short enc(float v) { return f32_to_bf16_bits(v); }
    static __device__ __forceinline__ float dec(unsigned short b) { return bf16_bits_to_f32(b); }
};
struct PlaneF16 {
    static __device__ __forceinline__ unsigned short enc(float v) { return f32_to_f16_bits(v); }
    static __device__ __forceinline__ float dec(unsigned short b) { return f16_bits_to_f32(b); }
};
template <typename F>
__device__ __forceinline__ void split2(float v, unsigned short& hi, unsigned short& lo) {
    hi = F::enc(v);
    lo = F::enc(v - F::dec(hi));
}
template <typename F>
__device__ __forceinline__ float dec_lo16(unsigned int w) { return F::dec((unsigned short)(w & 0xFFFFu)); }
template <typename F>
__device__ __forceinline__ float dec_hi16(unsigned int w) { return F::dec((unsigned short)(w >> 16)); }

// src: fp32 [B][src_rows][C] (row t of utterance b at b * src_bs + t * C), valid rows t < t_out
// dst: planes [B][dst_rows][3 C] at row dst_row0 + t; rows outside [0, t_out) are never written (they stay zero)
// mode 0: none, 1: relu, 2: elu, 3: relu mask, 4: elu mask (mask: a plane tensor of dst's geometry -- the stored activation)
template <int MODE, typename F>
__global__ __launch_bounds__(256) void split3_act_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                         const unsigned short* __restrict__ mask, int t_out, int c,
                                                         long src_bs, int dst_row0, long dst_bs) {
    const int b = blockIdx.y;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;  // 8 channels per thread: 32 bytes in, three 16-byte stores out
    if (i >= t_out * c) return;                           // (t_out * c < 2^31: checked by the host function)
    const int t = i / c, ch = i - t * c;
    const float* sp = src + (long)b * src_bs + i;
    const f32x4 va = *(const f32x4*)sp, vb = *(const f32x4*)(sp + 4);
    const long row = (long)b * dst_bs + (long)(dst_row0 + t) * (3 * c);
    float y[8];
    if (MODE == 3 || MODE == 4) {
        const u32x4 mh = *(const u32x4*)(mask + row + ch);
        u32x4 ml = (u32x4){0u, 0u, 0u, 0u};
        if (MODE == 4) ml = *(const u32x4*)(mask + row + c + ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[2 * j] = dec_lo16<F>(mh[j]) + (MODE == 4 ? dec_lo16<F>(ml[j]) : 0.f);
            y[2 * j + 1] = dec_hi16<F>(mh[j]) + (MODE == 4 ? dec_hi16<F>(ml[j]) : 0.f);
        }
    }
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = j < 4 ? va[j] : vb[j - 4];
        if (MODE == 1) v = fmaxf(v, 0.f);
        if (MODE == 2) v = v > 0.f ? v : expm1f(v);
        if (MODE == 3) v = y[j] > 0.f ? v : 0.f;
        if (MODE == 4) v = y[j] > 0.f ? v : v * (y[j] + 1.f);
        split2<F>(v, hi[j], lo[j]);
    }
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (unsigned)hi[2 * j] | ((unsigned)hi[2 * j + 1] << 16);
        l[j] = (unsigned)lo[2 * j] | ((unsigned)lo[2 * j + 1] << 16);
    }
    *(u32x4*)(dst + row + ch) = h;
    *(u32x4*)(dst + row + c + ch) = l;
    *(u32x4*)(dst + row + 2 * c + ch) = h;
}

// input packing (sl_pack_input for planes): src float[B][t_in][f] -> dst [B][rows][3 * c] at row dst_row0 + t, channels >= f zero
template <typename F>
__global__ __launch_bounds__(256) void pack_input3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                          int t_in, int f, int c, int dst_row0, long dst_bs) {
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)t_in * c) return;
    const int t = (int)(i / c), ch = (int)(i % c);
    unsigned short hi = 0, lo = 0;
    if (ch < f) split2<F>(src[((long)b * t_in + t) * f + ch], hi, lo);
    unsigned short* row = dst + (long)b * dst_bs + (long)(dst_row0 + t) * (3 * c);
    row[ch] = hi;
    row[c + ch] = lo;
    row[2 * c + ch] = hi;
}

// elementwise: v -> (float(hi), v - float(hi)) so that sl_pack_weights' own rounding of the two outputs yields hi and lo
__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ v, float* __restrict__ hi,
                                                        float* __restrict__ lo, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = v[i];
    const float h = bf16_bits_to_f32(f32_to_bf16_bits(x));
    hi[i] = h;
    lo[i] = x - h;
}

// master [k][cin][cout] fp32 -> the two packed bf16x3 operand copies in ONE pass (what sl_split3_weights + 2 x sl_pack_weights
// + 2 x sl_split3_assemble produce in five: 55 launches per optimisation step for the eleven layers):
//   w_fwd3 [cout][k][3 cin]   rows [w_hi | w_hi | w_lo] over cin        w_dgrad3 [cin][k-1-tap][3 cout]  likewise over cout
// with w_hi = bf16(v), w_lo = bf16(v - float(w_hi)).  32 x 32 LDS transpose per tap, as pack_weights_kernel.
template <typename F>
__global__ __launch_bounds__(256) void pack_weights3_kernel(const float* __restrict__ wm, unsigned short* __restrict__ wf,
                                                            unsigned short* __restrict__ wd, int k, int cin, int cout,
                                                            float scale) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int ci0 = blockIdx.y * 32;
    const int co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + ty + r * 8;
        const float v = wm[((long)tap * cin + ci) * cout + co0 + tx] * scale;  // (a power of two: exact)
        tile[ty + r * 8][tx] = v;
        if (wd) {
            unsigned short h, l;
            split2<F>(v, h, l);
            unsigned short* o = wd + ((long)ci * k + (k - 1 - tap)) * 3 * cout + co0 + tx;
            o[0] = h;
            o[cout] = h;
            o[2 * cout] = l;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + ty + r * 8;
        const float v = tile[tx][ty + r * 8];
        unsigned short h, l;
        split2<F>(v, h, l);
        unsigned short* o = wf + ((long)co * k + tap) * 3 * cin + ci0 + tx;
        o[0] = h;
        o[cin] = h;
        o[2 * cin] = l;
    }
}

// rows of `width` bf16: dst[r] = [a[r] | a[r] | b[r]]   (packed weight rows [w_hi | w_hi | w_lo])
__global__ __launch_bounds__(256) void assemble3_kernel(const unsigned short* __restrict__ a,
                                                        const unsigned short* __restrict__ b,
                                                        unsigned short* __restrict__ dst, long rows, int width) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;  // 8 elements (16 bytes) per thread
    if (i >= rows * width) return;
    const long r = i / width;
    const int c = (int)(i % width);
    const u32x4 va = *(const u32x4*)(a + i), vb = *(const u32x4*)(b + i);
    unsigned short* o = dst + r * 3 * width + c;
    *(u32x4*)o = va;
    *(u32x4*)(o + width) = va;
    *(u32x4*)(o + 2 * width) = vb;
}

// dw[tap][ci][co] = RA[tap'][f * fstride + ci][co] + RB[tap'][f * rb_fstride + ci][co] + RA[tap'][f * fstride + c_in + ci][co],
// tap = frames * tap' + f.  RA = weight gradient of (x planes, g_hi), RB = of (x planes, g_lo), both float
// [taps / frames][r_cin][c_out]: the hh, hl and lh terms.  (rb_fstride: the x operand of RB may be a narrower window of the
// row than RA's -- the two hi planes that meet in the middle of a pair row of the striding layer.)
__global__ __launch_bounds__(256) void wgrad_combine3_kernel(const float* __restrict__ ra, const float* __restrict__ rb,
                                                             float* __restrict__ dw, int taps, int c_in, int c_out,
                                                             int frames, int fstride, int ra_cin, int rb_cin, int rb_fstride,
                                                             float scale) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (long)taps * c_in * c_out) return;
    const int co = (int)(i % c_out);
    const long rest = i / c_out;
    const int ci = (int)(rest % c_in), tap = (int)(rest / c_in);
    const int tv = tap / frames, f = tap % frames;
    const float* a = ra + ((long)tv * ra_cin + f * fstride + ci) * c_out + co;
    const float* b = rb + ((long)tv * rb_cin + f * rb_fstride + ci) * c_out + co;
    const f32x4 hh = *(const f32x4*)a, hl = *(const f32x4*)b, lh = *(const f32x4*)(a + (long)c_in * c_out);
    *(f32x4*)(dw + i) = ((hh + hl) + lh) * scale;  // (scale: 1, or the power of two that undoes the f16x3 gradient scaling)
}

// db[co] = sum over the valid frames of (g_hi + g_lo), deterministic two-stage: grid (C / 64, BG_CHUNKS) partial sums over
// interleaved frames (chunk j takes the frames j, j + BG_CHUNKS, ... of every utterance), then a fixed-order sum of the chunks
constexpr int BG_CHUNKS = 64;
template <typename F>
__global__ __launch_bounds__(256) void bias_grad3_partial_kernel(const unsigned short* __restrict__ g, float* __restrict__ part,
                                                                 int batch, int t_out, int c, int row0, long bs) {
    __shared__ float sh[4][64];
    const int ch = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    float acc = 0.f;
    for (int b = 0; b < batch; ++b)
        for (int t = blockIdx.y * 4 + sub; t < t_out; t += 4 * BG_CHUNKS) {
            const unsigned short* row = g + (long)b * bs + (long)(row0 + t) * (3 * c);
            acc += F::dec(row[ch]) + F::dec(row[c + ch]);
        }
    sh[sub][threadIdx.x & 63] = acc;
    __syncthreads();
    if (threadIdx.x < 64)
        part[(long)blockIdx.y * c + ch] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void bias_grad3_final_kernel(const float* __restrict__ part, float* __restrict__ db, int c,
                                                               float scale) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    float s = 0.f;
    for (int j = 0; j < BG_CHUNKS; ++j) s += part[(long)j * c + ch];
    db[ch] = s * scale;
}

// Dropout on a plane tensor (rows of [hi | lo | hi] over c channels).  The keep decision of the element in row r, channel ch
// is dropout_bits(seed, r * c + ch): the index misc.hip's dropout_kernel uses on the single-plane tensor of the same logical
// shape, so a seed draws the same masks on every path.  The value is re-split after the arithmetic (scaling the two planes
// separately would round each to bf16 and leave 2^-9, not 2^-17).
//   MODE 0  forward:   dst = keep ? (hi + lo) * scale : 0                       (Keras inverted dropout, training phase)
//   MODE 1  scale:     dst = (hi + lo) * scale                                  (the 1 / (1 - rate) left of d dropout / dx
//                                                                                behind a ReLU: its mask epilogue saw the
//                                                                                post-dropout activation)
//   MODE 2  ELU bwd:   dst = keep ? (hi + lo) * scale * elu'(z) : 0, elu'(z) = m > 0 ? 1 : m * keep_prob + 1 with m the stored
//                      post-dropout activation y (planes, same geometry)
template <int MODE, typename F>
__global__ __launch_bounds__(256) void split3_dropout_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                             const unsigned short* __restrict__ y, long n, int c,
                                                             unsigned int threshold, float scale, float keep_prob,
                                                             unsigned long long seed) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;  // logical element index (row * c + channel), 8 per thread
    if (i >= n) return;
    const long r = i / c;
    const int ch = (int)(i - r * c);
    const long row = r * 3 * c;
    const u32x4 sh = *(const u32x4*)(src + row + ch), sl = *(const u32x4*)(src + row + c + ch);
    u32x4 yh = (u32x4){0u, 0u, 0u, 0u}, yl = yh;
    if (MODE == 2) {
        yh = *(const u32x4*)(y + row + ch);
        yl = *(const u32x4*)(y + row + c + ch);
    }
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned hw = sh[j >> 1], lw = sl[j >> 1];
        float v = (j & 1) ? dec_hi16<F>(hw) + dec_hi16<F>(lw) : dec_lo16<F>(hw) + dec_lo16<F>(lw);
        const bool keep = MODE == 1 || dropout_bits(seed, (unsigned long long)(i + j)) >= threshold;
        float d = scale;
        if (MODE == 2) {
            const unsigned a = yh[j >> 1], b = yl[j >> 1];
            const float m = (j & 1) ? dec_hi16<F>(a) + dec_hi16<F>(b) : dec_lo16<F>(a) + dec_lo16<F>(b);
            d = m > 0.f ? scale : scale * (m * keep_prob + 1.f);
        }
        v = keep ? v * d : 0.f;
        split2<F>(v, hi[j], lo[j]);
    }
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (unsigned)hi[2 * j] | ((unsigned)hi[2 * j + 1] << 16);
        l[j] = (unsigned)lo[2 * j] | ((unsigned)lo[2 * j + 1] << 16);
    }
    *(u32x4*)(dst + row + ch) = h;
    *(u32x4*)(dst + row + c + ch) = l;
    *(u32x4*)(dst + row + 2 * c + ch) = h;
}

}  // namespace

// ---- host side.  Every plane entry point exists twice: sl_split3* (bf16 planes) and sl_splitf16* (fp16 planes); the twins that
// touch scaled quantities (weights, bias gradients) take the power-of-two scale.
template <typename F>
static int dropout_impl(const char* who, const void* src, void* dst, const void* y, int64_t rows, int channels, int mode,
                        float rate, uint64_t seed, void* stream) {
    SL_CHECK_ARG(src && dst && rows > 0 && channels > 0 && channels % 8 == 0, "%s: channels must be a multiple of 8", who);
    SL_CHECK_ARG(mode >= 0 && mode <= 2 && (mode != 2 || y), "%s: mode 0..2, mode 2 needs the stored activation", who);
    SL_CHECK_ARG(rate >= 0.f && rate < 1.f, "%s: rate %f outside [0, 1)", who, (double)rate);
    const unsigned int threshold = (unsigned int)((double)rate * 4294967296.0);
    const float scale = 1.f / (1.f - rate);
    const long n = (long)rows * channels;
    const dim3 grid((unsigned)((n / 8 + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
#define SL_DROP3(M_)                                                                                                       \
    hipLaunchKernelGGL((split3_dropout_kernel<M_, F>), grid, dim3(256), 0, s, (const unsigned short*)src, (unsigned short*)dst, \
                       (const unsigned short*)y, n, channels, threshold, scale, 1.f - rate, (unsigned long long)seed)
    switch (mode) {
        case 0: SL_DROP3(0); break;
        case 1: SL_DROP3(1); break;
        default: SL_DROP3(2); break;
    }
#undef SL_DROP3
    return sl_check_launch(who);
}
extern "C" int sl_split3_dropout(const void* src, void* dst, const void* y, int64_t rows, int channels, int mode, float rate,
                                 uint64_t seed, void* stream) {
    return dropout_impl<PlaneBf16>("sl_split3_dropout", src, dst, y, rows, channels, mode, rate, seed, stream);
}
extern "C" int sl_splitf16_dropout(const void* src, void* dst, const void* y, int64_t rows, int channels, int mode, float rate,
                                   uint64_t seed, void* stream) {
    return dropout_impl<PlaneF16>("sl_splitf16_dropout", src, dst, y, rows, channels, mode, rate, seed, stream);
}

template <typename F>
static int split_impl(const char* who, const float* src, void* dst, const void* mask, int batch, int t_out, int channels,
                      int64_t src_batch_stride, int dst_row0, int64_t dst_batch_stride, int mode, void* stream) {
    SL_CHECK_ARG(src && dst && batch > 0 && t_out > 0 && channels > 0 && channels % 8 == 0 &&
                     (long)t_out * channels < (1L << 31),
                 "%s: channels must be a multiple of 8, t_out * channels below 2^31", who);
    SL_CHECK_ARG(mode >= 0 && mode <= 4 && (mode < 3 || mask), "%s: mode 0..4, modes 3 / 4 need the mask tensor", who);
    const long n8 = ((long)t_out * channels + 7) / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256), batch);
    hipStream_t s = (hipStream_t)stream;
#define SL_SPLIT3(M_)                                                                                                  \
    hipLaunchKernelGGL((split3_act_kernel<M_, F>), grid, dim3(256), 0, s, src, (unsigned short*)dst, (const unsigned short*)mask, \
                       t_out, channels, (long)src_batch_stride, dst_row0, (long)dst_batch_stride)
    switch (mode) {
        case 0: SL_SPLIT3(0); break;
        case 1: SL_SPLIT3(1); break;
        case 2: SL_SPLIT3(2); break;
        case 3: SL_SPLIT3(3); break;
        default: SL_SPLIT3(4); break;
    }
#undef SL_SPLIT3
    return sl_check_launch(who);
}
extern "C" int sl_split3(const float* src, void* dst, const void* mask, int batch, int t_out, int channels,
                         int64_t src_batch_stride, int dst_row0, int64_t dst_batch_stride, int mode, void* stream) {
    return split_impl<PlaneBf16>("sl_split3", src, dst, mask, batch, t_out, channels, src_batch_stride, dst_row0,
                                 dst_batch_stride, mode, stream);
}
extern "C" int sl_splitf16(const float* src, void* dst, const void* mask, int batch, int t_out, int channels,
                           int64_t src_batch_stride, int dst_row0, int64_t dst_batch_stride, int mode, void* stream) {
    return split_impl<PlaneF16>("sl_splitf16", src, dst, mask, batch, t_out, channels, src_batch_stride, dst_row0,
                                dst_batch_stride, mode, stream);
}

template <typename F>
static int pack_input_impl(const char* who, const float* src, void* dst, int batch, int t_in, int f, int channels, int dst_row0,
                           int64_t dst_batch_stride, void* stream) {
    SL_CHECK_ARG(src && dst && batch > 0 && t_in > 0 && f > 0 && channels >= f, "%s: bad arguments", who);
    const long n = (long)t_in * channels;
    hipLaunchKernelGGL(pack_input3_kernel<F>, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst, t_in, f, channels, dst_row0, (long)dst_batch_stride);
    return sl_check_launch(who);
}
extern "C" int sl_split3_pack_input(const float* src, void* dst, int batch, int t_in, int f, int channels, int dst_row0,
                                    int64_t dst_batch_stride, void* stream) {
    return pack_input_impl<PlaneBf16>("sl_split3_pack_input", src, dst, batch, t_in, f, channels, dst_row0, dst_batch_stride, stream);
}
extern "C" int sl_splitf16_pack_input(const float* src, void* dst, int batch, int t_in, int f, int channels, int dst_row0,
                                      int64_t dst_batch_stride, void* stream) {
    return pack_input_impl<PlaneF16>("sl_splitf16_pack_input", src, dst, batch, t_in, f, channels, dst_row0, dst_batch_stride, stream);
}

extern "C" int sl_split3_weights(const float* v, float* hi, float* lo, size_t n, void* stream) {
    SL_CHECK_ARG(v && hi && lo && n > 0, "sl_split3_weights: bad arguments");
    hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, hi, lo,
                       (long)n);
    return sl_check_launch("sl_split3_weights");
}

template <typename F>
static int pack_weights_impl(const char* who, const float* w_master, void* w_fwd3, void* w_dgrad3, int k, int cin_pad,
                             int cout_pad, float scale, void* stream) {
    SL_CHECK_ARG(w_master && w_fwd3 && k > 0 && cin_pad > 0 && cout_pad > 0 && cin_pad % 32 == 0 && cout_pad % 32 == 0,
                 "%s: channel counts must be multiples of 32", who);
    hipLaunchKernelGGL(pack_weights3_kernel<F>, dim3(cout_pad / 32, cin_pad / 32, k), dim3(256), 0, (hipStream_t)stream, w_master,
                       (unsigned short*)w_fwd3, (unsigned short*)w_dgrad3, k, cin_pad, cout_pad, scale);
    return sl_check_launch(who);
}
extern "C" int sl_split3_pack_weights(const float* w_master, void* w_fwd3, void* w_dgrad3, int k, int cin_pad, int cout_pad,
                                      void* stream) {
    return pack_weights_impl<PlaneBf16>("sl_split3_pack_weights", w_master, w_fwd3, w_dgrad3, k, cin_pad, cout_pad, 1.f, stream);
}
extern "C" int sl_splitf16_pack_weights(const float* w_master, void* w_fwd3, void* w_dgrad3, int k, int cin_pad, int cout_pad,
                                        float scale, void* stream) {
    SL_CHECK_ARG(scale > 0.f, "sl_splitf16_pack_weights: scale must be positive (a power of two)");
    return pack_weights_impl<PlaneF16>("sl_splitf16_pack_weights", w_master, w_fwd3, w_dgrad3, k, cin_pad, cout_pad, scale, stream);
}

extern "C" int sl_split3_assemble(const void* a, const void* b, void* dst, int64_t rows, int width, void* stream) {
    SL_CHECK_ARG(a && b && dst && rows > 0 && width > 0 && width % 8 == 0, "sl_split3_assemble: width must be a multiple of 8");
    const long n8 = rows * width / 8;
    hipLaunchKernelGGL(assemble3_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)a, (const unsigned short*)b, (unsigned short*)dst, (long)rows, width);
    return sl_check_launch("sl_split3_assemble");
}

extern "C" int sl_split3_wgrad_combine_scaled(const float* ra, const float* rb, float* dw, int taps, int c_in, int c_out,
                                              int frames, int fstride, int ra_cin, int rb_cin, int rb_fstride, float scale,
                                              void* stream) {
    SL_CHECK_ARG(ra && rb && dw && taps > 0 && c_in > 0 && c_out > 0 && c_out % 4 == 0 && (frames == 1 || frames == 2) &&
                     taps % frames == 0 && fstride >= 0 && rb_fstride >= 0 && ra_cin >= (frames - 1) * fstride + 2 * c_in &&
                     rb_cin >= (frames - 1) * rb_fstride + c_in && scale > 0.f,
                 "sl_split3_wgrad_combine: bad arguments");
    const long n4 = (long)taps * c_in * c_out / 4;
    hipLaunchKernelGGL(wgrad_combine3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ra, rb,
                       dw, taps, c_in, c_out, frames, fstride, ra_cin, rb_cin, rb_fstride, scale);
    return sl_check_launch("sl_split3_wgrad_combine");
}
extern "C" int sl_split3_wgrad_combine(const float* ra, const float* rb, float* dw, int taps, int c_in, int c_out,
                                       int frames, int fstride, int ra_cin, int rb_cin, int rb_fstride, void* stream) {
    return sl_split3_wgrad_combine_scaled(ra, rb, dw, taps, c_in, c_out, frames, fstride, ra_cin, rb_cin, rb_fstride, 1.f, stream);
}

extern "C" size_t sl_split3_bias_grad_workspace_bytes(int channels) {
    return channels > 0 ? (size_t)BG_CHUNKS * channels * sizeof(float) : 0;
}

template <typename F>
static int bias_grad_impl(const char* who, const void* g, float* db, int batch, int t_out, int channels, int g_row0,
                          int64_t g_batch_stride, float scale, void* workspace, size_t workspace_bytes, void* stream) {
    SL_CHECK_ARG(g && db && batch > 0 && t_out > 0 && channels > 0 && channels % 64 == 0 && scale > 0.f, "%s: bad arguments", who);
    if (workspace == nullptr || workspace_bytes < sl_split3_bias_grad_workspace_bytes(channels)) {
        sl_set_error("%s: workspace too small", who);
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bias_grad3_partial_kernel<F>, dim3(channels / 64, BG_CHUNKS), dim3(256), 0, s, (const unsigned short*)g,
                       (float*)workspace, batch, t_out, channels, g_row0, (long)g_batch_stride);
    int rc = sl_check_launch(who);
    if (rc != SL_OK) return rc;
    hipLaunchKernelGGL(bias_grad3_final_kernel, dim3((channels + 255) / 256), dim3(256), 0, s, (const float*)workspace, db,
                       channels, scale);
    return sl_check_launch(who);
}
extern "C" int sl_split3_bias_grad(const void* g, float* db, int batch, int t_out, int channels, int g_row0,
                                   int64_t g_batch_stride, void* workspace, size_t workspace_bytes, void* stream) {
    return bias_grad_impl<PlaneBf16>("sl_split3_bias_grad", g, db, batch, t_out, channels, g_row0, g_batch_stride, 1.f, workspace,
                                     workspace_bytes, stream);
}
extern "C" int sl_splitf16_bias_grad(const void* g, float* db, int batch, int t_out, int channels, int g_row0,
                                     int64_t g_batch_stride, float scale, void* workspace, size_t workspace_bytes, void* stream) {
    return bias_grad_impl<PlaneF16>("sl_splitf16_bias_grad", g, db, batch, t_out, channels, g_row0, g_batch_stride, scale,
                                    workspace, workspace_bytes, stream);
}
