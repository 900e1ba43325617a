// spectrogram.hip -- audio -> power-level spectrogram and z-normalisation on the GPU.
//
// Replaces the librosa / numpy front end of the reference (speechless/labeled_example.py:99-100 librosa.stft(n_fft 512,
// hop 128), :93-97 |D|^2, :150-158 10 log10 with the -150 dB floor, :28-29 z_normalize).  The mel projection in between
// (labeled_example.py:106-109: a (128 x 257) matrix applied to the dB matrix) is a 1 x 1 convolution and runs on the
// exact-fp32 MFMA kernel of conv_f32.hip through sl_conv1d_nt; see speechless_amd/spectrogram.py.
//
// HBM-bound by nature (0.5 MB of samples in, 1.3 MB of dB values out per 1000 frames); the Fourier transform is a
// radix-2 FFT in LDS, one wave per frame, several frames resident per CU.
#include "common.h"

namespace {

constexpr int FFT_MAX = 1024;

// FRAMES_PER_WG frames of one utterance per work-group of four waves, one wave per frame at a time.
// The window and the twiddle factors are computed once per work-group (they were 768 sincospi per FRAME: most of the 126 us
// the first version of this kernel took for 32 x 1001 frames).  The n_fft real samples of a frame go through a COMPLEX FFT of
// half the length (z[n] = x[2n] + i x[2n+1]; radix-2 decimation in time in the wave's own LDS buffer, input written in
// bit-reversed order, a lone wave's LDS operations execute in order: no barrier) and are split into the spectrum of the
// real sequence afterwards:  X[k] = (Z[k] + conj Z[N/2-k]) / 2  -  i/2 e^{-2 pi i k/N} (Z[k] - conj Z[N/2-k]),  k = 0 .. N/2.
// -> |X_k|^2 -> dB with floor -> row t of the output.
constexpr int FRAMES_PER_WG = 16;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ float2 cmul(float2 w, float2 x) { return make_float2(w.x * x.x - w.y * x.y, w.x * x.y + w.y * x.x); }

__global__ __launch_bounds__(256) void stft_power_db_kernel(const float* __restrict__ audio,
                                                           const long* __restrict__ offsets,
                                                           const int* __restrict__ lengths, float* __restrict__ out,
                                                           int n_fft, int log2n, int hop, int row_stride,
                                                           long batch_stride, float min_db, int max_frames) {
    // LDS (12 KB at n_fft = 512 -> eight work-groups = 32 waves per CU): tw float2[half] | z float2[4][half] | win float[n_fft]
    extern __shared__ float2 stft_lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int len = lengths[b];
    const int n_frames = 1 + len / hop;
    const int half = n_fft / 2;  // length of the complex transform; bins = half + 1
    float2* tw = stft_lds;                         // e^{-2 pi i k / n_fft}, k < n_fft / 2
    float2* zbuf = tw + half;                      // one half-length complex buffer per wave
    float* win = (float*)(zbuf + 4 * half);        // periodic Hann window
    const int log2h = log2n - 1;
    for (int n = tid; n < n_fft; n += 256) {
        float sn, cs;
        sincospif(2.f * (float)n / (float)n_fft, &sn, &cs);
        win[n] = 0.5f - 0.5f * cs;
    }
    for (int k = tid; k < half; k += 256) {
        float sn, cs;
        sincospif(-2.f * (float)k / (float)n_fft, &sn, &cs);
        tw[k] = make_float2(cs, sn);
    }
    __syncthreads();
    const float* y = audio + offsets[b];
    float2* z = zbuf + wave * half;
    const int t_first = blockIdx.x * FRAMES_PER_WG;
    for (int q = wave; q < FRAMES_PER_WG; q += 4) {
        const int t = t_first + q;
        if (t >= max_frames) break;
        float* row = out + (long)b * batch_stride + (long)t * row_stride;
        if (t >= n_frames) {  // frames beyond this utterance: zero rows (the batch is padded with zeros, net.py:583)
            for (int k = lane; k < row_stride; k += 64) row[k] = 0.f;
            continue;
        }
        for (int n = lane; n < half; n += 64) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int idx = t * hop + 2 * n + e - half;  // center=True: the frame is centred on sample t * hop
                if (idx < 0) idx = -idx;               // np.pad(mode="reflect"): edge sample not repeated
                if (idx >= len) idx = 2 * (len - 1) - idx;
                v[e] = win[2 * n + e] * y[idx];
            }
            z[(int)(__brev((unsigned)n) >> (32 - log2h))] = make_float2(v[0], v[1]);
        }
        wave_lds_sync();
        // two radix-2 stages per pass over the buffer, the four points of a pair of butterflies held in registers (the
        // arithmetic of the plain radix-2 recursion, half its LDS round trips and index arithmetic)
        int s = 1;
        for (; s + 1 <= log2h; s += 2) {
            const int lq = s - 1, L = 1 << lq;  // quarter of the block this pass completes
            const int st1 = n_fft >> s, st2 = n_fft >> (s + 1);
            for (int j = lane; j < half / 4; j += 64) {
                const int pos = j & (L - 1);
                const int base = ((j >> lq) << (s + 1)) + pos;
                const float2 a0 = z[base], a1 = z[base + L], a2 = z[base + 2 * L], a3 = z[base + 3 * L];
                const float2 w1 = tw[pos * st1], w2 = tw[pos * st2], w3 = tw[(pos + L) * st2];
                const float2 u1 = cmul(w1, a1), u3 = cmul(w1, a3);
                const float2 b0 = make_float2(a0.x + u1.x, a0.y + u1.y), b1 = make_float2(a0.x - u1.x, a0.y - u1.y);
                const float2 b2 = make_float2(a2.x + u3.x, a2.y + u3.y), b3 = make_float2(a2.x - u3.x, a2.y - u3.y);
                const float2 v2 = cmul(w2, b2), v3 = cmul(w3, b3);
                z[base] = make_float2(b0.x + v2.x, b0.y + v2.y);
                z[base + 2 * L] = make_float2(b0.x - v2.x, b0.y - v2.y);
                z[base + L] = make_float2(b1.x + v3.x, b1.y + v3.y);
                z[base + 3 * L] = make_float2(b1.x - v3.x, b1.y - v3.y);
            }
            wave_lds_sync();
        }
        if (s <= log2h) {  // an odd number of stages: the last one alone
            const int mh = 1 << (s - 1), tstep = n_fft >> s;
            for (int j = lane; j < half / 2; j += 64) {
                const int pos = j & (mh - 1);
                const int i0 = ((j >> (s - 1)) << s) + pos;
                const float2 u = z[i0], x = cmul(tw[pos * tstep], z[i0 + mh]);
                z[i0] = make_float2(u.x + x.x, u.y + x.y);
                z[i0 + mh] = make_float2(u.x - x.x, u.y - x.y);
            }
            wave_lds_sync();
        }
        for (int k = lane; k < row_stride; k += 64) {
            float v = 0.f;
            if (k <= half) {
                const int ka = k & (half - 1), kb = (half - k) & (half - 1);
                const float2 a = z[ka], c = z[kb];                                  // Z[k]; conj Z[N/2 - k] = (c.x, -c.y)
                const float er = 0.5f * (a.x + c.x), ei = 0.5f * (a.y - c.y);
                const float dr = 0.5f * (a.x - c.x), di = 0.5f * (a.y + c.y);       // (Z[k] - conj Z[N/2-k]) / 2
                const float orr = di, oi = -dr;                                     // times -i
                const float2 w = k == half ? make_float2(-1.f, 0.f) : tw[ka];
                const float xr = er + w.x * orr - w.y * oi, xi = ei + w.x * oi + w.y * orr;
                const float pw = xr * xr + xi * xi;
                // 10 log10 p = (10 / log2 10) log2 p on the native base-2 logarithm
                v = pw == 0.f ? min_db : fmaxf(3.0102999566398120f * __log2f(pw), min_db);
            }
            row[k] = v;  // padded lanes are zero: the mel projection contracts over the padded row
        }
        wave_lds_sync();  // the buffer is rewritten by the wave's next frame
    }
}

// z-normalisation statistics of one utterance: mean and population standard deviation over frames[b] x f values, two
// passes, double accumulation (numpy computes them in float64).  ZCH work-groups per utterance and pass (one work-group
// per utterance took 94 us of a 0.26 ms front end): pass 0 leaves per-chunk sums, pass 1 per-chunk sums of squared
// deviations from the mean it rebuilds from pass 0's chunks -- always summed in chunk order, so the statistics of an
// utterance do not depend on the batch it sits in.   part: double[B][2][ZCH]
constexpr int ZCH = 32;
__device__ __forceinline__ double znorm_mean(const double* part, int b, long n) {
    double s = 0.0;
    for (int j = 0; j < ZCH; ++j) s += part[((long)b * 2 + 0) * ZCH + j];
    return s / (double)n;
}
template <int PASS>
__global__ __launch_bounds__(256) void znorm_partial_kernel(const float* __restrict__ src, const int* __restrict__ frames,
                                                            double* __restrict__ part, int f, int row_stride,
                                                            long batch_stride) {
    __shared__ double sh[256];
    const int b = blockIdx.y, j = blockIdx.x;
    const int rows = frames[b];
    const long n = (long)rows * f;
    const int per = (rows + ZCH - 1) / ZCH;
    const int t0 = j * per, t1 = min(rows, t0 + per);
    const double mean = PASS == 1 ? znorm_mean(part, b, n) : 0.0;
    const float* base = src + (long)b * batch_stride;
    double acc = 0.0;
    const int m = max(t1 - t0, 0) * f;  // (a chunk of one utterance: well inside 32 bits)
    for (int i = threadIdx.x; i < m; i += 256) {
        const int q = i / f, c = i - q * f;
        const double v = (double)base[(long)(t0 + q) * row_stride + c];
        acc += PASS == 0 ? v : (v - mean) * (v - mean);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[((long)b * 2 + PASS) * ZCH + j] = sh[0];
}

// dst[b][t][c] = (src - mean_b) / std_b for t < frames[b], 0 beyond (the zero padding of the batch, net.py:583-586)
__global__ __launch_bounds__(256) void znorm_apply_kernel(const float* __restrict__ src, const int* __restrict__ frames,
                                                          const double* __restrict__ part, float* __restrict__ dst,
                                                          int max_frames, int f, int row_stride, long batch_stride) {
    __shared__ double stats[2];  // mean, 1 / standard deviation: once per work-group (they were 64 loads per ELEMENT)
    const int b = blockIdx.y;
    const int rows = frames[b];
    if (threadIdx.x == 0) {
        const long n = (long)rows * f;
        const double mean = n > 0 ? znorm_mean(part, b, n) : 0.0;
        double sq = 0.0;
        for (int j = 0; j < ZCH; ++j) sq += part[((long)b * 2 + 1) * ZCH + j];
        stats[0] = mean;
        stats[1] = n > 0 ? sqrt(sq / (double)n) : 1.0;
    }
    __syncthreads();
    const double mean = stats[0], sd = stats[1];
    // work-group = ZROWS rows of the utterance; a thread walks its columns
    constexpr int ZROWS = 8;
    const int t0 = blockIdx.x * ZROWS;
    const int nrows = min(ZROWS, max_frames - t0);
    const float* sb = src + (long)b * batch_stride;
    float* db = dst + (long)b * max_frames * f;
    for (int i = threadIdx.x; i < nrows * f; i += 256) {
        const int q = i / f, c = i - q * f, t = t0 + q;
        db[(long)t * f + c] = t < rows ? (float)(((double)sb[(long)t * row_stride + c] - mean) / sd) : 0.f;
    }
}

}  // namespace

extern "C" int sl_stft_power_db(const float* audio, const int64_t* offsets, const int32_t* lengths, float* out, int batch,
                                int max_frames, int n_fft, int hop, int row_stride, int64_t batch_stride, float min_db,
                                void* stream) {
    SL_CHECK_ARG(audio && offsets && lengths && out, "sl_stft_power_db: null pointer");
    SL_CHECK_ARG(batch > 0 && max_frames > 0 && hop > 0, "sl_stft_power_db: batch, max_frames and hop must be positive");
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    SL_CHECK_ARG((1 << log2n) == n_fft && n_fft >= 64 && n_fft <= FFT_MAX,
                 "sl_stft_power_db: n_fft = %d must be a power of two in [64, %d]", n_fft, FFT_MAX);
    SL_CHECK_ARG(row_stride >= n_fft / 2 + 1 && batch_stride >= (int64_t)max_frames * row_stride,
                 "sl_stft_power_db: output rows are too short for %d bins", n_fft / 2 + 1);
    const size_t lds = (size_t)(n_fft / 2) * 5 * sizeof(float2) + (size_t)n_fft * sizeof(float);
    hipLaunchKernelGGL(stft_power_db_kernel, dim3((max_frames + FRAMES_PER_WG - 1) / FRAMES_PER_WG, batch), dim3(256), lds,
                       (hipStream_t)stream, audio, (const long*)offsets, lengths, out, n_fft, log2n, hop, row_stride,
                       (long)batch_stride, min_db, max_frames);
    return sl_check_launch("sl_stft_power_db");
}

extern "C" size_t sl_z_normalize_workspace_bytes(int batch) {
    return batch > 0 ? (size_t)batch * 2 * ZCH * sizeof(double) : 0;
}

extern "C" int sl_z_normalize(const float* src, const int32_t* frames, float* dst, int batch, int max_frames, int f,
                              int src_row_stride, int64_t src_batch_stride, void* workspace, size_t workspace_bytes,
                              void* stream) {
    SL_CHECK_ARG(src && frames && dst, "sl_z_normalize: null pointer");
    SL_CHECK_ARG(batch > 0 && max_frames > 0 && f > 0 && src_row_stride >= f, "sl_z_normalize: bad shape");
    if (workspace == nullptr || workspace_bytes < sl_z_normalize_workspace_bytes(batch)) {
        sl_set_error("sl_z_normalize: workspace too small (%zu < %zu)", workspace_bytes,
                     sl_z_normalize_workspace_bytes(batch));
        return SL_ERR_WORKSPACE_TOO_SMALL;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(znorm_partial_kernel<0>, dim3(ZCH, batch), dim3(256), 0, s, src, frames, (double*)workspace, f,
                       src_row_stride, (long)src_batch_stride);
    hipLaunchKernelGGL(znorm_partial_kernel<1>, dim3(ZCH, batch), dim3(256), 0, s, src, frames, (double*)workspace, f,
                       src_row_stride, (long)src_batch_stride);
    int rc = sl_check_launch("sl_z_normalize(stats)");
    if (rc != SL_OK) return rc;
    hipLaunchKernelGGL(znorm_apply_kernel, dim3((unsigned)((max_frames + 7) / 8), batch), dim3(256), 0, s, src, frames,
                       (const double*)workspace, dst, max_frames, f, src_row_stride, (long)src_batch_stride);
    return sl_check_launch("sl_z_normalize(apply)");
}
