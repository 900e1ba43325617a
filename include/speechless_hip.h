/*
 * speechless_hip.h -- C-ABI of the MI355X (gfx950) implementation of the speechless Wav2Letter hot path.
 *
 * The reference (juliuskunze/speechless) exposes NO native / FFI interface for this path: every operator below
 * replaces an *implicit* third-party op that speechless/net.py reaches through Keras/TensorFlow.  Each entry
 * point cites the reference call site it stands in for (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; device pointers are raw HBM addresses (e.g. tensor.data_ptr()).
 *   - return 0 on success, a negative sl_status on error; never throw, never abort, never allocate.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no hidden synchronisation.
 *   - no global mutable state; sl_last_error() is thread-local.
 *
 * Data layout in HBM ("halo'd channels-last"): an activation tensor is [batch][rows][channels] with
 *   rows     = halo_top + padded_time + halo_bottom   (halo rows and rows >= valid time are ZERO, always)
 *   channels = channel count padded to a multiple of 128 (padded lanes are ZERO, always)
 * so a SAME-padded conv tap is a plain row-shifted view and no kernel needs bounds checks on reads.
 * padded_time must be a multiple of SL_TIME_TILE (256): the conv kernels read whole time tiles of up to 256 rows, i.e.
 * rows x_row0 .. x_row0 + ceil(t_out / 256) * 256 + taps - 2 of every utterance must lie inside its `rows`.
 */
#ifndef SPEECHLESS_HIP_H
#define SPEECHLESS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SL_VERSION 1
#define SL_TIME_TILE 256

typedef enum sl_status {
    SL_OK = 0,
    SL_ERR_INVALID_ARGUMENT = -1,
    SL_ERR_UNSUPPORTED = -2,
    SL_ERR_WORKSPACE_TOO_SMALL = -3,
    SL_ERR_LAUNCH_FAILED = -4
} sl_status;

/* SL_F16 (round 6): fp16 operands on v_mfma_f32_16x16x32_f16 with fp32 or hi + lo PLANE outputs -- sl_conv1d_nt (out_f32 = 1
 * or 2), sl_conv1d_wgrad and sl_conv1d_wgrad_multi only: the "f16x3" parity path (every fp32 value as two fp16 planes, three
 * MFMA terms per product; speechless_amd/engine_x3.py).  No reference counterpart: the reference computes in fp32. */
typedef enum sl_dtype { SL_BF16 = 0, SL_F32 = 1, SL_F16 = 2 } sl_dtype;

/* epilogue of sl_conv1d_nt */
typedef enum sl_epilogue {
    SL_EPI_NONE = 0,      /* y = acc                                  (dgrad into a linear layer)              */
    SL_EPI_BIAS = 1,      /* y = acc + bias[co]                       (output_conv logits, net.py:328-330)     */
    SL_EPI_BIAS_RELU = 2, /* y = max(acc + bias[co], 0)               (Conv1D(activation="relu"), net.py:304)  */
    SL_EPI_RELU_MASK = 3, /* y = acc * (mask[b,t,co] > 0)             (autodiff through relu, net.py:389,550)  */
    SL_EPI_BIAS_ELU = 4,  /* y = elu(acc + bias[co])                  (activation="elu", main.py:71-78)         */
    SL_EPI_ELU_MASK = 5   /* y = acc * (m > 0 ? 1 : m + 1), m = mask  (autodiff through elu via its output)      */
} sl_epilogue;

/*
 * Geometry of one stride-1 "row-shifted GEMM" convolution.  A stride-2 layer (striding_conv, net.py:317-319) is
 * presented in its PAIR VIEW: two consecutive input frames form one row of 2*Cin channels, which turns the
 * k=48/stride-2 conv into a k=24/stride-1 conv over the same memory (no padding waste, no strided loads).
 *
 *   out[b][y_row0 + t][co] = epi( sum_{tap < taps} sum_{c < cin} x[b][x_row0 + t + tap][c] * w[co][tap][c] )
 *   for t in [0, t_out), co in [0, cout)
 */
typedef struct sl_conv_geom {
    int32_t batch;          /* B                                                                        */
    int32_t t_out;          /* valid output rows per utterance (T'); rows >= t_out are never written     */
    int32_t taps;           /* taps of the (pair-)view                                                   */
    int32_t cin;            /* contraction channels per tap, multiple of 64                              */
    int32_t cout;           /* output channels, multiple of 128                                          */
    int32_t x_row0;         /* first input row read by (t = 0, tap = 0)                                  */
    int32_t x_row_stride;   /* elements between input rows (>= cin)                                      */
    int64_t x_batch_stride; /* elements between utterances in x                                          */
    int32_t y_row0;         /* output row of t = 0                                                       */
    int32_t y_row_stride;   /* elements between output rows (>= cout)                                    */
    int64_t y_batch_stride; /* elements between utterances in y (and in mask)                            */
    float acc_scale;        /* SL_F16 only: sl_conv1d_nt multiplies the accumulator by this before bias / mask (the f16x3 path
                             * stores weights and gradients scaled by powers of two); 0 = 1.  Ignored by the other dtypes. */
} sl_conv_geom;

int sl_version(void);
const char* sl_last_error(void);

/* Measurement hook (bench.py's roofline leg): the two HIP events (hipEvent_t, created with timing enabled) are attached to
 * the dispatch of the MAIN kernel of the next sl_conv1d_nt / sl_conv1d_wgrad / sl_conv1d_wgrad_grouped call
 * (hipExtLaunchKernelGGL: they carry the kernel's own begin and end timestamps) -- not to its split-K epilogue or reduction
 * tail -- and the hook disarms itself.  The duration between them is what `rocprofv3 --kernel-trace --stats` reports for
 * that kernel.  Either pointer may be NULL. */
int sl_profile_next_kernel(void* start_event, void* stop_event);

/* ---- conv forward and input-gradient (both are the same row-shifted NT GEMM) ---------------------------------
 * Replaces: Keras Conv1D forward (net.py:304-305 -> TF conv2d/bias_add/relu) and, with flipped/transposed packed
 * weights, TF Conv2DBackpropInput reached by autodiff from net.py:389,550.
 *   x      : [B][rows][x_row_stride]  dtype
 *   w      : packed weights [cout][taps][cin] dtype (see sl_pack_weights)
 *   bias   : float[cout] or NULL (SL_EPI_BIAS*)
 *   mask   : same geometry as y, dtype (SL_EPI_RELU_MASK) or NULL
 *   y      : [B][rows][y_row_stride]; dtype, or float when out_f32 = 1.  out_f32 = 2 (bf16 only; epilogues NONE, BIAS,
 *            BIAS_RELU, RELU_MASK): the result leaves the kernel as bf16x3 planes -- rows [hi | lo | hi] of 3 x cout
 *            channels (y_row_stride >= 3 cout), hi = bf16(v), lo = bf16(v - hi) of the fp32 value v after the epilogue;
 *            RELU_MASK then reads the mask's hi plane in y's geometry (what sl_split3 does in a second pass over HBM).
 *   cfg    : 0 = library picks the tile shape / pipeline depth / split-K for this geometry (measured table);
 *            otherwise (tuner / tests) wm | wn<<4 | stages<<8 | ksplit<<12 | it<<20 | m32<<24 | (1+log2 gm)<<25 |
 *            slab<<29 | interleaved<<30: wm x wn waves, each a (16*it) x 64 patch (it = 0 means 4; it = 5: the it = 4
 *            tile with eight waves, the two waves of a SIMD splitting the k-halves; m32: 32x32x16 MFMA, 64x64 patch);
 *            stages = ring slots, +8 = register-pipelined loop; gm = m-tiles per raster block; slab = chunk-major
 *            kernel with the activation slab in LDS; interleaved = hand-interleaved MFMA / LDS-read / request
 *            streams.  SL_ERR_INVALID_ARGUMENT for a shape that is not instantiated
 *            or that the geometry rules out.  SL_F32: 0 = exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32, 128 x 128
 *            tiles), 1 = the plain VALU FMA kernel (independent cross-check of the former).
 *   workspace: sl_conv1d_nt_workspace_bytes(geom, dtype, cfg) bytes (split-K partial tiles; 0 when not split).
 */
size_t sl_conv1d_nt_workspace_bytes(const sl_conv_geom* geom, int dtype, int cfg);
int sl_conv1d_nt(const void* x, const void* w, const float* bias, const void* mask, void* y,
                 const sl_conv_geom* geom, int epilogue, int dtype, int out_f32, int cfg, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ---- a RUN of identical stride-1 convolutions in one launch (bf16) --------------------------------------------------
 * Replaces: the seven Conv1D(250, 7) `inner_conv_i` of net.py:321-323 -- n_layers consecutive sl_conv1d_nt calls of the
 * same geometry (256 padded channels in and out, odd taps <= 9, every tensor in the same halo'd layout) whose
 * intermediate activations stay in LDS; each layer's output is still written to HBM (ys[i]: the backward pass needs it).
 *   epilogue SL_EPI_BIAS_RELU: forward; layer i: ys[i] = relu(conv(i == 0 ? x : ys[i-1], ws[i]) + biases[i])
 *   epilogue SL_EPI_RELU_MASK: input gradients; layer i: ys[i] = conv(i == 0 ? x : ys[i-1], ws[i]) * (masks[i] > 0),
 *            ws[i] the dgrad operand (taps flipped) of the i-th convolution from the TOP of the run
 *   geom: the geometry ONE layer of the run would pass to sl_conv1d_nt.  sl_conv1d_chain_supported() = 1 when the run
 *   fits; otherwise (or for fp32 / ELU / dropout between the layers) use n_layers calls of sl_conv1d_nt.
 */
int sl_conv1d_chain_supported(const sl_conv_geom* geom, int n_layers, int dtype);
/* Measurement / test hook: output frames per work-group of sl_conv1d_chain.  0 (default) = chosen per launch (64, or 48 where
 * that fills the chip's rounds of 256 work-groups better: long utterances in small batches), 48 / 64 = forced.  Process-wide. */
int sl_conv1d_chain_select(int tile_rows);
int sl_conv1d_chain(const void* x, void* const* ys, const void* const* ws, const float* const* biases,
                    const void* const* masks, const sl_conv_geom* geom, int n_layers, int epilogue, int dtype,
                    void* stream);

/* ---- conv weight gradient -------------------------------------------------------------------------------------
 * Replaces: TF Conv2DBackpropFilter reached by autodiff from net.py:389,550.
 *   dw[tap][c][co] = sum_b sum_{t < t_pad} x[b][x_row0 + t + tap][c] * g[b][g_row0 + t][co]     (fp32 out)
 * geom: batch,taps,cin,cout,x_* as above; y_* describe g (y_row0 = g_row0, ...); t_out = valid rows of g
 * (rows >= t_out of g are zero by the layout invariant, so the kernel sums whole 64-row chunks).
 * Deterministic: split-K partials go to `workspace` and are reduced in a fixed order.
 * cfg: 0 = library picks tile shape / ring depth / batch split (measured table); otherwise
 *      wm | wn<<4 | stages<<8 | splits<<12 (tile = 64*wm input channels x 64*wn output channels; splits 0 = auto).
 *      SL_F32: 0 = exact-fp32 MFMA kernel (channel counts that are multiples of 128), 1 = VALU FMA kernel.
 */
size_t sl_conv1d_wgrad_workspace_bytes(const sl_conv_geom* geom, int dtype, int cfg);
int sl_conv1d_wgrad(const void* x, const void* g, float* dw, const sl_conv_geom* geom, int dtype, int cfg,
                    void* workspace, size_t workspace_bytes, void* stream);

/* The same weight gradient for `groups` layers of IDENTICAL geometry in one launch (bf16 only): layer q reads
 * x + q*x_group_stride and g + q*g_group_stride (elements) and writes dw + q*dw_group_stride (floats).  The seven
 * inner_conv_i (net.py:321-323) are such a group: one grid of 7 x 28 tiles needs 3 batch splits instead of 16 per
 * layer, runs 80-step contractions instead of 16-step ones and one reduction instead of seven. */
size_t sl_conv1d_wgrad_grouped_workspace_bytes(const sl_conv_geom* geom, int groups, int cfg);
int sl_conv1d_wgrad_grouped(const void* x, const void* g, float* dw, const sl_conv_geom* geom, int groups,
                            int64_t x_group_stride, int64_t g_group_stride, int64_t dw_group_stride, int cfg,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- backward of a 1x1 convolution onto <= 32 real output channels in ONE launch (bf16) -----------------------------------
 * Replaces, for output_conv (net.py:326-330: Conv1D(grapheme_set_size, 1)), the pair sl_conv1d_nt(dgrad) +
 * sl_conv1d_wgrad reached by autodiff from net.py:389,550: both gradients of that layer are HBM-bound passes over its 2000-
 * channel input, which this call reads ONCE.
 *   x       : the layer's input [B][rows][x_row_stride] bf16 -- also the stored activation whose sign (ReLU) / value (ELU)
 *             is the mask of dx
 *   g       : gradient w.r.t. the layer's output, [B][rows][y_row_stride] bf16, columns >= k_real zero
 *   w_dgrad : the layer's dgrad operand [cin][1][cout] bf16 (sl_pack_weights)
 *   dx      : gradient w.r.t. x, same geometry as x:  dx = (g . w^T) * act'(x);  frames in [t_out, ceil(t_out / 64) * 64)
 *             are written with the zeros they hold by the layout invariant
 *   dw      : float[cin][cout], complete on return (columns >= 32 zero); its row cin - 1 is the bias gradient when x carries
 *             the ones channel (sl_bias_grad_from_wgrad)
 *   geom    : the layer's WEIGHT-GRADIENT geometry (taps = 1; x_* describe x and dx, y_* describe g), cin % 128 == 0
 *   epilogue: SL_EPI_RELU_MASK or SL_EPI_ELU_MASK
 *   cfg     : 0 = library picks the number of work-groups (one per CU); otherwise (tuner / tests) that number
 *   workspace: sl_conv1d_backward_1x1_workspace_bytes() bytes (per-work-group partial sums of dw, combined in a fixed order
 *             by a small second kernel of the same call: deterministic, no float atomics). */
int sl_conv1d_backward_1x1_supported(const sl_conv_geom* geom, int k_real, int dtype);
size_t sl_conv1d_backward_1x1_workspace_bytes(const sl_conv_geom* geom, int k_real, int dtype, int cfg);
int sl_conv1d_backward_1x1(const void* x, const void* g, const void* w_dgrad, void* dx, float* dw, const sl_conv_geom* geom,
                           int epilogue, int k_real, int dtype, int cfg, void* workspace, size_t workspace_bytes,
                           void* stream);
/* The same for ONE PART of the batch (x, g, dx point at the part's first utterance, geom->batch = its utterances):
 * accumulate = 0 writes dw (the first part), accumulate = 1 adds this part's sum to what dw holds (later parts; same stream,
 * fixed order: deterministic).  The engine runs the top of the step by half-batches so that the CTC lattice of one half
 * (net.py:402-406, a handful of latency-bound waves) runs under the top layers of the other half (Engine.split_top). */
int sl_conv1d_backward_1x1_part(const void* x, const void* g, const void* w_dgrad, void* dx, float* dw,
                                const sl_conv_geom* geom, int epilogue, int k_real, int dtype, int cfg, int accumulate,
                                void* workspace, size_t workspace_bytes, void* stream);

/* How many CUs the library's grid choosers may count on: 0 = all (default 256), else 64 .. 256.  Per calling THREAD
 * (thread-local; round 6 -- it was process-wide), read when a launch is enqueued on that thread.  The MFMA kernels take a whole CU per work-group and size their grids to whole rounds of the chip; when
 * communication kernels own some CUs during backward (data-parallel runs: GradBucketReducer sets 256 - its channel count) the
 * split / segment / tile-height choosers of sl_conv1d_nt, sl_conv1d_wgrad[_grouped|_multi], sl_conv1d_chain and
 * sl_conv1d_backward_1x1 plan their rounds for that many CUs instead.  (No reference counterpart: the reference is
 * single-device, main.py:14-24.) */
int sl_set_available_cus(int cus);

/* ---- the weight gradients of SEVERAL layers in one balanced launch (bf16) ------------------------------------------------
 * Replaces a sequence of sl_conv1d_wgrad / sl_conv1d_wgrad_grouped calls (TF Conv2DBackpropFilter of several Conv1D layers,
 * net.py:389,550) for layers with few 256 x 256 tiles -- inner_conv_1..7 and striding_conv of net.py:317-323: the
 * (tile, 64-frame step) space of all jobs is cut into one equal range per CU, so no CU idles behind a batch split whose
 * granularity is an utterance; per-range partial tiles are added in a fixed order by a second kernel of the same call
 * (deterministic).  job.geom: the layer's weight-gradient geometry as for sl_conv1d_wgrad; every job needs the same
 * batch and t_out, cin % 256 == 0, cout % 256 == 0.  SL_ERR_UNSUPPORTED otherwise (use the per-layer calls). */
#define SL_WGRAD_MULTI_MAX_JOBS 16
typedef struct sl_wgrad_job {
    const void* x;     /* the layer's input  (geom.x_*) */
    const void* g;     /* the gradient w.r.t. its output (geom.y_*) */
    float* dw;         /* float[taps][cin][cout] */
    sl_conv_geom geom;
} sl_wgrad_job;
size_t sl_conv1d_wgrad_multi_workspace_bytes(const sl_wgrad_job* jobs, int n_jobs, int dtype);
int sl_conv1d_wgrad_multi(const sl_wgrad_job* jobs, int n_jobs, int dtype, void* workspace, size_t workspace_bytes,
                          void* stream);

/* bias gradient db[co] = sum_{b,t} g[b][g_row0+t][co] (fp32 out, deterministic two-stage).  Same autodiff site. */
size_t sl_bias_grad_workspace_bytes(const sl_conv_geom* geom);
int sl_bias_grad(const void* g, float* db, const sl_conv_geom* geom, int dtype, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ---- weight packing -------------------------------------------------------------------------------------------
 * master weights are fp32 in the Keras kernel layout [k][cin_pad][cout_pad] (net.py:251-255 evidence for
 * (kernel_size, in_channels, filters)).  Produces
 *   w_fwd [cout_pad][k][cin_pad]            (forward operand)      and, if w_dgrad != NULL,
 *   w_dgrad[cin_pad][k][cout_pad]  with taps flipped: w_dgrad[ci][j][co] = w[k-1-j][ci][co]
 */
int sl_pack_weights(const float* w_master, void* w_fwd, void* w_dgrad, int k, int cin_pad, int cout_pad,
                    int dtype, void* stream);

/* ---- input packing  (net.py:578-587 zero-pad + cast; rows offset by the SAME left pad) -------------------------
 * src: float[B][t_in][f] (dense, host-packed batch already resident in HBM)
 * dst: [B][dst_rows][dst_row_stride] dtype, PRE-ZEROED; frame t lands in row dst_row0 + t.
 */
int sl_pack_input(const float* src, void* dst, int batch, int t_in, int f, int dst_row0, int dst_row_stride,
                  int64_t dst_batch_stride, int dtype, void* stream);
/* Raw-wave input (net.py:310-312: `wave_conv`, Conv1D(250, kernel 250, stride 160, padding="same") over the samples in front
 * of striding_conv when use_raw_wave_input=True): the sample windows of the t_out = ceil(t_in / stride) output frames as rows of
 * k * cin columns (zero padded to dst_row_stride), pad_left = TF's SAME rule for this t_in -- over which the layer is a 1 x 1
 * GEMM for sl_conv1d_nt (forward, output straight into the pair-view input buffer of the stack) and sl_conv1d_wgrad.
 *   audio: float[B][t_in][cin];   dst: [B][rows >= t_out][dst_row_stride] dtype at dst_batch_stride elements per utterance */
int sl_wave_frames(const float* audio, void* dst, int batch, int t_in, int cin, int k, int stride, int pad_left, int t_out,
                   int dst_row_stride, int64_t dst_batch_stride, int dtype, void* stream);
/* The same with a ONES CHANNEL: padding channel `ones_channel` (f <= ones_channel < dst_row_stride) of every packed frame is set
 * to 1 (-1: none).  Its weight rows are and stay zero, so the forward pass does not see it; the first layer's weight-gradient
 * GEMM then leaves that layer's BIAS gradient in row ones_channel of dW (sl_bias_grad_from_wgrad) -- the bias gradient of
 * striding_conv (net.py:317-319) without a pass of its own over the layer's gradient tensor. */
int sl_pack_input_ones(const float* src, void* dst, int batch, int t_in, int f, int dst_row0, int dst_row_stride,
                       int64_t dst_batch_stride, int ones_channel, int dtype, void* stream);

/* ---- output softmax (net.py:131,328-330) + the Keras ctc_batch_cost prologue log(p+eps) re-normalised -----------
 * logits: float, row t of utterance b at logits + b*logit_batch_stride + t*logit_stride (first k valid).
 * probs,logq: float[B][t_out][k] dense.
 * logq = log_softmax(log(probs + eps)) = what tf.nn.ctc_loss sees after its own softmax (net.py:405-406).
 */
int sl_softmax_logq(const float* logits, float* probs, float* logq, int batch, int t_out, int k, int logit_stride,
                    int64_t logit_batch_stride, float eps, void* stream);

/* ---- output layer fused with the softmax (bf16): the 1x1 convolution onto k <= 32 classes (net.py:326-330,
 * Conv1D(grapheme_set_size, 1, activation="softmax")) and everything sl_softmax_logq computes, in one launch that
 * streams the layer's input once and never writes the logits unless asked to.
 *   x, geom (batch, t_out, taps = 1, cin, cout >= 32 rows of w, x_*): as for sl_conv1d_nt;  w: packed [cout][1][cin]
 *   probs, logq: float[B][t_out][k];  logits: optional float, row t of utterance b at b*logit_batch_stride +
 *   t*logit_stride (first k written), or NULL.   sl_output_softmax_supported() = 1 when the geometry fits the kernel
 *   (otherwise use sl_conv1d_nt + sl_softmax_logq).
 */
int sl_output_softmax_supported(const sl_conv_geom* geom, int k, int dtype);
/* Measurement / test hook: which kernel sl_output_softmax runs.  0 (default) = automatic: the weights in REGISTERS (each of
 * the four waves of a work-group owns a quarter of the input channels and its own LDS-DMA ring; cin = 256 ... 2048) where
 * that applies, else in LDS; 1 = always the LDS kernel; 2 = as 0.  Process-wide. */
int sl_output_softmax_select(int variant);
int sl_output_softmax(const void* x, const void* w, const float* bias, float* probs, float* logq, float* logits,
                      const sl_conv_geom* geom, int k, int logit_stride, int64_t logit_batch_stride, float eps, int dtype,
                      void* stream);

/* ---- CTC loss and gradient (net.py:402-406: keras.backend.ctc_batch_cost -> tf.nn.ctc_loss) --------------------
 * labels: int32[B][l_max] (padding ignored, grapheme_enconding.py:28 uses -1); blank = k-1 (grapheme_enconding.py:125).
 * loss:   float[B]  (-log p(label | x); +inf when no valid alignment exists)
 * dlogits: gradient of  grad_scale * sum_b loss[b]  w.r.t. the PRE-softmax logits of output_conv, written into the
 *          halo'd tensor described by (g_row0, g_row_stride, g_batch_stride), dtype `dtype`; frames >=
 *          input_len[b] get zeros.  grad_scale = 1/B realises Keras' mean over the batch (net.py:389).
 * workspace: sl_ctc_workspace_bytes(...) bytes (alpha/beta lattices).
 * Two lattice kernels: with labels of up to 255 graphemes and k <= 63 a probability-domain one (one wave per utterance
 * and direction, eight lattice states per lane, block floating point with one exponent per lane and 8 / 16 frames, no
 * transcendental, no LDS exchange and no barrier on the T'-long sequential path) in doubles; the gradient kernel then
 * checks every frame's posteriors against 1, and an utterance that lost mass to underflow is redone in the log domain --
 * in doubles since round 6: as accurate as the lattice it replaces -- by the last work-group of the same launch (no further
 * launches; none of the regimes of a training run -- near-uniform start, blank collapse, a net that has learnt its labels
 * -- needs it; alignments with next to no slack, labels filling > 90 % of the frames, can).  Labels beyond 255 graphemes go through the
 * log-domain lattice kernel (one thread per lattice state, LDS row exchange + barrier per frame).  Results agree to fp32
 * round-off.
 */
size_t sl_ctc_workspace_bytes(int batch, int t_out, int l_max);
int sl_ctc_loss_grad(const float* probs, const float* logq, const int32_t* labels, const int32_t* label_len,
                     const int32_t* input_len, float* loss, void* dlogits, int batch, int t_out, int k, int l_max,
                     int g_row0, int g_row_stride, int64_t g_batch_stride, int dtype, float eps, float grad_scale,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Measurement / test hook: which lattice sl_ctc_loss_grad runs.  0 (default) = as described above, 1 = log-domain lattice
 * only; probability-domain lattice in doubles: 2 = without the repair pass, 3 = and then EVERY utterance redone by the
 * repair pass, 4 = + repair; in floats (faster, but the repair pass is needed in some regimes: ctc.hip:WaveReal): 6 / 7 / 5
 * likewise; 8 / 9 (round 6) = the double lattice on a PAIR of waves per utterance and direction (four states per lane, the
 * boundary state handed over through an LDS mailbox, the downstream wave half a block behind) with / without the repair
 * pass; 10 / 11 (round 6) = the lone double lattice wave with a HELPER wave beside it that fetches the probabilities and gathers
 * every lattice lane's emissions into 48 contiguous bytes of LDS per frame (three wide LDS reads per frame instead of five, no
 * staging work in the lattice wave) with / without the repair pass.  Process-wide; not for concurrent use with sl_ctc_loss_grad. */
int sl_ctc_select(int variant);

/* ---- greedy decode (net.py:452-454 tf.nn.ctc_greedy_decoder, merge_repeated=True; numpy twin
 *      grapheme_enconding.py:34-57): per-frame argmax (first max wins) for t < input_len, merge repeats, drop blank.
 * out: int32[B][t_out] filled with -1 past out_len[b] (sparse_to_dense default, net.py:436). frame_argmax: optional
 * int32[B][t_out] raw per-frame argmax (for parity attribution), or NULL.
 */
int sl_greedy_decode(const float* probs, const int32_t* input_len, int32_t* out, int32_t* out_len,
                     int32_t* frame_argmax, int batch, int t_out, int k, int blank, void* stream);

/* The same fused update for SEVERAL layers in one launch (the small layers' launches are pure latency): layer i's
 * block starts `offset` floats into param / grad / m / v (weights [k][cin_pad][cout_pad] followed by cout_pad biases). */
#define SL_ADAM_MAX_LAYERS 16
typedef struct {
    int64_t offset;  /* first weight of the layer, in floats from the four base pointers (multiple of 4) */
    void* w_fwd;     /* [cout_pad][k][cin_pad] dtype */
    void* w_dgrad;   /* [cin_pad][k][cout_pad] dtype, taps flipped; NULL for a layer without input gradient */
    int32_t k, cin_pad, cout_pad;
} sl_adam_layer;
int sl_adam_pack_layers(float* param, const float* grad, float* m, float* v, const sl_adam_layer* layers, int n_layers,
                        int dtype, int step, float lr, float beta1, float beta2, float eps, void* stream);
/* The operand rewrite alone (sl_pack_weights for several layers in one launch, same table): used when the masters were
 * updated elsewhere -- the data-parallel step with a sharded optimizer runs sl_adam_step on this rank's slice of the flat
 * buffers and all-gathers the masters (speechless_amd/parallel.py; the reference has no counterpart, main.py:14-24). */
int sl_pack_layers(const float* param, const sl_adam_layer* layers, int n_layers, int dtype, void* stream);

/* ---- Bias gradients out of the weight-gradient GEMM ("ones channel") ------------------------------------------------
 * The channel padding of an activation tensor (250 -> 256, 2000 -> 2048) is multiplied through every GEMM anyway.  When the
 * LAST padded output channel of layer i-1 carries the constant 1 on every valid frame (bias 1, zero weights; the host side
 * sets that up), row cin_pad - 1 of layer i's weight gradient is  dW[k][cin_pad-1][co] = sum over the frames whose tap-k
 * input frame is valid of g[frame][co]  -- for the tap that reads the frame itself (k = pad_left) exactly the bias gradient
 * of layer i (BiasAddGrad, net.py:389 autodiff), computed by sl_conv1d_wgrad at no extra cost instead of by sl_bias_grad's
 * extra pass over g.  This call moves it to the bias-gradient slot and zeroes the row for all taps, so that the padded
 * weights stay zero under the optimizer.  grads: the flat fp32 gradient buffer; per layer the element offsets of dW
 * ([k][cin_pad][cout_pad]) and db ([cout_pad]); copy = 0: zero the row only (e.g. dropout was applied to the ones).
 */
#define SL_BGW_MAX_LAYERS 16
typedef struct sl_bgw_layer {
    int64_t w_off, b_off;
    int32_t k, cin_pad, cout_pad, tap;
} sl_bgw_layer;
int sl_bias_grad_from_wgrad(float* grads, const sl_bgw_layer* layers, int n_layers, int copy, void* stream);

/* ---- Dropout (net.py:301-303: a Keras Dropout(rate) layer in front of every conv except the last three; training
 * phase only, `dropout=None` in every reference configuration) ------------------------------------------------------
 * sl_dropout: dst[i] = keep_i ? src[i] / (1 - rate) : 0 over n elements (dst may be src), keep_i a pure function of
 * (seed, i) -> a step is reproducible from its seed.  Zero halo / padding elements stay zero, so it can run over a
 * whole halo'd tensor.  Because the activation is stored AFTER dropout, the ReLU-mask epilogue of the next dgrad
 * (mask = stored value > 0) applies the keep mask for free; sl_scale supplies the remaining 1 / (1 - rate) factor of
 * the backward pass (x[i] *= scale).
 */
int sl_dropout(const void* src, void* dst, size_t n, int dtype, float rate, uint64_t seed, void* stream);
int sl_scale(void* x, size_t n, int dtype, float scale, void* stream);
/* Dropout behind an ELU layer (activation="elu", main.py:71-78, with dropout=...): a stored zero cannot tell a dropped
 * element from elu(z) == 0, so the input gradient is produced with SL_EPI_NONE and this pass applies both factors:
 *   g[i] = keep_i ? g[i] / (1 - rate) * (y[i] > 0 ? 1 : y[i] * (1 - rate) + 1) : 0
 * with y the stored post-dropout activation and keep_i recomputed from (seed, i) exactly as sl_dropout drew it. */
int sl_elu_dropout_backward(void* g, const void* y, size_t n, int dtype, float rate, uint64_t seed, void* stream);

/* ---- Keras-2.0 Adam (net.py:132): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps) ------------------- */
int sl_adam_step(float* param, const float* grad, float* m, float* v, size_t n, int step, float lr, float beta1,
                 float beta2, float eps, void* stream);

/* Same Adam update for ONE layer block laid out [k][cin_pad][cout_pad] weights followed by cout_pad biases (param,
 * grad, m, v point at the block's first weight), fused with sl_pack_weights: the pass that updates the fp32 masters
 * also rewrites w_fwd [cout_pad][k][cin_pad] and (if non-NULL) w_dgrad [cin_pad][k][cout_pad] (taps flipped). */
int sl_adam_pack_layer(float* param, const float* grad, float* m, float* v, void* w_fwd, void* w_dgrad, int k,
                       int cin_pad, int cout_pad, int dtype, int step, float lr, float beta1, float beta2, float eps,
                       void* stream);

/* ---- "bf16x3": the fast parity path ---------------------------------------------------------------------------------------
 * north_star: greedy-decoded indices bit-exact against the reference's fp32 CPU path (net.py:417-436 on Keras / TF float32),
 * gradients within 1e-3.  Every fp32 value is carried as two bf16 numbers (hi = bf16(v), lo = bf16(v - hi)) in THREE planes
 * per tensor row, [hi | lo | hi] (3 x channels), against packed weight rows [w_hi | w_hi | w_lo]: the unchanged
 * sl_conv1d_nt over 3 x channels then computes x_hi w_hi + x_lo w_hi + x_hi w_lo in fp32 (out_f32 = 1 into a staging buffer, or out_f32 = 2 straight into planes),
 * and these HBM-bound helpers move between the fp32 staging form and the planes.  See csrc/split3.hip.
 *   sl_split3            fp32 [B][src rows][channels] (valid rows t < t_out) -> planes [B][rows][3 * channels] at row
 *                        dst_row0 + t.  mode 0: copy; 1: relu; 2: elu (Conv1D activations, net.py:304); 3 / 4: multiply by the
 *                        ReLU / ELU derivative taken from `mask`, the stored activation in plane form (autodiff, net.py:389)
 *   sl_split3_pack_input float[B][t_in][f] -> planes, channels >= f zero (net.py:578-587)
 *   sl_split3_weights    v -> (float(hi), v - float(hi)): sl_pack_weights of the two arrays yields the hi and lo operands
 *   sl_split3_assemble   rows of `width` bf16: dst[r] = [a[r] | a[r] | b[r]]
 *   sl_split3_wgrad_combine  RA = sl_conv1d_wgrad of (x planes [hi | lo ...], g_hi), RB = of (x planes [hi ...], g_lo), float
 *                        [taps / frames][r*_cin][c_out] -> dw[tap][ci][co] = hh + hl + lh;  frames = 2, fstride = 3 * c_in for
 *                        the pair view of a stride-2 layer (two frames' planes per row); rb_fstride = the frame stride inside
 *                        RB's rows (= fstride, or c_in when RB's x operand was the [hi of frame 0 | hi of frame 1] window
 *                        in the middle of the pair row)
 *   sl_split3_bias_grad  db[co] = sum over valid frames of g_hi + g_lo (two stages, fixed order; workspace from
 *                        sl_split3_bias_grad_workspace_bytes) */
int sl_split3(const float* src, void* dst, const void* mask, int batch, int t_out, int channels, int64_t src_batch_stride,
              int dst_row0, int64_t dst_batch_stride, int mode, void* stream);
int sl_split3_pack_input(const float* src, void* dst, int batch, int t_in, int f, int channels, int dst_row0,
                         int64_t dst_batch_stride, void* stream);
int sl_split3_weights(const float* v, float* hi, float* lo, size_t n, void* stream);
int sl_split3_assemble(const void* a, const void* b, void* dst, int64_t rows, int width, void* stream);
/* The bf16x3 operand copies of one layer from its fp32 master [k][cin_pad][cout_pad] in one pass: w_fwd3 [cout][k][3 cin]
 * = rows [w_hi | w_hi | w_lo], w_dgrad3 [cin][k-1-tap][3 cout] likewise (or NULL); what sl_split3_weights + 2 x
 * sl_pack_weights + 2 x sl_split3_assemble produce, bit for bit. */
int sl_split3_pack_weights(const float* w_master, void* w_fwd3, void* w_dgrad3, int k, int cin_pad, int cout_pad,
                           void* stream);
/* sl_adam_pack_layers for the bf16x3 path: Keras-2.0 Adam on the fp32 masters of up to SL_ADAM_MAX_LAYERS layers and their
 * [w_hi | w_hi | w_lo] operand copies (layers[i].w_fwd / .w_dgrad = the 3-plane tensors) rewritten in the same pass. */
int sl_split3_adam_pack_layers(float* param, const float* grad, float* m, float* v, const sl_adam_layer* layers, int n_layers,
                               int step, float lr, float beta1, float beta2, float eps, void* stream);
int sl_split3_wgrad_combine(const float* ra, const float* rb, float* dw, int taps, int c_in, int c_out, int frames,
                            int fstride, int ra_cin, int rb_cin, int rb_fstride, void* stream);
size_t sl_split3_bias_grad_workspace_bytes(int channels);
int sl_split3_bias_grad(const void* g, float* db, int batch, int t_out, int channels, int g_row0, int64_t g_batch_stride,
                        void* workspace, size_t workspace_bytes, void* stream);
/* Dropout on a plane tensor of `rows` rows of [hi | lo | hi] over `channels` channels (Keras Dropout, net.py:301-303, on the
 * bf16x3 path).  The keep decision of (row r, channel c) is the one sl_dropout draws for element r * channels + c of the
 * single-plane tensor of the same logical shape, so a seed means the same masks on every path; values are re-split after the
 * arithmetic.  mode 0: dst = keep ? v / (1 - rate) : 0 (forward);  mode 1: dst = v / (1 - rate) (what is left of
 * d dropout / dx behind a ReLU, cf. sl_scale);  mode 2: dst = keep ? v / (1 - rate) * elu'(z) : 0 with y the stored
 * post-dropout activation (cf. sl_elu_dropout_backward).  dst may be src. */
int sl_split3_dropout(const void* src, void* dst, const void* y, int64_t rows, int channels, int mode, float rate,
                      uint64_t seed, void* stream);

/* ---- "f16x3" (round 6): the same three-plane scheme on fp16 pairs -- hi = fp16(v), lo = fp16(v - hi): 22 significand bits
 * where |v| >= 2^-3 and an absolute 2^-25 below (the lo plane runs into fp16 denormals, which v_mfma_f32_16x16x32_f16 keeps:
 * tools/f16_denorm_probe.hip) against bf16x3's 16-17, at the same three MFMA terms and the same rate; range 65504, so the engine
 * stores weights and back-propagated gradients multiplied by powers of two (exact) and the kernels divide them out again:
 * sl_conv1d_nt(dtype = SL_F16) through sl_conv_geom.acc_scale, the helpers below through `scale`.  Same reference rows as
 * bf16x3 (net.py:304, 389, 402-406: one fp32 arithmetic).  Twins of the sl_split3* entry points above, fp16 planes:
 *   sl_splitf16, sl_splitf16_pack_input, sl_splitf16_dropout    as sl_split3, sl_split3_pack_input, sl_split3_dropout
 *   sl_splitf16_pack_weights   operand copies of scale * w
 *   sl_splitf16_adam_pack_layers  Adam on the masters, operand copies of w_scale * w
 *   sl_splitf16_bias_grad      db = scale * sum over valid frames of (g_hi + g_lo)
 *   sl_split3_wgrad_combine_scaled  dw = scale * (hh + hl + lh)   (fp32 in and out: either plane format) */
int sl_splitf16(const float* src, void* dst, const void* mask, int batch, int t_out, int channels, int64_t src_batch_stride,
                int dst_row0, int64_t dst_batch_stride, int mode, void* stream);
int sl_splitf16_pack_input(const float* src, void* dst, int batch, int t_in, int f, int channels, int dst_row0,
                           int64_t dst_batch_stride, void* stream);
int sl_splitf16_pack_weights(const float* w_master, void* w_fwd3, void* w_dgrad3, int k, int cin_pad, int cout_pad, float scale,
                             void* stream);
int sl_splitf16_adam_pack_layers(float* param, const float* grad, float* m, float* v, const sl_adam_layer* layers, int n_layers,
                                 int step, float lr, float beta1, float beta2, float eps, float w_scale, void* stream);
int sl_split3_wgrad_combine_scaled(const float* ra, const float* rb, float* dw, int taps, int c_in, int c_out, int frames,
                                   int fstride, int ra_cin, int rb_cin, int rb_fstride, float scale, void* stream);
int sl_splitf16_bias_grad(const void* g, float* db, int batch, int t_out, int channels, int g_row0, int64_t g_batch_stride,
                          float scale, void* workspace, size_t workspace_bytes, void* stream);
int sl_splitf16_dropout(const void* src, void* dst, const void* y, int64_t rows, int channels, int mode, float rate,
                        uint64_t seed, void* stream);

/* ---- audio front end (speechless/labeled_example.py:99-160, 28-29) ---------------------------
 * sl_stft_power_db: librosa.stft(y, n_fft, hop_length) with its defaults (periodic Hann window of n_fft samples,
 * center=True with reflect padding of n_fft/2, 1 + len/hop frames, 1 + n_fft/2 bins; labeled_example.py:99-100), then
 * |D|^2 (:93-97) and the power level 10 log10(.) with `min_db` for zero power and as the floor (:150-158), in one pass.
 *   audio   : float, all utterances back to back; utterance b starts at audio + offsets[b] and has lengths[b] samples
 *             (lengths[b] > n_fft / 2: reflect padding)
 *   out     : float[B][max_frames][row_stride] at batch_stride; row t of utterance b holds its 1 + n_fft/2 levels followed
 *             by zeros up to row_stride; rows t >= 1 + lengths[b] / hop are zero.  Time-major, so that the mel projection
 *             (labeled_example.py:106-109: a matrix applied to the LEVEL spectrogram) is a 1 x 1 sl_conv1d_nt over it.
 * n_fft: power of two in [64, 1024].
 */
int sl_stft_power_db(const float* audio, const int64_t* offsets, const int32_t* lengths, float* out, int batch,
                     int max_frames, int n_fft, int hop, int row_stride, int64_t batch_stride, float min_db,
                     void* stream);

/* z_normalize (labeled_example.py:28-29, 136-140): per utterance (a - mean(a)) / std(a) over its frames[b] x f values
 * (population standard deviation, float64 statistics like numpy), written densely as float[B][max_frames][f]; rows
 * t >= frames[b] are zero -- exactly the zero-padded input batch of net.py:578-587, ready for sl_pack_input.
 * workspace: sl_z_normalize_workspace_bytes(batch) bytes. */
size_t sl_z_normalize_workspace_bytes(int batch);
int sl_z_normalize(const float* src, const int32_t* frames, float* dst, int batch, int max_frames, int f,
                   int src_row_stride, int64_t src_batch_stride, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHLESS_HIP_H */
