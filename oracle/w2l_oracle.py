"""CPU oracle for the speechless Wav2Letter hot path (numpy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing under ``speechless_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

PARITY PINNING STATUS
---------------------
* Greedy decode + label codec: PINNED by the reference's own known-answer tests
  (speechless/test/test_ctc_decoders.py:19-41, test_grapheme_encoding.py:9-31)
  and by fixtures generated from the importable reference module
  ``speechless.grapheme_enconding`` (tests/golden/make_golden.py).
* CTC core (loss, gradient w.r.t. the logits, greedy decoder): PINNED against the
  known-answer vectors of TensorFlow's own unit tests for tf.nn.ctc_loss /
  tf.nn.ctc_greedy_decoder (ctc_loss_op_test.py testBasic, ctc_decoder_ops_test.py),
  the third-party ops net.py:402-406 / 452-454 bottom out in:
  tests/golden/tf_known_answers.json (provenance and self-check in the file).
  keras.backend.ctc_batch_cost on a padded batch: Keras' own backend test
  (backend_test.py::test_ctc, same data, atol 1e-5) is in the same fixture.
* Convolution (TF "SAME" padding incl. the asymmetric stride-2 case, forward, input and
  filter gradients): PINNED on TensorFlow's conv_ops_test.py known answers restated as
  1-D problems (same fixture, section "conv").
* Full-size Conv1D stack beyond those small cases, and Adam: **parity unpinned**.  The arithmetic
  of the reference lives in Keras 2.0.x / TensorFlow 1.x (un-vendored, un-pinned,
  not importable in the build container) and the
  reference holds no golden vectors for it.  This file restates the published
  algorithms (TF "SAME" padding rule, Keras ``ctc_batch_cost`` ->
  ``tf.nn.ctc_loss`` incl. the log(p+1e-8) re-softmax quirk, Keras-2.0 Adam) and
  is guarded by independent checks: brute-force CTC path enumeration, torch-CPU
  ``F.conv1d`` / ``F.ctc_loss`` autograd (oracle/w2l_torch_cpu.py) and finite
  differences (tests/test_oracle.py).

Reference call sites restated here (all paths relative to /root/reference):
  topology ............ speechless/net.py:291-341
  conv layer .......... speechless/net.py:297-305 (Keras Conv1D, padding="same")
  ctc loss ............ speechless/net.py:402-406 (keras.backend.ctc_batch_cost)
  mean-over-batch ..... speechless/net.py:389
  greedy decode ....... speechless/net.py:417-436,452-454,468-475;
                        speechless/grapheme_enconding.py:34-57
  batch packing ....... speechless/net.py:578-607; grapheme_enconding.py:25-32
  optimizer ........... speechless/net.py:132 (keras.optimizers.Adam(1e-4))
"""
from itertools import product as _product

import numpy as np

NEG_INF = -np.inf


# ----------------------------------------------------------------------------------------------
# topology  (net.py:291-341)
# ----------------------------------------------------------------------------------------------
class LayerSpec:
    def __init__(self, name, kernel_size, stride, cin, cout, activation):
        self.name = name
        self.kernel_size = kernel_size
        self.stride = stride
        self.cin = cin
        self.cout = cout
        self.activation = activation  # "relu" | "softmax" | "linear"

    def __repr__(self):
        return "LayerSpec({}, k={}, s={}, {}->{}, {})".format(
            self.name, self.kernel_size, self.stride, self.cin, self.cout, self.activation)


def layer_specs(input_size_per_time_step, grapheme_set_size, main_filter_count=250, out_filter_count=2000,
                activation="relu", output_activation="softmax", inner_count=7,
                striding_kernel=48, inner_kernel=7, big_kernel=32, use_raw_wave_input=False, wave_kernel=250,
                wave_stride=160):
    """The 11-layer spectrogram-input stack of net.py:307-330 (sizes parameterised so that tests can
    build shrunken nets with the same structure); use_raw_wave_input: `wave_conv` (250 taps, stride 160,
    net.py:310-312) in front of it, striding_conv then reads its filters."""
    specs = []
    if use_raw_wave_input:
        specs.append(LayerSpec("wave_conv", wave_kernel, wave_stride, input_size_per_time_step, main_filter_count,
                               activation))
        input_size_per_time_step = main_filter_count
    specs.append(LayerSpec("striding_conv", striding_kernel, 2, input_size_per_time_step, main_filter_count, activation))
    for i in range(1, inner_count + 1):
        specs.append(LayerSpec("inner_conv_{}".format(i), inner_kernel, 1, main_filter_count, main_filter_count,
                               activation))
    specs.append(LayerSpec("big_conv_1", big_kernel, 1, main_filter_count, out_filter_count, activation))
    specs.append(LayerSpec("big_conv_2", 1, 1, out_filter_count, out_filter_count, activation))
    specs.append(LayerSpec("output_conv", 1, 1, out_filter_count, grapheme_set_size, output_activation))
    return specs


def glorot_uniform_weights(specs, seed, dtype=np.float32):
    """Keras default init (glorot_uniform kernel, zero bias): limit = sqrt(6 / (fan_in + fan_out)) with
    fan_in = k*Cin, fan_out = k*Cout.  Kernel layout (k, Cin, Cout) as in net.py:251-255."""
    rng = np.random.RandomState(seed)
    weights = []
    for s in specs:
        limit = np.sqrt(6.0 / (s.kernel_size * s.cin + s.kernel_size * s.cout))
        w = rng.uniform(-limit, limit, size=(s.kernel_size, s.cin, s.cout)).astype(dtype)
        b = np.zeros((s.cout,), dtype=dtype)
        weights.append((w, b))
    return weights


# ----------------------------------------------------------------------------------------------
# bf16 rounding mirror (used to model the HIP kernels' storage rounding points)
# ----------------------------------------------------------------------------------------------
def round_to_bf16(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (what v_cvt_pk_bf16_f32 / a software RNE does)."""
    x32 = np.ascontiguousarray(x, dtype=np.float32)
    u = x32.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = rounded.astype(np.uint32).view(np.float32).reshape(x32.shape)
    # NaN stays NaN (not produced on the hot path); inf stays inf.
    return out


# ----------------------------------------------------------------------------------------------
# Conv1D, TF "SAME" padding  (net.py:304-305)
# ----------------------------------------------------------------------------------------------
def same_padding(t_in, kernel_size, stride):
    t_out = -(-t_in // stride)
    pad_total = max((t_out - 1) * stride + kernel_size - t_in, 0)
    pad_left = pad_total // 2
    return t_out, pad_left, pad_total - pad_left


def conv1d_preactivation(x, w, b, stride):
    """z[b,t,co] = bias[co] + sum_{k,ci} x[b, t*s + k - padL, ci] * W[k,ci,co], zero outside [0,T)."""
    bsz, t_in, cin = x.shape
    k, cin_w, cout = w.shape
    assert cin == cin_w
    t_out, pad_l, pad_r = same_padding(t_in, k, stride)
    xp = np.zeros((bsz, t_in + pad_l + pad_r, cin), dtype=x.dtype)
    xp[:, pad_l:pad_l + t_in] = x
    z = np.zeros((bsz, t_out, cout), dtype=np.result_type(x.dtype, w.dtype))
    for tap in range(k):
        rows = xp[:, tap: tap + (t_out - 1) * stride + 1: stride, :]
        z += rows @ w[tap]
    return z + b


def activate(z, activation):
    if activation == "relu":
        return np.maximum(z, 0)
    if activation == "linear":
        return z
    if activation == "softmax":
        return softmax(z)
    if activation == "elu":
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    raise ValueError(activation)


def softmax(z):
    m = z.max(axis=-1, keepdims=True)
    e = np.exp(z - m)
    return e / e.sum(axis=-1, keepdims=True)


def conv1d_backward(x, w, stride, dz):
    """Given dz = dL/d(pre-activation), returns (dx, dW, db) for one SAME conv layer."""
    bsz, t_in, cin = x.shape
    k, _, cout = w.shape
    t_out, pad_l, pad_r = same_padding(t_in, k, stride)
    xp = np.zeros((bsz, t_in + pad_l + pad_r, cin), dtype=x.dtype)
    xp[:, pad_l:pad_l + t_in] = x
    dxp = np.zeros_like(xp, dtype=dz.dtype)
    dw = np.zeros(w.shape, dtype=dz.dtype)
    for tap in range(k):
        sl = slice(tap, tap + (t_out - 1) * stride + 1, stride)
        rows = xp[:, sl, :]
        dw[tap] = np.einsum("btc,bto->co", rows, dz, optimize=True)
        dxp[:, sl, :] += dz @ w[tap].T
    db = dz.sum(axis=(0, 1))
    return dxp[:, pad_l:pad_l + t_in], dw, db


# ----------------------------------------------------------------------------------------------
# forward stack
# ----------------------------------------------------------------------------------------------
def forward_stack(specs, weights, input_batch, bf16_mirror=False, keep=False, input_scales=None):
    """Runs the 11 conv layers.  Returns probabilities (B,T',K) (and, with keep=True, the list of layer
    inputs and pre-activations needed for backprop).

    input_scales: optional list (one entry per layer, None = identity) of arrays multiplied onto that layer's INPUT:
    Keras' training-phase Dropout in front of a conv (net.py:301-303) with the mask made explicit, i.e.
    keep_mask / (1 - rate) (inverted dropout: kept activations are scaled up, the rest are zero).

    bf16_mirror=True models the HIP bf16 path: weights and every stored activation are rounded to bf16
    (accumulation stays fp32/fp64); the output layer's logits are kept in fp32 (never stored as bf16)."""
    x = input_batch
    if bf16_mirror:
        x = round_to_bf16(x).astype(input_batch.dtype)
    xs, zs = [], []
    for spec, (w, b) in zip(specs, weights):
        if bf16_mirror:
            w = round_to_bf16(w).astype(w.dtype)
        if input_scales is not None and input_scales[len(xs)] is not None:
            x = x * input_scales[len(xs)]
            if bf16_mirror:
                x = round_to_bf16(x).astype(w.dtype)
        z = conv1d_preactivation(x, w, b, spec.stride)
        xs.append(x)
        zs.append(z)
        x = activate(z, spec.activation)
        if bf16_mirror and spec.activation != "softmax":
            x = round_to_bf16(x).astype(z.dtype)
    return (x, xs, zs) if keep else x


# ----------------------------------------------------------------------------------------------
# CTC  (net.py:402-406 -> keras.backend.ctc_batch_cost -> tf.nn.ctc_loss)
# ----------------------------------------------------------------------------------------------
def _logsumexp2(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore"):
        r = m + np.log(np.exp(a - m) + np.exp(b - m))
    return np.where(np.isneginf(m), NEG_INF, r)


def ctc_log_q(probs, eps=1e-8):
    """Keras feeds log(p + eps) to tf.nn.ctc_loss, which re-applies softmax: q = (p+eps)/sum(p+eps)."""
    u = np.log(probs + probs.dtype.type(eps))
    m = u.max(axis=-1, keepdims=True)
    return u - (m + np.log(np.exp(u - m).sum(axis=-1, keepdims=True)))


def ctc_single(log_q, label, blank):
    """Log-space alpha/beta for one utterance.  log_q: (T,K) log-probabilities of the *scored* frames,
    label: list of ints (no blanks).  Returns (loss, dL/du (T,K)) where u are the logits TF's op sees
    (TF convention: gradient = q - occupancy)."""
    t_len, k = log_q.shape
    ext = [blank]
    for c in label:
        ext += [int(c), blank]
    ext = np.array(ext, dtype=np.int64)
    s_len = len(ext)
    dtype = log_q.dtype
    can_skip = np.zeros(s_len, dtype=bool)
    can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])

    alpha = np.full((t_len, s_len), NEG_INF, dtype=dtype)
    beta = np.full((t_len, s_len), NEG_INF, dtype=dtype)
    emit = log_q[:, ext]  # (T,S)
    if t_len == 0:
        return np.inf, np.zeros_like(log_q)
    alpha[0, 0] = emit[0, 0]
    if s_len > 1:
        alpha[0, 1] = emit[0, 1]
    for t in range(1, t_len):
        prev = alpha[t - 1]
        acc = prev.copy()
        acc[1:] = _logsumexp2(acc[1:], prev[:-1])
        skip = np.full(s_len, NEG_INF, dtype=dtype)
        skip[2:] = np.where(can_skip[2:], prev[:-2], NEG_INF)
        acc = _logsumexp2(acc, skip)
        alpha[t] = acc + emit[t]
    beta[t_len - 1, s_len - 1] = emit[t_len - 1, s_len - 1]
    if s_len > 1:
        beta[t_len - 1, s_len - 2] = emit[t_len - 1, s_len - 2]
    for t in range(t_len - 2, -1, -1):
        nxt = beta[t + 1]
        acc = nxt.copy()
        acc[:-1] = _logsumexp2(acc[:-1], nxt[1:])
        skip = np.full(s_len, NEG_INF, dtype=dtype)
        skip[:-2] = np.where(can_skip[2:], nxt[2:], NEG_INF)
        acc = _logsumexp2(acc, skip)
        beta[t] = acc + emit[t]
    log_p = alpha[t_len - 1, s_len - 1]
    if s_len > 1:
        log_p = _logsumexp2(log_p, alpha[t_len - 1, s_len - 2])
    q = np.exp(log_q)
    if np.isneginf(log_p):
        # TF: "No valid path found." -> loss = inf, gradient = softmax (from memory of TF 1.x
        # ctc_loss_calculator; not verifiable offline, edge case only).
        return np.inf, q
    with np.errstate(invalid="ignore"):
        log_gamma = alpha + beta - emit - log_p  # (T,S) state posteriors
    gamma = np.where(np.isfinite(log_gamma), np.exp(log_gamma), 0).astype(dtype)
    occ = np.zeros((t_len, k), dtype=dtype)
    np.add.at(occ, (np.arange(t_len)[:, None], ext[None, :]), gamma)
    return -log_p, q - occ


def ctc_batch_cost(probs, labels, prediction_lengths, label_lengths, eps=1e-8, blank=None):
    """Restates keras.backend.ctc_batch_cost semantics.  probs (B,T,K); labels (B,Lmax) padded (any value
    beyond label_length, grapheme_enconding.py:28 uses -1).  Returns (loss (B,), dL/dprobs (B,T,K)) where
    dL/dprobs is the gradient of the *per-utterance* loss w.r.t. the softmax output of the net."""
    bsz, _, k = probs.shape
    blank = k - 1 if blank is None else blank
    log_q = ctc_log_q(probs, eps)
    losses = np.zeros(bsz, dtype=probs.dtype)
    dprobs = np.zeros_like(probs)
    for i in range(bsz):
        tl = int(prediction_lengths[i])
        ll = int(label_lengths[i])
        loss, du = ctc_single(log_q[i, :tl], list(labels[i, :ll]), blank)
        losses[i] = loss
        # u = log(p + eps)  ->  dL/dp = du / (p + eps); frames >= prediction_length get zero gradient.
        dprobs[i, :tl] = du / (probs[i, :tl] + probs.dtype.type(eps))
    return losses, dprobs


def softmax_backward(probs, dprobs):
    inner = (probs * dprobs).sum(axis=-1, keepdims=True)
    return probs * (dprobs - inner)


def ctc_brute_force(probs_t_k, label, blank, eps=1e-8):
    """-log sum over all K^T alignments that collapse to `label` (tiny cases only).  Pure Python."""
    t_len, k = probs_t_k.shape
    p = probs_t_k.astype(np.float64) + eps
    q = p / p.sum(axis=-1, keepdims=True)
    total = 0.0
    target = list(label)
    for path in _product(range(k), repeat=t_len):
        collapsed = []
        prev = None
        for c in path:
            if c != prev and c != blank:
                collapsed.append(c)
            prev = c
        if collapsed == target:
            pr = 1.0
            for t, c in enumerate(path):
                pr *= q[t, c]
            total += pr
    return -np.log(total) if total > 0 else np.inf


# ----------------------------------------------------------------------------------------------
# loss + all 22 gradients (net.py:359-390: mean over the batch of per-utterance CTC loss)
# ----------------------------------------------------------------------------------------------
def loss_and_gradients(specs, weights, input_batch, labels, prediction_lengths, label_lengths, eps=1e-8,
                       bf16_mirror=False, frozen_layer_count=0, input_scales=None):
    """Returns dict(probs, losses (B,), mean_loss, grads [(dW, db)] * n_layers, dlogits).
    bf16_mirror: False | True (weights, stored activations and stored gradients rounded to bf16, as the HIP bf16 path
    stores them) | "fp32_g" (the same with the back-propagated signal kept unrounded).
    input_scales: see forward_stack (explicit dropout masks); the input gradient of a layer passes through the same
    multiplier on its way to the previous layer's activation."""
    probs, xs, zs = forward_stack(specs, weights, input_batch, bf16_mirror=bf16_mirror, keep=True,
                                  input_scales=input_scales)
    bsz = input_batch.shape[0]
    losses, dprobs = ctc_batch_cost(probs, labels, prediction_lengths, label_lengths, eps)
    assert specs[-1].activation == "softmax"
    dz = softmax_backward(probs, dprobs) / bsz
    dlogits = dz.copy()
    grads = [(np.zeros_like(w), np.zeros_like(b)) for (w, b) in weights]
    dzs = [None] * len(specs)  # dL/d(pre-activation) per layer, as consumed by that layer's wgrad/dgrad
    # frozen layers (net.py:335-339) are the FIRST frozen_layer_count layers: no dW/db for them and no
    # dgrad below the first trainable layer.
    for li in range(len(specs) - 1, frozen_layer_count - 1, -1):
        spec = specs[li]
        w, _ = weights[li]
        if bf16_mirror:
            w = round_to_bf16(w).astype(w.dtype)
            if bf16_mirror != "fp32_g":  # "fp32_g": what-if experiment, back-propagated signal stored unrounded
                dz = round_to_bf16(dz).astype(dz.dtype)
        dx, dw, db = conv1d_backward(xs[li], w, spec.stride, dz)
        grads[li] = (dw, db)
        dzs[li] = dz
        if li == frozen_layer_count:
            break
        if input_scales is not None and input_scales[li] is not None:
            dx = dx * input_scales[li]
        prev_spec = specs[li - 1]
        if prev_spec.activation == "relu":
            dz = dx * (zs[li - 1] > 0)
        elif prev_spec.activation == "linear":
            dz = dx
        elif prev_spec.activation == "elu":
            dz = dx * np.where(zs[li - 1] > 0, 1.0, np.exp(np.minimum(zs[li - 1], 0)))
        else:
            raise ValueError(prev_spec.activation)
    return dict(probs=probs, losses=losses, mean_loss=losses.mean(), grads=grads, dlogits=dlogits, dzs=dzs)


# ----------------------------------------------------------------------------------------------
# greedy decode  (net.py:452-454 + 468-475; grapheme_enconding.py:34-57)
# ----------------------------------------------------------------------------------------------
def greedy_decode_indices(probs, prediction_lengths, blank=None, merge_repeated=True):
    """argmax per frame (first max wins, like numpy.argmax / Eigen maxCoeff) for t < prediction_length,
    merge consecutive repeats, THEN drop blanks (test_ctc_decoders.py:40: 'A A _ A A' -> [0, 0])."""
    bsz, _, k = probs.shape
    blank = k - 1 if blank is None else blank
    out = []
    for i in range(bsz):
        idx = np.argmax(probs[i, :int(prediction_lengths[i])], axis=-1)
        seq = []
        prev = None
        for c in idx:
            c = int(c)
            if not (merge_repeated and c == prev) and c != blank:
                seq.append(c)
            prev = c
        out.append(seq)
    return out


def frame_argmax_and_margin(probs):
    """Per-frame argmax and the top1-top2 margin (used to attribute argmax flips to near-ties)."""
    idx = np.argmax(probs, axis=-1)
    srt = np.sort(probs, axis=-1)
    return idx, srt[..., -1] - srt[..., -2]


# ----------------------------------------------------------------------------------------------
# batch packing  (net.py:578-607, grapheme_enconding.py:25-32)
# ----------------------------------------------------------------------------------------------
def pack_input_batch(spectrograms, length_ratio=2):
    t_max = max(s.shape[0] for s in spectrograms)
    f = spectrograms[0].shape[1]
    batch = np.zeros((len(spectrograms), t_max, f))  # float64 like the reference (net.py:583)
    for i, s in enumerate(spectrograms):
        batch[i, :s.shape[0], :s.shape[1]] = s
    return batch, [s.shape[0] // length_ratio for s in spectrograms]


def pack_label_batch(encoded_labels):
    l_max = max(len(l) for l in encoded_labels)
    out = -np.ones((len(encoded_labels), l_max), dtype=np.int32)
    for i, l in enumerate(encoded_labels):
        out[i, :len(l)] = np.array(l, dtype=np.int32)
    return out


# ----------------------------------------------------------------------------------------------
# Keras 2.0 Adam  (net.py:132)
# ----------------------------------------------------------------------------------------------
def keras_adam_step(p, g, m, v, step, lr=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
    """step is the 1-based iteration count AFTER increment (Keras: t = iterations + 1)."""
    lr_t = lr * np.sqrt(1.0 - beta_2 ** step) / (1.0 - beta_1 ** step)
    m_t = beta_1 * m + (1.0 - beta_1) * g
    v_t = beta_2 * v + (1.0 - beta_2) * g * g
    p_t = p - lr_t * m_t / (np.sqrt(v_t) + epsilon)
    return p_t, m_t, v_t
