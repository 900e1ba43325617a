"""Device-side engine of the Wav2Letter hot path: owns HBM buffers (torch tensors = plumbing only) and sequences the
hand-written gfx950 kernels of libspeechless_hip.so for forward, CTC, backward, Adam.

Reference path being replaced (paths relative to the reference root): the Keras graph built by
speechless/net.py:291-341 (11 x Conv1D), :359-390 (loss_net, mean CTC loss, Adam 1e-4) and the two backend functions
:350-357 / :456-459.

HBM layout (see include/speechless_hip.h): every activation / gradient tensor is [B][HALO + Tt_pad + HALO][C_pad]
channels-last with zero halo rows, zero rows beyond the valid time and zero padded channels, so that a SAME-padded
conv tap is a row-shifted view.  The stride-2 first layer reads its input in the PAIR VIEW ([rows/2][2*C]).
Master weights / gradients / Adam moments live in ONE flat fp32 buffer each (layer order), in the Keras kernel layout
(k, Cin_pad, Cout_pad); the gradient buffer is what the data-parallel all-reduce operates on.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import ConvGeom, lib

HALO = 16
TIME_TILE = 256  # SL_TIME_TILE: the conv kernels read whole time tiles of up to 256 rows


def _round_up(x, m):
    return (x + m - 1) // m * m


class LayerSpec:
    def __init__(self, name, kernel_size, stride, cin, cout, activation):
        self.name = name
        self.kernel_size = kernel_size
        self.stride = stride
        self.cin = cin
        self.cout = cout
        self.activation = activation


def wav2letter_layer_specs(input_size_per_time_step, grapheme_set_size, activation="relu",
                           output_activation="softmax", main_filter_count=250, out_filter_count=2000, inner_count=7,
                           striding_kernel=48, inner_kernel=7, big_kernel=32):
    """Topology of reference net.py:307-330 (spectrogram input).  Sizes are parameters only so that tests can build
    shrunken stacks of the same structure."""
    specs = [LayerSpec("striding_conv", striding_kernel, 2, input_size_per_time_step, main_filter_count, activation)]
    for i in range(1, inner_count + 1):
        specs.append(LayerSpec("inner_conv_{}".format(i), inner_kernel, 1, main_filter_count, main_filter_count,
                               activation))
    specs.append(LayerSpec("big_conv_1", big_kernel, 1, main_filter_count, out_filter_count, activation))
    specs.append(LayerSpec("big_conv_2", 1, 1, out_filter_count, out_filter_count, activation))
    specs.append(LayerSpec("output_conv", 1, 1, out_filter_count, grapheme_set_size, output_activation))
    return specs


def same_padding(t_in, kernel_size, stride):
    """TF 'SAME': T_out = ceil(T/s); pad_total = max((T_out-1)*s + k - T, 0); extra padding goes right."""
    t_out = -(-t_in // stride)
    pad_total = max((t_out - 1) * stride + kernel_size - t_in, 0)
    return t_out, pad_total // 2, pad_total - pad_total // 2


class LayerPlan:
    def __init__(self, index, spec, cin_pad, cout_pad, w_off, b_off):
        self.index = index
        self.spec = spec
        self.cin_pad = cin_pad
        self.cout_pad = cout_pad
        k = spec.kernel_size
        if spec.stride == 2:
            if k % 2:
                raise NotImplementedError("stride-2 layers need an even kernel size (pair view)")
            self.taps_view = k // 2
            self.cin_view = 2 * cin_pad
            # pair view needs pad_left odd/even consistent with row offset; pad_left of SAME stride 2, even k is k/2-1
            self.pad_left = (k - 2) // 2 if k >= 2 else 0
            self.pad_right = None  # depends on T parity, not needed in the pair view
        else:
            self.taps_view = k
            self.cin_view = cin_pad
            self.pad_left = (k - 1) // 2
            self.pad_right = (k - 1) - self.pad_left
        self.w_off = w_off
        self.w_numel = k * cin_pad * cout_pad
        self.b_off = b_off


class _Buffers:
    """All HBM tensors of one (batch, padded frames) geometry.  Batches of any length whose output frames round up to
    the same multiple of TIME_TILE share one set of buffers (the reference's training generator, corpus.py:224-226,
    pads every batch to its own longest member, so the frame count changes with nearly every step): set_length()
    re-targets the geometry descriptors and keeps the layout invariant (rows beyond the valid time are zero) by clearing
    only the rows between the new length and the previous high-water mark."""

    def __init__(self, eng, batch, tt_pad):
        dev = eng.device
        dt = eng.torch_dtype
        p0 = eng.plans[0]
        self.batch = batch
        self.tt_pad = tt_pad
        self.t_in = None
        self.t_out = None
        self.rows = HALO + self.tt_pad + HALO
        self.rows0 = 2 * (self.tt_pad + p0.taps_view)
        self.x0 = torch.zeros((batch, self.rows0, p0.cin_pad), dtype=dt, device=dev)
        self.x0_dropped = None  # dropout(x0), allocated by the first training forward with dropout
        self.dropped = False    # the activations of the last forward are post-dropout
        n = len(eng.plans)
        self.y = [None] * (n - 1)
        self._blocks = []  # every halo'd allocation (runs of identical layers are one), for set_length()'s clearing
        # a run of identical layers (the seven inner_conv_i) keeps its inputs y[s-1..e-1] in ONE allocation so that
        # the grouped weight-gradient launch can address layer q as base + q*stride
        for (s0, e0) in eng.runs:
            block = torch.zeros((e0 - s0 + 1, batch, self.rows, eng.plans[s0].cin_pad), dtype=dt, device=dev)
            self._blocks.append(block)
            for q in range(e0 - s0 + 1):
                self.y[s0 - 1 + q] = block[q]
        for p in eng.plans[:-1]:
            if self.y[p.index] is None:
                self.y[p.index] = torch.zeros((batch, self.rows, p.cout_pad), dtype=dt, device=dev)
                self._blocks.append(self.y[p.index].unsqueeze(0))
        self.logits = torch.zeros((batch, self.tt_pad, eng.plans[-1].cout_pad), dtype=torch.float32, device=dev)
        k = eng.grapheme_set_size
        # dense [B][T'][K] / [B][T'] results: flat allocations for the longest batch, viewed per length
        self._probs_flat = torch.zeros((batch * self.tt_pad * k,), dtype=torch.float32, device=dev)
        self._logq_flat = torch.zeros((batch * self.tt_pad * k,), dtype=torch.float32, device=dev)
        self._decoded_flat = torch.zeros((batch * self.tt_pad,), dtype=torch.int32, device=dev)
        self._argmax_flat = torch.zeros((batch * self.tt_pad,), dtype=torch.int32, device=dev)
        self.g = [None] * n  # allocated lazily by ensure_backward()
        self.decoded_len = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.input_len = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.loss = torch.zeros((batch,), dtype=torch.float32, device=dev)
        self.fwd_geom = []
        for p in eng.plans:
            g = ConvGeom()
            g.batch = batch
            g.t_out = self.tt_pad
            g.taps = p.taps_view
            g.cin = p.cin_view
            g.cout = p.cout_pad
            if p.index == 0:
                g.x_row0 = 0
                g.x_row_stride = p.cin_view
                g.x_batch_stride = self.rows0 * p.cin_pad
            else:
                g.x_row0 = HALO - p.pad_left
                g.x_row_stride = p.cin_pad
                g.x_batch_stride = self.rows * p.cin_pad
            if p.index == n - 1:
                g.y_row0 = 0
                g.y_row_stride = p.cout_pad
                g.y_batch_stride = self.tt_pad * p.cout_pad
            else:
                g.y_row0 = HALO
                g.y_row_stride = p.cout_pad
                g.y_batch_stride = self.rows * p.cout_pad
            self.fwd_geom.append(g)
        self.wgrad_geom = [None] * n
        self.dgrad_geom = [None] * n
        self.bwd_ready = False
        self.nt_ws = None
        self.wgrad_ws = None
        self.launch_lists = {}   # recorded launch lists (Engine._replay); dropped whenever a pointer they hold changes
        self._ws_sized = set()   # output lengths whose workspace needs have been checked
        self._clean_in = 0       # input frames / output rows up to which stale data may sit in the buffers
        self._clean_out = 0

    def set_length(self, eng, t_in):
        """Re-targets the buffers at batches of t_in input frames (same tt_pad)."""
        p0 = eng.plans[0]
        t_out, pad_l, _ = same_padding(t_in, p0.spec.kernel_size, p0.spec.stride)
        assert pad_l == p0.pad_left and _round_up(t_out, TIME_TILE) == self.tt_pad
        # rows [new length, high-water mark) still hold the previous, longer batch: the kernels never write rows
        # beyond the valid time, so they are cleared here (nothing to do while the lengths grow)
        if t_in < self._clean_in:
            self.x0[:, p0.pad_left + t_in: p0.pad_left + self._clean_in].zero_()
        if t_out < self._clean_out:
            for block in self._blocks:
                block[:, :, HALO + t_out: HALO + self._clean_out].zero_()
        self._clean_in, self._clean_out = t_in, t_out
        if t_in == self.t_in:
            return
        self.t_in, self.t_out = t_in, t_out
        k = eng.grapheme_set_size
        b = self.batch
        self.probs = self._probs_flat[:b * t_out * k].view(b, t_out, k)
        self.logq = self._logq_flat[:b * t_out * k].view(b, t_out, k)
        self.decoded = self._decoded_flat[:b * t_out].view(b, t_out)
        self.frame_argmax = self._argmax_flat[:b * t_out].view(b, t_out)
        for geoms in (self.fwd_geom, self.wgrad_geom, self.dgrad_geom):
            for g in geoms:
                if g is not None:
                    g.t_out = t_out
        if t_out not in self._ws_sized:  # split counts (hence workspace sizes) depend on the number of time tiles
            self._ws_sized.add(t_out)
            self.size_nt_workspace(eng, self.fwd_geom, "fwd")
            if self.bwd_ready:
                self.size_backward_workspaces(eng)

    def size_nt_workspace(self, eng, geoms, kind):
        need = 16
        for p, g in zip(eng.plans, geoms):
            if g is not None:
                need = max(need, lib().raw("sl_conv1d_nt_workspace_bytes")(
                    ctypes.byref(g), eng.dtype_code, eng.nt_cfg.get((kind, p.spec.name), 0)))
        if self.nt_ws is None or self.nt_ws.numel() < need:
            self.nt_ws = torch.empty((need,), dtype=torch.uint8, device=eng.device)
            self.launch_lists = {}

    def ensure_backward(self, eng):
        if self.bwd_ready:
            return
        dev, dt = eng.device, eng.torch_dtype
        n = len(eng.plans)
        first = eng.frozen_layer_count
        for (s0, e0) in eng.runs:  # gradients g[s..e] of a run of identical layers: one allocation (grouped wgrad)
            lo = max(s0, first)
            if e0 >= lo:
                block = torch.zeros((e0 - lo + 1, self.batch, self.rows, eng.plans[lo].cout_pad), dtype=dt, device=dev)
                self._blocks.append(block)
                for q in range(e0 - lo + 1):
                    self.g[lo + q] = block[q]
        for p in eng.plans[first:]:
            if self.g[p.index] is None:
                self.g[p.index] = torch.zeros((self.batch, self.rows, p.cout_pad), dtype=dt, device=dev)
                self._blocks.append(self.g[p.index].unsqueeze(0))
            wg = ConvGeom()
            f = self.fwd_geom[p.index]
            for name, _ in ConvGeom._fields_:
                setattr(wg, name, getattr(f, name))
            wg.y_row0 = HALO
            wg.y_row_stride = p.cout_pad
            wg.y_batch_stride = self.rows * p.cout_pad
            self.wgrad_geom[p.index] = wg
            if p.index > first:
                dg = ConvGeom()
                dg.batch = self.batch
                dg.t_out = self.t_out
                dg.taps = p.spec.kernel_size
                dg.cin = p.cout_pad
                dg.cout = p.cin_pad
                dg.x_row0 = HALO - p.pad_right
                dg.x_row_stride = p.cout_pad
                dg.x_batch_stride = self.rows * p.cout_pad
                dg.y_row0 = HALO
                dg.y_row_stride = p.cin_pad
                dg.y_batch_stride = self.rows * p.cin_pad
                self.dgrad_geom[p.index] = dg
        self.bias_ws = None
        self.ctc_ws = None
        self.ctc_ws_lmax = -1
        self.labels = None
        self.label_len = torch.zeros((self.batch,), dtype=torch.int32, device=dev)
        self.bwd_ready = True
        self.size_backward_workspaces(eng)

    def size_backward_workspaces(self, eng):
        L = lib()
        first = eng.frozen_layer_count
        ws_bytes = 0
        bias_ws = 0
        for p in eng.plans[first:]:
            wg = self.wgrad_geom[p.index]
            ws_bytes = max(ws_bytes, L.raw("sl_conv1d_wgrad_workspace_bytes")(
                ctypes.byref(wg), eng.dtype_code, eng.nt_cfg.get(("wgrad", p.spec.name), 0)))
            bias_ws = max(bias_ws, L.raw("sl_bias_grad_workspace_bytes")(ctypes.byref(wg)))
        self.size_nt_workspace(eng, self.dgrad_geom, "dgrad")
        if eng.dtype == "bf16":
            for (s0, e0) in eng.runs:
                lo = max(s0, first)
                if e0 > lo:
                    ws_bytes = max(ws_bytes, L.raw("sl_conv1d_wgrad_grouped_workspace_bytes")(
                        ctypes.byref(self.wgrad_geom[lo]), e0 - lo + 1, 0))
        if self.wgrad_ws is None or self.wgrad_ws.numel() < ws_bytes:
            self.wgrad_ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=eng.device)
            self.launch_lists = {}
        if self.bias_ws is None or self.bias_ws.numel() < bias_ws:
            self.bias_ws = torch.empty((max(bias_ws, 16),), dtype=torch.uint8, device=eng.device)
            self.launch_lists = {}

    def ensure_ctc(self, eng, l_max):
        if self.ctc_ws is not None and l_max <= self.ctc_ws_lmax:
            return
        need = lib().raw("sl_ctc_workspace_bytes")(self.batch, self.tt_pad, l_max)  # covers every length
        self.ctc_ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=eng.device)
        self.ctc_ws_lmax = l_max
        self.labels = torch.zeros((self.batch, l_max), dtype=torch.int32, device=eng.device)


class Engine:
    """Forward / CTC / backward / Adam on one MI355X.  dtype 'bf16' (bf16 storage, fp32 accumulate, fp32 CTC) or
    'f32' (parity path)."""

    def __init__(self, specs, grapheme_set_size, dtype="bf16", device="cuda:0", ctc_epsilon=1e-8,
                 frozen_layer_count=0, lr=1e-4, beta_1=0.9, beta_2=0.999, adam_epsilon=1e-8):
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("speechless_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                       "there is no CPU fallback for the hot path")
        self.lib = lib()
        self.device = torch.device(device)
        self.dtype = dtype
        if dtype == "bf16":
            self.torch_dtype, self.dtype_code = torch.bfloat16, _lib.SL_BF16
        elif dtype == "f32":
            self.torch_dtype, self.dtype_code = torch.float32, _lib.SL_F32
        else:
            raise ValueError("dtype must be 'bf16' or 'f32'")
        self.specs = specs
        self.grapheme_set_size = grapheme_set_size
        self.ctc_epsilon = ctc_epsilon
        self.frozen_layer_count = frozen_layer_count
        self.lr, self.beta_1, self.beta_2, self.adam_epsilon = lr, beta_1, beta_2, adam_epsilon
        self.adam_iterations = 0
        for i, s in enumerate(specs):
            if s.stride not in (1, 2) or (s.stride == 2 and i != 0):
                raise NotImplementedError("only the first layer may stride (spectrogram-input stack, net.py:317)")
            hidden_ok = s.activation in ("relu", "elu") if i < len(specs) - 1 else s.activation == "softmax"
            if not hidden_ok:
                raise NotImplementedError(
                    "HIP path supports relu/elu hidden layers and a softmax output layer (got {!r} on {})".format(
                        s.activation, s.name))
        if specs[-1].cout != grapheme_set_size:
            raise ValueError("output layer width must equal the grapheme set size")
        self.plans = []
        off = 0
        # pair view of the striding layer = 2*cin_pad channels; the wgrad tile needs that to be a multiple of 128
        cin_pad = _round_up(specs[0].cin, 64)
        for i, s in enumerate(specs):
            cout_pad = _round_up(s.cout, 128)
            w_off = off
            off += s.kernel_size * cin_pad * cout_pad
            b_off = off
            off += cout_pad
            self.plans.append(LayerPlan(i, s, cin_pad, cout_pad, w_off, b_off))
            cin_pad = cout_pad
        self.param_numel = off
        # runs of >= 2 consecutive stride-1 layers with identical padded geometry (net.py:321-323: inner_conv_1..7)
        self.runs = []
        i = 1
        while i < len(self.plans):
            j = i
            key = lambda q: (q.spec.kernel_size, q.spec.stride, q.cin_pad, q.cout_pad)  # noqa: E731
            while j + 1 < len(self.plans) and key(self.plans[j + 1]) == key(self.plans[i]) and \
                    self.plans[i].cin_pad == self.plans[i].cout_pad:
                j += 1
            if j > i:
                self.runs.append((i, j))
            i = j + 1
        self.group_wgrad = True
        dev = self.device
        self.params = torch.zeros((off,), dtype=torch.float32, device=dev)
        self.grads = torch.zeros((off,), dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros((off,), dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros((off,), dtype=torch.float32, device=dev)
        self.w_fwd = [torch.zeros((p.cout_pad, p.spec.kernel_size, p.cin_pad), dtype=self.torch_dtype, device=dev)
                      for p in self.plans]
        self.w_dgrad = [torch.zeros((p.cin_pad, p.spec.kernel_size, p.cout_pad), dtype=self.torch_dtype, device=dev)
                        if p.index > 0 else None for p in self.plans]
        self._packed_dirty = True
        self._buffers = {}
        self.max_cached_shapes = 4  # (batch, frames rounded up to 512) geometries kept allocated
        # Keras Dropout(rate) in front of every conv except the last three (net.py:301-303, 326-330); training steps
        # only.  None = off (every reference configuration).
        self.dropout_rate = None
        self.dropout_seed = 0
        self._dropout_steps = 0
        self.cur = None
        self.timeline = None
        self.kernel_timeline = None  # (set of tags, list of (tag, start, stop)): see _launch
        self._side_stream = None
        self.overlap_wgrad = False
        self.defer_bias_grads = os.environ.get("SL_DEFER_BGRAD", "1") != "0"  # A/B knob, see backward()
        # "ones channel": the last padded output channel of every hidden layer carries the constant 1 (bias 1, zero weights),
        # so the next layer's weight-gradient GEMM -- which multiplies the padding through anyway -- leaves that layer's
        # BIAS gradient in row cin_pad - 1 of dW (sl_bias_grad_from_wgrad, include/speechless_hip.h): ten of the eleven
        # sl_bias_grad passes over g (two launches each, on the side stream, 0.07 ms of the config-3 step by taking
        # bandwidth and power from the GEMMs beside them) become one small launch.  SL_ONES_CHANNEL=0: the old passes.
        self.ones_channel = os.environ.get("SL_ONES_CHANNEL", "1") == "1"
        self._bgw_tables = {}
        # forward + CTC + backward of a resident step replayed from a hipGraph (one graph per batch geometry): the ~60
        # launches and 4 cross-stream hand-overs of the step cost the host ~0.7 ms of Python / ctypes per step otherwise
        self.use_graph = os.environ.get("SL_USE_GRAPH", "0") == "1"
        self.fuse_output_softmax = os.environ.get("SL_FUSE_OUTPUT", "1") != "0"  # A/B knob: sl_output_softmax
        # train_step_resident: Adam of a layer runs under the rest of backward (see backward()).  Measured on MI355X
        # (tools/step_ab.py): 2.546 ms/step either way -- the HBM-bound update slows the MFMA kernels it overlaps by as
        # much as it costs alone -- so it is off by default, which also keeps per-kernel timings clean.
        self.early_adam = False
        self.nt_cfg = {}  # optional per-launch tile configuration overrides {("fwd"|"dgrad", layer name): cfg word}
        # Launch lists: the ~60 C-ABI calls and 4 stream hand-overs of a step are recorded the first time a buffer set
        # runs them and replayed afterwards with their arguments already marshalled -- the Python around each launch
        # (tensor views of the flat parameter buffers, data_ptr() calls, geometry look-ups: ~10 us per launch) was what
        # capped the host-fed loop below the rate of the resident step.  Geometries are passed by reference and
        # re-targeted in place (_Buffers.set_length), so one list serves every batch length of a buffer set.
        # runs of identical layers (inner_conv_1..7) as ONE launch with the activations kept in LDS (sl_conv1d_chain):
        # forward 124 us against 7 x 23.6 = 165 us of single launches, input gradients 126 against 185 (config 3;
        # DESIGN.md section 3.1).  SL_CHAIN=0 restores the single launches (A/B measurements).
        self.use_chain = os.environ.get("SL_CHAIN", "1") == "1"
        self._chain_tables = {}
        self.use_launch_lists = os.environ.get("SL_LAUNCH_LISTS", "1") != "0"
        self._rec = None
        self._adam_tables = {}

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _launch(self, tag, name, *args):
        """One C-ABI call.  With self.timeline set (a list), brackets it with HIP events on the launch stream so
        that bench.py can read per-kernel durations live (tag = logical kernel instance, e.g. 'fwd:big_conv_1')."""
        if self.kernel_timeline is not None and tag in self.kernel_timeline[0]:
            # events immediately around the MAIN kernel of this call (sl_profile_next_kernel), none anywhere else: the
            # step runs as in the timed region and the duration is what rocprofv3 reports for that kernel
            start = torch.cuda.Event(enable_timing=True)
            stop = torch.cuda.Event(enable_timing=True)
            start.record()  # creates the HIP events; the library records them again around the kernel
            stop.record()
            self.lib.call("sl_profile_next_kernel", start.cuda_event, stop.cuda_event)
            self.lib.call(name, *args)
            self.kernel_timeline[1].append((tag, start, stop))
            return
        if self.timeline is None:
            self.lib.call(name, *args)
            if self._rec is not None:  # building a launch list (see _replay): the raw entry point and its arguments
                self._rec.append((0, self.lib.raw(name), args, name))
            return
        start = torch.cuda.Event(enable_timing=True)
        stop = torch.cuda.Event(enable_timing=True)
        start.record()
        self.lib.call(name, *args)
        stop.record()
        self.timeline.append((tag, start, stop))

    def _hand_over(self, src, dst):
        """dst waits for everything enqueued on src so far (event record on src + wait on dst)."""
        ev = torch.cuda.Event()
        ev.record(src)
        dst.wait_event(ev)
        if self._rec is not None:
            self._rec.append((1, ev, src, dst))

    def _replay(self, ops, callback=None):
        for op in ops:
            kind = op[0]
            if kind == 0:
                rc = op[1](*op[2])
                if rc != 0:
                    raise _lib.HipLibraryError("{} failed with status {}: {}".format(op[3], rc, self.lib.last_error()))
            elif kind == 1:
                op[1].record(op[2])
                op[3].wait_event(op[1])
            else:
                callback(op[1])

    def _launch_list(self, buf, key):
        """The recorded launch list of `key` for this buffer set, or None (then the caller runs eagerly; with
        self._rec set by start_recording() that run records the list)."""
        if not self.use_launch_lists or self.timeline is not None or self.kernel_timeline is not None or \
                self._rec is not None:
            return None
        return buf.launch_lists.get(key)

    def buffers(self, batch, t_in):
        """Buffers for batches of `batch` utterances padded to t_in frames: one set per (batch, output frames rounded
        up to TIME_TILE), re-targeted at t_in (see _Buffers.set_length)."""
        p0 = self.plans[0]
        t_out, _, _ = same_padding(t_in, p0.spec.kernel_size, p0.spec.stride)
        key = (batch, _round_up(max(t_out, 1), TIME_TILE))
        buf = self._buffers.get(key)
        if buf is None:
            if len(self._buffers) >= self.max_cached_shapes:  # bound HBM use when many batch shapes are seen
                self._buffers.pop(next(iter(self._buffers)))
            buf = _Buffers(self, batch, key[1])
            self._buffers[key] = buf
        buf.set_length(self, t_in)
        return buf

    def layer_param_views(self, tensor, plan):
        k = plan.spec.kernel_size
        w = tensor[plan.w_off: plan.w_off + plan.w_numel].view(k, plan.cin_pad, plan.cout_pad)
        b = tensor[plan.b_off: plan.b_off + plan.cout_pad]
        return w, b

    def bucket_ranges(self):
        """Flat-gradient ranges in the order they become ready during backward: the three output layers first
        (81 % of the bytes), then the rest."""
        n = len(self.plans)
        split = max(n - 3, self.frozen_layer_count)
        ranges = [(self.plans[split].w_off, self.param_numel)]
        if split > self.frozen_layer_count:
            ranges.append((self.plans[self.frozen_layer_count].w_off, self.plans[split].w_off))
        return ranges, split

    # ------------------------------------------------------------------ weights
    def set_weights(self, weights):
        """weights: [(W (k,Cin,Cout), b (Cout,))] numpy, Keras layout."""
        assert len(weights) == len(self.plans)
        self.params.zero_()
        for p, (w, b) in zip(self.plans, weights):
            s = p.spec
            if tuple(w.shape) != (s.kernel_size, s.cin, s.cout) or tuple(b.shape) != (s.cout,):
                raise ValueError("weights of layer {} have shape {} / {}".format(s.name, w.shape, b.shape))
            wv, bv = self.layer_param_views(self.params, p)
            wv[:, :s.cin, :s.cout] = torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32)).to(self.device)
            bv[:s.cout] = torch.as_tensor(np.ascontiguousarray(b, dtype=np.float32)).to(self.device)
            if self._has_ones_output(p):
                bv[p.cout_pad - 1] = 1.0  # relu(0 * x + 1) = elu(1) = 1: the ones channel (see self.ones_channel)
        self._packed_dirty = True

    def _has_ones_output(self, plan):
        """hidden layer whose output has channel padding: its last padded channel is the constant 1"""
        return self.ones_channel and plan.index < len(self.plans) - 1 and plan.cout_pad > plan.spec.cout

    def _ones_input_layers(self, first):
        """trainable layers whose input carries a ones channel: their bias gradient is row cin_pad - 1 of dW"""
        return [i for i in range(max(first, 1), len(self.plans)) if self._has_ones_output(self.plans[i - 1])]

    def _bias_grads_from_wgrad(self, layers, copy, stream):
        """One sl_bias_grad_from_wgrad launch for `layers` (their weight gradients are complete on `stream`)."""
        layers = list(layers)
        for lo in range(0, len(layers), 16):
            chunk = tuple(layers[lo:lo + 16])
            table = self._bgw_tables.get(chunk)
            if table is None:
                table = (_lib.BgwLayer * len(chunk))()
                for entry, i in zip(table, chunk):
                    q = self.plans[i]
                    entry.w_off, entry.b_off = q.w_off, q.b_off
                    entry.k, entry.cin_pad, entry.cout_pad, entry.tap = q.spec.kernel_size, q.cin_pad, q.cout_pad, q.pad_left
                self._bgw_tables[chunk] = table
            self._launch("bgrad_from_wgrad:{}..{}".format(self.plans[chunk[0]].spec.name, self.plans[chunk[-1]].spec.name),
                         "sl_bias_grad_from_wgrad", self.grads.data_ptr(), table, len(chunk), 1 if copy else 0,
                         stream.cuda_stream)

    def _unpad(self, tensor):
        out = []
        for p in self.plans:
            s = p.spec
            wv, bv = self.layer_param_views(tensor, p)
            out.append((wv[:, :s.cin, :s.cout].contiguous().cpu().numpy(), bv[:s.cout].contiguous().cpu().numpy()))
        return out

    def get_weights(self):
        return self._unpad(self.params)

    def get_gradients(self):
        return self._unpad(self.grads)

    def get_optimizer_state(self):
        """Adam moments in the Keras layout (per layer (m_W, m_b), (v_W, v_b)), the step count and the dropout step
        counter: everything beyond the weights that the next step depends on."""
        return {"m": self._unpad(self.adam_m), "v": self._unpad(self.adam_v), "iterations": int(self.adam_iterations),
                "dropout_steps": int(self._dropout_steps)}

    def set_optimizer_state(self, state):
        for name, flat in (("m", self.adam_m), ("v", self.adam_v)):
            flat.zero_()
            for p, (w, b) in zip(self.plans, state[name]):
                s = p.spec
                if tuple(w.shape) != (s.kernel_size, s.cin, s.cout) or tuple(b.shape) != (s.cout,):
                    raise ValueError("optimizer state of layer {} has shape {} / {}".format(s.name, w.shape, b.shape))
                wv, bv = self.layer_param_views(flat, p)
                wv[:, :s.cin, :s.cout] = torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32)).to(self.device)
                bv[:s.cout] = torch.as_tensor(np.ascontiguousarray(b, dtype=np.float32)).to(self.device)
        self.adam_iterations = int(state["iterations"])
        self._dropout_steps = int(state.get("dropout_steps", 0))

    def repack_weights(self):
        st = self._stream()
        for p in self.plans:
            wv, _ = self.layer_param_views(self.params, p)
            wd = self.w_dgrad[p.index]
            self._launch("pack:" + p.spec.name, "sl_pack_weights", wv.data_ptr(), self.w_fwd[p.index].data_ptr(),
                          wd.data_ptr() if wd is not None else None, p.spec.kernel_size, p.cin_pad, p.cout_pad,
                          self.dtype_code, st)
        self._packed_dirty = False

    # ------------------------------------------------------------------ forward
    def load_input(self, input_batch):
        """input_batch: (B,T,F) numpy (any float dtype; the reference packs float64, net.py:583) or a float32 torch
        tensor already resident in HBM."""
        if isinstance(input_batch, np.ndarray):
            src = torch.from_numpy(np.ascontiguousarray(input_batch, dtype=np.float32)).to(self.device,
                                                                                            non_blocking=True)
        else:
            src = input_batch.to(device=self.device, dtype=torch.float32).contiguous()
        batch, t_in, f = src.shape
        if f != self.specs[0].cin:
            raise ValueError("input has {} bins per frame, the net expects {}".format(f, self.specs[0].cin))
        buf = self.buffers(batch, t_in)
        p0 = self.plans[0]
        self._launch("pack_input", "sl_pack_input", src.data_ptr(), buf.x0.data_ptr(), batch, t_in, f, p0.pad_left, p0.cin_pad,
                      buf.rows0 * p0.cin_pad, self.dtype_code, self._stream())
        self.cur = buf
        self._src_keepalive = src
        return buf

    def _chain_table(self, kind, layers, buf):
        """ctypes pointer tables of sl_conv1d_chain for the given layers (plan indices in launch order), cached per
        (buffer set, kind).  kind 'fwd': ys = activations, ws = forward operands, biases; kind 'dgrad': layer i of the list
        is the input gradient of plan i: ys = g[i - 1], ws = dgrad operands, masks = y[i - 1]."""
        key = (kind, tuple(layers), id(buf))
        hit = self._chain_tables.get(key)
        if hit is not None and hit[0] is buf:
            return hit[1]
        n = len(layers)
        arr = ctypes.c_void_p * n
        if kind == "fwd":
            ys = arr(*[buf.y[i].data_ptr() for i in layers])
            ws = arr(*[self.w_fwd[i].data_ptr() for i in layers])
            aux = arr(*[self.layer_param_views(self.params, self.plans[i])[1].data_ptr() for i in layers])
        else:
            ys = arr(*[buf.g[i - 1].data_ptr() for i in layers])
            ws = arr(*[self.w_dgrad[i].data_ptr() for i in layers])
            aux = arr(*[buf.y[i - 1].data_ptr() for i in layers])
        if len(self._chain_tables) > 64:
            self._chain_tables.clear()
        self._chain_tables[key] = (buf, (ys, ws, aux))
        return ys, ws, aux

    def _chain_ok(self, buf, layers):
        """the fused kernel takes a run of ReLU layers of the bf16 path whose geometry it supports"""
        if not self.use_chain or self.dtype != "bf16" or len(layers) < 2:
            return False
        if any(self.specs[i].activation != "relu" for i in layers):
            return False
        return bool(self.lib.raw("sl_conv1d_chain_supported")(ctypes.byref(buf.fwd_geom[layers[0]]), len(layers),
                                                              self.dtype_code))

    def _dropout_layers(self):
        """Indices of the layers with a Dropout in front of them (all but the last three, net.py:326-330)."""
        return range(0, max(len(self.plans) - 3, 0))

    def forward(self, input_batch=None, training=False):
        """Runs the 11 conv layers + softmax.  Returns the probability tensor (B,T',K) fp32 in HBM.
        training=True applies dropout (if self.dropout_rate) to the inputs of the first n-3 layers: the packed input
        goes through sl_dropout into a second buffer, every other activation is dropped in place right after the
        layer that produced it (so the stored activation is the post-dropout one the backward pass needs)."""
        buf = self.load_input(input_batch) if input_batch is not None else self.cur
        if self._packed_dirty:
            self.repack_weights()
        st = self._stream()
        n = len(self.plans)
        rate = self.dropout_rate if training else None
        buf.dropped = bool(rate)
        fuse_out = self.fuse_output_softmax and self.dtype == "bf16" and bool(self.lib.raw("sl_output_softmax_supported")(
            ctypes.byref(buf.fwd_geom[n - 1]), self.grapheme_set_size, self.dtype_code))
        # launch list (no dropout): everything below takes its frame count from the geometries, except the unfused
        # softmax, which gets it by value -> then the list is per length
        key = None if rate else ("fwd", st, fuse_out, self.use_chain, tuple(sorted(self.nt_cfg.items())),
                                 None if fuse_out else buf.t_out)
        ops = self._launch_list(buf, key) if key is not None else None
        if ops is not None:
            self._replay(ops)
            return buf.probs
        record = key is not None and self.use_launch_lists and self.timeline is None and \
            self.kernel_timeline is None and self._rec is None
        if not record:
            return self._forward_eager(buf, rate, fuse_out, st)
        self._rec = []
        try:
            probs = self._forward_eager(buf, rate, fuse_out, st)
            buf.launch_lists[key] = self._rec
            return probs
        finally:
            self._rec = None

    def _forward_eager(self, buf, rate, fuse_out, st):
        n = len(self.plans)
        x = buf.x0
        if rate:
            self._dropout_steps += 1
            seed0 = (self.dropout_seed * 1000003 + self._dropout_steps) * 64
            buf.dropout_seed0 = seed0  # ELU layers: backward recomputes the keep decisions (sl_elu_dropout_backward)
            if buf.x0_dropped is None:
                buf.x0_dropped = torch.zeros_like(buf.x0)
            self._launch("dropout:input", "sl_dropout", buf.x0.data_ptr(), buf.x0_dropped.data_ptr(), buf.x0.numel(),
                         self.dtype_code, rate, seed0, st)
            x = buf.x0_dropped
        chained = {}  # first layer of a run -> the run, when it goes through sl_conv1d_chain
        if not rate:
            for (s0, e0) in self.runs:
                if e0 < n - 1 and self._chain_ok(buf, list(range(s0, e0 + 1))) and not any(
                        ("fwd", self.specs[i].name) in self.nt_cfg for i in range(s0, e0 + 1)):
                    chained[s0] = list(range(s0, e0 + 1))
        skip_until = -1
        for p in self.plans:
            last = p.index == n - 1
            if p.index <= skip_until:
                x = buf.y[p.index]
                continue
            if p.index in chained:
                layers = chained[p.index]
                ys, ws, biases = self._chain_table("fwd", layers, buf)
                self._launch("fwd:{}..{}".format(self.specs[layers[0]].name, self.specs[layers[-1]].name),
                             "sl_conv1d_chain", x.data_ptr(), ys, ws, biases, None, ctypes.byref(buf.fwd_geom[p.index]),
                             len(layers), _lib.EPI_BIAS_RELU, self.dtype_code, st)
                skip_until = layers[-1]
                x = buf.y[p.index]
                continue
            y = buf.logits if last else buf.y[p.index]
            _, bias = self.layer_param_views(self.params, p)
            if last and fuse_out:  # output layer + softmax + log(p + eps) re-normalisation in one launch
                self._launch("fwd:" + p.spec.name, "sl_output_softmax", x.data_ptr(), self.w_fwd[p.index].data_ptr(),
                             bias.data_ptr(), buf.probs.data_ptr(), buf.logq.data_ptr(), None,
                             ctypes.byref(buf.fwd_geom[p.index]), self.grapheme_set_size, p.cout_pad,
                             buf.tt_pad * p.cout_pad, self.ctc_epsilon, self.dtype_code, st)
                return buf.probs
            self._launch("fwd:" + p.spec.name, "sl_conv1d_nt", x.data_ptr(), self.w_fwd[p.index].data_ptr(), bias.data_ptr(), None,
                          y.data_ptr(), ctypes.byref(buf.fwd_geom[p.index]),
                          _lib.EPI_BIAS if last else
                          (_lib.EPI_BIAS_ELU if p.spec.activation == "elu" else _lib.EPI_BIAS_RELU),
                          self.dtype_code, 1 if last else 0,
                          self.nt_cfg.get(("fwd", p.spec.name), 0), buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
            if rate and (p.index + 1) in self._dropout_layers():
                self._launch("dropout:" + p.spec.name, "sl_dropout", y.data_ptr(), y.data_ptr(), y.numel(),
                             self.dtype_code, rate, seed0 + p.index + 1, st)
            x = y
        self._launch("softmax", "sl_softmax_logq", buf.logits.data_ptr(), buf.probs.data_ptr(), buf.logq.data_ptr(), buf.batch,
                      buf.t_out, self.grapheme_set_size, self.plans[-1].cout_pad, buf.tt_pad * self.plans[-1].cout_pad,
                      self.ctc_epsilon, st)
        return buf.probs

    def set_input_lengths(self, prediction_lengths):
        buf = self.cur
        buf.input_len.copy_(torch.as_tensor(np.asarray(prediction_lengths, dtype=np.int32).reshape(-1)),
                            non_blocking=True)

    def greedy_decode(self, prediction_lengths=None):
        """Greedy CTC decode of the current probabilities.  Returns (list of index lists, frame argmax (B,T') numpy)."""
        buf = self.cur
        if prediction_lengths is not None:
            self.set_input_lengths(prediction_lengths)
        k = self.grapheme_set_size
        self._launch("decode", "sl_greedy_decode", buf.probs.data_ptr(), buf.input_len.data_ptr(), buf.decoded.data_ptr(),
                      buf.decoded_len.data_ptr(), buf.frame_argmax.data_ptr(), buf.batch, buf.t_out, k, k - 1,
                      self._stream())
        dec = buf.decoded.cpu().numpy()
        lens = buf.decoded_len.cpu().numpy()
        return [list(map(int, dec[i, :lens[i]])) for i in range(buf.batch)], buf.frame_argmax.cpu().numpy()

    # ------------------------------------------------------------------ loss + backward
    def set_labels(self, label_batch, label_lengths, prediction_lengths):
        """label_batch: int (B,Lmax) padded with anything (reference pads -1); lengths: (B,) or (B,1)."""
        buf = self.cur
        buf.ensure_backward(self)
        labels = np.asarray(label_batch, dtype=np.int32)
        lab_len = np.asarray(label_lengths, dtype=np.int32).reshape(-1)
        if labels.ndim != 2 or labels.shape[0] != buf.batch:
            raise ValueError("label batch must be (B, Lmax)")
        k = self.grapheme_set_size
        for i in range(buf.batch):
            row = labels[i, :lab_len[i]]
            if row.size and (row.min() < 0 or row.max() >= k - 1):
                raise ValueError("label {} holds an index outside [0, {}) (blank is {})".format(i, k - 1, k - 1))
        l_max = max(int(labels.shape[1]), 1)
        if labels.shape[1] == 0:
            labels = np.zeros((buf.batch, 1), dtype=np.int32)
        buf.ensure_ctc(self, l_max)
        if buf.labels.shape[1] != labels.shape[1]:
            buf.labels = torch.zeros((buf.batch, labels.shape[1]), dtype=torch.int32, device=self.device)
        buf.labels.copy_(torch.from_numpy(np.ascontiguousarray(labels)), non_blocking=True)
        buf.label_len.copy_(torch.from_numpy(lab_len), non_blocking=True)
        self.set_input_lengths(prediction_lengths)

    def set_labels_resident(self, labels_dev, label_len_dev, input_len_dev):
        """set_labels for int32 tensors that already live in HBM (the staged input pipeline copies them on its copy
        stream): labels (B, Lmax >= 1) with every entry of row b below label_len[b] in [0, K-1), lengths (B,).  The
        tensors are used in place -- the caller keeps them alive and unchanged until the step's kernels have run."""
        buf = self.cur
        buf.ensure_backward(self)
        if labels_dev.dim() != 2 or labels_dev.shape[0] != buf.batch or labels_dev.shape[1] < 1:
            raise ValueError("label batch must be (B, Lmax >= 1)")
        buf.ensure_ctc(self, int(labels_dev.shape[1]))
        buf.labels = labels_dev
        buf.label_len = label_len_dev
        buf.input_len = input_len_dev

    def ctc(self, grad_scale=None, with_grad=True):
        """Per-utterance CTC loss of the current probabilities (tensor (B,) in HBM) and, into g[last], the gradient
        w.r.t. the output_conv logits of grad_scale * sum_b loss_b (default 1/B: Keras' mean, net.py:389)."""
        buf = self.cur
        buf.ensure_backward(self)
        last = len(self.plans) - 1
        if grad_scale is None:
            grad_scale = 1.0 / buf.batch
        l_max = buf.labels.shape[1]
        self._launch("ctc", "sl_ctc_loss_grad", buf.probs.data_ptr(), buf.logq.data_ptr(), buf.labels.data_ptr(),
                      buf.label_len.data_ptr(), buf.input_len.data_ptr(), buf.loss.data_ptr(), buf.g[last].data_ptr(),
                      buf.batch, buf.t_out, self.grapheme_set_size, l_max, HALO, self.plans[last].cout_pad,
                      buf.rows * self.plans[last].cout_pad, self.dtype_code, self.ctc_epsilon, grad_scale,
                      buf.ctc_ws.data_ptr(), buf.ctc_ws.numel(), self._stream())
        return buf.loss

    def backward(self, on_bucket_ready=None, early_adam=False, reducer=None):
        """wgrad / bias-grad / dgrad for every trainable layer, output layer first.

        early_adam: the fused Adam + operand repack of a layer (HBM-bound, 0.2 ms per step in total, 0.13 ms of it in
        the three output layers whose gradients are complete FIRST) is enqueued as soon as that layer's wgrad, bias
        grad and dgrad are: on the side stream (single GPU), or on the reducer's communication stream behind the
        all-reduce of the layer's gradient bucket (data parallel).  It then runs underneath the MFMA-bound kernels of
        the layers below instead of after them.  Results are identical to backward() followed by adam_step().

        The HBM-bound bias gradients (they only stream g[i] once) run on a SIDE stream underneath the MFMA-bound
        wgrad/dgrad kernels instead of in front of them.  With self.overlap_wgrad the weight gradients move to the side
        stream as well, under the critical dgrad chain of the MAIN stream (they only need g[i] and the saved
        activation; the short layers put at most one work-group on a CU, so kernels of different layers co-reside).
        That is worth +1.3 % at B=32 but makes per-kernel event/rocprof durations overlap, so it is off by default.
        on_bucket_ready(i) is called once the launches that complete gradient bucket i (bucket_ranges) are enqueued."""
        buf = self.cur
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)  # (ROCm offers no priority below the default)
        side = self._side_stream
        # launch list: the default schedule only (no dropout masks to rescale, no early Adam, no second wgrad stream)
        plain = not (early_adam or self.overlap_wgrad or buf.dropped or os.environ.get("SL_DEFER_BGRAD") == "skip")
        key = ("bwd", main.cuda_stream, on_bucket_ready is not None, self.defer_bias_grads, self.ones_channel,
               self.frozen_layer_count,
               self.group_wgrad, self.use_chain, tuple(sorted(self.nt_cfg.items()))) if plain else None
        ops = self._launch_list(buf, key) if key is not None else None
        if ops is not None:
            self._replay(ops, on_bucket_ready)
            return
        record = key is not None and self.use_launch_lists and self.timeline is None and \
            self.kernel_timeline is None and self._rec is None
        if not record:
            self._backward_eager(buf, main, side, on_bucket_ready, early_adam, reducer)
            return
        self._rec = []
        try:
            self._backward_eager(buf, main, side, on_bucket_ready, early_adam, reducer)
            buf.launch_lists[key] = self._rec
        finally:
            self._rec = None

    def _backward_eager(self, buf, main, side, on_bucket_ready, early_adam, reducer):
        first = self.frozen_layer_count
        _, split = self.bucket_ranges()
        grouped = {}  # layer index -> (lo, hi) of the run whose weight gradients are computed in one grouped launch
        if self.group_wgrad and self.dtype == "bf16" and not self.overlap_wgrad:
            for (s0, e0) in self.runs:
                lo = max(s0, first)
                if e0 > lo and not (lo <= split <= e0):  # keep the bucket boundary simple
                    for q in range(lo, e0 + 1):
                        grouped[q] = (lo, e0)

        def join_side():
            self._hand_over(side, main)

        def bucket_ready(index):
            on_bucket_ready(index)
            if self._rec is not None:
                self._rec.append((2, index))

        if early_adam:
            if self._packed_dirty:
                self.repack_weights()
            self.adam_iterations += 1
        dp = reducer is not None and (reducer.world_size > 1 or reducer.force)
        bucket_layers = []  # data parallel: layers of the gradient bucket being completed

        def adam_after_this_layer(layers):
            """layers: their wgrad / dgrad launches are all enqueued on MAIN (and their bias grads on SIDE)."""
            issued = torch.cuda.Event()
            issued.record(main)
            if not dp:
                with torch.cuda.stream(side):
                    side.wait_event(issued)
                    self._adam_layers(layers, side.cuda_stream)
            else:
                bucket_layers.extend(layers)

        # input gradients of a run of identical ReLU layers in one launch (sl_conv1d_chain): keyed by the TOP layer
        dchain, dchain_skip = {}, set()
        if not buf.dropped:
            for (s0, e0) in self.runs:
                lo_d = max(s0, first + 1)
                layers = list(range(e0, lo_d - 1, -1))
                if len(layers) >= 2 and self.use_chain and self.dtype == "bf16" and \
                        all(self.specs[i - 1].activation == "relu" for i in layers) and \
                        not any(("dgrad", self.specs[i].name) in self.nt_cfg for i in layers) and \
                        bool(self.lib.raw("sl_conv1d_chain_supported")(ctypes.byref(buf.dgrad_geom[e0]), len(layers),
                                                                       self.dtype_code)):
                    dchain[e0] = layers
                    dchain_skip.update(layers[1:])
        # bias gradients out of the weight-gradient GEMM (self.ones_channel): which layers, and whether the row holds the
        # bias gradient (the ones were not touched by dropout) or only has to be zeroed before the optimizer sees it
        ones_in = self._ones_input_layers(first)
        simple = not (early_adam or self.overlap_wgrad)
        ones_db = set(ones_in) if (simple and not buf.dropped) else set()
        # (those two need every layer's bias gradient at once; defer_bias_grads = False restores one hand-over per layer)
        defer = self.defer_bias_grads and not (early_adam or self.overlap_wgrad)
        pending, pending_bytes = [], 0  # layers whose bias-gradient launch is still owed to the side stream
        for p in reversed(self.plans[first:]):
            i = p.index
            x = (buf.x0_dropped if buf.dropped else buf.x0) if i == 0 else buf.y[i - 1]
            dw, db = self.layer_param_views(self.grads, p)
            # bias gradient of layer i: deferred until enough work has piled up for one hand-over to the side stream.
            # Every hand-over is an event record on MAIN, and the record costs MAIN ~6 us of pipeline drain
            # (profiles/r01j: 11 records = the only gaps in the step's timeline); g[i] stays intact until the next step,
            # so the small layers' bias gradients can wait for a common hand-over.
            if os.environ.get("SL_DEFER_BGRAD") != "skip" and i not in ones_db:  # ("skip": timing experiment only)
                pending.append(i)
                pending_bytes += buf.g[i].numel() * buf.g[i].element_size()
            if pending and (not defer or pending_bytes >= (128 << 20) or i <= first + 1 or
                            (on_bucket_ready is not None and i == split)):
                # g[j], j in pending (CTC gradient or a previous dgrad) are complete at this point of MAIN
                self._hand_over(main, side)
                wgrad_stream = side if self.overlap_wgrad else main
                with torch.cuda.stream(side):
                    if self.overlap_wgrad:
                        self._launch("wgrad:" + p.spec.name, "sl_conv1d_wgrad", x.data_ptr(), buf.g[i].data_ptr(),
                                     dw.data_ptr(), ctypes.byref(buf.wgrad_geom[i]), self.dtype_code,
                                     self.nt_cfg.get(("wgrad", p.spec.name), 0), buf.wgrad_ws.data_ptr(),
                                     buf.wgrad_ws.numel(), wgrad_stream.cuda_stream)
                        if i in ones_in:
                            self._bias_grads_from_wgrad([i], False, wgrad_stream)
                    for j in pending:
                        _, db_j = self.layer_param_views(self.grads, self.plans[j])
                        self._launch("bgrad:" + self.plans[j].spec.name, "sl_bias_grad", buf.g[j].data_ptr(),
                                     db_j.data_ptr(), ctypes.byref(buf.wgrad_geom[j]), self.dtype_code,
                                     buf.bias_ws.data_ptr(), buf.bias_ws.numel(), side.cuda_stream)
                    if self.overlap_wgrad and on_bucket_ready is not None and i == split:
                        bucket_ready(0)
                del pending[:]
                pending_bytes = 0
            if i in grouped:
                lo, hi = grouped[i]
                if i == lo:  # every g[lo..hi] is complete now: one launch for the whole run
                    plo = self.plans[lo]
                    dw_lo, _ = self.layer_param_views(self.grads, plo)
                    stride_elems = buf.batch * buf.rows * plo.cin_pad
                    self._launch("wgrad:{}..{}".format(plo.spec.name, self.plans[hi].spec.name),
                                 "sl_conv1d_wgrad_grouped", buf.y[lo - 1].data_ptr(), buf.g[lo].data_ptr(),
                                 dw_lo.data_ptr(), ctypes.byref(buf.wgrad_geom[lo]), hi - lo + 1, stride_elems,
                                 stride_elems, plo.w_numel + plo.cout_pad, 0, buf.wgrad_ws.data_ptr(),
                                 buf.wgrad_ws.numel(), main.cuda_stream)
                    if not simple:
                        self._bias_grads_from_wgrad([j for j in range(lo, hi + 1) if j in ones_in], False, main)
            elif not self.overlap_wgrad:
                self._launch("wgrad:" + p.spec.name, "sl_conv1d_wgrad", x.data_ptr(), buf.g[i].data_ptr(),
                             dw.data_ptr(), ctypes.byref(buf.wgrad_geom[i]), self.dtype_code,
                             self.nt_cfg.get(("wgrad", p.spec.name), 0), buf.wgrad_ws.data_ptr(),
                             buf.wgrad_ws.numel(), main.cuda_stream)
                if not simple and i in ones_in:
                    self._bias_grads_from_wgrad([i], False, main)
                if on_bucket_ready is not None and i == split:
                    if simple:
                        self._bias_grads_from_wgrad([j for j in ones_in if j >= split], bool(ones_db), main)
                    join_side()
                    bucket_ready(0)
            if i in dchain:
                layers = dchain[i]
                ys, ws, masks = self._chain_table("dgrad", layers, buf)
                self._launch("dgrad:{}..{}".format(self.specs[layers[0]].name, self.specs[layers[-1]].name),
                             "sl_conv1d_chain", buf.g[i].data_ptr(), ys, ws, None, masks,
                             ctypes.byref(buf.dgrad_geom[i]), len(layers), _lib.EPI_RELU_MASK, self.dtype_code,
                             main.cuda_stream)
            elif i in dchain_skip:
                pass
            elif i > first:
                elu = self.specs[i - 1].activation == "elu"
                elu_dropped = elu and buf.dropped and i in self._dropout_layers()
                self._launch("dgrad:" + p.spec.name, "sl_conv1d_nt", buf.g[i].data_ptr(), self.w_dgrad[i].data_ptr(),
                             None, None if elu_dropped else buf.y[i - 1].data_ptr(), buf.g[i - 1].data_ptr(),
                             ctypes.byref(buf.dgrad_geom[i]),
                             _lib.EPI_NONE if elu_dropped else (_lib.EPI_ELU_MASK if elu else _lib.EPI_RELU_MASK),
                             self.dtype_code, 0, self.nt_cfg.get(("dgrad", p.spec.name), 0),
                             buf.nt_ws.data_ptr(), buf.nt_ws.numel(), main.cuda_stream)
                if elu_dropped:
                    # a stored zero is ambiguous behind an ELU (dropped, or elu(z) == 0): both factors of the chain rule
                    # in one elementwise pass that recomputes the keep decisions from the step's seed
                    self._launch("dropout_elu_bwd:" + p.spec.name, "sl_elu_dropout_backward", buf.g[i - 1].data_ptr(),
                                 buf.y[i - 1].data_ptr(), buf.g[i - 1].numel(), self.dtype_code, self.dropout_rate,
                                 buf.dropout_seed0 + i, main.cuda_stream)
                elif buf.dropped and i in self._dropout_layers():
                    # the dgrad epilogue's mask (stored activation > 0) already applied the keep mask: the stored
                    # activation is post-dropout; what is left of d dropout / dx is the 1 / (1 - rate) factor
                    self._launch("dropout_scale:" + p.spec.name, "sl_scale", buf.g[i - 1].data_ptr(), buf.g[i - 1].numel(),
                                 self.dtype_code, 1.0 / (1.0 - self.dropout_rate), main.cuda_stream)
            if early_adam:
                if i in grouped:
                    if i == grouped[i][0]:
                        adam_after_this_layer(list(range(grouped[i][0], grouped[i][1] + 1)))
                else:
                    adam_after_this_layer([i])
                if dp and i == split:  # bucket 0 is being reduced on the communication stream: Adam goes behind it
                    reducer.run_after_reduce(lambda st, ls=list(bucket_layers): self._adam_layers(ls, st))
                    del bucket_layers[:]
        if self.overlap_wgrad:
            if on_bucket_ready is not None and split > first:
                with torch.cuda.stream(side):
                    bucket_ready(1)
            join_side()
        else:
            if simple:
                rest = [j for j in ones_in if on_bucket_ready is None or j < split]
                if rest:
                    self._bias_grads_from_wgrad(rest, bool(ones_db), main)
            join_side()
            if on_bucket_ready is not None and split > first:
                bucket_ready(1)
        if early_adam and dp:
            if split > first:
                reducer.run_after_reduce(lambda st, ls=list(bucket_layers): self._adam_layers(ls, st))
            del bucket_layers[:]

    def adam_step(self, fused=True):
        """Keras-2.0 Adam on the flat fp32 masters.  fused=True: one kernel per trainable layer that applies Adam AND
        rewrites the layer's two bf16 operand copies in the same pass (no separate repack); fused=False: one flat
        elementwise launch, operands repacked lazily by the next forward()."""
        self.adam_iterations += 1
        st = self._stream()
        if not fused:
            self._launch("adam", "sl_adam_step", self.params.data_ptr(), self.grads.data_ptr(),
                         self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.param_numel, self.adam_iterations,
                         self.lr, self.beta_1, self.beta_2, self.adam_epsilon, st)
            self._packed_dirty = True
            return
        if self._packed_dirty:
            self.repack_weights()  # frozen layers keep these copies; trainable ones are rewritten below
        self._adam_layers(range(self.frozen_layer_count, len(self.plans)), st)

    def _adam_layers(self, layers, st):
        """Fused Adam + bf16 operand repack of the given layers on stream st (self.adam_iterations already counts
        this step): ONE launch for all of them (sl_adam_pack_layers), 16 layers per call at most."""
        layers = list(layers)
        for lo in range(0, len(layers), 16):
            chunk = layers[lo:lo + 16]
            table = self._adam_tables.get(tuple(chunk))
            if table is None:  # (the operand copies never move: built once per set of layers)
                table = (_lib.AdamLayer * len(chunk))()
                for entry, i in zip(table, chunk):
                    p = self.plans[i]
                    wd = self.w_dgrad[p.index]
                    entry.offset = p.w_off
                    entry.w_fwd = self.w_fwd[p.index].data_ptr()
                    entry.w_dgrad = wd.data_ptr() if wd is not None else None
                    entry.k, entry.cin_pad, entry.cout_pad = p.spec.kernel_size, p.cin_pad, p.cout_pad
                self._adam_tables[tuple(chunk)] = table
            self._launch("adam:{}..{}".format(self.plans[chunk[0]].spec.name, self.plans[chunk[-1]].spec.name),
                         "sl_adam_pack_layers", self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                         self.adam_v.data_ptr(), table, len(chunk), self.dtype_code, self.adam_iterations, self.lr,
                         self.beta_1, self.beta_2, self.adam_epsilon, st)

    def train_step(self, input_batch, label_batch, label_lengths, prediction_lengths, reducer=None):
        """One full optimisation step (forward, CTC, backward, [gradient all-reduce], Adam, weight repack).
        Returns the per-utterance loss tensor (B,) in HBM (not synchronised)."""
        self.load_input(input_batch)
        self.set_labels(label_batch, label_lengths, prediction_lengths)
        return self.train_step_resident(reducer)

    def _graph_eligible(self, reducer):
        return (self.use_graph and reducer is None and self.dtype == "bf16" and not self.dropout_rate and
                self.timeline is None and self.kernel_timeline is None and not self.early_adam and
                not self.overlap_wgrad)

    def _graph_step(self):
        """forward + CTC + backward from a captured hipGraph.  The kernels read the step's inputs through fixed
        pointers (x0 / labels / lengths of this geometry's buffers), so a replay computes on whatever was loaded into
        them; Adam stays outside the graph (its step count is a kernel argument)."""
        buf = self.cur
        # everything a launch of the step reads through a pointer or takes as an argument and that can differ between
        # two steps on this geometry: the label / length tensors are used in place (set_labels_resident) and the CTC
        # workspace grows with the longest label row seen
        key = (buf.labels.data_ptr(), buf.label_len.data_ptr(), buf.input_len.data_ptr(), int(buf.labels.shape[1]),
               buf.ctc_ws.data_ptr(), buf.t_in)
        graphs = buf.__dict__.setdefault("graphs", {})
        g = graphs.get(key)
        if g is None:
            warm = buf.__dict__.setdefault("graph_warm", {})
            if warm.get(key, 0) < 1 or self._packed_dirty:  # an eager step first: lazy allocations, clean operands
                warm[key] = warm.get(key, 0) + 1
                return None
            if len(graphs) >= 8:  # label tensors that never repeat (no slot recycling): graphs cannot pay off
                return None
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.forward(training=True)
                self.ctc(grad_scale=1.0 / buf.batch)
                self.backward()
            graphs[key] = g
        if self._packed_dirty:  # set_weights() since the capture: the operand repack is not part of the graph
            self.repack_weights()
        g.replay()
        return buf.loss

    def train_step_resident(self, reducer=None):
        """Same, with input / labels / lengths already resident in HBM (bench.py's timed region)."""
        if self._graph_eligible(reducer):
            loss = self._graph_step()
            if loss is not None:
                self.adam_step()
                return loss
        self.forward(training=True)
        world = reducer.world_size if reducer is not None else 1
        loss = self.ctc(grad_scale=1.0 / (self.cur.batch * world))
        early = self.early_adam and not self.overlap_wgrad
        if reducer is None:
            self.backward(early_adam=early)
        else:
            early = early and reducer.overlap
            self.backward(on_bucket_ready=reducer.reduce_bucket, early_adam=early, reducer=reducer)
            if not early and (reducer.world_size > 1 or reducer.force):
                # bucket 0 (the three output layers, 81 % of the bytes) finished reducing under the rest of backward:
                # update those layers while the small last bucket is still on the wire, then the rest
                _, split = self.bucket_ranges()
                first = self.frozen_layer_count
                if self._packed_dirty:
                    self.repack_weights()
                self.adam_iterations += 1
                reducer.wait_next()
                self._adam_layers(range(split, len(self.plans)), self._stream())
                reducer.wait_all()
                if split > first:
                    self._adam_layers(range(first, split), self._stream())
                return loss
            reducer.wait_all()
        if not early:
            self.adam_step()
        return loss
