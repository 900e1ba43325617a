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
from .buffers import _Buffers
from .engine_front import FrontLayerMixin
from .engine_split import SplitTopMixin
from .engine_x3 import X3Mixin
from ._hipevents import TimingEvent
from .plan import HALO, TIME_TILE, LayerPlan, LayerSpec, _round_up, same_padding, wav2letter_layer_specs  # noqa: F401


class Engine(X3Mixin, SplitTopMixin, FrontLayerMixin):
    """Forward / CTC / backward / Adam on one MI355X.  dtype 'bf16' (bf16 storage, fp32 accumulate, fp32 CTC: the
    benchmarked path), 'f32' (parity path: fp32 storage, exact-fp32 MFMA) or 'bf16x3' (the fast parity path: every value as
    hi + lo bf16 planes, three bf16 MFMA terms per product, fp32 accumulate; csrc/split3.hip)."""

    FRONT_DGRAD_ROW0 = 7          # first pair row of x0's gradient that is computed (frames start at pair row 11)
    FRONT_DGRAD_EXTRA_ROWS = 17   # pair rows computed beyond the output frames: up to row T' + 23 = the last frame's

    def __init__(self, specs, grapheme_set_size, dtype="bf16", device="cuda:0", ctc_epsilon=1e-8,
                 frozen_layer_count=0, lr=1e-4, beta_1=0.9, beta_2=0.999, adam_epsilon=1e-8, forward_only=False):
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("speechless_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                       "there is no CPU fallback for the hot path")
        self.lib = lib()
        self.device = torch.device(device)
        self.dtype = dtype
        self.planes = 1
        if dtype == "bf16":
            self.torch_dtype, self.dtype_code = torch.bfloat16, _lib.SL_BF16
        elif dtype == "f32":
            self.torch_dtype, self.dtype_code = torch.float32, _lib.SL_F32
        elif dtype == "bf16x3":
            self.torch_dtype, self.dtype_code, self.planes = torch.bfloat16, _lib.SL_BF16, 3
        elif dtype == "f16x3":
            # the plane scheme on fp16 pairs (round 6; engine_x3.py): 22 operand bits instead of 16-17 at the same three MFMA
            # terms.  fp16's range is 65504 and its precision below 2^-3 absolute, so weights and back-propagated gradients
            # are STORED multiplied by powers of two (exact) that bring typical magnitudes to ~1; the kernels divide them
            # out again (sl_conv_geom.acc_scale, the `scale` arguments of the sl_splitf16* helpers).
            self.torch_dtype, self.dtype_code, self.planes = torch.float16, _lib.SL_F16, 3
        else:
            raise ValueError("dtype must be 'bf16', 'f32', 'bf16x3' or 'f16x3'")
        self.x3_f16 = dtype == "f16x3"
        # f16x3: operand copies hold w_scale * w (glorot-uniform weights of this stack are 0.01 ... 0.05: x 64 -> ~1; representable
        # up to |w| < 1000), gradient planes hold g_scale * g (the CTC gradient of a frame is <= 1 / B: x 4096 stays below 2^12)
        self.w_scale = 64.0 if self.x3_f16 else 1.0
        self.g_scale = 4096.0 if self.x3_f16 else 1.0
        # Raw-wave input (reference net.py:310-312: `wave_conv`, 250 taps at stride 160 over the samples, in front of
        # striding_conv): the FRONT layer.  It is a GEMM over gathered sample windows (sl_wave_frames: K = 250 * Cin
        # columns per output frame, 3 GFLOP per 32 x 8 s -- nothing next to the stack) whose output lands directly in the
        # pair-view input buffer of the stack below; the eleven layers behind it, their plans, indices and launches are
        # untouched.  Internally the front plan has index len(plans) (its parameters sit at the END of the flat buffers);
        # the public order of set_weights / get_weights / get_gradients puts it first, as the reference's layer list does.
        self.front_spec = None
        self.front_frozen = False
        if specs and specs[0].stride > 2:
            self.front_spec = specs[0]
            specs = list(specs[1:])
            self.front_frozen = frozen_layer_count >= 1
            frozen_layer_count = max(frozen_layer_count - 1, 0)
            if self.front_spec.activation not in ("relu", "elu"):
                raise NotImplementedError("the raw-wave layer takes a relu / elu activation")
            if dtype == "f16x3":
                raise NotImplementedError("raw-wave input runs on 'bf16', 'f32' and 'bf16x3' (f16x3: spectrogram input)")
        self.specs = specs
        self.all_specs = ([self.front_spec] if self.front_spec is not None else []) + list(specs)
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
        # (a first layer that does not stride reads plain rows: its own padded width must be that multiple -- ADVICE r5)
        cin_pad = _round_up(specs[0].cin, 64 if specs[0].stride == 2 else 128)
        for i, s in enumerate(specs):
            cout_pad = _round_up(s.cout, 128)
            w_off = off
            off += s.kernel_size * cin_pad * cout_pad
            b_off = off
            off += cout_pad
            self.plans.append(LayerPlan(i, s, cin_pad, cout_pad, w_off, b_off))
            cin_pad = cout_pad
        self.front_plan = None
        if self.front_spec is not None:
            fs = self.front_spec
            p0 = self.plans[0]
            if fs.cout != specs[0].cin or p0.cin_pad % 128:
                raise NotImplementedError("the raw-wave layer's filters must be the next layer's inputs, padded to a multiple "
                                          "of 128 (250 -> 256)")
            k_real = fs.kernel_size * fs.cin
            gemm = LayerSpec(fs.name, 1, 1, k_real, fs.cout, fs.activation)  # the layer as the 1 x 1 GEMM it is launched as
            self.front_plan = LayerPlan(len(self.plans), gemm, _round_up(k_real, 128), p0.cin_pad, off, off + _round_up(k_real, 128) * p0.cin_pad)
            off = self.front_plan.b_off + p0.cin_pad
        self.all_plans = self.plans + ([self.front_plan] if self.front_plan is not None else [])
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
        # forward_only: an evaluation engine (forward, CTC loss, decode) over masters another engine trains -- no gradient
        # buffer, no moments (Wav2Letter.eval_engine aliases `params` to the training engine's)
        self.forward_only = bool(forward_only)
        self.grads = self.adam_m = self.adam_v = None
        if not self.forward_only:
            self.grads = torch.zeros((off,), dtype=torch.float32, device=dev)
            self.adam_m = torch.zeros((off,), dtype=torch.float32, device=dev)
            self.adam_v = torch.zeros((off,), dtype=torch.float32, device=dev)
        self._weights_set_count = 0
        pl = self.planes  # bf16x3: packed weight rows are [w_hi | w_hi | w_lo]
        self.w_fwd = [torch.zeros((p.cout_pad, p.spec.kernel_size, p.cin_pad * pl), dtype=self.torch_dtype, device=dev)
                      for p in self.all_plans]
        # (with a front layer the first layer of the stack needs its input gradient too: its dgrad operand in the pair view)
        self.w_dgrad = [torch.zeros((p.cin_pad, p.spec.kernel_size, p.cout_pad * pl), dtype=self.torch_dtype, device=dev)
                        if (0 < p.index < len(self.plans) or (p.index == 0 and self.front_plan is not None)) else None
                        for p in self.all_plans]
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
        # "ones channel": the last padded output channel of every hidden layer carries the constant 1 (bias 1, zero weights),
        # so the next layer's weight-gradient GEMM -- which multiplies the padding through anyway -- leaves that layer's
        # BIAS gradient in row cin_pad - 1 of dW (sl_bias_grad_from_wgrad, include/speechless_hip.h): ten of the eleven
        # sl_bias_grad passes over g (two launches each, on the side stream, 0.07 ms of the config-3 step by taking
        # bandwidth and power from the GEMMs beside them) become one small launch.  SL_ONES_CHANNEL=0: the old passes.
        self.ones_channel = os.environ.get("SL_ONES_CHANNEL", "1") == "1"
        self._bgw_tables = {}
        self.fuse_output_softmax = os.environ.get("SL_FUSE_OUTPUT", "1") != "0"  # A/B knob: sl_output_softmax
        # both gradients of the 1x1 output layer in one launch that reads the layer's input once (sl_conv1d_backward_1x1):
        # 0.038 ms against 0.040 + 0.032 ms of dgrad + wgrad launches at config 3.  SL_FUSE_OUTPUT_BWD=0: the two launches.
        self.fuse_output_backward = os.environ.get("SL_FUSE_OUTPUT_BWD", "1") != "0"
        # the weight gradients of the layers with few 256 x 256 tiles (the run of inner layers, striding_conv) in ONE launch
        # whose (tile, 64-frame step) space is cut into one equal range per CU (sl_conv1d_wgrad_multi) instead of a grouped
        # launch + a 128 x 128-tile launch with utterance-granular batch splits.  SL_WGRAD_MULTI=0: those launches.
        self.use_wgrad_multi = os.environ.get("SL_WGRAD_MULTI", "1") != "0"
        # ... also for a layer whose input channels are not a multiple of the 256-wide tile (257 bins: the 640-wide pair view
        # of striding_conv): its last tile starts at cin - 256 and overlaps the one before (A/B knob)
        self.multi_overlap_tiles = True
        self.small_bias_pass_on_main = os.environ.get("SL_BIAS_MAIN", "1") != "0"  # A/B knob
        self.x3_fused_epilogue = os.environ.get("SL_X3_FUSED_EPILOGUE", "1") != "0"  # bf16x3: activation + plane split in the NT epilogue
        # bf16x3, striding layer: the g_lo partial of its weight gradient against the [hi0 | hi1] window of the pair rows
        # instead of the whole row (buffers.py; A/B knob, read when a buffer set's backward geometries are built)
        self.x3_b_window = os.environ.get("SL_X3_B_WINDOW", "1") != "0"
        self.nt_cfg = {}  # optional per-launch tile configuration overrides {("fwd"|"dgrad", layer name): cfg word}
        # Launch lists: the ~60 C-ABI calls and 4 stream hand-overs of a step are recorded the first time a buffer set
        # runs them and replayed afterwards with their arguments already marshalled -- the Python around each launch
        # (tensor views of the flat parameter buffers, data_ptr() calls, geometry look-ups: ~10 us per launch) was what
        # capped the host-fed loop below the rate of the resident step.  Geometries are passed by reference and
        # re-targeted in place (_Buffers.set_length), so one list serves every batch length of a buffer set.
        # runs of identical layers (inner_conv_1..7) as ONE launch with the activations kept in LDS (sl_conv1d_chain):
        # forward 124 us against 7 x 23.6 = 165 us of single launches, input gradients 126 against 185 (config 3;
        # HISTORY.md section 3.1).  SL_CHAIN=0 restores the single launches (A/B measurements).
        self.use_chain = os.environ.get("SL_CHAIN", "1") == "1"
        self.use_launch_lists = os.environ.get("SL_LAUNCH_LISTS", "1") != "0"
        if self.planes > 1:  # the fused launches read and write single-plane bf16 tensors
            self.use_chain = self.fuse_output_softmax = self.fuse_output_backward = self.group_wgrad = False
            self._x3_bias_ws = None
        # Split top: the CTC lattice is a handful of latency-bound waves (one per utterance and direction, T' sequential
        # frames: 0.106 ms at config 3, 0.39 ms at config 5) with the rest of the chip idle, and it sits between forward and
        # backward.  The training step therefore runs its TOP by half-batches: big_conv_1 / big_conv_2 / output_conv forward of
        # half A, then A's CTC on a side stream UNDER the same layers of half B; B's CTC under the input gradients of A's top
        # layers.  Per-utterance results do not depend on the half an utterance is in; weight gradients stay whole-batch
        # launches (sl_conv1d_backward_1x1_part accumulates the output layer's).  Used where it pays (_split_parts: long
        # utterances in small batches, i.e. configuration 5); SL_SPLIT_TOP=0: the whole-batch sequence everywhere.
        self.split_top = os.environ.get("SL_SPLIT_TOP", "1") != "0"
        self.split_min_tiles = None
        # Data-parallel runs: CUs the communication kernels are expected to own while a bucket is on the wire.  The MFMA
        # kernels take a whole CU per work-group and their grids are sized to whole rounds of the chip, so backward() tells
        # the library's choosers to plan for 256 - comm_cus (sl_set_available_cus) while an exchange can be in flight -- only
        # then: forward runs with the whole chip.  0 = no hint.  Set by train_step_resident from the reducer (comm_cus).
        self.comm_cus = 0
        # Data-parallel runs, A/B knob for the first N > 1 run (SL_SPLIT_LAST_BUCKET=1): the one exchange
        # nothing covers is the last bucket {striding_conv, inner_conv_1..7} (19 MB), closed by the last kernel of backward.
        # With the flag the run of identical layers is cut at `split_last_at`: its fused input-gradient launch and the balanced
        # weight-gradient launch become two each (+2 launches per step), and the upper part's gradients (inner_conv_4..7,
        # 7 MB) close as a bucket of their own under the lower part's launches.  Off by default: on one GPU it only costs.
        self.split_last_bucket = os.environ.get("SL_SPLIT_LAST_BUCKET", "0") == "1"
        self.split_last_at = 4
        self._cu_hint_active = 0  # what sl_set_available_cus was last told by this engine's backward (restored after sizing calls)
        self._ctc_streams = None
        self._rec = None
        self._adam_tables = {}
        self._sharded_reducer = None  # the reducer of the last step, if that step ran Adam on this rank's slices only

    # ------------------------------------------------------------------ plumbing

    def cu_hints(self):
        """the sl_set_available_cus settings this engine launches under (workspaces are sized for all of them)"""
        return [0, 256 - self.comm_cus] if self.comm_cus else [0]

    def set_comm_cus(self, comm_cus):
        """see self.comm_cus; re-sizes the split workspaces of the existing buffer sets for the new choosers' decisions"""
        comm_cus = int(comm_cus or 0)
        if comm_cus == self.comm_cus:
            return
        if comm_cus and not 0 < comm_cus <= 128:
            raise ValueError("comm_cus must be in 0 .. 128")
        self.comm_cus = comm_cus
        for buf in self._buffers.values():
            buf.launch_lists = {}
            buf.multi_tables = {}  # (their workspaces are sized for the CU settings in use: rebuilt by the next backward)
            buf.wgrad_multi_ws = None
            buf.size_nt_workspace(self, buf.fwd_geom, "fwd")
            if buf.bwd_ready:
                buf.size_backward_workspaces(self)

    @property
    def weights_version(self):
        """changes whenever the fp32 masters changed (set_weights, every optimizer step): what an engine sharing them compares"""
        return (self._weights_set_count, self.adam_iterations)

    def _set_cu_hint(self, cus):
        """sl_set_available_cus for the launches this THREAD enqueues from here on (the library keeps the setting per thread)
        and this engine's record of it -- one op of a recorded backward, so that a replayed step tracks it too and the sizing
        helpers that run in between (_part_geom, _wgrad_multi_workspace_need) restore the right value"""
        self.lib.call("sl_set_available_cus", cus)
        self._cu_hint_active = cus

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
                self._rec.append((0, self.lib.raw(name), args, name, tag, torch.cuda.current_stream(self.device)))
            return
        start, stop = TimingEvent(), TimingEvent()  # (timestamps only: no system-scope release between the launches)
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
        if self.timeline is not None:
            return self._replay_timeline(ops, callback)
        if self.kernel_timeline is not None:
            return self._replay_profiled(ops, callback)
        for op in ops:
            kind = op[0]
            if kind == 0:
                rc = op[1](*op[2])
                if rc != 0:
                    raise _lib.HipLibraryError("{} failed with status {}: {}".format(op[3], rc, self.lib.last_error()))
            elif kind == 1:
                op[1].record(op[2])
                op[3].wait_event(op[1])
            elif kind == 2:
                callback(op[1])
            else:  # a step of the sequence that has to be marshalled afresh every time (pointers / sizes that change per batch)
                op[1](*op[2])

    def _replay_timeline(self, ops, callback):
        """_replay with HIP events on the launch stream around every C-ABI call (self.timeline, bench.py's per-launch table):
        the step still runs from its recorded lists -- marshalled eagerly, the Python between the launches starves the GPU and
        the MFMA-bound kernels behind the gaps run at lower clocks (round 5: their durations came out 17 % above the kernel's
        own timestamps)"""
        out = self.timeline
        for op in ops:
            kind = op[0]
            if kind == 0:
                start, stop = TimingEvent(), TimingEvent()
                start.record(op[5])  # (the stream the launch was recorded on: bias passes run on the side stream)
                rc = op[1](*op[2])
                stop.record(op[5])
                out.append((op[4], start, stop))
                if rc != 0:
                    raise _lib.HipLibraryError("{} failed with status {}: {}".format(op[3], rc, self.lib.last_error()))
            elif kind == 1:
                op[1].record(op[2])
                op[3].wait_event(op[1])
            elif kind == 2:
                callback(op[1])
            else:
                op[1](*op[2])

    def _replay_profiled(self, ops, callback):
        """_replay with events attached to the main kernels of the launches named in self.kernel_timeline (bench.py's roofline
        leg): the step runs from its recorded lists exactly as in the timed region -- launched eagerly, the Python between the
        launches leaves gaps on the GPU and the kernels behind them run at other clocks (live 0.237 against 0.212 ms under
        rocprofv3 in the same process, round 5)"""
        tags, out = self.kernel_timeline
        for op in ops:
            kind = op[0]
            if kind == 0:
                if op[4] in tags:
                    start = torch.cuda.Event(enable_timing=True)
                    stop = torch.cuda.Event(enable_timing=True)
                    start.record()  # creates the HIP events; the library attaches them to the kernel's dispatch
                    stop.record()
                    self.lib.call("sl_profile_next_kernel", start.cuda_event, stop.cuda_event)
                    out.append((op[4], start, stop))
                rc = op[1](*op[2])
                if rc != 0:
                    raise _lib.HipLibraryError("{} failed with status {}: {}".format(op[3], rc, self.lib.last_error()))
            elif kind == 1:
                op[1].record(op[2])
                op[3].wait_event(op[1])
            elif kind == 2:
                callback(op[1])
            else:
                op[1](*op[2])

    def _eager_op(self, fn, *args):
        """fn(*args) now; while a launch list is being recorded it becomes ONE op of the list that calls fn again at
        replay (the launches inside are not recorded: their arguments change from batch to batch)"""
        rec, self._rec = self._rec, None
        try:
            fn(*args)
        finally:
            self._rec = rec
        if rec is not None:
            rec.append((3, fn, args))

    def _launch_list(self, buf, key):
        """The recorded launch list of `key` for this buffer set, or None (then the caller runs eagerly; with
        self._rec set by start_recording() that run records the list)."""
        if not self.use_launch_lists or self._rec is not None:
            return None
        return buf.launch_lists.get(key)  # (with self.timeline / self.kernel_timeline set: _replay_timeline / _replay_profiled)

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

    def bucket_plan(self):
        """Gradient buckets of the data-parallel exchange in the order backward() completes them: a list of
        (layer indices, (lo, hi)) with [lo, hi) the bucket's contiguous range of the flat gradient buffer.
        {output_conv, big_conv_2} (17 % of the bytes, complete after the second weight gradient of the step), {big_conv_1}
        (64 %, its exchange runs under big_conv_1's input gradient and the whole inner run), the run of identical inner layers
        (one grouped weight-gradient launch at the very end of backward), then what is left (striding_conv, whose weight
        gradient is the last kernel of backward: the only exchange nothing covers) -- one bucket with the run when
        sl_conv1d_wgrad_multi writes both (_wgrad_multi_layers).  Frozen layers are in no bucket."""
        n = len(self.plans)
        first = self.frozen_layer_count
        groups = []
        if n >= 3:
            groups.append(list(range(max(n - 2, first), n)))
            groups.append(list(range(max(n - 3, first), n - 2)))
        else:
            groups.append(list(range(first, n)))
        rest_hi = max(n - 3, first) if n >= 3 else first
        for (s0, e0) in reversed(self.runs):  # runs inside the remaining layers become buckets of their own
            lo = max(s0, first)
            if e0 < rest_hi and e0 >= lo:
                if e0 + 1 < rest_hi:
                    groups.append(list(range(e0 + 1, rest_hi)))
                groups.append(list(range(lo, e0 + 1)))
                rest_hi = lo
        if rest_hi > first:
            groups.append(list(range(first, rest_hi)))
        # one launch writes the weight gradients of the striding layer and of the run above it: one bucket
        multi = self._wgrad_multi_layers(first)
        if multi and multi[0] == 0:
            # every bucket that holds a layer of the launch becomes part of ONE bucket closed at the launch (its lowest
            # layer) -- whole buckets, so that with several runs a layer between them is neither left out of the merged
            # range nor reduced twice; the merged layers are then contiguous from layer 0 up
            span = set(range(multi[0], multi[-1] + 1))
            merged = sorted(set(l for g in groups if set(g) & span for l in g))
            groups = [g for g in groups if not set(g) & span]
            assert merged == list(range(merged[0], merged[-1] + 1)), merged
            parts = self._wgrad_multi_groups(first)
            if len(parts) == 2 and sorted(parts[0] + parts[1]) == merged:
                groups += parts  # split_last_bucket: the upper part of the run closes first, as a bucket of its own
            else:
                groups.append(merged)
        plan = []
        for layers in groups:
            if layers:
                hi_layer = self.plans[layers[-1]]
                plan.append((layers, (self.plans[layers[0]].w_off, hi_layer.b_off + hi_layer.cout_pad)))
        if self.front_plan is not None and not self.front_frozen and first == 0:
            fp = self.front_plan  # (raw-wave front layer: its gradients are the last launches of backward)
            plan.append(([fp.index], (fp.w_off, fp.b_off + fp.cout_pad)))
        return plan

    def bucket_ranges(self):
        """Flat-gradient ranges of bucket_plan(), in completion order (what GradBucketReducer takes)."""
        return [r for _, r in self.bucket_plan()]

    # ------------------------------------------------------------------ weights

    def set_weights(self, weights):
        """weights: [(W (k,Cin,Cout), b (Cout,))] numpy, Keras layout."""
        assert len(weights) == len(self.all_plans)
        self.params.zero_()
        for p, (w, b) in zip(self._public_plans(), weights):
            s = p.spec
            w = self._front_to_gemm(p, w)
            if tuple(w.shape) != (s.kernel_size, s.cin, s.cout) or tuple(b.shape) != (s.cout,):
                raise ValueError("weights of layer {} have shape {} / {}".format(s.name, w.shape, b.shape))
            wv, bv = self.layer_param_views(self.params, p)
            wv[:, :s.cin, :s.cout] = torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32)).to(self.device)
            bv[:s.cout] = torch.as_tensor(np.ascontiguousarray(b, dtype=np.float32)).to(self.device)
            if self._has_ones_output(p):
                bv[p.cout_pad - 1] = 1.0  # relu(0 * x + 1) = elu(1) = 1: the ones channel (see self.ones_channel)
        self._packed_dirty = True
        self._weights_set_count += 1

    def _has_ones_output(self, plan):
        """hidden layer whose output has channel padding: its last padded channel is the constant 1"""
        hidden = plan is self.front_plan or plan.index < len(self.plans) - 1
        return self.ones_channel and hidden and plan.cout_pad > plan.spec.cout

    def _has_ones_input(self):
        """the packed INPUT carries a ones channel (sl_pack_input_ones) where its bins leave a padding channel free: 257 bins in
        rows of 320 (configuration 5) -- not 128 mel bins, which fill their rows.  Single-plane paths only."""
        if self.front_plan is not None:  # the front layer's last padded filter is the constant 1 (its bias; _has_ones_output)
            return self._has_ones_output(self.front_plan)
        return self.ones_channel and self.planes == 1 and self.plans[0].cin_pad > self.specs[0].cin

    def _ones_input_layers(self, first):
        """trainable layers whose input carries a ones channel: their bias gradient is row cin_pad - 1 of dW"""
        lower = [0] if (first == 0 and self._has_ones_input()) else []
        return lower + [i for i in range(max(first, 1), len(self.plans)) if self._has_ones_output(self.plans[i - 1])]

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
        for p in self._public_plans():
            s = p.spec
            wv, bv = self.layer_param_views(tensor, p)
            out.append((self._front_from_gemm(p, wv[:, :s.cin, :s.cout].contiguous().cpu().numpy()),
                        bv[:s.cout].contiguous().cpu().numpy()))
        return out

    def get_weights(self):
        return self._unpad(self.params)

    def get_gradients(self):
        """The gradients of the last backward pass in the Keras layout.  (After a data-parallel step with a sharded
        optimizer only this rank's slice of every bucket holds the reduced gradient: reduce-scatter, not all-reduce.)"""
        return self._unpad(self.grads)

    def get_optimizer_state(self):
        """Adam moments in the Keras layout (per layer (m_W, m_b), (v_W, v_b)), the step count and the dropout step
        counter: everything beyond the weights that the next step depends on.
        After a step with a sharded optimizer (GradBucketReducer(shard_optimizer=True)) every rank holds the moments of its
        own slice of each bucket only: they are all-gathered here first, bucket by bucket with the slices of gather_bucket --
        a COLLECTIVE call in that case (every rank must make it; Wav2Letter.train does, each rank saves at the same epoch)."""
        reducer = self._sharded_reducer
        if reducer is not None and (reducer.world_size > 1 or reducer.force):
            for flat in (self.adam_m, self.adam_v):
                for b in range(len(reducer.ranges)):
                    reducer.gather_bucket(b, flat)
            reducer.wait_all()
        return {"m": self._unpad(self.adam_m), "v": self._unpad(self.adam_v), "iterations": int(self.adam_iterations),
                "dropout_steps": int(self._dropout_steps)}

    def set_optimizer_state(self, state):
        for name, flat in (("m", self.adam_m), ("v", self.adam_v)):
            flat.zero_()
            for p, (w, b) in zip(self._public_plans(), state[name]):
                s = p.spec
                w = self._front_to_gemm(p, w)
                if tuple(w.shape) != (s.kernel_size, s.cin, s.cout) or tuple(b.shape) != (s.cout,):
                    raise ValueError("optimizer state of layer {} has shape {} / {}".format(s.name, w.shape, b.shape))
                wv, bv = self.layer_param_views(flat, p)
                wv[:, :s.cin, :s.cout] = torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32)).to(self.device)
                bv[:s.cout] = torch.as_tensor(np.ascontiguousarray(b, dtype=np.float32)).to(self.device)
        self.adam_iterations = int(state["iterations"])
        self._dropout_steps = int(state.get("dropout_steps", 0))

    def repack_weights(self):
        if self.planes > 1:
            return self._repack_weights_x3()
        st = self._stream()
        for p in self.all_plans:
            wv, _ = self.layer_param_views(self.params, p)
            wd = self.w_dgrad[p.index]
            k, cin = self._pack_dims(p)
            self._launch("pack:" + p.spec.name, "sl_pack_weights", wv.data_ptr(), self.w_fwd[p.index].data_ptr(),
                          wd.data_ptr() if wd is not None else None, k, cin, p.cout_pad, self.dtype_code, st)
        self._packed_dirty = False

    def _pack_dims(self, p):
        """(taps, input channels) the operand copies of plan p are packed with: the PAIR VIEW for a strided first layer that
        has a dgrad operand (same bytes for the forward operand; the flipped taps of the input-gradient operand are the
        pair view's 24, not the layer's 48)"""
        if p.index == 0 and p.spec.stride == 2 and self.w_dgrad[0] is not None and self.planes == 1:
            return p.taps_view, p.cin_view
        # (bf16x3: a frame's three planes sit side by side in a row, so the pair view of the FORWARD operand is the layer's
        # own (48, cin_pad) packing; the pair view of the input-gradient operand is packed apart, _pack_pair_dgrad_x3)
        return p.spec.kernel_size, p.cin_pad

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
        if self.front_plan is not None:
            return self._load_front_input(src)
        if f != self.specs[0].cin:
            raise ValueError("input has {} bins per frame, the net expects {}".format(f, self.specs[0].cin))
        buf = self.buffers(batch, t_in)
        p0 = self.plans[0]
        if self.planes > 1:
            self._launch("pack_input", self._x3("sl_split3_pack_input"), src.data_ptr(), buf.x0.data_ptr(), batch, t_in, f, p0.cin_pad,
                         p0.pad_left, buf.rows0 * p0.cin_pad * self.planes, self._stream())
            self.cur = buf
            self._src_keepalive = src
            return buf
        self._launch("pack_input", "sl_pack_input_ones", src.data_ptr(), buf.x0.data_ptr(), batch, t_in, f, p0.pad_left,
                     p0.cin_pad, buf.rows0 * p0.cin_pad, p0.cin_pad - 1 if self._has_ones_input() else -1, self.dtype_code,
                     self._stream())
        self.cur = buf
        self._src_keepalive = src
        return buf

    def _chain_table(self, kind, layers, buf):
        """ctypes pointer tables of sl_conv1d_chain for the given layers (plan indices in launch order), cached on the
        buffer set (freed with it).  kind 'fwd': ys = activations, ws = forward operands, biases; kind 'dgrad': layer i of the list
        is the input gradient of plan i: ys = g[i - 1], ws = dgrad operands, masks = y[i - 1]."""
        key = (kind, tuple(layers))
        hit = buf.chain_tables.get(key)
        if hit is not None:
            return hit
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
        buf.chain_tables[key] = (ys, ws, aux)
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

    def forward(self, input_batch=None, training=False, split_ctc=None):
        """Runs the 11 conv layers + softmax.  Returns the probability tensor (B,T',K) fp32 in HBM.
        training=True applies dropout (if self.dropout_rate) to the inputs of the first n-3 layers: the packed input
        goes through sl_dropout into a second buffer, every other activation is dropped in place right after the
        layer that produced it (so the stored activation is the post-dropout one the backward pass needs).
        split_ctc=(grad_scale, a) (train_step_resident, a = split_top_plan()): the top three layers run in two parts of the
        batch -- utterances [0, a) and [a, B) -- and each part's CTC loss + gradient (ctc(grad_scale)) is launched on a side
        stream as soon as its probabilities exist; backward() picks the parts up (see self.split_top)."""
        buf = self.load_input(input_batch) if input_batch is not None else self.cur
        if self._packed_dirty:
            self.repack_weights()
        st = self._stream()
        n = len(self.plans)
        rate = self.dropout_rate if training else None
        buf.dropped = bool(rate)
        buf.split_pending = 0
        if self.front_plan is not None and not rate and buf.front_frames_dropped:
            self._front_gather(buf, buf.front_src)  # (the last step gathered DROPPED samples; not part of any launch list)
        fuse_out = self.fuse_output_softmax and self.dtype == "bf16" and bool(self.lib.raw("sl_output_softmax_supported")(
            ctypes.byref(buf.fwd_geom[n - 1]), self.grapheme_set_size, self.dtype_code))
        # launch list (no dropout): everything below takes its frame count from the geometries, except the unfused
        # softmax, which gets it by value -> then the list is per length
        key = None if rate else ("fwd", st, fuse_out, self.use_chain, tuple(sorted(self.nt_cfg.items())),
                                 None if fuse_out else buf.t_out, split_ctc)
        if split_ctc is not None:
            # the part geometries (and the workspace they may re-allocate) exist BEFORE a list is looked up or recorded: a
            # launch of the lower layers already in the list being recorded would keep a freed nt_ws pointer
            self._presize_parts(buf, ("fwd", "dgrad") if buf.bwd_ready else ("fwd",), split_ctc[1])
        ops = self._launch_list(buf, key) if key is not None else None
        if ops is not None:
            self._replay(ops)
            buf.split_pending = split_ctc[1] if split_ctc is not None else 0
            return buf.probs
        record = key is not None and self.use_launch_lists and self.timeline is None and \
            self.kernel_timeline is None and self._rec is None
        if not record:
            return self._forward_eager(buf, rate, fuse_out, st, split_ctc)
        self._rec = []
        try:
            probs = self._forward_eager(buf, rate, fuse_out, st, split_ctc)
            buf.launch_lists[key] = self._rec
            return probs
        finally:
            self._rec = None

    def _forward_eager(self, buf, rate, fuse_out, st, split_ctc=None):
        if self.planes > 1:
            return self._forward_x3(buf, st, rate)
        n = len(self.plans)
        x = buf.x0
        seed0 = 0
        if rate:
            self._dropout_steps += 1
            seed0 = (self.dropout_seed * 1000003 + self._dropout_steps) * 64
            buf.dropout_seed0 = seed0  # ELU layers: backward recomputes the keep decisions (sl_elu_dropout_backward)
        if self.front_plan is not None:
            self._front_forward(buf, rate, seed0, st)
        if rate:
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
            if split_ctc is not None and p.index == n - 3:
                return self._forward_top_split(buf, x, st, split_ctc[0], split_ctc[1])
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
        if self.planes > 1:  # fp32 dL/dlogits into the staging buffer, then into the planes of g[last]
            cp = self.plans[last].cout_pad
            buf.stage32[:buf.batch * buf.tt_pad * cp].zero_()  # (padded classes: the CTC kernel writes the k real ones)
            self._launch("ctc", "sl_ctc_loss_grad", buf.probs.data_ptr(), buf.logq.data_ptr(), buf.labels.data_ptr(),
                         buf.label_len.data_ptr(), buf.input_len.data_ptr(), buf.loss.data_ptr(), buf.stage32.data_ptr(),
                         buf.batch, buf.t_out, self.grapheme_set_size, l_max, 0, cp, buf.tt_pad * cp, _lib.SL_F32,
                         self.ctc_epsilon, grad_scale * self.g_scale, buf.ctc_ws.data_ptr(), buf.ctc_ws.numel(), self._stream())
            self._launch("split:ctc", self._x3("sl_split3"), buf.stage32.data_ptr(), buf.g[last].data_ptr(), None, buf.batch, buf.t_out,
                         cp, buf.tt_pad * cp, HALO, buf.rows * cp * self.planes, 0, self._stream())
            return buf.loss
        self._launch("ctc", "sl_ctc_loss_grad", buf.probs.data_ptr(), buf.logq.data_ptr(), buf.labels.data_ptr(),
                      buf.label_len.data_ptr(), buf.input_len.data_ptr(), buf.loss.data_ptr(), buf.g[last].data_ptr(),
                      buf.batch, buf.t_out, self.grapheme_set_size, l_max, HALO, self.plans[last].cout_pad,
                      buf.rows * self.plans[last].cout_pad, self.dtype_code, self.ctc_epsilon, grad_scale,
                      buf.ctc_ws.data_ptr(), buf.ctc_ws.numel(), self._stream())
        return buf.loss

    def backward(self, on_bucket_ready=None):
        """wgrad / bias-grad / dgrad for every trainable layer, output layer first.

        Bias gradients: layers whose input carries the ones channel get theirs out of the weight-gradient GEMM
        (_bias_grads_from_wgrad); the others (striding_conv; every layer when dropout touched the ones) by an HBM-bound
        sl_bias_grad pass over g[i] on a SIDE stream underneath the MFMA-bound wgrad / dgrad kernels.  Every hand-over to
        the side stream is an event record on the main stream and drains it for ~6 us (profiles/r01j: the only gaps of the
        step's timeline), and g[i] stays intact until the next step, so passes are collected until >= 128 MB of gradient
        are owed (or a bucket / the end of backward needs them).
        on_bucket_ready(b) is called once every launch that writes gradient bucket b (bucket_plan) is enqueued on the main
        stream -- the data-parallel reducer starts that bucket's exchange there."""
        buf = self.cur
        if self.forward_only:
            raise RuntimeError("this engine was built forward_only (evaluation): it has no gradient buffers")
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)  # (ROCm offers no priority below the default)
        side = self._side_stream
        # launch list: not with dropout (its scale passes take the rate by value)
        key = None if buf.dropped else ("bwd", main.cuda_stream, buf.split_pending, on_bucket_ready is not None, self.ones_channel,
                                        self.frozen_layer_count, self.group_wgrad, self.use_chain, self.fuse_output_backward,
                                        self.use_wgrad_multi, self.multi_overlap_tiles, self.small_bias_pass_on_main,
                                        self.split_last_bucket, self.split_last_at,
                                        tuple(sorted(self.nt_cfg.items())),
                                        buf.t_out if self.planes > 1 else None)  # (bf16x3 helpers take it by value)
        try:
            ops = self._launch_list(buf, key) if key is not None else None
            if ops is not None:
                self._replay(ops, on_bucket_ready)
                return
            record = key is not None and self.use_launch_lists and self.timeline is None and \
                self.kernel_timeline is None and self._rec is None
            if not record:
                self._backward_eager(buf, main, side, on_bucket_ready)
                return
            self._rec = []
            try:
                self._backward_eager(buf, main, side, on_bucket_ready)
                buf.launch_lists[key] = self._rec
            finally:
                self._rec = None
        except BaseException:
            if on_bucket_ready is not None and self.comm_cus:
                # raised between setting and clearing the CU hint: the library's setting is process-wide state shared by
                # every engine of the process -- do not leave it behind
                self._cu_hint_active = 0
                self.lib.raw("sl_set_available_cus")(0)
            raise

    def _grouped_wgrad_runs(self, first):
        """layer index -> (lo, hi) of the run whose weight gradients are one grouped launch (bf16 path)"""
        grouped = {}
        if self.group_wgrad and self.dtype == "bf16":
            for (s0, e0) in self.runs:
                lo = max(s0, first)
                if e0 > lo:
                    for q in range(lo, e0 + 1):
                        grouped[q] = (lo, e0)
        return grouped

    def _dgrad_chains(self, buf, first):
        """input gradients of a run of identical ReLU layers in one launch (sl_conv1d_chain): {top layer: layers, top
        first}, and the set of layers such a launch covers besides its top layer"""
        dchain, skip = {}, set()
        if buf.dropped or not self.use_chain or self.dtype != "bf16":
            return dchain, skip
        cut = self._split_last_cut(first)
        for (s0, e0) in self.runs:
            layers = list(range(e0, max(s0, first + 1) - 1, -1))
            if len(layers) >= 2 and all(self.specs[i - 1].activation == "relu" for i in layers) and \
                    not any(("dgrad", self.specs[i].name) in self.nt_cfg for i in layers) and \
                    bool(self.lib.raw("sl_conv1d_chain_supported")(ctypes.byref(buf.dgrad_geom[e0]), len(layers),
                                                                   self.dtype_code)):
                pieces = [layers]
                if cut is not None and s0 < cut <= e0:
                    # split_last_bucket: input gradients of layers e0 .. cut first (they complete g[cut .. e0 - 1], all the
                    # upper weight-gradient launch still needs), the rest behind that launch
                    upper, lower = [i for i in layers if i >= cut], [i for i in layers if i < cut]
                    ok = all(len(q) == 1 or bool(self.lib.raw("sl_conv1d_chain_supported")(
                        ctypes.byref(buf.dgrad_geom[q[0]]), len(q), self.dtype_code)) for q in (upper, lower) if q)
                    if ok:
                        pieces = [q for q in (upper, lower) if q]
                for q in pieces:
                    if len(q) >= 2:
                        dchain[q[0]] = q
                        skip.update(q[1:])
        return dchain, skip

    def _split_last_cut(self, first):
        """split_last_bucket: the layer at which the balanced weight-gradient launch (and the fused input-gradient launch of
        the run it covers) is cut in two, or None"""
        if not self.split_last_bucket or self.planes > 1:
            return None
        multi = self._wgrad_multi_layers(first)
        cut = self.split_last_at
        if not multi or multi[0] != 0 or not (multi[0] < cut <= multi[-1]) or cut <= first:
            return None
        return cut

    def _wgrad_multi_groups(self, first, grouped=None):
        """the balanced weight-gradient launches of backward, in the order they are enqueued: one for all of
        _wgrad_multi_layers, or -- split_last_bucket -- the layers from the cut up first, then the rest"""
        multi = self._wgrad_multi_layers(first, grouped)
        if not multi:
            return []
        cut = self._split_last_cut(first)
        if cut is None:
            return [multi]
        return [[i for i in multi if i >= cut], [i for i in multi if i < cut]]

    def _launch_wgrad(self, buf, i, grouped, st):
        """weight gradient of layer i -- or, at the lowest layer of a grouped run, of the whole run"""
        p = self.plans[i]
        if i in grouped:
            lo, hi = grouped[i]
            if i != lo:
                return  # every g[lo..hi] is complete only at the lowest layer: one launch for the run there
            plo = self.plans[lo]
            dw_lo, _ = self.layer_param_views(self.grads, plo)
            stride_elems = buf.batch * buf.rows * plo.cin_pad
            self._launch("wgrad:{}..{}".format(plo.spec.name, self.plans[hi].spec.name), "sl_conv1d_wgrad_grouped",
                         buf.y[lo - 1].data_ptr(), buf.g[lo].data_ptr(), dw_lo.data_ptr(),
                         ctypes.byref(buf.wgrad_geom[lo]), hi - lo + 1, stride_elems, stride_elems,
                         plo.w_numel + plo.cout_pad, 0, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
            return
        x = (buf.x0_dropped if buf.dropped else buf.x0) if i == 0 else buf.y[i - 1]
        dw, _ = self.layer_param_views(self.grads, p)
        self._launch("wgrad:" + p.spec.name, "sl_conv1d_wgrad", x.data_ptr(), buf.g[i].data_ptr(), dw.data_ptr(),
                     ctypes.byref(buf.wgrad_geom[i]), self.dtype_code, self.nt_cfg.get(("wgrad", p.spec.name), 0),
                     buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)

    def _fused_output_backward(self, buf, i, first, grouped, dchain, dchain_skip):
        """layer i's weight AND input gradient in one sl_conv1d_backward_1x1 launch: the 1x1 output layer on the bf16 path"""
        if not self.fuse_output_backward or i != len(self.plans) - 1 or i <= first or self.dtype != "bf16":
            return False
        if i in grouped or i in dchain or i in dchain_skip or buf.bwd1x1_ws is None:
            return False
        name = self.specs[i].name
        if ("wgrad", name) in self.nt_cfg or ("dgrad", name) in self.nt_cfg:
            return False
        if self.specs[i - 1].activation not in ("relu", "elu") or (buf.dropped and i in self._dropout_layers()):
            return False
        return bool(self.lib.raw("sl_conv1d_backward_1x1_supported")(ctypes.byref(buf.wgrad_geom[i]),
                                                                     self.grapheme_set_size, self.dtype_code))

    def _launch_output_backward(self, buf, i, st):
        p = self.plans[i]
        dw, _ = self.layer_param_views(self.grads, p)
        epi = _lib.EPI_ELU_MASK if self.specs[i - 1].activation == "elu" else _lib.EPI_RELU_MASK
        self._launch("bwd:" + p.spec.name, "sl_conv1d_backward_1x1", buf.y[i - 1].data_ptr(), buf.g[i].data_ptr(),
                     self.w_dgrad[i].data_ptr(), buf.g[i - 1].data_ptr(), dw.data_ptr(), ctypes.byref(buf.wgrad_geom[i]),
                     epi, self.grapheme_set_size, self.dtype_code, 0, buf.bwd1x1_ws.data_ptr(), buf.bwd1x1_ws.numel(), st)

    def _launch_dgrad(self, buf, i, st):
        """input gradient of layer i into g[i - 1], through the activation of layer i - 1 (and its dropout)"""
        p = self.plans[i]
        elu = self.specs[i - 1].activation == "elu"
        elu_dropped = elu and buf.dropped and i in self._dropout_layers()
        self._launch("dgrad:" + p.spec.name, "sl_conv1d_nt", buf.g[i].data_ptr(), self.w_dgrad[i].data_ptr(),
                     None, None if elu_dropped else buf.y[i - 1].data_ptr(), buf.g[i - 1].data_ptr(),
                     ctypes.byref(buf.dgrad_geom[i]),
                     _lib.EPI_NONE if elu_dropped else (_lib.EPI_ELU_MASK if elu else _lib.EPI_RELU_MASK),
                     self.dtype_code, 0, self.nt_cfg.get(("dgrad", p.spec.name), 0),
                     buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
        if elu_dropped:
            # a stored zero is ambiguous behind an ELU (dropped, or elu(z) == 0): both factors of the chain rule
            # in one elementwise pass that recomputes the keep decisions from the step's seed
            self._launch("dropout_elu_bwd:" + p.spec.name, "sl_elu_dropout_backward", buf.g[i - 1].data_ptr(),
                         buf.y[i - 1].data_ptr(), buf.g[i - 1].numel(), self.dtype_code, self.dropout_rate,
                         buf.dropout_seed0 + i, st)
        elif buf.dropped and i in self._dropout_layers():
            # the dgrad epilogue's mask (stored activation > 0) already applied the keep mask: the stored
            # activation is post-dropout; what is left of d dropout / dx is the 1 / (1 - rate) factor
            self._launch("dropout_scale:" + p.spec.name, "sl_scale", buf.g[i - 1].data_ptr(), buf.g[i - 1].numel(),
                         self.dtype_code, 1.0 / (1.0 - self.dropout_rate), st)

    def _wgrad_multi_layers(self, first, grouped=None):
        """layers whose weight gradients go into ONE sl_conv1d_wgrad_multi launch (at the lowest of them): the runs of
        identical layers and the striding layer below them.  The same with and without a data-parallel exchange (the step
        is then bit-identical either way); bucket_plan() closes the striding layer's bucket together with the run's."""
        if self.planes > 1:
            return self._wgrad_multi_layers_x3(first)
        if grouped is None:
            grouped = self._grouped_wgrad_runs(first)
        if not self.use_wgrad_multi or self.dtype != "bf16" or not grouped:
            return []

        def fits(i):  # (input channels: whole 256-wide tiles, or -- round 4 -- a last tile overlapping its neighbour)
            cin = self.plans[i].cin_view
            return cin >= 256 and cin % (64 if self.multi_overlap_tiles else 256) == 0 and \
                self.plans[i].cout_pad % 256 == 0 and ("wgrad", self.specs[i].name) not in self.nt_cfg
        layers = sorted(grouped)
        if not all(fits(i) for i in layers) or len(layers) > 15:
            return []
        if first == 0 and self.plans[0].spec.stride == 2 and layers[0] == 1 and fits(0):
            layers = [0] + layers
        return layers

    def _wgrad_multi_workspace_need(self, table, n_jobs):
        """workspace of a balanced weight-gradient launch under EVERY sl_set_available_cus setting this engine launches
        under: its segment count is 2 * CUs / tiles, so a table first used under the data-parallel CU hint would otherwise
        get a workspace too small for a later step without the hint"""
        need = 0
        for hint in self.cu_hints():
            self.lib.call("sl_set_available_cus", hint)
            need = max(need, self.lib.raw("sl_conv1d_wgrad_multi_workspace_bytes")(table, n_jobs, self.dtype_code))
        self.lib.call("sl_set_available_cus", self._cu_hint_active)
        return need

    def _wgrad_multi_table(self, buf, layers):
        """job table of a balanced weight-gradient launch over `layers`, built once per buffer set; makes sure the shared
        workspace covers it.  backward() builds the tables of ALL its balanced launches before the first of them is enqueued:
        a workspace re-allocated for the second launch would leave the first one's recorded launch with a freed pointer."""
        key = (tuple(layers), buf.dropped)
        table = buf.multi_tables.get(key)
        if table is None:
            table = (_lib.WgradJob * len(layers))()
            for job, i in zip(table, layers):
                x = (buf.x0_dropped if buf.dropped else buf.x0) if i == 0 else buf.y[i - 1]
                dw, _ = self.layer_param_views(self.grads, self.plans[i])
                job.x, job.g, job.dw = x.data_ptr(), buf.g[i].data_ptr(), dw.data_ptr()
                for name, _ in ConvGeom._fields_:
                    setattr(job.geom, name, getattr(buf.wgrad_geom[i], name))
            buf.multi_tables[key] = table
            need = self._wgrad_multi_workspace_need(table, len(layers))
            if buf.wgrad_multi_ws is None or buf.wgrad_multi_ws.numel() < need:
                if buf.wgrad_multi_ws is not None:
                    buf.launch_lists = {}  # (a recorded list may hold the pointer that is freed here)
                buf.wgrad_multi_ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=self.device)
        return table

    def _launch_wgrad_multi(self, buf, layers, st):
        table = self._wgrad_multi_table(buf, layers)
        self._launch("wgrad:{}..{}".format(self.specs[layers[0]].name, self.specs[layers[-1]].name),
                     "sl_conv1d_wgrad_multi", table, len(layers), self.dtype_code, buf.wgrad_multi_ws.data_ptr(),
                     buf.wgrad_multi_ws.numel(), st)

    def _backward_eager(self, buf, main, side, on_bucket_ready):
        if self.planes > 1:
            return self._backward_x3(buf, main.cuda_stream, on_bucket_ready)
        first = self.frozen_layer_count
        grouped = self._grouped_wgrad_runs(first)
        dchain, dchain_skip = self._dgrad_chains(buf, first)
        multi = {}  # layer -> the layers of the balanced launch that writes its weight gradient (at the lowest of them)
        for launch_layers in self._wgrad_multi_groups(first, grouped):
            self._wgrad_multi_table(buf, launch_layers)
            for i in launch_layers:
                multi[i] = launch_layers
        # bias gradients out of the weight-gradient GEMM (self.ones_channel): which layers, and whether the row holds the
        # bias gradient (the ones were not touched by dropout) or only has to be zeroed before the optimizer sees it
        ones_in = self._ones_input_layers(first)
        ones_db = set() if buf.dropped else set(ones_in)
        # data parallel: bucket b is complete once the weight gradient of its LOWEST layer is enqueued
        bucket_at = {}
        if on_bucket_ready is not None:
            for b, (layers, _) in enumerate(self.bucket_plan()):
                bucket_at[layers[0]] = (b, layers)
        side_busy = [False]

        def flush_bias_passes(pending):
            """sl_bias_grad passes of the layers in `pending` (their g is complete at this point of MAIN) on SIDE -- a lone
            small pass (the first layer's 8 MB, the only one left beside the ones channel) on MAIN: two hand-overs cost the
            main stream 6.5 us each (rocprof timeline), the pass itself 9"""
            if len(pending) == 1 and buf.g[pending[0]].numel() * buf.g[pending[0]].element_size() <= (32 << 20) \
                    and self.small_bias_pass_on_main:
                j = pending[0]
                _, db_j = self.layer_param_views(self.grads, self.plans[j])
                self._launch("bgrad:" + self.plans[j].spec.name, "sl_bias_grad", buf.g[j].data_ptr(), db_j.data_ptr(),
                             ctypes.byref(buf.wgrad_geom[j]), self.dtype_code, buf.bias_ws.data_ptr(),
                             buf.bias_ws.numel(), main.cuda_stream)
                del pending[:]
                return
            self._hand_over(main, side)
            with torch.cuda.stream(side):
                for j in pending:
                    _, db_j = self.layer_param_views(self.grads, self.plans[j])
                    self._launch("bgrad:" + self.plans[j].spec.name, "sl_bias_grad", buf.g[j].data_ptr(),
                                 db_j.data_ptr(), ctypes.byref(buf.wgrad_geom[j]), self.dtype_code,
                                 buf.bias_ws.data_ptr(), buf.bias_ws.numel(), side.cuda_stream)
            side_busy[0] = True
            del pending[:]

        def join_side():
            if side_busy[0]:
                self._hand_over(side, main)
                side_busy[0] = False

        n = len(self.plans)
        hint = on_bucket_ready is not None and self.comm_cus
        split = buf.split_pending
        if split:  # the CTC ran in two parts: so do the input gradients of the top three layers (see self.split_top)
            self._backward_top_split(buf, main, split)
        pending, pending_bytes = [], 0  # layers whose bias-gradient pass is still owed to the side stream
        for p in reversed(self.plans[first:]):
            i = p.index
            if i not in ones_db:
                pending.append(i)
                pending_bytes += buf.g[i].numel() * buf.g[i].element_size()
            closes_bucket = i in bucket_at
            if pending and (pending_bytes >= (128 << 20) or i == first or closes_bucket):
                flush_bias_passes(pending)
                pending_bytes = 0
            fused_bwd = self._fused_output_backward(buf, i, first, grouped, dchain, dchain_skip)
            if split and i == n - 1:
                pass  # weight + input gradient of the output layer: done by halves above
            elif fused_bwd:
                self._launch_output_backward(buf, i, main.cuda_stream)
            elif i in multi:
                if i == multi[i][0]:  # every gradient tensor the launch reads is complete at its lowest layer
                    self._launch_wgrad_multi(buf, multi[i], main.cuda_stream)
            else:
                self._launch_wgrad(buf, i, grouped, main.cuda_stream)
            if closes_bucket:
                b, layers = bucket_at[i]
                rows = [j for j in layers if j in ones_in]
                if rows:
                    self._bias_grads_from_wgrad(rows, bool(ones_db), main)
                join_side()
                on_bucket_ready(b)
                if self._rec is not None:
                    self._rec.append((2, b))
                if hint and b == 0:  # from here on communication kernels may own CUs: the choosers plan for the rest
                    self._eager_op(self._set_cu_hint, 256 - self.comm_cus)
            if i in dchain:
                layers = dchain[i]
                ys, ws, masks = self._chain_table("dgrad", layers, buf)
                self._launch("dgrad:{}..{}".format(self.specs[layers[0]].name, self.specs[layers[-1]].name),
                             "sl_conv1d_chain", buf.g[i].data_ptr(), ys, ws, None, masks,
                             ctypes.byref(buf.dgrad_geom[i]), len(layers), _lib.EPI_RELU_MASK, self.dtype_code,
                             main.cuda_stream)
            elif i > first and i not in dchain_skip and not fused_bwd and not (split and i >= n - 3):
                self._launch_dgrad(buf, i, main.cuda_stream)
        if on_bucket_ready is None:
            if ones_in:
                self._bias_grads_from_wgrad(ones_in, bool(ones_db), main)
            join_side()
        if self.front_plan is not None and first == 0 and not self.front_frozen:
            self._front_backward(buf, main.cuda_stream)
            if on_bucket_ready is not None:  # the front layer's parameters: the last bucket of bucket_plan()
                b = len(self.bucket_plan()) - 1
                on_bucket_ready(b)
                if self._rec is not None:
                    self._rec.append((2, b))
        if hint:
            self._eager_op(self._set_cu_hint, 0)

    def adam_step(self, fused=True):
        """Keras-2.0 Adam on the flat fp32 masters.  fused=True: one kernel per trainable layer that applies Adam AND
        rewrites the layer's two bf16 operand copies in the same pass (no separate repack); fused=False: one flat
        elementwise launch, operands repacked lazily by the next forward()."""
        if self.forward_only:
            raise RuntimeError("this engine was built forward_only (evaluation): it has no optimizer state")
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
        self._adam_layers(self._trainable_layers(), st)

    def _trainable_layers(self):
        """internal indices of the layers the optimizer updates (the front layer, if any, has index len(plans))"""
        layers = list(range(self.frozen_layer_count, len(self.plans)))
        if self.front_plan is not None and not self.front_frozen:
            layers.append(self.front_plan.index)
        return layers

    def _adam_table(self, chunk):
        table = self._adam_tables.get(tuple(chunk))
        if table is None:  # (the operand copies never move: built once per set of layers)
            table = (_lib.AdamLayer * len(chunk))()
            for entry, i in zip(table, chunk):
                p = self.all_plans[i]
                wd = self.w_dgrad[p.index]
                if self.planes > 1 and p.index == 0 and p is not self.front_plan:
                    wd = None  # bf16x3 under a raw-wave front layer: _pack_pair_dgrad_x3 rewrites this operand
                entry.offset = p.w_off
                entry.w_fwd = self.w_fwd[p.index].data_ptr()
                entry.w_dgrad = wd.data_ptr() if wd is not None else None
                entry.k, entry.cin_pad = self._pack_dims(p)
                entry.cout_pad = p.cout_pad
            self._adam_tables[tuple(chunk)] = table
        return table

    def _adam_layers(self, layers, st):
        """Fused Adam + bf16 operand repack of the given layers on stream st (self.adam_iterations already counts
        this step): ONE launch for all of them (sl_adam_pack_layers), 16 layers per call at most."""
        layers = list(layers)
        for lo in range(0, len(layers), 16):
            chunk = layers[lo:lo + 16]
            table = self._adam_table(chunk)
            tag = "adam:{}..{}".format(self.all_plans[chunk[0]].spec.name, self.all_plans[chunk[-1]].spec.name)
            if self.planes == 3 and self.x3_f16:
                self._launch(tag, "sl_splitf16_adam_pack_layers", self.params.data_ptr(), self.grads.data_ptr(),
                             self.adam_m.data_ptr(), self.adam_v.data_ptr(), table, len(chunk), self.adam_iterations,
                             self.lr, self.beta_1, self.beta_2, self.adam_epsilon, self.w_scale, st)
            elif self.planes == 3:  # bf16x3: the [w_hi | w_hi | w_lo] operand rows are rewritten in the same pass
                self._launch(tag, "sl_split3_adam_pack_layers", self.params.data_ptr(), self.grads.data_ptr(),
                             self.adam_m.data_ptr(), self.adam_v.data_ptr(), table, len(chunk), self.adam_iterations,
                             self.lr, self.beta_1, self.beta_2, self.adam_epsilon, st)
                if 0 in chunk and self.w_dgrad[0] is not None:
                    self._pack_pair_dgrad_x3(st)
            else:
                self._launch(tag, "sl_adam_pack_layers", self.params.data_ptr(), self.grads.data_ptr(),
                             self.adam_m.data_ptr(), self.adam_v.data_ptr(), table, len(chunk), self.dtype_code,
                             self.adam_iterations, self.lr, self.beta_1, self.beta_2, self.adam_epsilon, st)

    def train_step(self, input_batch, label_batch, label_lengths, prediction_lengths, reducer=None):
        """One full optimisation step (forward, CTC, backward, [gradient all-reduce], Adam, weight repack).
        Returns the per-utterance loss tensor (B,) in HBM (not synchronised)."""
        self.load_input(input_batch)
        self.set_labels(label_batch, label_lengths, prediction_lengths)
        return self.train_step_resident(reducer)

    def train_step_resident(self, reducer=None):
        """Same, with input / labels / lengths already resident in HBM (bench.py's timed region)."""
        dp = reducer is not None and (reducer.world_size > 1 or reducer.force)
        world = reducer.world_size if reducer is not None else 1
        self._sharded_reducer = reducer if (dp and reducer.shard_optimizer) else None
        self.set_comm_cus(getattr(reducer, "comm_cus", 0) if dp else 0)
        grad_scale = 1.0 / (self.cur.batch * world)
        part = self.split_top_plan(self.cur)
        if part:
            self.forward(training=True, split_ctc=(grad_scale, part))  # ... and the CTC, in two parts (self.split_top)
            loss = self.cur.loss
        else:
            self.forward(training=True)
            loss = self.ctc(grad_scale=grad_scale)
        if not dp:
            self.backward()
            self.adam_step()
            return loss
        self.backward(on_bucket_ready=reducer.reduce_bucket)
        # every bucket's exchange was started the moment its last weight gradient was enqueued; the big ones finished
        # under the rest of backward.  Update bucket by bucket, in the order the exchanges complete.
        if self._packed_dirty:
            self.repack_weights()
        self.adam_iterations += 1
        st = self._stream()
        plan = self.bucket_plan()
        if not reducer.shard_optimizer:
            for layers, _ in plan:
                reducer.wait_next()
                self._adam_layers(layers, st)
            return loss
        # sharded optimizer: each bucket was reduce-SCATTERED; this rank holds the summed gradient of its 1/world slice
        # only, updates that slice (plain elementwise Adam, 1/world of the optimizer's HBM traffic), and the updated
        # fp32 masters are all-gathered in place; the operand copies are rewritten from the gathered masters
        for b, (layers, (lo, hi)) in enumerate(plan):
            reducer.wait_next()
            slo, shi = reducer.shard_of(lo, hi)
            self._launch("adam_shard:{}".format(b), "sl_adam_step", self.params[slo:shi].data_ptr(),
                         self.grads[slo:shi].data_ptr(), self.adam_m[slo:shi].data_ptr(), self.adam_v[slo:shi].data_ptr(),
                         shi - slo, self.adam_iterations, self.lr, self.beta_1, self.beta_2, self.adam_epsilon, st)
            reducer.gather_bucket(b, self.params)
        for layers, _ in plan:
            reducer.wait_next()
            self._pack_layers(layers, st)
        return loss

    def _pack_layers(self, layers, st):
        """both operand copies of the given layers rewritten from the fp32 masters: one launch (sl_pack_layers)"""
        if self.planes == 3:  # bf16x3: one launch per layer
            for i in layers:
                p = self.all_plans[i]
                wv, _ = self.layer_param_views(self.params, p)
                wd = self.w_dgrad[p.index]
                pair = wd is not None and p.index == 0 and p is not self.front_plan  # (its pair view: _pack_pair_dgrad_x3)
                self._pack_weights_x3("pack3:" + p.spec.name, wv, self.w_fwd[p.index], wd if (wd is not None and not pair) else None,
                                      p.spec.kernel_size, p.cin_pad, p.cout_pad, st)
                if pair:
                    self._pack_pair_dgrad_x3(st)
            return
        layers = list(layers)
        for lo in range(0, len(layers), 16):
            chunk = layers[lo:lo + 16]
            self._launch("pack:{}..{}".format(self.all_plans[chunk[0]].spec.name, self.all_plans[chunk[-1]].spec.name),
                         "sl_pack_layers", self.params.data_ptr(), self._adam_table(chunk), len(chunk), self.dtype_code,
                         st)
