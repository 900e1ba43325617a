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
                           striding_kernel=48, inner_kernel=7, big_kernel=32, use_raw_wave_input=False, wave_kernel=250,
                           wave_stride=160):
    """Topology of reference net.py:307-330; use_raw_wave_input: `wave_conv` (250 taps at stride 160 over the samples,
    net.py:310-312) in front of striding_conv, which then reads its filters instead of spectrogram bins.  Sizes are
    parameters only so that tests can build shrunken stacks of the same structure."""
    specs = []
    if use_raw_wave_input:
        specs.append(LayerSpec("wave_conv", wave_kernel, wave_stride, input_size_per_time_step, main_filter_count, activation))
        input_size_per_time_step = main_filter_count
    specs.append(LayerSpec("striding_conv", striding_kernel, 2, input_size_per_time_step, main_filter_count, activation))
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
        pl = eng.planes  # 3 on the bf16x3 path: every tensor row holds the planes [hi | lo | hi] (csrc/split3.hip)
        self.batch = batch
        self.tt_pad = tt_pad
        self.t_in = None
        self.t_out = None
        self.rows = HALO + self.tt_pad + HALO
        self.rows0 = 2 * (self.tt_pad + p0.taps_view)
        self.x0 = torch.zeros((batch, self.rows0, p0.cin_pad * pl), dtype=dt, device=dev)
        self.x0_dropped = None  # dropout(x0), allocated by the first training forward with dropout
        self.dropped = False    # the activations of the last forward are post-dropout
        n = len(eng.plans)
        self.y = [None] * (n - 1)
        self._blocks = []  # every halo'd allocation (runs of identical layers are one), for set_length()'s clearing
        # a run of identical layers (the seven inner_conv_i) keeps its inputs y[s-1..e-1] in ONE allocation so that
        # the grouped weight-gradient launch can address layer q as base + q*stride
        for (s0, e0) in eng.runs:
            block = torch.zeros((e0 - s0 + 1, batch, self.rows, eng.plans[s0].cin_pad * pl), dtype=dt, device=dev)
            self._blocks.append(block)
            for q in range(e0 - s0 + 1):
                self.y[s0 - 1 + q] = block[q]
        for p in eng.plans[:-1]:
            if self.y[p.index] is None:
                self.y[p.index] = torch.zeros((batch, self.rows, p.cout_pad * pl), dtype=dt, device=dev)
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
            g.cin = p.cin_view * pl
            g.cout = p.cout_pad
            if p.index == 0:
                g.x_row0 = 0
                g.x_row_stride = p.cin_view * pl
                g.x_batch_stride = self.rows0 * p.cin_pad * pl
            else:
                g.x_row0 = HALO - p.pad_left
                g.x_row_stride = p.cin_pad * pl
                g.x_batch_stride = self.rows * p.cin_pad * pl
            if p.index == n - 1 or pl > 1:  # fp32 out: the logits -- and on the bf16x3 path every layer's staging buffer
                g.y_row0 = 0
                g.y_row_stride = p.cout_pad
                g.y_batch_stride = self.tt_pad * p.cout_pad
            else:
                g.y_row0 = HALO
                g.y_row_stride = p.cout_pad
                g.y_batch_stride = self.rows * p.cout_pad
            self.fwd_geom.append(g)
        # bf16x3: fp32 staging buffer of a layer's pre-activations / input gradients (sl_conv1d_nt out_f32 -> sl_split3)
        self.stage32 = torch.empty((batch * self.tt_pad * max(p.cout_pad for p in eng.plans),), dtype=torch.float32,
                                   device=dev) if pl > 1 else None
        self.plane_geoms = {}  # bf16x3: (kind, layer) -> geometry whose output side describes a plane tensor
        # front layer (raw-wave input): gathered sample windows [B][2 tt_pad][K_pad], the gradient w.r.t. the stack's input in
        # the pair-view layout of x0, and the three geometries of the launches around them (Engine._front_*)
        self.frames = self.gx0 = self.front_geom = self.front_dgrad_geom = None
        if eng.front_plan is not None:
            fp = eng.front_plan
            self.frames = torch.zeros((batch, 2 * tt_pad, fp.cin_pad), dtype=dt, device=dev)
            g = ConvGeom()
            g.batch, g.t_out, g.taps, g.cin, g.cout = batch, 2 * tt_pad, 1, fp.cin_pad, fp.cout_pad
            g.x_row0, g.x_row_stride, g.x_batch_stride = 0, fp.cin_pad, 2 * tt_pad * fp.cin_pad
            g.y_row0, g.y_row_stride, g.y_batch_stride = p0.pad_left, p0.cin_pad, self.rows0 * p0.cin_pad
            self.front_geom = g  # forward (x = frames, y = x0) and weight gradient (x = frames, "y" = gx0): t_out = input frames
        self.half_geoms = {}   # split top (Engine.split_top): (kind, layer) -> the layer's geometry for half the batch
        self.ctc_done = [None, None]  # split top: events behind the CTC launches of the two half-batches
        self.ctc_half_bytes = 0
        self.split_pending = 0        # utterances in the first part if the last forward ran the CTC in two parts (else 0)
        self.wgrad_r = None  # bf16x3: the two partial weight gradients (RA | RB) in front of sl_split3_wgrad_combine
        self.wgrad_geom_b = [None] * n
        self.wgrad_geom = [None] * n
        self.dgrad_geom = [None] * n
        self.bwd_ready = False
        self.nt_ws = None
        self.wgrad_ws = None
        self.launch_lists = {}   # recorded launch lists (Engine._replay); dropped whenever a pointer they hold changes
        self.chain_tables = {}   # sl_conv1d_chain pointer tables of this buffer set (Engine._chain_table)
        self.multi_tables = {}   # sl_conv1d_wgrad_multi job tables (their geometries follow set_length)
        self.wgrad_multi_ws = None
        self._ws_sized_fwd = set()   # output lengths whose forward / backward workspace needs have been checked
        self._ws_sized_bwd = set()   # (a length first seen by predict() and trained on later still gets its dgrad sizing)
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
            if self.gx0 is not None:
                self.gx0[:, p0.pad_left + t_in: p0.pad_left + self._clean_in].zero_()
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
        for table in self.multi_tables.values():
            for job in (table[0] if isinstance(table, tuple) else table):  # (bf16x3: (table, partial buffers, ...))
                job.geom.t_out = t_out
        for g in self.plane_geoms.values():
            g.t_out = t_out
        for g in self.half_geoms.values():
            g.t_out = t_out
        if self.front_geom is not None:
            self.front_geom.t_out = t_in
        if self.front_dgrad_geom is not None:
            self.front_dgrad_geom.t_out = t_out + eng.FRONT_DGRAD_EXTRA_ROWS
        if t_out not in self._ws_sized_fwd:  # split counts (hence workspace sizes) depend on the number of time tiles
            self._ws_sized_fwd.add(t_out)
            self.size_nt_workspace(eng, self.fwd_geom, "fwd")
        if self.bwd_ready and t_out not in self._ws_sized_bwd:
            self.size_backward_workspaces(eng)

    def size_nt_workspace(self, eng, geoms, kind):
        need = 16
        for hint in eng.cu_hints():  # (the split choosers consult sl_set_available_cus: size for every setting in use)
            lib().call("sl_set_available_cus", hint)
            for p, g in zip(eng.plans, geoms):
                if g is not None:
                    need = max(need, lib().raw("sl_conv1d_nt_workspace_bytes")(
                        ctypes.byref(g), eng.dtype_code, eng.nt_cfg.get((kind, p.spec.name), 0)))
        lib().call("sl_set_available_cus", 0)
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
                block = torch.zeros((e0 - lo + 1, self.batch, self.rows, eng.plans[lo].cout_pad * eng.planes), dtype=dt,
                                    device=dev)
                self._blocks.append(block)
                for q in range(e0 - lo + 1):
                    self.g[lo + q] = block[q]
        for p in eng.plans[first:]:
            if self.g[p.index] is None:
                # (layer 0 under a front layer: its input-gradient launch reads up to a time tile past the last utterance's
                # rows -- one utterance of zero slack behind the batch)
                slack = 1 if (p.index == 0 and eng.front_plan is not None) else 0
                full = torch.zeros((self.batch + slack, self.rows, p.cout_pad * eng.planes), dtype=dt, device=dev)
                self.g[p.index] = full[:self.batch]
                self._g0_keepalive = full
                self._blocks.append(self.g[p.index].unsqueeze(0))
            pl = eng.planes
            wg = ConvGeom()
            f = self.fwd_geom[p.index]
            for name, _ in ConvGeom._fields_:
                setattr(wg, name, getattr(f, name))
            wg.y_row0 = HALO
            wg.y_row_stride = p.cout_pad * pl
            wg.y_batch_stride = self.rows * p.cout_pad * pl
            self.wgrad_geom[p.index] = wg
            if pl > 1:
                # bf16x3: two launches.  A: the [hi | lo] prefix of x against g_hi (hh and lh in one (2 Cin) x Cout product),
                # B: x_hi against g_lo (hl); sl_split3_wgrad_combine adds the three blocks.  The pair view of the striding
                # layer has the planes of two frames in a row, so there the whole row is the x operand of both.
                wb = ConvGeom()
                for name, _ in ConvGeom._fields_:
                    setattr(wb, name, getattr(wg, name))
                wg.cin = p.cin_view * pl if p.index == 0 else 2 * p.cin_pad
                wb.cin = p.cin_view * pl if p.index == 0 else p.cin_pad
                self.wgrad_geom_b[p.index] = wb
            if p.index > first:
                dg = ConvGeom()
                dg.batch = self.batch
                dg.t_out = self.t_out
                dg.taps = p.spec.kernel_size
                dg.cin = p.cout_pad * pl
                dg.cout = p.cin_pad
                dg.x_row0 = HALO - p.pad_right
                dg.x_row_stride = p.cout_pad * pl
                dg.x_batch_stride = self.rows * p.cout_pad * pl
                if pl > 1:  # fp32 into the staging buffer, sl_split3 applies the mask and writes the planes
                    dg.y_row0 = 0
                    dg.y_row_stride = p.cin_pad
                    dg.y_batch_stride = self.tt_pad * p.cin_pad
                else:
                    dg.y_row0 = HALO
                    dg.y_row_stride = p.cin_pad
                    dg.y_batch_stride = self.rows * p.cin_pad
                self.dgrad_geom[p.index] = dg
        if eng.front_plan is not None and first == 0 and not eng.front_frozen:
            # dL/d(x0) in x0's own pair-view layout: pair row r = sum over the 24 pair taps j of g0[r - j] . Wpair[j]^T, as an
            # NT launch over g0 with flipped taps.  Frames start at pair row 11, so rows from 7 on are computed: the launch
            # then reads g0 from its first halo row (16 - 23 + 7 = 0) and never in front of the buffer.
            p0 = eng.plans[0]
            self.gx0 = torch.zeros_like(self.x0)
            dg = ConvGeom()
            dg.batch, dg.t_out, dg.taps = self.batch, (self.t_out or 0) + eng.FRONT_DGRAD_EXTRA_ROWS, p0.taps_view
            dg.cin, dg.cout = p0.cout_pad, p0.cin_view
            dg.x_row0, dg.x_row_stride, dg.x_batch_stride = 0, p0.cout_pad, self.rows * p0.cout_pad
            dg.y_row0, dg.y_row_stride, dg.y_batch_stride = eng.FRONT_DGRAD_ROW0, p0.cin_view, self.rows0 * p0.cin_pad
            self.front_dgrad_geom = dg
        self.bias_ws = None
        self.bwd1x1_ws = None
        self.ctc_ws = None
        self.labels = None
        self.label_len = torch.zeros((self.batch,), dtype=torch.int32, device=dev)
        self.bwd_ready = True
        self.size_backward_workspaces(eng)

    def size_backward_workspaces(self, eng):
        L = lib()
        self._ws_sized_bwd.add(self.t_out)
        first = eng.frozen_layer_count
        ws_bytes = 0
        bias_ws = 0
        for hint in eng.cu_hints():
            L.call("sl_set_available_cus", hint)
            for p in eng.plans[first:]:
                wg = self.wgrad_geom[p.index]
                ws_bytes = max(ws_bytes, L.raw("sl_conv1d_wgrad_workspace_bytes")(
                    ctypes.byref(wg), eng.dtype_code, eng.nt_cfg.get(("wgrad", p.spec.name), 0)))
                bias_ws = max(bias_ws, L.raw("sl_bias_grad_workspace_bytes")(ctypes.byref(wg)))
        L.call("sl_set_available_cus", 0)
        self.size_nt_workspace(eng, self.dgrad_geom, "dgrad")
        if eng.front_plan is not None:
            for g in (self.front_geom, self.front_dgrad_geom):
                if g is not None:
                    need = L.raw("sl_conv1d_nt_workspace_bytes")(ctypes.byref(g), eng.dtype_code, 0)
                    if self.nt_ws.numel() < need:
                        self.nt_ws = torch.empty((need,), dtype=torch.uint8, device=eng.device)
                        self.launch_lists = {}
            ws_bytes = max(ws_bytes, L.raw("sl_conv1d_wgrad_workspace_bytes")(ctypes.byref(self.front_geom), eng.dtype_code, 0))
            bias_ws = max(bias_ws, L.raw("sl_bias_grad_workspace_bytes")(ctypes.byref(self.front_geom)))
        if eng.planes > 1:
            need = max(p.taps_view * (self.wgrad_geom[p.index].cin + self.wgrad_geom_b[p.index].cin) * p.cout_pad
                       for p in eng.plans[first:])
            if self.wgrad_r is None or self.wgrad_r.numel() < need:
                self.wgrad_r = torch.empty((need,), dtype=torch.float32, device=eng.device)
            for p in eng.plans[first:]:
                ws_bytes = max(ws_bytes, L.raw("sl_conv1d_wgrad_workspace_bytes")(
                    ctypes.byref(self.wgrad_geom_b[p.index]), eng.dtype_code, 0))
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
        last = len(eng.plans) - 1
        if eng.dtype == "bf16" and last > first:
            # (its split count only shrinks with fewer CUs: the whole-chip size covers every hint)
            need = L.raw("sl_conv1d_backward_1x1_workspace_bytes")(ctypes.byref(self.wgrad_geom[last]),
                                                                   eng.grapheme_set_size, eng.dtype_code, 0)
            # a PART of the batch (Engine.split_top) can pick more chunk ranges than the whole batch does (33 chunks: 11
            # ranges of 3, 32 chunks: 16 of 2): size for the most any chunk count gets -- that of a very long batch
            many = ConvGeom()
            for name, _ in ConvGeom._fields_:
                setattr(many, name, getattr(self.wgrad_geom[last], name))
            many.batch = 4096
            need = max(need, L.raw("sl_conv1d_backward_1x1_workspace_bytes")(ctypes.byref(many), eng.grapheme_set_size,
                                                                             eng.dtype_code, 0))
            if need and (self.bwd1x1_ws is None or self.bwd1x1_ws.numel() < need):
                self.bwd1x1_ws = torch.empty((need,), dtype=torch.uint8, device=eng.device)
                self.launch_lists = {}

    def ensure_ctc(self, eng, l_max):
        """CTC workspace for label rows of up to l_max graphemes: sized in BYTES and never shrunk (the library's need
        is monotonic in l_max since round 3, but a buffer set that has served long labels keeps its allocation)."""
        need = lib().raw("sl_ctc_workspace_bytes")(self.batch, self.tt_pad, l_max)  # covers every length
        if eng.split_top and self.batch >= 2:  # two parts of the batch at a time (up to B - 1 utterances), each with its own workspace
            half = _round_up(lib().raw("sl_ctc_workspace_bytes")(self.batch - 1, self.tt_pad, l_max), 256)
            self.ctc_half_bytes = max(self.ctc_half_bytes, half)
            need = max(need, 2 * self.ctc_half_bytes)
        if self.ctc_ws is None or self.ctc_ws.numel() < need:
            self.ctc_ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=eng.device)
            self.launch_lists = {}
        if self.labels is None or self.labels.shape[1] < l_max:
            self.labels = torch.zeros((self.batch, l_max), dtype=torch.int32, device=eng.device)


class Engine:
    """Forward / CTC / backward / Adam on one MI355X.  dtype 'bf16' (bf16 storage, fp32 accumulate, fp32 CTC: the
    benchmarked path), 'f32' (parity path: fp32 storage, exact-fp32 MFMA) or 'bf16x3' (the fast parity path: every value as
    hi + lo bf16 planes, three bf16 MFMA terms per product, fp32 accumulate; csrc/split3.hip)."""

    FRONT_DGRAD_ROW0 = 7          # first pair row of x0's gradient that is computed (frames start at pair row 11)
    FRONT_DGRAD_EXTRA_ROWS = 17   # pair rows computed beyond the output frames: up to row T' + 23 = the last frame's

    def __init__(self, specs, grapheme_set_size, dtype="bf16", device="cuda:0", ctc_epsilon=1e-8,
                 frozen_layer_count=0, lr=1e-4, beta_1=0.9, beta_2=0.999, adam_epsilon=1e-8):
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
        else:
            raise ValueError("dtype must be 'bf16', 'f32' or 'bf16x3'")
        # Raw-wave input (reference net.py:310-312: `wave_conv`, 250 taps at stride 160 over the samples, in front of
        # striding_conv): the FRONT layer.  It is a GEMM over gathered sample windows (sl_wave_frames: K = 250 * Cin
        # columns per output frame, 3 GFLOP per 32 x 8 s -- nothing next to the stack) whose output lands directly in the
        # pair-view input buffer of the stack below; the eleven layers behind it, their plans, indices and launches are
        # untouched.  Internally the front plan has index len(plans) (its parameters sit at the END of the flat buffers);
        # the public order of set_weights / get_weights / get_gradients puts it first, as the reference's layer list does.
        self.front_spec = None
        self.front_frozen = False
        if specs and specs[0].stride > 2:
            if dtype == "bf16x3":
                raise NotImplementedError("raw-wave input is implemented on the bf16 and f32 paths")
            self.front_spec = specs[0]
            specs = list(specs[1:])
            self.front_frozen = frozen_layer_count >= 1
            frozen_layer_count = max(frozen_layer_count - 1, 0)
            if self.front_spec.activation not in ("relu", "elu"):
                raise NotImplementedError("the raw-wave layer takes a relu / elu activation")
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
        cin_pad = _round_up(specs[0].cin, 64)
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
        self.grads = torch.zeros((off,), dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros((off,), dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros((off,), dtype=torch.float32, device=dev)
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
            buf.size_nt_workspace(self, buf.fwd_geom, "fwd")
            if buf.bwd_ready:
                buf.size_backward_workspaces(self)

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
            elif kind == 2:
                callback(op[1])
            else:  # a step of the sequence that has to be marshalled afresh every time (pointers / sizes that change per batch)
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
            groups = [g for g in groups if not set(g) & span] + [merged]
            assert merged == list(range(merged[0], merged[-1] + 1)), merged
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

    def _public_plans(self):
        """the plans in the public layer order (the reference's layer list): the front layer first"""
        return ([self.front_plan] if self.front_plan is not None else []) + self.plans

    def _front_to_gemm(self, plan, w):
        """the front layer's Keras kernel (k, Cin, Cout) as the (1, k * Cin, Cout) matrix it is stored and launched as"""
        if plan is self.front_plan and np.ndim(w) == 3 and w.shape[0] == self.front_spec.kernel_size:
            return np.reshape(w, (1, w.shape[0] * w.shape[1], w.shape[2]))
        return w

    def _front_from_gemm(self, plan, w):
        if plan is self.front_plan:
            fs = self.front_spec
            return np.reshape(w, (fs.kernel_size, fs.cin, fs.cout))
        return w

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

    def _repack_weights_x3(self):
        """bf16x3 operand copies: rows [w_hi | w_hi | w_lo] in both operand layouts, w_hi = bf16(w), w_lo = bf16(w - w_hi),
        one launch per layer (sl_split3_pack_weights; the five-launch sequence it replaces -- split, two packs, two
        assembles -- was 55 launches and 0.45 ms of a 7.1 ms optimisation step)."""
        st = self._stream()
        for p in self.plans:
            wv, _ = self.layer_param_views(self.params, p)
            wd = self.w_dgrad[p.index]
            self._launch("pack3:" + p.spec.name, "sl_split3_pack_weights", wv.data_ptr(), self.w_fwd[p.index].data_ptr(),
                         wd.data_ptr() if wd is not None else None, p.spec.kernel_size, p.cin_pad, p.cout_pad, st)
        self._packed_dirty = False

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
        if p.index == 0 and p.spec.stride == 2 and self.w_dgrad[0] is not None:
            return p.taps_view, p.cin_view
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
            self._launch("pack_input", "sl_split3_pack_input", src.data_ptr(), buf.x0.data_ptr(), batch, t_in, f, p0.cin_pad,
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

    # ------------------------------------------------------------------ front layer (raw-wave input, see __init__)
    def _load_front_input(self, src):
        """src: float32 (B, T, Cin) samples in HBM.  Gathers the sample windows of the ceil(T / stride) output frames (the
        weight-independent half of the front layer; its GEMM is part of forward())."""
        fs = self.front_spec
        batch, t_audio, f = src.shape
        if f != fs.cin:
            raise ValueError("input has {} values per sample, the net expects {}".format(f, fs.cin))
        t1, pad_l, _ = same_padding(t_audio, fs.kernel_size, fs.stride)
        buf = self.buffers(batch, t1)
        buf.front_src, buf.front_geometry = src, (t_audio, t1, pad_l)
        self._front_gather(buf, src)
        self.cur = buf
        self._src_keepalive = src
        return buf

    def _front_gather(self, buf, src):
        fs, fp = self.front_spec, self.front_plan
        t_audio, t1, pad_l = buf.front_geometry
        self._launch("wave_frames", "sl_wave_frames", src.data_ptr(), buf.frames.data_ptr(), buf.batch, t_audio, fs.cin,
                     fs.kernel_size, fs.stride, pad_l, t1, fp.cin_pad, buf.frames.stride(0), self.dtype_code, self._stream())
        buf.front_frames_dropped = False

    def _front_forward(self, buf, rate, seed0, st):
        """wave_conv: bias + activation epilogue of a 1 x 1 NT launch over the gathered windows, written into x0 where
        sl_pack_input would have put a spectrogram.  With dropout (a Dropout sits in front of wave_conv too, net.py:301-303)
        the samples are dropped first (seed offset 63: the stack's layers use 0 .. n) and the windows gathered again."""
        fp = self.front_plan
        if rate:
            if getattr(buf, "front_src_dropped", None) is None or buf.front_src_dropped.shape != buf.front_src.shape:
                buf.front_src_dropped = torch.empty_like(buf.front_src)
            self._launch("dropout:samples", "sl_dropout", buf.front_src.data_ptr(), buf.front_src_dropped.data_ptr(),
                         buf.front_src.numel(), _lib.SL_F32, rate, seed0 + 63, st)
            self._front_gather(buf, buf.front_src_dropped)
            buf.front_frames_dropped = True
        elif buf.front_frames_dropped:
            self._front_gather(buf, buf.front_src)
        _, bias = self.layer_param_views(self.params, fp)
        self._launch("fwd:" + fp.spec.name, "sl_conv1d_nt", buf.frames.data_ptr(), self.w_fwd[fp.index].data_ptr(),
                     bias.data_ptr(), None, buf.x0.data_ptr(), ctypes.byref(buf.front_geom),
                     _lib.EPI_BIAS_ELU if fp.spec.activation == "elu" else _lib.EPI_BIAS_RELU, self.dtype_code, 0, 0,
                     buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)

    def _front_backward(self, buf, st):
        """behind the stack's backward: x0's gradient (pair-view NT launch over g[0] with the flipped pair taps, masked by
        wave_conv's stored activation), then wave_conv's weight and bias gradient from the gathered windows"""
        fp, p0 = self.front_plan, self.plans[0]
        x0 = buf.x0_dropped if buf.dropped else buf.x0
        elu = fp.spec.activation == "elu"
        elu_dropped = elu and buf.dropped
        self._launch("dgrad:" + p0.spec.name, "sl_conv1d_nt", buf.g[0].data_ptr(), self.w_dgrad[0].data_ptr(), None,
                     None if elu_dropped else x0.data_ptr(), buf.gx0.data_ptr(), ctypes.byref(buf.front_dgrad_geom),
                     _lib.EPI_NONE if elu_dropped else (_lib.EPI_ELU_MASK if elu else _lib.EPI_RELU_MASK), self.dtype_code, 0,
                     0, buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
        if elu_dropped:
            self._launch("dropout_elu_bwd:" + p0.spec.name, "sl_elu_dropout_backward", buf.gx0.data_ptr(), x0.data_ptr(),
                         buf.gx0.numel(), self.dtype_code, self.dropout_rate, buf.dropout_seed0, st)
        elif buf.dropped:
            self._launch("dropout_scale:" + p0.spec.name, "sl_scale", buf.gx0.data_ptr(), buf.gx0.numel(), self.dtype_code,
                         1.0 / (1.0 - self.dropout_rate), st)
        dw, db = self.layer_param_views(self.grads, fp)
        self._launch("wgrad:" + fp.spec.name, "sl_conv1d_wgrad", buf.frames.data_ptr(), buf.gx0.data_ptr(), dw.data_ptr(),
                     ctypes.byref(buf.front_geom), self.dtype_code, 0, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
        self._launch("bgrad:" + fp.spec.name, "sl_bias_grad", buf.gx0.data_ptr(), db.data_ptr(), ctypes.byref(buf.front_geom),
                     self.dtype_code, buf.bias_ws.data_ptr(), buf.bias_ws.numel(), st)

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
        if self.front_plan is not None and not rate and getattr(buf, "front_frames_dropped", False):
            self._front_gather(buf, buf.front_src)  # (the last step gathered DROPPED samples; not part of any launch list)
        fuse_out = self.fuse_output_softmax and self.dtype == "bf16" and bool(self.lib.raw("sl_output_softmax_supported")(
            ctypes.byref(buf.fwd_geom[n - 1]), self.grapheme_set_size, self.dtype_code))
        # launch list (no dropout): everything below takes its frame count from the geometries, except the unfused
        # softmax, which gets it by value -> then the list is per length
        key = None if rate else ("fwd", st, fuse_out, self.use_chain, tuple(sorted(self.nt_cfg.items())),
                                 None if fuse_out else buf.t_out, split_ctc)
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

    # ------------------------------------------------------------------ split top (see self.split_top)
    def split_top_plan(self, buf):
        """How the training step on `buf` runs its top three layers and the CTC in two parts of the batch: the number of
        utterances in the first part, or 0 (whole-batch step)."""
        state = (self.split_top, bool(self.dropout_rate), self.frozen_layer_count, self.fuse_output_softmax,
                 self.fuse_output_backward, tuple(sorted(self.nt_cfg)), buf.bwd_ready, buf.bwd1x1_ws is not None,
                 buf.labels is not None)
        if getattr(buf, "_split_ok", (None, None))[0] != state:
            buf._split_ok = (state, self._split_top_ok(buf))
        return self._split_parts(buf) if buf._split_ok[1] else 0

    def _split_parts(self, buf):
        """Measured rule (MI355X, tools/split_by_bucket.py, rocprofv3 kernel traces under profiles/r04_trace_*): the top
        layers run 256 x 256 tiles, ONE work-group per CU, and a CTC lattice wave cannot share a CU with such a work-group
        (its registers fill the SIMDs) -- so while a part's lattice runs, 3 waves per utterance hold CUs of their own and a
        launch of exactly 256 tiles needs a second round.  The split pays where the two launches of the widest layer, each
        next to the other part's lattice waves, take no more rounds of 256 work-groups than the whole-batch launch: config
        5's buckets of 384 / 896 / 960 tiles in halves (-0.09 / -0.22 / -0.09 ms), 640 tiles as 3 + 5 utterances (240 + 400
        tiles = 1 + 2 rounds), not 512 / 768 / 1024 (whole rounds already: +0.05 ... +0.11 ms when halved) and not config
        3 (512 tiles: +0.10 ms).  Among the splits that qualify the most even one is taken (equal halves also keep an
        utterance's results independent of the part it is in: the two launches then pick the same K split)."""
        b = buf.batch
        if self.split_min_tiles is not None:  # measurement hook: halves, from a tile count on (0 = always)
            return b // 2 if (b % 2 == 0 and self._top_tiles(buf) >= self.split_min_tiles) else 0
        cus = 256
        per_utt = self._top_tiles(buf) // b
        whole = -(-(b * per_utt) // cus)
        best, best_key = 0, None
        for a in range(1, b):
            rounds = -(-(a * per_utt + 3 * (b - a)) // cus) + -(-((b - a) * per_utt + 3 * a) // cus)
            if rounds <= whole:
                key = (abs(2 * a - b), a)
                if best_key is None or key < best_key:
                    best, best_key = a, key
        return best

    def _top_tiles(self, buf):
        widest = max(self.plans[i].cout_pad for i in range(len(self.plans) - 3, len(self.plans) - 1))
        return buf.batch * (-(-buf.t_out // 256)) * (-(-widest // 256))

    def _split_top_ok(self, buf):
        n = len(self.plans)
        if not self.split_top or self.dtype != "bf16" or self.dropout_rate or n < 4 or buf.batch < 2:
            return False
        if self.frozen_layer_count >= n - 3 or not self.fuse_output_softmax or not self.fuse_output_backward:
            return False
        top = (n - 3, n - 2, n - 1)
        if any(s0 <= i <= e0 for (s0, e0) in self.runs for i in top) or any(self.plans[i].spec.stride != 1 for i in top):
            return False
        if any((kind, self.specs[i].name) in self.nt_cfg for kind in ("fwd", "dgrad", "wgrad") for i in top):
            return False
        if any(self.specs[i].activation not in ("relu", "elu") for i in (n - 4, n - 3, n - 2)):
            return False
        if not buf.bwd_ready or buf.bwd1x1_ws is None or buf.labels is None:
            return False
        return bool(self.lib.raw("sl_output_softmax_supported")(ctypes.byref(buf.fwd_geom[n - 1]), self.grapheme_set_size,
                                                                self.dtype_code)) \
            and bool(self.lib.raw("sl_conv1d_backward_1x1_supported")(ctypes.byref(buf.wgrad_geom[n - 1]),
                                                                        self.grapheme_set_size, self.dtype_code))

    def _part_geom(self, buf, kind, i, count):
        """geometry of layer i (kind 'fwd' / 'dgrad' / 'wgrad') for `count` utterances; follows set_length like the others"""
        g = buf.half_geoms.get((kind, i, count))
        if g is None:
            src = {"fwd": buf.fwd_geom, "dgrad": buf.dgrad_geom, "wgrad": buf.wgrad_geom}[kind][i]
            g = ConvGeom()
            for name, _ in ConvGeom._fields_:
                setattr(g, name, getattr(src, name))
            g.batch = count
            buf.half_geoms[(kind, i, count)] = g
            if kind in ("fwd", "dgrad"):  # (a part of the batch may pick more K splits: make sure the workspace covers it)
                need = self.lib.raw("sl_conv1d_nt_workspace_bytes")(ctypes.byref(g), self.dtype_code, 0)
                if buf.nt_ws is None or buf.nt_ws.numel() < need:
                    buf.nt_ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
                    buf.launch_lists = {}
        return g

    @staticmethod
    def _utt_ptr(t, first):
        """address of utterance `first` of a tensor whose first dimension is the batch"""
        return t.data_ptr() + first * t.stride(0) * t.element_size()

    def _forward_top_split(self, buf, x, st, grad_scale, a):
        """big_conv_1, big_conv_2, output_conv + softmax and the CTC, part by part: utterances [0, a), then [a, B)"""
        n = len(self.plans)
        for h, (first, count) in enumerate(((0, a), (a, buf.batch - a))):
            xin = x
            for i in (n - 3, n - 2):
                p = self.plans[i]
                _, bias = self.layer_param_views(self.params, p)
                y = buf.y[i]
                self._launch("fwd:" + p.spec.name, "sl_conv1d_nt", self._utt_ptr(xin, first), self.w_fwd[i].data_ptr(),
                             bias.data_ptr(), None, self._utt_ptr(y, first),
                             ctypes.byref(self._part_geom(buf, "fwd", i, count)),
                             _lib.EPI_BIAS_ELU if p.spec.activation == "elu" else _lib.EPI_BIAS_RELU, self.dtype_code, 0, 0,
                             buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
                xin = y
            self._eager_op(self._top_part_tail, buf, h, first, count, grad_scale)
        buf.split_pending = a
        return buf.probs

    def _top_part_tail(self, buf, h, first, count, grad_scale):
        """output layer + softmax of one part on the main stream, then its CTC loss + gradient on a side stream.  Marshalled
        afresh every step: the dense probability tensors (a part's offset depends on the frame count), the label tensors
        (the staged pipeline hands over new ones per batch) and the label width are per-batch values."""
        n = len(self.plans)
        last = n - 1
        p = self.plans[last]
        k = self.grapheme_set_size
        main = torch.cuda.current_stream(self.device)
        if self._ctc_streams is None:
            self._ctc_streams = [torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)]
        side = self._ctc_streams[h]
        _, bias = self.layer_param_views(self.params, p)
        dense = first * buf.t_out * k * 4  # probs / log q: [B][T'][K] floats
        self._launch("fwd:" + p.spec.name, "sl_output_softmax", self._utt_ptr(buf.y[last - 1], first),
                     self.w_fwd[last].data_ptr(), bias.data_ptr(), buf.probs.data_ptr() + dense, buf.logq.data_ptr() + dense,
                     None, ctypes.byref(self._part_geom(buf, "fwd", last, count)), k, p.cout_pad, buf.tt_pad * p.cout_pad,
                     self.ctc_epsilon, self.dtype_code, main.cuda_stream)
        ready = torch.cuda.Event()
        ready.record(main)
        labels = buf.labels if buf.labels.is_contiguous() else buf.labels.contiguous()
        l_max = labels.shape[1]
        with torch.cuda.stream(side):
            side.wait_event(ready)
            self._launch("ctc", "sl_ctc_loss_grad", buf.probs.data_ptr() + dense, buf.logq.data_ptr() + dense,
                         labels.data_ptr() + first * l_max * 4, buf.label_len.data_ptr() + first * 4,
                         buf.input_len.data_ptr() + first * 4, buf.loss.data_ptr() + first * 4,
                         self._utt_ptr(buf.g[last], first), count, buf.t_out, k, l_max, HALO, p.cout_pad,
                         buf.rows * p.cout_pad, self.dtype_code, self.ctc_epsilon, grad_scale,
                         buf.ctc_ws.data_ptr() + h * buf.ctc_half_bytes, buf.ctc_half_bytes, side.cuda_stream)
            done = torch.cuda.Event()
            done.record(side)
        buf.ctc_done[h] = done
        buf._split_labels_keepalive = labels

    def _wait_ctc_half(self, buf, h, main):
        main.wait_event(buf.ctc_done[h])

    def _backward_top_split(self, buf, main, a):
        """the input gradients of the top three layers (and the output layer's weight gradient) part by part, each part
        behind its own CTC: the second part's lattice runs under the first part's launches here"""
        n = len(self.plans)
        st = main.cuda_stream
        last = n - 1
        for h, (first, count) in enumerate(((0, a), (a, buf.batch - a))):
            self._eager_op(self._wait_ctc_half, buf, h, main)
            p = self.plans[last]
            dw, _ = self.layer_param_views(self.grads, p)
            epi = _lib.EPI_ELU_MASK if self.specs[last - 1].activation == "elu" else _lib.EPI_RELU_MASK
            self._launch("bwd:" + p.spec.name, "sl_conv1d_backward_1x1_part", self._utt_ptr(buf.y[last - 1], first),
                         self._utt_ptr(buf.g[last], first), self.w_dgrad[last].data_ptr(),
                         self._utt_ptr(buf.g[last - 1], first), dw.data_ptr(),
                         ctypes.byref(self._part_geom(buf, "wgrad", last, count)), epi, self.grapheme_set_size,
                         self.dtype_code, 0, h, buf.bwd1x1_ws.data_ptr(), buf.bwd1x1_ws.numel(), st)
            for i in (n - 2, n - 3):
                q = self.plans[i]
                elu = self.specs[i - 1].activation == "elu"
                self._launch("dgrad:" + q.spec.name, "sl_conv1d_nt", self._utt_ptr(buf.g[i], first), self.w_dgrad[i].data_ptr(),
                             None, self._utt_ptr(buf.y[i - 1], first), self._utt_ptr(buf.g[i - 1], first),
                             ctypes.byref(self._part_geom(buf, "dgrad", i, count)),
                             _lib.EPI_ELU_MASK if elu else _lib.EPI_RELU_MASK, self.dtype_code, 0, 0, buf.nt_ws.data_ptr(),
                             buf.nt_ws.numel(), st)

    def _plane_geom(self, buf, kind, i, channels):
        """the NT geometry of layer i (kind 'fwd' / 'dgrad') with its OUTPUT side describing a bf16x3 plane tensor of
        `channels` padded channels (rows of 3 x channels behind HALO halo rows) instead of the fp32 staging buffer"""
        g = buf.plane_geoms.get((kind, i))
        if g is None:
            src = (buf.fwd_geom if kind == "fwd" else buf.dgrad_geom)[i]
            g = ConvGeom()
            for name, _ in ConvGeom._fields_:
                setattr(g, name, getattr(src, name))
            g.y_row0, g.y_row_stride, g.y_batch_stride = HALO, self.planes * channels, buf.rows * channels * self.planes
            buf.plane_geoms[(kind, i)] = g
        return g

    def _dropout_x3(self, tag, src, dst, y, channels, mode, seed, st):
        """sl_split3_dropout over a whole plane tensor (halo rows and padding included: zeros stay zeros)"""
        self._launch(tag, "sl_split3_dropout", src.data_ptr(), dst.data_ptr(), y.data_ptr() if y is not None else None,
                     src.numel() // (self.planes * channels), channels, mode, self.dropout_rate, seed, st)

    def _forward_x3(self, buf, st, rate=None):
        """bf16x3: every layer = the unchanged NT kernel over the three planes.  ReLU layers: bias, ReLU and the split into
        planes in the kernel's own epilogue (out_f32 = 2); ELU layers: fp32 into the staging buffer + sl_split3.  The last
        layer's fp32 logits go to the softmax as on the other paths.  Dropout (training, net.py:301-303): sl_split3_dropout
        on the packed input (into a second buffer) and in place on every activation that feeds a layer with a Dropout in
        front of it -- the same (seed, element) keep decisions as sl_dropout draws on the single-plane paths."""
        n = len(self.plans)
        x = buf.x0
        if rate:
            self._dropout_steps += 1
            seed0 = buf.dropout_seed0 = (self.dropout_seed * 1000003 + self._dropout_steps) * 64
            if buf.x0_dropped is None:
                buf.x0_dropped = torch.zeros_like(buf.x0)
            self._dropout_x3("dropout:input", buf.x0, buf.x0_dropped, None, self.plans[0].cin_pad, 0, seed0, st)
            x = buf.x0_dropped

        def drop(p):
            if rate and (p.index + 1) in self._dropout_layers():
                y = buf.y[p.index]
                self._dropout_x3("dropout:" + p.spec.name, y, y, None, p.cout_pad, 0, seed0 + p.index + 1, st)

        for p in self.plans:
            last = p.index == n - 1
            _, bias = self.layer_param_views(self.params, p)
            cfg = self.nt_cfg.get(("fwd", p.spec.name), 0)
            if not last and p.spec.activation == "relu" and self.x3_fused_epilogue:
                y = buf.y[p.index]
                self._launch("fwd:" + p.spec.name, "sl_conv1d_nt", x.data_ptr(), self.w_fwd[p.index].data_ptr(),
                             bias.data_ptr(), None, y.data_ptr(), ctypes.byref(self._plane_geom(buf, "fwd", p.index, p.cout_pad)),
                             _lib.EPI_BIAS_RELU, self.dtype_code, 2, cfg, buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
                drop(p)
                x = y
                continue
            out = buf.logits if last else buf.stage32
            self._launch("fwd:" + p.spec.name, "sl_conv1d_nt", x.data_ptr(), self.w_fwd[p.index].data_ptr(), bias.data_ptr(),
                         None, out.data_ptr(), ctypes.byref(buf.fwd_geom[p.index]), _lib.EPI_BIAS, self.dtype_code, 1,
                         cfg, buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
            if not last:
                y = buf.y[p.index]
                self._launch("split:" + p.spec.name, "sl_split3", buf.stage32.data_ptr(), y.data_ptr(), None, buf.batch,
                             buf.t_out, p.cout_pad, buf.tt_pad * p.cout_pad, HALO, buf.rows * p.cout_pad * self.planes,
                             2 if p.spec.activation == "elu" else 1, st)
                drop(p)
                x = y
        self._launch("softmax", "sl_softmax_logq", buf.logits.data_ptr(), buf.probs.data_ptr(), buf.logq.data_ptr(), buf.batch,
                     buf.t_out, self.grapheme_set_size, self.plans[-1].cout_pad, buf.tt_pad * self.plans[-1].cout_pad,
                     self.ctc_epsilon, st)
        return buf.probs

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
                         self.ctc_epsilon, grad_scale, buf.ctc_ws.data_ptr(), buf.ctc_ws.numel(), self._stream())
            self._launch("split:ctc", "sl_split3", buf.stage32.data_ptr(), buf.g[last].data_ptr(), None, buf.batch, buf.t_out,
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
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)  # (ROCm offers no priority below the default)
        side = self._side_stream
        # launch list: not with dropout (its scale passes take the rate by value)
        key = None if buf.dropped else ("bwd", main.cuda_stream, buf.split_pending, on_bucket_ready is not None, self.ones_channel,
                                        self.frozen_layer_count, self.group_wgrad, self.use_chain, self.fuse_output_backward,
                                        self.use_wgrad_multi, self.multi_overlap_tiles, self.small_bias_pass_on_main,
                                        tuple(sorted(self.nt_cfg.items())),
                                        buf.t_out if self.planes > 1 else None)  # (bf16x3 helpers take it by value)
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
        for (s0, e0) in self.runs:
            layers = list(range(e0, max(s0, first + 1) - 1, -1))
            if len(layers) >= 2 and all(self.specs[i - 1].activation == "relu" for i in layers) and \
                    not any(("dgrad", self.specs[i].name) in self.nt_cfg for i in layers) and \
                    bool(self.lib.raw("sl_conv1d_chain_supported")(ctypes.byref(buf.dgrad_geom[e0]), len(layers),
                                                                   self.dtype_code)):
                dchain[e0] = layers
                skip.update(layers[1:])
        return dchain, skip

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

    def _backward_x3(self, buf, st, on_bucket_ready=None):
        """bf16x3 backward: per layer the weight gradient of the [hi | lo] prefixes + sl_split3_wgrad_combine, the input
        gradient through the unchanged NT kernel (fp32 staging) + sl_split3 with the activation mask.  Bias gradients:
        row cin_pad - 1 of dW where the input carries the ones channel (hi = 1, lo = 0), sl_split3_bias_grad elsewhere (and
        everywhere when dropout touched the ones).  Everything runs on ONE stream, so a gradient bucket (bucket_plan) is
        complete the moment the launches of its lowest layer are enqueued: on_bucket_ready(b) is called there, exactly as
        _backward_eager does for the single-plane paths."""
        first = self.frozen_layer_count
        pl = self.planes
        ones_in = self._ones_input_layers(first)
        ones_db = set() if buf.dropped else set(ones_in)  # rows that hold a bias gradient (else: only to be zeroed)
        main = torch.cuda.current_stream(self.device)
        bucket_at = {}
        if on_bucket_ready is not None:
            for b, (layers, _) in enumerate(self.bucket_plan()):
                bucket_at[layers[0]] = (b, layers)
        # the runs of identical layers (inner_conv_1..7): their 2 x 7 partial weight gradients (x planes against g_hi, against
        # g_lo) in ONE balanced launch (sl_conv1d_wgrad_multi, a job per partial) at the lowest layer of the run -- they
        # were 14 launches of 31 us + their reductions, 0.6 ms of the 6.8 ms step
        multi = {}
        if self.use_wgrad_multi:
            for (s0, e0) in self.runs:
                lo = max(s0, first)
                layers = list(range(lo, e0 + 1))
                if len(layers) >= 2 and 2 * len(layers) <= 16 and all(
                        buf.wgrad_geom[i].cin % 256 == 0 and buf.wgrad_geom_b[i].cin % 256 == 0 and
                        self.plans[i].cout_pad % 256 == 0 and self.plans[i].spec.stride == 1 and
                        ("wgrad", self.specs[i].name) not in self.nt_cfg for i in layers):
                    for i in layers:
                        multi[i] = layers

        def combine(p, ra, rb):
            dw, _ = self.layer_param_views(self.grads, p)
            frames = 2 if p.spec.stride == 2 else 1
            self._launch("combine:" + p.spec.name, "sl_split3_wgrad_combine", ra.data_ptr(), rb.data_ptr(), dw.data_ptr(),
                         p.spec.kernel_size, p.cin_pad, p.cout_pad, frames, pl * p.cin_pad if frames == 2 else 0,
                         buf.wgrad_geom[p.index].cin, buf.wgrad_geom_b[p.index].cin, st)

        for p in reversed(self.plans[first:]):
            i = p.index
            x = (buf.x0_dropped if buf.dropped else buf.x0) if i == 0 else buf.y[i - 1]
            dw, db = self.layer_param_views(self.grads, p)
            wa, wb = buf.wgrad_geom[i], buf.wgrad_geom_b[i]
            if i in multi:
                if i == multi[i][0]:  # every gradient tensor of the run is complete here
                    self._launch_wgrad_multi_x3(buf, multi[i], st, combine)
            else:
                ra = buf.wgrad_r
                rb = buf.wgrad_r[p.taps_view * wa.cin * p.cout_pad:]
                g_lo = buf.g[i].data_ptr() + p.cout_pad * 2  # plane P1 of every row
                cfg = self.nt_cfg.get(("wgrad", p.spec.name), 0)
                self._launch("wgrad:" + p.spec.name, "sl_conv1d_wgrad", x.data_ptr(), buf.g[i].data_ptr(), ra.data_ptr(),
                             ctypes.byref(wa), self.dtype_code, cfg, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
                self._launch("wgrad_lo:" + p.spec.name, "sl_conv1d_wgrad", x.data_ptr(), g_lo, rb.data_ptr(),
                             ctypes.byref(wb), self.dtype_code, cfg, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
                combine(p, ra, rb)
            if i not in ones_db:
                if self._x3_bias_ws is None:
                    self._x3_bias_ws = torch.empty((self.lib.raw("sl_split3_bias_grad_workspace_bytes")(
                        max(q.cout_pad for q in self.plans)),), dtype=torch.uint8, device=self.device)
                self._launch("bgrad:" + p.spec.name, "sl_split3_bias_grad", buf.g[i].data_ptr(), db.data_ptr(), buf.batch,
                             buf.t_out, p.cout_pad, HALO, buf.rows * p.cout_pad * pl, self._x3_bias_ws.data_ptr(),
                             self._x3_bias_ws.numel(), st)
            if i in bucket_at:
                b, layers = bucket_at[i]
                rows = [j for j in layers if j in ones_in]
                if rows:
                    self._bias_grads_from_wgrad(rows, bool(ones_db), main)
                on_bucket_ready(b)
                if self._rec is not None:
                    self._rec.append((2, b))
            dropped_in = buf.dropped and i in self._dropout_layers()  # a Dropout sits between y[i - 1] and layer i
            if i > first and self.specs[i - 1].activation == "elu" and dropped_in:
                # a stored zero is ambiguous behind an ELU: plain input gradient, then both factors of the chain rule with the
                # keep decisions recomputed from the step's seed (cf. sl_elu_dropout_backward)
                self._launch("dgrad:" + p.spec.name, "sl_conv1d_nt", buf.g[i].data_ptr(), self.w_dgrad[i].data_ptr(), None,
                             None, buf.stage32.data_ptr(), ctypes.byref(buf.dgrad_geom[i]), _lib.EPI_NONE, self.dtype_code, 1,
                             self.nt_cfg.get(("dgrad", p.spec.name), 0), buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
                self._launch("split:dgrad:" + p.spec.name, "sl_split3", buf.stage32.data_ptr(), buf.g[i - 1].data_ptr(),
                             None, buf.batch, buf.t_out, p.cin_pad, buf.tt_pad * p.cin_pad, HALO,
                             buf.rows * p.cin_pad * pl, 0, st)
                self._dropout_x3("dropout_elu_bwd:" + p.spec.name, buf.g[i - 1], buf.g[i - 1], buf.y[i - 1], p.cin_pad, 2,
                                 buf.dropout_seed0 + i, st)
                continue
            if i > first and self.specs[i - 1].activation == "relu" and self.x3_fused_epilogue:
                # the ReLU mask (the hi plane of the stored activation) and the split into planes in the NT kernel's epilogue
                self._launch("dgrad:" + p.spec.name, "sl_conv1d_nt", buf.g[i].data_ptr(), self.w_dgrad[i].data_ptr(), None,
                             buf.y[i - 1].data_ptr(), buf.g[i - 1].data_ptr(),
                             ctypes.byref(self._plane_geom(buf, "dgrad", i, p.cin_pad)), _lib.EPI_RELU_MASK, self.dtype_code,
                             2, self.nt_cfg.get(("dgrad", p.spec.name), 0), buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
            elif i > first:
                self._launch("dgrad:" + p.spec.name, "sl_conv1d_nt", buf.g[i].data_ptr(), self.w_dgrad[i].data_ptr(), None,
                             None, buf.stage32.data_ptr(), ctypes.byref(buf.dgrad_geom[i]), _lib.EPI_NONE, self.dtype_code, 1,
                             self.nt_cfg.get(("dgrad", p.spec.name), 0), buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
                self._launch("split:dgrad:" + p.spec.name, "sl_split3", buf.stage32.data_ptr(), buf.g[i - 1].data_ptr(),
                             buf.y[i - 1].data_ptr(), buf.batch, buf.t_out, p.cin_pad, buf.tt_pad * p.cin_pad, HALO,
                             buf.rows * p.cin_pad * pl, 4 if self.specs[i - 1].activation == "elu" else 3, st)
            if i > first and dropped_in:
                # the ReLU mask (stored activation > 0) already applied the keep mask: the stored activation is the
                # post-dropout one; what is left of d dropout / dx is the factor 1 / (1 - rate)
                self._dropout_x3("dropout_scale:" + p.spec.name, buf.g[i - 1], buf.g[i - 1], None, p.cin_pad, 1, 0, st)
        if on_bucket_ready is None and ones_in:
            self._bias_grads_from_wgrad(ones_in, bool(ones_db), main)

    def _launch_wgrad_multi_x3(self, buf, layers, st, combine):
        """bf16x3: the partial weight gradients RA (x planes [hi | lo] against g_hi) and RB (x plane hi against g_lo) of
        every layer of a run as jobs of one sl_conv1d_wgrad_multi launch, then sl_split3_wgrad_combine per layer"""
        key = ("x3",) + tuple(layers)
        entry = buf.multi_tables.get(key)
        if entry is None:
            sizes = [(self.plans[i].taps_view * buf.wgrad_geom[i].cin * self.plans[i].cout_pad,
                      self.plans[i].taps_view * buf.wgrad_geom_b[i].cin * self.plans[i].cout_pad) for i in layers]
            scratch = torch.empty((sum(a + b for a, b in sizes),), dtype=torch.float32, device=self.device)
            table = (_lib.WgradJob * (2 * len(layers)))()
            parts, off = [], 0
            for n, (i, (na, nb)) in enumerate(zip(layers, sizes)):
                ra, rb = scratch[off:off + na], scratch[off + na:off + na + nb]
                off += na + nb
                parts.append((ra, rb))
                x = buf.y[i - 1]
                for job, (g_ptr, out, geom) in zip((table[2 * n], table[2 * n + 1]),
                                                   ((buf.g[i].data_ptr(), ra, buf.wgrad_geom[i]),
                                                    (buf.g[i].data_ptr() + self.plans[i].cout_pad * 2, rb,
                                                     buf.wgrad_geom_b[i]))):
                    job.x, job.g, job.dw = x.data_ptr(), g_ptr, out.data_ptr()
                    for name, _ in ConvGeom._fields_:
                        setattr(job.geom, name, getattr(geom, name))
            need = self.lib.raw("sl_conv1d_wgrad_multi_workspace_bytes")(table, len(table), self.dtype_code)
            ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=self.device)
            entry = buf.multi_tables[key] = (table, parts, scratch, ws)
        table, parts, _, ws = entry
        self._launch("wgrad:{}..{}".format(self.specs[layers[0]].name, self.specs[layers[-1]].name),
                     "sl_conv1d_wgrad_multi", table, len(table), self.dtype_code, ws.data_ptr(), ws.numel(), st)
        for i, (ra, rb) in zip(layers, parts):
            combine(self.plans[i], ra, rb)

    def _wgrad_multi_layers(self, first, grouped=None):
        """layers whose weight gradients go into ONE sl_conv1d_wgrad_multi launch (at the lowest of them): the runs of
        identical layers and the striding layer below them.  The same with and without a data-parallel exchange (the step
        is then bit-identical either way); bucket_plan() closes the striding layer's bucket together with the run's."""
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

    def _launch_wgrad_multi(self, buf, layers, st):
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
            need = self.lib.raw("sl_conv1d_wgrad_multi_workspace_bytes")(table, len(layers), self.dtype_code)
            if buf.wgrad_multi_ws is None or buf.wgrad_multi_ws.numel() < need:
                buf.wgrad_multi_ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=self.device)
        self._launch("wgrad:{}..{}".format(self.specs[layers[0]].name, self.specs[layers[-1]].name),
                     "sl_conv1d_wgrad_multi", table, len(layers), self.dtype_code, buf.wgrad_multi_ws.data_ptr(),
                     buf.wgrad_multi_ws.numel(), st)

    def _backward_eager(self, buf, main, side, on_bucket_ready):
        if self.planes > 1:
            return self._backward_x3(buf, main.cuda_stream, on_bucket_ready)
        first = self.frozen_layer_count
        grouped = self._grouped_wgrad_runs(first)
        dchain, dchain_skip = self._dgrad_chains(buf, first)
        multi = self._wgrad_multi_layers(first, grouped)
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
                if i == multi[0]:  # every gradient tensor the launch reads is complete at its lowest layer
                    self._launch_wgrad_multi(buf, multi, main.cuda_stream)
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
                    self._launch("cu_hint", "sl_set_available_cus", 256 - self.comm_cus)
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
            self._launch("cu_hint", "sl_set_available_cus", 0)

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
            if self.planes == 3:  # bf16x3: the [w_hi | w_hi | w_lo] operand rows are rewritten in the same pass
                self._launch(tag, "sl_split3_adam_pack_layers", self.params.data_ptr(), self.grads.data_ptr(),
                             self.adam_m.data_ptr(), self.adam_v.data_ptr(), table, len(chunk), self.adam_iterations,
                             self.lr, self.beta_1, self.beta_2, self.adam_epsilon, st)
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
                p = self.plans[i]
                wv, _ = self.layer_param_views(self.params, p)
                wd = self.w_dgrad[p.index]
                self._launch("pack3:" + p.spec.name, "sl_split3_pack_weights", wv.data_ptr(), self.w_fwd[p.index].data_ptr(),
                             wd.data_ptr() if wd is not None else None, p.spec.kernel_size, p.cin_pad, p.cout_pad, st)
            return
        layers = list(layers)
        for lo in range(0, len(layers), 16):
            chunk = layers[lo:lo + 16]
            self._launch("pack:{}..{}".format(self.all_plans[chunk[0]].spec.name, self.all_plans[chunk[-1]].spec.name),
                         "sl_pack_layers", self.params.data_ptr(), self._adam_table(chunk), len(chunk), self.dtype_code,
                         st)
