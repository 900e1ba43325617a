"""HBM buffer sets of the engine: every activation / gradient tensor, workspace and geometry descriptor of one (batch, padded
frames) geometry (layout: include/speechless_hip.h and DESIGN.md section 2)."""
import ctypes

import torch

from ._lib import ConvGeom, lib
from .plan import HALO, TIME_TILE, _round_up, same_padding


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
            g.acc_scale = 1.0 / eng.w_scale  # (f16x3: the operand copies hold w_scale * w; 1 on every other path)
            self.fwd_geom.append(g)
        # bf16x3: fp32 staging buffer of a layer's pre-activations / input gradients (sl_conv1d_nt out_f32 -> sl_split3)
        stage = batch * self.tt_pad * max(p.cout_pad for p in eng.plans)
        if eng.front_plan is not None:  # the front layer stages its ELU pre-activations (one row per INPUT frame) and the
            # pair rows of x0's gradient here (front_stage_geom, front_dgrad_geom)
            stage = max(stage, batch * 2 * self.tt_pad * eng.front_plan.cout_pad,
                        batch * (self.tt_pad + eng.FRONT_DGRAD_EXTRA_ROWS) * p0.cin_view)
        self.stage32 = torch.empty((stage,), dtype=torch.float32, device=dev) if pl > 1 else None
        self.plane_geoms = {}  # bf16x3: (kind, layer) -> geometry whose output side describes a plane tensor
        # front layer (raw-wave input): gathered sample windows [B][2 tt_pad][K_pad], the gradient w.r.t. the stack's input in
        # the pair-view layout of x0, and the three geometries of the launches around them (Engine._front_*)
        self.frames = self.gx0 = self.front_geom = self.front_dgrad_geom = self.front_wgrad_geom_a = None
        self.front_src = self.front_src_dropped = None  # the samples the windows were gathered from (and their dropped copy)
        self.front_geometry = None                       # (samples, output frames, left padding) of the current batch
        self.front_frames_dropped = False                # frames hold the windows of DROPPED samples (last training step)
        self.dropout_seed0 = 0                           # seed base of the last training forward with dropout
        self._split_ok = (None, False)                   # cached Engine._split_top_ok for the engine state it was computed under
        self._split_labels_keepalive = self._g0_keepalive = None
        self.frames32 = self.front_stage_geom = self.front_wgrad_geom_b = None  # bf16x3 only, see below
        if eng.front_plan is not None:
            fp = eng.front_plan
            self.frames = torch.zeros((batch, 2 * tt_pad, fp.cin_pad * pl), dtype=dt, device=dev)
            g = ConvGeom()
            g.batch, g.t_out, g.taps, g.cin, g.cout = batch, 2 * tt_pad, 1, fp.cin_pad * pl, fp.cout_pad
            g.x_row0, g.x_row_stride, g.x_batch_stride = 0, fp.cin_pad * pl, 2 * tt_pad * fp.cin_pad * pl
            g.y_row0, g.y_row_stride, g.y_batch_stride = p0.pad_left, p0.cin_pad * pl, self.rows0 * p0.cin_pad * pl
            self.front_geom = g  # forward (x = frames, y = x0) and weight gradient (x = frames, "y" = gx0): t_out = input frames
            if pl > 1:
                # bf16x3: the windows are gathered in fp32 and split into planes (sl_wave_frames + sl_split3); an ELU front
                # layer goes through the fp32 staging buffer (front_stage_geom) like every ELU layer of this path; the weight
                # gradient is two launches -- the [hi | lo] prefix of the windows against gx0's hi plane (front_geom with
                # cin = 2 K_pad) and their hi plane against gx0's lo plane (front_wgrad_geom_b) -- plus sl_split3_wgrad_combine
                self.frames32 = torch.zeros((batch, 2 * tt_pad, fp.cin_pad), dtype=torch.float32, device=dev)
                sg = ConvGeom()
                for name, _ in ConvGeom._fields_:
                    setattr(sg, name, getattr(g, name))
                sg.y_row0, sg.y_row_stride, sg.y_batch_stride = 0, fp.cout_pad, 2 * tt_pad * fp.cout_pad
                self.front_stage_geom = sg
                wa, wb = ConvGeom(), ConvGeom()
                for w in (wa, wb):
                    for name, _ in ConvGeom._fields_:
                        setattr(w, name, getattr(g, name))
                wa.cin, wb.cin = 2 * fp.cin_pad, fp.cin_pad
                self.front_wgrad_geom_a, self.front_wgrad_geom_b = wa, wb
        self.half_geoms = {}   # split top (Engine.split_top): (kind, layer) -> the layer's geometry for half the batch
        self.ctc_done = [None, None]  # split top: events behind the CTC launches of the two half-batches
        self.ctc_half_bytes = 0
        self.split_pending = 0        # utterances in the first part if the last forward ran the CTC in two parts (else 0)
        self.wgrad_r = None  # bf16x3: the two partial weight gradients (RA | RB) in front of sl_split3_wgrad_combine
        self.wgrad_geom_b = [None] * n
        self.x3_window = False  # bf16x3, striding layer: RB's x operand is the [hi0 | hi1] window of the pair row (ensure_backward)
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
            if self.frames is not None:
                # the front layer's weight gradient contracts whole 64-row chunks of `frames` against gx0, and with ELU the
                # input-gradient launch leaves non-zero rows of gx0 just past the valid frames (elu'(0) = 1 keeps what
                # FRONT_DGRAD_EXTRA_ROWS computes): the sample windows of a longer previous batch must not sit under them
                self.frames[:, t_in: self._clean_in].zero_()
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
        for g in (self.front_geom, self.front_stage_geom, getattr(self, "front_wgrad_geom_a", None), self.front_wgrad_geom_b):
            if g is not None:
                g.t_out = t_in
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
        lib().call("sl_set_available_cus", eng._cu_hint_active)
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
                # layer has the planes of two frames in a row -- [hi0 | lo0 | hi0 | hi1 | lo1 | hi1] -- so there A takes the row
                # up to lo1 (five planes where that is a whole number of 128-wide tiles, else all six) and B the window
                # [hi0 | hi1] in its middle (Engine.x3_b_window: x pointer + 2 cin_pad, frame stride cin_pad inside RB): two
                # planes instead of six (round 5; the whole row for both was 0.18 + 0.18 ms at config 3)
                wb = ConvGeom()
                for name, _ in ConvGeom._fields_:
                    setattr(wb, name, getattr(wg, name))
                if p.index == 0 and p.spec.stride == 2:
                    # (the window decision is frozen HERE, with the geometries: the launch-time pointer offset and the
                    # combine read it back from the buffer set, not from the engine attribute -- ADVICE r5)
                    self.x3_window = bool(eng.x3_b_window)
                    wg.cin = 5 * p.cin_pad if (self.x3_window and (5 * p.cin_pad) % 128 == 0) else p.cin_view * pl
                    wb.cin = 2 * p.cin_pad if self.x3_window else p.cin_view * pl
                else:  # (a stride-1 first layer included: its rows are one frame's [hi | lo | hi] like everyone's)
                    wg.cin, wb.cin = 2 * p.cin_pad, p.cin_pad
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
                dg.acc_scale = 1.0 / eng.w_scale  # (f16x3: g_scale * g in, g_scale * g out; the weights' scale is divided out)
                self.dgrad_geom[p.index] = dg
        if eng.front_plan is not None and first == 0 and not eng.front_frozen:
            # dL/d(x0) in x0's own pair-view layout: pair row r = sum over the 24 pair taps j of g0[r - j] . Wpair[j]^T, as an
            # NT launch over g0 with flipped taps.  Frames start at pair row 11, so rows from 7 on are computed: the launch
            # then reads g0 from its first halo row (16 - 23 + 7 = 0) and never in front of the buffer.
            p0 = eng.plans[0]
            self.gx0 = torch.zeros_like(self.x0)
            dg = ConvGeom()
            dg.batch, dg.t_out, dg.taps = self.batch, (self.t_out or 0) + eng.FRONT_DGRAD_EXTRA_ROWS, p0.taps_view
            pl = eng.planes
            dg.cin, dg.cout = p0.cout_pad * pl, p0.cin_view
            dg.x_row0, dg.x_row_stride, dg.x_batch_stride = 0, p0.cout_pad * pl, self.rows * p0.cout_pad * pl
            dg.y_row0, dg.y_row_stride, dg.y_batch_stride = eng.FRONT_DGRAD_ROW0, p0.cin_view, self.rows0 * p0.cin_pad
            if pl > 1:
                # bf16x3: fp32 pair rows into the staging buffer (row 0 = pair row FRONT_DGRAD_ROW0; a fixed batch stride for
                # the longest batch of this buffer set), from where sl_split3 applies the activation mask and writes the planes
                # of the FRAME rows (a pair row of 2 cin_pad floats = two frame rows of cin_pad)
                dg.y_row0, dg.y_batch_stride = 0, (self.tt_pad + eng.FRONT_DGRAD_EXTRA_ROWS) * p0.cin_view
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
        L.call("sl_set_available_cus", eng._cu_hint_active)
        self.size_nt_workspace(eng, self.dgrad_geom, "dgrad")
        if eng.front_plan is not None:
            for g in (self.front_geom, self.front_stage_geom, self.front_dgrad_geom):
                if g is not None:
                    need = L.raw("sl_conv1d_nt_workspace_bytes")(ctypes.byref(g), eng.dtype_code, 0)
                    if self.nt_ws.numel() < need:
                        self.nt_ws = torch.empty((need,), dtype=torch.uint8, device=eng.device)
                        self.launch_lists = {}
            for g in ((self.front_wgrad_geom_a, self.front_wgrad_geom_b) if eng.planes > 1 else (self.front_geom,)):
                ws_bytes = max(ws_bytes, L.raw("sl_conv1d_wgrad_workspace_bytes")(ctypes.byref(g), eng.dtype_code, 0))
            bias_ws = max(bias_ws, L.raw("sl_bias_grad_workspace_bytes")(ctypes.byref(self.front_geom)))
        if eng.planes > 1:
            need = max(p.taps_view * (self.wgrad_geom[p.index].cin + self.wgrad_geom_b[p.index].cin) * p.cout_pad
                       for p in eng.plans[first:])
            if eng.front_plan is not None:
                need = max(need, 3 * eng.front_plan.cin_pad * eng.front_plan.cout_pad)
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
