"""The bf16x3 path of the engine (Engine(dtype="bf16x3")): every fp32 value as hi + lo bf16 planes, a product as three bf16 MFMA
terms through the unchanged NT / TN kernels (csrc/split3.hip; DESIGN.md section 3.1, HISTORY.md section 3.6).  Methods of Engine."""
import ctypes

import torch

from . import _lib
from ._lib import ConvGeom
from .plan import HALO


class X3Mixin:
    def _x3(self, name):
        """the entry point of a plane helper for this engine's plane format: sl_split3* (bf16 pairs) or its fp16 twin"""
        return name.replace("sl_split3", "sl_splitf16", 1) if self.x3_f16 else name

    def _pack_weights_x3(self, tag, w_master, w_fwd, w_dgrad, k, cin, cout, st):
        """both operand copies [w_hi | w_hi | w_lo] of one layer from its fp32 master (f16x3: of w_scale * w)"""
        args = (w_master.data_ptr(), w_fwd.data_ptr(), w_dgrad.data_ptr() if w_dgrad is not None else None, k, cin, cout)
        if self.x3_f16:
            self._launch(tag, "sl_splitf16_pack_weights", *args, self.w_scale, st)
        else:
            self._launch(tag, "sl_split3_pack_weights", *args, st)

    def _repack_weights_x3(self):
        """bf16x3 operand copies: rows [w_hi | w_hi | w_lo] in both operand layouts, w_hi = bf16(w), w_lo = bf16(w - w_hi),
        one launch per layer (sl_split3_pack_weights; the five-launch sequence it replaces -- split, two packs, two
        assembles -- was 55 launches and 0.45 ms of a 7.1 ms optimisation step)."""
        st = self._stream()
        for p in self.all_plans:  # (the raw-wave front layer included; a striding layer with a dgrad operand: its pair view)
            wv, _ = self.layer_param_views(self.params, p)
            wd = self.w_dgrad[p.index]
            if p.index == 0 and p is not self.front_plan:
                wd = None  # (only there under a front layer, and then in the pair view: _pack_pair_dgrad_x3 below)
            k, cin = self._pack_dims(p)
            self._pack_weights_x3("pack3:" + p.spec.name, wv, self.w_fwd[p.index], wd, k, cin, p.cout_pad, st)
        if self.w_dgrad[0] is not None:
            self._pack_pair_dgrad_x3(st)
        self._packed_dirty = False

    def _pack_pair_dgrad_x3(self, st):
        """the input-gradient operand of the striding layer under a raw-wave front layer: the PAIR VIEW of its taps (24 flipped
        pair taps x 2 cin_pad pair channels, rows [w_hi | w_hi | w_lo] over the output channels) -- sl_split3_pack_weights on
        the pair view of the masters; the forward operand it writes beside it goes to a scratch buffer (the layer's real
        forward operand is packed with its own 48 taps: a frame's planes are contiguous in a row of the input buffer)"""
        p0 = self.plans[0]
        if getattr(self, "_w_fwd0_pair_scratch", None) is None:
            self._w_fwd0_pair_scratch = torch.empty_like(self.w_fwd[0])
        wv, _ = self.layer_param_views(self.params, p0)
        self._pack_weights_x3("pack3_pair:" + p0.spec.name, wv, self._w_fwd0_pair_scratch, self.w_dgrad[0], p0.taps_view,
                              p0.cin_view, p0.cout_pad, st)

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
        self._launch(tag, self._x3("sl_split3_dropout"), src.data_ptr(), dst.data_ptr(), y.data_ptr() if y is not None else None,
                     src.numel() // (self.planes * channels), channels, mode, self.dropout_rate, seed, st)

    def _forward_x3(self, buf, st, rate=None):
        """bf16x3: every layer = the unchanged NT kernel over the three planes.  ReLU layers: bias, ReLU and the split into
        planes in the kernel's own epilogue (out_f32 = 2); ELU layers: fp32 into the staging buffer + sl_split3.  The last
        layer's fp32 logits go to the softmax as on the other paths.  Dropout (training, net.py:301-303): sl_split3_dropout
        on the packed input (into a second buffer) and in place on every activation that feeds a layer with a Dropout in
        front of it -- the same (seed, element) keep decisions as sl_dropout draws on the single-plane paths."""
        n = len(self.plans)
        x = buf.x0
        seed0 = 0
        if rate:
            self._dropout_steps += 1
            seed0 = buf.dropout_seed0 = (self.dropout_seed * 1000003 + self._dropout_steps) * 64
        if self.front_plan is not None:
            self._front_forward(buf, rate, seed0, st)
        if rate:
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
                self._launch("split:" + p.spec.name, self._x3("sl_split3"), buf.stage32.data_ptr(), y.data_ptr(), None, buf.batch,
                             buf.t_out, p.cout_pad, buf.tt_pad * p.cout_pad, HALO, buf.rows * p.cout_pad * self.planes,
                             2 if p.spec.activation == "elu" else 1, st)
                drop(p)
                x = y
        self._launch("softmax", "sl_softmax_logq", buf.logits.data_ptr(), buf.probs.data_ptr(), buf.logq.data_ptr(), buf.batch,
                     buf.t_out, self.grapheme_set_size, self.plans[-1].cout_pad, buf.tt_pad * self.plans[-1].cout_pad,
                     self.ctc_epsilon, st)
        return buf.probs

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
        for launch_layers in self._x3_multi_runs(first):
            for i in launch_layers:
                multi[i] = launch_layers

        def combine(p, ra, rb):
            dw, _ = self.layer_param_views(self.grads, p)
            frames = 2 if p.spec.stride == 2 else 1
            fstride = pl * p.cin_pad if frames == 2 else 0
            window = frames == 2 and buf.x3_window  # RB's x operand was the [hi0 | hi1] window of the pair row
            args = (ra.data_ptr(), rb.data_ptr(), dw.data_ptr(), p.spec.kernel_size, p.cin_pad, p.cout_pad, frames, fstride,
                    buf.wgrad_geom[p.index].cin, buf.wgrad_geom_b[p.index].cin, p.cin_pad if window else fstride)
            if self.x3_f16:  # the partial sums are g_scale * dW (the gradient planes are stored scaled)
                self._launch("combine:" + p.spec.name, "sl_split3_wgrad_combine_scaled", *args, 1.0 / self.g_scale, st)
            else:
                self._launch("combine:" + p.spec.name, "sl_split3_wgrad_combine", *args, st)

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
                x_b = x.data_ptr() + (2 * p.cin_pad * 2 if (i == 0 and p.spec.stride == 2 and buf.x3_window) else 0)
                self._launch("wgrad_lo:" + p.spec.name, "sl_conv1d_wgrad", x_b, g_lo, rb.data_ptr(),
                             ctypes.byref(wb), self.dtype_code, cfg, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
                combine(p, ra, rb)
            if i not in ones_db:
                if self._x3_bias_ws is None:
                    self._x3_bias_ws = torch.empty((self.lib.raw("sl_split3_bias_grad_workspace_bytes")(
                        max(q.cout_pad for q in self.plans)),), dtype=torch.uint8, device=self.device)
                args = (buf.g[i].data_ptr(), db.data_ptr(), buf.batch, buf.t_out, p.cout_pad, HALO, buf.rows * p.cout_pad * pl)
                if self.x3_f16:
                    self._launch("bgrad:" + p.spec.name, "sl_splitf16_bias_grad", *args, 1.0 / self.g_scale,
                                 self._x3_bias_ws.data_ptr(), self._x3_bias_ws.numel(), st)
                else:
                    self._launch("bgrad:" + p.spec.name, "sl_split3_bias_grad", *args, self._x3_bias_ws.data_ptr(),
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
                self._launch("split:dgrad:" + p.spec.name, self._x3("sl_split3"), buf.stage32.data_ptr(), buf.g[i - 1].data_ptr(),
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
                self._launch("split:dgrad:" + p.spec.name, self._x3("sl_split3"), buf.stage32.data_ptr(), buf.g[i - 1].data_ptr(),
                             buf.y[i - 1].data_ptr(), buf.batch, buf.t_out, p.cin_pad, buf.tt_pad * p.cin_pad, HALO,
                             buf.rows * p.cin_pad * pl, 4 if self.specs[i - 1].activation == "elu" else 3, st)
            if i > first and dropped_in:
                # the ReLU mask (stored activation > 0) already applied the keep mask: the stored activation is the
                # post-dropout one; what is left of d dropout / dx is the factor 1 / (1 - rate)
                self._dropout_x3("dropout_scale:" + p.spec.name, buf.g[i - 1], buf.g[i - 1], None, p.cin_pad, 1, 0, st)
        if on_bucket_ready is None and ones_in:
            self._bias_grads_from_wgrad(ones_in, bool(ones_db), main)
        if self.front_plan is not None and first == 0 and not self.front_frozen:
            self._front_backward(buf, st)
            if on_bucket_ready is not None:  # the front layer's parameters: the last bucket of bucket_plan()
                b = len(self.bucket_plan()) - 1
                on_bucket_ready(b)
                if self._rec is not None:
                    self._rec.append((2, b))

    def _wgrad_multi_layers_x3(self, first):
        """bf16x3: no launch writes the weight gradients of the striding layer AND of a run (bucket_plan() merges nothing)"""
        return []

    def _x3_multi_runs(self, first):
        """bf16x3: the runs of identical layers whose 2 partial weight gradients per layer are jobs of ONE
        sl_conv1d_wgrad_multi launch at the lowest layer of the run (16 jobs at most): x operands of whole 256-wide tiles
        against 256-wide output tiles.  Shapes only (the same for every buffer set)."""
        out = []
        if not self.use_wgrad_multi:
            return out
        for (s0, e0) in self.runs:
            lo = max(s0, first)
            layers = list(range(lo, e0 + 1))

            def fits_run(i):
                p = self.plans[i]  # A: the [hi | lo] prefix (2 cin_pad) against g_hi, B: x_hi (cin_pad) against g_lo
                return p.cin_pad % 256 == 0 and p.cout_pad % 256 == 0 and p.spec.stride == 1 and \
                    ("wgrad", self.specs[i].name) not in self.nt_cfg
            if len(layers) >= 2 and 2 * len(layers) <= 16 and all(fits_run(i) for i in layers):
                # (round 5: striding_conv's two partials as jobs of the same launch -- 144 more tiles over the 768-wide pair
                # rows -- were measured SLOWER: 6.328 against 6.215 ms per config-3 step in a same-box A/B,
                # profiles/r05_x3_multi_striding_ab.txt; they stay two 128 x 128-tile launches)
                out.append(layers)
        return out

    def _launch_wgrad_multi_x3(self, buf, layers, st, combine):
        """bf16x3: the partial weight gradients RA (x planes [hi | lo] against g_hi) and RB (x plane hi against g_lo) of
        every layer of a run as jobs of one sl_conv1d_wgrad_multi launch, then sl_split3_wgrad_combine per layer"""
        key = ("x3", buf.dropped) + tuple(layers)
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
                x = (buf.x0_dropped if buf.dropped else buf.x0) if i == 0 else buf.y[i - 1]
                for job, (g_ptr, out, geom) in zip((table[2 * n], table[2 * n + 1]),
                                                   ((buf.g[i].data_ptr(), ra, buf.wgrad_geom[i]),
                                                    (buf.g[i].data_ptr() + self.plans[i].cout_pad * 2, rb,
                                                     buf.wgrad_geom_b[i]))):
                    job.x, job.g, job.dw = x.data_ptr(), g_ptr, out.data_ptr()
                    for name, _ in ConvGeom._fields_:
                        setattr(job.geom, name, getattr(geom, name))
            need = self._wgrad_multi_workspace_need(table, len(table))
            ws = torch.empty((max(need, 16),), dtype=torch.uint8, device=self.device)
            entry = buf.multi_tables[key] = (table, parts, scratch, ws)
        table, parts, _, ws = entry
        self._launch("wgrad:{}..{}".format(self.specs[layers[0]].name, self.specs[layers[-1]].name),
                     "sl_conv1d_wgrad_multi", table, len(table), self.dtype_code, ws.data_ptr(), ws.numel(), st)
        for i, (ra, rb) in zip(layers, parts):
            combine(self.plans[i], ra, rb)
