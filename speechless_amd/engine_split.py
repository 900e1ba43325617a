"""Split top (Engine.split_top): the top three layers and the CTC of a training step in two parts of the batch, the CTC of one
part on a side stream under the top layers of the other.  Methods of Engine."""
import ctypes

import torch

from . import _lib
from ._lib import ConvGeom
from .plan import HALO


class SplitTopMixin:
    def split_top_plan(self, buf):
        """How the training step on `buf` runs its top three layers and the CTC in two parts of the batch: the number of
        utterances in the first part, or 0 (whole-batch step)."""
        state = (self.split_top, bool(self.dropout_rate), self.frozen_layer_count, self.fuse_output_softmax,
                 self.fuse_output_backward, tuple(sorted(self.nt_cfg)), buf.bwd_ready, buf.bwd1x1_ws is not None,
                 buf.labels is not None)
        if buf._split_ok[0] != state:
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
                need = 0
                for hint in self.cu_hints():
                    self.lib.call("sl_set_available_cus", hint)
                    need = max(need, self.lib.raw("sl_conv1d_nt_workspace_bytes")(ctypes.byref(g), self.dtype_code, 0))
                self.lib.call("sl_set_available_cus", self._cu_hint_active)
                if buf.nt_ws is None or buf.nt_ws.numel() < need:
                    buf.nt_ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
                    buf.launch_lists = {}
        return g

    @staticmethod
    def _utt_ptr(t, first):
        """address of utterance `first` of a tensor whose first dimension is the batch"""
        return t.data_ptr() + first * t.stride(0) * t.element_size()

    def _presize_parts(self, buf, kinds, a):
        """Creates the part geometries of the top layers for both parts BEFORE any launch of a (possibly recorded) sequence
        reads buf.nt_ws: _part_geom may have to re-allocate that workspace, and a launch already appended to the list being
        recorded would keep the freed pointer.  Sized under every CU setting in use."""
        n = len(self.plans)
        for count in (a, buf.batch - a):
            for kind in kinds:
                for i in (n - 3, n - 2):
                    self._part_geom(buf, kind, i, count)

    def _forward_top_split(self, buf, x, st, grad_scale, a):
        """big_conv_1, big_conv_2, output_conv + softmax and the CTC, part by part: utterances [0, a), then [a, B)"""
        n = len(self.plans)
        self._presize_parts(buf, ("fwd", "dgrad") if buf.bwd_ready else ("fwd",), a)
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
        # ONE contiguous copy per step, made for the first part and kept until the next step's first part replaces it (by
        # then both parts' CTC launches have been waited for by backward): a copy per part would free part 0's while its
        # lattice may still be reading it
        if h == 0 or buf._split_labels_keepalive is None:
            buf._split_labels_keepalive = buf.labels if buf.labels.is_contiguous() else buf.labels.contiguous()
        labels = buf._split_labels_keepalive
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

    def _wait_ctc_half(self, buf, h, main):
        main.wait_event(buf.ctc_done[h])

    def _backward_top_split(self, buf, main, a):
        """the input gradients of the top three layers (and the output layer's weight gradient) part by part, each part
        behind its own CTC: the second part's lattice runs under the first part's launches here"""
        n = len(self.plans)
        st = main.cuda_stream
        last = n - 1
        self._presize_parts(buf, ("dgrad",), a)
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
