"""Raw-wave input (reference net.py:310-312, `wave_conv`): the front layer in front of the stack -- sample windows gathered by
sl_wave_frames, a 1 x 1 GEMM into the pair-view input buffer, and the input gradient of the stack's first layer that its
weight gradient needs.  Methods of Engine."""
import ctypes

import numpy as np
import torch

from . import _lib
from .plan import same_padding


class FrontLayerMixin:
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
        if self.planes > 1:  # bf16x3: fp32 windows, then the planes [hi | lo | hi] of every window row
            self._launch("wave_frames", "sl_wave_frames", src.data_ptr(), buf.frames32.data_ptr(), buf.batch, t_audio, fs.cin,
                         fs.kernel_size, fs.stride, pad_l, t1, fp.cin_pad, buf.frames32.stride(0), _lib.SL_F32, self._stream())
            self._launch("split:wave_frames", "sl_split3", buf.frames32.data_ptr(), buf.frames.data_ptr(), None, buf.batch, t1,
                         fp.cin_pad, buf.frames32.stride(0), 0, buf.frames.stride(0), 0, self._stream())
        else:
            self._launch("wave_frames", "sl_wave_frames", src.data_ptr(), buf.frames.data_ptr(), buf.batch, t_audio, fs.cin,
                         fs.kernel_size, fs.stride, pad_l, t1, fp.cin_pad, buf.frames.stride(0), self.dtype_code, self._stream())
        buf.front_frames_dropped = False

    def _front_forward(self, buf, rate, seed0, st):
        """wave_conv: bias + activation epilogue of a 1 x 1 NT launch over the gathered windows, written into x0 where
        sl_pack_input would have put a spectrogram.  With dropout (a Dropout sits in front of wave_conv too, net.py:301-303)
        the samples are dropped first (seed offset 63: the stack's layers use 0 .. n) and the windows gathered again."""
        fp = self.front_plan
        if rate:
            if buf.front_src_dropped is None or buf.front_src_dropped.shape != buf.front_src.shape:
                buf.front_src_dropped = torch.empty_like(buf.front_src)
            self._launch("dropout:samples", "sl_dropout", buf.front_src.data_ptr(), buf.front_src_dropped.data_ptr(),
                         buf.front_src.numel(), _lib.SL_F32, rate, seed0 + 63, st)
            self._front_gather(buf, buf.front_src_dropped)
            buf.front_frames_dropped = True
        elif buf.front_frames_dropped:
            self._front_gather(buf, buf.front_src)
        _, bias = self.layer_param_views(self.params, fp)
        if self.planes > 1:
            return self._front_forward_x3(buf, bias, st)
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
        if self.planes > 1:
            return self._front_backward_x3(buf, x0, elu, elu_dropped, st)
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

    # ------------------------------------------------------------------ bf16x3 (round 5)
    def _front_forward_x3(self, buf, bias, st):
        """wave_conv on the planes: the unchanged NT kernel over the window rows [hi | lo | hi] against [w_hi | w_hi | w_lo];
        ReLU: bias, activation and the split into x0's planes in its epilogue (out_f32 = 2); ELU: fp32 staging + sl_split3"""
        fp, p0 = self.front_plan, self.plans[0]
        if fp.spec.activation == "relu" and self.x3_fused_epilogue:
            self._launch("fwd:" + fp.spec.name, "sl_conv1d_nt", buf.frames.data_ptr(), self.w_fwd[fp.index].data_ptr(),
                         bias.data_ptr(), None, buf.x0.data_ptr(), ctypes.byref(buf.front_geom), _lib.EPI_BIAS_RELU,
                         self.dtype_code, 2, 0, buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
            return
        self._launch("fwd:" + fp.spec.name, "sl_conv1d_nt", buf.frames.data_ptr(), self.w_fwd[fp.index].data_ptr(),
                     bias.data_ptr(), None, buf.stage32.data_ptr(), ctypes.byref(buf.front_stage_geom), _lib.EPI_BIAS,
                     self.dtype_code, 1, 0, buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)
        self._launch("split:" + fp.spec.name, "sl_split3", buf.stage32.data_ptr(), buf.x0.data_ptr(), None, buf.batch,
                     buf.front_geom.t_out, fp.cout_pad, buf.front_stage_geom.y_batch_stride, p0.pad_left, buf.x0.stride(0),
                     2 if fp.spec.activation == "elu" else 1, st)

    def _front_backward_x3(self, buf, x0, elu, elu_dropped, st):
        """x0's gradient: the pair-view NT launch over the planes of g[0] (flipped pair taps, _pack_pair_dgrad_x3) into the
        fp32 staging buffer, sl_split3 through the stored activation of wave_conv into gx0's planes FRAME row by frame row
        (a pair row of 2 cin_pad floats is two frame rows); then wave_conv's weight gradient as the two partial launches of
        this path + sl_split3_wgrad_combine, and its bias gradient"""
        fp, p0 = self.front_plan, self.plans[0]
        pl = self.planes
        dg = buf.front_dgrad_geom
        self._launch("dgrad:" + p0.spec.name, "sl_conv1d_nt", buf.g[0].data_ptr(), self.w_dgrad[0].data_ptr(), None, None,
                     buf.stage32.data_ptr(), ctypes.byref(dg), _lib.EPI_NONE, self.dtype_code, 1, 0, buf.nt_ws.data_ptr(),
                     buf.nt_ws.numel(), st)
        mode = 0 if elu_dropped else (4 if elu else 3)
        self._launch("split:dgrad:" + p0.spec.name, "sl_split3", buf.stage32.data_ptr(), buf.gx0.data_ptr(),
                     None if elu_dropped else x0.data_ptr(), buf.batch, 2 * dg.t_out, p0.cin_pad, dg.y_batch_stride,
                     2 * self.FRONT_DGRAD_ROW0, buf.gx0.stride(0), mode, st)
        if elu_dropped:
            self._dropout_x3("dropout_elu_bwd:" + p0.spec.name, buf.gx0, buf.gx0, x0, p0.cin_pad, 2, buf.dropout_seed0, st)
        elif buf.dropped:
            self._dropout_x3("dropout_scale:" + p0.spec.name, buf.gx0, buf.gx0, None, p0.cin_pad, 1, 0, st)
        dw, db = self.layer_param_views(self.grads, fp)
        ra = buf.wgrad_r
        rb = buf.wgrad_r[2 * fp.cin_pad * fp.cout_pad:]
        g_lo = buf.gx0.data_ptr() + p0.cin_pad * 2  # plane P1 of every row
        self._launch("wgrad:" + fp.spec.name, "sl_conv1d_wgrad", buf.frames.data_ptr(), buf.gx0.data_ptr(), ra.data_ptr(),
                     ctypes.byref(buf.front_wgrad_geom_a), self.dtype_code, 0, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
        self._launch("wgrad_lo:" + fp.spec.name, "sl_conv1d_wgrad", buf.frames.data_ptr(), g_lo, rb.data_ptr(),
                     ctypes.byref(buf.front_wgrad_geom_b), self.dtype_code, 0, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
        self._launch("combine:" + fp.spec.name, "sl_split3_wgrad_combine", ra.data_ptr(), rb.data_ptr(), dw.data_ptr(), 1,
                     fp.cin_pad, fp.cout_pad, 1, 0, 2 * fp.cin_pad, fp.cin_pad, 0, st)
        if self._x3_bias_ws is None:
            self._x3_bias_ws = torch.empty((self.lib.raw("sl_split3_bias_grad_workspace_bytes")(
                max(q.cout_pad for q in self.plans)),), dtype=torch.uint8, device=self.device)
        self._launch("bgrad:" + fp.spec.name, "sl_split3_bias_grad", buf.gx0.data_ptr(), db.data_ptr(), buf.batch,
                     buf.front_geom.t_out, fp.cout_pad, p0.pad_left, buf.gx0.stride(0), self._x3_bias_ws.data_ptr(),
                     self._x3_bias_ws.numel(), st)
