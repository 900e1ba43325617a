"""Topology and per-layer plans of the Wav2Letter stack (reference net.py:291-341): layer specs, TF 'SAME' padding, padded
channel counts and the layer's place in the flat parameter buffers.  Shared by the engine and its buffer sets."""

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
