"""ORACLE (test infrastructure) -- ``paddle.nn.layer.conv._ConvNd`` (the base the reference's Conv2DValid extends,
squeezeformer/conv2d.py:10).  Weight layout [out, in/groups, *kernel]."""
import numpy as _np

from paddle.nn import Layer  # paddle.nn is partially initialised here; Layer is already defined


def _ntuple(v, n):
    if isinstance(v, (int, _np.integer)):
        return [int(v)] * n
    v = [int(a) for a in v]
    return v * n if len(v) == 1 else v


class _ConvNd(Layer):
    def __init__(self, in_channels, out_channels, kernel_size, transposed, dims, stride=1, padding=0,
                 padding_mode="zeros", output_padding=0, dilation=1, groups=1, weight_attr=None, bias_attr=None,
                 data_format="NCHW"):
        super().__init__()
        if transposed:
            raise NotImplementedError("paddle shim: transposed conv")
        if padding_mode != "zeros":
            raise NotImplementedError("paddle shim: conv padding_mode " + padding_mode)
        if in_channels % groups != 0 or out_channels % groups != 0:
            raise ValueError("channels must be divisible by groups")
        self._in_channels = in_channels
        self._out_channels = out_channels
        self._groups = groups
        self._data_format = data_format
        self._kernel_size = _ntuple(kernel_size, dims)
        self._stride = _ntuple(stride, dims)
        self._dilation = _ntuple(dilation, dims)
        self._padding = padding
        self._param_attr = weight_attr
        self._bias_attr = bias_attr
        shape = [out_channels, in_channels // groups] + self._kernel_size
        self.weight = self.create_parameter(shape, attr=weight_attr)
        b = self.create_parameter([out_channels], attr=bias_attr, is_bias=True)
        if b is None:
            self.bias = None
        else:
            self.bias = b
