"""Hand-written gfx950 convolution engine (autograd wrappers over dvs_conv_* of libdvslam_hip.so)."""


def supported(x, weight, stride, padding, reflect_pad):
    """True when a hand-written kernel exists for this problem."""
    return False


def conv2d(x, weight, bias, stride, padding, reflect_pad):
    raise NotImplementedError
