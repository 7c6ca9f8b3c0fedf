"""Ordered (slab / workspace) weight gradients -- the deterministic backward of the two big weight-gradient kernel families:
the Winograd kernel (dvs_conv3x3_wino_wgrad_ws / _gen_ws) and the split-K implicit GEMM (dvs_conv2d_wgrad_ws).  Each must (1) agree
with the atomic form and with torch's fp64 gradient to the tolerance of the atomic kernels' own tests (1e-4 of the tensor max)
and (2) repeat BIT FOR BIT, which the atomic form does not promise.  Reference arithmetic: the weight gradient of nn.Conv2d as
used by torchvision's BasicBlock (model/resnet_encoder.py:83-111) and Conv3x3 (model/layers.py:26-41)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _ref(x, dy, wshape, stride, pad, reflect=False):
    w = torch.zeros(wshape, device=x.device, dtype=torch.float64, requires_grad=True)
    xx = x.double()
    if reflect:
        xx, pad = F.pad(xx, (1, 1, 1, 1), mode="reflect"), 0
    F.conv2d(xx, w, None, stride, pad).backward(dy.double())
    return w.grad


@pytest.fixture
def ordered():
    from deep_visual_slam_amd import conv as DC
    old = DC._WGRAD_ORDERED
    DC._WGRAD_ORDERED = True
    yield DC
    DC._WGRAD_ORDERED = old


@pytest.mark.parametrize("B,ci,co,h,w", [(2, 64, 64, 24, 32), (3, 128, 96, 13, 27), (12, 64, 64, 120, 160), (2, 512, 512, 15, 20)])
def test_winograd_weight_gradient_ordered(gpu_device, ordered, B, ci, co, h, w):
    DC = ordered
    torch.manual_seed(B * h + co)
    x = torch.randn(B, ci, h, w, device=gpu_device).contiguous(memory_format=CL)
    dy = torch.randn(B, co, h, w, device=gpu_device).contiguous(memory_format=CL)
    ref = _ref(x, dy, (co, ci, 3, 3), 1, 1)
    a = DC.conv3x3_wino_wgrad(x, dy, (co, ci, 3, 3))
    b = DC.conv3x3_wino_wgrad(x, dy, (co, ci, 3, 3))
    assert torch.equal(a, b)                                          # fixed summation order
    assert float((a.double() - ref).abs().max() / ref.abs().max()) < 1e-4
    sink = torch.full((co, ci, 3, 3), 0.5, device=gpu_device).contiguous(memory_format=CL)
    assert DC.conv3x3_wino_wgrad(x, dy, (co, ci, 3, 3), dw_out=sink) is None
    assert float(((sink - 0.5).double() - ref).abs().max() / ref.abs().max()) < 1e-4      # adds into the sink
    DC._WGRAD_ORDERED = False
    c = DC.conv3x3_wino_wgrad(x, dy, (co, ci, 3, 3))                 # the atomic form
    assert float((a - c).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("up", [False, True])
def test_winograd_decoder_weight_gradient_ordered(gpu_device, ordered, up):
    DC = ordered
    torch.manual_seed(5)
    B, c1, c2, co, hs, ws = 2, 64, (32 if up else 0), 64, 12, 16
    H, W = (2 * hs, 2 * ws) if up else (hs, ws)
    x = torch.randn(B, c1, hs, ws, device=gpu_device).contiguous(memory_format=CL)
    x2 = torch.randn(B, c2, H, W, device=gpu_device).contiguous(memory_format=CL) if up else None
    dz = torch.randn(B, co, H, W, device=gpu_device).contiguous(memory_format=CL)
    full = torch.cat([F.interpolate(x, scale_factor=2, mode="nearest"), x2], 1) if up else x
    ref = _ref(full, dz, (co, c1 + c2, 3, 3), 1, 1, reflect=True)
    a, _ = DC.conv3x3_wino_wgrad_gen(x, x2, dz, (co, c1 + c2, 3, 3))
    b, _ = DC.conv3x3_wino_wgrad_gen(x, x2, dz, (co, c1 + c2, 3, 3))
    assert torch.equal(a, b)
    assert float((a.double() - ref).abs().max() / ref.abs().max()) < 1e-4


@pytest.mark.parametrize("B,ci,co,h,w,k,stride,pad", [(2, 64, 128, 24, 32, 3, 2, 1), (2, 64, 128, 24, 32, 1, 2, 0), (12, 64, 128, 120, 160, 3, 2, 1),
                                                      (3, 32, 48, 17, 23, 3, 1, 1), (2, 256, 512, 30, 40, 1, 2, 0)])
def test_implicit_gemm_weight_gradient_ordered(gpu_device, ordered, B, ci, co, h, w, k, stride, pad):
    DC = ordered
    torch.manual_seed(ci + k)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = torch.randn(B, ci, h, w, device=gpu_device).contiguous(memory_format=CL)
    dy = torch.randn(B, co, ho, wo, device=gpu_device).contiguous(memory_format=CL)
    ref = _ref(x, dy, (co, ci, k, k), stride, pad)
    a, _ = DC.conv2d_wgrad(x, dy, (co, ci, k, k), stride, pad, False, False)
    b, _ = DC.conv2d_wgrad(x, dy, (co, ci, k, k), stride, pad, False, False)
    assert torch.equal(a, b)
    assert float((a.double() - ref).abs().max() / ref.abs().max()) < 1e-4
    sink = torch.full((co, ci, k, k), -0.25, device=gpu_device).contiguous(memory_format=CL)
    DC.conv2d_wgrad(x, dy, (co, ci, k, k), stride, pad, False, False, dw_out=sink)
    assert float(((sink + 0.25).double() - ref).abs().max() / ref.abs().max()) < 1e-4
    DC._WGRAD_ORDERED = False
    c, _ = DC.conv2d_wgrad(x, dy, (co, ci, k, k), stride, pad, False, False)
    assert float((a - c).abs().max() / ref.abs().max()) < 2e-6


def test_workspace_query_and_short_workspace(gpu_device):
    import ctypes as C
    from deep_visual_slam_amd import _lib, conv as DC
    l = _lib.lib()
    assert l.dvs_conv3x3_wino_wgrad_workspace(12, 120, 160, 64, 64, 0) == 4 * 64 * 9 * 1024 * 4      # 4 blocks x 64 splits
    assert l.dvs_conv3x3_wino_wgrad_workspace(12, 120, 160, 48, 64, 0) == 0                          # not a multiple of 32
    x = torch.randn(2, 64, 24, 32, device=gpu_device).contiguous(memory_format=CL)
    dy = torch.randn(2, 64, 24, 32, device=gpu_device).contiguous(memory_format=CL)
    dw = torch.zeros(64, 64, 3, 3, device=gpu_device).contiguous(memory_format=CL)
    ws = torch.empty(16, device=gpu_device)
    rc = l.dvs_conv3x3_wino_wgrad_ws(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 2, 24, 32, 64, 64, 0, ws.data_ptr(), 64, _lib.stream())
    assert rc < 0 and b"workspace" in l.dvs_last_error()
    d = DC._desc(2, 64, 24, 32, (128, 64, 3, 3), 2, 1, False)
    f = DC._fusion(x, None, None, None, False, False)
    need = l.dvs_conv2d_wgrad_workspace(C.byref(d), C.byref(f), 0, 0)
    assert need > 0 and need % (128 * 128 * 4) == 0
