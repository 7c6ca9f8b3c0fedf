"""Weight gradient of the four one-channel disparity heads at batch 12 (dvs_conv2d_head_bwd with dx = NULL): time per launch and the
gradient against torch.  DVS_HEAD_WGRAD_IN=0 times the output-indexed row form.  tools/head_bench.py [batch]"""
import ctypes as C, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import _lib, conv as DC
from deep_visual_slam_amd.conv import ACT, check, _desc

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
CL = torch.channels_last
torch.manual_seed(0)
for cin, H, W in ((16, 480, 640), (32, 240, 320), (64, 120, 160), (128, 60, 80)):
    x = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=CL)
    w = (torch.randn(1, cin, 3, 3, device=dev) * 0.1).contiguous(memory_format=CL)
    b = torch.randn(1, device=dev) * 0.1
    y = torch.sigmoid(F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), w, b)).contiguous(memory_format=CL)
    dy = torch.randn_like(y)
    d = _desc(B, cin, H, W, w.shape, 1, 1, True)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)

    def run():
        check(_lib.lib().dvs_conv2d_head_bwd_res(x.data_ptr(), w.data_ptr(), y.data_ptr(), dy.data_ptr(), None, dw.data_ptr(), db.data_ptr(),
                                                 C.byref(d), ACT["sigmoid"], None, _lib.stream()), "dvs_conv2d_head_bwd")
    run()
    torch.cuda.synchronize()
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = torch.sigmoid(F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), wr, br))
    gw, gb = torch.autograd.grad(yr, [wr, br], dy)
    ew = float((dw - gw).abs().max() / gw.abs().max())
    eb = float((db - gb).abs().max() / gb.abs().max())
    for _ in range(3):
        run()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 20 * 1e3
    mb = (x.numel() + 2 * y.numel()) * 4 / 1e6
    dx = torch.empty_like(x)

    def bwd():
        check(_lib.lib().dvs_conv2d_head_bwd_res(x.data_ptr(), w.data_ptr(), y.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                 C.byref(d), ACT["sigmoid"], None, _lib.stream()), "dvs_conv2d_head_bwd")
    xr = x.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(torch.sigmoid(F.conv2d(F.pad(xr, (1,) * 4, mode="reflect"), w, b)), [xr], dy)
    for _ in range(3):
        bwd()
    a.record()
    for _ in range(20):
        bwd()
    e.record()
    torch.cuda.synchronize()
    usb = a.elapsed_time(e) / 20 * 1e3
    ex = float((dx - gx).abs().max() / gx.abs().max())
    yo = torch.empty_like(y)

    def fwd():
        check(_lib.lib().dvs_conv2d_head_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), yo.data_ptr(), C.byref(d), ACT["sigmoid"], _lib.stream()),
              "dvs_conv2d_head_fwd")
    for _ in range(3):
        fwd()
    a.record()
    for _ in range(20):
        fwd()
    e.record()
    torch.cuda.synchronize()
    usf = a.elapsed_time(e) / 20 * 1e3
    ef = float((yo - y).abs().max())
    print("Cin %3d  %3dx%3d  wgrad %7.1f us  %6.1f MB  %5.2f TB/s   dw err %.1e  db err %.1e | fwd %7.1f us  %5.2f TB/s  err %.1e | dgrad + wgrad %7.1f us  dx err %.1e" % (
        cin, H, W, us, mb, mb / us, ew, eb, usf, (x.numel() + y.numel()) * 4 / 1e6 / usf, ef, usb, ex))
