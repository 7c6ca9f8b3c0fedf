"""Host-side logic that needs no GPU: the lazy `outputs` dict, FusedAdam as a torch.optim.Optimizer (scheduler
acceptance, torch.optim.Adam-compatible checkpoints), the folded-BatchNorm generation counter, and the loud failure of
every primitive off the GPU / outside the kernels' shape coverage (no library fallback in the product)."""
import pickle

import pytest
import torch


# ------------------------------------------------------------------------------------------ LazyOutputs
def _lazy():
    from deep_visual_slam_amd.learner_new import LazyOutputs
    out = LazyOutputs({("disp", 0): 1, ("cam_T_cam", 0, -1): 2})
    calls = []

    def fill(o):
        calls.append(1)
        o[("depth", 0)] = 3
        o["identity_selection/0"] = 4

    out._thunk = fill
    return out, calls


def test_lazy_outputs_present_keys_do_not_materialise():
    out, calls = _lazy()
    assert out[("disp", 0)] == 1 and ("disp", 0) in out and out.get(("cam_T_cam", 0, -1)) == 2
    assert not calls


@pytest.mark.parametrize("access", ["getitem", "contains", "get", "iter", "keys", "items", "values", "len", "copy", "dict", "eq",
                                    "pickle"])
def test_lazy_outputs_every_read_access_sees_the_full_schema(access):
    """vo/learner_new.py:132-172,241 puts these keys into a plain dict; whatever a caller does with it must see them."""
    out, calls = _lazy()
    want = {("disp", 0), ("cam_T_cam", 0, -1), ("depth", 0), "identity_selection/0"}
    if access == "getitem":
        assert out[("depth", 0)] == 3
    elif access == "contains":
        assert ("depth", 0) in out and "nope" not in out
    elif access == "get":
        assert out.get("identity_selection/0") == 4 and out.get("nope", 7) == 7
    elif access == "iter":
        assert set(iter(out)) == want
    elif access == "keys":
        assert set(out.keys()) == want
    elif access == "items":
        assert dict(out.items())[("depth", 0)] == 3
    elif access == "values":
        assert sorted(out.values()) == [1, 2, 3, 4]
    elif access == "len":
        assert len(out) == 4
    elif access == "copy":
        assert set(out.copy()) == want
    elif access == "dict":
        assert set(dict(out)) == want
    elif access == "eq":
        assert out == {("disp", 0): 1, ("cam_T_cam", 0, -1): 2, ("depth", 0): 3, "identity_selection/0": 4}
    elif access == "pickle":
        assert set(pickle.loads(pickle.dumps(out))) == want
    assert calls == [1]
    with pytest.raises(KeyError):
        out[("nope", 0)]
    assert calls == [1]                      # materialised once


# ------------------------------------------------------------------------------------------ FusedAdam
def _two_nets():
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, bias=True), torch.nn.BatchNorm2d(8))
    b = torch.nn.Sequential(torch.nn.Linear(5, 3))
    b.add_module("fc", torch.nn.Linear(3, 2))          # stands in for torchvision's unused encoder.fc (no arena slot)
    return a, b


def _fused(a, b, **kw):
    from deep_visual_slam_amd import dp
    named = [("0." + n, p) for n, p in a.named_parameters()] + [("1." + n, p) for n, p in b.named_parameters() if not n.startswith("fc.")]
    flat = dp.FlatParams(named)
    ref_order = list(a.parameters()) + list(b.parameters())      # what vo/train.py:114-117 hands to Adam
    return flat, dp.FusedAdam(flat, lr=1e-4, params=ref_order, **kw), ref_order


def test_fused_adam_is_an_optimizer_and_takes_the_reference_scheduler():
    """vo/train.py:120-124: PolynomialLR(self.optimizer, total_iters=epoch, power=0.9)."""
    a, b = _two_nets()
    flat, opt, _ = _fused(a, b)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.PolynomialLR(opt, total_iters=10, power=0.9)
    assert opt.param_groups[0]["initial_lr"] == 1e-4
    opt.step_count = 1                       # pretend a step happened (the kernel itself needs the GPU)
    sched.step()
    assert abs(opt.lr - 1e-4 * (1 - 1 / 10) ** 0.9) < 1e-12
    assert "last_epoch" in sched.state_dict()


def test_fused_adam_state_dict_is_torch_adam_layout_and_round_trips():
    a, b = _two_nets()
    flat, opt, order = _fused(a, b)
    assert opt.state_dict()["state"] == {}                  # like torch: no state before the first step
    opt.step_count = 3
    opt.exp_avg.copy_(torch.randn_like(opt.exp_avg))
    opt.exp_avg_sq.copy_(torch.rand_like(opt.exp_avg_sq))
    sd = opt.state_dict()
    # same layout as torch.optim.Adam over the reference's parameter list (the two `fc` tensors have no state)
    ref = torch.optim.Adam(order, lr=1e-4)
    for p in order[:-2]:
        p.grad = torch.zeros_like(p)
    ref.step()
    rsd = ref.state_dict()
    assert set(sd["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "amsgrad", "params"}
    assert sd["param_groups"][0]["params"] == rsd["param_groups"][0]["params"] == list(range(len(order)))
    assert set(sd["state"]) == set(rsd["state"]) == set(range(len(order) - 2))
    for i, p in enumerate(order[:-2]):
        assert sd["state"][i]["exp_avg"].shape == p.shape and sd["state"][i]["exp_avg"].is_contiguous()
        assert float(sd["state"][i]["step"]) == 3.0
    # torch.optim.Adam loads it ...
    ref.load_state_dict(sd)
    assert torch.equal(ref.state[order[0]]["exp_avg"], sd["state"][0]["exp_avg"])
    # ... and a second FusedAdam restores the arena from torch's own state_dict
    a2, b2 = _two_nets()
    flat2, opt2, order2 = _fused(a2, b2)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.step_count == 3
    for p_, o_ in zip(flat.tensors, flat.offsets):                # slot by slot (the alignment padding between slots is not state)
        n_ = p_.numel()
        assert torch.equal(opt2.exp_avg[o_:o_ + n_], opt.exp_avg[o_:o_ + n_])
        assert torch.equal(opt2.exp_avg_sq[o_:o_ + n_], opt.exp_avg_sq[o_:o_ + n_])
    # conv weights are stored [Cout][kh][kw][Cin] in the arena but checkpointed in the logical shape
    conv_state = sd["state"][0]["exp_avg"]
    slot = [i for i, t in enumerate(flat.tensors) if t is order[0]][0]
    o = flat.offsets[slot]
    assert torch.equal(opt.exp_avg[o:o + conv_state.numel()].view(8, 3, 3, 4).permute(0, 3, 1, 2), conv_state)
    # round-1 flat layout still loads
    opt2.load_state_dict({"step": 5, "exp_avg": opt.exp_avg * 2, "exp_avg_sq": opt.exp_avg_sq, "param_groups": [{"lr": 3e-5}]})
    assert opt2.step_count == 5 and opt2.lr == 3e-5


def test_fused_adam_rejects_unsupported_options_and_mismatched_checkpoints():
    a, b = _two_nets()
    flat, opt, order = _fused(a, b)
    with pytest.raises(ValueError):
        opt.load_state_dict({"state": {}, "param_groups": [{"params": [0, 1]}]})
    from deep_visual_slam_amd import dp
    with pytest.raises(ValueError):
        dp.FusedAdam(flat, params=order[1:])                    # an arena tensor missing from the numbering


def test_zero_grad_keeps_the_arena_views():
    a, b = _two_nets()
    flat, opt, order = _fused(a, b)
    flat.grads.fill_(1.0)
    for p in order:
        p.grad = None                                            # what the reference's zero_grad(set_to_none=True) does
    opt.zero_grad(set_to_none=True)
    assert float(flat.grads.abs().max()) == 0.0
    for p, o in zip(flat.tensors, flat.offsets):
        assert p.grad is not None and p.grad.data_ptr() == flat.grads.data_ptr() + 4 * o


# ------------------------------------------------------------------------------------------ fold generation
def test_fold_cache_follows_raw_pointer_writers():
    """dvs_adam_step / dvs_bn_fwd write weights and running statistics through raw pointers: torch's _version does not
    move, nn_ops.bump_generation() (called by FusedAdam.step and the training-mode BatchNorm wrappers) must."""
    from deep_visual_slam_amd import nn_ops
    torch.manual_seed(0)
    w = torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(8).eval()
    w_f, _ = nn_ops.folded_bn(w, bn)
    assert nn_ops.folded_bn(w, bn)[0] is w_f
    w.permute(0, 2, 3, 1).view(-1).numpy()[:] *= 0.5                                  # a write torch's version counter does not see
    v = w._version
    assert nn_ops.folded_bn(w, bn)[0] is w_f and w._version == v  # stale -- which is exactly the hazard
    nn_ops.bump_generation()
    w_g, _ = nn_ops.folded_bn(w, bn)
    assert w_g is not w_f and torch.allclose(w_g, w / torch.sqrt(bn.running_var + bn.eps).view(-1, 1, 1, 1))


# ------------------------------------------------------------------------------------------ no library fallback
def test_primitives_fail_loudly_without_gpu_or_kernel():
    from deep_visual_slam_amd import _lib, nn_ops
    import deep_visual_slam_amd.nn_ops as N
    src = open(N.__file__).read()
    for banned in ("F.conv2d", "F.batch_norm", "F.max_pool2d", "F.interpolate", "F.pad", "DVS_CONV_BACKEND"):
        assert banned not in src, banned
    x = torch.zeros(1, 16, 8, 8)
    w = torch.zeros(16, 16, 3, 3)
    bn = torch.nn.BatchNorm2d(16)
    for call in (lambda: nn_ops.conv2d(x, w), lambda: nn_ops.conv_bn_act(x, w, bn), lambda: nn_ops.max_pool_3x3_s2(x),
                 lambda: nn_ops.upsample_nearest2x(x)):
        with pytest.raises(_lib.DvsError):
            call()


# ------------------------------------------------------------------------------------------ input pipeline (host side)
def test_jitter_records_and_intrinsics_pyramid():
    import numpy as np
    from deep_visual_slam_amd import synth
    from deep_visual_slam_amd.input_pipeline import JitterParams, intrinsics_pyramid
    jp = JitterParams(5, np.random.default_rng(0))
    assert jp.order.shape == (5, 4) and all(sorted(r) == [0, 1, 2, 3] for r in jp.order.tolist())
    assert (np.abs(jp.factor[:, :3] - 1) <= 0.3).all() and (np.abs(jp.factor[:, 3]) <= 0.2).all()
    jp.apply[:] = [True, False, True, True, False]
    rec = jp.records(3)
    assert rec.shape == (15, 8) and rec.dtype == np.int32
    assert (rec[3:6, :4] == -1).all() and (rec[0:3, :4] == jp.order[0]).all()           # frames of a sample share a record
    assert np.array_equal(rec[7, 4:].view(np.float32), jp.factor[2])
    # K / inv_K pyramid: same numbers as the synthetic-sample builder that follows vo/dataset/common.py:65-75
    ref = synth.intrinsics(2, 480, 640)
    got = intrinsics_pyramid(ref[("K", 0)].numpy(), 480, 640)
    for s in range(4):
        assert torch.allclose(got[("K", s)], ref[("K", s)], atol=1e-6) and torch.allclose(got[("inv_K", s)], ref[("inv_K", s)], atol=1e-8)


def test_color_jitter_oracle_identity_and_range():
    """The restated ColorJitter: factor 1 / shift 0 is the identity (up to the hsv round trip), output stays in [0, 1]."""
    from oracle import input_pipeline as OI
    img = torch.rand(3, 20, 30, generator=torch.Generator().manual_seed(0))
    same = OI.color_jitter(img, [0, 1, 2, 3], [1.0, 1.0, 1.0, 0.0])
    assert float((same - img).abs().max()) < 1e-5
    out = OI.color_jitter(img, [3, 1, 0, 2], [1.3, 0.7, 1.3, -0.2])
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0 and float((out - img).abs().max()) > 0.05


def test_winograd_cost_model_and_eligibility():
    """conv.wino_pays / wino_eligible / wino_dec_eligible are host logic: which launches take the Winograd kernels."""
    import torch
    from deep_visual_slam_amd import conv as DC
    enc = [(64, 120, 160), (128, 60, 80), (256, 30, 40), (512, 15, 20)]          # stride-1 3x3 layers of ResNet-18 at 480x640
    assert all(DC.wino_pays(12, h, w, c, c) and DC.wino_pays(24, h, w, c, c) for c, h, w in enc)
    assert [DC.wino_pays(1, h, w, c, c) for c, h, w in enc] == [True, False, False, False]
    assert [DC.wino_pays(4, h, w, c, c) for c, h, w in enc] == [True, True, False, False]
    w = torch.empty(128, 64, 3, 3)
    assert DC.wino_eligible(w, 1, 1, False, None, None, False, None)
    assert not DC.wino_eligible(w, 2, 1, False, None, None, False, None)          # stride 2
    assert not DC.wino_eligible(w, 1, 1, True, None, None, False, None)           # reflection pad: the decoder's variant
    assert not DC.wino_eligible(w, 1, 1, False, "relu", None, False, None)        # fused activation
    assert not DC.wino_eligible(torch.empty(128, 48, 3, 3), 1, 1, False, None, None, False, None)      # K % 16, K >= 64
    assert not DC.wino_eligible(torch.empty(128, 64, 1, 1), 1, 0, False, None, None, False, None)
    x, skip = torch.empty(2, 64, 6, 8), torch.empty(2, 64, 12, 16)
    wd = torch.empty(64, 128, 3, 3)
    assert DC.wino_dec_eligible(wd, 1, 1, True, "elu", x, skip, False, None)
    assert not DC.wino_dec_eligible(wd, 1, 1, True, "elu", x, torch.empty(2, 32, 12, 16), False, None)     # channels do not add up
    assert not DC.wino_dec_eligible(wd, 1, 1, True, "sigmoid", x, skip, False, None)
    assert DC.wino_wgrad_eligible((128, 64, 3, 3)) and not DC.wino_wgrad_eligible((48, 64, 3, 3))
    # decoder weight gradient on the Winograd kernel's gathers: sources in 32-channel blocks, one round of workgroups of work
    dec = [(512, 256, 15, 20), (512, 256, 30, 40), (256, 128, 30, 40), (256, 128, 60, 80), (128, 64, 60, 80), (128, 64, 120, 160),
           (64, 32, 120, 160), (96, 32, 240, 320)]                                 # (Cin, Cout, H, W) of upconv_4_0 ... upconv_1_1
    assert all(DC.wino_dec_wgrad_pays(b, h, w_, ci, co) for b in (2, 4, 12) for ci, co, h, w_ in dec)
    assert not DC.wino_dec_wgrad_pays(2, 6, 8, 64, 64) and not DC.wino_dec_wgrad_pays(1, 15, 20, 512, 256)
    assert DC.wino_dec_wgrad_eligible((64, 128, 3, 3), x, skip) and DC.wino_dec_wgrad_eligible((32, 64, 3, 3), x, None)
    assert not DC.wino_dec_wgrad_eligible((16, 64, 3, 3), x, None)                 # Cout in 32-blocks
    assert not DC.wino_dec_wgrad_eligible((64, 80, 3, 3), x, torch.empty(2, 16, 12, 16))      # skip channels in 32-blocks
    assert not DC.wino_dec_wgrad_eligible((64, 64, 3, 3), torch.empty(2, 64, 1, 8), None)     # ReflectionPad2d(1) needs two rows


def test_packed_weights_release_only_their_own_entries():
    """PackedWeights.release() of a late-collected owner must leave the entries a newer owner registered at the
    same (reused) address alone."""
    import weakref
    from deep_visual_slam_amd import conv as DC

    class _W:                                  # stands in for a CUDA weight: release() only looks at data_ptr()
        def __init__(self, addr):
            self.addr = addr

        def data_ptr(self):
            return self.addr

    old_w, new_w = _W(4096), _W(4096)          # the allocator handed the freed arena's address to the next arena
    pack = DC.PackedWeights.__new__(DC.PackedWeights)
    pack.weights = [old_w]
    DC._prepacked[4096] = ("pack-of-new", 0, (1,), weakref.ref(new_w))
    DC._wino_packed[4096] = ["u", "uf", 0, (1,), weakref.ref(new_w)]
    try:
        pack.release()
        assert DC._prepacked[4096][0] == "pack-of-new" and DC._wino_packed[4096][0] == "u"
        DC._prepacked[4096] = ("pack-of-old", 0, (1,), weakref.ref(old_w))
        pack.release()
        assert 4096 not in DC._prepacked and 4096 in DC._wino_packed
    finally:
        DC._prepacked.pop(4096, None)
        DC._wino_packed.pop(4096, None)


def test_stream_override_is_scoped_and_nests():
    """_lib.on_stream hands the dvs_* calls another stream without switching torch's current stream: the handle is visible inside
    the block only, nests, and is restored when the block raises."""
    from deep_visual_slam_amd import _lib

    class S:
        def __init__(self, h):
            self.cuda_stream = h

    assert getattr(_lib._stream_tls, "override", None) is None
    with _lib.on_stream(S(111)):
        assert _lib.stream() == 111
        with _lib.on_stream(S(222)):
            assert _lib.stream() == 222
        assert _lib.stream() == 111
    assert getattr(_lib._stream_tls, "override", None) is None
    try:
        with _lib.on_stream(S(333)):
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert getattr(_lib._stream_tls, "override", None) is None


def test_bf16_kernels_are_opt_in():
    """The patch kernels of the bf16 mode are only ever chosen while the mode is on; the default precision is fp32."""
    import torch
    from deep_visual_slam_amd import _lib, conv
    assert _lib.precision() == "fp32"
    w = torch.zeros(64, 64, 3, 3)
    x = torch.zeros(1, 64, 8, 8)
    assert not conv.p16_eligible(w, 1, 1, False, None, None, False, None)
    assert not conv.p16_dec_eligible(w, 1, 1, True, "elu", x, None, False, None)
    assert conv._wino_on() == conv._WINO
