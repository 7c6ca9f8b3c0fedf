"""The reference caller's literal sequences on the HIP path (vo/train.py is not importable here -- it pulls in
tensorboard / the datasets -- so its two step functions are restated line for line):

  * train_mono_step (vo/train.py:173-199): optimizer.zero_grad(set_to_none=True); process_batch; backward;
    torch.optim.Adam.step(); losses[k].detach().cpu() -- with STOCK torch.optim.Adam over the plain modules, two steps,
    against oracle networks + oracle chain + torch.optim.Adam on the CPU (losses of both steps, updated weights);
  * valid_mono_step (vo/train.py:201-217, called under .eval(), :311-331): eval() + no_grad through process_batch
    (folded-BatchNorm inference path), against the oracle's eval-mode networks;
  * train -> eval -> FusedAdam step -> eval: the folded-BatchNorm cache must follow the raw-pointer writers
    (nn_ops.folded_bn generation counter);
  * eval() WITH autograd (validation loss without no_grad): running statistics, differentiable.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
B, H, W = 2, 96, 128
LR = 1e-4


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cfg():
    return {"Train": dict(num_source=1, batch_size=B, img_h=H, img_w=W, smoothness_ratio=0.001, auto_mask=True,
                          ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}


def _nets(dev, seed=0):
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.posenet_single import PoseNet
    torch.manual_seed(seed)
    dn, pn = DepthNet(18, pretrained=False), PoseNet(18, pretrained=False, num_input_images=2)
    sd_d = {k: v.clone() for k, v in dn.state_dict().items()}
    sd_p = {k: v.clone() for k, v in pn.state_dict().items()}
    return dn.to(dev), pn.to(dev), sd_d, sd_p


def _oracle_losses(sample, sd_d, sd_p, noise, train, update=None):
    from oracle import loss_chain as OL, networks as ON
    tgt, left, right = sample[("target_image", 0)], sample[("source_left", 0)], sample[("source_right", 0)]
    ud, up = ({}, {}) if update is None else update
    disp = ON.depthnet(tgt, sd_d, train=train, update=ud)
    aa_l, t_l = ON.posenet(torch.cat([left, tgt], 1), sd_p, train=train, update=up)
    sd_p2 = dict(sd_p)
    sd_p2.update({k: v for k, v in up.items()})              # the second PoseNet call sees the first call's running stats
    aa_r, t_r = ON.posenet(torch.cat([tgt, right], 1), sd_p2, train=train, update=up)
    _, losses = OL.loss_chain(sample, [disp[("disp", s)] for s in range(4)], (aa_l, t_l, aa_r, t_r), noise)
    return losses


def test_stock_train_mono_step_two_steps(gpu_device):
    from deep_visual_slam_amd import synth
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    dn, pn, sd_d, sd_p = _nets(gpu_device)
    dn.train()
    pn.train()
    sample = synth.parity_sample(B, H, W)
    g = torch.Generator().manual_seed(7)
    noises = [[torch.randn(B, 2, H, W, generator=g) for _ in range(4)] for _ in range(2)]
    # ---- CPU side: oracle + torch.optim.Adam
    cd = {k: v.clone().requires_grad_(v.is_floating_point() and ".fc." not in k and "running" not in k) for k, v in sd_d.items()}
    cp = {k: v.clone().requires_grad_(v.is_floating_point() and ".fc." not in k and "running" not in k) for k, v in sd_p.items()}
    copt = torch.optim.Adam([v for v in list(cd.values()) + list(cp.values()) if v.requires_grad], lr=LR)
    ref_losses = []
    for it in range(2):
        copt.zero_grad(set_to_none=True)
        ud, up = {}, {}
        losses = _oracle_losses(sample, cd, cp, noises[it], True, (ud, up))
        losses["loss"].backward()
        copt.step()
        with torch.no_grad():
            for d, u in ((cd, ud), (cp, up)):
                for k, v in u.items():
                    d[k] = v.detach().clone()
        ref_losses.append({k: float(v) for k, v in losses.items()})
    # ---- GPU side: vo/train.py:114-117 optimiser, :173-199 step, unchanged
    optimizer = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), lr=LR)
    learner = MonodepthTrainer(dn, pn, _cfg(), gpu_device)
    for it in range(2):
        learner._noise = torch.stack(noises[it]).to(gpu_device)
        optimizer.zero_grad(set_to_none=True)
        outputs, losses = learner.process_batch(dict(sample))
        total_loss = losses["loss"]
        total_loss.backward()
        optimizer.step()
        total_loss = total_loss.detach()
        for key in losses:
            losses[key] = losses[key].detach().cpu()
        assert sorted(losses) == ["loss", "loss/0", "loss/1", "loss/2", "loss/3"]
        tol = 2e-4 if it == 0 else 1e-3          # step 2 runs on weights that went through one sign-like Adam update
        for k, v in ref_losses[it].items():
            assert abs(float(losses[k]) - v) < tol * abs(v), (it, k, float(losses[k]), v)
    torch.cuda.synchronize()
    # Adam's first steps move every weight by ~lr * sign(g): elements whose tiny gradient has the other sign differ by
    # up to 2 * lr * steps; everything else must agree to a small fraction of lr
    n_bad = n_all = 0
    for mod, ref in ((dn, cd), (pn, cp)):
        for k, p in mod.named_parameters():
            if ".fc." in k:
                assert torch.equal(p.detach().cpu(), ref[k].detach())      # never touched on either side
                continue
            d = (p.detach().cpu() - ref[k].detach()).abs()
            assert float(d.max()) <= 8.0 * LR, k
            n_bad += int((d > 0.2 * LR).sum())
            n_all += d.numel()
    assert n_bad / n_all < 0.02, n_bad / n_all
    for mod, ref in ((dn, cd), (pn, cp)):
        sd = mod.state_dict()
        for k in ("encoder.encoder.bn1.running_mean", "encoder.encoder.layer4.1.bn2.running_var"):
            assert rel(sd[k], ref[k]) < 1e-3, k
        assert int(sd["encoder.encoder.bn1.num_batches_tracked"]) == int(ref["encoder.encoder.bn1.num_batches_tracked"])


def test_valid_mono_step_eval_no_grad(gpu_device):
    """vo/train.py:311-331: depth_net.eval(); pose_net.eval(); valid_mono_step (torch.no_grad) -> process_batch."""
    from deep_visual_slam_amd import synth
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    dn, pn, sd_d, sd_p = _nets(gpu_device, seed=1)
    # non-trivial running statistics / affine parameters, same on both sides
    gen = torch.Generator().manual_seed(5)
    for sd, mod in ((sd_d, dn), (sd_p, pn)):
        for k in sd:
            if k.endswith("running_mean"):
                sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.2
            elif k.endswith("running_var"):
                sd[k] = torch.rand(sd[k].shape, generator=gen) + 0.5
            elif ".bn" in k and k.endswith(".weight"):
                sd[k] = torch.rand(sd[k].shape, generator=gen) + 0.5
        mod.load_state_dict(sd)
    dn.eval()
    pn.eval()
    sample = synth.parity_sample(B, H, W)
    g = torch.Generator().manual_seed(9)
    noise = [torch.randn(B, 2, H, W, generator=g) for _ in range(4)]
    with torch.no_grad():
        ref = _oracle_losses(sample, sd_d, sd_p, noise, False)
    learner = MonodepthTrainer(dn, pn, _cfg(), gpu_device)
    learner._noise = torch.stack(noise).to(gpu_device)
    with torch.no_grad():
        outputs, losses = learner.process_batch(dict(sample))
    total_loss = losses["loss"].detach()
    assert not total_loss.requires_grad
    for k in ref:
        assert abs(float(losses[k]) - float(ref[k])) < 2e-4 * abs(float(ref[k])), k
    assert outputs[("depth", 0)].shape == (B, 1, H, W)                       # what PlotTool / the commented line reads
    assert int(dn.state_dict()["encoder.encoder.bn1.num_batches_tracked"]) == 0   # eval: buffers untouched


def test_eval_after_fused_adam_step_uses_the_new_weights(gpu_device):
    """train -> eval -> train step with dp.FusedAdam (raw-pointer weight + running-stat updates) -> eval: the second eval
    pass must be computed from the updated weights, not from the first pass's cached BatchNorm folds."""
    from deep_visual_slam_amd import dp, synth
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from oracle import networks as ON
    dn, pn, _, _ = _nets(gpu_device, seed=2)
    flat = dp.FlatParams(dp.trainable_parameters(dn, pn))
    opt = dp.FusedAdam(flat, lr=1e-2, params=list(dn.parameters()) + list(pn.parameters()))   # large lr: visible change
    learner = MonodepthTrainer(dn, pn, _cfg(), gpu_device)
    sample = synth.parity_sample(B, H, W)
    x = sample[("target_image", 0)]
    pair = torch.cat([sample[("source_left", 0)], x], 1)

    def eval_pass():
        dn.eval()
        pn.eval()
        with torch.no_grad():
            d = dn(x.to(gpu_device))[("disp", 0)].clone()
            aa, t = pn(pair.to(gpu_device))
        return d, aa.clone(), t.clone()

    def oracle_pass():
        sd_d = {k: v.detach().cpu().clone() for k, v in dn.state_dict().items()}
        sd_p = {k: v.detach().cpu().clone() for k, v in pn.state_dict().items()}
        with torch.no_grad():
            d = ON.depthnet(x, sd_d, train=False)[("disp", 0)]
            aa, t = ON.posenet(pair, sd_p, train=False)
        return d, aa, t

    d0, aa0, t0 = eval_pass()
    r0 = oracle_pass()
    assert rel(d0, r0[0]) < 2e-4 and rel(aa0, r0[1]) < 2e-4
    dn.train()
    pn.train()
    opt.zero_grad(set_to_none=True)
    _, losses = learner.process_batch(dict(sample))
    losses["loss"].backward()
    opt.step()
    d1, aa1, t1 = eval_pass()
    r1 = oracle_pass()
    assert rel(r1[0], r0[0]) > 1e-3                       # the step really changed the function
    assert rel(d1, r1[0]) < 2e-4 and rel(aa1, r1[1]) < 5e-4 and rel(t1, r1[2]) < 5e-4
    # Adam-layout checkpoint written by the fused optimiser loads into torch.optim.Adam over the same parameters
    sd = opt.state_dict()
    ref_opt = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), lr=1e-2)
    ref_opt.load_state_dict(sd)
    p0 = next(iter(dn.parameters()))
    assert float(ref_opt.state[p0]["step"]) == 1.0 and ref_opt.state[p0]["exp_avg"].shape == p0.shape
    # and the reference scheduler drives it (vo/train.py:120-124)
    sched = torch.optim.lr_scheduler.PolynomialLR(opt, total_iters=30, power=0.9)
    sched.step()
    assert opt.lr < 1e-2


def test_eval_mode_with_autograd_is_differentiable_and_uses_running_stats(gpu_device):
    """Validation loss computed WITHOUT no_grad (the reference allows it): eval-mode BatchNorm = running statistics,
    gradients flow to the conv weights and to gamma / beta; PoseNet's pair batching must not be required."""
    from oracle import networks as ON
    dn, pn, sd_d, sd_p = _nets(gpu_device, seed=3)
    gen = torch.Generator().manual_seed(6)
    for sd, mod in ((sd_d, dn), (sd_p, pn)):
        for k in sd:
            if k.endswith("running_mean"):
                sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.2
            elif k.endswith("running_var"):
                sd[k] = torch.rand(sd[k].shape, generator=gen) + 0.5
        mod.load_state_dict(sd)
    dn.eval()
    pn.eval()
    torch.manual_seed(8)
    x6 = torch.rand(2 * B, 6, H, W)
    cp = {k: v.clone().requires_grad_(v.is_floating_point() and ".fc." not in k and "running" not in k) for k, v in sd_p.items()}
    aa_r, t_r = ON.posenet(x6, cp, train=False)
    cot = torch.randn(2 * B, 1, 1, 3)
    ((aa_r + t_r) * cot).sum().backward()
    aa, t = pn(x6.to(gpu_device), pairs=2)              # what MonodepthTrainer._predict_poses issues
    assert aa.requires_grad and rel(aa, aa_r) < 2e-4 and rel(t, t_r) < 2e-4
    pn.zero_grad(set_to_none=True)
    ((aa + t) * cot.to(gpu_device)).sum().backward()
    # eval mode: no batch coupling, but one ReLU branch flip still moves a tensor by ~1/sqrt(elements of that map)
    # (tools/grad_truth.py --eval-bn: fp32 CPU vs fp64 shows the same steps); smallest map here: 4 x 3 x 4 x 256
    flip = 1.0 / (2 * B * 3 * 4 * 256) ** 0.5
    worst = max((rel(p.grad, cp[n].grad), n) for n, p in pn.named_parameters() if ".fc." not in n)
    assert worst[0] < 2e-3 + 2 * flip, worst
    assert int(pn.state_dict()["encoder.encoder.bn1.num_batches_tracked"]) == 0
    # DepthNet: all four disparity maps, gradient to a BatchNorm gamma deep in the encoder
    x3 = torch.rand(B, 3, H, W)
    cd = {k: v.clone().requires_grad_(v.is_floating_point() and ".fc." not in k and "running" not in k) for k, v in sd_d.items()}
    ref = ON.depthnet(x3, cd, train=False)
    out = dn(x3.to(gpu_device))
    for s in range(4):
        assert rel(out[("disp", s)], ref[("disp", s)]) < 2e-4
    sum(ref[("disp", s)].mean() for s in range(4)).backward()
    sum(out[("disp", s)].mean() for s in range(4)).backward()
    flip = 1.0 / (B * 3 * 4 * 512) ** 0.5
    for n in ("encoder.encoder.layer3.0.bn2.weight", "encoder.encoder.layer2.0.downsample.1.bias", "encoder.encoder.conv1.weight"):
        assert rel(dict(dn.named_parameters())[n].grad, cd[n].grad) < 2e-3 + 2 * flip, n
    assert rel(dict(dn.named_parameters())["decoder.9.conv.conv.weight"].grad, cd["decoder.9.conv.conv.weight"].grad) < 1e-4


def test_one_line_optimizer_swap_keeps_the_caller_sequence(gpu_device):
    """replace only the optimiser construction of vo/train.py:114-117 by dp.FusedAdam over a FlatParams arena;
    `optimizer.zero_grad(set_to_none=True)`, `backward()`, `optimizer.step()`, the scheduler and the `.cpu()` calls stay.
    Two steps against the same sequence with stock torch.optim.Adam on identical weights and inputs."""
    from deep_visual_slam_amd import dp, gradsink, synth
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    sample = synth.parity_sample(B, H, W)
    g = torch.Generator().manual_seed(7)
    noises = [torch.stack([torch.randn(B, 2, H, W, generator=g) for _ in range(4)]).to(gpu_device) for _ in range(2)]
    results = []
    for fused in (False, True):
        gradsink.reset_streams()
        dn, pn, _, _ = _nets(gpu_device, seed=0)
        dn.train()
        pn.train()
        if fused:
            flat = dp.FlatParams(dp.trainable_parameters(dn, pn))
            optimizer = dp.FusedAdam(flat, lr=LR, params=list(dn.parameters()) + list(pn.parameters()))
        else:
            optimizer = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), lr=LR)
        scheduler = torch.optim.lr_scheduler.PolynomialLR(optimizer, total_iters=30, power=0.9)
        learner = MonodepthTrainer(dn, pn, _cfg(), gpu_device)
        seen = []
        for it in range(2):
            learner._noise = noises[it]
            optimizer.zero_grad(set_to_none=True)
            outputs, losses = learner.process_batch(dict(sample))
            total_loss = losses["loss"]
            total_loss.backward()
            optimizer.step()
            for key in losses:
                losses[key] = losses[key].detach().cpu()
            seen.append({k: float(v) for k, v in losses.items()})
        scheduler.step()
        torch.cuda.synchronize()
        results.append((seen, {k: v.detach().cpu().clone() for k, v in list(dn.named_parameters()) + list(pn.named_parameters())},
                        optimizer.param_groups[0]["lr"]))
    (l_a, w_a, lr_a), (l_b, w_b, lr_b) = results
    assert lr_a == lr_b and lr_a < LR
    for k in l_a[0]:
        assert abs(l_a[0][k] - l_b[0][k]) < 1e-5 * abs(l_a[0][k])              # same weights, same kernels (BatchNorm statistics are summed with float atomics: 2e-6 run to run)
        assert abs(l_a[1][k] - l_b[1][k]) < 1e-3 * abs(l_a[1][k])              # after one sign-like Adam update
    n_bad = n_all = 0
    for k in w_a:
        d = (w_a[k] - w_b[k]).abs()
        n_bad += int((d > 0.2 * LR).sum())
        n_all += d.numel()
    assert n_bad / n_all < 0.02, n_bad / n_all


def _train_mono_step(learner, optimizer, sample, use_amp, scaler=None, poison=None):
    """vo/train.py:173-199 line for line (`use_amp` defaults to True there: vo/train.py:44); `poison` plants a non-finite
    gradient between backward and the scaler's step (test hook)."""
    from torch.amp import autocast
    optimizer.zero_grad(set_to_none=True)
    if use_amp:
        with autocast(device_type="cuda", enabled=use_amp):
            outputs, losses = learner.process_batch(sample)
        total_loss = losses["loss"]
        scaler.scale(total_loss).backward()
        if poison is not None:
            poison()
        scaler.step(optimizer)
        scaler.update()
    else:
        outputs, losses = learner.process_batch(sample)
        total_loss = losses["loss"]
        total_loss.backward()
        optimizer.step()
    total_loss = total_loss.detach()
    for key in losses:
        losses[key] = losses[key].detach().cpu()
    return total_loss, outputs, losses


@pytest.mark.parametrize("fused", [False, True])
def test_amp_branch_of_train_mono_step(gpu_device, fused):
    """The branch the reference takes by default (vo/train.py:44 `use_amp` = True, :126-128 GradScaler, :177-185 autocast around
    process_batch, scaler.scale(loss).backward(), scaler.step, scaler.update), with stock Adam and with dp.FusedAdam.  Every kernel
    of this package computes in fp32 whatever autocast says and GradScaler's factor is a power of two, so two AMP steps must
    land on the weights of two plain steps (same criterion as the optimiser-swap test: Adam's first steps are sign-like); an
    injected inf must skip the optimiser step and halve the scale."""
    from torch.amp import GradScaler
    from deep_visual_slam_amd import dp, gradsink, synth
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    sample = synth.parity_sample(B, H, W)
    g = torch.Generator().manual_seed(7)
    noises = [torch.stack([torch.randn(B, 2, H, W, generator=g) for _ in range(4)]).to(gpu_device) for _ in range(3)]
    results = []
    for use_amp in (False, True):
        gradsink.reset_streams()
        dn, pn, _, _ = _nets(gpu_device, seed=0)
        dn.train()
        pn.train()
        if fused:
            flat = dp.FlatParams(dp.trainable_parameters(dn, pn))
            optimizer = dp.FusedAdam(flat, lr=LR, params=list(dn.parameters()) + list(pn.parameters()))
        else:
            optimizer = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), lr=LR)
        scaler = GradScaler() if use_amp else None
        learner = MonodepthTrainer(dn, pn, _cfg(), gpu_device)
        seen = []
        for it in range(2):
            learner._noise = noises[it]
            _, outputs, losses = _train_mono_step(learner, optimizer, dict(sample), use_amp, scaler)
            seen.append({k: float(v) for k, v in losses.items()})
            assert outputs[("disp", 0)].dtype == torch.float32
        torch.cuda.synchronize()
        named = [("d." + k, v) for k, v in dn.named_parameters()] + [("p." + k, v) for k, v in pn.named_parameters()]
        weights = {k: v.detach().cpu().clone() for k, v in named}
        results.append((seen, weights))
        if use_amp:
            # third step with a poisoned gradient: no weight may move, the scale halves (GradScaler's backoff)
            scale0 = scaler.get_scale()
            p_bad = dict(dn.named_parameters())["decoder.9.conv.conv.weight"]

            def poison():
                p_bad.grad[0, 0, 0, 0] = float("inf")         # (.grad is a [Cout][kh][kw][Cin]-stored view of the arena)

            learner._noise = noises[2]
            _train_mono_step(learner, optimizer, dict(sample), True, scaler, poison)
            torch.cuda.synchronize()
            for k, v in named:
                assert torch.equal(v.detach().cpu(), weights[k]), k
            assert scaler.get_scale() == 0.5 * scale0
            # and the step after that trains again
            _train_mono_step(learner, optimizer, dict(sample), True, scaler)
            torch.cuda.synchronize()
            moved = max(float((v.detach().cpu() - weights[k]).abs().max()) for k, v in named if ".fc." not in k)
            assert 0.0 < moved <= 2.0 * LR
    (l_a, w_a), (l_b, w_b) = results
    for k in l_a[0]:
        assert abs(l_a[0][k] - l_b[0][k]) < 1e-5 * abs(l_a[0][k])
        assert abs(l_a[1][k] - l_b[1][k]) < 1e-3 * abs(l_a[1][k])
    n_bad = n_all = 0
    for k in w_a:
        d = (w_a[k] - w_b[k]).abs()
        n_bad += int((d > 0.2 * LR).sum())
        n_all += d.numel()
    assert n_bad / n_all < 0.02, n_bad / n_all


def test_loss_that_bypasses_the_chain_joins_the_side_streams(gpu_device):
    """the trainer moves the networks into its arena (gradient sinks, weight gradients on side streams), but the
    backward pass need not start at the fused loss chain -- a supervised loss on DepthNet's outputs, clip_grad_norm_, a stock
    optimiser.  The first side-stream use of a pass queues the end-of-pass fence (gradsink.StreamSet), so the stock optimiser
    reads complete gradients: two steps against the same sequence without the arena (DVS_ARENA=0 path: plain autograd)."""
    from deep_visual_slam_amd import gradsink, synth
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    sample = synth.parity_sample(B, H, W)
    x = sample[("target_image", 0)].to(gpu_device)
    torch.manual_seed(11)
    gt = [torch.rand(B, 1, H >> s, W >> s, device=gpu_device) for s in range(4)]
    results = []
    for arena in (False, True):
        gradsink.reset_streams()
        dn, pn, _, _ = _nets(gpu_device, seed=4)
        dn.train()
        cfg = _cfg()
        cfg["Train"]["arena"] = arena
        learner = MonodepthTrainer(dn, pn, cfg, gpu_device)
        assert (learner.arena is not None) == arena
        optimizer = torch.optim.Adam(dn.parameters(), lr=LR)
        grads = None
        for it in range(2):
            optimizer.zero_grad(set_to_none=True)
            if arena:
                learner._arena_prepare()                      # what process_batch does at the start of a step
            with gradsink.use(learner.streams):
                out = dn(x)
            loss = sum(((out[("disp", s)] - gt[s]) ** 2).mean() for s in range(4))
            loss.backward()
            total = torch.nn.utils.clip_grad_norm_(dn.parameters(), 1e9)      # reads every .grad right after backward
            if it == 0:
                grads = {k: p.grad.detach().clone() for k, p in dn.named_parameters() if p.grad is not None}
            optimizer.step()
        torch.cuda.synchronize()
        results.append((float(loss), float(total), grads, {k: v.detach().cpu().clone() for k, v in dn.named_parameters()}))
    (la, ta, ga, wa), (lb, tb, gb, wb) = results
    assert abs(la - lb) < 1e-4 * abs(la) and abs(ta - tb) < 2e-3 * abs(ta)
    for k in ga:                                               # first-step gradients: same weights, same kernels
        assert rel(gb[k], ga[k]) < 5e-3, k
    n_bad = n_all = 0
    for k in wa:
        d = (wa[k] - wb[k]).abs()
        n_bad += int((d > 0.2 * LR).sum())
        n_all += d.numel()
    assert n_bad / n_all < 0.02, n_bad / n_all
