"""Data-parallel path on the GPU: two ranks sharing one MI355X (gloo carries the CUDA tensors), the real
networks, gradient sinks, PoseNet / weight-gradient side streams and the bucketed all-reduce issued from the
ready hooks.  A bucket reduced before all of its gradients were written (a missing stream fence) leaves the
two ranks with different arenas, which is what this test looks for."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from deep_visual_slam_amd import _lib, dp, gradsink, synth
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from deep_visual_slam_amd.posenet_single import PoseNet
    _lib.set_deterministic(True)           # same ReLU branches in the local and the reduced pass: the sums can be compared tightly
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    B, H, W = 2, 96, 128
    cfg = {"Train": dict(num_source=1, batch_size=B, img_h=H, img_w=W, smoothness_ratio=0.001, auto_mask=True,
                         ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}
    torch.manual_seed(5)                                   # same weights on both ranks
    dn = DepthNet(18, pretrained=False).to(dev).train()
    pn = PoseNet(18, pretrained=False, num_input_images=2).to(dev).train()
    flat = dp.FlatParams(dp.trainable_parameters(dn, pn))
    tr = MonodepthTrainer(dn, pn, cfg, dev)
    sample = {k: v.to(dev) for k, v in synth.parity_sample(B, H, W, seed=10 + rank).items()}   # different data per rank
    g = torch.Generator().manual_seed(3)
    noise = torch.stack([torch.randn(B, 2, H, W, generator=g) for _ in range(4)]).to(dev)
    # local gradients, no exchange (GradSync does not exist yet: its hooks would reduce)
    tr._noise = noise
    _, losses = tr.process_batch(dict(sample))
    losses["loss"].backward()
    gradsink.join()
    torch.cuda.synchronize()
    local = flat.grads.clone()
    flat.zero_grad()
    torch.cuda.synchronize()
    # the real thing
    sync = dp.GradSync(flat, bucket_bytes=4 << 20)         # ~26 buckets
    out = []
    for step in range(2):                                  # twice: bucket bookkeeping must reset
        tr._noise = noise
        _, losses = tr.process_batch(dict(sample))
        losses["loss"].backward()
        sync.finish()
        torch.cuda.synchronize()
        out.append(flat.grads.clone().cpu().numpy())
        flat.zero_grad()
        torch.cuda.synchronize()
    q.put((rank, local.cpu().numpy(), out, len(sync.buckets), list(sync.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reduce_identical_arenas(gpu_device):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][3] > 8
    for step in range(2):
        a, b = got[0][2][step], got[1][2][step]
        assert np.array_equal(a, b), "ranks disagree after the all-reduce: %d elements" % int((a != b).sum())
    # the reduced arena is the sum of the two local gradients (deterministic forward: only the smooth rounding noise of the
    # backward's float atomics separates the two passes; round 1 had to allow 2e-2 for flipped ReLU branches)
    ref = got[0][1].astype(np.float64) + got[1][1].astype(np.float64)
    err = np.linalg.norm(got[0][2][0] - ref) / np.linalg.norm(ref)
    per_bucket = [(float(np.linalg.norm(got[0][2][0][s:e] - ref[s:e]) / (np.linalg.norm(ref[s:e]) + 1e-30)), s, e)
                  for s, e in got[0][4]]
    assert err < 5e-5, (err, sorted(per_bucket, reverse=True)[:6], float(np.linalg.norm(got[0][2][0])), float(np.linalg.norm(ref)))


def test_direct_rccl_allreduce_single_rank(gpu_device):
    """include/dvslam_rccl.h on the GPU: a one-rank communicator (what this one-GPU box can hold; the 8-GPU run is the
    driver's): unique id, init, the sum all-reduce on the communicator's own stream ordered behind the producer stream,
    wait, destroy.  With one rank the sum is the identity."""
    from deep_visual_slam_amd import dp
    comm = dp.RcclComm(gpu_device)
    assert comm.world == 1 and comm.rank == 0 and comm.stream.cuda_stream != torch.cuda.current_stream().cuda_stream
    x = torch.arange(1 << 20, device=gpu_device, dtype=torch.float32)
    y = x * 2.0                                     # produced on the current stream; the all-reduce must wait for it
    comm.all_reduce_(y)
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(y, x * 2.0)
    # as GradSync uses it: buckets of the arena reduced from hooks, finish() waits for the communicator's stream
    p = torch.nn.Parameter(torch.randn(1000, device=gpu_device))
    flat = dp.FlatParams([("p", p)])
    sync = dp.GradSync(flat, comm=comm)
    flat.grads.fill_(3.0)
    sync._reduce(0, flat.numel)
    sync.world = 2                                  # exercise finish()'s communicator wait
    sync._reduced = [True] * len(sync.buckets)
    sync.finish()
    torch.cuda.synchronize()
    assert float(flat.grads.min()) == 3.0 and float(flat.grads.max()) == 3.0
    comm.close()
