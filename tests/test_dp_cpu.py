"""Data-parallel path on CPU: flat arena + bucketed all-reduce over gloo, world_size 2."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 3, padding=1),
                               torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from deep_visual_slam_amd import dp
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    m = _model()
    flat = dp.FlatParams(dp.trainable_parameters(m))
    sync = dp.GradSync(flat, bucket_bytes=256)          # tiny buckets -> several all-reduces per step
    assert len(sync.buckets) > 1
    torch.manual_seed(100)
    x = torch.randn(4, 3, 8, 8)
    res = []
    for step in range(2):                                 # two steps: bucket bookkeeping must reset
        xs = x[rank * 2:(rank + 1) * 2] + step
        loss = m(xs).pow(2).mean()
        loss.backward()
        sync.finish()
        res.append((flat.grads * sync.grad_scale).clone())
        flat.zero_grad()
    q.put((rank, [r.numpy() for r in res], flat.names, flat.offsets))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    # single-process reference on the full batch (mean over ranks of rank-mean losses == full-batch mean)
    from deep_visual_slam_amd import dp
    m = _model()
    flat = dp.FlatParams(dp.trainable_parameters(m))
    torch.manual_seed(100)
    x = torch.randn(4, 3, 8, 8)
    for step in range(2):
        m(x + step).pow(2).mean().backward()
        ref = flat.grads.clone().numpy()
        flat.zero_grad()
        for r in range(world):
            assert abs(got[r][1][step] - ref).max() < 1e-6
    assert (got[0][1][0] == got[1][1][0]).all()           # ranks hold identical reduced gradients


def test_flat_params_are_views_in_backward_order():
    from deep_visual_slam_amd import dp
    m = _model()
    flat = dp.FlatParams(dp.trainable_parameters(m))
    assert flat.names[0].endswith("4.bias") and flat.names[-1].endswith("0.weight")
    assert all(o % 4 == 0 for o in flat.offsets)
    for p, o in zip(flat.tensors, flat.offsets):
        assert p.data_ptr() == flat.params.data_ptr() + 4 * o
        assert p.grad.data_ptr() == flat.grads.data_ptr() + 4 * o
    m(torch.randn(1, 3, 4, 4)).sum().backward()
    assert float(flat.grads.abs().sum()) > 0
    for p in m.parameters():
        p.grad = None
    flat.reattach()
    assert all(p.grad is not None for p in m.parameters())


def test_fc_head_is_excluded():
    from deep_visual_slam_amd import dp
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.posenet_single import PoseNet
    torch.manual_seed(0)
    named = dp.trainable_parameters(DepthNet(18, False), PoseNet(18, False, 2))
    assert sum(p.numel() for _, p in named) == 26828186
    assert not any(".fc." in n for n, _ in named)


def test_post_accumulate_hook_fires_for_none_gradients():
    """dp.GradSync counts a parameter as complete when its post-accumulate-grad hook fires.  Gradient sinks
    (gradsink.py) make the autograd Functions return None for their parameters; the hook must still fire, once,
    after the last use."""
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            return x * w

        @staticmethod
        def backward(ctx, g):
            return g, None

    w = torch.nn.Parameter(torch.ones(3))
    w.grad = torch.zeros(3)
    x = torch.ones(3, requires_grad=True)
    fired = []
    w.register_post_accumulate_grad_hook(lambda p: fired.append(1))
    (F.apply(x, w) + F.apply(2 * x, w)).sum().backward()
    assert fired == [1]
    assert float(w.grad.abs().max()) == 0.0


def test_bucket_layout_follows_the_streams():
    """Buckets tile the arena in order, never span parameters registered under different streams (the two networks run
    their backward passes concurrently), and each stream segment ends in a small tail bucket (the gradients that only
    exist at the very end of the step)."""
    import torch
    from deep_visual_slam_amd import dp
    torch.manual_seed(0)
    a = torch.nn.Sequential(*[torch.nn.Linear(256, 256) for _ in range(12)])      # 12 x 65 792 parameters
    b = torch.nn.Sequential(*[torch.nn.Linear(256, 256) for _ in range(7)])
    flat = dp.FlatParams(dp.trainable_parameters(a, b), grad_sinks=False)
    marker = object()
    streams = {id(p): marker for p in b.parameters()}
    gs = dp.GradSync(flat, bucket_bytes=1 << 20, hook_streams=streams)            # 262 144 floats per bucket
    assert gs.buckets[0][0] == 0 and gs.buckets[-1][1] == flat.numel
    assert all(x[1] == y[0] for x, y in zip(gs.buckets, gs.buckets[1:]))
    owner = [streams.get(id(p)) for p in flat.tensors]
    for bkt in range(len(gs.buckets)):
        members = {owner[i] for i, bo in enumerate(gs.bucket_of) if bo == bkt}
        assert len(members) == 1, "bucket %d spans two streams" % bkt
    # the last bucket of each segment is the small tail (<= bucket/8 elements, or the segment's last tensor alone when that
    # is larger), the others reach the requested size
    seg_last = [i for i in range(len(owner)) if i + 1 == len(owner) or owner[i + 1] is not owner[i]]
    seg_last_bucket = [gs.bucket_of[i] for i in seg_last]
    for i, bkt in zip(seg_last, seg_last_bucket):
        s, e = gs.buckets[bkt]
        assert e - s <= max((1 << 20) // 4 // 8, flat.tensors[i].numel()) + 256
    full = [e - s for k, (s, e) in enumerate(gs.buckets) if k not in seg_last_bucket and k + 1 not in seg_last_bucket]
    assert all(n >= (1 << 20) // 4 for n in full)


class _RecordingComm:
    """Stands in for dp.RcclComm on the CPU: same two entry points, carried by gloo, recording what GradSync hands it."""

    def __init__(self):
        self.calls = []

    def all_reduce_(self, tensor):
        import torch.distributed as dist
        self.calls.append(("one", [(tensor.storage_offset(), tensor.storage_offset() + tensor.numel())]))
        dist.all_reduce(tensor)

    def all_reduce_ranges_(self, base, ranges):
        import torch.distributed as dist
        self.calls.append(("group", [(int(s), int(e)) for s, e in ranges]))
        for s, e in ranges:
            dist.all_reduce(base[s:e])

    def wait(self):
        pass


def _worker_comm(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from deep_visual_slam_amd import dp
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    out = []
    for use_comm in (False, True):
        m = _model()
        flat = dp.FlatParams(dp.trainable_parameters(m))
        # two "networks": the first conv's parameters on one (fake) stream key, the rest on another, as bench.py keys PoseNet's
        streams = {id(p): ("a" if i < 2 else "b") for i, p in enumerate(m.parameters())}
        comm = _RecordingComm() if use_comm else None
        sync = dp.GradSync(flat, bucket_bytes=256, hook_streams=streams, comm=comm)
        # the tail rule is sized for megabyte buckets (2 MB tails); on this toy model mark each network's last bucket by hand
        keys = [streams[id(p)] for p in flat.tensors]
        sync.deferred = {sync.bucket_of[i] for i in range(len(keys)) if i + 1 == len(keys) or keys[i + 1] != keys[i]}
        torch.manual_seed(100)
        x = torch.randn(4, 3, 8, 8)
        m(x[rank * 2:(rank + 1) * 2]).pow(2).mean().backward()
        sync.finish()
        out.append((flat.grads.clone().numpy(), list(sync.buckets), sorted(sync.deferred), comm.calls if comm else None))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_hands_the_comm_exactly_the_bucket_ranges():
    """with a communicator (the direct RCCL path) GradSync must reduce exactly the element ranges the
    torch.distributed path reduces -- every bucket once, the deferred tails as one group -- and end with the same gradients."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_comm, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in range(world):
        (g_plain, buckets, _, _), (g_comm, buckets2, deferred, calls) = got[rank]
        assert buckets == buckets2
        assert abs(g_plain - g_comm).max() == 0.0
        handed = sorted(r for _, rs in calls for r in rs)
        assert handed == sorted(buckets)                      # every bucket exactly once, nothing else
        groups = [rs for kind, rs in calls if kind == "group"]
        assert len(deferred) == 2                            # one tail per "network"
        assert len(groups) == 1 and sorted(groups[0]) == sorted(buckets[b] for b in deferred)
