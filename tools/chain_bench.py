"""Time the fused loss-chain kernels alone (HIP events inside the library) at BASELINE sizes."""
import sys, json
import torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import ops, synth, dp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = 480, 640
dev = torch.device("cuda:0")
sample = synth.throughput_sample(B, H, W, device=dev)
g = torch.Generator().manual_seed(0)
disps = [(torch.rand(B, 1, H >> s, W >> s, generator=g) * 0.8 + 0.1).to(dev).requires_grad_(True) for s in range(S)]
poses = [p.to(dev).requires_grad_(True) for p in synth.parity_poses(B)]
def step():
    T_l = ops.pose_to_mat(poses[0][:, 0], poses[1][:, 0], True)
    T_r = ops.pose_to_mat(poses[2][:, 0], poses[3][:, 0], False)
    losses, _, _ = ops.loss_chain(sample[("target_image", 0)], sample[("source_left", 0)], sample[("source_right", 0)],
                                  sample[("K", 0)], sample[("inv_K", 0)], T_l, T_r, disps, seed=1)
    losses.mean().backward()
for _ in range(3): step()
torch.cuda.synchronize()
dp.profile_enable(True)
for _ in range(10): step()
torch.cuda.synchronize()
prof = dp.profile_read()
dp.profile_enable(False)
bytes_fwd = {4: 45.87e6, 1: 12.29e6}.get(S, 45.87e6 * S / 4) * B
out = {k: {"avg_ms": ms / n, "GBps_algorithmic": bytes_fwd / (ms / n * 1e-3) / 1e9} for k, (ms, n, _w) in prof.items()}
print(json.dumps({"B": B, "S": S, **out}))
