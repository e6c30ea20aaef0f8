"""arrival times (s_memtime) of the 512 tiles of a twin ring-apply launch at the rendezvous, without the spin (option ring_apply
bits 2 | 16): which boxes of an image are late, and by how much?"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ganslate_amd.hip.ops import HipOps
from ganslate_amd.nn.native.spec import ConvSpec, lower
from ganslate_amd.nn.native.twin import Twin

ops = HipOps()
dev = ops.device
C, N, H = 256, 16, 64
spec = ConvSpec("conv", C, C, 3, 1, 1, pad_mode="reflect")
low = lower(spec, H, H)
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).to(dev)
gy, y, g2 = rnd(N, H, H, C), rnd(N, H, H, C), rnd(N, H, H, C)
packs = (torch.randn(2, low.dgrad_index.size + 64, generator=g) * 0.05).to(torch.bfloat16).to(dev)
mr = torch.rand(N * 2 * C, device=dev) + 0.5
dy, tot = torch.empty_like(y), torch.empty_like(y)
w = Twin(packs[0], packs[1])
plan = ops.fused_ring_plan(low.dgrad_ring, N, C, twin=True)
sync = ops.ring_apply_plan(low.dgrad_ring, N, C, twin=True)
fz = {"y": y, "mean_rstd": mr, "g2": g2, "partial": plan[1], "fold": 1, "fold_mode": "reflect", "act": "none", "slope": 0.2}
ops.set_option("ring_apply", 1 | 2 | 16)
for rep in range(5):
    ops.gconv_ring_apply(low.dgrad_ring, gy, w, dy, tot, fz, sync)
torch.cuda.synchronize()
t = sync[256:256 + 2 * 512].cpu().view(512, 2).to(torch.int64)
t = (t[:, 0] & 0xffffffff) | (t[:, 1] << 32)
t = t.view(16, 16, 2).double()            # [image][box][channel tile]
sync.zero_()
ops.set_option("ring_apply", 1)
print("arrival in us (s_memrealtime, 100 MHz) relative to the earliest tile of the launch; rows = images, boxes 0..15 of channel tile 0")
t0 = t.min()
for n in range(16):
    grp = (t[n] - t0) / 100.0
    print(f"img {n:2d}: spread nt0 {grp[:, 0].max() - grp[:, 0].min():5.1f}  nt1 {grp[:, 1].max() - grp[:, 1].min():5.1f} | "
          + " ".join(f"{v:5.1f}" for v in grp[:, 0]))
