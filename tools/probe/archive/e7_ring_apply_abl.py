"""timing ablations of gs_gconv_ring_apply's rendezvous (option ring_apply bits: 2 = no spin, 4 = no slot sums, 8 = no atomics at
all; results are wrong with any of them): where do the ~20 us per tile go?"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ganslate_amd.hip.ops import HipOps
from ganslate_amd.nn.native.spec import ConvSpec, lower
from ganslate_amd.nn.native.twin import Twin
from tools.bench_trunk import timed

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
for has_g2 in (True, False):
    fz = {"y": y, "mean_rstd": mr, "g2": g2 if has_g2 else None, "partial": plan[1], "fold": 1, "fold_mode": "reflect",
          "act": "none" if has_g2 else "relu", "slope": 0.2}
    base = timed(lambda: ops.gconv(low.dgrad_ring, gy, w, None, dy, fuse=fz), 200)
    print(f"g2={has_g2}: ring launch alone {base:6.1f} us")
    for bits, label in ((1, "full"), (3, "no spin"), (7, "no spin, no slot sums"), (15, "no atomics, no slot sums")):
        ops.set_option("ring_apply", bits)
        t = timed(lambda: ops.gconv_ring_apply(low.dgrad_ring, gy, w, dy, tot if has_g2 else None, fz, sync), 200)
        sync.zero_()
        print(f"   apply inside, {label:26s} {t:6.1f} us  (+{t - base:5.1f})")
    ops.set_option("ring_apply", 1)
