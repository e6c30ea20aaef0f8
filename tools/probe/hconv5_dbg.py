import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ganslate_amd.hip.ops import HipOps
from ganslate_amd.nn.native.spec import ConvSpec, lower
from tests.test_ops_gpu import make_layer
ops = HipOps(); dev = ops.device
sizes = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (8, 16, 16)
spec, N = ConvSpec("conv", 16, 16, 5, 1, 2, dims=3), 1
low, master, bias, fpack, dpack = make_layer(spec, sizes, 31)
g = torch.Generator().manual_seed(32)
x = torch.randn(N, *sizes, 16, generator=g).to(torch.bfloat16)
outs = {}
for o in (1, 0):
    ops.set_option("hconv5", o)
    y = torch.zeros(N, *sizes, 16, dtype=torch.bfloat16, device=dev)
    ops.gconv_classes(low.fwd, x.to(dev), fpack.to(dev), bias.to(dev), y)
    torch.cuda.synchronize()
    outs[o] = y.float().cpu()
d = (outs[1] - outs[0]).abs()
print("max diff", d.max().item(), "ref scale", outs[0].abs().max().item())
bad = (d > 0.1).nonzero()
print("bad count", bad.shape[0], "of", d.numel())
for ax, name in enumerate(["n", "z", "y", "x", "co"]):
    vals, cnt = torch.unique(bad[:, ax], return_counts=True)
    print(name, list(zip(vals.tolist(), cnt.tolist()))[:40])
print("finite", torch.isfinite(outs[1]).all().item())
