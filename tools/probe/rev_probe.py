import sys; sys.path.insert(0, "/root/repo")
import torch
from oracle import torch_ref
from oracle.ops_ref import RefOps
from ganslate_amd.nn.native import backend
from ganslate_amd.hip.ops import HipOps
from ganslate_amd.nn.generators import Vnet2D, Vnet3D
from tests.test_cyclegan_gpu import cosine, rel_l2
hip = HipOps()
for dims, ch, c, blocks, shape in [(3, 1, 8, ((1, 2), (2, 1)), (1, 1, 16, 24, 32)), (2, 2, 8, ((1, 2), (2, 1)), (2, 2, 64, 96)), (2, 2, 8, None, (1, 2, 128, 192))]:
    V, R = (Vnet3D, torch_ref.Vnet3D) if dims == 3 else (Vnet2D, torch_ref.Vnet2D)
    kw = {} if blocks is None else dict(down_blocks=blocks[0], up_blocks=blocks[1])
    shadow = R(ch, ch, c, use_inverse=True, **kw)
    sd = torch_ref.seeded_state_dict(shadow, 81); shadow.load_state_dict(sd)
    g = torch.Generator().manual_seed(82)
    x = torch.rand(shape, generator=g) * 2 - 1
    gy, gr = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    xa = x.clone().requires_grad_(); ya = shadow(xa); ra = shadow(ya, inverse=True)
    ((ya * gy).sum() + (ra * gr).sum()).backward()
    res = {}
    for name, ops in (("hip", hip), ("emu", RefOps(act_dtype=torch.bfloat16))):
        backend.set_ops(ops)
        net = V(ch, ch, "instance", c, **kw); net.load_state_dict(sd)
        xb = x.clone().to(ops.device).requires_grad_(); yb = net(xb); rb = net(yb, inverse=True)
        ((yb * gy.to(ops.device)).sum() + (rb * gr.to(ops.device)).sum()).backward()
        res[name] = (yb.detach().cpu(), rb.detach().cpu(), xb.grad.cpu(), {k: v.float().cpu() for k, v in net.grads_state_dict().items()})
    backend.set_ops(hip)
    normed = {nd.name for nd in net.nodes if nd.norm}
    print(shape, "y", rel_l2(res["hip"][0], ya.detach()), "r", rel_l2(res["hip"][1], ra.detach()), "r emu", rel_l2(res["hip"][1], res["emu"][1]))
    worst = {"fp32": (0, 1, ""), "emu": (0, 1, ""), "emu_vs_fp32": (0, 1, "")}
    for n, p in shadow.named_parameters():
        if n.startswith("encoder.") or (n.endswith(".bias") and n[:-5] in normed): continue
        if p.dim() == 1: continue      # PReLU slopes / biases: looked at separately
        for key, a, b in (("fp32", res["hip"][3][n], p.grad), ("emu", res["hip"][3][n], res["emu"][3][n]), ("emu_vs_fp32", res["emu"][3][n], p.grad)):
            r, cs = rel_l2(a, b), cosine(a, b)
            if r > worst[key][0]: worst[key] = (r, cs, n)
    print("  worst:", {k: (round(v[0], 3), round(v[1], 3), v[2]) for k, v in worst.items()})
    print("  gx: fp32", rel_l2(res["hip"][2], xa.grad), cosine(res["hip"][2], xa.grad), "emu", rel_l2(res["hip"][2], res["emu"][2]))
