"""The optimiser update in chunks under the backward pass (NativeAdam.arm_early, NativeNet._early_step_at): Pix2Pix's generator
takes one backward pass per step (ganslate/nn/gans/paired/pix2pix.py:84-88: backward_G(); optimizers['G'].step()), so the layers
the pass is done with may be updated while it goes on. Same arithmetic per element: parameters, moments and weight packs must
equal the update-after-backward form bit for bit."""
import pytest
import torch

from ganslate_amd.nn.native import backend
from oracle.ops_ref import RefOps

from .helpers import build_product_pix2pix, run_product_pix2pix_steps


@pytest.fixture()
def fp32_oracle_backend():
    backend.set_ops(RefOps(act_dtype=torch.float32))
    yield
    backend.set_ops(None)


def _run(monkeypatch, mode, n_steps=2):
    """(callers take the fp32_oracle_backend fixture: the product recipe on the CPU oracle ops)"""
    from ganslate_amd.nn.optim import NativeAdam
    monkeypatch.setenv("GS_EARLY_ADAM", mode)
    monkeypatch.setattr(NativeAdam, "EARLY_MIN", 256)
    c = dict(size=[32, 32], batch=1, steps=n_steps, n_iters=100, n_iters_decay=100, num_downs=5, ngf=8, use_dropout=False,
             n_layers=2, lambda_pix2pix=30.0, seed=35)
    model = build_product_pix2pix(c)
    chunks = []
    opt = model.optimizers["G"]
    inner = opt._update_range

    def spy(p, net, start, end):
        chunks.append((start, end))
        return inner(p, net, start, end)
    monkeypatch.setattr(opt, "_update_range", spy)
    fused = []
    ops = backend.get_ops()
    inner_wa = ops.wgrad_adam
    monkeypatch.setattr(ops, "wgrad_adam", lambda w, *a, **k: (fused.append(id(w)), inner_wa(w, *a, **k))[1])
    ranged = []
    inner_rg = ops.adam_step_dev_ranges
    monkeypatch.setattr(ops, "adam_step_dev_ranges",
                        lambda p, g, m, v, r, *a, **k: (ranged.append([tuple(x) for x in r.tolist()]), inner_rg(p, g, m, v, r, *a, **k))[1])
    logs = run_product_pix2pix_steps(model, c, n_steps)
    G = model.networks["G"]
    st = opt.state[G.master]
    packs = next(iter(G._packs.values()))
    G.refresh_packs(torch.zeros(1, 3, 32, 32))
    return {"master": G.master.detach().clone(), "m": st["exp_avg"].clone(), "v": st["exp_avg_sq"].clone(),
            "fpack": packs["fpack"].clone(), "dpack": packs["dpack"].clone(), "logs": logs, "chunks": chunks,
            "numel": G.numel, "step": st["step"], "fused": list(fused),
            "tr": set(getattr(G, "_tr_fresh", ())), "ranged": list(ranged)}


def test_chunked_update_equals_update_after_backward(fp32_oracle_backend, monkeypatch):
    late = _run(monkeypatch, "0")
    early = _run(monkeypatch, "1")
    assert not late["chunks"] and not late["fused"] and early["step"] == late["step"] == 2
    assert len(early["fused"]) == 2 * 10, "every layer's weight gradient of the 5-level U-Net took the fused launch (oracle ops)"
    # ... and wrote its transposed pack (data-gradient pack of the convs, forward pack of the transposed convs) with it
    assert early["tr"] == {(i, "d") for i in range(5)} | {(i, "f") for i in range(5, 10)} and not late["tr"]
    # the first step handed the whole flat buffer over in adjoining chunks (several); the second one — the network's layers took
    # the fused launch in the first — updated what the fused launches left (the biases) in ONE multi-range launch behind the pass
    ch = early["chunks"]
    assert len(ch) >= 3
    covered = sorted(ch)
    assert covered[0][0] == 0 and covered[-1][1] == early["numel"]
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), covered
    assert len(early["ranged"]) == 1 and not late["ranged"]
    (rng,) = early["ranged"]
    assert len(rng) == 10 and all(b - a <= 64 for a, b in rng), "ten bias vectors"
    for k in ("master", "m", "v", "fpack", "dpack"):
        assert torch.equal(early[k], late[k]), k
    assert early["logs"] == late["logs"]
