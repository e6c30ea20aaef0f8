"""Tolerances for loss curves after the first optimiser step, derived from a committed fixture instead of hard-coded.

tests/golden/envelope.json (oracle/gen_golden_r2.py) holds the REAL reference's 100-iteration loss curve of one
configuration (64x64, batch 2, horse2zebra hyper-parameters) computed twice — with 1 and with 8 intra-op threads. Same
code, same seeds, same batches: only the summation order inside the reference's own conv / reduction kernels differs
(relative perturbation ~1e-7 at iteration 1). The gap between the two curves is what the training dynamics do to a
perturbation of that size: ~5e-5 at iteration 2, 1e-3 at 3, 1e-2 at 5-7, 0.1-0.5 from iteration ~10 on (Adam's first
updates are +-lr*sign(g): a sign flip of a rounding-level gradient moves a weight by 2*lr, and a GAN amplifies it).
No implementation — the reference included — can be asked to match "1e-3 over 100 steps" point-wise; what can be
asked is to stay inside the reference's own scatter.

    envelope(family, s)      cumulative max over iterations <= s of the relative gap of that loss family
    tolerance(key, s, eps0)  the scatter the reference itself shows for an initial perturbation eps0:
                             K * envelope(family, s + shift(eps0)), floored at eps0

shift(eps0) is the first iteration at which the reference's own fp32 reordering noise has grown to eps0: a pipeline
that starts with a perturbation eps0 (bf16 storage: 2e-2 on the adversarial terms, 5e-3 on the L1 cycle terms at
iteration 0, measured and asserted by the step-0 tests) is that many iterations "ahead" on the same amplification
curve. K = 3 covers that the fixture is one sample of the scatter, not its maximum.
"""
import json
from functools import lru_cache
from pathlib import Path

GOLD = Path(__file__).parent / "golden"
K = 3.0
FAMILIES = {"adv": ("G_AB", "G_BA", "D_A", "D_B", "G", "D"), "cycle": ("cycle_A", "cycle_B", "idt_A", "idt_B")}


def family(key):
    return "cycle" if key.startswith(("cycle", "idt", "pix2pix", "NCE")) else "adv"


@lru_cache()
def _curves():
    e = json.loads((GOLD / "envelope.json").read_text())
    return e["config"], e["threads_1"], e["threads_8"]


@lru_cache()
def _cummax(fam):
    _, a, b = _curves()
    out, run = [], 0.0
    for sa, sb in zip(a, b):
        for k, vb in sb["losses"].items():
            if family(k) == fam:
                run = max(run, abs(sa["losses"][k] - vb) / abs(vb))
        out.append(run)
    return out


def envelope(fam, step):
    c = _cummax(fam)
    return c[min(max(step, 0), len(c) - 1)]


def shift(fam, eps0):
    c = _cummax(fam)
    return next((s for s, v in enumerate(c) if v >= eps0), len(c) - 1)


def tolerance(key, step, eps0):
    """relative tolerance for loss `key` at iteration `step` (0-based) of a run whose iteration-0 error is eps0"""
    fam = family(key)
    if step == 0:
        return eps0
    return max(eps0, K * envelope(fam, step + shift(fam, eps0)))


@lru_cache()
def window_tolerance(fam, width=10):
    """K x the largest gap between the two reference runs' window means of that loss family"""
    _, a, b = _curves()
    worst = 0.0
    for k in b[0]["losses"]:
        if family(k) != fam:
            continue
        for lo in range(0, len(b), width):
            ma = sum(s["losses"][k] for s in a[lo:lo + width]) / len(a[lo:lo + width])
            mb = sum(s["losses"][k] for s in b[lo:lo + width]) / len(b[lo:lo + width])
            worst = max(worst, abs(ma - mb) / abs(mb))
    return K * worst


def reference_curve():
    """(config, per-iteration records) of the 8-thread reference run"""
    cfg, _, b = _curves()
    return cfg, b


EPS0 = {"adv": 2e-2, "cycle": 5e-3}      # iteration-0 tolerances of the bf16 HIP path (same weights, same batch)


def step_tolerance(key, s, eps0=None):
    """bf16 HIP path vs the reference's golden losses: iteration 0 is arithmetic parity at bf16 level (EPS0, asserted
    as such); later iterations get the reference's own scatter for a perturbation of that size"""
    fam = family(key)
    return tolerance(key, s, (eps0 or EPS0)[fam])
