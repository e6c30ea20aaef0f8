"""Cycle-consistency and identity losses — interface and weighting of ganslate/nn/losses/cyclegan_losses.py:7-101:
cycle_A = lambda_AB * [alpha*SSIMdist + (1-alpha)*L1](real_A, rec_A); idt_B = lambda_AB * lambda_idt * L1(idt_B, real_B)
(note the reference pairs idt_B with lambda_AB and idt_A with lambda_BA, :50-52)."""
from .functional import l1_loss, scalar_affine, ssim_distance_autograd


class CycleGANLosses:

    def __init__(self, conf):
        opt = conf.train.gan.optimizer
        self.lambda_AB, self.lambda_BA = opt.lambda_AB, opt.lambda_BA
        self.criterion_cycle = CycleLoss(opt.proportion_ssim)
        self.criterion_idt = IdentityLoss(opt.lambda_identity) if opt.lambda_identity > 0 else None

    def is_using_identity(self):
        return bool(self.criterion_idt)

    def __call__(self, visuals):
        real_A, real_B = visuals["real_A"], visuals["real_B"]
        rec_A, rec_B = visuals["rec_A"], visuals["rec_B"]
        idt_A, idt_B = visuals["idt_A"], visuals["idt_B"]
        if self.criterion_idt and (idt_A is None or idt_B is None):
            raise ValueError("idt_A and/or idt_B is not computed but the identity loss is defined.")
        if type(self.criterion_cycle) is CycleLoss and type(self.criterion_idt) in (IdentityLoss, type(None)):
            # the library's own criterions: every weighted term of every loss, then ONE launch for the scalar algebra
            names = ["cycle_A", "cycle_B"]
            parts = [[(self.lambda_AB * w, x) for w, x in self.criterion_cycle.terms(real_A, rec_A)],
                     [(self.lambda_BA * w, x) for w, x in self.criterion_cycle.terms(real_B, rec_B)]]
            if self.criterion_idt:
                names += ["idt_B", "idt_A"]
                parts += [[(self.lambda_AB * w, x) for w, x in self.criterion_idt.terms(idt_B, real_B)],
                          [(self.lambda_BA * w, x) for w, x in self.criterion_idt.terms(idt_A, real_A)]]
            xs = [x for p in parts for _, x in p]
            rows, k = [], 0
            for p in parts:
                rows.append([0.0] * k + [w for w, _ in p] + [0.0] * (len(xs) - k - len(p)))
                k += len(p)
            return dict(zip(names, scalar_affine(xs, rows)))
        losses = {
            "cycle_A": self.lambda_AB * self.criterion_cycle(real_A, rec_A),
            "cycle_B": self.lambda_BA * self.criterion_cycle(real_B, rec_B),
        }
        if self.criterion_idt:
            losses["idt_B"] = self.lambda_AB * self.criterion_idt(idt_B, real_B)
            losses["idt_A"] = self.lambda_BA * self.criterion_idt(idt_A, real_A)
        return losses


class CycleLoss:

    def __init__(self, proportion_ssim):
        self.alpha = proportion_ssim
        self.beta = 1 - proportion_ssim

    def terms(self, real, reconstructed):
        """[(weight, 0-d loss)] whose weighted sum is the loss"""
        l1 = l1_loss(reconstructed, real)
        if self.alpha > 0:
            return [(self.alpha, ssim_distance_autograd(reconstructed, real)), (self.beta, l1)]
        return [(1.0, l1)]

    def __call__(self, real, reconstructed):
        t = self.terms(real, reconstructed)
        return scalar_affine([x for _, x in t], [[w for w, _ in t]])[0] if len(t) > 1 else t[0][1]


class IdentityLoss:

    def __init__(self, lambda_identity):
        self.lambda_identity = lambda_identity

    def terms(self, idt, real):
        return [(self.lambda_identity, l1_loss(idt, real))]

    def __call__(self, idt, real):
        (w, x), = self.terms(idt, real)
        return scalar_affine([x], [[w]])[0]
