"""PatchNCE (InfoNCE over feature patches) — ganslate/nn/losses/cut_losses.py:5-43.

`CUT` itself does not go through this class: the patch MLP, the L2 normalisation, the logits [q.k+ , q.k_j (j != i,
diagonal -> -10)] / T and the cross-entropy against class 0 run as ONE hand-written kernel family (`csrc/patchnce.hip`,
`gs_patchnce_forward/backward`) behind `FeaturePatchMLP.nce_loss` (`ganslate_amd/nn/gans/unpaired/cut.py`).

The class stays importable and WORKING for user recipes that instantiate the reference's criterion by name on their own
(already normalised) features: it is the reference's arithmetic in torch ops on whatever device the features live on — not a
hand-written kernel, and it says so once per process. Same constructor and call signature, same per-row loss vector."""
import logging

import torch

_warned = False


class PatchNCELoss:

    def __init__(self, conf):
        self.batch_size = conf.train.batch_size
        self.nce_T = conf.train.gan.optimizer.nce_T

    def to(self, device):
        return self

    def __call__(self, feat_q, feat_k):
        return self.forward(feat_q, feat_k)

    def forward(self, feat_q, feat_k):
        """feat_q / feat_k: [batch * patches, dim] (target / source features of one level) -> loss per row"""
        global _warned
        if not _warned:
            logging.getLogger("ganslate_amd").warning(
                "PatchNCELoss: stand-alone criterion running as torch ops; CUT's own loss is the fused kernel family behind "
                "FeaturePatchMLP.nce_loss (csrc/patchnce.hip)")
            _warned = True
        rows, dim = feat_q.shape[:2]
        feat_k = feat_k.detach()
        l_pos = (feat_q * feat_k).sum(1, keepdim=True)                       # the positive: same patch of the other image
        q = feat_q.view(self.batch_size, -1, dim)
        k = feat_k.view(self.batch_size, -1, dim)
        patches = q.shape[1]
        l_neg = torch.bmm(q, k.transpose(2, 1))                               # the other patches of the SAME image
        eye = torch.eye(patches, device=feat_q.device, dtype=torch.bool)[None]
        l_neg = l_neg.masked_fill(eye, -10.0).view(-1, patches)               # (a patch against itself: exp(-10) ~ 0)
        out = torch.cat((l_pos, l_neg), dim=1) / self.nce_T
        return torch.nn.functional.cross_entropy(out, torch.zeros(rows, dtype=torch.long, device=feat_q.device),
                                                 reduction="none")
