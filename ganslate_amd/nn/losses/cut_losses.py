"""PatchNCE (InfoNCE over feature patches) — arithmetic of ganslate/nn/losses/cut_losses.py:5-43:
logits = [q.k+ , q.k_j (j != i, diagonal -> -10)] / T, cross-entropy against class 0, `feat_k` detached, negatives
are the other patches of the SAME image (bmm batched by train.batch_size).
The tensors are tiny ((B*256) x 256 per layer); they run as plain library GEMMs + softmax on the device."""
import torch


class PatchNCELoss:

    def __init__(self, conf):
        self.batch_size = conf.train.batch_size
        self.nce_T = conf.train.gan.optimizer.nce_T

    def to(self, device):
        return self

    def __call__(self, feat_q, feat_k):
        bs, dim = feat_q.shape[:2]
        feat_k = feat_k.detach()
        l_pos = (feat_q * feat_k).sum(1, keepdim=True)
        q = feat_q.view(self.batch_size, -1, dim)
        k = feat_k.view(self.batch_size, -1, dim)
        n = q.size(1)
        l_neg = torch.bmm(q, k.transpose(2, 1))
        l_neg = l_neg.masked_fill(torch.eye(n, device=q.device, dtype=torch.bool)[None], -10.0).view(-1, n)
        out = torch.cat((l_pos, l_neg), dim=1) / self.nce_T
        return torch.nn.functional.cross_entropy(out, torch.zeros(out.size(0), dtype=torch.long, device=q.device),
                                                 reduction="none")
