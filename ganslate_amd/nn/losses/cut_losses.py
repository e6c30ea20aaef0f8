"""PatchNCE (InfoNCE over feature patches) — ganslate/nn/losses/cut_losses.py:5-43.

In this package the loss is not a stand-alone criterion: the patch MLP, the L2 normalisation, the logits
[q.k+ , q.k_j (j != i, diagonal -> -10)] / T and the cross-entropy against class 0 run as ONE hand-written kernel family
(`csrc/patchnce.hip`, `gs_patchnce_forward/backward`) behind `FeaturePatchMLP.nce_loss`
(`ganslate_amd/nn/gans/unpaired/cut.py`), which is what `CUT` calls. A torch-ops restatement used to live here; it was
never on the product path and silently left the HIP kernels when instantiated from a user recipe, so the name now fails
loudly and points at the fused entry."""


class PatchNCELoss:

    def __init__(self, conf):
        raise NotImplementedError(
            "ganslate_amd has no stand-alone PatchNCELoss: the loss is fused with FeaturePatchMLP "
            "(FeaturePatchMLP.nce_loss -> gs_patchnce_forward / gs_patchnce_backward, csrc/patchnce.hip). "
            "Call mlp.nce_loss(target_feats, source_feats, batch, nce_T, lambda_nce) as CUT._calculate_nce_loss does.")
