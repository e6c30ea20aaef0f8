"""ganslate/nn/utils.py restated for the native backend: bias rule (:71-80), LR schedule (:83-99), init (:8-36)."""
from torch.optim import lr_scheduler


def init_net(network, conf, device):
    network.init_weights(conf.train.gan.weight_init_type, conf.train.gan.weight_init_gain)
    return network.to(device)


def require_instance_norm(norm_type):
    if norm_type == "batch":
        raise NotImplementedError("norm_type 'batch' has no MI355X-native path: every BASELINE config uses "
                                  "InstanceNorm (configs/base.py:55 default)")
    if norm_type != "instance":
        raise NotImplementedError(f"Normalization layer `{norm_type}` not supported")


def is_bias_before_norm(norm_type="instance"):
    if norm_type == "instance":
        return True
    elif norm_type == "batch":
        return False
    raise NotImplementedError(f"Normalization layer `{norm_type}` not supported")


def get_scheduler(optimizer, conf):
    """constant LR for n_iters, then linear decay to zero over n_iters_decay (nn/utils.py:83-99)"""

    def lambda_rule(iter_idx):
        start_iter = 1
        if conf.train.checkpointing.load_iter:
            start_iter += conf.train.checkpointing.load_iter
        return 1.0 - max(0, iter_idx + start_iter - conf.train.n_iters) / float(conf.train.n_iters_decay + 1)

    return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
