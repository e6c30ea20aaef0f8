"""Resnet2D generator on the HIP executor — same constructor, layer order, padding, bias rule and state_dict
names as ganslate/nn/generators/resnet/resnet2d.py:14-93:
c7s1-64, d128, d256, n x R256, u128, u64, c7s1-out, tanh; ReflectionPad before the k7 and residual convs;
InstanceNorm2d(affine=False) + ReLU; ConvTranspose2d(3, s2, p1, op1) up-sampling (always biased)."""
import os
from dataclasses import dataclass

import torch

from .... import configs
from ...native.net import NativeNet, Node
from ...native.spec import ConvSpec
from ...utils import is_bias_before_norm, require_instance_norm


@dataclass
class Resnet2DConfig(configs.base.BaseGeneratorConfig):
    n_residual_blocks: int = 9


def resnet_nodes(in_channels, out_channels, use_bias, n, dims=2, wfold=None):
    """layer list shared by Resnet2D (reflect padding, `encoder` alias — resnet2d.py:24,46,80) and Resnet3D
    (replicate padding, no `encoder` — resnet3d.py:15,24,78)"""
    pad_mode = "reflect" if dims == 2 else "replicate"
    enc = (lambda *names: tuple(names)) if dims == 2 else (lambda *names: ())
    conv = lambda *a, **k: ConvSpec(*a, dims=dims, **k)
    # the k7 convs at the image boundary have 1-3 channels on one side: their W taps are folded into the channel axis
    # so the 16-wide matrix tile is not mostly padding (csrc/wfold.hip); GS_WFOLD=0 keeps the plain lowering (A/B runs)
    if wfold is None:
        wfold = os.environ.get("GS_WFOLD", "1") != "0"
    wf_in = "in" if (wfold and 7 * in_channels <= 32) else ""
    wf_out = "out" if (wfold and 7 * out_channels <= 32) else ""
    nodes = [Node(conv("conv", in_channels, 64, 7, 1, 3, pad_mode=pad_mode, bias=use_bias, wfold=wf_in), True, "relu",
                  name="model.1", aliases=enc("encoder.1"))]
    feats = 64
    for d in range(2):
        idx = 4 + 3 * d
        nodes.append(Node(conv("conv", feats, feats * 2, 3, 2, 1, bias=use_bias), True, "relu",
                          name=f"model.{idx}", aliases=enc(f"encoder.{idx}")))
        feats *= 2
    for b in range(n):
        idx = 10 + b
        src = len(nodes) - 1                      # node whose output enters the block (x + conv_block(x))
        nodes.append(Node(conv("conv", feats, feats, 3, 1, 1, pad_mode=pad_mode, bias=use_bias), True,
                          "relu", name=f"model.{idx}.conv_block.1", aliases=enc(f"encoder.{idx}.conv_block.1")))
        nodes.append(Node(conv("conv", feats, feats, 3, 1, 1, pad_mode=pad_mode, bias=use_bias), True,
                          "none", res=src, name=f"model.{idx}.conv_block.5",
                          aliases=enc(f"encoder.{idx}.conv_block.5")))
    for u in range(2):
        idx = 10 + n + 3 * u
        nodes.append(Node(conv("convT", feats, feats // 2, 3, 2, 1, 1), True, "relu", name=f"model.{idx}"))
        feats //= 2
    nodes.append(Node(conv("conv", feats, out_channels, 7, 1, 3, pad_mode=pad_mode, bias=use_bias, wfold=wf_out),
                      False, "none", name=f"model.{17 + n}"))
    return nodes


class _ImageTapFn(torch.autograd.Function):
    """features of the ReflectionPad2d(pad) output of an fp32 NCHW image at sampled flat pixel ids -> [N, P, C]
    (gs_image_tap_gather); backward scatters through the same reflection map (gs_image_tap_scatter)"""

    @staticmethod
    def forward(ctx, x, pid, pad, ops):
        x = x.contiguous()
        ctx.ops, ctx.pid, ctx.pad, ctx.shape = ops, pid, pad, tuple(x.shape)
        return ops.image_tap_gather(x.detach(), pid, pad)

    @staticmethod
    def backward(ctx, g):
        return ctx.ops.image_tap_scatter(g, ctx.pid, ctx.shape, ctx.pad), None, None, None


class Resnet2D(NativeNet):

    def __init__(self, in_channels, out_channels, norm_type, n_residual_blocks=9):
        require_instance_norm(norm_type)
        use_bias = is_bias_before_norm(norm_type)
        n = n_residual_blocks
        # `encoder` aliases the first 10+n modules of `model` (resnet2d.py:46)
        nodes = resnet_nodes(in_channels, out_channels, use_bias, n, dims=2)
        self.encoder_nodes = 3 + 2 * n              # nodes that make up the reference's `encoder`
        self.n_residual_blocks = n
        super().__init__(nodes, in_channels, out_channels, out_act="tanh")

    # ---- feature taps for CUT (ganslate/nn/gans/unpaired/cut.py:297-312 walks `self.encoder` module by module) ----
    def encoder_len(self):
        return 10 + self.n_residual_blocks

    def encoder_tap(self, e):
        """encoder module index -> ("pad", None) | ("y", node) | ("x", node), channels.
        The reference appends the module OUTPUT; outputs of InstanceNorm modules are later overwritten in place by the
        following nn.ReLU(inplace=True), so taps on 2/5/8 observe the activated tensor (same as 3/6/9)."""
        assert 0 <= e < self.encoder_len(), f"encoder has {self.encoder_len()} layers"
        if e == 0:
            return ("pad", None), self.in_channels
        if e < 10:
            node, sub = (e - 1) // 3, (e - 1) % 3
            return (("y" if sub == 0 else "x"), node), self.nodes[node].spec.cout
        node = 3 + 2 * (e - 10) + 1
        return ("x", node), self.nodes[node].spec.cout

    def extract_patch_features(self, x, layers, ids, detached=False):
        """features of `layers` sampled at pixel ids (one LongTensor per layer) -> list of [N, P, C] fp32.
        detached: the caller will not differentiate them (CUT's source / key patches, cut_losses.py:16): if a recorded pass
        over this very tensor is still alive (fake_B = G(real_A) of the same iteration) they are read out of ITS activations
        instead of running the encoder again; else the encoder runs without being recorded."""
        import torch
        taps, where = [], []
        for e in layers:
            tap, _ = self.encoder_tap(e)
            where.append(tap)
            if tap[0] != "pad":
                taps.append(tap)
        tap_ids = [i for i, t in zip(ids, where) if t[0] != "pad"]
        x_pad = x
        if taps and len(taps) < len(where) and not detached:      # the image feeds the encoder AND the padded-image level
            from ...losses.functional import fanout
            x_pad, x = fanout(x)
        if detached and taps:
            xc = x.contiguous().float()
            rec = self.recorded_pass(xc)
            if rec is None:
                with torch.no_grad():
                    _, saved = self._forward(xc.detach(), save=True, stop=max(node for _, node in taps))
                n0 = 0
            else:
                saved, n0 = rec            # (the pass may have carried other batches in front of this one)
            feats, nb = [], xc.shape[0]
            for (kind, node), pid in zip(taps, tap_ids):
                src = saved.ys[node] if kind == "y" else saved.acts[node + 1]
                feats.append(self.ops.tap_gather(src[n0:n0 + nb], pid, self.nodes[node].spec.cout))
            native = iter(feats)
        else:
            native = iter(self.forward_taps(x, taps, tap_ids)) if taps else iter(())
        out = []
        for tap, pid in zip(where, ids):
            if tap[0] == "pad":      # layer 0 = the ReflectionPad2d(3) output of the boundary image
                out.append(_ImageTapFn.apply(x_pad.float(), pid, 3, self.ops))
            else:
                out.append(next(native))
        return out

    def extract_patch_features_parts(self, xs, layers, ids_per_part):
        """extract_patch_features for several batches in ONE encoder pass, each with its own pixel ids (CUT's target
        patches of fake_B and idt_B) -> per batch a list of [N_p, P, C] fp32"""
        taps, where = [], []
        for e in layers:
            tap, _ = self.encoder_tap(e)
            where.append(tap)
            if tap[0] != "pad":
                taps.append(tap)
        xs_pad = xs
        if taps and len(taps) < len(where):      # every image feeds the encoder AND the padded-image level
            from ...losses.functional import fanout
            pairs = [fanout(x) for x in xs]
            xs_pad, xs = [p[0] for p in pairs], [p[1] for p in pairs]
        native = self.forward_taps_parts(xs, taps, [[i for i, t in zip(ids, where) if t[0] != "pad"]
                                                     for ids in ids_per_part]) if taps else [() for _ in xs]
        out = []
        for x, ids, nat in zip(xs_pad, ids_per_part, native):
            nat, feats = iter(nat), []
            for tap, pid in zip(where, ids):
                if tap[0] == "pad":      # layer 0 = the ReflectionPad2d(3) output of the boundary image
                    feats.append(_ImageTapFn.apply(x.float(), pid, 3, self.ops))
                else:
                    feats.append(next(nat))
            out.append(feats)
        return out

    def tap_dims(self, e, H, W):
        """(rows, columns) of encoder layer e for an H x W input: the ReflectionPad2d(3) output, the k7 conv block, the two
        stride-2 blocks (k3 p1: ceil(n / 2)), the residual trunk"""
        if e == 0:
            return H + 6, W + 6
        if e < 4:
            return H, W
        if e < 7:
            return (H + 1) // 2, (W + 1) // 2
        return ((H + 1) // 2 + 1) // 2, ((W + 1) // 2 + 1) // 2

    def tap_extent(self, e, H, W):
        """number of pixels of encoder layer e for an H x W input"""
        h, w = self.tap_dims(e, H, W)
        return h * w
