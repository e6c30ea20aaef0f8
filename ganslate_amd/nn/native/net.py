"""Executor for conv networks (Resnet2D, PatchGAN2D, ...) on the HIP kernel library.

A network is a list of `Node`s (conv -> [InstanceNorm] -> activation [+ residual]). The executor owns
  * one flat fp32 master buffer (OTI-layout weights + biases of every layer), exposed as a single
    torch.nn.Parameter whose .grad is the flat fp32 gradient buffer the weight-gradient kernels accumulate
    into (this is also the bucket the data-parallel all-reduce works on),
  * the bf16 forward / data-gradient weight packs, refreshed from the master after every optimiser step,
  * hand-written forward and backward passes (no per-layer autograd): the whole network is ONE
    torch.autograd.Function, so recipe code keeps the reference's shape — `loss.backward()` drives it
    (ganslate/nn/gans/base.py:155-170) and images cross the boundary as NCHW fp32 tensors like the
    reference's `visuals` (cyclegan.py:39).

Reference semantics: layer order / hyper-parameters of ganslate/nn/generators/resnet/resnet2d.py:14-93 and
ganslate/nn/discriminators/patchgan/patchgan2d.py:17-66; weight init of ganslate/nn/utils.py:13-36.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .backend import get_ops
from .spec import ConvSpec, Lowered, lower
from .twin import Twin


import os
import weakref
# (GS_BWD_ORDER=0 — backward passes of one network unordered across streams — is gone: since the one-split weight-gradient
# launches add to dw with plain loads / stores (gs_wgrad_desc.dw_fresh note in ganslate_hip.h), overlapping them loses updates)
_BWD_ORDER = True
# GS_WGRAD_STREAM=1: networks a recipe marks (wgrad_side_stream) launch their weight-gradient kernels on a stream of their own.
# Nothing inside a backward pass waits for a weight gradient — the chain is apply -> data gradient -> apply -> ... — so the
# (compute-bound, long) weight-gradient launches can fill in beside the chain's memory-bound norm passes
_WGRAD_STREAM = os.environ.get("GS_WGRAD_STREAM", "0") == "1"
_wgrad_streams = {}


@dataclass
class Node:
    spec: ConvSpec
    norm: bool = True
    act: str = "relu"            # applied after the norm, or in the conv epilogue when norm is False
    slope: float = 0.2
    res: Optional[int] = None    # index of the node whose output is added after the norm (residual block tail)
    name: str = ""               # torch module path of the conv in the reference net, e.g. "model.1"
    aliases: Tuple[str, ...] = ()  # additional state_dict prefixes holding the same tensors
    attn: str = ""               # module path of a SelfAttentionBlock applied to this node's output (nn/attention.py);
                                 # its parameters are the Extras `attention_extras(attn, channels)`


@dataclass
class Extra:
    """A non-conv parameter vector kept in the flat master buffer after the conv layers (nn.PReLU slopes)."""
    name: str                    # state_dict key, e.g. "in_ab.relu.weight"
    size: int
    init: float = 0.25           # nn.PReLU default; ganslate's init_weights leaves it alone (nn/utils.py:13-36);
                                 # None: a conv weight drawn by init_weights (N(0, gain), nn/utils.py:19-20)
    aliases: Tuple[str, ...] = ()
    shape: Optional[Tuple[int, ...]] = None      # torch shape of the tensor in state dicts (default: the flat vector)


ATTN_KEYS = {"gamma": "gamma", "wq": "query_conv.weight", "bq": "query_conv.bias", "wk": "key_conv.weight",
             "bk": "key_conv.bias", "wv": "value_conv.weight", "bv": "value_conv.bias"}


def attention_extras(prefix: str, C: int, dims: int = 3) -> List["Extra"]:
    """parameters of a SelfAttentionBlock(C) (ganslate/nn/attention.py:16-22) as Extras, in the order torch yields them
    (gamma — a direct parameter of the block — before the three 1x1 convs' weights and biases)"""
    one = (1,) * dims
    d = C // 8
    return [Extra(f"{prefix}.gamma", 1, 0.0, shape=(1,)),
            Extra(f"{prefix}.query_conv.weight", d * C, None, shape=(d, C) + one),
            Extra(f"{prefix}.query_conv.bias", d, 0.0, shape=(d,)),
            Extra(f"{prefix}.key_conv.weight", d * C, None, shape=(d, C) + one),
            Extra(f"{prefix}.key_conv.bias", d, 0.0, shape=(d,)),
            Extra(f"{prefix}.value_conv.weight", C * C, None, shape=(C, C) + one),
            Extra(f"{prefix}.value_conv.bias", C, 0.0, shape=(C,))]


class _Saved:
    __slots__ = ("x_img", "acts", "ys", "mrs", "out_img", "lows", "N", "attn", "__weakref__")


class NativeNet:
    """Hand-written fwd/bwd of a conv network over an ops backend (HIP in production)."""

    def __init__(self, nodes: List[Node], in_channels: int, out_channels: int, out_act: str = "none", ops=None,
                 extras: Optional[List[Extra]] = None):
        self.ops = ops if ops is not None else get_ops()
        self.device = self.ops.device
        self.nodes = nodes
        self.in_channels, self.out_channels, self.out_act = in_channels, out_channels, out_act
        assert nodes[0].spec.cin == in_channels and nodes[-1].spec.cout == out_channels
        assert not nodes[-1].norm, "last node feeds the image boundary directly"
        self.dims = nodes[0].spec.dims          # 2 = NCHW images, 3 = NCDHW volumes
        for i, nd in enumerate(nodes):           # W-folded convs only exist at the image boundary (csrc/wfold.hip)
            assert nd.spec.wfold in ("", "in", "out") and (nd.spec.wfold != "in" or i == 0) and \
                (nd.spec.wfold != "out" or i == len(nodes) - 1), "W-fold: 'in' on the first, 'out' on the last layer"
        # ---- flat parameter layout: [w_0 | b_0 | w_1 | b_1 | ...] ------------------------------------------
        self.w_off, self.b_off, off = [], [], 0
        for nd in nodes:
            self.w_off.append(off); off += nd.spec.master_numel
            self.b_off.append(off); off += nd.spec.cout_p
        self.extras = list(extras or [])
        self._extra_by_name = {ex.name: ex for ex in self.extras}
        self.x_off = {}
        for ex in self.extras:
            self.x_off[ex.name] = off
            off += (ex.size + 7) // 8 * 8
        self.numel = off
        self.master = torch.nn.Parameter(torch.zeros(off, dtype=torch.float32, device=self.device))
        self.master.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.master._owner_net = self
        self._token = torch.zeros(1, requires_grad=True, device=self.device)
        self._low_cache: Dict[Tuple[int, int], list] = {}
        self._packs: Dict[Tuple[int, int], dict] = {}
        self._packs_dirty = True
        self.training = True
        # data-parallel state (set by parallelize())
        self._dist = None
        self._fw_pending = 0
        self._reduce_handles = []
        self._reduced_buckets = set()
        self.grad_dirty = False
        self._wgrad_seen = set()           # layers whose weight gradient has been written since the buffer was cleared
        self._deferred = {}          # node -> (wgrad desc, dense, gathered) held back for a merged launch
        self.multi_stream_passes = False   # set by a recipe that runs passes of this network on several streams
        self.external_reduce = False       # data-parallel gradients are reduced by the caller (BaseGAN graph runner)

    # ---- torch.nn.Module-like surface used by BaseGAN ----------------------------------------------------------
    def parameters(self):
        return [self.master]

    @property
    def requires_grad(self) -> bool:
        return self.master.requires_grad

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        return self

    def __call__(self, x):
        return self.forward(x)

    # ---- weights ------------------------------------------------------------------------------------------------------
    def init_weights(self, init_type="normal", gain=0.02):
        """ganslate/nn/utils.py:13-36 — N(0, gain) (or xavier/kaiming/orthogonal) conv weights, zero biases; drawn
        with torch's CPU RNG in layer order."""
        from torch.nn import init
        flat = torch.zeros(self.numel, dtype=torch.float32)
        for i, nd in enumerate(self.nodes):
            w = torch.empty(nd.spec.torch_weight_shape())
            if init_type == "normal":
                init.normal_(w, 0.0, gain)
            elif init_type == "xavier":
                init.xavier_normal_(w, gain=gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(w, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(w, gain=gain)
            else:
                raise NotImplementedError(f"initialization method `{init_type}` is not implemented")
            flat[self.w_off[i]:self.w_off[i] + nd.spec.master_numel] = nd.spec.master_from_torch(w).reshape(-1)
        for ex in self.extras:
            if ex.init is None:          # a conv weight kept as an Extra (SelfAttentionBlock's 1x1 convs): same rule as above
                w = torch.empty(ex.shape if ex.shape is not None else (ex.size,))
                if init_type == "normal":
                    init.normal_(w, 0.0, gain)
                elif init_type == "xavier":
                    init.xavier_normal_(w, gain=gain)
                elif init_type == "kaiming":
                    init.kaiming_normal_(w, a=0, mode="fan_in")
                elif init_type == "orthogonal":
                    init.orthogonal_(w, gain=gain)
                else:
                    raise NotImplementedError(f"initialization method `{init_type}` is not implemented")
                flat[self.x_off[ex.name]:self.x_off[ex.name] + ex.size] = w.reshape(-1)
            else:
                flat[self.x_off[ex.name]:self.x_off[ex.name] + ex.size] = ex.init
        with torch.no_grad():
            self.master.copy_(flat.to(self.device))
        self.mark_packs_dirty()

    def attn_tensors(self, prefix, grad=False):
        """the SelfAttentionBlock `prefix`'s parameters (or their gradients) as views of the flat buffer, keyed like
        ops.attn_forward wants them"""
        buf = self.master.grad if grad else self.master.detach()
        out = {}
        for k, suffix in ATTN_KEYS.items():
            ex = self._extra_by_name[f"{prefix}.{suffix}"]
            v = buf[self.x_off[ex.name]:self.x_off[ex.name] + ex.size]
            out[k] = v.view(ex.shape[0], -1) if suffix.endswith("conv.weight") else v
        return out

    def extra(self, name, grad=False):
        """view of an extra parameter vector (padded to 8) in the master / gradient buffer"""
        o = self.x_off[name]
        n = (next(e.size for e in self.extras if e.name == name) + 7) // 8 * 8
        return (self.master.grad if grad else self.master.detach())[o:o + n]

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """torch-layout tensors under the reference's module names (incl. aliases such as encoder.N == model.N)."""
        sd = {}
        m = self.master.detach()
        for i, nd in enumerate(self.nodes):
            w = nd.spec.torch_from_master(m[self.w_off[i]:self.w_off[i] + nd.spec.master_numel])
            b = m[self.b_off[i]:self.b_off[i] + nd.spec.cout].clone()
            for prefix in (nd.name,) + tuple(nd.aliases):
                sd[f"{prefix}.weight"] = w
                if nd.spec.bias:
                    sd[f"{prefix}.bias"] = b
        for ex in self.extras:
            v = m[self.x_off[ex.name]:self.x_off[ex.name] + ex.size].clone()
            if ex.shape is not None:
                v = v.view(ex.shape)
            for key in (ex.name,) + tuple(ex.aliases):
                sd[key] = v
        return sd

    def load_state_dict(self, sd, strict=True):
        flat = torch.zeros(self.numel, dtype=torch.float32)
        for i, nd in enumerate(self.nodes):
            w = sd[f"{nd.name}.weight"].detach().float().cpu()
            assert tuple(w.shape) == tuple(nd.spec.torch_weight_shape()), (nd.name, w.shape)
            flat[self.w_off[i]:self.w_off[i] + nd.spec.master_numel] = nd.spec.master_from_torch(w).reshape(-1)
            if nd.spec.bias and f"{nd.name}.bias" in sd:
                flat[self.b_off[i]:self.b_off[i] + nd.spec.cout] = sd[f"{nd.name}.bias"].detach().float().cpu()
        for ex in self.extras:
            flat[self.x_off[ex.name]:self.x_off[ex.name] + ex.size] = sd[ex.name].detach().float().cpu().reshape(-1)
        with torch.no_grad():
            self.master.copy_(flat.to(self.device))
        self.mark_packs_dirty()

    def grads_state_dict(self) -> Dict[str, torch.Tensor]:
        """gradients in torch layout (for parity tests / inspection)."""
        sd = {}
        g = self.master.grad
        for i, nd in enumerate(self.nodes):
            sd[f"{nd.name}.weight"] = nd.spec.torch_from_master(g[self.w_off[i]:self.w_off[i] + nd.spec.master_numel])
            sd[f"{nd.name}.bias"] = g[self.b_off[i]:self.b_off[i] + nd.spec.cout].clone()
        for ex in self.extras:
            v = g[self.x_off[ex.name]:self.x_off[ex.name] + ex.size].clone()
            sd[ex.name] = v.view(ex.shape) if ex.shape is not None else v
        return sd

    # ---- flat buffers <-> the reference's per-parameter tensors (optimizer state in checkpoints) ------------------
    def reference_parameter_order(self):
        """state_dict keys of this network's parameters in the order torch's Module.parameters() yields them in the
        reference network (index i of an `optimizer.state_dict()['state']` saved by the reference, base.py:244-245)"""
        keys = []
        for nd in self.nodes:
            keys.append(f"{nd.name}.weight")
            if nd.spec.bias:
                keys.append(f"{nd.name}.bias")
        return keys + [ex.name for ex in self.extras]

    def flat_to_tensors(self, flat) -> Dict[str, torch.Tensor]:
        """a flat fp32 buffer laid out like the master (Adam moments, gradients) as torch-layout tensors by key"""
        out = {}
        for i, nd in enumerate(self.nodes):
            out[f"{nd.name}.weight"] = nd.spec.torch_from_master(flat[self.w_off[i]:self.w_off[i] + nd.spec.master_numel])
            if nd.spec.bias:
                out[f"{nd.name}.bias"] = flat[self.b_off[i]:self.b_off[i] + nd.spec.cout].clone()
        for ex in self.extras:
            v = flat[self.x_off[ex.name]:self.x_off[ex.name] + ex.size].clone()
            out[ex.name] = v.view(ex.shape) if ex.shape is not None else v
        return out

    def tensors_to_flat(self, tensors, flat):
        """inverse of flat_to_tensors: writes torch-layout tensors into a flat buffer (padding stays zero)"""
        host = torch.zeros(self.numel, dtype=torch.float32)
        for i, nd in enumerate(self.nodes):
            w = tensors[f"{nd.name}.weight"].detach().float().cpu()
            host[self.w_off[i]:self.w_off[i] + nd.spec.master_numel] = nd.spec.master_from_torch(w).reshape(-1)
            if nd.spec.bias:
                host[self.b_off[i]:self.b_off[i] + nd.spec.cout] = tensors[f"{nd.name}.bias"].detach().float().cpu()
        for ex in self.extras:
            host[self.x_off[ex.name]:self.x_off[ex.name] + ex.size] = tensors[ex.name].detach().float().cpu().reshape(-1)
        with torch.no_grad():
            flat.copy_(host.to(flat.device))

    def mark_packs_dirty(self, ident_fresh=None, tr_fresh=()):
        """the master moved. ident_fresh: the pack set whose row-major groups the optimiser wrote with the update itself
        (fused_pack_targets) — that set only needs its transposed segments refreshed; tr_fresh: (layer, 'f' | 'd') pairs whose
        transposed segments were written too (gs_wgrad_adam with gs_adam_fuse.tr_*): the refresh leaves them out"""
        self._packs_dirty = True
        for pk in self._packs.values():
            pk["ident_fresh"] = pk is ident_fresh
            pk["tr_fresh"] = frozenset(tr_fresh) if pk is ident_fresh else frozenset()
            if pk["tr_fresh"]:      # the reduced refresh plans now (host work, a download): the refresh itself may be captured
                for which in ("f", "d"):
                    skip = frozenset(i for i, w in pk["tr_fresh"] if w == which)
                    if skip:
                        self._tiled_plan_without(pk, which, skip)
        self._recent_passes = {}          # (the weights moved: recorded activations no longer describe this network)

    def transposed_tables(self, i):
        """(which, base int32[T], kp int32[T], pack) of layer i's TRANSPOSED pack — pack element of W[p][t][q] = base[t] + q kp[t] +
        p, base[t] < 0 for a tap in no class — or None: derived from the layer's gather table and verified element by element
        (a layer whose table does not have this form keeps the refresh from the master). One pack set only."""
        if len(self._packs) != 1:
            return None
        (key, pk), = self._packs.items()
        cache = pk.setdefault("tr_tables", {})
        if i in cache:
            return cache[i]
        lw, sp = self._lowered(*key)[i], self.nodes[i].spec
        n = sp.master_numel
        wg = lw.wgrad
        P, T, Q = wg.P, wg.T, wg.Q
        out = None
        if P * T * Q == n and P > 1 and Q > 1:
            for which, index, off in (("f", lw.fwd_index, pk["f_off"][i]), ("d", lw.dgrad_index, pk["d_off"][i])):
                idx = np.asarray(index).reshape(-1)
                valid = np.nonzero(idx >= 0)[0]
                inv = np.full(n, -1, np.int64)
                inv[idx[valid]] = valid
                pos = inv.reshape(P, T, Q)
                base, kp = pos[0, :, 0].copy(), pos[0, :, 1] - pos[0, :, 0]
                live = base >= 0
                if not live.any() or (pos[1, live, 0] - base[live] != 1).any():
                    continue                                     # (row-major pack: consecutive along q, the optimiser writes it)
                kp = np.where(live, kp, 0)
                want = base[None, :, None] + np.arange(Q)[None, None, :] * kp[None, :, None] + np.arange(P)[:, None, None]
                ok = (np.array_equal(pos[:, live, :], want[:, live, :]) and (pos[:, ~live, :] < 0).all()
                      and valid.size == int(live.sum()) * P * Q and not ((base[live] + off) % 8).any() and not (kp[live] % 8).any()
                      and (kp[live] > 0).all())
                if ok:
                    b = np.where(live, base + off, -1).astype(np.int32)
                    out = (which, torch.from_numpy(b).to(self.device), torch.from_numpy(kp.astype(np.int32)).to(self.device),
                           pk[which + "pack"])
                break
        cache[i] = out
        return out

    def _tiled_plan_without(self, pk, which, skip):
        """(seg, tiles) of pack `which` without the transposed segments of the layers in `skip`"""
        cache = pk.setdefault("seg_without", {})
        ck = (which, skip)
        if ck not in cache:
            plan, offs = pk[which + "_plan"], pk[which + "_off"]
            total = pk[which + "pack"].numel()
            rng = [(offs[i], offs[i + 1] if i + 1 < len(offs) else total) for i in sorted(skip)]
            seg = plan["seg"].cpu().numpy()
            keep = [r for r in seg if not any(a <= r[0] < b for a, b in rng)]
            tiles, rows = 0, []
            for r in keep:
                rows.append((r[0], r[1], r[2], r[3], tiles))
                tiles += (int(r[2]) + 63) // 64 * (int(r[3]) // 64)
            cache[ck] = (torch.from_numpy(np.asarray(rows, np.int64).reshape(-1, 5)).to(self.device), tiles)
        return cache[ck]

    def fused_pack_targets(self):
        """(pack set, (inv_f, fpack, inv_d, dpack)) for gs_adam_step_dev_packs, or None: inv_x[i] = the group of 8 pack
        elements that IS master elements 8 i .. 8 i + 7 (-1: none). Built once per pack set; only for a network with ONE pack
        set (one input size) — with several, every set refreshes from the master as before. GS_ADAM_PACKS=0 switches it off."""
        if os.environ.get("GS_ADAM_PACKS", "1") == "0" or len(self._packs) != 1 or self.master.data_ptr() % 16:
            return None
        (pk,) = self._packs.values()
        if "fused" not in pk:
            n8 = (self.numel + 7) // 8
            out, any_fused = [], False
            for which in ("f", "d"):
                plan = pk[which + "_plan"]
                groups = plan["groups"].cpu().numpy()
                inv = np.full(n8, -1, np.int32)
                j = np.nonzero((groups >= 0) & (groups % 8 == 0) & (groups + 8 <= self.numel // 4 * 4))[0]
                i = groups[j] // 8
                first = np.unique(i, return_index=True)[1]          # (a master group feeds at most one pack group here)
                j, i = j[first], i[first]
                inv[i] = j
                rest = groups.copy()
                rest[j] = -3                                          # the leftover launch skips them
                left = int(((rest >= 0) | (rest == -2)).sum())
                any_fused = any_fused or len(j) > 0
                out.append((torch.from_numpy(inv).to(self.device) if len(j) else None,
                            torch.from_numpy(rest).to(self.device) if left else None, left))
                if pk[which + "pack"].data_ptr() % 16:
                    any_fused = False
            pk["fused"] = out if any_fused else None
        if pk["fused"] is None:
            return None
        (inv_f, _, _), (inv_d, _, _) = pk["fused"]
        return pk, (inv_f, pk["fpack"] if inv_f is not None else None, inv_d, pk["dpack"] if inv_d is not None else None)

    # A recorded pass keeps its activations until its backward pass has run. A recipe that needs DETACHED features of the
    # same input again (CUT's source patches: the reference runs the encoder a second time on real_A / real_B,
    # cut.py:205-211, 297-312) can read them out of that pass instead of launching it again: same weights, same input,
    # same kernels — the same bits.
    def _remember_pass(self, x, saved, n0=0):
        if not hasattr(self, "_recent_passes"):
            self._recent_passes = {}
        if len(self._recent_passes) >= 4:
            self._recent_passes.pop(next(iter(self._recent_passes)))
        # (a weak reference: the autograd node owns the pass; remembering it must not keep its activations alive past the
        # backward pass — the V-Net executors do not clear theirs, and use_memory_saving exists to hold less, not more)
        self._recent_passes[(x.data_ptr(), tuple(x.shape))] = (x._version, weakref.ref(saved), n0)

    def recorded_pass(self, x):
        """(saved state, first image) of a recorded full pass that contained exactly this tensor (same storage, shape and
        version) and whose activations are still alive, or None"""
        ent = getattr(self, "_recent_passes", {}).get((x.data_ptr(), tuple(x.shape)))
        saved = ent[1]() if ent is not None else None
        acts = getattr(saved, "acts", None)
        if saved is None or ent[0] != x._version or acts is None or any(a is None for a in acts):
            return None
        return saved, ent[2]

    def refresh_packs(self, x):
        """bring the bf16 packs for inputs shaped like x up to date now (on the current stream), so that passes launched
        on several streams afterwards only read them"""
        self._get_packs(*tuple(x.shape[2:]))

    # Passes of one network that run on different streams (the two cycles of a CycleGAN step) accumulate into the same
    # gradient buffer and hand operands to each other for merged weight-gradient launches: a backward pass starts after
    # the previous backward pass of the SAME network has finished, whatever stream that ran on. Only networks a recipe
    # marks (multi_stream_passes) pay for the events.
    def _order_backward_begin(self):
        if self.device.type == "cuda" and _BWD_ORDER and self.multi_stream_passes:
            ev = getattr(self, "_bwd_done", None)
            # An event from before a stream capture began (or from inside one that has ended) orders nothing here: the
            # capture's own begin / end already does. A pass on the stream the previous one ran on is ordered by the
            # stream itself (and a captured side stream waiting on its own event crashes hipStreamEndCapture, ROCm 7.2).
            if ev is not None and ev[1] == torch.cuda.is_current_stream_capturing() and \
                    ev[2] != torch.cuda.current_stream():
                torch.cuda.current_stream().wait_event(ev[0])

    def _order_backward_end(self):
        if self.device.type == "cuda" and _BWD_ORDER and self.multi_stream_passes:
            from ...utils.streams import new_event
            ev = new_event()
            ev.record()
            self._bwd_done = (ev, torch.cuda.is_current_stream_capturing(), torch.cuda.current_stream())

    # ---- lowering / packs (per input size) -----------------------------------------------------------------------
    def _lowered(self, *sizes) -> List[Lowered]:
        key = tuple(sizes)
        if key not in self._low_cache:
            lows, cur = [], key
            for nd in self.nodes:
                lw = lower(nd.spec, *cur)
                lows.append(lw)
                cur = lw.out_dims
            self._low_cache[key] = lows
        return self._low_cache[key]

    @staticmethod
    def _repack_plan(lows, idx, offs, which):
        """How a network's pack is refreshed from the master: two launches, both with ONE index per 8 pack elements (the
        element-wise refresh reads a 4-byte index per 2-byte element: 40 % of its bytes).
          groups  int32 [pack elements / 8], along k over the whole pack: base index of 8 consecutive master elements;
                  -1 = padding (zeros), -2 = an irregular group (reads its own entries of the element-wise table; does not
                  occur in the reference's layer types), -3 = part of a transposed segment;
          seg     int64 [nseg, 5] = (pack offset, offset into tgroups, rows, Kp, first tile) of every class segment whose
                  master indices are consecutive along the ROWS (the transposed packs: data-gradient pack of a conv, forward
                  pack of a transposed conv), all refreshed by one tiled launch; tgroups int32 = their [rows / 8][Kp] bases.
        Returns dict(groups, seg, tgroups, tiles, need_index)."""
        total = sum(x.size for x in idx)
        assert total % 8 == 0
        allidx = np.concatenate([x.reshape(-1) for x in idx]) if idx else np.zeros(0, np.int64)
        t = allidx.reshape(-1, 8)
        groups = t[:, 0].copy()
        want = np.where(groups[:, None] >= 0, groups[:, None] + np.arange(8)[None, :], -1)
        groups[(t != want).any(1)] = -2
        seg, tg, tiles, goff = [], [], 0, 0
        for i, lw in enumerate(lows):
            for g in getattr(lw, which):
                start, n = offs[i] + g.pack_offset, g.w_rows * g.Kp
                if g.w_rows % 8 or start % 8 or not (groups[start // 8:(start + n) // 8] == -2).any():
                    continue
                tt = allidx[start:start + n].reshape(g.w_rows // 8, 8, g.Kp)
                base = tt[:, 0, :]
                if not np.array_equal(tt, np.where(base[:, None, :] >= 0, base[:, None, :] + np.arange(8)[None, :, None], -1)):
                    continue
                seg.append((start, goff, g.w_rows, g.Kp, tiles))
                tg.append(base.reshape(-1))
                goff += base.size
                tiles += (g.w_rows + 63) // 64 * (g.Kp // 64)
                groups[start // 8:(start + n) // 8] = -3
        return {"groups": groups.astype(np.int32), "seg": np.asarray(seg, np.int64).reshape(-1, 5),
                "tgroups": np.concatenate(tg).astype(np.int32) if tg else np.zeros(0, np.int32), "tiles": tiles,
                "need_index": bool((groups == -2).any())}

    def _get_packs(self, *sizes):
        """bf16 packs are size-independent except for the parity-class split, which only depends on the spec;
        one pack set per input-size key keeps the bookkeeping trivial (a net sees one or two sizes in practice)."""
        key = tuple(sizes)
        pk = self._packs.get(key)
        if pk is None:
            lows = self._lowered(*sizes)
            f_idx, d_idx, f_off, d_off = [], [], [], []
            fo = do = 0
            for i, lw in enumerate(lows):
                fi = lw.fwd_index.astype(np.int64); fi[fi >= 0] += self.w_off[i]
                di = lw.dgrad_index.astype(np.int64); di[di >= 0] += self.w_off[i]
                f_idx.append(fi); d_idx.append(di); f_off.append(fo); d_off.append(do)
                fo += fi.size; do += di.size
            def to_dev(plan, idx):
                d = {k: torch.from_numpy(plan[k]).to(self.device) for k in ("groups", "seg", "tgroups")}
                d["tiles"] = plan["tiles"]
                # the element-wise table is only uploaded when some group needs it (4 bytes per weight otherwise, twice)
                d["index"] = (torch.from_numpy(np.concatenate(idx).astype(np.int32)).to(self.device)
                              if plan["need_index"] else None)
                return d
            pk = {
                "f_plan": to_dev(self._repack_plan(lows, f_idx, f_off, "fwd"), f_idx),
                "d_plan": to_dev(self._repack_plan(lows, d_idx, d_off, "dgrad"), d_idx),
                "f_off": f_off, "d_off": d_off,
                # + 64 elements of slack so the last row's padded K-steps stay inside the allocation
                "fpack": torch.zeros(fo + 64, dtype=self.ops.act_dtype, device=self.device),
                "dpack": torch.zeros(do + 64, dtype=self.ops.act_dtype, device=self.device),
                "fresh": False,
            }
            self._packs[key] = pk
        if self._packs_dirty:
            for p in self._packs.values():
                p["fresh"] = False
            self._packs_dirty = False
        if not pk["fresh"]:
            m = self.master.detach()
            ident = pk.get("ident_fresh") and pk.get("fused")         # row-major groups written by the optimiser's launch
            for k, which in enumerate(("f", "d")):
                plan, pack = pk[which + "_plan"], pk[which + "pack"]
                n8 = plan["groups"].numel()
                if ident and ident[k][0] is not None:
                    if ident[k][2]:                                   # groups the fused launch could not take
                        self.ops.repack_groups(m, ident[k][1], pack[:n8 * 8], plan["index"])
                else:
                    self.ops.repack_groups(m, plan["groups"], pack[:n8 * 8], plan["index"])
                skip = frozenset(i for i, w in (pk.get("tr_fresh") or ()) if w == which) if ident else frozenset()
                if skip:                                              # their fused launches wrote the transposed groups too
                    seg, tiles = self._tiled_plan_without(pk, which, skip)
                    if tiles:
                        self.ops.repack_tiled_groups(m, plan["tgroups"], pack, seg, tiles)
                elif plan["tiles"]:
                    self.ops.repack_tiled_groups(m, plan["tgroups"], pack, plan["seg"], plan["tiles"])
            pk["fresh"], pk["ident_fresh"], pk["tr_fresh"] = True, False, frozenset()
        return pk

    # ---- forward ----------------------------------------------------------------------------------------------------------
    def forward(self, x):
        assert x.dim() == 2 + self.dims and x.shape[1] == self.in_channels, \
            f"expected N x {self.in_channels} x {'D x ' if self.dims == 3 else ''}H x W"
        x = x.contiguous().float()
        record = torch.is_grad_enabled() and (x.requires_grad or self.requires_grad)
        if not record:
            out, _ = self._forward(x.detach(), save=False)
            return out
        return _NetFn.apply(x, self._token, self)

    def forward_parts(self, xs):
        """several batches through this network as ONE pass (they follow each other in the batch; per-sample InstanceNorm:
        every image's output is what its own pass gives) -> one output per batch. CUT's G(real_A) and G(real_B),
        cut.py:160-166."""
        xs = tuple(t.contiguous().float() for t in xs)
        assert all(t.shape[1:] == xs[0].shape[1:] for t in xs), "forward_parts: batches of one image shape"
        record = torch.is_grad_enabled() and (self.requires_grad or any(t.requires_grad for t in xs))
        if not record:
            out, _ = self._forward(tuple(t.detach() for t in xs), save=False)
            return _split_parts(out, xs)
        return _PartsFn.apply(self._token, self, *xs)

    def forward_parts_cat(self, pairs):
        """forward_parts for batches that are each torch.cat([a, b], dim=1) of two images (the conditional discriminator:
        D(cat(real_A, fake_B)), pix2pix.py:70,80): the images are converted side by side into the first activation, no
        concatenated tensor exists -> one output per batch"""
        pairs = [(a.contiguous().float(), b.contiguous().float()) for a, b in pairs]
        flat = [t for pair in pairs for t in pair]
        record = torch.is_grad_enabled() and (self.requires_grad or any(t.requires_grad for t in flat))
        if not record:
            out, _ = self._forward(tuple(ChannelCat((a.detach(), b.detach())) for a, b in pairs), save=False)
            return _split_parts(out, pairs_shape(pairs))
        return _CatPartsFn.apply(self._token, self, *flat)

    def forward_taps(self, x, taps, ids):
        """sampled features of intermediate nodes: taps = [("x"|"y", node)], ids = [LongTensor[P]] (pixel indices);
        returns a tuple of [N, P, C] fp32 tensors that autograd can differentiate into the network"""
        x = x.contiguous().float()
        return _TapFn.apply(self._token, self, tuple(taps), (tuple(ids),), x)

    def forward_taps_parts(self, xs, taps, ids_per_part):
        """forward_taps for several batches in ONE encoder pass, each with its own pixel ids -> per batch a tuple of
        [N_p, P, C] tensors"""
        xs = tuple(t.contiguous().float() for t in xs)
        outs = _TapFn.apply(self._token, self, tuple(taps), tuple(tuple(i) for i in ids_per_part), *xs)
        k = len(taps)
        return [outs[p * k:(p + 1) * k] for p in range(len(xs))]

    def _forward(self, x, save, stop=None, tw=None):
        """stop = index of the last node to run (encoder-only passes of CUT, cut.py:297-312); None = whole net.
        tw = a second network of identical architecture (nn/native/twin.py): x is then the pair (xa, xb) and the pass runs
        as ONE batch of 2N images, images [0, N) through this network's weights and [N, 2N) through tw's."""
        ops, dev = self.ops, self.device
        # twin: x = (images of this network, images of tw), each a tuple of tensors that follow each other in the batch
        # (one network: a tensor, or a tuple of batches that run as one pass — forward_parts)
        xs = tuple(x[0]) + tuple(x[1]) if tw is not None else (tuple(x) if isinstance(x, (tuple, list)) else (x,))
        sizes = tuple(xs[0].shape[2:])
        N = sum(t.shape[0] for t in xs)
        if tw is not None:
            assert 2 * sum(t.shape[0] for t in x[0]) == N, "twin pass: both networks take the same number of images"
        lows = self._lowered(*sizes)
        pk = self._get_packs(*sizes)
        m = self.master.detach()
        fpack_all = pk["fpack"]
        if tw is not None:
            assert stop is None, "twin passes run whole networks"
            m = Twin(m, tw.master.detach())
            fpack_all = Twin(fpack_all, tw._get_packs(*sizes)["fpack"])
        sp0 = self.nodes[0].spec
        a = torch.empty(N, *sizes, sp0.cin_p, dtype=self.ops.act_dtype, device=dev)
        n0 = 0
        for xh in xs:
            ah = a[n0:n0 + xh.shape[0]]
            n0 += xh.shape[0]
            if isinstance(xh, ChannelCat):
                assert sp0.wfold != "in", "channel pairs enter through the plain image conversion"
                ops.image_pair_to_act(xh[0], xh[1], ah)
            elif sp0.wfold == "in":      # the W taps of the stem become channels while the image is converted
                ops.image_unfold(xh, ah, sp0.k, sp0.pad, sp0.pad_mode)
            else:
                ops.image_to_act(xh, ah)
        acts, ys, mrs, attn_saved = [a], [], [], {}
        for i, (nd, lw) in enumerate(zip(self.nodes, lows)):
            if stop is not None and i > stop:
                break
            if i > 0 and self.nodes[i - 1].attn:       # SelfAttentionBlock on the previous node's output
                xo, sv = ops.attn_forward(acts[-1], self.attn_tensors(self.nodes[i - 1].attn), need_backward=bool(save))
                attn_saved[i - 1] = sv if save else None
                acts[-1] = xo
            sp = nd.spec
            bias = m[self.b_off[i]:self.b_off[i] + sp.cout_p]
            fpack = fpack_all[pk["f_off"][i]:]
            y = torch.empty(N, *lw.out_dims, sp.cout_p, dtype=self.ops.act_dtype, device=dev)
            if nd.norm:
                slots, offs = 0, []
                for g in lw.fwd:
                    offs.append(slots)
                    slots += ops.stat_slots(g, N, twin=tw is not None, multi=lw.fwd if len(lw.fwd) > 1 else None)
                part = torch.empty(N * slots * 2 * sp.cout_p, dtype=torch.float32, device=dev)
                ops.gconv_classes(lw.fwd, acts[-1], fpack, bias, y, stats=part, stats_slots=slots, stats_slot0s=offs)
                mr = torch.empty(N * 2 * sp.cout_p, dtype=torch.float32, device=dev)
                xo = torch.empty_like(y)
                res = acts[nd.res + 1] if nd.res is not None else None
                # statistics finalised inside the apply launch (no separate slot-sum kernel)
                ops.inorm_stats_act_forward(y, part, slots, mr, res, xo, act=nd.act, slope=nd.slope)
                ys.append(y if save else None); mrs.append(mr if save else None)
                acts.append(xo)
            else:
                assert nd.res is None
                wf_out = sp.wfold == "out"     # bias and activation move behind the shift-add
                ops.gconv_classes(lw.fwd, acts[-1], fpack, None if wf_out else bias, y,
                                  act="none" if wf_out else nd.act, slope=nd.slope)
                ys.append(None); mrs.append(None)
                acts.append(y)
        out = None
        if stop is None:
            lw, spl = lows[-1], self.nodes[-1].spec
            if spl.wfold == "out":
                out = torch.empty(N, self.out_channels, *spl.out_hw(*lw.in_dims), dtype=torch.float32, device=dev)
                ops.shiftadd_to_image(acts[-1], m[self.b_off[-1]:self.b_off[-1] + spl.cout_p], out, spl.k,
                                      act=self.out_act)
            else:
                out = torch.empty(N, self.out_channels, *lw.out_dims, dtype=torch.float32, device=dev)
                ops.act_to_image(acts[-1], out, act=self.out_act)
        if not save:
            return out, None
        s = _Saved()
        s.x_img, s.acts, s.ys, s.mrs, s.out_img, s.lows, s.N = (x if torch.is_tensor(x) else xs), acts, ys, mrs, out, lows, N
        s.attn = attn_saved
        return out, s

    # ---- backward ---------------------------------------------------------------------------------------------------------
    def _backward(self, s: _Saved, g_img, need_input_grad: bool, want_w: bool, start=None, inj_x=None, inj_y=None,
                  tw=None):
        """g_img = gradient of the output image, or None for a partial pass that starts at node `start` and is driven
        only by injected gradients: inj_x[i] / inj_y[i] = sparse gradients [(first image, images, pixel ids [P], g [images,
        P, c])] w.r.t. the output / raw conv output of node i (feature taps of CUT's PatchNCE loss).
        tw: the pass recorded by _forward(..., tw) — g_img holds one gradient per input part (None = that part's output took
        no part in the loss), the gradients of images [N/2, N) go to tw's flat gradient buffer, and the input gradients come
        back as a tuple, one per part."""
        inj_x, inj_y = inj_x or {}, inj_y or {}
        ops, dev = self.ops, self.device
        nodes, lows, N = self.nodes, s.lows, s.N
        parts = not torch.is_tensor(s.x_img)        # the pass took several batches (twin pass / forward_parts)
        x_imgs = s.x_img if parts else (s.x_img,)
        Nh = N // 2 if tw is not None else N        # images per network
        for net in (self, tw):
            if net is not None and net.master.grad is None:
                net.master.grad = torch.zeros(net.numel, dtype=torch.float32, device=dev)
        pk = self._get_packs(*x_imgs[0].shape[2:])
        grad, dpack_all = self.master.grad, pk["dpack"]
        if tw is not None:
            grad = Twin(grad, tw.master.grad)
            dpack_all = Twin(dpack_all, tw._get_packs(*x_imgs[0].shape[2:])["dpack"])
        last = len(nodes) - 1 if start is None else start
        # gradient w.r.t. the output of node i: (tensor on padded domain, fold, extra, pad mode of the fold)
        pending = None
        if g_img is not None:
            ga = torch.empty_like(s.acts[-1])
            n0 = 0
            for gh, xh in zip(g_img if parts else (g_img,), x_imgs):     # one gradient per input part
                gah, outh = ga[n0:n0 + xh.shape[0]], s.out_img[n0:n0 + xh.shape[0]]
                n0 += xh.shape[0]
                if gh is None:           # this part's output took no part in the loss
                    gah.zero_()
                elif nodes[-1].spec.wfold == "out":
                    ops.shiftadd_to_image_backward(gh.contiguous().float(), outh, gah, nodes[-1].spec.k, act=self.out_act)
                else:
                    ops.act_to_image_backward(gh.contiguous().float(), outh, gah, act=self.out_act)
            pending = (ga, 0, None, "reflect")
        skip: Dict[int, torch.Tensor] = {}
        db_items = []        # bias gradients of the convs in front of norms: one batched launch at the end of the pass
        final_pass = want_w and self._dist is not None and self._fw_pending == 0 and not self.external_reduce
        # weight gradients on their own stream (see _WGRAD_STREAM): not when this pass reduces buckets as it goes
        wst = None
        if _WGRAD_STREAM and want_w and dev.type == "cuda" and getattr(self, "wgrad_side_stream", False) and not final_pass:
            wst = _wgrad_streams.get(dev.index)
            if wst is None:
                wst = _wgrad_streams[dev.index] = torch.cuda.Stream(device=dev)
        wst_used = False

        def on_wgrad_stream(fn, *tensors):
            """fn() behind everything launched so far, on the weight-gradient stream; `tensors` are its operands"""
            nonlocal wst_used
            if wst is None:
                return fn()
            from ...utils.streams import new_event
            ev = new_event()
            ev.record()
            wst.wait_event(ev)
            for t in tensors:
                if t is not None:
                    t.record_stream(wst)
            wst_used = True
            with torch.cuda.stream(wst):
                return fn()
        # another recorded forward of this net still awaits its backward (G_AB(real_A) and G_AB(fake_A) in one step):
        # hold the weight gradients of mergeable layers back and issue both passes as one launch then
        more_passes = want_w and self._fw_pending > 0 and start is None
        for i in range(last, -1, -1):
            nd, lw, sp = nodes[i], lows[i], nodes[i].spec
            if i in inj_x:            # tapped feature gradients join the gradient of this node's output (the pad adjoint is
                if pending is None:   # linear: they are added on the padded domain at their own pixels)
                    pending = (ops.zeros_like_act(s.acts[i + 1]), 0, None, "reflect")
                assert len(lw.out_dims) == 2, "feature taps: 2-D networks"
                for n0, np_, pid, g in inj_x[i]:
                    ops.tap_scatter_add(pending[0][n0:n0 + np_], pid, g, lw.out_dims[-1], f0=pending[1])
            if nd.attn:
                # the gradient arrives w.r.t. the block's output; the block's backward turns it into the gradient w.r.t. this
                # node's own output (and adds the block's parameter gradients)
                g_att, f_att = pending[0], pending[1]
                assert f_att == 0 and pending[2] is None and len(pending) == 4, "attention output: plain gradient expected"
                gx_att = ops.attn_backward(s.attn[i], g_att, self.attn_tensors(nd.attn),
                                           self.attn_tensors(nd.attn, grad=True) if want_w else None)
                pending = (gx_att, 0, None, "reflect")
                if want_w:
                    self.grad_dirty = True
            g_pad, fold, g2, fmode = pending[:4]
            pre = pending[4] if len(pending) > 4 else None
            applied = pending[5] if len(pending) > 5 else None
            x_out = s.acts[i + 1]
            # ---- gradient w.r.t. the conv output y ---------------------------------------------------------------
            need_total = nd.res is not None
            if applied is not None:
                # the data-gradient launch of the layer above already ran this norm's backward (gs_gconv_ring_apply): dy, the
                # total gradient for the skip path and the per-image totals for the bias gradient are there
                dy, total = applied["dy"], applied["total"]
                if want_w and sp.bias:
                    db = grad[self.b_off[i]:self.b_off[i] + sp.cout_p]
                    holder, off = pre[1], pre[1].numel() - N * 3 * sp.cout_p
                    if tw is None:
                        items = [(holder, off, s.mrs[i], db, N, sp.cout_p, lw.out_pixels)]
                    else:
                        hm = s.mrs[i].numel() // 2
                        items = [(holder, off + h * Nh * 3 * sp.cout_p, s.mrs[i][h * hm:(h + 1) * hm], db.half(h), Nh,
                                  sp.cout_p, lw.out_pixels) for h in (0, 1)]
                    if final_pass:
                        ops.norm_bias_grads(items)
                    else:
                        db_items += items
            elif nd.norm or nd.act != "none" or fold > 0 or g2 is not None:
                dy = torch.empty_like(x_out)
                gsum = torch.empty_like(x_out) if (need_total and (fold > 0 or g2 is not None)) else None
                if nd.norm:
                    # the bias gradient of a conv in front of an InstanceNorm comes out of the norm's reduction sums
                    db = grad[self.b_off[i]:self.b_off[i] + sp.cout_p] if (want_w and sp.bias) else None
                    # (data parallel, bucketed: the bucket holding db is all-reduced as soon as this layer is done -> inline)
                    inline_db = final_pass and tw is None
                    sums = ops.inorm_act_backward(g_pad, g2, s.ys[i], s.mrs[i], dy, gsum, fold=fold, fold_mode=fmode,
                                                  act=nd.act, slope=nd.slope, bias_grad=db if inline_db else None,
                                                  pre=pre)
                    if db is not None and not inline_db:
                        if tw is None:
                            items = [(sums[0], sums[1], s.mrs[i], db, N, sp.cout_p, lw.out_pixels)]
                        else:        # the per-image totals / statistics of the two halves lie behind each other
                            hm = s.mrs[i].numel() // 2
                            items = [(sums[0], sums[1] + h * Nh * 3 * sp.cout_p, s.mrs[i][h * hm:(h + 1) * hm],
                                      db.half(h), Nh, sp.cout_p, lw.out_pixels) for h in (0, 1)]
                        if final_pass:       # (bucketed reduction: this layer's bucket may be reduced right behind it)
                            ops.norm_bias_grads(items)
                        else:
                            db_items += items
                else:
                    ops.inorm_act_backward(g_pad, g2, x_out, None, dy, gsum, fold=fold, fold_mode=fmode, act=nd.act,
                                           slope=nd.slope)
                total = gsum if gsum is not None else g_pad
            else:
                dy, total = g_pad, g_pad
            if need_total:
                skip[nd.res] = total
            if i in inj_y:
                assert tw is None, "feature taps: single-network passes"
                if dy is g_pad or dy is total:      # (shared with the skip path: the taps go into a copy)
                    dy = dy.clone()
                for n0, np_, pid, g in inj_y[i]:
                    ops.tap_scatter_add(dy[n0:n0 + np_], pid, g, lw.out_dims[-1])
                    if want_w and sp.bias and nd.norm:
                        # a feature tapped from the RAW conv output (CUT's nce layer 4, cut.py:297-312) sees the bias: the
                        # norm's reduction sums above give the (zero) bias gradient of the path through the norm only
                        ops.tap_rows_sum(g, grad[self.b_off[i]:self.b_off[i] + sp.cout])
            # ---- parameter gradients ---------------------------------------------------------------------------------
            if want_w:
                dw = grad[self.w_off[i]:self.w_off[i] + sp.master_numel]
                a_t, g_t = (dy, s.acts[i]) if sp.kind == "conv" else (s.acts[i], dy)
                held = self._deferred.pop(i, None)
                if held is not None and (held[0] is not lw.wgrad or held[1].shape != a_t.shape or held[3] is not tw):
                    self._flush_held(i, held)     # other input size / other twin partner: cannot share a launch
                    held = None
                if held is not None:
                    if dev.type == "cuda":      # the other pass may have run (and allocated) on another stream
                        held[1].record_stream(torch.cuda.current_stream())
                        held[2].record_stream(torch.cuda.current_stream())
                    self._wgrad_written(i, tw)
                    on_wgrad_stream(lambda: ops.wgrad(lw.wgrad, a_t, g_t, dw, pair=(held[1], held[2])),
                                    a_t, g_t, held[1], held[2])
                elif more_passes and ops.can_merge_wgrad(lw.wgrad):
                    self._deferred[i] = (lw.wgrad, a_t, g_t, tw)      # (noted as written when it is launched)
                else:
                    fresh = self._wgrad_written(i, tw)
                    on_wgrad_stream(lambda: ops.wgrad(lw.wgrad, a_t, g_t, dw, fresh=fresh), a_t, g_t)
                if sp.bias and not nd.norm:
                    for h in range(N // Nh):
                        gh_ = grad.half(h) if tw is not None else grad
                        dyh = dy[h * Nh:(h + 1) * Nh]
                        if sp.wfold == "out":   # channels [0, cout) of dy are the dw = 0 slice = the plain output gradient
                            ops.bias_grad(dyh, sp.cout, gh_[self.b_off[i]:self.b_off[i] + sp.cout])
                        else:
                            ops.bias_grad(dyh, sp.cout_p, gh_[self.b_off[i]:self.b_off[i] + sp.cout_p])
                self.grad_dirty = True
                if tw is not None:
                    tw.grad_dirty = True
                if final_pass:
                    self._maybe_reduce_bucket(i)
                    if tw is not None:
                        tw._maybe_reduce_bucket(i)
            # ---- data gradient ------------------------------------------------------------------------------------------
            if i > 0 or need_input_grad:
                f = lw.dgrad_fold
                dpack = dpack_all[pk["d_off"][i]:]
                g2n = skip.pop(i - 1, None)
                fmode_n = sp.pad_mode if f else "reflect"
                # the reduction pass of the previous layer's InstanceNorm backward rides in this launch's epilogue
                # (wide stride-1 layers; not when a tapped feature gradient is still to be added to gx)
                plan = None
                if i > 0 and nodes[i - 1].norm and len(lw.dgrad) == 1 and (i - 1) not in inj_x and not nodes[i - 1].attn:
                    plan = ops.fused_norm_plan(lw.dgrad[0], N, sp.cin_p, twin=tw is not None)
                elif i > 0 and nodes[i - 1].norm and len(lw.dgrad) == 4 and f == 0 and (i - 1) not in inj_x \
                        and not nodes[i - 1].attn:
                    # the four parity classes of a stride-2 conv's data gradient as one halo-resident launch (hconvt.hip)
                    plan = ops.fused_multi_plan(lw.dgrad, N, sp.cin_p, twin=tw is not None)
                ring = ops.fused_ring_plan(lw.dgrad_ring, N, sp.cin_p, twin=tw is not None) \
                    if (plan is not None and nodes[i - 1].act != "tanh") else None     # (the ring form has no tanh' path)
                if ring is None:
                    gx = torch.empty(N, *lw.dgrad_dims, sp.cin_p, dtype=self.ops.act_dtype, device=dev)
                # ... and where every workgroup of that launch is resident at once, the WHOLE norm backward does: the launch
                # writes the previous layer's dy (and the total gradient its skip path wants) instead of gx. Passes of this
                # network on several streams could run two such launches at once (each waits for its own workgroups): not then.
                sync = ops.ring_apply_plan(lw.dgrad_ring, N, sp.cin_p, twin=tw is not None) \
                    if (ring is not None and not self.multi_stream_passes and (i - 1) not in inj_y) else None
                if sync is not None:
                    fz = {"y": s.ys[i - 1], "mean_rstd": s.mrs[i - 1], "g2": g2n, "partial": ring[1], "fold": f,
                          "fold_mode": fmode_n, "act": nodes[i - 1].act, "slope": nodes[i - 1].slope}
                    dy_prev = torch.empty(N, *lw.in_dims, sp.cin_p, dtype=self.ops.act_dtype, device=dev)
                    want_total = nodes[i - 1].res is not None and g2n is not None
                    tot_prev = torch.empty_like(dy_prev) if want_total else None
                    ops.gconv_ring_apply(lw.dgrad_ring, dy, dpack, dy_prev, tot_prev, fz, sync)
                    assert nodes[i - 1].res is None or g2n is not None, "a residual join always brings its skip gradient"
                    pending = (None, 0, g2n, fmode_n, ring, {"dy": dy_prev, "total": tot_prev})
                elif ring is not None:
                    # reflect-padded wide 3x3 layer: the launch folds the ring of padded-domain pixels itself, the gradient
                    # arrives on the unpadded domain
                    gx = torch.empty(N, *lw.in_dims, sp.cin_p, dtype=self.ops.act_dtype, device=dev)
                    ops.gconv(lw.dgrad_ring, dy, dpack, None, gx,
                              fuse={"y": s.ys[i - 1], "mean_rstd": s.mrs[i - 1], "g2": g2n, "partial": ring[1], "fold": f,
                                    "fold_mode": fmode_n, "act": nodes[i - 1].act, "slope": nodes[i - 1].slope})
                    pending = (gx, 0, g2n, fmode_n, ring)
                elif plan is not None:
                    fz = {"y": s.ys[i - 1], "mean_rstd": s.mrs[i - 1], "g2": g2n, "partial": plan[1], "fold": f,
                          "fold_mode": fmode_n, "act": nodes[i - 1].act, "slope": nodes[i - 1].slope}
                    if len(lw.dgrad) == 1:
                        ops.gconv(lw.dgrad[0], dy, dpack, None, gx, fuse=fz)
                    else:
                        ops.gconv_classes(lw.dgrad, dy, dpack, None, gx, fuse=fz)
                    pending = (gx, f, g2n, fmode_n, plan)
                else:
                    ops.gconv_classes(lw.dgrad, dy, dpack, None, gx)
                    pending = (gx, f, g2n, fmode_n)
            if start is None:
                s.acts[i + 1] = None  # release as we go
        if db_items:
            ops.norm_bias_grads(db_items)
        if wst_used:           # whatever follows this pass (the optimiser, another pass of this network) sees its weight gradients
            from ...utils.streams import wait_stream
            wait_stream(torch.cuda.current_stream(), wst)
        if not need_input_grad:
            return None
        gx, f, _, fmode = pending[:4]
        sp0 = nodes[0].spec
        g_ins, n0 = [], 0
        need, k = getattr(self, "_cat_need", None), 0
        for xh in x_imgs:
            gxh = gx[n0:n0 + xh.shape[0]]
            n0 += xh.shape[0]
            if isinstance(xh, ChannelCat):               # (fold is 0 here: the pair form has no padded stem)
                assert f == 0
                want = need[k:k + 2] if need is not None else (True, True)
                k += 2
                ga = torch.empty_like(xh[0]) if want[0] else None
                gb = torch.empty_like(xh[1]) if want[1] else None
                if ga is not None or gb is not None:
                    ops.image_pair_to_act_backward(gxh, ga, gb, xh[0].shape[1], xh[1].shape[1])
                g_ins.append((ga, gb))
                continue
            g_in = torch.empty_like(xh)
            if sp0.wfold == "in":
                ops.image_unfold_backward(gxh, g_in, sp0.k, sp0.pad, f, sp0.pad_mode)
            else:
                ops.image_to_act_backward(gxh, g_in, fold=f, fold_mode=fmode)
            g_ins.append(g_in)
        return tuple(g_ins) if parts else g_ins[0]

    # ---- data parallelism (reference: DistributedDataParallel per network, base.py:172-189) -------------
    def parallelize(self, process_group=None, bucket_bytes=8 << 20):
        """Mark the net data-parallel: master is broadcast from rank 0 (DDP ctor, C3) and the flat gradient is
        all-reduced in buckets as the last backward pass of a step produces them (C4)."""
        import torch.distributed as dist
        self._dist = process_group if process_group is not None else dist.group.WORLD
        with torch.no_grad():
            dist.broadcast(self.master.data, 0, group=self._dist)
        self.mark_packs_dirty()
        # buckets = contiguous [start, end) element ranges of whole layers, built from the LAST layer backwards. The Extras at
        # the tail of the flat buffer (SelfAttentionBlock parameters, whose gradients are written by the block's backward on
        # an EARLIER node's output than the bucket boundary suggests) are a bucket of their own that no layer triggers:
        # finish_grad_reduction reduces it, after the whole backward pass.
        x_start = min(self.x_off.values()) if self.extras else self.numel
        self._buckets, end, cur = [], x_start, x_start
        for i in range(len(self.nodes) - 1, -1, -1):
            cur = self.w_off[i]
            if (end - cur) * 4 >= bucket_bytes or i == 0:
                self._buckets.append((i, cur, end))   # ready once node i's wgrad has been issued
                end = cur
        if self.extras:
            self._buckets.append((-1, x_start, self.numel))
        self._bucket_at = {i: (s, e) for i, s, e in self._buckets if i >= 0}
        return self

    # ---- optimiser chunks under the backward pass (NativeAdam.arm_early) ---------------------------------------------
    # The flat master holds the layers in node order (w_0, b_0, w_1, b_1, ...) and a backward pass walks the nodes last to first:
    # once node i has launched its parameter gradients and its data gradient, nothing of this pass reads or writes the
    # parameters (or bf16 packs) of nodes >= i again, so the optimiser may update [w_off[i], cursor) on another stream while the
    # pass goes on — for a network that takes exactly ONE backward pass per optimiser step (Pix2Pix's generator: 178 M
    # parameters, an update of 1.1 ms at HBM rate that used to start when the backward pass had ended).
    def _early_step_at(self, i):
        fn = getattr(self, "_early_step", None)
        if fn is None:
            return
        end = self._early_cursor
        start = self.w_off[i]
        if end - start < (self._early_min if i > 0 else 1):
            return
        self._early_cursor = start
        fn(self, start, end)

    def _maybe_reduce_bucket(self, i):
        rng = self._bucket_at.get(i)
        if rng is None:
            return
        import torch.distributed as dist
        s, e = rng
        h = dist.all_reduce(self.master.grad[s:e], op=dist.ReduceOp.SUM, group=self._dist, async_op=True)
        self._reduce_handles.append(h)
        self._reduced_buckets.add(i)

    def wgrad_fresh(self, i) -> bool:
        """True for the first weight gradient of layer i since the optimiser cleared the gradient buffer (grad_dirty is
        dropped by the update / zero_grad and raised by every executor behind its writes): its slice still holds zeros, which
        a single-contributor weight-gradient launch may use to store instead of read-add-store (gs_wgrad_desc.dw_fresh; the
        U-Net's 17-34 M-weight layers are pure output traffic). Call exactly once per weight-gradient launch, before it."""
        if not self.grad_dirty:
            self._wgrad_seen.clear()
        fresh = i not in self._wgrad_seen
        self._wgrad_seen.add(i)
        return fresh

    def _wgrad_written(self, i, tw=None) -> bool:
        """EVERY weight-gradient launch of layer i goes through here (merged pairs, flushed holds and twin launches included:
        a launch that is not noted would make a later one look like the first): notes the write in this network (and its twin
        partner) and says whether the slice was fresh in all of them"""
        fresh = self.wgrad_fresh(i)
        if tw is not None:
            fresh = tw.wgrad_fresh(i) and fresh
        return fresh

    def _flush_held(self, i, held):
        wd, a_t, g_t, tw = held
        grad = self.master.grad if tw is None else Twin(self.master.grad, tw.master.grad)
        self._wgrad_written(i, tw)
        self.ops.wgrad(wd, a_t, g_t, grad[self.w_off[i]:self.w_off[i] + self.nodes[i].spec.master_numel])

    def flush_deferred_wgrads(self):
        """weight gradients held back for a merged launch whose partner pass never came"""
        lead = getattr(self, "_twin_lead", None)      # second network of a TwinNet: its share is held by the first
        if lead is not None:
            lead.flush_deferred_wgrads()
        for i, held in sorted(self._deferred.items()):
            self._flush_held(i, held)
        self._deferred = {}

    def finish_grad_reduction(self) -> float:
        """Called by the optimiser before the update. Returns the factor the summed gradient must be scaled by."""
        self.flush_deferred_wgrads()
        if self._dist is None:
            return 1.0
        if self.external_reduce:       # a captured step: the runner all-reduces the flat gradient between its graphs
            import torch.distributed as dist
            self._fw_pending = 0
            self._reduce_handles, self._reduced_buckets = [], set()
            return 1.0 / dist.get_world_size(self._dist)
        import torch.distributed as dist
        if self.grad_dirty:
            # buckets the last backward pass did not reach (partial encoder passes, custom recipes) go now
            for i, s, e in self._buckets:
                if i not in self._reduced_buckets:
                    self._reduce_handles.append(dist.all_reduce(self.master.grad[s:e], op=dist.ReduceOp.SUM,
                                                                group=self._dist, async_op=True))
        for h in self._reduce_handles:
            h.wait()          # stream-level wait on GPU backends; no host sync
        self._reduce_handles = []
        self._reduced_buckets = set()
        self._fw_pending = 0
        return 1.0 / dist.get_world_size(self._dist)


def pairs_shape(pairs):
    return tuple(ChannelCat(p) for p in pairs)


def _split_parts(out, parts):
    outs, n0 = [], 0
    for t in parts:
        outs.append(out[n0:n0 + t.shape[0]])
        n0 += t.shape[0]
    return tuple(outs)


class _TapFn(torch.autograd.Function):
    """Encoder-only pass returning sampled feature patches [N, P, C] (fp32) of several nodes as one autograd node:
    forward gathers rows of the NHWC activations (a patch is one pixel's channel vector), backward scatters the patch
    gradients back and runs the partial backward pass. Several batches (each with its own pixel ids) ride in one pass;
    outputs come batch by batch, tap by tap."""

    @staticmethod
    def forward(ctx, token, net, taps, ids, *xs):
        stop = max(node for _, node in taps)
        xs = tuple(t.detach() for t in xs)
        _, saved = net._forward(xs if len(xs) > 1 else xs[0], save=True, stop=stop)
        ctx.net, ctx.saved, ctx.taps, ctx.ids, ctx.stop = net, saved, taps, ids, stop
        ctx.sizes = tuple(t.shape[0] for t in xs)
        ctx.need_x = any(ctx.needs_input_grad[4:])
        ctx.want_w = net.requires_grad
        if ctx.want_w:
            net._fw_pending += 1
        outs, n0 = [], 0
        for np_, pids in zip(ctx.sizes, ids):
            for (kind, node), pid in zip(taps, pids):
                src = saved.ys[node] if kind == "y" else saved.acts[node + 1]
                outs.append(net.ops.tap_gather(src[n0:n0 + np_], pid, net.nodes[node].spec.cout))
            n0 += np_
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        net, s = ctx.net, ctx.saved
        if ctx.want_w:
            net._fw_pending -= 1
        inj = {"x": {}, "y": {}}
        k, n0 = len(ctx.taps), 0
        for p, (np_, pids) in enumerate(zip(ctx.sizes, ctx.ids)):
            for (kind, node), pid, g in zip(ctx.taps, pids, grads[p * k:(p + 1) * k]):
                if g is not None:       # sparse: (first image, images, pixel ids, [images, P, c] gradient)
                    inj[kind].setdefault(node, []).append((n0, np_, pid, g))
            n0 += np_
        if ctx.want_w:      # passes that write parameter gradients are ordered per network
            net._order_backward_begin()
        gx = net._backward(s, None, ctx.need_x, ctx.want_w, start=ctx.stop, inj_x=inj["x"], inj_y=inj["y"])
        if ctx.want_w:
            net._order_backward_end()
        ctx.saved = None
        if gx is None:
            return (None,) * (4 + len(ctx.sizes))
        gx = gx if isinstance(gx, tuple) else (gx,)
        return (None, None, None, None) + tuple(g if need else None for g, need in zip(gx, ctx.needs_input_grad[4:]))


class ChannelCat(tuple):
    """(a, b): two image batches that enter a network side by side along the channel axis — torch.cat([a, b], dim=1) without
    the launch (NativeNet.forward_parts_cat); looks like the concatenated tensor where the executor asks for a shape"""

    @property
    def shape(self):
        a, b = self
        return torch.Size((a.shape[0], a.shape[1] + b.shape[1]) + tuple(a.shape[2:]))


class _CatPartsFn(torch.autograd.Function):
    """forward_parts over batches given as channel pairs: inputs come flattened (a_0, b_0, a_1, b_1, ...)"""

    @staticmethod
    def forward(ctx, token, net, *ts):
        xs = tuple(ChannelCat((ts[i].detach(), ts[i + 1].detach())) for i in range(0, len(ts), 2))
        out, saved = net._forward(xs, save=True)
        ctx.net, ctx.saved = net, saved
        ctx.need = tuple(ctx.needs_input_grad[2:])
        ctx.want_w = net.requires_grad
        if ctx.want_w:
            net._fw_pending += 1
        ctx.set_materialize_grads(False)
        return _split_parts(out, xs)

    @staticmethod
    def backward(ctx, *grads):
        net = ctx.net
        if ctx.want_w:
            net._fw_pending -= 1
            net._order_backward_begin()
        for g in grads:
            if g is not None and g.is_cuda:
                g.record_stream(torch.cuda.current_stream())
        net._cat_need = ctx.need                       # which images want a gradient (NativeNet._backward reads it)
        gx = net._backward(ctx.saved, grads, any(ctx.need), ctx.want_w)
        net._cat_need = None
        if ctx.want_w:
            net._order_backward_end()
        ctx.saved = None
        if gx is None:
            return (None,) * (2 + len(ctx.need))
        flat = [t for pair in gx for t in pair]
        return (None, None) + tuple(g if need else None for g, need in zip(flat, ctx.need))


class _PartsFn(torch.autograd.Function):
    """several batches through the whole network as one autograd node (NativeNet.forward_parts)"""

    @staticmethod
    def forward(ctx, token, net, *xs):
        xs = tuple(t.detach() for t in xs)
        out, saved = net._forward(xs, save=True)
        n0 = 0
        for t in xs:
            net._remember_pass(t, saved, n0)
            n0 += t.shape[0]
        ctx.net, ctx.saved = net, saved
        ctx.need = tuple(ctx.needs_input_grad[2:])
        ctx.want_w = net.requires_grad
        if ctx.want_w:
            net._fw_pending += 1
        ctx.set_materialize_grads(False)
        return _split_parts(out, xs)

    @staticmethod
    def backward(ctx, *grads):
        net = ctx.net
        if ctx.want_w:
            net._fw_pending -= 1
            net._order_backward_begin()
        for g in grads:
            if g is not None and g.is_cuda:
                g.record_stream(torch.cuda.current_stream())
        gx = net._backward(ctx.saved, grads, any(ctx.need), ctx.want_w)
        if ctx.want_w:
            net._order_backward_end()
        ctx.saved = None
        if gx is None:
            return (None,) * (2 + len(grads))
        return (None, None) + tuple(g if need else None for g, need in zip(gx, ctx.need))


class _NetFn(torch.autograd.Function):
    """The whole network as one autograd node."""

    @staticmethod
    def forward(ctx, x, token, net: NativeNet):
        out, saved = net._forward(x.detach(), save=True)
        net._remember_pass(x, saved)
        ctx.net, ctx.saved = net, saved
        ctx.need_x = ctx.needs_input_grad[0]
        ctx.want_w = net.requires_grad     # autograd semantics: decided when the graph is recorded
        if ctx.want_w:
            net._fw_pending += 1
        return out

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        if ctx.want_w:
            net._fw_pending -= 1
        if ctx.want_w:      # passes that write parameter gradients are ordered per network
            net._order_backward_begin()
        if g.is_cuda:           # the incoming gradient may come from another stream's allocator pool
            g.record_stream(torch.cuda.current_stream())
        gx = net._backward(ctx.saved, g, ctx.need_x, ctx.want_w)
        if ctx.want_w:
            net._order_backward_end()
        ctx.saved = None
        return gx, None, None
