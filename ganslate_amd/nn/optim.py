"""Fused flat Adam over the executor's master buffers (gs_adam_step). Subclasses torch.optim.Optimizer so the
reference's LambdaLR schedule (nn/utils.py:83-99) and `param_groups[0]['lr']` logging (base.py:318-319) work
unchanged; constructor signature mirrors torch.optim.Adam(params, lr, betas) as used in cyclegan.py:81-82."""
import os

import torch

from .native.backend import get_ops


class NativeAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        for p in params:
            if getattr(p, "_owner_net", None) is None:
                raise TypeError("NativeAdam only optimises the flat master parameters of NativeNet instances")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.external_prepare = False
        self.deferred_to = None

    def prepare(self):
        """Host side of one update: advance the step counters and upload the step-dependent scalars (learning rate from
        the scheduler, bias corrections) to each parameter's device hyper vector. `step()` calls it itself unless a
        captured step owns the launches (external_prepare), in which case the caller prepares before every replay."""
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                if "hyper" not in st:
                    st["hyper"] = torch.zeros(6, dtype=torch.float32, device=p.device)
                st["step"] += 1
                t = st["step"]
                host = torch.tensor([group["lr"], b1, b2, group["eps"], 1.0 - b1 ** t, (1.0 - b2 ** t) ** 0.5],
                                    dtype=torch.float64).float()
                if p.is_cuda:       # pinned + non_blocking: a pageable upload would make the host wait for the stream
                    host = host.pin_memory()
                st["hyper"].copy_(host, non_blocking=True)

    # ---- the update in chunks under the backward pass -----------------------------------------------------------------------
    # torch's `loss.backward(); optimizer.step()` (pix2pix.py:84-88) updates after the whole backward pass. A layer's parameters
    # are done with once its own gradients are launched (NativeNet._early_step_at), so for networks that take ONE backward pass
    # per step the update of the layers already passed can be launched while the pass goes on: same arithmetic per element.
    # Pix2Pix's 178 M-parameter generator (update 1.1 ms at HBM rate): as launches BETWEEN the pass's own, each chunk right
    # behind the weight gradients it consumes, the step takes 3.65 instead of 3.82 ms; on a stream of its own beside the pass
    # it takes 4.1 ms (the HBM-bound chunks slow the pass's weight-streaming launches more than they hide;
    # profiles/r06_ab_pix2pix.txt). arm_early() before backward(), step() as always (it updates what the pass did not hand over).
    EARLY_MIN = 1 << 21          # smallest chunk worth a launch (elements)

    def arm_early(self, stream=None):
        """stream: the CUDA stream the chunks run on (None: the current one — tests). Returns False (and arms nothing) where the
        early form does not apply: data-parallel networks (the gradient is all-reduced first), executors without the hook."""
        nets = [p._owner_net for group in self.param_groups for p in group["params"]]
        if self.deferred_to is not None or any(getattr(n, "_dist", None) is not None or not hasattr(n, "_early_step_at")
                                               or getattr(n, "_twin_lead", None) is not None for n in nets):
            return False
        if not self.external_prepare:
            self.prepare()
        self._early = {"stream": stream, "prepared": True, "launched": False,
                       "tr": os.environ.get("GS_WGRAD_ADAM_TR", "1") != "0"}       # (A/B switch: transposed packs by the fused launch)
        fuse = os.environ.get("GS_WGRAD_ADAM", "1") != "0"       # (A/B switch)
        for net in nets:
            # a network whose large layers took the fused launch last step: what is left (biases, small layers) goes in ONE
            # multi-range launch behind the pass instead of a dozen chunk launches between its launches
            ranged = fuse and getattr(net, "_last_fused", 0) > 0 and os.environ.get("GS_ADAM_RANGES", "1") != "0"      # (A/B switch)
            net._early_step = None if ranged else self._early_chunk
            # layers of few pixels: weight gradient + update in one launch (gs_wgrad_adam) — the executor asks per layer
            net._early_fuse = self._fuse_args if fuse else None
            net._early_fused, net._early_tr = [], []
            net._early_cursor = net.b_off[-1] + net.nodes[-1].spec.cout_p      # end of the node parameters
            net._early_min = int(os.environ.get("GS_EARLY_ADAM_MIN", self.EARLY_MIN))
        return True

    @staticmethod
    def _pack_slices(net, start, end):
        tgt = net.fused_pack_targets() if hasattr(net, "fused_pack_targets") else None
        if not tgt:
            return None
        inv_f, fpack, inv_d, dpack = tgt[1]
        g0, g1 = start // 8, (end + 7) // 8
        return (inv_f[g0:g1] if inv_f is not None else None, fpack, inv_d[g0:g1] if inv_d is not None else None, dpack)

    def _fuse_args(self, net, i):
        """((p, m, v, hyper, packs, transposed tables) of layer i's weights for ops.wgrad_adam, which pack the tables are of)"""
        p = net.master
        a, b = net.w_off[i], net.w_off[i] + net.nodes[i].spec.master_numel
        if p.grad is None or a % 8 or p.data_ptr() % 16:
            return None, None
        st = self.state[p]
        tr = net.transposed_tables(i) if (self._early["tr"] and hasattr(net, "transposed_tables")) else None
        if tr is not None and not hasattr(net, "fused_pack_targets"):
            tr = None
        return (p.data[a:b], st["exp_avg"][a:b], st["exp_avg_sq"][a:b], st["hyper"], self._pack_slices(net, a, b),
                tr[1:] if tr is not None else None), (tr[0] if tr is not None else None)

    @torch.no_grad()
    def _update_range(self, p, net, start, end):
        """the update of [start, end) of the flat buffer, minus the layers that were updated with their weight gradient"""
        st = self.state[p]
        pos = start
        for a, b in sorted(getattr(net, "_early_fused", ())) + [(end, end)]:
            a, b = min(max(a, start), end), min(max(b, start), end)
            if a > pos:
                get_ops().adam_step_dev(p.data[pos:a], p.grad[pos:a], st["exp_avg"][pos:a], st["exp_avg_sq"][pos:a],
                                        st["hyper"], grad_scale=1.0, zero_grad=True, packs=self._pack_slices(net, pos, a))
            pos = max(pos, b)

    @staticmethod
    def _rest_ranges(p, net):
        """(device int64 [n][2], longest) ranges of the flat buffer outside the layers in net._early_fused — built when a step has
        told which layers take the fused launch (host work and an upload: the step that uses it may be a captured one)"""
        pos, ranges = 0, []
        for a, b in sorted(net._early_fused) + [(p.numel(), p.numel())]:
            if a > pos:
                ranges.append((pos, a))
            pos = max(pos, b)
        key = tuple(ranges)
        cache = net.__dict__.setdefault("_rest_ranges_cache", {})
        if key not in cache:
            cache[key] = (torch.tensor(ranges, dtype=torch.int64, device=p.device).reshape(-1, 2), max(b - a for a, b in ranges))
        return cache[key]

    @torch.no_grad()
    def _update_rest(self, p, net, tgt):
        """everything of the flat buffer outside the layers that were updated with their weight gradient, in one launch"""
        st = self.state[p]
        dev, max_len = self._rest_ranges(p, net)
        get_ops().adam_step_dev_ranges(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], dev, max_len, st["hyper"],
                                       grad_scale=1.0, zero_grad=True, packs=tgt[1] if tgt else None)

    def _early_chunk(self, net, start, end):
        from ..utils import streams
        p = net.master
        if p.grad is None or start % 8 or start >= end:
            net._early_cursor = end          # (not handed over: step() takes it)
            return
        side = self._early["stream"]
        self._early["launched"] = True
        if side is None:
            self._update_range(p, net, start, end)
            return
        ev = streams.new_event()
        ev.record()
        side.wait_event(ev)
        ops = get_ops()
        cap = int(os.environ.get("GS_EARLY_ADAM_BLOCKS", "0")) if hasattr(ops, "lib") else 0
        if cap:      # (the grid is fixed when the launch is enqueued)
            ops.lib.gs_set_option(b"adam_blocks", cap)
        with torch.cuda.stream(side):
            self._update_range(p, net, start, end)
        if cap:
            ops.lib.gs_set_option(b"adam_blocks", 8192)

    @torch.no_grad()
    def step(self, closure=None):
        if self.deferred_to is not None:     # a data-parallel captured step: the update is launched after the all-reduce
            self.deferred_to.append(self)
            return
        early = getattr(self, "_early", None)
        if not self.external_prepare and not (early and early["prepared"]):
            self.prepare()
        self.launch()

    @torch.no_grad()
    def launch(self):
        ops = get_ops()
        early, self._early = getattr(self, "_early", None), None
        if early and early["stream"] is not None and early["launched"]:
            from ..utils import streams
            streams.wait_stream(torch.cuda.current_stream(), early["stream"])
        for group in self.param_groups:
            for p in group["params"]:
                net = p._owner_net
                if early:
                    net._early_step = net._early_fuse = None
                if p.grad is None:
                    continue
                scale = net.finish_grad_reduction()
                st = self.state[p]
                # the update writes the row-major bf16 weight packs as it goes where the network has one pack set
                tgt = net.fused_pack_targets() if hasattr(net, "fused_pack_targets") else None
                nodes_end = net.b_off[-1] + net.nodes[-1].spec.cout_p if early else 0      # (armed networks are NativeNets)
                if early and net._early_fused and net._early_cursor == nodes_end and hasattr(ops, "adam_step_dev_ranges"):
                    self._update_rest(p, net, tgt)       # nothing handed over in chunks: the rest of the buffer in one launch
                elif early and (net._early_cursor < net.b_off[-1] + net.nodes[-1].spec.cout_p or net._early_fused):
                    # chunks of this pass are done (or under way on the joined stream): the layers it did not hand over and
                    # whatever follows the node parameters in the flat buffer
                    nodes_end = net.b_off[-1] + net.nodes[-1].spec.cout_p
                    for a, b in ((0, net._early_cursor), (nodes_end, p.numel())):
                        if b > a:
                            self._update_range(p, net, a, b)
                else:
                    ops.adam_step_dev(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], st["hyper"], grad_scale=scale,
                                      zero_grad=True, packs=tgt[1] if tgt else None)
                net.grad_dirty = False
                if early:
                    net._last_fused = len(net._early_fused)
                    if net._last_fused and not (p.is_cuda and torch.cuda.is_current_stream_capturing()):
                        self._rest_ranges(p, net)        # (for the next step's one-launch form)
                    net._early_fused = []
                    net._tr_fresh = frozenset(getattr(net, "_early_tr", ())) if tgt else frozenset()
                    net._early_tr = []
                if tgt:
                    net.mark_packs_dirty(ident_fresh=tgt[0], tr_fresh=getattr(net, "_tr_fresh", ()) if early else ())
                else:
                    net.mark_packs_dirty()

    def mark_updated(self):
        """a captured launch() was replayed: same host bookkeeping as launch(), no launches"""
        for group in self.param_groups:
            for p in group["params"]:
                net = p._owner_net
                if p.grad is None:
                    continue
                tgt = net.fused_pack_targets() if hasattr(net, "fused_pack_targets") else None
                # (a replayed step that was captured with the early form wrote the same transposed segments again)
                net.mark_packs_dirty(**({"ident_fresh": tgt[0], "tr_fresh": getattr(net, "_tr_fresh", ())} if tgt else {}))

    def state_dict(self):
        sd = super().state_dict()       # the hyper vectors are derived state: rebuilt by the next prepare()
        sd["state"] = {k: {kk: vv for kk, vv in st.items() if kk != "hyper"} for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        """moments are loaded INTO the existing device buffers (a captured step keeps their addresses)"""
        old = {p: dict(st) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        for p, st_old in old.items():
            st_new = self.state[p]
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st_old and k in st_new:
                    st_old[k].copy_(st_new[k])
                    st_new[k] = st_old[k]
            if "hyper" in st_old:
                st_new["hyper"] = st_old["hyper"]

    # ---- the reference's checkpoint format (torch.optim.Adam.state_dict over per-layer parameters, base.py:244-245) --
    def reference_state_dict(self):
        """This optimiser's state as `torch.optim.Adam(itertools.chain(net.parameters() ...)).state_dict()` of the
        reference would hold it: one entry per torch parameter (conv weight / bias / PReLU slope, torch layouts), indexed
        in the networks' parameters() order (cyclegan.py:70-82: G_AB then G_BA; D_B then D_A)."""
        state, index = {}, 0
        groups = []
        for group in self.param_groups:
            ids = []
            for p in group["params"]:
                net, st = p._owner_net, self.state.get(p, {})
                moments = {k: net.flat_to_tensors(st[k].detach()) for k in ("exp_avg", "exp_avg_sq") if k in st}
                for key in net.reference_parameter_order():
                    if moments:
                        state[index] = {"step": torch.tensor(float(st["step"])),
                                        "exp_avg": moments["exp_avg"][key].cpu(),
                                        "exp_avg_sq": moments["exp_avg_sq"][key].cpu()}
                    ids.append(index)
                    index += 1
            g = {k: v for k, v in group.items() if k != "params"}
            g.update(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                     fused=None, params=ids)
            groups.append(g)
        return {"state": state, "param_groups": groups}

    def load_reference_state_dict(self, sd):
        """inverse of reference_state_dict: accepts what the reference wrote (`optimizer_G` / `optimizer_D`)"""
        index = 0
        for group, saved in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "betas", "eps", "initial_lr"):
                if k in saved:
                    group[k] = tuple(saved[k]) if k == "betas" else saved[k]
            for p in group["params"]:
                net = p._owner_net
                keys = net.reference_parameter_order()
                entries = [sd["state"].get(index + i) for i in range(len(keys))]
                index += len(keys)
                if any(e is None for e in entries):
                    if all(e is None for e in entries):
                        continue                      # optimiser never stepped
                    raise ValueError("reference optimizer state covers only part of a network's parameters")
                st = self.state[p]
                st.setdefault("exp_avg", torch.zeros_like(p))
                st.setdefault("exp_avg_sq", torch.zeros_like(p))
                st["step"] = int(float(entries[0]["step"]))
                for mk in ("exp_avg", "exp_avg_sq"):
                    net.tensors_to_flat({key: e[mk] for key, e in zip(keys, entries)}, st[mk])
        n_saved = sum(len(g["params"]) for g in sd["param_groups"])
        if n_saved != index:
            raise ValueError(f"reference optimizer state has {n_saved} parameters, these networks have {index}")

    def zero_grad(self, set_to_none=True):
        """The update kernel already cleared the gradient it consumed; only buffers written since are cleared."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None and p._owner_net.grad_dirty:
                    p.grad.zero_()
                    p._owner_net.grad_dirty = False
