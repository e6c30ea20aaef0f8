"""BaseGAN — the plugin boundary the Trainer talks to (ganslate/nn/gans/base.py:16-321): public dicts
`networks / optimizers / losses / visuals / metrics`, abstract `set_input / forward / optimize_parameters /
init_criterions / init_optimizers`, concrete `setup / backward / parallelize_networks / save_checkpoint /
load_networks / set_requires_grad / infer / get_loggable_data / update_learning_rate`.

Differences that are deliberate: networks are HIP executors (NativeNet) instead of nn.Modules;
`train.mixed_precision` selects nothing — activations are always bf16 with fp32 master weights/statistics/
losses (apex AMP O1 fp16, base.py:118-126, has no MI355X counterpart); data parallelism is the executor's own
bucketed all-reduce over RCCL instead of DistributedDataParallel (base.py:172-189)."""
import logging
import os
from abc import ABC, abstractmethod
from pathlib import Path

import torch

from ...utils import communication, io, streams
from ...utils.builders import build_D, build_G
from ...utils.metrics import TrainingMetrics
from ..utils import get_scheduler


class BaseGAN(ABC):
    # ---- captured training step (no counterpart in the reference) ---------------------------
    # A recipe whose optimize_parameters enqueues the same launch sequence every iteration sets graph_capturable and
    # names the visuals that set_input fills. The second call of optimize_parameters is then captured into a hipGraph
    # (torch.cuda.CUDAGraph on the launch stream) and every later iteration is one graph launch: the ~500 kernel
    # launches of a CycleGAN step cost the host ~20-25 ms to enqueue one by one but only ~15 ms to execute.
    graph_capturable = False
    input_visuals = ("real_A", "real_B")
    side_stream_names = ()       # streams the recipe forks work onto (created up front, outside any capture)

    def __init__(self, conf):
        self.logger = logging.getLogger("ganslate_amd")
        self.conf = conf
        self.is_train = self.conf.mode == "train"
        self.device = self._specify_device()
        self.output_dir = conf[conf.mode].output_dir
        self.visuals, self.metrics, self.losses, self.optimizers, self.networks = {}, {}, {}, {}, {}

    def init_networks(self):
        for name in self.networks.keys():
            if name.startswith("G"):
                direction = "BA" if name.endswith("_BA") else "AB"
                self.networks[name] = build_G(self.conf, direction, self.device)
            elif name.startswith("D"):
                domain = "A" if name.endswith("_A") else "B"
                self.networks[name] = build_D(self.conf, domain, self.device)

    @abstractmethod
    def init_criterions(self):
        """Initialize criterions (losses)"""

    @abstractmethod
    def init_optimizers(self):
        """Initialize optimizers"""

    def init_metrics(self):
        self.training_metrics = TrainingMetrics(self.conf)

    def init_schedulers(self):
        self.schedulers = [get_scheduler(optim, self.conf) for optim in self.optimizers.values()]

    def _specify_device(self):
        from ..native.backend import get_ops
        if torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl":
            dev = torch.device(f"cuda:{communication.get_local_rank()}")
            torch.cuda.set_device(dev)
        else:
            dev = get_ops().device          # raises loudly when neither the GPU nor the library is there
        if not self.conf[self.conf.mode].cuda and dev.type == "cuda":
            self.logger.warning("`cuda: False` requested, but this build only has the MI355X path; running on %s", dev)
        return get_ops().device

    @abstractmethod
    def set_input(self, input):
        """Unpack input data from the dataloader."""

    @abstractmethod
    def forward(self):
        """Run forward pass."""

    @abstractmethod
    def optimize_parameters(self):
        """Calculate losses, gradients, and update network weights; called in every training iteration"""

    def setup(self):
        assert "G" in self.networks or "G_AB" in self.networks, "The (main) generator has to be named `G` or `G_AB`."
        if self.conf[self.conf.mode].mixed_precision:
            self.logger.info("mixed_precision: bf16 activations / fp32 master weights are always on in this build; "
                             "apex opt_level is ignored")
        from ..native.backend import get_ops
        getattr(get_ops(), "sync_options", lambda: None)()      # GS_* kernel-selection switches -> library options
        self.init_networks()
        if self.is_train:
            self.init_criterions()
            self.init_optimizers()
            self.init_metrics()
            self.init_schedulers()
        else:
            self.eval()
            if len(self.networks.keys()) != 1:
                raise ValueError("When inferring there should be only one network initialized - generator.")
        if self.conf[self.conf.mode].checkpointing.load_iter:
            self.load_networks(self.conf[self.conf.mode].checkpointing.load_iter)
        if int(os.environ.get("WORLD_SIZE", 1)) > 1 or os.environ.get("GS_FORCE_DDP") == "1":
            self.parallelize_networks()      # GS_FORCE_DDP: the data-parallel path with a 1-rank group (tests)
        if self.is_train:
            self._init_step_graph()

    # ---- second launch stream -------------------------------------------------------------
    # The discriminators' update of an iteration only needs the forward pass' images and the discriminator weights the
    # generator loss has already read, so its launches can run beside the generators' backward pass instead of after
    # it: same arithmetic, same host order, another HIP stream. Recipes mark the point from which the side work may
    # start (fork_side_work), wrap it (side_work) and join before the discriminator optimiser step.
    def _side_stream_enabled(self, name):
        from ..native.backend import get_ops
        want = os.environ.get("GS_SIDE_STREAM", "1")        # "0": none, "1": all, or a comma list of names
        if not ((want == "1" or name in want.split(",")) and self.device.type == "cuda"
                and getattr(get_ops(), "name", "") == "hip"):
            return False
        # data parallel: only inside a captured step, where the gradient all-reduce happens between the graphs; a
        # launch-by-launch iteration issues its bucketed all-reduces from inside backward and stays on one stream
        return not self._data_parallel_nets() or torch.cuda.is_current_stream_capturing()

    def fork_side_work(self, name="D"):
        """everything launched so far on the current stream happens-before the next side_work(name) block"""
        if not self._side_stream_enabled(name):
            return
        if not hasattr(self, "_side"):
            self._side = {}
        if name not in self._side:
            self._side[name] = {"stream": torch.cuda.Stream(device=self.device), "fork": None, "busy": False}
        st = self._side[name]
        st["fork"] = streams.new_event()
        st["fork"].record()

    def side_work(self, name="D"):
        import contextlib
        st = getattr(self, "_side", {}).get(name)
        if st is None or st["fork"] is None:
            return contextlib.nullcontext()
        st["stream"].wait_event(st["fork"])
        st["busy"] = True
        return torch.cuda.stream(st["stream"])

    def join_side_work(self, name="D", last=True):
        """the current stream waits for the side stream. last=False keeps the fork alive: autograd runs the backward of
        what was launched in side_work(name) on that stream again, and since the networks accumulate their parameter
        gradients themselves (no AccumulateGrad leaves) the engine does not join it — the recipe joins once more after
        its backward()."""
        st = getattr(self, "_side", {}).get(name)
        if st is not None and st["busy"]:
            streams.wait_stream(torch.cuda.current_stream(), st["stream"])
            if last:
                st["busy"], st["fork"] = False, None

    def arm_early_update(self, name):
        """before the backward pass of a network group that takes ONE backward pass per step: its optimiser updates the layers
        the pass is done with while the pass goes on (NativeAdam.arm_early) — by default as launches of the same stream, between
        the pass's own (Pix2Pix: 3.82 -> 3.65 ms per step, profiles/r06_ab_pix2pix.txt). GS_EARLY_ADAM=0: the update after the
        pass; =stream: the chunks on the 'opt' stream beside the pass (measured: slower, 4.1 ms)."""
        mode = os.environ.get("GS_EARLY_ADAM", "1")
        optim = self.optimizers[name]
        if mode == "0" or not hasattr(optim, "arm_early"):
            return False
        if mode != "stream":
            return optim.arm_early(None)
        st = getattr(self, "_side", {}).get("opt")
        if st is None or not self._side_stream_enabled("opt"):
            return False
        return optim.arm_early(st["stream"])

    # ---- captured training step -------------------------------------------------------------------------------
    def _init_step_graph(self):
        """Decide whether iterations run as graph replays. Off for backends without streams (the CPU test oracle) and
        with GS_STEP_GRAPH=0. Data-parallel runs replay two graphs per iteration with the gradient all-reduce between
        them (see _capture_step)."""
        from ..native.backend import get_ops
        self._graph, self._graph_shapes, self._graph_calls, self._graph_broken = None, None, 0, False
        self._graph_update = None
        nets_capturable = all(getattr(net, "graph_capturable", True) for net in self.networks.values())
        self.step_graph_enabled = (self.graph_capturable and nets_capturable
                                   and os.environ.get("GS_STEP_GRAPH", "1") != "0"
                                   and getattr(get_ops(), "name", "") == "hip" and self.device.type == "cuda")
        if self.device.type == "cuda":
            self._side = {n: {"stream": torch.cuda.Stream(device=self.device), "fork": None, "busy": False}
                          for n in self.side_stream_names}
        if self.step_graph_enabled:
            self._eager_set_input, self._eager_step = self.set_input, self.optimize_parameters
            self.set_input, self.optimize_parameters = self._graph_set_input, self._graph_step

    def _native_nets(self):
        """the executors behind `networks`: a composite (MultiScalePatchGAN3D: one PatchGAN3D per scale) counts as its parts"""
        out = []
        for net in self.networks.values():
            out.extend(net.native_children() if hasattr(net, "native_children") else [net])
        return out

    def _data_parallel_nets(self):
        return [net for net in self._native_nets() if getattr(net, "_dist", None) is not None]

    def _step_pools(self):
        """ImagePools in the order optimize_parameters queries them (their coin flips are drawn before a replay)"""
        return []

    def _set_external_host_state(self, on):
        for optim in self.optimizers.values():
            optim.external_prepare = on
        for pool in self._step_pools():
            pool.external_draw = on
        for net in self._native_nets():             # per-iteration host state of the networks (dropout seeds)
            if hasattr(net, "external_draw"):
                net.external_draw = on

    def _prepare_host_state(self):
        """what the host contributes to one iteration besides the launches: optimiser step counters / learning rates /
        bias corrections and the image pools' coin flips, uploaded to the device vectors the captured kernels read"""
        batch = self.visuals[self.input_visuals[0]].shape[0]
        for pool in self._step_pools():      # drawn in query order, like the reference's sequential step
            pool.draw(batch)
        for optim in self.optimizers.values():
            optim.prepare()
        for net in self._native_nets():
            if hasattr(net, "prepare_host_state"):
                net.prepare_host_state()

    def _rewind_host_state(self):
        """re-arm the per-step host cursors over what _prepare_host_state drew (no new draws): the same iteration is
        about to be recorded again"""

    def _graph_set_input(self, input):
        self._eager_set_input(input)
        if self._graph is not None:
            for name in self.input_visuals:
                static, new = self._static_inputs[name], self.visuals[name]
                if static.shape == new.shape and static.dtype == new.dtype:
                    static.copy_(new)
                    self.visuals[name] = static

    def _input_shapes(self):
        return tuple((tuple(self.visuals[n].shape), self.visuals[n].dtype) for n in self.input_visuals)

    def _graph_step(self):
        self._graph_calls += 1
        if self._graph is not None and self._input_shapes() == self._graph_shapes and self.step_graph_enabled:
            self._prepare_host_state()
            self._replay()
            self.visuals.update(self._graph_out[0]); self.losses.update(self._graph_out[1])
            self.metrics.update(self._graph_out[2])
            return
        if self._graph is None and self._graph_calls >= 2 and not self._graph_broken and self.step_graph_enabled:
            return self._capture_step()
        self._set_external_host_state(False)
        self._eager_step()

    def _replay(self):
        self._graph.replay()
        self._mark_replayed_updates()
        if self._graph_update is not None:       # data parallel: sum the flat gradients, then the optimiser launches
            import torch.distributed as dist
            timing = getattr(self, "reduce_timing", None)      # bench.py: [(event, event)] around the exposed reduction
            if timing is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for net in self._reduced_nets:
                dist.all_reduce(net.master.grad, op=dist.ReduceOp.SUM, group=net._dist)
            if timing is not None:
                e1.record()
                timing.append((e0, e1))
            self._graph_update.replay()

    def _mark_replayed_updates(self):
        """a replay moved the masters behind the host's back: the packs an eager pass (validation, inference) between two
        replays finds are a mix — row-major groups written by the captured fused Adam, transposed segments and leftover
        groups from the start of that replay. Tell the executors, so that such a pass refreshes what the update did not."""
        for optim in self.optimizers.values():
            mark = getattr(optim, "mark_updated", None)
            if mark is not None:
                mark()

    def _capture_step(self):
        """Record this iteration's launches (capture does not execute them), then run it as the first replay.

        Data-parallel runs: the optimisers' step() calls made by the recipe are held back while the first graph records
        forward, backward and everything else, and a second graph holds the held-back Adam launches (the 1/world factor
        is folded into them). Nothing in an iteration reads the updated weights before its end, so moving the updates
        behind the last backward pass changes no arithmetic. The gradient reduction comes in two forms:
          "between"  collectives stay outside the graphs: the flat gradient of every network is all-reduced launch by
                     launch between the two graphs (one collective per network on RCCL's own stream, not overlapped);
          "captured" the bucketed all-reduces the executors issue during their last backward pass are CAPTURED into the
                     first graph (RCCL's kernels become nodes on their own branch, joined by the event wait in
                     finish_grad_reduction), so a replayed iteration overlaps the reduction of the upper buckets and of
                     the discriminators with the remaining backward work.
        Default for world > 1: "between" — the form with nothing of RCCL inside a graph; a multi-rank hang inside a replayed
        collective has no fallback, and no multi-GPU box has run the captured form yet. GS_DDP_GRAPH_COLLECTIVES=auto opts
        into the self-check (BOTH forms are built and checked against each other on this very iteration before either is
        trusted, _ddp_self_check: "captured" is kept when the two agree); =0 / =1 force a form. A single-rank group
        (GS_FORCE_DDP tests) defaults to auto: there is nobody to hang with."""
        self.logger.info("capturing the training step into a hipGraph (GS_STEP_GRAPH=0 runs it launch by launch, "
                         "GS_SIDE_STREAM=0 on one stream)")
        self._static_inputs = {n: self.visuals[n].clone() for n in self.input_visuals}
        self.visuals.update(self._static_inputs)
        self._set_external_host_state(True)
        self._prepare_host_state()
        dp_nets = self._data_parallel_nets()
        if dp_nets:
            import torch.distributed as dist
            default = "0" if dist.get_world_size(dp_nets[0]._dist) > 1 else "auto"
            want = os.environ.get("GS_DDP_GRAPH_COLLECTIVES", default)
        else:
            want = "none"
        self.ddp_form_requested = want
        if want in ("0", "1", "none"):
            form = {"0": "between", "1": "captured", "none": None}[want]
            chosen = self._capture_graphs(dp_nets, form)
        else:
            chosen = self._ddp_self_check(dp_nets)
        self._graph, self._graph_update, self._graph_out, form = chosen
        self._graph_collectives = form == "captured"
        self._reduced_nets = dp_nets if form == "between" else []
        self._graph_shapes = self._input_shapes()
        self.visuals.update(self._graph_out[0]); self.losses.update(self._graph_out[1])
        self.metrics.update(self._graph_out[2])
        self._replay()

    def _capture_graphs(self, dp_nets, form):
        """-> (step graph, update graph or None, (visuals, losses, metrics) of the recorded step, form)"""
        pending = [] if dp_nets else None
        self._rewind_host_state()            # (the self-check records the step twice on one set of draws)
        for optim in self.optimizers.values():
            optim.deferred_to = pending
        for net in dp_nets:
            net.external_reduce = form == "between"
        torch.cuda.synchronize()
        graph, update = torch.cuda.CUDAGraph(), None
        # No garbage collection while the capture is open: a cyclic-garbage sweep that happens to run between two launches
        # may finalise streams / events / graphs of objects that died earlier (another model, a finished validation run), and
        # destroying those is not permitted during a global-mode capture — the process aborts inside a destructor
        # (seen in the test-suite: "Fatal Python error: Aborted ... Garbage-collecting" under _capture_step).
        import gc
        gc_was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            # data parallel: RCCL's watchdog thread may poll events of earlier collectives while this thread captures;
            # only this thread's calls are policed then
            mode = "thread_local" if dp_nets else "global"
            with torch.cuda.graph(graph, capture_error_mode=mode):
                self._eager_step()
                for net in dp_nets:
                    net.flush_deferred_wgrads()
                    if form == "captured":       # buckets the backward passes did not reach, and the join of all of them
                        net.finish_grad_reduction()
            if dp_nets:
                for net in dp_nets:              # (reduced already, one way or the other: the update only scales)
                    net.external_reduce = True
                update = torch.cuda.CUDAGraph()
                with torch.cuda.graph(update, capture_error_mode=mode):
                    for optim in pending:
                        optim.launch()
        except Exception as e:      # a recipe with a host-dependent launch sequence: stay eager, loudly
            self._graph_broken = True
            streams.release_events()
            self._set_external_host_state(False)
            torch.cuda.synchronize()
            raise RuntimeError(f"{type(self).__name__}: the training step could not be captured into a hipGraph "
                               f"({e}); set GS_STEP_GRAPH=0 to run it launch by launch") from e
        finally:
            if gc_was_enabled:
                gc.enable()
            for optim in self.optimizers.values():
                optim.deferred_to = None
            for net in dp_nets:
                net.external_reduce = False
        streams.release_events()
        return graph, update, (dict(self.visuals), dict(self.losses), dict(self.metrics)), form

    def _ddp_self_check(self, dp_nets):
        """Build both reduction forms and make the first multi-GPU run explain itself: each form's step graph is replayed
        once on THIS iteration's inputs with the weights frozen (the update graphs are not replayed; image pools are put
        back, gradients zeroed in between), and the reduced flat gradients must agree — between the forms (bit for bit up
        to 2 ranks, where a + b has one summation order; to 1e-5 of the largest gradient beyond, where a bucket's ring
        order differs from the whole buffer's) and between the ranks (bit for bit: an all-reduce hands every rank the same
        sums). All ranks take the same decision; the outcome is logged and kept in `ddp_self_check`."""
        import torch.distributed as dist
        group = dp_nets[0]._dist
        world = dist.get_world_size(group)
        forms = {"between": self._capture_graphs(dp_nets, "between")}
        pools = [(p, p.images.clone()) for p in self._step_pools() if getattr(p, "images", None) is not None]

        def replay(name):
            for net in dp_nets:
                net.master.grad.zero_()
            forms[name][0].replay()
            if name == "between":
                for net in dp_nets:
                    dist.all_reduce(net.master.grad, op=dist.ReduceOp.SUM, group=net._dist)
            out = [net.master.grad.clone() for net in dp_nets]
            for pool, saved in pools:
                pool.images.copy_(saved)
            return out
        grads = {"between": replay("between")}
        # the captured form is an offer, not a requirement: a runtime that cannot record or replay the collectives inside a
        # graph (an exception, on every rank alike) leaves the run on the between form
        # a rank decides locally whether its capture worked (an out-of-memory condition hits one rank, not all), so the
        # ranks AGREE on the outcome before any of them replays a graph that holds collectives: a rank that fell back
        # would sit in the between form's all-reduce while the others wait inside the captured one.
        failed, fatal = None, None
        try:
            forms["captured"] = self._capture_graphs(dp_nets, "captured")
        except Exception as e:      # noqa: BLE001
            if isinstance(e.__cause__, _HOST_LOGIC_ERRORS):      # a bug in the recipe's host code is not "not capturable":
                fatal = e                                        # raised on EVERY rank, after the agreement below
            failed = e
            forms.pop("captured", None)
            self._graph_broken = False
            self._set_external_host_state(True)
            torch.cuda.synchronize()
        # 1: captured here, 0: not capturable here, -1: host-logic error here. MIN over the ranks: every rank learns the worst
        # outcome BEFORE any of them raises or replays (a rank-local raise in front of this collective would leave the others
        # blocked in it until the timeout)
        flag = torch.tensor([-1.0 if fatal is not None else (0.0 if failed is not None else 1.0)], device=dp_nets[0].master.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if flag.item() < 0.0:
            if fatal is not None:
                raise fatal
            raise RuntimeError("data-parallel self-check: another rank hit a host-logic error while capturing the step")
        if flag.item() != 1.0 and failed is None:
            failed = RuntimeError("another rank could not capture the collectives")
            forms.pop("captured", None)
        if failed is None:
            grads["captured"] = replay("captured")
        for net in dp_nets:
            net.master.grad.zero_()
        if failed is not None:
            for pool, saved in pools:
                pool.images.copy_(saved)
            self.ddp_self_check = {"world": world, "forms_agree": False, "kept": "between", "error": str(failed)[:300]}
            self.logger.warning(f"data-parallel self-check over {world} rank(s): the collectives could not be captured into the "
                                f"step graph ({failed}); keeping the all-reduce between the two graphs")
            return forms["between"]
        worst, scale = 0.0, 0.0
        for ga, gb in zip(grads["between"], grads["captured"]):
            worst = max(worst, (ga - gb).abs().max().item())
            scale = max(scale, ga.abs().max().item())
        agree = worst == 0.0 if world <= 2 else worst <= 1e-5 * scale
        finite = all(torch.isfinite(g).all().item() for g in grads["captured"]) and scale > 0.0
        # the same sums on every rank: compare a checksum of the captured form's gradients across the group
        chk = torch.stack([g.double().sum() for g in grads["captured"]] + [g.double().abs().sum() for g in grads["captured"]])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        ranks_agree = bool(torch.equal(lo, hi))
        verdict = torch.tensor([1.0 if (agree and finite and ranks_agree) else 0.0], device=chk.device)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
        ok = verdict.item() == 1.0
        self.ddp_self_check = {"world": world, "forms_agree": bool(agree), "max_abs_diff": worst, "max_abs_grad": scale,
                               "ranks_agree": ranks_agree, "finite": bool(finite), "kept": "captured" if ok else "between"}
        (self.logger.info if ok else self.logger.warning)(
            f"data-parallel self-check over {world} rank(s): forms {'agree' if agree else 'DISAGREE'} (max |diff| {worst:.3e} "
            f"of max |grad| {scale:.3e}), ranks {'agree' if ranks_agree else 'DISAGREE'} -> keeping the "
            f"{'captured (overlapped) collectives' if ok else 'all-reduce between the two graphs'}")
        keep = forms.pop("captured" if ok else "between")
        forms.clear()
        return keep

    def backward(self, loss, optimizer, retain_graph=False, loss_id=0):
        # the root gradient is a constant kept for the run (loss.backward() alone fills a fresh one every time)
        unit = getattr(self, "_unit_grad", None)
        if unit is None or unit.device != loss.device or loss.dim() != 0 or loss.dtype != unit.dtype:
            if loss.dim() != 0 or (loss.is_cuda and torch.cuda.is_current_stream_capturing()):
                return loss.backward(retain_graph=retain_graph)
            unit = self._unit_grad = torch.ones((), dtype=loss.dtype, device=loss.device)
        loss.backward(gradient=unit, retain_graph=retain_graph)

    def parallelize_networks(self):
        if not torch.distributed.is_initialized():
            raise RuntimeError("Multi-GPU runs must be launched in distributed mode (torchrun / "
                               "torch.distributed.run), one process per GPU.")
        for name in self.networks.keys():
            self.networks[name].parallelize()

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            scheduler.step()

    def save_checkpoint(self, iter_idx):
        checkpoint = {}
        path = Path(self.output_dir) / f"checkpoints/{iter_idx}.pth"
        io.mkdirs(path.parent)
        for name, net in self.networks.items():
            checkpoint[name] = {k: v.cpu() for k, v in net.state_dict().items()}
        for name, optim in self.optimizers.items():      # reference saves only G and D (base.py:244-245)
            # per-parameter Adam state in the reference's layout (a reference run can resume from it and vice versa)
            checkpoint[f"optimizer_{name}"] = optim.reference_state_dict() if hasattr(optim, "reference_state_dict") \
                else optim.state_dict()
        torch.save(checkpoint, path)

    def load_networks(self, iter_idx):
        path = Path(self.output_dir).resolve() / f"checkpoints/{iter_idx}.pth"
        checkpoint = torch.load(path, map_location="cpu")
        self.logger.info(f"Loaded the checkpoint from `{path}`")
        for name in self.networks.keys():
            self.networks[name].load_state_dict(checkpoint[name])
        if self.is_train and self.conf[self.conf.mode].checkpointing.load_optimizers:
            for name, optim in self.optimizers.items():
                key = f"optimizer_{name}"
                if key not in checkpoint:
                    self.logger.info(f"{key}: not in the checkpoint; starting from scratch")
                elif _is_native_optimizer_state(checkpoint[key], optim):       # round-1 checkpoints: flat layout
                    optim.load_state_dict(checkpoint[key])
                elif hasattr(optim, "load_reference_state_dict"):
                    optim.load_reference_state_dict(checkpoint[key])           # raises on a shape / count mismatch
                else:
                    optim.load_state_dict(checkpoint[key])

    def set_requires_grad(self, networks, requires_grad=False):
        if not isinstance(networks, list):
            networks = [networks]
        for net in networks:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad

    def eval(self):
        for name in self.networks.keys():
            self.networks[name].eval()

    def infer(self, input):
        generator = "G" if "G" in self.networks.keys() else "G_AB"
        with torch.no_grad():
            return self.networks[generator].forward(input)

    def get_loggable_data(self):
        learning_rates = {f"lr_{name}": optim.param_groups[0]["lr"] for name, optim in self.optimizers.items()}
        return learning_rates, self.losses, self.visuals, self.metrics


_HOST_LOGIC_ERRORS = (IndexError, KeyError, TypeError, AttributeError, AssertionError, NameError, ValueError)


def _is_native_optimizer_state(state, optim):
    try:
        n_saved = sum(len(g["params"]) for g in state["param_groups"])
        n_here = sum(len(g["params"]) for g in optim.param_groups)
        return n_saved == n_here
    except (KeyError, TypeError):
        return False
