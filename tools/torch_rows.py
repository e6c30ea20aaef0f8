#!/usr/bin/env python
"""Which lines of the recipe still launch torch's own kernels (run on the GPU box).

    python tools/torch_rows.py [--workload cyclegan|cut|pix2pix] [--batch 8]

Runs a few launch-by-launch steps (GS_STEP_GRAPH=0) under a TorchDispatchMode and prints, for every aten operator that
computes on device tensors (views, allocations and metadata operators excluded), how often per step and from which
ganslate_amd line it was called (backward passes: the line inside the autograd node). Everything the step computes is
meant to run in libganslate_hip's kernels; what shows up here is scalar algebra / glue left on torch."""
import argparse
import collections
import traceback
import os
import sys
from pathlib import Path

os.environ.setdefault("GS_STEP_GRAPH", "0")
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from ganslate_amd.utils.builders import build_gan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cyclegan", choices=["cyclegan", "cut", "pix2pix"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    if a.workload == "pix2pix":
        model = build_gan(bench.make_pix2pix_conf(1, 10 ** 6))
        shape = (1, 3, 256, 512)
    elif a.workload == "cut":
        model = build_gan(bench.make_cut_conf(a.batch, 256, 10 ** 6))
        shape = (a.batch, 3, 256, 256)
    else:
        model = build_gan(bench.make_conf(a.batch, 256, 10 ** 6))
        shape = (a.batch, 3, 256, 256)
    batch = {k: (torch.rand(shape, generator=g) * 2 - 1).to(dev) for k in ("A", "B")}

    def step():
        model.set_input(batch)
        model.optimize_parameters()
        model.update_learning_rate()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rows = collections.Counter()
    skip = ("empty", "view", "as_strided", "detach", "alias", "_unsafe_view", "reshape", "select", "slice", "expand",
            "permute", "transpose", "t.", "unsqueeze", "squeeze", "narrow", "split", "unbind", "_local_scalar_dense",
            "is_pinned", "lift_fresh", "set_", "resize_", "record_stream", "chunk", "unfold", "contiguous",
            "_reshape_alias", "zeros_like", "empty_like", "empty_strided", "new_empty", "result_type", "is_same_size")

    class Rows(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func).replace("aten.", "aten::")
            base = name.split("::")[-1]
            flat = [t for t in torch.utils._pytree.tree_leaves((args, kwargs, out)) if torch.is_tensor(t)]
            if any(t.is_cuda for t in flat) and not any(base.startswith(s) for s in skip):
                where = "?"
                for fr in reversed(traceback.extract_stack()):
                    if "ganslate_amd/" in fr.filename:
                        where = f"{fr.filename.split('ganslate_amd/')[-1]}:{fr.lineno} {fr.name}"
                        break
                big = max((t.numel() for t in flat if t.is_cuda), default=0)
                rows[(name, where, "scalar" if big <= 1 else f"<= {big} elements")] += 1
            return out

    with Rows():
        for _ in range(a.steps):
            step()
    torch.cuda.synchronize()
    print(f"# {a.workload}: aten operators computing on device tensors, per step ({a.steps} steps traced)")
    for (name, where, kern), n in sorted(rows.items(), key=lambda kv: -kv[1]):
        print(f"{n / a.steps:6.1f}  {name:28s} {where:80s} {kern}")


if __name__ == "__main__":
    main()
