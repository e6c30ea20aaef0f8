#!/usr/bin/env python
"""Headline benchmark: training images/sec of the CycleGAN ResNet-9 256x256 bf16 step (BASELINE.json configs[1]:
horse2zebra hyper-parameters, batch 8 per GPU) on N MI355X of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N          # no WORLD_SIZE in the environment: starts the N ranks itself (child process)

A "step" = set_input + optimize_parameters of ganslate_amd.nn.gans.unpaired.CycleGAN (what the reference's
`t_comp` brackets, engines/trainer.py:56-60) on a synthetic batch already resident in HBM. One process per GPU;
gradients are averaged over RCCL inside the step. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# algorithmic work (BASELINE.md §3 / SURVEY.md §8d): conv MACs x 2 of one CycleGAN step per image pair
GFLOP_PER_IMAGE = 1289.9
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
# HBM bytes per launch of the three residual-conv kernels at the headline shape come from rocprofv3 PMC passes (bench.py
# cannot run the profiler on itself): 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes, gfx950 correction of
# MI355X_MICROARCH.md. tools/pmc_hbm_json.py writes them to this file from the passes of tools/pmc_step.sh; bench.py only
# READS it ({label: {"bytes_per_launch", "images_per_launch", ...}}); a missing file or label gives `traffic: null`.
HBM_JSON = ROOT / "profiles" / "r06_trunk_hbm.json"


def hbm_bytes(label, images_per_launch):
    """(bytes per launch, source) measured for this kernel form AT this launch size, or (None, None)"""
    try:
        rec = json.loads(HBM_JSON.read_text()).get(label)
    except (OSError, ValueError):
        return None, None
    if not rec or abs(rec.get("images_per_launch", 0) - images_per_launch) > 0.5:
        return None, None
    return rec["bytes_per_launch"], f"profiles/{HBM_JSON.name} ({rec.get('source', 'tools/pmc_step.sh')})"


def algorithmic_bytes(label, images, hw, C=256, taps=9, nets=1):
    """HBM bytes one launch of a residual-conv form has to move at least (bf16 activations [images, hw, C], fp32 dW; a twin
    launch carries the weights / weight gradients of `nets` = 2 networks):
    forward: x + y + W; fused data gradient: dY + y + g2 read, dX written, + W; weight-gradient pair: two (a, g) operand
    pairs read, dW [C][taps][C] fp32 written once"""
    act = int(images * hw * C * 2)
    w = C * taps * C * 2 * nets
    return {"rb_fwd": 2 * act + w, "rb_dgrad": 4 * act + w, "rb_wgrad_pair": 4 * act + 2 * w,
            "rb_wgrad": 2 * act + 2 * w}.get(label)


def make_pix2pix_conf(batch, n_iters):
    """BASELINE configs[2]: cityscapes_label2photo pix2pix.yaml:25-48 — Unet2D(num_downs 7, ngf 128, dropout) +
    PatchGAN2D(n_layers 4, 6 ch), 256x512 (a parity-test workload, measured with --workload pix2pix)"""
    from ganslate_amd.configs.config import Config
    from ganslate_amd.configs.omegalite import OmegaConf
    from ganslate_amd.configs.utils import init_config
    y = OmegaConf.create({
        "train": {
            "output_dir": "/tmp/ganslate_amd_bench", "cuda": True, "batch_size": batch,
            "n_iters": n_iters, "n_iters_decay": n_iters,
            "dataset": {"_target_": "ganslate.data.SyntheticImageDataset", "final_size": [256, 512]},
            "gan": {
                "_target_": "ganslate.nn.gans.paired.Pix2PixConditionalGAN",
                "generator": {"_target_": "ganslate.nn.generators.Unet2D", "num_downs": 7, "ngf": 128,
                              "use_dropout": True, "in_out_channels": {"AB": [3, 3]}},
                "discriminator": {"_target_": "ganslate.nn.discriminators.PatchGAN2D", "n_layers": 4,
                                  "in_channels": {"B": 6}},
                "optimizer": {"lr_D": 0.0001, "lr_G": 0.0002, "lambda_pix2pix": 30.0},
            },
            "metrics": {"discriminator_evolution": True},
        }})
    return init_config(y, Config)


def make_volume_conf(batch, size, n_iters, generator="resnet"):
    """BASELINE configs[4] shape family (3-D CycleGAN on 128^3 single-channel patches, conv3d implicit GEMM) with the
    reference's Resnet3D(9 blocks) + PatchGAN3D(3 layers) (33.2 TFLOP per pair)"""
    from ganslate_amd.configs.config import Config
    from ganslate_amd.configs.omegalite import OmegaConf
    from ganslate_amd.configs.utils import init_config
    y = OmegaConf.create({
        "train": {
            "output_dir": "/tmp/ganslate_amd_bench", "cuda": True, "batch_size": batch,
            "n_iters": n_iters, "n_iters_decay": n_iters,
            "dataset": {"_target_": "ganslate.data.SyntheticImageDataset", "image_channels": 1,
                        "final_size": [size, size, size]},
            "gan": {
                "_target_": "ganslate.nn.gans.unpaired.CycleGAN", "pool_size": 50,
                # vnet: the brats yaml networks (projects/brats_mri_sequence_translation/experiments/cyclegan.yaml)
                "generator": ({"_target_": "ganslate.nn.generators.Vnet3D", "use_memory_saving": False,
                               "use_inverse": False, "down_blocks": [2, 2, 3], "up_blocks": [3, 3, 3],
                               "in_out_channels": {"AB": [1, 1]}} if generator == "vnet" else
                              {"_target_": "ganslate.nn.generators.Resnet3D", "n_residual_blocks": 9,
                               "in_out_channels": {"AB": [1, 1]}}),
                "discriminator": {"_target_": "ganslate.nn.discriminators.PatchGAN3D",
                                  "n_layers": 2 if generator == "vnet" else 3, "in_channels": {"B": 1}},
                "optimizer": {"lambda_AB": 10.0, "lambda_BA": 10.0, "lambda_identity": 0, "proportion_ssim": 0,
                              "lr_D": 0.0002, "lr_G": 0.0002},
            },
            "metrics": {"discriminator_evolution": True, "ssim": False},
        }})
    return init_config(y, Config)


def make_cut_conf(batch, size, n_iters):
    """BASELINE configs[3]: horse2zebra CUT (PatchNCE) — ResNet-9 + PatchGAN-3 + the patch-feature MLP with the
    reference defaults (cut.py:16-40: nce_layers 0,4,8,12,16; 256 patches; lambda_nce_idt 0.5) and the horse2zebra
    learning rates"""
    from ganslate_amd.configs.config import Config
    from ganslate_amd.configs.omegalite import OmegaConf
    from ganslate_amd.configs.utils import init_config
    y = OmegaConf.create({
        "train": {
            "output_dir": "/tmp/ganslate_amd_bench", "cuda": True, "batch_size": batch,
            "n_iters": n_iters, "n_iters_decay": n_iters,
            "dataset": {"_target_": "ganslate.data.SyntheticImageDataset", "final_size": [size, size]},
            "gan": {
                "_target_": "ganslate.nn.gans.unpaired.CUT",
                "generator": {"_target_": "ganslate.nn.generators.Resnet2D", "n_residual_blocks": 9,
                              "in_out_channels": {"AB": [3, 3]}},
                "discriminator": {"_target_": "ganslate.nn.discriminators.PatchGAN2D", "n_layers": 3,
                                  "in_channels": {"B": 3}},
                "optimizer": {"lr_D": 0.0002, "lr_G": 0.0002},
            },
            "metrics": {"discriminator_evolution": True},
        }})
    return init_config(y, Config)


def make_conf(batch, size, n_iters):
    from ganslate_amd.configs.config import Config
    from ganslate_amd.configs.omegalite import OmegaConf
    from ganslate_amd.configs.utils import init_config
    y = OmegaConf.create({
        "train": {
            "output_dir": "/tmp/ganslate_amd_bench", "cuda": True, "batch_size": batch,
            "n_iters": n_iters, "n_iters_decay": n_iters,
            "dataset": {"_target_": "ganslate.data.SyntheticImageDataset", "final_size": [size, size]},
            "gan": {
                "_target_": "ganslate.nn.gans.unpaired.CycleGAN", "pool_size": 50,
                "generator": {"_target_": "ganslate.nn.generators.Resnet2D", "n_residual_blocks": 9,
                              "in_out_channels": {"AB": [3, 3]}},
                "discriminator": {"_target_": "ganslate.nn.discriminators.PatchGAN2D", "n_layers": 3,
                                  "in_channels": {"B": 3}},
                # projects/horse2zebra/experiments/default.yaml:43-49
                "optimizer": {"lambda_AB": 10.0, "lambda_BA": 10.0, "lambda_identity": 0, "proportion_ssim": 0,
                              "lr_D": 0.0002, "lr_G": 0.0002},
            },
            "metrics": {"discriminator_evolution": True, "ssim": True},
        }})
    return init_config(y, Config)


def cpu_baseline(size=256, steps=5):
    """The reference step restated in stock torch fp32 (oracle/torch_ref.py) on the host cores: BASELINE config 1
    shape (batch 1). Bounded sample: 1 warm-up + `steps` >= 5 timed steps; `cores` = the threads actually used."""
    import platform
    import torch
    from oracle.torch_ref import CycleGANStep
    cores = os.cpu_count() or 1
    model = CycleGANStep(seed=0)
    g = torch.Generator().manual_seed(1234)
    A = torch.rand(1, 3, size, size, generator=g) * 2 - 1
    B = torch.rand(1, 3, size, size, generator=g) * 2 - 1
    # BASELINE.md §4 plans torch.set_num_threads(os.cpu_count()); why the count is capped at 64 on the 256-thread hosts of
    # this pool (oversubscription, measured in round 5) is recorded in BASELINE.md §4 — the line reports only what this run did
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    model.step(A, B)                         # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        model.step(A, B)
    dt = (time.perf_counter() - t0) / steps
    cpu = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), cpu)
    except OSError:
        pass
    return {"value": round(1.0 / dt, 4), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": f"oracle/torch_ref.CycleGANStep (stock torch fp32), batch 1, {size}x{size}, 1 warm-up + "
                      f"{steps} timed steps, {threads} threads of {cores} host cpus ({cpu})"}


# BASELINE configs[2..4] (per-GPU shapes, SURVEY.md §8 GFLOP per unit): timed for a few steps behind the headline so that the
# driver's line shows them too
SECONDARY = {
    "pix2pix": {"config": "cityscapes pix2pix (BASELINE configs[2]): U-Net(7,128) + PatchGAN-4, 256x512, batch 1",
                "unit": "img/s", "gflop_per_unit": 371.5},
    "cut": {"config": "horse2zebra CUT (BASELINE configs[3]): ResNet-9 + PatchGAN-3 + PatchNCE, 256x256, batch 8",
            "unit": "img/s", "gflop_per_unit": 1389.0},
    "brats": {"config": "brats 3-D CycleGAN (BASELINE configs[4]): Vnet3D + PatchGAN3D-2, 128^3 patches, batch 1",
              "unit": "vol/s", "gflop_per_unit": 27692.0},
}


def _trunk_roofline(model, step, batch, size, timing_steps=3):
    """`roofline` block of a 2-D ResNet-9 recipe: HIP events around every launch of the three residual-conv forms in a few
    launch-by-launch single-stream steps (the way main() times the headline's), the form with the largest per-step total"""
    import torch
    from ganslate_amd.nn.native import backend
    ops = backend.get_ops()

    def select(kind, spec, flag):
        if kind == "gconv" and spec.T == 9 and spec.Ci == 256 and spec.Co == 256 and spec.si == 1 and spec.so == 1:
            return "rb_dgrad" if (flag or spec.border == "zero") else "rb_fwd"
        if kind == "wgrad" and spec.T == 9 and spec.P == 256 and spec.Q == 256 and spec.si == 1:
            return "rb_wgrad_pair" if flag else "rb_wgrad"
        return None
    graphed = bool(getattr(model, "_graph", None) is not None and model.step_graph_enabled)
    ops.enable_kernel_timing(select)
    model.step_graph_enabled = False
    side = os.environ.get("GS_SIDE_STREAM")
    os.environ["GS_SIDE_STREAM"] = "0"
    try:
        for _ in range(timing_steps):
            step()
        torch.cuda.synchronize()
    finally:
        ops.disable_kernel_timing()
        if side is None:
            del os.environ["GS_SIDE_STREAM"]
        else:
            os.environ["GS_SIDE_STREAM"] = side
        model.step_graph_enabled = graphed
    res, imgs = ops.kernel_timing_result(), ops.kernel_timing_images()
    hw = (size // 4) ** 2
    rows = {}
    for label, (n, ms) in res.items():
        if not n:
            continue
        nimg = imgs.get(label) or batch
        flop = 2.0 * hw * nimg * 256 * 2304 * (2 if label == "rb_wgrad_pair" else 1)
        rows[label] = {"launches_per_step": round(n / timing_steps, 1), "images_per_launch": round(nimg, 1), "avg_ms": round(ms, 4),
                       "ms_per_step": round(n / timing_steps * ms, 3), "tflops": round(flop / (ms * 1e-3) / 1e12, 1)}
    if not rows:
        return None
    dom = max(rows, key=lambda k: rows[k]["ms_per_step"])
    names = {"rb_fwd": "hconvw_kernel<9> (forward)", "rb_dgrad": "hconvw_kernel<9, RING> (fused data gradient)",
             "rb_wgrad_pair": "hwgrad_wide_kernel<9> (two passes)", "rb_wgrad": "hwgrad_wide_kernel<9> (one pass)"}
    dimg = rows[dom]["images_per_launch"]
    return {"bound": "mfma", "achieved": rows[dom]["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(rows[dom]["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": None,
            "algorithmic_bytes": algorithmic_bytes(dom, dimg, hw),
            "kernel": f"{names[dom]}: 3x3 256->256 reflect conv over {int(dimg)} images per launch", "avg_ms": rows[dom]["avg_ms"],
            "launches_timed": res[dom][0], "residual_conv_kernels": rows,
            "timed_in": f"{timing_steps} launch-by-launch single-stream steps behind the timed region"}


def run_secondary(dev, steps=10, warmup=4):
    """value / ms_per_step / fraction of the dense bf16 MFMA peak of the other BASELINE workloads on this GPU, `steps` timed
    iterations each after `warmup` (the second iteration captures the step graph). Same step definition as the headline."""
    import gc
    import torch
    from ganslate_amd.utils.builders import build_gan
    out = {}
    for name, meta in SECONDARY.items():
        try:
            out[name] = _run_secondary_one(dev, name, meta, steps, warmup)
        except Exception as exc:          # noqa: BLE001  (one workload failing must not take the others, or the line, with it)
            out[name] = {"config": meta["config"], "error": f"{type(exc).__name__}: {exc}"[:500]}
        gc.collect()
        torch.cuda.empty_cache()
    return out


def _run_secondary_one(dev, name, meta, steps, warmup):
    import torch
    from ganslate_amd.utils.builders import build_gan
    out = {}
    g = torch.Generator().manual_seed(4321)
    if name == "pix2pix":
        batch, shape, conf = 1, (1, 3, 256, 512), make_pix2pix_conf(1, 10 ** 6)
    elif name == "cut":
        batch, shape, conf = 8, (8, 3, 256, 256), make_cut_conf(8, 256, 10 ** 6)
    else:
        batch, shape, conf = 1, (1, 1, 128, 128, 128), make_volume_conf(1, 128, 10 ** 6, "vnet")
    model = build_gan(conf)
    data = {"A": (torch.rand(shape, generator=g) * 2 - 1).to(dev), "B": (torch.rand(shape, generator=g) * 2 - 1).to(dev)}

    def step():
        model.set_input(data)
        model.optimize_parameters()
        model.update_learning_rate()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = {k: float(v.detach()) for k, v in model.losses.items() if v is not None}
    assert all(v == v and abs(v) < 1e6 for v in losses.values()), f"{name}: non-finite losses {losses}"
    value = batch * steps / dt
    out[name] = {"config": meta["config"], "value": round(value, 3), "unit": meta["unit"], "steps": steps,
                 "ms_per_step": round(1e3 * dt / steps, 3),
                 "step_tflops": round(value * meta["gflop_per_unit"] / 1e3, 1),
                 "step_mfma_frac": round(value * meta["gflop_per_unit"] / 1e3 / PEAK_BF16_TFLOPS, 4)}
    if name == "cut":
        out[name]["roofline"] = _trunk_roofline(model, step, batch, shape[-1])
    del model, data, step
    return out[name]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE config: 8)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--workload", default="cyclegan", choices=["cyclegan", "pix2pix", "cut", "cyclegan3d", "brats", "revgan"],
                    help="cyclegan = the headline (BASELINE configs[1]); pix2pix = configs[2] (batch 1, 256x512); "
                         "cut = configs[3] (PatchNCE, batch 8, 256x256); "
                         "cyclegan3d = 3-D CycleGAN on 128^3 volumes with Resnet3D (configs[4] shape, batch 1); "
                         "brats = configs[4] with the brats yaml's own networks (Vnet3D + PatchGAN3D-2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short timed runs of the other BASELINE configs attached to the headline line as `secondary`")
    args = ap.parse_args()

    # a timed kernel must never be one that was told to skip work: the ablation switches of round 1 are compiled out
    # of the library; refuse to run if somebody still sets them, or any other kernel "variant" selector
    bad = [k for k in os.environ if k.startswith("GS_") and (k.endswith("_VARIANT") or k.endswith("_ABL"))]
    if bad:
        raise SystemExit(f"bench.py refuses to run with {bad} set: ablation / variant switches are not benchmarks")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start one rank per GPU as a CHILD process (torch.distributed.run) before this
        # process has touched the GPU, and exit with its code (a process that has initialised the GPU must not exec)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    # The ONE JSON line must be the only thing this process puts on stdout: RCCL prints a version block to stdout when its
    # communicator is torn down (after everything else), which would follow the JSON line of a data-parallel run. File
    # descriptor 1 is pointed at stderr for the whole run and the JSON line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist
    from ganslate_amd.utils import communication
    from ganslate_amd.utils.builders import build_gan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    if world > 1:
        communication.init_distributed()
    elif os.environ.get("GS_FORCE_DDP") == "1":       # the data-parallel code path with a 1-rank RCCL group
        os.environ.setdefault("LOCAL_RANK", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29571", rank=0, world_size=1)
    dev = torch.device(f"cuda:{local_rank}")

    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234 + rank)
    if args.workload == "pix2pix":
        args.no_cpu_baseline = args.no_kernel_timing = True
        if args.batch == 8:
            args.batch = 1
        model = build_gan(make_pix2pix_conf(args.batch, 10 ** 6))
        shape = (args.batch, 3, 256, 512)
    elif args.workload == "cut":
        args.no_cpu_baseline = args.no_kernel_timing = True
        model = build_gan(make_cut_conf(args.batch, args.size, 10 ** 6))
        shape = (args.batch, 3, args.size, args.size)
    elif args.workload in ("cyclegan3d", "brats"):
        args.no_cpu_baseline = True
        if args.workload == "brats":
            args.no_kernel_timing = True
        if args.batch == 8:
            args.batch = 1
        if args.size == 256:
            args.size = 128
        model = build_gan(make_volume_conf(args.batch, args.size, 10 ** 6,
                                           "vnet" if args.workload == "brats" else "resnet"))
        shape = (args.batch, 1, args.size, args.size, args.size)
    elif args.workload == "revgan":
        # projects/brats_mri_sequence_translation/experiments/revgan.yaml: RevGAN with Piresnet3D(depth 5, 32 channels,
        # memory saving + inverse) and PatchGAN3D(2 layers), lambda 5 / 5, lr_G 4e-4; 32 x 176 x 176 patches in the yaml
        args.no_cpu_baseline = args.no_kernel_timing = True
        if args.batch == 8:
            args.batch = 1
        from ganslate_amd.configs.omegalite import OmegaConf
        from ganslate_amd.configs.config import Config
        from ganslate_amd.configs.utils import init_config
        y = OmegaConf.create({"train": {
            "output_dir": "/tmp/ganslate_amd_bench", "cuda": True, "batch_size": args.batch, "n_iters": 10 ** 6,
            "n_iters_decay": 10 ** 6,
            "dataset": {"_target_": "ganslate.data.SyntheticImageDataset", "image_channels": 1,
                        "final_size": [32, 176, 176]},
            "gan": {"_target_": "ganslate.nn.gans.unpaired.RevGAN", "pool_size": 50,
                    "generator": {"_target_": "ganslate.nn.generators.Piresnet3D", "use_memory_saving": True,
                                  "use_inverse": True, "depth": 5, "in_out_channels": {"AB": [1, 1]}},
                    "discriminator": {"_target_": "ganslate.nn.discriminators.PatchGAN3D", "n_layers": 2,
                                      "in_channels": {"B": 1, "A": 1}},
                    "optimizer": {"lambda_AB": 5.0, "lambda_BA": 5.0, "lambda_identity": 0, "proportion_ssim": 0,
                                  "lr_D": 0.0002, "lr_G": 0.0004}},
            "metrics": {"discriminator_evolution": True, "ssim": False}}})
        model = build_gan(init_config(y, Config))
        shape = (args.batch, 1, 32, 176, 176)
    else:
        model = build_gan(make_conf(args.batch, args.size, 10 ** 6))
        shape = (args.batch, 3, args.size, args.size)
    batch = {"A": (torch.rand(shape, generator=g) * 2 - 1).to(dev), "B": (torch.rand(shape, generator=g) * 2 - 1).to(dev)}

    def step():
        model.set_input(batch)
        model.optimize_parameters()
        model.update_learning_rate()

    for _ in range(args.warmup):
        step()

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        model.reduce_timing = []      # two-graph data-parallel path: events around the all-reduce between the graphs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_dt = time.perf_counter() - t0          # time the host needed to enqueue the K steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ops = next(iter(model.networks.values())).ops
    graphed = bool(getattr(model, "_graph", None) is not None and model.step_graph_enabled)
    timing = None
    if not args.no_kernel_timing:
        # HIP events around every launch of the three residual-block 3x3 conv forms (forward, fused data gradient,
        # merged weight-gradient pair) on the stream they are launched on (torch's current stream). A replayed hipGraph
        # cannot hold event records, so when the timed region ran as graph replays the events are taken in a few
        # launch-by-launch, single-stream steps of the same workload right after it (same kernels, same arguments, no
        # other stream sharing the chip — which is also how rocprofv3's kernel trace runs them; profiles/ holds its
        # averages for the same kernels). The kernel with the largest per-step total goes into `roofline`.
        rb_taps = 27 if args.workload == "cyclegan3d" else 9

        def select(kind, spec, flag):
            if kind == "gconv" and spec.T == rb_taps and spec.Ci == 256 and spec.Co == 256 and spec.si == 1 and spec.so == 1:
                return "rb_dgrad" if (flag or spec.border == "zero") else "rb_fwd"
            if kind == "wgrad" and spec.T == rb_taps and spec.P == 256 and spec.Q == 256 and spec.si == 1:
                return "rb_wgrad_pair" if flag else "rb_wgrad"
            return None
        timing = ops.enable_kernel_timing(select)
        if graphed:
            model.step_graph_enabled = False
        side = os.environ.get("GS_SIDE_STREAM")
        os.environ["GS_SIDE_STREAM"] = "0"       # one stream: a launch's duration is its own, not a share of the chip
        timing_steps = min(args.steps, 5)
        for _ in range(timing_steps):
            step()
        torch.cuda.synchronize()
        ops.disable_kernel_timing()
        if side is None:
            del os.environ["GS_SIDE_STREAM"]
        else:
            os.environ["GS_SIDE_STREAM"] = side
        if graphed:
            model.step_graph_enabled = True

    losses = {k: float(v.detach()) for k, v in model.losses.items() if v is not None}
    assert all(v == v and abs(v) < 1e6 for v in losses.values()), f"non-finite losses: {losses}"

    if rank == 0 and args.workload == "pix2pix":
        value = args.batch * world * args.steps / dt
        emit({"metric": "training images/sec, Pix2Pix U-Net(7,128)+PatchGAN-4 256x512 bf16",
                          "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "synthetic", "config": {"workload": "cityscapes pix2pix (BASELINE configs[2])",
                                                          "global_batch": args.batch * world},
                          "step_tflops": round(value * 371.5 / 1e3, 1),
                          "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 3), "step_graph": graphed})
    elif rank == 0 and args.workload == "cut":
        value = args.batch * world * args.steps / dt
        emit({"metric": "training images/sec, CUT ResNet-9 + PatchGAN-3 + PatchNCE 256x256 bf16",
                          "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "synthetic",
                          "config": {"workload": f"horse2zebra CUT (BASELINE configs[3]), {args.size}x{args.size}, "
                                                 f"batch {args.batch} per GPU, nce_layers 0/4/8/12/16, 256 patches",
                                     "global_batch": args.batch * world, "parallelism": f"dp{world}"},
                          "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 3), "step_graph": graphed})
    elif rank == 0 and args.workload == "revgan":
        value = args.batch * world * args.steps / dt
        emit({"metric": "training volumes/sec, RevGAN Piresnet3D(5, 32) + PatchGAN3D-2 32x176x176 bf16",
                          "value": round(value, 3), "unit": "vol/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "synthetic",
                          "config": {"workload": "brats revgan.yaml networks and patch size (not a BASELINE configuration), "
                                                 "activation recompute on", "global_batch": args.batch * world,
                                     "parallelism": f"dp{world}"},
                          "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 3), "step_graph": graphed})
    elif rank == 0 and args.workload in ("cyclegan3d", "brats"):
        value = args.batch * world * args.steps / dt
        vnet = args.workload == "brats"
        # SURVEY.md §8 rows a15 / a16 at 128^3
        tflop_per_pair = (27.7 if vnet else 33.2) * (args.size / 128.0) ** 3
        nets = "Vnet3D(16; 2,2,3 / 3,3,3) + PatchGAN3D-2" if vnet else "Resnet3D-9 + PatchGAN3D-3"
        out = {"metric": f"training volumes/sec, 3-D CycleGAN {nets} bf16",
               "value": round(value, 4), "unit": "vol/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic",
               "config": {"workload": f"3-D CycleGAN {args.size}^3 single-channel patches (BASELINE configs[4] "
                                      f"shape), {nets}, lsgan, lambda 10/10",
                          "global_batch": args.batch * world, "parallelism": f"dp{world}"},
               "step_tflops": round(value * tflop_per_pair, 1),
               "step_mfma_frac": round(value * tflop_per_pair / (PEAK_BF16_TFLOPS * world), 4),
               "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 3), "step_graph": graphed}
        if timing is not None:
            n, ms = ops.kernel_timing_result().get("rb_fwd", (0, 0.0))
            vox = (args.size // 4) ** 3
            flop = 2.0 * vox * args.batch * 256 * 6912
            tf = flop / (ms * 1e-3) / 1e12 if n else 0.0
            out["roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "traffic": None,
                               "kernel": "gconv_kernel (3x3x3 256->256 replicate conv, M=%d N=256 K=6912)"
                                         % (vox * args.batch),
                               "launches_timed": n, "avg_ms": round(ms, 4)}
        emit(out)
    elif rank == 0:
        images = args.batch * world * args.steps
        value = images / dt
        out = {
            "metric": "training images/sec, CycleGAN ResNet-9 256x256 bf16", "value": round(value, 2),
            "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"horse2zebra CycleGAN ResNet-9 + PatchGAN-3, {args.size}x{args.size}, "
                                   f"batch {args.batch} per GPU, lsgan, lambda 10/10, pool 50, Adam(2e-4, 0.5), "
                                   "SSIM + D-output metrics on",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 3), "step_graph": graphed,
            "step_tflops": round(value * GFLOP_PER_IMAGE / 1e3, 1),
            "step_mfma_frac": round(value * GFLOP_PER_IMAGE / 1e3 / (PEAK_BF16_TFLOPS * world), 4),
        }
        if world == 1:
            out["ddp_path"] = "single process: no gradient reduction"
            out["ddp_self_check"] = None
        if world > 1:
            rt = getattr(model, "reduce_timing", None) or []
            # the reduction is NOT overlapped on this path: everything between the end of the backward graph and the start of
            # the optimiser graph that the stream spends inside RCCL (rank 0's view)
            out["exposed_allreduce_ms_per_step"] = round(sum(a.elapsed_time(b) for a, b in rt) / max(len(rt), 1), 3) \
                if rt else None
            out["ddp_path"] = (("graph(forward + backward with the bucketed RCCL all-reduces captured inside it, overlapped "
                                "with the remaining backward) | graph(Adam)")
                               if getattr(model, "_graph_collectives", False) else
                               ("graph(forward + backward) | one RCCL all-reduce per network on the flat gradient | "
                                "graph(Adam)")) if graphed else \
                "launch by launch: bucketed all-reduce overlapped with the last backward pass"
            # both forms were built and compared on the second iteration (BaseGAN._ddp_self_check); None: a form was forced
            out["ddp_self_check"] = getattr(model, "ddp_self_check", None)
            out["ddp_form_requested"] = getattr(model, "ddp_form_requested", None)   # "0" (default, world > 1) | "1" | "auto"
        if timing is not None:
            res = ops.kernel_timing_result()
            imgs = ops.kernel_timing_images()                    # images per launch: a twin launch covers both generators
            hw = (args.size // 4) ** 2
            names = {"rb_fwd": "hconvw_kernel<9> (forward, halo-resident)",
                     "rb_dgrad": ("hconvw_kernel<9, RING> (data gradient on the unpadded domain, reflect ring "
                                  "folded in-launch, fused norm-backward reduction)" if (ops.get_option("hconvw_ring") and rb_taps == 9)
                                  else "gconv_kernel<288, 128> (data gradient on the padded domain + fused norm-backward "
                                       "reduction)"),
                     "rb_wgrad_pair": "hwgrad_wide_kernel<9> (weight gradient of two backward passes in one launch)",
                     "rb_wgrad": "hwgrad_wide_kernel<9> (weight gradient, one pass)"}
            kernels = {}
            for label, (n, ms) in res.items():
                nimg = imgs.get(label) or args.batch
                # 2*M*N*K of one 3x3 256->256 conv over the launch's images (x 2 operand pairs for a merged weight gradient)
                flop = 2.0 * hw * nimg * 256 * 2304 * (2 if label == "rb_wgrad_pair" else 1)
                kernels[label] = {"kernel": names[label], "launches_per_step": round(n / timing_steps, 1),
                                  "images_per_launch": round(nimg, 1),
                                  "avg_ms": round(ms, 4), "ms_per_step": round(n / timing_steps * ms, 3),
                                  "gflop_per_launch": round(flop / 1e9, 2),
                                  "tflops": round(flop / (ms * 1e-3) / 1e12, 1) if n else 0.0,
                                  "frac": round(flop / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if n else 0.0}
            dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
            dimg = kernels[dom]["images_per_launch"]
            hbm, hbm_src = hbm_bytes(dom, dimg) if args.size == 256 else (None, None)
            out["roofline"] = {"bound": "mfma", "achieved": kernels[dom]["tflops"], "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": kernels[dom]["frac"], "traffic": hbm,
                               "traffic_source": hbm_src,
                               "traffic_note": "PMC bytes of the same kernel at the same launch size, measured on another box of "
                                               "the pool than this run's `achieved` (HIP events here)" if hbm is not None else None,
                               "algorithmic_bytes": algorithmic_bytes(dom, dimg, hw, nets=2 if dimg > args.batch else 1),
                               "kernel": f"{names[dom]}: 3x3 256->256 reflect conv, M={int(hw * dimg)} N=256 K=2304"
                                         + (" x 2 passes" if dom == "rb_wgrad_pair" else "")
                                         + (f" (twin launch: {int(dimg)} images = both generators' batch {args.batch})"
                                            if dimg > args.batch else ""),
                               "dominant_of": "the three residual-conv launch forms, by per-step total of their own "
                                              "HIP-event timings",
                               "launches_timed": res[dom][0], "avg_ms": kernels[dom]["avg_ms"],
                               "timed_in": "%d launch-by-launch single-stream steps right after the timed region%s"
                                           % (timing_steps, " (which ran as hipGraph replays)" if graphed else ""),
                               # an event pair brackets the launch AND its dispatch gap: 2-3 us above the kernel-only
                               # duration rocprofv3 reports for the same kernel (profiles/r02_step_kernel_stats_graph_v4.txt)
                               "event_bracket_overhead_us": "2-3"}
            out["residual_conv_kernels"] = kernels
        # (the headline line is what the driver reads: a failure in an attached measurement is reported in its place, never
        # instead of the line)
        if world == 1 and not args.no_secondary and args.batch == 8 and args.size == 256:
            del model
            try:
                out["secondary"] = run_secondary(dev)
            except Exception as exc:      # noqa: BLE001
                out["secondary"] = {"error": f"{type(exc).__name__}: {exc}"[:500]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.size)
            except Exception as exc:      # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"[:500]}
        emit(out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
