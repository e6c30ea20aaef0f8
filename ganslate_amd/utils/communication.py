"""Process-group helpers (ganslate/utils/communication.py:17-116, 153-285 restated): one process per GPU,
rendez-vous from the torchrun / `--use_env` environment; backend "nccl" is RCCL over xGMI on ROCm. A gloo
backend is accepted for the CPU test-suite (GANSLATE_DIST_BACKEND=gloo)."""
import os

import torch
import torch.distributed as dist


def init_distributed():
    if os.environ.get("WORLD_SIZE", None):
        world = int(os.environ.get("WORLD_SIZE", 1))
        if world > 1:
            backend = os.environ.get("GANSLATE_DIST_BACKEND", "nccl")
            if backend == "nccl":
                torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
            if not dist.is_initialized():
                import datetime
                dist.init_process_group(backend=backend, init_method="env://",
                                        timeout=datetime.timedelta(minutes=10))
            synchronize()
        else:
            raise ValueError("Distributed ON but but running single process.")


def synchronize():
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dist.barrier()


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_local_rank() -> int:
    return int(os.environ["LOCAL_RANK"]) if dist.is_available() and dist.is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_backend_compatible_device():
    return torch.device("cuda" if dist.get_backend() == "nccl" else "cpu")


def shared_random_seed() -> int:
    """same seed on every rank (communication.py:101-116): rank 0's draw is broadcast"""
    seed = torch.randint(2 ** 31, (1,))
    if dist.is_available() and dist.is_initialized():
        seed = seed.to(get_backend_compatible_device())
        dist.broadcast(seed, 0)
    return int(seed)


def reduce_dict(d, average=True):
    """stacked reduce of a dict of 0-d tensors to rank 0 (communication.py:226-250) — ONE collective, called at
    logging time only (the reference's per-iteration timer reduces, C6, are deliberately not reproduced)."""
    world = get_world_size()
    if world < 2 or not d:
        return d
    names = sorted(d.keys())
    vals = torch.stack([torch.as_tensor(d[k], dtype=torch.float32).to(get_backend_compatible_device())
                        for k in names])
    dist.reduce(vals, dst=0)
    if dist.get_rank() == 0 and average:
        vals /= world
    return {k: v for k, v in zip(names, vals)}
