"""Selection of the kernel backend. Production has exactly one: the HIP library. `set_ops` exists so the CPU
test-suite can run the executor against the op-level oracle (oracle/ops_ref.py); nothing in the package calls it."""
_ops = None


def get_ops():
    global _ops
    if _ops is None:
        from ...hip.ops import HipOps   # raises loudly if libganslate_hip.so or the GPU is missing
        _ops = HipOps()
    return _ops


def set_ops(ops):
    global _ops
    _ops = ops
