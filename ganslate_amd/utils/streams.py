"""Stream / event helpers for the multi-stream training step.

hipStreamEndCapture walks the events that were recorded during the capture; an event destroyed before the capture ends
leaves a dangling entry there (segfault in ROCm 7.2 once a few such events exist). Events made through `new_event()` are
therefore kept alive until `release_events()`, which the graph runner calls after the capture has ended."""
import torch

_keep = []


def new_event():
    ev = torch.cuda.Event()
    if torch.cuda.is_current_stream_capturing():
        _keep.append(ev)
    return ev


def wait_stream(waiter, other):
    """waiter.wait_stream(other) with an event that outlives a running capture"""
    ev = new_event()
    ev.record(other)
    waiter.wait_event(ev)


def release_events():
    _keep.clear()
