"""hipGraph capture, the one way every capture of this package is made.

Why this exists (root cause of the round-4 SIGABRT of ``pytest -m gpu``, reproduced by ``tools/repro_gc_capture.py``): on ROCm
``at::cuda::CUDAGraph::~CUDAGraph()`` (ATen/hip/HIPGraph.cpp:324 of torch 2.10) makes a HIP call that is "not permitted when stream is
capturing" (hipErrorStreamCaptureUnsupported) -- in EVERY capture error mode -- and throws from the destructor, i.e.
``std::terminate`` -> SIGABRT of the whole interpreter.  A dead graph reaches its destructor inside somebody else's capture when it sits in
a Python reference cycle (a stepper of a model that was dropped: plan <-> closures <-> stepper) and the cyclic garbage collector happens to
run between ``capture_begin`` and ``capture_end``; since torch 2.9 ``torch.cuda.graph.__enter__`` no longer calls ``gc.collect()`` itself
(``torch.compiler.config.force_cudagraph_gc`` is off).  Whether the collector fires inside the window depends on the allocation count of the
process up to then, which is why the suite passed under ``-v`` and died under ``-q``.

So: collect BEFORE the capture (dead graphs are destroyed while that is legal), keep the collector off inside it (reference counting still
frees what the recorded code itself drops; nothing this package records drops a graph), and check only this thread's calls against the capture
("thread_local": a process group's watchdog thread that queries events while we record must not fail the capture -- train.py found that one
in round 4)."""
import contextlib
import gc

import torch


@contextlib.contextmanager
def capture(graph: "torch.cuda.CUDAGraph", pool=None, stream=None, capture_error_mode: str = "thread_local"):
    """``with capture(g): <launches on torch.cuda.current_stream()>`` -- torch.cuda.graph with the collector handled as above."""
    gc.collect()
    was_on = gc.isenabled()
    gc.disable()
    try:
        kw = {}
        if pool is not None:
            kw["pool"] = pool
        if stream is not None:
            kw["stream"] = stream
        with torch.cuda.graph(graph, capture_error_mode=capture_error_mode, **kw):
            yield graph
    finally:
        if was_on:
            gc.enable()
