import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# every test writes its node id here before it starts (flush + fsync), and to stderr: a hard abort of the interpreter (SIGABRT from the
# HIP runtime, a GPU memory fault) then names the test that was running -- `-x -q` alone prints dots and a dead process prints nothing
BREADCRUMB = os.environ.get("JEN1_TEST_BREADCRUMB", os.path.join(ROOT, "gpurun_out", "last_test.txt"))
_crumb = {"n": 0, "fh": None, "trace": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "isolated: fault injection -- the body runs in a spawned process; collected last")


def _crumb_file():
    if _crumb["fh"] is None:
        try:
            os.makedirs(os.path.dirname(BREADCRUMB), exist_ok=True)
            _crumb["fh"] = open(BREADCRUMB, "a")
        except OSError:
            _crumb["fh"] = False
    return _crumb["fh"]


def _redirect_faulthandler():
    """pytest's faulthandler plugin dumps every thread's stack to stderr on a fatal signal: several KB that push the name of the running test
    out of any log tail. Send that dump to a file next to the breadcrumb instead; stderr then ends with the `[jen1-test N] <node id>` line of
    the test that died and whatever the HIP runtime printed."""
    if _crumb["trace"] is not None:
        return
    import faulthandler
    try:
        _crumb["trace"] = open(os.path.join(os.path.dirname(BREADCRUMB), "faulthandler.txt"), "a")
        faulthandler.enable(file=_crumb["trace"], all_threads=True)
    except OSError:
        _crumb["trace"] = False
    if os.environ.get("JEN1_ABORT_TRACE") == "1":
        # diagnostic: the NATIVE stack of a fatal signal (tools/native/abort_trace.c), printed before Python's own dump
        import ctypes
        so = os.path.join(ROOT, "tools", "native", "libabort_trace.so")
        if os.path.exists(so):
            tracer = ctypes.CDLL(so)
            tracer.jen1_abort_trace_install(os.path.join(os.path.dirname(BREADCRUMB), "native_stack.txt").encode())
            _crumb["tracer"] = tracer


def pytest_runtest_logstart(nodeid, location):
    _crumb["n"] += 1
    _crumb_file()
    _redirect_faulthandler()
    line = f"[jen1-test {_crumb['n']}] {nodeid}\n"
    if os.environ.get("JEN1_TEST_BREADCRUMB_STDERR", "1") != "0":
        # -q prints one dot per test; the id goes to the real stderr (not pytest's capture) so that it is in the tail of a crashed run
        try:
            os.write(2, line.encode())
        except OSError:
            pass
    fh = _crumb_file()
    if fh:
        fh.write(line)
        fh.flush()
        try:
            os.fsync(fh.fileno())
        except OSError:
            pass


def pytest_collection_modifyitems(config, items):
    """Fault-injection tests (marker `isolated`) go to the end of the run: whatever they do to the device or the process, every parity test
    has already reported."""
    tail = [it for it in items if it.get_closest_marker("isolated") is not None]
    if tail:
        head = [it for it in items if it.get_closest_marker("isolated") is None]
        items[:] = head + tail


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
