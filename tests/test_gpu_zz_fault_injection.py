"""-m gpu: fault injection and shared-device cases of the persistent deep-level launch (include/jen1_deep.h) and of graph capture.

Every body lives in tests/fault_cases.py and runs in a SPAWNED process with a model of its own; the test here judges the exit code (and
shows the child's output when it fails).  Marked ``isolated``: tests/conftest.py moves them behind every parity test of the run, whatever the
file order -- a case that kills its process, wedges a queue or leaves class-level scheduling state behind cannot take the parity evidence
with it (round 4: one abort 17 s into the run erased 529 results)."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.isolated]
HERE = os.path.dirname(os.path.abspath(__file__))
LIMIT_S = 420


def run_case(*args, expect_rc=0):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, os.path.join(HERE, "fault_cases.py"), *map(str, args)], capture_output=True, text=True,
                           timeout=LIMIT_S, env=env)
    except subprocess.TimeoutExpired as e:
        pytest.fail(f"fault case {args} did not finish within {LIMIT_S} s (killed):\n{(e.stdout or b'')[-2000:]}\n{(e.stderr or b'')[-4000:]}")
    if r.returncode != expect_rc:
        pytest.fail(f"fault case {args}: exit code {r.returncode} (wanted {expect_rc})\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-6000:]}")
    return r


def test_a_dying_case_fails_one_test_not_the_run():
    """the wrapper itself: a child that aborts is an exit code here, not the end of the pytest process"""
    r = run_case("abort", expect_rc=-6)
    assert "OK" not in r.stdout


def test_gc_inside_capture_window_does_not_abort():
    run_case("gc_in_capture")


@pytest.mark.parametrize("mode", ["f32", "bf16", "f32:long", "bf16:long"])
def test_nan_activations_flow_through_the_launch(mode):
    run_case("nan_flow", mode)


@pytest.mark.parametrize("which", ["deep", "long"])
def test_time_out_is_reported_and_cleared(which):
    run_case("time_out", which)


def test_forward_time_out_raises_through_the_public_call():
    run_case("forward_time_out")


@pytest.mark.parametrize("n", [2, 4])
def test_concurrent_persistent_launches_match_solo_runs(n):
    run_case("concurrent", n)
