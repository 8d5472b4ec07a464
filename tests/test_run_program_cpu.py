"""The helper the reference-binding tests run the reference's binaries with (tests/test_reference_binding_gpu.py::run_program), on the CPU:
a program that ends comes back with its output; one that does not end within the limit is described (threads, output) in a warning,
started once more, and fails the test the second time instead of holding the suite."""
import sys
import warnings

import pytest

import test_reference_binding_gpu as rb


def test_a_program_that_ends_returns_its_output(tmp_path):
    r = rb.run_program([sys.executable, "-c", "import sys; sys.stdin.read(); print('lnL = -1.5')"], tmp_path, 3, limit=30)
    assert r.returncode == 0 and b"lnL = -1.5" in r.stdout


def test_a_program_that_does_not_end_is_started_once_more_and_then_fails(tmp_path):
    marker = tmp_path / "runs"
    prog = "import time; open(%r, 'a').write('x'); print('started', flush=True); time.sleep(60)" % str(marker)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with pytest.raises(AssertionError, match="did not end within 1 s"):
            rb.run_program([sys.executable, "-c", prog], tmp_path, 1, limit=1)
    assert marker.read_text() == "xx"                                        # two runs, no more
    ours = [str(x.message) for x in w if "started again after" in str(x.message)]
    assert len(ours) == 1 and "threads (tid comm wchan syscall state)" in ours[0]


def test_a_program_that_ends_the_second_time_passes(tmp_path):
    marker = tmp_path / "runs"
    prog = ("import os, time; first = not os.path.exists(%r); open(%r, 'a').write('x'); print('run', flush=True)\n"
            "if first: time.sleep(60)") % (str(marker), str(marker))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = rb.run_program([sys.executable, "-c", prog], tmp_path, 1, limit=2)
    assert r.returncode == 0 and marker.read_text() == "xx" and len([x for x in w if "started again after" in str(x.message)]) == 1
