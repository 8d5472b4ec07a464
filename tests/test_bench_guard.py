"""bench.py's HeadlineGuard (N > 1): the measured headline gets out whatever the blocks behind it do.  CPU only — the class is plain Python;
the N = 2 bench itself runs in tests/test_multirank_gpu.py."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, time
sys.path.insert(0, %r)
import bench
out = {"metric": "m", "value": 1.5, "n_gpus": 2}
g = bench.HeadlineGuard(out if int(sys.argv[1]) == 0 else None, 1, int(sys.argv[1]), float(sys.argv[2]))
mode = sys.argv[3]
if mode == "hang":
    out["sweep"] = [1, 2]
    time.sleep(30)
elif mode == "throw":
    try:
        raise SystemExit("parity")
    except (Exception, SystemExit) as ex:
        g.fire("rank %%d: %%r" %% (int(sys.argv[1]), ex))
else:
    assert g.finish()
    time.sleep(0.5)
    print(bench.json.dumps(out))
"""


def run(rank, seconds, mode):
    return subprocess.run([sys.executable, "-c", SCRIPT % REPO, str(rank), str(seconds), mode], capture_output=True, text=True, timeout=60)


def test_guard_prints_the_headline_when_the_extras_hang():
    r = run(0, 0.3, "hang")
    assert r.returncode == 0
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.5 and line["sweep"] == [1, 2] and "did not finish" in line["extras_error"]


def test_guard_other_ranks_leave_quietly():
    r = run(1, 0.3, "hang")
    assert r.returncode == 0 and r.stdout.strip() == ""


def test_guard_reports_an_exception_and_the_normal_end_prints_once():
    r = run(0, 5, "throw")
    assert r.returncode == 0
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "parity" in line["extras_error"]
    r = run(0, 0.2, "ok")
    assert r.returncode == 0
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and "extras_error" not in json.loads(lines[0])
