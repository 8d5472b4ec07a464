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
    assert "parity" in line["extras_error"] and "correctness_failed" not in line
    r = run(0, 0.2, "ok")
    assert r.returncode == 0
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and "extras_error" not in json.loads(lines[0])


def test_guard_flags_a_failed_parity_gate_and_exits_non_zero():
    """A wrong number behind the headline is not an infrastructure failure: the line still gets out, flagged, and the status is 3."""
    script = SCRIPT.replace('g.fire("rank %%d: %%r" %% (int(sys.argv[1]), ex))', 'g.fire("rank %%d: %%r" %% (int(sys.argv[1]), ex), correctness=isinstance(ex, SystemExit))')
    r = subprocess.run([sys.executable, "-c", script % REPO, "0", "5", "throw"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["correctness_failed"] is True and "parity" in line["extras_error"] and line["value"] == 1.5


def _probe(cmd, **env):
    e = dict(os.environ, PAML_AMD_BENCH_LAUNCH_PROBE="1", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run(cmd, env=e, cwd=REPO, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment (how the driver starts the N = 1 bench) re-executes itself
    under torch.distributed.run with two ranks on 127.0.0.1; with the launcher around it (the documented convention) it does not."""
    bench = os.path.join(REPO, "bench.py")
    p = _probe([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p == {"launch_probe": True, "world": 2, "self_launched": True, "master_addr": "127.0.0.1"}
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = _probe([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), bench, "--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p["world"] == 2 and p["self_launched"] is False
    p = _probe([sys.executable, bench, "--gpus", "1", "--steps", "1", "--warmup", "0"])
    assert p["world"] == 1 and p["self_launched"] is False
