"""world > 1 on the GPU box of the test tier, which has ONE GPU: every rank on GPU 0, the collective library replaced by the
shared-memory stand-in tests/shim/librccl_shim.so (PAML_AMD_RCCL_LIB; real RCCL refuses two ranks per device).  Everything
else is the production path: paml_amd_comm_init(world = 2, 3), the exchange step on the engine's collective stream with its two
slots of partial sums, `bench.py --gpus 2` under torch.distributed.run (gloo as the courier of the id), `pamlh_lnl --gpus 2`
with its fork / pipe hand-over.  The all-reduced results must have the bits of the one-engine run (the ranks' zero-padded
partial-sum arrays are added — exact — and totalled in one fixed order) and match the reference's goldens."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SHIM = os.path.join(HERE, "shim", "librccl_shim.so")
WORKER = os.path.join(HERE, "shim", "rank_worker.py")


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def shim_env(**extra):
    if not os.path.exists(SHIM):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "shim")])
    return dict(os.environ, PAML_AMD_RCCL_LIB=SHIM, **extra)


def real_env(**extra):
    """The production environment of a multi-GPU node: the real collective library (whatever librccl.so.1 resolves to), one GPU per rank."""
    env = dict(os.environ, PAML_AMD_WORKER_REAL_RCCL="1", **extra)
    env.pop("PAML_AMD_RCCL_LIB", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def run_ranks(world, case, tmp_path, real=False, **extra_env):
    xdir = tmp_path / ("%s_w%d%s%s" % (case, world, "_real" if real else "", "".join("_" + v for v in extra_env.values())))
    xdir.mkdir()
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(xdir), case], env=real_env(**extra_env) if real else shim_env(**extra_env),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0].decode())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d of %d failed:\n%s" % (r, world, outs[r][-3000:])
    return [json.load(open(xdir / ("out%d.json" % r))) for r in range(world)]


@pytest.mark.parametrize("case", ["codon_jit", "codon_k3", "nuc_fused", "aa20"])
def test_ranks_on_one_gpu_match_the_single_engine_bit_for_bit(case, tmp_path):
    one = run_ranks(1, case, tmp_path)[0]
    for world in (2, 3):
        res = run_ranks(world, case, tmp_path)
        for r in res:
            for key in ("eval", "eval_device", "eval_batch", "eval_again"):
                assert r[key] == one[key], (case, world, key, r[key], one[key])
            # branch-local sums: per-block partials at global block positions, same fixed order -> same bits as well
            assert r["eval_branch"] == one["eval_branch"], (case, world)
            # (a shard of a small data set may fall under the size at which every 16-pattern group gets a CU: the cooperative form
            #  of the interpreter kernel instead of the one-wave-per-group form — same accumulation order, the bits above are equal)
            interp = {"mfma64_coop": "mfma64_gather", "mfma64_coopjit": "mfma64_gather"}
            assert interp.get(r["kernel"], r["kernel"]) == interp.get(one["kernel"], one["kernel"])
            if "eval_adg" in one:
                # the rate chain over the sites: class likelihoods gathered over the ranks (x + 0 is exact), the chain on every rank
                assert r["eval_adg"] == one["eval_adg"], (case, world)
                # the BEB grid: the shards' per-grid-point sums are all-reduced (a different summation order than the one engine's:
                # agreement to rounding, not bits); each rank holds the posteriors of its own patterns
                lo, hi = r["shard"]
                assert abs(r["beb"]["ln_fx"] - one["beb"]["ln_fx"]) <= 1e-11 * abs(one["beb"]["ln_fx"])
                assert abs(r["beb"]["ln_fx_classes"] - one["beb"]["ln_fx"]) <= 1e-11 * abs(one["beb"]["ln_fx"])
                for key in ("pr_last", "mean_w", "sd_w"):
                    assert np.allclose(r["beb"][key], one["beb"][key][lo:hi], rtol=1e-9, atol=1e-13), (case, world, key)
                assert np.allclose(np.array(r["beb"]["post"]), np.array(one["beb"]["post"])[:, lo:hi], rtol=1e-9, atol=1e-13)
        assert res[0]["shard"][1] == res[1]["shard"][0]
    # the evaluations of the eval_device run differ from each other (different branch lengths) and the first equals eval
    assert one["eval_device"][0] == one["eval"] and len(set(one["eval_device"])) == len(one["eval_device"])
    assert ("eval_adg" in one) == (case != "codon_jit")      # (every case with rate or site classes)


def test_two_ranks_with_overlapping_pruning_kernels(tmp_path):
    """125 000 codon patterns per rank: the evaluations of the eval_device run alternate between two pruning streams on every rank
    while their exchange steps queue up on the collective stream — same bits as the single engine."""
    one = run_ranks(1, "codon_big", tmp_path)[0]
    res = run_ranks(2, "codon_big", tmp_path)
    for r in res:
        for key in ("eval", "eval_device", "eval_batch", "eval_again", "eval_branch", "kernel"):
            assert r[key] == one[key], (key, r[key], one[key])
    assert one["kernel"] == "mfma64_jit" and len(set(one["eval_device"])) == len(one["eval_device"])


def test_two_ranks_with_a_collective_that_needs_a_cu_and_waits_for_its_peers(tmp_path):
    """The stand-in's device mode: ncclAllReduce returns at once; a kernel on the collective stream — 512 threads, 64 KB of LDS, a CU of
    its own — publishes the rank's partial sums, spins until the peer's have arrived and 20 us have passed, and adds them up.  The
    persistent pruning kernels of both ranks (two pruning streams each) compete with it for the CUs.  Same bits as the one engine."""
    one = run_ranks(1, "codon_big", tmp_path)[0]
    res = run_ranks(2, "codon_big", tmp_path, PAML_AMD_SHIM_DEVICE_US="20")
    for r in res:
        for key in ("eval", "eval_device", "eval_batch", "eval_again", "eval_branch", "kernel"):
            assert r[key] == one[key], (key, r[key], one[key])


def test_bench_on_two_ranks_of_one_gpu(tmp_path):
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, one process per rank), but both ranks on
    GPU 0 (PAML_AMD_BENCH_ONE_GPU=1: gloo carries the id and the timing's barrier / max).  Strong scaling of the same patterns:
    lnL_hex equals the one-rank run's."""
    env = dict(shim_env(), PAML_AMD_BENCH_ONE_GPU="1")
    common = ["--steps", "4", "--warmup", "2", "--patterns", "40000", "--no-cpu-baseline", "--sweep-steps", "2"]
    r1 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--no-extras"] + common, env=env, cwd=REPO,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r1.returncode == 0, r1.stderr.decode()[-3000:]
    one = json.loads(r1.stdout.decode().strip().splitlines()[-1])
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2"] + common, env=env, cwd=REPO,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert r2.returncode == 0, r2.stderr.decode()[-3000:]
    lines = [ln for ln in r2.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["lnL_hex"] == one["lnL_hex"]
    assert two["config"]["patterns_rank0"] < 40000
    assert [s["model"] for s in two["sweep"]] == ["M0", "M1a", "M2a", "M7", "M8"]
    assert two["weak"]["patterns_per_gpu"] == 40000 and two["weak"]["value"] > 0
    assert two["ms_per_step_readback"] > 0
    rep = two["c5_replicas"]      # configs[4] at N > 1: one NSsites model per rank, independent replicas
    assert sorted(m[0] for r in rep["per_rank"] for m in r["models"]) == ["M0", "M1a", "M2a", "M7", "M8"] and all(r["error"] is None for r in rep["per_rank"])
    assert 0 < rep["seconds"] == max(r["seconds"] for r in rep["per_rank"])


def test_pamlh_lnl_on_two_ranks_of_one_gpu(tmp_path):
    """The C driver's --gpus path: the parent forks rank 1 before touching the GPU, hands the id over a pipe, both ranks read the same
    files and keep their shard; lnL is the reference's."""
    from paml_amd import hostlib
    g = helpers.load_golden("mtcdna_branch")      # 607 site patterns: three reduction chunks
    ctl = os.path.join(helpers.GOLDEN, "ctl", "mtcdna_branch.ctl")
    cmd = [hostlib.DRIVER_PATH, "codeml", ctl] + ["%.6f" % v for v in g["x"]]
    r1 = subprocess.run(cmd, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r1.returncode == 0, r1.stdout.decode()
    (tmp_path / "w2").mkdir()
    r2 = subprocess.run(cmd + ["--gpus", "2", "--devices", "0,0"], cwd=tmp_path / "w2", env=shim_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r2.returncode == 0, r2.stdout.decode()

    def lnl(txt):
        return txt.split("lnL  =")[1].split()[0]
    assert lnl(r2.stdout.decode()) == lnl(r1.stdout.decode())
    assert abs(float(lnl(r2.stdout.decode())) - g["lnL"]) <= 2e-6
    assert "sharded over 2 GPUs" in r2.stdout.decode()
    # more ranks than reduction chunks: every rank refuses, nobody is left waiting in a collective call
    r9 = subprocess.run(cmd + ["--gpus", "8", "--devices", "0,0,0,0,0,0,0,0"], cwd=tmp_path / "w2", env=shim_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r9.returncode != 0 and "at most" in r9.stdout.decode()


# ---- multi-GPU boxes: the same checks on the REAL collective library, one GPU per rank ------------------------------------------------
def gpus_here():
    from paml_amd import engine
    return engine.device_count()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("case", ["codon_big", "nuc_fused", "codon_k3"])
def test_ranks_on_real_rccl_one_gpu_each_have_the_bits_of_one_engine(world, case, tmp_path):
    """What the stand-in emulates on the one-GPU test tier, on hardware when there is more than one GPU: `world` ranks, GPU r for rank r,
    RCCL over xGMI.  lnL (eval, runs of eval_device on two pruning streams, eval_batch), the branch-local sums and, with classes, the
    rate chain are those of the one-engine run, bit for bit; the exchange step's timed events are there; the library is a real librccl."""
    if gpus_here() < world:
        pytest.skip("needs %d GPUs, this box shows %d" % (world, gpus_here()))
    one = run_ranks(1, case, tmp_path)[0]
    res = run_ranks(world, case, tmp_path, real=True)
    for r, out in enumerate(res):
        for key in ("eval", "eval_device", "eval_batch", "eval_again", "eval_branch"):
            assert out[key] == one[key], (case, world, r, key, out[key], one[key])
        if "eval_adg" in one:
            assert out["eval_adg"] == one["eval_adg"]
        assert out["device"] == r
        assert all(v == one["eval"] for v in out["run16"]), (out["run16"], one["eval"])
        assert out["comm_stats"]["n"] > 0 and out["comm_stats"]["exchange_us"] > 0
        assert "rccl" in os.path.basename(out["comm_library"]).lower() and "shim" not in out["comm_library"]
    assert [o["shard"][0] for o in res[1:]] == [o["shard"][1] for o in res[:-1]]


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_itself_on_real_gpus(world, tmp_path):
    """`python bench.py --gpus N` with no launcher around it, on N real GPUs: one line, the one-rank bits, the collective library named."""
    if gpus_here() < world:
        pytest.skip("needs %d GPUs, this box shows %d" % (world, gpus_here()))
    common = ["--steps", "4", "--warmup", "2", "--patterns", "200000", "--no-cpu-baseline", "--sweep-steps", "2"]
    env = real_env()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r1 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--no-extras"] + common, env=env, cwd=REPO,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r1.returncode == 0, r1.stderr.decode()[-3000:]
    one = json.loads(r1.stdout.decode().strip().splitlines()[-1])
    rn = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world)] + common, env=env, cwd=REPO,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert rn.returncode == 0, rn.stderr.decode()[-3000:]
    lines = [ln for ln in rn.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    many = json.loads(lines[0])
    assert many["n_gpus"] == world and many["lnL_hex"] == one["lnL_hex"] and "extras_error" not in many
    assert "rccl" in many["config"]["collective_library"].lower()
    assert many["exchange"]["evaluations"] > 0


def test_bench_launches_itself_on_two_ranks_of_one_gpu(tmp_path):
    """The same self-launch on the one-GPU tier: `python bench.py --gpus 2`, no torch.distributed.run in the command, both ranks on GPU 0
    through the stand-in (PAML_AMD_BENCH_ONE_GPU=1).  One line; strong scaling of the same patterns: the one-rank bits."""
    env = dict(shim_env(), PAML_AMD_BENCH_ONE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--steps", "3", "--warmup", "1", "--patterns", "40000", "--no-cpu-baseline", "--no-extras"]
    r1 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1"] + common, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r1.returncode == 0, r1.stderr.decode()[-3000:]
    r2 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"] + common, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r2.returncode == 0, r2.stderr.decode()[-3000:]
    lines = [ln for ln in r2.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    one, two = json.loads(r1.stdout.decode().strip().splitlines()[-1]), json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["lnL_hex"] == one["lnL_hex"]
    assert two["config"]["collective_library"].endswith("librccl_shim.so")
