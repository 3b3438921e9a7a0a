"""bench.py end to end on the GPU box: the single-rank JSON contract, and the N > 1 path (row partition, per-step exchange,
max-over-ranks timing) rehearsed with two ranks sharing the box's one GPU over gloo — exactly the launch line the driver uses for
N > 1, with `--backend gloo` in place of RCCL. Reduced sizes (`--small`): these are plumbing checks, not measurements."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch_two_ranks(extra_args, env):
    """The driver's launch line for N > 1 (`--master-addr 127.0.0.1 --master-port P`). The port is picked just before the launch; if somebody else takes it
    in between (EADDRINUSE: a race in the test, not a bench failure) the launch is repeated once with another port."""
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
               "bench.py", "--gpus", "2"] + extra_args
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr or attempt == 1:
            return r
    return r


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_rank_contract():
    r = subprocess.run([sys.executable, "bench.py", "--small", "--steps", "5", "--warmup", "2", "--path", "blocked"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1.5 and d["roofline"]["unit"] == "GB/s"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert d["config"]["spmv_path"] == "blocked" and "model" not in d["config"]
    assert d["max_rel_err"] <= 1e-10 and d["parity"]["ok"] is True                 # the bench checks its own y against the oracle's
    assert set(d["cpu_baseline"]["by_threads"]) >= {"1"} and "reproducible" in d["config"] and "traffic_source" in d["roofline"]
    ref = d["cpu_baseline"]["reference_library"]                                   # the reference's mm/ call sequence on oneMKL, where its runtime exists
    assert "skipped" in ref or (ref["kind"] == "reference" and ref["value"] > 0 and ref["agrees_with_oracle"] is True), ref


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls --gpus 1) starts its own ranks before touching the GPU."""
    env = dict(os.environ, G4S_BENCH_SAME_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--small", "--backend", "gloo", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "dist" in d["config"]["exchange"]


@pytest.mark.parametrize("workload,exchange", [("rmat", "dist"), ("lap7", "dist"), ("rmat", "allgather"), ("lap7", "allgather")])
def test_bench_two_ranks_rehearsal(workload, exchange):
    env = dict(os.environ, G4S_BENCH_SAME_DEVICE="1")
    r = _launch_two_ranks(["--steps", "4", "--warmup", "1", "--small", "--backend", "gloo", "--workload", workload, "--exchange", exchange, "--no-cpu-baseline"], env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["scaling"] == "strong"
    assert "2 rank(s)" in d["config"]["partition"] and ("packed ncclSend" if exchange == "dist" else "ncclAllGather") in d["config"]["exchange"]


@pytest.mark.parametrize("fail", ["wire:1", "pre:wire:1", "create:1"])
def test_bench_two_ranks_fall_back_together(fail):
    """One rank's set-up of the packed exchange fails — after the wiring's collectives, BEFORE them (the peer is about to enter a point-to-point round the
    failing rank never posts: the readiness agreement in front of every collective phase keeps it out), or already in the local create: every rank takes the
    library's all-gather exchange instead (it needs no wiring), none hangs."""
    env = dict(os.environ, G4S_BENCH_SAME_DEVICE="1", G4S_DIST_FAIL=fail)
    r = _launch_two_ranks(["--steps", "3", "--warmup", "1", "--small", "--backend", "gloo", "--no-cpu-baseline"], env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and "ncclAllGather" in d["config"]["exchange"]
    assert "falls back" in r.stderr
