"""`python3 bench.py --gpus N` as the round-end driver invokes it (no torch.distributed.run in front): bench.py starts its N
ranks itself, as child processes, before touching a GPU.  These tests exercise that launcher on a CPU-only host through
--dry-launch (each rank reports its rendezvous and the device it would select, the ranks meet once over gloo), and the
refusals (too few devices; RCCL with every rank on device 0)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def run(*argv, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                          text=True)


def json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.parametrize("n,extra", [(2, []), (3, ["--config", "stream"]), (2, ["--config", "e2e", "--all-on-device0", "--backend", "gloo"])])
def test_self_launch_starts_n_ranks(n, extra):
    p = run("--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-launch", *extra)
    assert p.returncode == 0, p.stderr
    line = json_line(p.stdout)
    assert line["dry_launch"] and line["n_gpus"] == n
    ranks = line["ranks"]
    assert [r["rank"] for r in ranks] == list(range(n)) and [r["local_rank"] for r in ranks] == list(range(n))
    assert [r["device"] for r in ranks] == ([0] * n if "--all-on-device0" in extra else list(range(n)))
    assert len({r["pid"] for r in ranks}) == n                       # n processes ...
    assert len({r["master"] for r in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")   # ... one rendezvous


def test_single_rank_needs_no_launcher():
    line = json_line(run("--dry-launch").stdout)
    assert line["n_gpus"] == 1 and line["ranks"][0]["rank"] == 0


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="host has two devices: the run would start")
def test_refuses_when_fewer_devices_than_ranks():
    p = run("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert p.returncode == 2 and "visible HIP devices" in p.stderr and p.stdout.strip() == ""


def test_refuses_rccl_with_all_ranks_on_one_device():
    p = run("--gpus", "2", "--all-on-device0")
    assert p.returncode == 2 and "gloo" in p.stderr


@pytest.mark.skipif(torch.cuda.device_count() > 0, reason="needs a host without a device: the ranks must fail, loudly")
def test_children_exit_code_is_relayed():
    p = run("--gpus", "2", "--all-on-device0", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert p.returncode != 0 and "needs a HIP device" in (p.stderr + p.stdout)


def test_started_by_a_launcher_with_the_wrong_world_is_refused():
    p = run("--gpus", "4", "--dry-launch", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert p.returncode != 0 and "WORLD_SIZE 2 does not match --gpus 4" in p.stderr
