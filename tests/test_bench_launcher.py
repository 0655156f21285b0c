"""bench.py --gpus N must start N ranks itself when no launcher has: the parent spawns one child per
rank with the rendezvous environment set, relays rank 0's stdout, and fails when a rank fails."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

import bench


def _child(tmp_path, body):
    script = tmp_path / "child.py"
    script.write_text(textwrap.dedent(body))
    return [sys.executable, str(script)]


def test_spawns_one_child_per_rank(tmp_path):
    child = _child(tmp_path, """
        import json, os, sys
        r = os.environ["RANK"]
        open(os.path.join(sys.argv[1], "rank" + r + ".json"), "w").write(json.dumps(
            {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[2:]}))
    """)
    assert bench.spawn_ranks(4, [str(tmp_path), "--gpus", "4"], child=child) == 0
    seen = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(4)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2", "3"] and [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2", "3"]
    assert {s["WORLD_SIZE"] for s in seen} == {"4"} and {s["MASTER_ADDR"] for s in seen} == {"127.0.0.1"}
    assert len({s["MASTER_PORT"] for s in seen}) == 1 and all(s["argv"] == ["--gpus", "4"] for s in seen)


def test_fails_when_one_rank_fails_and_ends_the_others(tmp_path):
    child = _child(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)   # a rank waiting at a barrier for the failed one
    """)
    import time
    t0 = time.monotonic()
    assert bench.spawn_ranks(3, [], child=child) == 7
    assert time.monotonic() - t0 < 30


def test_only_rank0_reaches_stdout_and_gpus_flag_launches(tmp_path):
    # the whole command line: `bench.py --gpus 2` without WORLD_SIZE starts two ranks of itself; with
    # ZBRA_BENCH_DRY_RANK each rank only reports its place in the job (no GPU here)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ZBRA_BENCH_DRY_RANK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout  # what libraries print on stdout (gloo's peer count, RCCL's banner) must not reach it
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["rank"] == 0 and doc["config"]["partitioning"].startswith("columns x2")
    # the dry ranks went through the job's collectives (gloo): slowest rank's time, both ranks' units,
    # and one report per rank, in rank order
    assert doc["ranks_seen"] == 2 and doc["per_rank_device"] == ["dry:0", "dry:1"] and doc["per_rank_ms"] == [1.0, 2.0]
    assert doc["job_seconds"] == 0.002 and doc["job_units"] == 200.0


def _dry_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["ZBRA_BENCH_DRY_RANK"] = "1"
    env.update(extra)
    return env


def test_two_ranks_on_one_device_fail_the_job():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=_dry_env(ZBRA_BENCH_DRY_CLASH="1"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, (out.returncode, out.stderr)
    assert "ranks share a device" in out.stderr and not out.stdout.strip()


def test_single_rank_can_be_forced_through_the_collectives():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=_dry_env(ZBRA_BENCH_FORCE_DIST="1"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout
    doc = json.loads(out.stdout)
    assert doc["ranks_seen"] == 1 and doc["per_rank_device"] == ["dry:0"] and doc["job_units"] == 100.0


def test_shared_devices():
    from zbra_amd import sharding
    assert sharding.shared_devices(["a", "b", "c"]) == []
    assert sharding.shared_devices(["a", "b", "a", "b", "c"]) == ["a", "b"]
