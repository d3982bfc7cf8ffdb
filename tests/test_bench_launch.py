"""CPU: `python bench.py --gpus N` launches its own ranks (torch.distributed.run, rendezvous on 127.0.0.1) when no launcher
set WORLD_SIZE, and still runs as one rank of an external launcher; rank 0 prints exactly one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_bench_self_launches_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3 and lines[0]["ranks_seen"] == 2


def test_bench_under_an_external_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2


def test_data_parallel_step_end_to_end_on_the_host():
    """not only the rendezvous -- the data-parallel contrastive step of `bench.py --workload step --no-graph`
    itself, self-launched at world size 2: two pairs of views, flat gradient buckets averaged by all-reduce (gloo here, RCCL on a GPU
    node: the same GradientBuckets code), both ranks' optimizers stepping, max-over-ranks timing, ONE JSON line from rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "step", "--no-graph", "--plumbing-cpu",
                        "--size", "32", "--steps", "2", "--warmup", "1", "--sustain", "0"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["devices"]) == 2 and line["steps"] == 2
    assert line["finite"] and line["value"] > 0 and "flat buckets" in line["config"]["workload"]


def test_sharded_sliding_window_end_to_end_on_the_host():
    """the N-rank path of BASELINE configs[1] (`bench.py --sw-volume V --gpus N`) end to end where no GPU node is
    available: self-launched at world size 2 on gloo, the windows of one volume dealt to the ranks as z-ordered runs, partial sums on
    each rank's sub-volume, neighbour slab exchange, normalised z-slabs, max-over-ranks timing, ONE JSON line from rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sw-volume", "48", "--size", "32", "--plumbing-cpu",
                        "--steps", "1", "--warmup", "0", "--sustain", "0"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and len(line["devices"]) == 2 and line["steps"] == 1
    assert line["finite"] and line["value"] > 0 and "windows dealt to the ranks" in line["config"]["workload"]
