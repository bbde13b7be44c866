"""The N > 1 encoder layouts under a real process group on the GPU box: WORLD ranks (child processes, backend gloo, all
on the one MI355X of the box) run every layout of ``set_row_shard`` -- replicated, row-sharded with an all-gather per
layer, and the single all-gather of [X_node | Z] -- through ``propagate()`` and the pair stage and compare with the
unsharded result bit for bit (tests/dist_gpu_worker.py).  RCCL over xGMI needs more than one GPU, which the build loop
does not have; what this pins is everything around the collective: row blocks (even and ragged), the order of the
per-layer exchanges, the Z / Y hand-over of the single-gather layout, pair shards."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,ragged", [(2, False), (3, True)])
def test_encoder_layouts_under_a_process_group(world, ragged):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LPF_DIST_BACKEND="gloo", LPF_TEST_RAGGED="1" if ragged else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_rccl_world_of_one():
    """backend "nccl" (= RCCL), one rank, collectives forced: librccl is loaded and ``allgather_rows`` /
    ``measure_allgather_gbps`` / ``max_over_ranks`` / every ``set_row_shard`` layout run the library's collective
    calls on the device (tests/rccl_world1_worker.py).  The exchange over xGMI itself needs more than one GPU."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    assert "rccl world-1 ok" in p.stdout, p.stdout


def _bench(args, extra_env=None, timeout=900):
    env = dict(os.environ, LPF_DIST_BACKEND="gloo", LPF_LOCAL_DEVICE="0")
    env.pop("WORLD_SIZE", None)      # bench.py starts its own ranks
    env.pop("RANK", None)
    env.update(extra_env or {})
    before = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    after = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    left = [f for f in after - before if f.startswith("lpf_bench_setup")]
    return p, left


@pytest.mark.parametrize("encoder", ["replicated", "sharded", "gather_once"])
def test_bench_dress_rehearsal_eight_ranks_on_one_gpu(encoder):
    """``bench.py --gpus 8`` end to end on the box's ONE GPU (eight ranks under gloo, all on cuda:0): the self-launch, the
    set-up shared through /dev/shm (and removed again), the encoder layout, the rank-agreed launch path, the max-over-ranks
    timing and the one JSON line of rank 0 -- everything of an N = 8 run but the exchange over xGMI."""
    import json
    p, left = _bench(["--gpus", "8", "--config", "tiny", "--steps", "4", "--warmup", "2", "--repeats", "2", "--encoder",
                      encoder, "--no-cpu-baseline", "--weights", "random", "--no-bf16", "--spinup", "0"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert "pairs sharded x8" in d["config"]["parallelism"] and encoder in d["config"]["parallelism"]
    assert d["config"]["encoder_plan"]["chosen"] == encoder
    assert not left, f"set-up files left in /dev/shm: {left}"


def test_bench_capture_failure_on_one_rank_falls_back_everywhere():
    """A recording / capture that fails on ONE rank (injected) must put ALL ranks on the eager path -- the probes run
    barriers and reductions, a rank on another path would wait for ever -- and the run still ends with one JSON line."""
    import json
    p, left = _bench(["--gpus", "4", "--config", "tiny", "--steps", "4", "--warmup", "2", "--repeats", "2",
                      "--no-cpu-baseline", "--weights", "random", "--no-bf16", "--spinup", "0"],
                     {"LPF_BENCH_FAIL_CAPTURE_RANK": "2"})
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["launch"].startswith("eager"), d["config"]["launch"]
    assert not left
