"""`python bench.py --gpus N` as the driver types it, rehearsed on one GPU: the launcher starts the ranks itself, and with
--share-gpu the ranks share the device (gloo collectives, because RCCL refuses two ranks on one GPU), so the multi-rank code
path -- sharded replicas, the barrier / MAX reduction around the timed region, and for 2xov-remd the device-side exchange with
the energies handed over in torch-allocated memory and gathered across ranks -- runs end to end.  Not a performance number."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout          # rank 0 alone prints
    return json.loads(lines[0])


def test_one_rank_through_the_launcher_is_the_plain_run():
    d = _bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--replicas", "256", "--no-cpu-baseline", "--no-schedule-legs")
    assert d["n_gpus"] == 1 and d["value"] > 1e6 and "REHEARSAL" not in d["config"]["parallelism"]


def test_two_ranks_started_by_bench_itself_share_the_gpu():
    d = _bench("--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1", "--replicas", "128", "--no-cpu-baseline", "--no-schedule-legs")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "REHEARSAL" in d["config"]["parallelism"]
    assert d["value"] > 1e5 and d["config"]["replicas_per_gpu"] == 128


def test_two_rank_replica_exchange_goes_through_the_gather():
    """2xov ladder of 2 x 6 replicas: every iteration gathers the energies of both ranks and both sweep kernels take the same
    decisions; the line reports the acceptance of the first rungs."""
    d = _bench("--gpus", "2", "--share-gpu", "--workload", "2xov-remd", "--replicas", "6", "--md-steps", "20", "--steps", "6", "--warmup", "1",
               "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and len(d["config"]["remd_acceptance"]) == 8
    assert "all_gather_into_tensor" in d["config"]["remd_exchange"]
