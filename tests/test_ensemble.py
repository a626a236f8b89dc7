"""Multi-rank replica bookkeeping on CPU: sharding and temperature replica exchange over gloo (world_size 2).

The reference has no multi-process code (its REMD driver, scripts/mm_remd.py:85-149, hands a linear temperature
ladder to openmmtools' ReplicaExchangeSampler in one process), so there is no reference fixture for this: the swap
rule is checked against the closed-form Metropolis criterion min(1, exp((beta_i - beta_j)(U_i - U_j))) and the
multi-rank run against the single-rank run of the same seed ("parity unpinned").
"""
import os
import socket
import sys

import numpy as np
import pytest

from openawsem_amd import ensemble

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_partitions_every_replica_exactly_once():
    for n, world in [(1024, 8), (1024, 3), (12, 5), (7, 8), (1, 1)]:
        seen = []
        for rank in range(world):
            r = ensemble.shard(n, world, rank)
            assert len(r) in (n // world, n // world + 1)
            seen.extend(r)
        assert seen == list(range(n))


def test_counter_uniform_is_philox_4x32_10():
    """Known-answer vectors of Random123's Philox4x32-10 (kat_vectors: all-zero and all-ones counter / key): the device kernel
    (csrc/awsem_kernels.h) and the oracle share this generator, so the host-side sweep draws the numbers the device sweep draws."""
    assert ensemble._philox4x32_10((0, 0, 0, 0), 0, 0) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    f = 0xffffffff
    assert ensemble._philox4x32_10((f, f, f, f), f, f) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    c0, c1, _, _ = ensemble._philox4x32_10((4, 3, 0, 0x52454D44), 7, 0)
    assert ensemble.counter_uniform(7, 3, 4) == (((c0 << 32) | c1) >> 11) / 2.0 ** 53


def test_counter_uniform_is_deterministic_and_uniform():
    u = np.array([ensemble.counter_uniform(7, it, i) for it in range(200) for i in range(10)])
    assert np.all((u >= 0) & (u < 1))
    assert abs(u.mean() - 0.5) < 0.03 and abs(np.var(u) - 1 / 12) < 0.01
    assert ensemble.counter_uniform(7, 3, 4) == ensemble.counter_uniform(7, 3, 4)
    assert ensemble.counter_uniform(7, 3, 4) != ensemble.counter_uniform(8, 3, 4)


def test_exchange_follows_metropolis_criterion():
    T = np.linspace(300.0, 600.0, 2)
    beta = 1.0 / (ensemble.KB * T)
    # favourable swap (hot replica has the lower energy): always accepted
    remd = ensemble.ReplicaExchange(T, seed=1)
    scale = remd.exchange(np.array([-100.0, -300.0]))
    assert list(remd.slot) == [1, 0]
    assert np.allclose(scale, np.sqrt([T[1] / T[0], T[0] / T[1]]))
    # unfavourable swap: acceptance frequency -> exp(delta)
    dU = 4.0
    p_expected = np.exp((beta[0] - beta[1]) * (-dU))
    acc, n = 0, 4000
    for seed in range(n):
        remd = ensemble.ReplicaExchange(T, seed=seed)
        remd.exchange(np.array([-300.0, -300.0 + dU]))
        acc += int(remd.accepted[0])
    assert abs(acc / n - p_expected) < 4 * np.sqrt(p_expected * (1 - p_expected) / n)


def test_exchange_alternates_even_and_odd_pairs_and_keeps_a_permutation():
    T = np.linspace(200.0, 600.0, 12)      # the reference's default ladder (mm_remd.py:166-171)
    remd = ensemble.ReplicaExchange(T, seed=5)
    rng = np.random.default_rng(0)
    for it in range(50):
        U = rng.normal(-500, 50, size=12)
        before = remd.attempted.copy()
        remd.exchange(U)
        tried = np.flatnonzero(remd.attempted - before)
        assert np.all(tried % 2 == it % 2)
        assert sorted(remd.slot) == list(range(12))
    assert remd.accepted.sum() > 0


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
from openawsem_amd import ensemble
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
T = np.linspace(200.0, 600.0, 13)          # odd count: unequal shards (7 + 6)
remd = ensemble.ReplicaExchange(T, rank=rank, world=world, seed=11)
local = remd.local
slots, scales = [], []
for it in range(40):
    # synthetic 'potential energy' of replica r at its current temperature (deterministic, rank-independent)
    U_all = -400.0 + 0.5 * T[remd.slot] + 40.0 * np.sin(0.7 * np.arange(13) + it)
    s = remd.exchange(U_all[local.start:local.stop])
    slots.append(remd.slot.copy()); scales.append(s)
g = remd._gather                            # one all_gather_into_tensor per exchange, sizes worked out without a collective
np.savez(os.path.join({out!r}, f"rank{{rank}}.npz"), slots=np.array(slots), scales=np.array(scales),
         accepted=remd.accepted, start=local.start, stop=local.stop, pad=g.m, index=g.index, gathered=g.all.numpy())
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_over_gloo_take_identical_decisions(tmp_path):
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=120) == 0
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert (int(r0["start"]), int(r0["stop"]), int(r1["start"]), int(r1["stop"])) == (0, 7, 7, 13)
    assert np.array_equal(r0["slots"], r1["slots"])             # replicated bookkeeping stays in lock step
    # the gather buffer: shards padded to the longer one (7) plus one word holding the length sent, replica r at
    # rank * 8 + (r - start(rank)); both ranks hold the same
    assert int(r0["pad"]) == 7 and list(r0["index"]) == list(range(7)) + list(range(8, 14))
    assert np.array_equal(r0["gathered"], r1["gathered"]) and r0["gathered"].shape == (16,)
    assert r0["gathered"][7] == 7 and r0["gathered"][15] == 6
    assert np.array_equal(r0["accepted"], r1["accepted"]) and r0["accepted"].sum() > 0

    # single-rank run of the same ladder and seed: same decisions, and the per-rank scale vectors tile it
    T = np.linspace(200.0, 600.0, 13)
    remd = ensemble.ReplicaExchange(T, seed=11)
    for it in range(40):
        U_all = -400.0 + 0.5 * T[remd.slot] + 40.0 * np.sin(0.7 * np.arange(13) + it)
        s = remd.exchange(U_all)
        assert np.array_equal(remd.slot, r0["slots"][it])
        assert np.allclose(s[:7], r0["scales"][it]) and np.allclose(s[7:], r1["scales"][it])


_RAGGED_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
from openawsem_amd import ensemble
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
U = np.arange(13, dtype=np.float64) * 1.5
# (1) no _Gather: any split goes (here 3 + 10, not shard()'s 7 + 6)
cut = 3
mine = U[:cut] if rank == 0 else U[cut:]
got = ensemble.all_gather_energies(mine, world)
assert np.array_equal(got, U), got
# (2) with a _Gather the split must be shard()'s; a wrong length on ONE rank raises on BOTH (after the collective: nobody hangs)
g = ensemble._Gather(13, world, rank)
ok = ensemble.all_gather_energies(U[g.shards[rank].start:g.shards[rank].stop], world, gather=g)
assert np.array_equal(ok, U)
wrong = U[:7] if rank == 0 else U[7:12]          # rank 1 brings 5 values where its shard has 6
try:
    ensemble.all_gather_energies(wrong, world, gather=g)
    raised = ""
except ValueError as e:
    raised = str(e)
assert "[7, 5]" in raised and "[7, 6]" in raised, raised
# ... and the gather is still usable afterwards
assert np.array_equal(ensemble.all_gather_energies(U[g.shards[rank].start:g.shards[rank].stop], world, gather=g), U)
open(os.path.join({out!r}, f"ragged{{rank}}.ok"), "w").write(raised)
dist.barrier()
dist.destroy_process_group()
"""


def test_a_split_that_is_not_the_block_partition_fails_on_every_rank_instead_of_hanging(tmp_path):
    import subprocess
    script = tmp_path / "ragged.py"
    script.write_text(_RAGGED_WORKER.format(root=ROOT, out=str(tmp_path)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=120) == 0
    assert (tmp_path / "ragged0.ok").read_text() == (tmp_path / "ragged1.ok").read_text() != ""


def test_bench_launches_its_own_ranks_when_asked_for_several_gpus():
    """`python bench.py --gpus N` as typed (no torch.distributed.run around it) starts N ranks itself.  Without a GPU every rank
    stops at once with the product path's own message -- which shows that N children were started with a rendezvous environment,
    that the launcher returns their failure, and that nothing waits for ever."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the -m gpu rehearsal (tests/test_gpu_drivers.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 1
    assert "[bench] started 3 ranks" in res.stderr
    # (the launcher ends the other ranks as soon as one has failed -- they would wait in a collective for ever -- so a slow one may be
    # gone before it has printed its own message)
    assert 1 <= res.stderr.count("bench.py needs a GPU") <= 3


def test_swap_all_mixing_samples_the_boltzmann_distribution_over_assignments():
    """Closed-form detailed balance of openmmtools' default scheme (what the reference's mm_remd.py runs): with the energies held
    fixed, repeated swap-all mixing is a Markov chain over the n! assignments of replicas to rungs whose stationary distribution is
    P(s) ~ exp(-sum_r beta_{s(r)} U_r).  Three replicas, six assignments, 3000 exchanges of 27 attempts each."""
    import itertools
    T = np.array([300.0, 340.0, 390.0])
    U = np.array([-12.0, -9.5, -7.0])          # kJ/mol: differences of a few kT
    beta = 1.0 / (ensemble.KB * T)
    perms = list(itertools.permutations(range(3)))
    w = np.array([np.exp(-sum(beta[s[r]] * U[r] for r in range(3))) for s in perms])
    p_exact = w / w.sum()
    assert p_exact.min() > 0.03                # every assignment is visited often enough to test
    remd = ensemble.ReplicaExchange(T, seed=77, mixing="swap-all")
    counts = np.zeros(6)
    n_iter = 3000
    for _ in range(n_iter):
        remd.exchange(U)
        counts[perms.index(tuple(int(v) for v in remd.slot))] += 1
    p = counts / n_iter
    sigma = np.sqrt(p_exact * (1 - p_exact) / n_iter)
    assert np.all(np.abs(p - p_exact) < 4.5 * sigma), (p, p_exact)
    # neighbouring-rung counters only: every counted attempt had |s_i - s_j| = 1
    assert remd.attempted.shape == (2,) and (remd.accepted <= remd.attempted).all() and remd.attempted.min() > n_iter


def test_swap_all_draws_are_uniform_pairs_from_the_documented_stream():
    i, j, lnu = ensemble._swap_all_draws(seed=9, iteration=4, n=12)
    assert len(i) == 12 ** 3 and i.min() == 0 and i.max() == 11 and j.min() == 0 and j.max() == 11
    # attempt 7 by the scalar generator: counter (attempt, iteration lo, iteration hi ^ attempt hi, 'RMSA'), key seed
    c = ensemble._philox4x32_10((7, 4, 0, 0x524D5341), 9, 0)
    assert (c[0] * 12) >> 32 == i[7] and (c[1] * 12) >> 32 == j[7]
    u = (((c[2] << 32) | c[3]) >> 11) / float(1 << 53)
    assert abs(np.log(u) - lnu[7]) < 1e-15
    assert abs(np.exp(lnu).mean() - 0.5) < 0.03 and abs(np.bincount(i, minlength=12).std() / 144) < 0.2
