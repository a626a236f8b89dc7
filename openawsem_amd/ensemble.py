"""Replica ensembles across GPUs: sharding of independent trajectories and temperature replica exchange.

The reference has no multi-process code; its REMD driver hands everything to openmmtools' ReplicaExchangeSampler
in one process (openawsem/scripts/mm_remd.py:85-149: `replicas` temperatures on a linear ladder, `exchange_frequency`
Langevin steps between swap attempts).  Here replicas are block-partitioned over ranks (one process per GPU).  MD
needs no communication; an exchange step is ONE all-gather of the per-replica potential energies (N x 8 bytes,
latency-bound -- RCCL over xGMI on GPUs, gloo on CPU), after which every rank takes the same Metropolis decisions
from a shared counter-based random stream and swaps *temperature labels* (no coordinates move).
"""
from __future__ import annotations

import hashlib

import numpy as np

KB = 0.00831446261815324


def shard(n_total: int, world: int, rank: int):
    """Contiguous block of replica ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def counter_uniform(seed: int, *counters) -> float:
    """Deterministic U(0,1) from (seed, counters): identical on every rank without communication."""
    h = hashlib.blake2b(np.array([seed, *counters], dtype=np.int64).tobytes(), digest_size=8).digest()
    return (int.from_bytes(h, "little") >> 11) * (1.0 / (1 << 53))


def all_gather_energies(local: np.ndarray, world: int, device=None) -> np.ndarray:
    """Concatenate per-rank energy vectors (equal or unequal length) in rank order."""
    if world == 1:
        return np.asarray(local, dtype=np.float64)
    import torch
    import torch.distributed as dist
    n = torch.tensor([len(local)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    m = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(m, dtype=torch.float64, device=device)
    buf[:len(local)] = torch.as_tensor(np.asarray(local, dtype=np.float64), device=device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([o[:int(s.item())].cpu().numpy() for o, s in zip(out, sizes)])


class ReplicaExchange:
    """Temperature REMD bookkeeping.  `slot[r]` is the ladder index replica r currently runs at."""

    def __init__(self, temperatures, rank=0, world=1, seed=2026, device=None):
        self.T = np.asarray(temperatures, dtype=np.float64)
        self.n = len(self.T)
        self.rank, self.world, self.seed, self.device = rank, world, int(seed), device
        self.local = shard(self.n, world, rank)
        self.slot = np.arange(self.n)          # replica -> ladder index (replicated on every rank)
        self.iteration = 0
        self.attempted = np.zeros(self.n - 1, dtype=np.int64)
        self.accepted = np.zeros(self.n - 1, dtype=np.int64)

    def local_temperatures(self):
        return self.T[self.slot[self.local.start:self.local.stop]]

    def exchange(self, local_energies):
        """One sweep of neighbour swaps (even pairs on even iterations, odd pairs on odd ones).  Returns the
        velocity scale factors sqrt(T_new/T_old) of the local replicas."""
        U = all_gather_energies(np.asarray(local_energies, dtype=np.float64), self.world, self.device)
        assert len(U) == self.n
        self.last_energies = U             # every replica's potential energy at this exchange (replicated on every rank)
        beta = 1.0 / (KB * self.T)
        holder = np.argsort(self.slot)         # ladder index -> replica
        old_T = self.T[self.slot].copy()
        for i in range(self.iteration % 2, self.n - 1, 2):
            a, b = holder[i], holder[i + 1]    # a runs at T_i, b at T_{i+1}
            delta = (beta[i] - beta[i + 1]) * (U[a] - U[b])
            self.attempted[i] += 1
            if delta >= 0 or counter_uniform(self.seed, self.iteration, i) < np.exp(delta):
                self.slot[a], self.slot[b] = i + 1, i
                self.accepted[i] += 1
        self.iteration += 1
        new_T = self.T[self.slot]
        return np.sqrt(new_T / old_T)[self.local.start:self.local.stop]

    def acceptance(self):
        return self.accepted / np.maximum(self.attempted, 1)


def run_remd(context, remd: ReplicaExchange, n_iterations: int, steps_per_iteration: int, callback=None):
    """Drive an engine Context holding this rank's shard of the ladder: MD at the current temperatures, gather
    energies, swap labels, rescale velocities.  `context` is an openawsem_amd.engine.Context with
    n_replicas == len(remd.local)."""
    assert context.n_replicas == len(remd.local)
    for it in range(n_iterations):
        context.setReplicaTemperatures(remd.local_temperatures())
        context._step(steps_per_iteration)
        e = context.energies()
        scale = remd.exchange(e)
        context._ctx.scale_velocities(scale)
        if callback is not None:
            callback(it, e, remd)
    context.setReplicaTemperatures(remd.local_temperatures())
    return remd
