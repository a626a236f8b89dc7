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


_M32 = 0xFFFFFFFF


def _philox4x32_10(c, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011): the counter-based generator of the engine's kernels (csrc/awsem_kernels.h)."""
    c0, c1, c2, c3 = c
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _M32, p1 & _M32, ((p0 >> 32) ^ c3 ^ k1) & _M32, p0 & _M32
        k0, k1 = (k0 + 0x9E3779B9) & _M32, (k1 + 0xBB67AE85) & _M32
    return c0, c1, c2, c3


def counter_uniform(seed: int, iteration: int, pair: int) -> float:
    """Deterministic U(0,1) of (seed, iteration, pair), 53 bits: identical on every rank without communication, and the number
    awsem_remd_sweep_kernel draws on the device (counter (pair, iteration, 'REMD'), key seed)."""
    seed, iteration = int(seed) & 0xFFFFFFFFFFFFFFFF, int(iteration) & 0xFFFFFFFFFFFFFFFF
    c0, c1, _, _ = _philox4x32_10((int(pair) & _M32, iteration & _M32, iteration >> 32, 0x52454D44), seed & _M32, seed >> 32)
    return (((c0 << 32) | c1) >> 11) * (1.0 / (1 << 53))


def _philox4x32_10_np(c0, c1, c2, c3, k0, k1):
    """The same generator over numpy uint64 arrays (values kept below 2^32)."""
    M = np.uint64(_M32)
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M, p0 & M
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return c0, c1, c2, c3


def _swap_all_draws(seed: int, iteration: int, n: int):
    """The n^3 attempts of one swap-all iteration: replica indices (i, j) and ln u, the stream of awsem_remd_mix_all_kernel
    (counter (attempt, iteration, 'RMSA'), key seed; word 0 -> i, word 1 -> j by multiply-shift, words 2,3 -> 53-bit u)."""
    seed, iteration = int(seed) & 0xFFFFFFFFFFFFFFFF, int(iteration) & 0xFFFFFFFFFFFFFFFF
    a = np.arange(n ** 3, dtype=np.uint64)
    c0 = a & np.uint64(_M32)
    c1 = np.full_like(a, iteration & _M32)
    c2 = np.uint64(iteration >> 32) ^ (a >> np.uint64(32))
    c3 = np.full_like(a, 0x524D5341)
    w0, w1, w2, w3 = _philox4x32_10_np(c0, c1, c2, c3, seed & _M32, seed >> 32)
    i = ((w0 * np.uint64(n)) >> np.uint64(32)).astype(np.int64)
    j = ((w1 * np.uint64(n)) >> np.uint64(32)).astype(np.int64)
    u = (((w2 << np.uint64(32)) | w3) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    with np.errstate(divide="ignore"):
        lnu = np.where(u > 0.0, np.log(np.maximum(u, 1e-320)), -1e300)
    return i, j, lnu


class _Gather:
    """One all-gather of per-replica doubles across ranks: shard sizes are static, so they are worked out once (no collective),
    the buffers are allocated once, and an exchange is a single all_gather_into_tensor (RCCL on GPUs, gloo on CPU).  Every rank's
    block is padded to the longest shard plus ONE more word that carries the number of values the rank actually sent: after the
    gather every rank sees every length, so a caller whose split is not shard()'s fails on ALL ranks with the same error instead
    of leaving the others blocked in the collective."""

    def __init__(self, n_total, world, rank, device=None):
        import torch
        self.world, self.rank = world, rank
        self.shards = [shard(n_total, world, r) for r in range(world)]
        self.m = max(len(s) for s in self.shards)                    # shards are padded to the longest one
        self.stride = self.m + 1                                     # + the length word
        self.local = torch.zeros(self.stride, dtype=torch.float64, device=device)
        self.all = torch.zeros(world * self.stride, dtype=torch.float64, device=device)
        self.local[self.m] = len(self.shards[rank])                  # the device path (engine writes its energies here) always sends its shard
        # replica r sits at rank * stride + (r - start(rank)) of the gathered buffer
        self.index = np.concatenate([r * self.stride + np.arange(len(s)) for r, s in enumerate(self.shards)]).astype(np.int32)
        self._host = None

    def gather(self):
        import torch
        import torch.distributed as dist
        if self.all.is_cuda and dist.get_backend() == "gloo":
            # rehearsals of the multi-rank path on one GPU (bench.py --share-gpu): gloo moves host memory only
            if self._host is None:
                self._host = torch.zeros(self.world * self.stride, dtype=torch.float64)
            dist.all_gather_into_tensor(self._host, self.local.cpu())
            self.all.copy_(self._host)
        else:
            dist.all_gather_into_tensor(self.all, self.local)
        return self.all

    def lengths(self, gathered):
        return [int(gathered[r * self.stride + self.m]) for r in range(self.world)]


def _all_gather_ragged(local, world, device=None):
    """Per-rank vectors of ANY length, concatenated in rank order: the lengths first, then one padded gather."""
    import torch
    import torch.distributed as dist
    lens = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(lens, torch.tensor([len(local)], dtype=torch.int64, device=device))
    lens = [int(v) for v in lens.cpu()]
    m = max(max(lens), 1)
    mine = torch.zeros(m, dtype=torch.float64, device=device)
    mine[:len(local)] = torch.as_tensor(local, device=device)
    out = torch.zeros(world * m, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, mine)
    out = out.cpu().numpy()
    return np.concatenate([out[r * m:r * m + n] for r, n in enumerate(lens)])


def all_gather_energies(local: np.ndarray, world: int, device=None, gather=None) -> np.ndarray:
    """Per-rank energy vectors concatenated in rank order (host in, host out: the path of drivers that report every energy).
    Without `gather` the vectors may have any length per rank (two small collectives: lengths, then values).  With a `_Gather`
    (buffers allocated once, one collective) the split must be shard()'s; a rank that brings another length makes EVERY rank
    raise ValueError after the collective -- nobody is left waiting in it."""
    local = np.asarray(local, dtype=np.float64).reshape(-1)
    if world == 1:
        return local
    import torch
    if gather is None:
        return _all_gather_ragged(local, world, device)
    n = min(len(local), gather.m)
    gather.local[:n] = torch.as_tensor(local[:n])
    gather.local[gather.m] = len(local)
    out = gather.gather().cpu().numpy()
    gather.local[gather.m] = len(gather.shards[gather.rank])
    got, want = gather.lengths(out), [len(s) for s in gather.shards]
    if got != want:
        raise ValueError(f"all_gather_energies: per-rank lengths {got} are not the block partition {want} this gather was built for "
                         "(pass gather=None for vectors of arbitrary length)")
    return out[gather.index]


class ReplicaExchange:
    """Temperature REMD bookkeeping.  `slot[r]` is the ladder index replica r currently runs at."""

    def __init__(self, temperatures, rank=0, world=1, seed=2026, device=None, mixing="neighbours"):
        """mixing: "neighbours" -- one sweep of neighbouring-rung swaps of alternating parity per exchange (cheap, the bench default) --
        or "swap-all", openmmtools' default scheme and therefore what the reference's mm_remd.py runs (scripts/mm_remd.py:133-140
        passes no replica_mixing_scheme): n^3 attempts per exchange on uniformly drawn replica pairs."""
        if mixing in ("neighbors", "swap-neighbors"):
            mixing = "neighbours"
        if mixing not in ("neighbours", "swap-all"):
            raise ValueError(f"unknown replica mixing scheme {mixing!r}")
        self.mixing = mixing
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
        if self.world > 1 and getattr(self, "_gather", None) is None:
            self._gather = _Gather(self.n, self.world, self.rank, self.device)      # static shard sizes: no collective, buffers allocated once
        U = all_gather_energies(np.asarray(local_energies, dtype=np.float64), self.world, self.device, getattr(self, "_gather", None))
        assert len(U) == self.n
        self.last_energies = U             # every replica's potential energy at this exchange (replicated on every rank)
        beta = 1.0 / (KB * self.T)
        holder = np.argsort(self.slot)         # ladder index -> replica
        old_T = self.T[self.slot].copy()
        if self.mixing == "swap-all":
            self._mix_all(U, beta)
            self.iteration += 1
            return np.sqrt(self.T[self.slot] / old_T)[self.local.start:self.local.stop]
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

    def _mix_all(self, U, beta):
        """openmmtools' _mix_all_replicas restated for a temperature ladder: n^3 times, draw replicas i, j uniformly; with
        s = slot[.] the current rungs, log_p = -(u_ij + u_ji) + u_ii + u_jj with u_ab = beta_{s_b} U_a, i.e.
        (beta_si - beta_sj)(U_i - U_j); accept when log_p >= 0 or u < exp(log_p) and swap the two rungs.  Counters as on the
        device: attempts between neighbouring rungs only."""
        pi, pj, lnu = _swap_all_draws(self.seed, self.iteration, self.n)
        s = [int(v) for v in self.slot]
        b, Ul = [float(v) for v in beta], [float(v) for v in U]
        for k in range(len(pi)):
            i, j = int(pi[k]), int(pj[k])
            if i == j:
                continue
            si, sj = s[i], s[j]
            log_p = (b[si] - b[sj]) * (Ul[i] - Ul[j])
            near = abs(si - sj) == 1
            if near:
                self.attempted[min(si, sj)] += 1
            if log_p >= 0.0 or lnu[k] < log_p:
                s[i], s[j] = sj, si
                if near:
                    self.accepted[min(si, sj)] += 1
        self.slot = np.array(s)

    def acceptance(self):
        return self.accepted / np.maximum(self.attempted, 1)

    # ---- the same exchange with the bookkeeping on the device (include/awsem_hip.h: awsem_remd_*) -------------------------------
    def attach(self, ctx):
        """Hand the ladder to an engine handle (`openawsem_amd._capi.Context` holding this rank's block of replicas): from here on
        the ladder assignment, temperatures and velocity factors live on the device.  With several ranks the gather buffers are
        torch tensors allocated once; the engine writes its energies straight into them."""
        assert ctx.n_replicas == len(self.local)
        self._ctx = ctx
        self._gather = _Gather(self.n, self.world, self.rank, self.device) if self.world > 1 else None
        ctx.remd_setup(self.T, local_start=self.local.start, gather_index=None if self._gather is None else self._gather.index, seed=self.seed)
        ctx.remd_set_mixing(self.mixing)

    def exchange_on_device(self, group_mask=0xFFFFFFFF):
        """One sweep without a host hop: energies -> device buffer -> (all-gather) -> sweep kernel -> temperatures and velocity
        factors applied.  Single rank: three asynchronous launches on the engine's stream.  Several ranks: the engine's stream is
        joined before the gather and torch's stream after it (the two sit in different HIP runtimes), nothing is copied."""
        ctx = self._ctx
        if self._gather is None:
            ctx.remd_energies(group_mask)
            ctx.remd_sweep(self.iteration)
        else:
            import torch
            g = self._gather
            ctx.remd_energies(group_mask, g.local.data_ptr())
            ctx.synchronize()
            g.gather()
            if g.all.is_cuda:
                torch.cuda.current_stream().synchronize()
            ctx.remd_sweep(self.iteration, g.all.data_ptr())
        self.iteration += 1

    def pull(self):
        """Bring the bookkeeping back to the host (reporting / checkpoints): slot, attempted, accepted."""
        slot, att, acc, _ = self._ctx.remd_state()
        self.slot, self.attempted, self.accepted = slot.astype(np.int64), att, acc
        return self


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
