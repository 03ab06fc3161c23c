"""Corpus row-sharded search over the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm).

The reference searches on a single GPU (script/embed_for_neg_mining.sh:16); sharding is the build's
own multi-GPU form: every rank holds rows [lo, hi) of the corpus, sees every query,
produces a local top-k with GLOBAL indices as packed 8-byte keys, and the per-rank lists are exchanged with ONE
all-gather per chunk and merged on the device (abc_topk_packed / abc_topk_merge_packed).  The 0.95-positive rule needs s_pos[i] = q_i . c_i, which
only the shard owning row i can compute: owners fill their rows, one all-reduce(SUM) replicates it.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


MAX_ROWS = 0xFFFFFFFF      # exclusive: index 0xFFFFFFFF would collide with the reserved empty key


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of rank `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class Backend:
    """The three device operations the sharded driver needs.  The default binds the HIP kernels;
    CPU-only tests of the exchange logic bind a checker instead (tests/ only)."""
    topk: Callable          # (q, c_shard, k, upper_bound, idx_offset) -> (vals [nq,k], idx [nq,k] global, -1 = hole)
    rowdot: Callable        # (q_rows, c_rows) -> [n] f32
    merge: Callable         # (vals [P,nq,k], idx [P,nq,k], idx_limit) -> (vals [nq,k], idx [nq,k])
    # the packed pair: one 8-byte key per entry (score and global index), so that a chunk's exchange is ONE all-gather
    topk_packed: Optional[Callable] = None     # (q, c_shard, k, upper_bound, idx_offset) -> keys i64 [nq,k]
    merge_packed: Optional[Callable] = None    # (keys [P,nq,k]) -> (vals [nq,k], idx [nq,k])


def hip_backend() -> Backend:
    from . import search, kernels

    def _topk(q, c, k, ub, off):
        v, i, _ = search.topk(q, c, k, upper_bound=ub, idx_offset=off)
        return v, i

    def _rowdot(q, c):
        return kernels.rowdot(q.data, c.data, q_scale=q.scale, c_scale=c.scale)

    def _topk_packed(q, c, k, ub, off):
        if q.compute != c.compute:
            raise ValueError("query and corpus must use the same compute dtype")
        return kernels.topk_packed(q.data, c.data, k, upper_bound=ub, idx_offset=off, q_scale=q.scale, c_scale=c.scale)

    return Backend(topk=_topk, rowdot=_rowdot, merge=kernels.merge_topk, topk_packed=_topk_packed,
                   merge_packed=kernels.merge_packed)


class ShardedCorpus:
    """This rank's rows [lo, hi) of an n_total-row corpus."""

    def __init__(self, shard, lo: int, n_total: int, group=None, backend: Optional[Backend] = None,
                 exchange: Optional[bool] = None):
        if n_total >= MAX_ROWS:
            raise ValueError(f"n_total={n_total}: global row indices travel in the 32-bit index field of the packed "
                             f"(score, index) keys; a sharded corpus holds at most {MAX_ROWS - 1} rows")
        self.shard, self.lo, self.n_total = shard, lo, n_total
        self.hi = lo + shard.shape[0]
        self.group = group
        self.backend = backend or hip_backend()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # the exchange (packed lists -> all-gather -> merge; all-reduce of s_pos) runs whenever there are several ranks;
        # exchange=True forces it under a ONE-rank group too: the only way to execute the RCCL calls on a one-GPU box
        # (tests/test_gpu_zrccl_one_rank.py)
        self.exchange = (self.world > 1) if exchange is None else bool(exchange)
        if self.exchange and not dist.is_initialized():
            raise ValueError("exchange=True needs an initialised process group")

    # -- exchange steps -------------------------------------------------------------------------
    def _host_staged(self, t: torch.Tensor) -> bool:
        """Under a gloo group (rehearsals on one GPU, CPU-side tests) device tensors cross in host memory.  A single gloo
        collective on device tensors completes; what deadlocks is several ASYNCHRONOUS ones outstanding while the main
        thread keeps enqueueing kernels (gloo stages device tensors on its own worker threads and synchronises streams
        there): scripts/gloo_device_diag.py, DESIGN.md section 5.
        RCCL (backend "nccl", the production path) takes the device tensors as they are."""
        return t.is_cuda and self.exchange and dist.get_backend(self.group) == "gloo"

    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        if not self.exchange:
            return t.unsqueeze(0)
        # concatenated along dim 0 (the form both RCCL and gloo accept), viewed as [world, ...]
        dev = t.device
        if self._host_staged(t):
            t = t.cpu()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out.to(dev).view((self.world,) + tuple(t.shape))

    def positive_scores(self, q, offset: int) -> torch.Tensor:
        """s_pos[i] = q[i] . corpus[offset + i], replicated on every rank."""
        n = q.shape[0]
        dev = q.data.device if hasattr(q, "data") else q.device
        s = torch.zeros((n,), dtype=torch.float32, device=dev)
        a, b = max(self.lo, offset), min(self.hi, offset + n)     # global rows owned here
        if a < b:
            s[a - offset:b - offset] = self.backend.rowdot(q.rows(a - offset, b - offset),
                                                           self.shard.rows(a - self.lo, b - self.lo))
        if self.exchange:
            if self._host_staged(s):                        # gloo: through host memory (see _host_staged)
                h = s.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                s = h.to(dev)
            else:
                dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group)
        return s

    def topk(self, q, k: int, upper_bound: Optional[torch.Tensor] = None, chunk: int = 8192
             ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Exact global top-k of every query, replicated on every rank.  Queries go through the local sweep in chunks;
        a chunk's lists leave the sweep as packed 8-byte keys (score and global index), cross the GPUs in ONE all-gather of
        k * 8 bytes per query and are merged on the device before the next chunk's sweep is enqueued.  The order on the
        stream is sweep, gather, merge, sweep, ...: the sweep owns every CU (one workgroup per CU with all of its registers
        and LDS), so a collective enqueued beside it does not overlap with it -- it can only race it for CUs at the
        boundary between two sweeps, and a rank whose collective lost that race would hold its peers' collective kernels
        spinning for a whole sweep.  In order, the gather has the chip to itself for the ~0.1 ms it takes.
        (A backend without the packed pair -- CPU-side tests of the exchange logic -- exchanges (value f32, index i64)
        with two all-gathers.)"""
        n = q.shape[0]
        if not self.exchange:
            return self.backend.topk(q, self.shard, k, upper_bound, self.lo)
        packed = self.backend.topk_packed is not None and self.backend.merge_packed is not None
        outs_v, outs_i = [], []
        for a in range(0, max(n, 1), chunk):
            b = min(n, a + chunk)
            qc = q if (a == 0 and b == n) else q.rows(a, b)
            ub = None if upper_bound is None else upper_bound[a:b]
            if packed:
                parts = [self.backend.topk_packed(qc, self.shard, k, ub, self.lo)]
            else:
                parts = list(self.backend.topk(qc, self.shard, k, ub, self.lo))
            dev = parts[0].device
            if self._host_staged(parts[0]):                 # gloo: the lists cross in host memory
                parts = [t.cpu() for t in parts]
            bufs = []
            for t in parts:
                g = torch.empty((self.world * t.shape[0], k), dtype=t.dtype, device=t.device)
                dist.all_gather_into_tensor(g, t.contiguous(), group=self.group)
                bufs.append(g.to(dev).view(self.world, b - a, k))
            if packed:
                mv, mi = self.backend.merge_packed(bufs[0])
            else:
                mv, mi = self.backend.merge(bufs[0], bufs[1], self.n_total)
            outs_v.append(mv)
            outs_i.append(mi)
        if len(outs_v) == 1:
            return outs_v[0], outs_i[0]
        return torch.cat(outs_v, dim=0), torch.cat(outs_i, dim=0)

    def mine_negatives(self, q, k: int, ratio: float = 0.95, offset: int = 0, fill_masked_tail: bool = True,
                       chunk: int = 8192):
        """Sharded form of pretrain/create_negative_index.py:45-52 for queries whose positives are
        corpus rows offset .. offset+n-1.  Returns (idx, vals, s_pos), identical on every rank and identical
        to search.mine_negatives on the unsharded corpus: rows with fewer than k survivors get their tail filled
        with the masked columns by the same helper (a second, unbounded sharded search of those rows; the merged
        lists are replicated, so every rank sees the same rows and takes part in the same collectives)."""
        from .search import fill_masked_tail_rows
        s_pos = self.positive_scores(q, offset)
        ub = s_pos * ratio
        v, i = self.topk(q, k, upper_bound=ub, chunk=chunk)
        if fill_masked_tail and self.exchange:
            fill_masked_tail_rows(i, ub, lambda rows: self.topk(q.take(rows), k))
        elif fill_masked_tail:
            fill_masked_tail_rows(i, ub, lambda rows: self.backend.topk(q.take(rows), self.shard, k, None, self.lo))
        return i, v, s_pos
