"""Resumable hard-negative mining driver and the file formats either side of the search path.

Mirrors, with the same file names and JSON layout:

    main()                    pretrain/create_negative_index.py:10-43   (cand.pt / query.pt -> negative/negatives.json)
    negatives_for_sample      pretrain/create_negative_index.py:45-52   (in vllm_amd.search; the batched form is used here)
    query.pt / cand.pt        pretrain/embed_training_data.py:96-107    (torch.save of fp32 [N, D], row i of one pairs with row i of the other)
    negatives.json consumer   dataset_utils/conceptual_captions.py:143-159

The reference loops over the N queries one GEMV at a time and holds the whole result in a Python dict
until the end, so a crash at sample 3 299 999 loses everything.  Here the queries go through the
fused search kernel in blocks, every finished block is written to `negatives.parts/` (atomic rename),
a restarted run skips the blocks it finds there, and `negatives.json` is streamed out at the end in
exactly the text `json.dump(negative_dict, f)` produces: keys are the stringified row indices in
order, values the candidate rows best first.

No CPU path: the search runs in libabcsearch.so on the GPU (vllm_amd.search / vllm_amd.parallel).
"""
from __future__ import annotations

import ctypes
import json
import os
import sys
import time as _time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, parallel, search
from ._lib import AbcError

PARTS_DIR = "negatives.parts"
MANIFEST = "manifest.json"


# ------------------------------------------------------------------------------------------------
# file formats
# ------------------------------------------------------------------------------------------------
def save_embeddings(negative_path: str, q: torch.Tensor, c: torch.Tensor) -> None:
    """query.pt / cand.pt as the embedding pass leaves them (fp32 [N, D], CPU)."""
    os.makedirs(negative_path, exist_ok=True)
    torch.save(q.detach().to(torch.float32).cpu(), os.path.join(negative_path, "query.pt"))
    torch.save(c.detach().to(torch.float32).cpu(), os.path.join(negative_path, "cand.pt"))


def load_embeddings(negative_path: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """(query, cand) CPU tensors, memory-mapped so that a 54 GB fp32 table is paged in block by block."""
    out = []
    for name in ("query.pt", "cand.pt"):
        path = os.path.join(negative_path, name)
        if not os.path.exists(path):
            raise AbcError(f"{path} not found (run the embedding pass first)")
        try:
            t = torch.load(path, weights_only=True, mmap=True)
        except (RuntimeError, ValueError):          # legacy (non-zip) serialisation cannot be mapped
            t = torch.load(path, weights_only=True)
        if t.dim() != 2:
            raise AbcError(f"{name}: expected a [N, D] matrix, got {tuple(t.shape)}")
        out.append(t)
    q, c = out
    if q.shape != c.shape:
        raise AbcError(f"query.pt {tuple(q.shape)} and cand.pt {tuple(c.shape)} must be row-aligned pairs")
    return q, c


class _StagedRows:
    """Row blocks of a (possibly memory-mapped) CPU table on their way to the GPU: the copy out of the mapping (page-ins
    included) into one of two pinned staging buffers runs on worker threads, a slice each, the host-to-device copy and the
    conversion to the compute dtype on a side stream -- while the caller's stream works on the block before.

        st = _StagedRows(x, block, compute, device); st.prefetch(lo, hi); emb = st.take()      # strictly alternating
    """

    def __init__(self, x: torch.Tensor, block: int, compute: str, device: torch.device, workers: int = 8):
        self.x, self.block, self.compute, self.device = x, block, compute, device
        self.pinned = [torch.empty((block, x.shape[1]), dtype=x.dtype).pin_memory() for _ in range(2)]
        self.free_at = [None, None]                   # event: the slot's last host-to-device copy has run
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.workers = workers
        self.side = torch.cuda.Stream(device)
        self.slot = 0
        self.pending = None
        self.waited = 0.0                             # seconds the caller spent waiting for page-ins (diagnostics)

    def prefetch(self, lo: int, hi: int) -> None:
        assert self.pending is None and 0 < hi - lo <= self.block
        slot, m = self.slot, hi - lo
        if self.free_at[slot] is not None:
            self.free_at[slot].synchronize()          # two blocks back: long done
        buf = self.pinned[slot]
        step = -(-m // self.workers)
        futs = [self.pool.submit(buf[a:min(a + step, m)].copy_, self.x[lo + a:lo + min(a + step, m)])
                for a in range(0, m, step)]
        self.pending = (slot, m, futs)
        self.slot ^= 1

    def take(self) -> search.Embeddings:
        """The prefetched block as resident Embeddings, usable on the CURRENT stream."""
        slot, m, futs = self.pending
        self.pending = None
        t0 = _time.perf_counter()
        for f in futs:
            f.result()
        self.waited += _time.perf_counter() - t0
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.side):
            raw = self.pinned[slot][:m].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.free_at[slot] = ev
            emb = search.Embeddings(raw, self.compute, self.device)
            ready = torch.cuda.Event()
            ready.record(self.side)
        cur.wait_event(ready)
        for t in (raw, emb.data, emb.scale):          # allocated on the side stream, used on the caller's
            if t is not None:
                t.record_stream(cur)
        return emb

    def close(self) -> None:
        self.pool.shutdown(wait=True)


class _Download:
    """A block's result on its way back: the [m, top] index table and the "some row has fewer than top survivors" flag go
    to one of two pinned host buffers with asynchronous copies; wait() blocks until they have arrived (by then the
    next block's sweep is running)."""

    def __init__(self, block: int, top: int, device: torch.device):
        self.device = device
        self.waited = 0.0                             # seconds spent waiting for the GPU (diagnostics)
        self.host = [(torch.empty((block, top), dtype=torch.int64).pin_memory(), torch.empty((), dtype=torch.bool).pin_memory())
                     for _ in range(2)]

    def start(self, slot: int, idx: torch.Tensor):
        arr_t, holes = self.host[slot]
        arr_t[:idx.shape[0]].copy_(idx, non_blocking=True)
        holes.copy_((idx[:, -1] < 0).any(), non_blocking=True)       # holes sit at the tail of a row
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return slot, idx.shape[0], ev

    def wait(self, handle) -> Tuple[np.ndarray, bool]:
        slot, m, ev = handle
        t0 = _time.perf_counter()
        ev.synchronize()
        self.waited += _time.perf_counter() - t0
        arr_t, holes = self.host[slot]
        return arr_t[:m].numpy(), bool(holes)

    def refresh(self, handle, idx: torch.Tensor) -> np.ndarray:
        """After a tail fill changed idx on the device: fetch it again (synchronous; rare)."""
        slot, m, _ = handle
        arr_t, _ = self.host[slot]
        arr_t[:m].copy_(idx)
        return arr_t[:m].numpy()


def upload(x: torch.Tensor, compute: str, device=None, rows: Optional[Tuple[int, int]] = None,
           block: int = 1 << 16) -> search.Embeddings:
    """Rows [lo, hi) of a (possibly memory-mapped) CPU table as resident Embeddings, converted block by
    block so that neither the host nor the device ever holds a second full-size copy; block i+1 is paged in and
    copied while block i converts (_StagedRows)."""
    lo, hi = rows if rows is not None else (0, x.shape[0])
    dev = search._device(device)
    if hi - lo <= block or not x.device.type == "cpu":
        return search.Embeddings(x[lo:hi], compute, dev)
    st = _StagedRows(x, block, compute, dev)
    out = object.__new__(search.Embeddings)
    out.compute = compute
    out.data = out.scale = None
    pos = lo
    st.prefetch(pos, min(pos + block, hi))
    try:
        while pos < hi:
            n = min(block, hi - pos)
            part = st.take()
            if pos + n < hi:
                st.prefetch(pos + n, min(pos + n + block, hi))
            if out.data is None:
                out.data = torch.empty((hi - lo, part.data.shape[1]), dtype=part.data.dtype, device=dev)
                out.scale = None if part.scale is None else torch.empty((hi - lo,), dtype=torch.float32, device=dev)
            out.data[pos - lo:pos - lo + n] = part.data
            if out.scale is not None:
                out.scale[pos - lo:pos - lo + n] = part.scale
            pos += n
    finally:
        st.close()
    return out


def format_negatives(first_row: int, arr: np.ndarray) -> bytes:
    """The bytes json.dump writes for rows first_row.. of the {int: [ints]} dict (no braces), entries < 0 dropped:
    abc_format_negatives, host code in the library (a Python loop over 3.3 M x 100 integers takes minutes)."""
    lib = _lib.load()
    a = np.ascontiguousarray(arr, dtype=np.int64)
    n, k = (a.shape[0], a.shape[1]) if a.ndim == 2 else (0, 0)
    if n == 0:
        return b""
    cap = -int(lib.abc_format_negatives(a.ctypes.data, n, k, int(first_row), None, 0))
    buf = ctypes.create_string_buffer(cap)
    got = int(lib.abc_format_negatives(a.ctypes.data, n, k, int(first_row), ctypes.addressof(buf), cap))
    if got <= 0:
        raise AbcError(f"abc_format_negatives failed ({got})")
    return buf.raw[:got]


class _JsonStream:
    """negatives.json written block by block (tmp file, renamed by close()): "{" + blocks joined by ", " + "}"."""

    def __init__(self, path: str):
        self.path, self.tmp = path, path + ".tmp"
        self.f = open(self.tmp, "wb")
        self.f.write(b"{")
        self.rows, self.first = 0, True

    def add(self, start: int, arr: np.ndarray) -> None:
        if start != self.rows:
            raise AbcError(f"negatives blocks out of order: expected row {self.rows}, got {start}")
        text = format_negatives(start, arr)
        if text:
            self.f.write((b"" if self.first else b", ") + text)
            self.first = False
        self.rows += arr.shape[0]

    def close(self) -> int:
        self.f.write(b"}")
        self.f.close()
        os.replace(self.tmp, self.path)
        return self.rows

    def abort(self) -> None:
        self.f.close()
        if os.path.exists(self.tmp):
            os.remove(self.tmp)


def write_negatives_json(path: str, blocks: Iterator[Tuple[int, np.ndarray]]) -> int:
    """Stream {"0": [..], "1": [..], ...} to `path` (tmp + rename).  `blocks` yields (first row, int array
    [n, k]) in row order; entries < 0 (no candidate) are dropped.  Byte-identical to json.dump of the
    reference's {int: list} dict."""
    js = _JsonStream(path)
    try:
        for start, arr in blocks:
            js.add(start, arr)
    except BaseException:
        js.abort()
        raise
    return js.close()


def load_negatives(path: str) -> Dict[str, List[int]]:
    with open(path, "rb") as f:
        return json.loads(f.read())


# consumer contract: dataset_utils/conceptual_captions.py:143-159
def check_negative_index(n_meta: int, negative_meta: Dict[str, List[int]], negatives: Optional[int] = None) -> None:
    """The adapter's load-time assertion (len(meta) == len(negative_meta), :147) plus what
    _attach_negatives needs: entry str(idx) holds at least 5*negatives candidate rows, all inside meta."""
    if len(negative_meta) != n_meta:
        raise AbcError(f"negatives.json has {len(negative_meta)} entries for {n_meta} samples")
    if negatives is not None:
        for key, lst in negative_meta.items():
            if len(lst) < 5 * negatives:
                raise AbcError(f"sample {key}: {len(lst)} negatives, the sampler indexes up to {5 * negatives - 1}")
            if lst and (min(lst) < 0 or max(lst) >= n_meta):
                raise AbcError(f"sample {key}: candidate row outside [0, {n_meta})")


def sample_negative_rows(negative_meta: Dict[str, List[int]], idx: int, negatives: int,
                         rng: Optional[np.random.RandomState] = None) -> List[int]:
    """Candidate rows _attach_negatives picks for sample idx (:150-156): one of every 5 consecutive
    ranks, `negatives` groups; the same np.random calls in the same order as the reference."""
    offset = 5
    starts = np.arange(negatives) * offset
    offsets = (rng.randint(0, offset, size=negatives) if rng is not None
               else np.random.randint(0, offset, size=negatives))
    neg_idx = starts + offsets
    negatives_for_idx = negative_meta[str(idx)]
    return [negatives_for_idx[i] for i in neg_idx]


# ------------------------------------------------------------------------------------------------
# the driver
# ------------------------------------------------------------------------------------------------
def _part_name(start: int) -> str:
    return f"part-{start:012d}.npy"


def _load_manifest(parts_dir: str) -> Optional[dict]:
    p = os.path.join(parts_dir, MANIFEST)
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f)


def _finished_blocks(parts_dir: str, starts: Sequence[int], sizes: Sequence[int], top: int) -> Dict[int, str]:
    done = {}
    for s, n in zip(starts, sizes):
        p = os.path.join(parts_dir, _part_name(s))
        if os.path.exists(p):
            try:
                a = np.load(p, mmap_mode="r", allow_pickle=False)
            except (ValueError, OSError):
                continue                          # torn file from a crash before the rename: redo the block
            if a.shape == (n, top):
                done[s] = p
    return done


def create_negative_index(negative_path: str, top: int = 100, ratio: float = 0.95, compute: str = "f32",
                          block: int = 8192, resume: bool = True, keep_parts: bool = False,
                          progress=None, timings: Optional[dict] = None) -> str:
    """create_negative_index.main() for a directory holding query.pt / cand.pt; returns the path of
    negatives.json.  Under torch.distributed (one process per GPU) the corpus is row-sharded over the
    ranks, every rank takes part in every block and rank 0 writes the files.

    The loop is a software pipeline: while block t sweeps, block t+1 is paged in from query.pt, copied and converted on a
    side stream, and block t-1's lists come back through pinned memory, go to their part file and are appended to
    negatives.json (formatted by the library's host code), so the GPU never waits for the files.  `timings`, when given,
    receives the seconds spent in {"load", "upload", "loop", "finalize"}."""
    _t = [_time.perf_counter()]

    def _lap(name: str) -> None:
        if timings is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            now = _time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - _t[0]
            _t[0] = now
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    q_cpu, c_cpu = load_embeddings(negative_path)
    _lap("load")
    n, d = q_cpu.shape
    top_eff = min(top, n)
    parts_dir = os.path.join(negative_path, PARTS_DIR)
    out_path = os.path.join(negative_path, "negatives.json")
    config = {"n": n, "d": d, "top": top_eff, "ratio": ratio, "compute": compute, "block": block}
    starts = list(range(0, n, block))
    sizes = [min(block, n - s) for s in starts]

    done: Dict[int, str] = {}
    if rank == 0:
        os.makedirs(parts_dir, exist_ok=True)
        if resume and _load_manifest(parts_dir) == config:
            done = _finished_blocks(parts_dir, starts, sizes, top_eff)
        else:
            for name in os.listdir(parts_dir):
                os.remove(os.path.join(parts_dir, name))
            with open(os.path.join(parts_dir, MANIFEST), "w") as f:
                json.dump(config, f)
    if world > 1:
        box = [sorted(done)]
        dist.broadcast_object_list(box, src=0)
        done_starts = set(box[0])
    else:
        done_starts = set(done)

    todo = [(s, m) for s, m in zip(starts, sizes) if s not in done_starts]
    js = _JsonStream(out_path) if rank == 0 else None
    next_json = 0                                       # index into `starts` of the next block negatives.json needs

    def drain_json(upto: int) -> None:
        """Append every finished block before starts[upto] to negatives.json, in row order (blocks found on disk from
        an earlier run are formatted as they are reached)."""
        nonlocal next_json
        while js is not None and next_json < upto:
            s0 = starts[next_json]
            js.add(s0, np.load(os.path.join(parts_dir, _part_name(s0)), allow_pickle=False))
            next_json += 1

    try:
        if todo:
            lo, hi = parallel.shard_bounds(n, world, rank)
            _lap("load")
            shard = upload(c_cpu, compute, rows=(lo, hi))
            _lap("upload")
            dev = shard.data.device
            corpus = parallel.ShardedCorpus(shard, lo, n)
            stage = _StagedRows(q_cpu, block, compute, dev)
            down = _Download(block, top_eff, dev)
            inflight = None                             # (start, rows, download handle, query block, upper bounds, idx on device)

            def finish(job) -> None:
                """Block `job` has left the GPU: fill rows with fewer than `top` survivors (rare; a second search, the
                same collectives on every rank), write the part file, extend negatives.json."""
                s0, m0, handle, qb0, ub0, idx0 = job
                arr, holes = down.wait(handle)
                if holes:
                    if world > 1:
                        search.fill_masked_tail_rows(idx0, ub0, lambda rows: corpus.topk(qb0.take(rows), top_eff))
                    else:
                        search.fill_masked_tail_rows(idx0, ub0, lambda rows: search.topk(qb0.take(rows), shard, top_eff,
                                                                                          idx_offset=lo)[:2])
                    arr = down.refresh(handle, idx0)
                if rank == 0:
                    tmp = os.path.join(parts_dir, _part_name(s0) + ".tmp")
                    with open(tmp, "wb") as f:
                        np.save(f, arr, allow_pickle=False)
                    os.replace(tmp, os.path.join(parts_dir, _part_name(s0)))
                    drain_json(starts.index(s0) + 1)
                if progress is not None:
                    progress(s0 + m0, n)

            stage.prefetch(todo[0][0], todo[0][0] + todo[0][1])
            try:
                for t, (s, m) in enumerate(todo):
                    qb = stage.take()                   # block t on the device (its upload ran behind block t-1's sweep)
                    if t + 1 < len(todo):
                        stage.prefetch(todo[t + 1][0], todo[t + 1][0] + todo[t + 1][1])
                    # one code path at every world size (world 1: no collective, the local sweep is the result)
                    idx, _, s_pos = corpus.mine_negatives(qb, top_eff, ratio=ratio, offset=s, fill_masked_tail=False)
                    handle = down.start(t & 1, idx)
                    if inflight is not None:            # block t-1: written out while block t sweeps
                        finish(inflight)
                    inflight = (s, m, handle, qb, None if s_pos is None else s_pos * ratio, idx)
                if inflight is not None:
                    finish(inflight)
            finally:
                stage.close()
            _lap("loop")
            if timings is not None:
                timings["loop_wait_pagein"] = getattr(stage, "waited", 0.0)
                timings["loop_wait_gpu"] = getattr(down, "waited", 0.0)
        if world > 1:
            dist.barrier()
        if rank == 0:
            drain_json(len(starts))
            rows = js.close()
            js = None
            if rows != n:
                raise AbcError(f"negatives.json holds {rows} rows, expected {n}")
            if not keep_parts:
                for name in os.listdir(parts_dir):
                    os.remove(os.path.join(parts_dir, name))
                os.rmdir(parts_dir)
    finally:
        if js is not None:
            js.abort()
    if world > 1:
        dist.barrier()
    _lap("finalize")
    return out_path


def _negative_dir(arg: str) -> str:
    """The reference takes the training JSON config and derives <adapter.root>/negative from the dataset
    name (create_negative_index.py:12-29; roots come from CC_ROOT / CC_PRETRAIN_ROOT / MSCOCO_ROOT ...).
    Accepted here: that directory itself, a dataset root containing negative/, or the JSON config."""
    if os.path.isdir(arg):
        sub = os.path.join(arg, "negative")
        return sub if os.path.isdir(sub) and not os.path.exists(os.path.join(arg, "query.pt")) else arg
    if arg.endswith(".json") and os.path.exists(arg):
        with open(arg) as f:
            name = json.load(f).get("training_dataset_name")
        env = {"cc": "CC_ROOT", "cc128k": "CC_ROOT", "cc_pretrain": "CC_PRETRAIN_ROOT", "mscoco": "MSCOCO_ROOT",
               "mscoco_neg": "MSCOCO_ROOT", "vg": "VG_ROOT", "wiki_instruct": "WIKI_INSTRUCT_ROOT"}.get(name)
        if env is None or env not in os.environ:
            raise AbcError(f"dataset {name!r}: set {env or 'the dataset root variable'} or pass the negative/ directory")
        return os.path.join(os.environ[env], "negative")
    raise AbcError(f"{arg}: neither a directory nor a JSON config")


def main(argv: Optional[Sequence[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print("usage: python -m vllm_amd.mining <negative dir | dataset root | config.json> [top] [compute]", file=sys.stderr)
        return 2
    path = _negative_dir(argv[0])
    top = int(argv[1]) if len(argv) > 1 else 100
    compute = argv[2] if len(argv) > 2 else "f32"
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")

    def progress(done, total):
        print(f"\rProcessing Samples: {done}/{total}", end="", file=sys.stderr, flush=True)

    out = create_negative_index(path, top=top, compute=compute, progress=progress)
    print(f"\nwrote {out}", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
