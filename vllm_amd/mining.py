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

import json
import os
import sys
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import parallel, search
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


def upload(x: torch.Tensor, compute: str, device=None, rows: Optional[Tuple[int, int]] = None,
           block: int = 1 << 16) -> search.Embeddings:
    """Rows [lo, hi) of a (possibly memory-mapped) CPU table as resident Embeddings, converted block by
    block so that neither the host nor the device ever holds a second full-size copy."""
    lo, hi = rows if rows is not None else (0, x.shape[0])
    first = search.Embeddings(x[lo:min(lo + block, hi)], compute, device)
    if hi - lo <= block:
        return first
    out = object.__new__(search.Embeddings)
    out.compute = compute
    out.data = torch.empty((hi - lo, x.shape[1]), dtype=first.data.dtype, device=first.data.device)
    out.scale = None if first.scale is None else torch.empty((hi - lo,), dtype=torch.float32, device=first.data.device)
    pos = 0
    part = first
    while True:
        n = part.data.shape[0]
        out.data[pos:pos + n] = part.data
        if out.scale is not None:
            out.scale[pos:pos + n] = part.scale
        pos += n
        if lo + pos >= hi:
            break
        part = search.Embeddings(x[lo + pos:min(lo + pos + block, hi)], compute, first.data.device)
    return out


def _row_text(i: int, row: Sequence[int]) -> str:
    return f'"{i}": [' + ", ".join(str(int(j)) for j in row) + "]"


def write_negatives_json(path: str, blocks: Iterator[Tuple[int, np.ndarray]]) -> int:
    """Stream {"0": [..], "1": [..], ...} to `path` (tmp + rename).  `blocks` yields (first row, int array
    [n, k]) in row order; entries < 0 (no candidate) are dropped.  Byte-identical to json.dump of the
    reference's {int: list} dict."""
    tmp = path + ".tmp"
    n_rows = 0
    with open(tmp, "w") as f:
        f.write("{")
        first = True
        for start, arr in blocks:
            if start != n_rows:
                raise AbcError(f"negatives blocks out of order: expected row {n_rows}, got {start}")
            lines = []
            for r in range(arr.shape[0]):
                row = arr[r]
                lines.append(_row_text(start + r, row[row >= 0]))
            if lines:
                f.write(("" if first else ", ") + ", ".join(lines))
                first = False
            n_rows += arr.shape[0]
        f.write("}")
    os.replace(tmp, path)
    return n_rows


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
                          progress=None) -> str:
    """create_negative_index.main() for a directory holding query.pt / cand.pt; returns the path of
    negatives.json.  Under torch.distributed (one process per GPU) the corpus is row-sharded over the
    ranks, every rank takes part in every block and rank 0 writes the files."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    q_cpu, c_cpu = load_embeddings(negative_path)
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
    if todo:
        lo, hi = parallel.shard_bounds(n, world, rank)
        shard = upload(c_cpu, compute, rows=(lo, hi))
        corpus = parallel.ShardedCorpus(shard, lo, n)
        for s, m in todo:
            qb = search.Embeddings(q_cpu[s:s + m], compute, shard.data.device)
            # one code path at every world size (world 1: no collective, the local sweep is the result)
            idx, _, _ = corpus.mine_negatives(qb, top_eff, ratio=ratio, offset=s)
            if rank == 0:
                arr = idx.to(torch.int64).cpu().numpy()
                tmp = os.path.join(parts_dir, _part_name(s) + ".tmp")
                with open(tmp, "wb") as f:
                    np.save(f, arr, allow_pickle=False)
                os.replace(tmp, os.path.join(parts_dir, _part_name(s)))
            if progress is not None:
                progress(s + m, n)
    if world > 1:
        dist.barrier()
    if rank == 0:
        def blocks():
            for s in starts:
                yield s, np.load(os.path.join(parts_dir, _part_name(s)), allow_pickle=False)
        rows = write_negatives_json(out_path, blocks())
        if rows != n:
            raise AbcError(f"negatives.json holds {rows} rows, expected {n}")
        if not keep_parts:
            for name in os.listdir(parts_dir):
                os.remove(os.path.join(parts_dir, name))
            os.rmdir(parts_dir)
    if world > 1:
        dist.barrier()
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
