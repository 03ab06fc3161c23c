"""Retrieval eval harness: Karpathy-split recall@1/5/10, same inputs, dict semantics and printed lines as

    eval_mscoco   evaluate/eval_mscoco.py:56-99     (env MSCOCO_EVAL -> dataset_coco.json)
    eval_flickr   evaluate/eval_flickr30k.py:56-100 (env FLICKR_EVAL -> dataset_flickr30k.json)
    intersect     evaluate/eval_mscoco.py:15-16

`fxn(x, dtype="image"|"text") -> Tensor[1, D]` is the embedding function the reference gets from
evaluate/embed_function.py; the backbone behind it is out of scope here, so the
caller supplies it.  The search itself is ONE fused GPU top-10 per direction, reused for k = 1, 5, 10
(the reference runs the full matmul + a 500 MB device-to-host copy six times).  Results are keyed by
the raw caption string / image path exactly as in the reference, so duplicated captions collapse to
the last occurrence (eval_mscoco.py:45-48).
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from . import search
from ._lib import AbcError

TOPKS = (1, 5, 10)


def intersect(l1, l2) -> bool:
    return len(set.intersection(set(l1), set(l2))) > 0


def load_karpathy_test(json_path: str, image_dir: Optional[str] = None) -> List[dict]:
    """The "test" split of a Karpathy JSON with x["image"] set the way the two scripts do:
    <dir of json>/<x.filepath>/<x.filename> for MSCOCO (eval_mscoco.py:66-67), <dir of json>/<image_dir>/
    <x.filename> for Flickr30k (eval_flickr30k.py:65-68, image_dir = "flickr30k_images")."""
    with open(json_path, "rb") as f:
        images = json.loads(f.read())["images"]
    test = list(filter(lambda x: x["split"] == "test", images))
    root = os.path.dirname(json_path)
    for x in test:
        sub = image_dir if image_dir is not None else x["filepath"]
        x["image"] = os.path.join(root, sub, x["filename"])
    return test


def embed_test_split(test: Sequence[dict], fxn: Callable) -> Tuple[list, list]:
    """(images, text) lists of (id, Tensor[1, D]) in the reference's order (eval_mscoco.py:69-74)."""
    text = []
    for x in test:
        text.extend(x["sentences"])
    images = [(i["image"], fxn(i["image"], dtype="image")) for i in test]
    text = [(i["raw"], fxn(i["raw"], dtype="text")) for i in text]
    return images, text


def recall_i2t(test: Sequence[dict], cand: Dict[object, list]) -> float:
    """eval_mscoco.py:76-84: hit if any of the image's captions is among its predictions."""
    acc = 0
    for x in test:
        targets = [i["raw"] for i in x["sentences"]]
        preds = cand[x["image"]]
        if intersect(preds, targets):
            acc += 1
    return acc / len(test)


def recall_t2i(test: Sequence[dict], cand: Dict[object, list]) -> float:
    """eval_mscoco.py:87-99: one trial per (image, caption) pair, hit if the image is among the
    predictions stored under the caption string."""
    acc = 0
    cntr = 0
    for x in test:
        for t in x["sentences"]:
            targets = [x["image"]]
            preds = cand[t["raw"]]
            if intersect(preds, targets):
                acc += 1
            cntr += 1
    return acc / cntr


def eval_retrieval(test: Sequence[dict], images: list, text: list, compute: str = "f32", out=print) -> Dict[str, float]:
    """Both directions, k in (1, 5, 10); prints the reference's lines ("i2t top1 is 0.1234") and returns
    {"i2t@1": ..., ..., "t2i@10": ...}."""
    if not test:
        raise AbcError("empty test split")
    results = {}
    i2t = search.get_topk_candidates_multi(images, text, TOPKS, compute=compute)
    for topk in TOPKS:
        acc = recall_i2t(test, i2t[topk])
        out(f"i2t top{topk} is {acc:.4f}")
        results[f"i2t@{topk}"] = acc
    t2i = search.get_topk_candidates_multi(text, images, TOPKS, compute=compute)
    for topk in TOPKS:
        acc = recall_t2i(test, t2i[topk])
        out(f"t2i top{topk} is {acc:.4f}")
        results[f"t2i@{topk}"] = acc
    return results


def eval_mscoco(fxn: Callable, compute: str = "f32", out=print) -> Dict[str, float]:
    path = os.environ.get("MSCOCO_EVAL", None)
    assert path is not None
    test = load_karpathy_test(path)
    images, text = embed_test_split(test, fxn)
    return eval_retrieval(test, images, text, compute=compute, out=out)


def eval_flickr(fxn: Callable, compute: str = "f32", out=print) -> Dict[str, float]:
    path = os.environ.get("FLICKR_EVAL", None)
    assert path is not None
    test = load_karpathy_test(path, image_dir="flickr30k_images")
    images, text = embed_test_split(test, fxn)
    return eval_retrieval(test, images, text, compute=compute, out=out)
