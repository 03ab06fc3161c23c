"""Helpers for tests/golden/glue.npz + glue_karpathy.json: outputs of the reference's OWN get_topk_candidates /
eval_mscoco / eval_flickr / negatives_for_sample (oracle/gen_golden.py::gen_glue compiled those function definitions out
of /root/reference/evaluate/eval_mscoco.py, eval_flickr30k.py and pretrain/create_negative_index.py and ran them on CPU).

The embeddings are integers / 128 with d = 64, so every score is an exact multiple of 2^-14 in fp32 whatever the
summation order: the CPU reference, the oracle and the MFMA kernels produce bit-identical scores, and equal scores are
true ties -- the only place where the reference's order (torch.topk / argsort, unspecified among ties) may differ."""
import json
import os

import numpy as np
import torch


class Glue:
    def __init__(self, golden_dir):
        self.z = np.load(os.path.join(golden_dir, "glue.npz"))
        self.json_path = os.path.join(golden_dir, "glue_karpathy.json")
        with open(self.json_path) as f:
            self.test = [x for x in json.load(f)["images"] if x["split"] == "test"]
        self.caps = [str(c) for c in self.z["eval_caps"]]
        self.cap_of = {c: i for i, c in enumerate(self.caps)}
        self.img_emb = torch.from_numpy(self.z["eval_img_grid"].astype(np.float32) / 128)
        self.cap_emb = torch.from_numpy(self.z["eval_cap_grid"].astype(np.float32) / 128)

    def lines(self, name):
        return [str(x) for x in self.z[f"{name}_lines"]]

    def paths(self, sub=None):
        root = os.path.dirname(self.json_path)
        return [os.path.join(root, sub if sub is not None else x["filepath"], x["filename"]) for x in self.test]

    def fxn(self, sub=None):
        """The embedding function the eval scripts are handed (evaluate/embed_function.py's contract: [1, D] fp32)."""
        row = {p: i for i, p in enumerate(self.paths(sub))}

        def f(item, dtype="image"):
            return self.img_emb[row[item]].unsqueeze(0) if dtype == "image" else self.cap_emb[self.cap_of[item]].unsqueeze(0)
        return f

    def lists(self, sub=None):
        f = self.fxn(sub)
        images = [(p, f(p, dtype="image")) for p in self.paths(sub)]
        text = [(s["raw"], f(s["raw"], dtype="text")) for x in self.test for s in x["sentences"]]
        return images, text

    def check_dicts(self, get_topk_candidates, ks=(1, 3, 5, 10)):
        """get_topk_candidates(queries, candidates, k) against the dicts the reference returned (both directions)."""
        images, text = self.lists()
        paths = self.paths()
        for k in ks:
            d = get_topk_candidates(images, text, k)
            assert list(d) == paths
            got = [[self.cap_of[c] for c in d[p]] for p in paths]
            assert got == self.z[f"i2t_k{k}"].tolist(), f"i2t k={k}"
            d = get_topk_candidates(text, images, k)
            assert sorted(d) == sorted(self.caps) and len(d) == len(self.caps)      # duplicated strings collapsed
            got = [[paths.index(p) for p in d[c]] for c in self.caps]
            assert got == self.z[f"t2i_k{k}"].tolist(), f"t2i k={k}"

    # ---- a3 -----------------------------------------------------------------------------------------------------
    def mining_inputs(self):
        q = torch.from_numpy(self.z["mine_q_grid"].astype(np.float32) / 128)
        c = torch.from_numpy(self.z["mine_c_grid"].astype(np.float32) / 128)
        return q, c

    def check_negatives(self, negative_dict, rows=None):
        """negative_dict[idx] against the reference's lists: the same exact score at every position among the survivors
        of the 0.95 rule, the same index wherever that score is unique in the row, and a tail (fewer than 100
        survivors) made of distinct masked columns, as the reference's argsort over -inf entries gives."""
        qg, cg = self.z["mine_q_grid"].astype(np.int64), self.z["mine_c_grid"].astype(np.int64)
        ref = self.z["mine_negatives"]
        n = qg.shape[0]
        stats = {"rows": 0, "tied_positions": 0, "short_rows": 0}
        for idx in (range(n) if rows is None else rows):
            s_int = cg @ qg[idx]                                        # units of 2^-14, exact
            s = (s_int.astype(np.float64) / 16384).astype(np.float32)   # exactly the fp32 scores
            thr = np.float32(s[idx]) * np.float32(0.95)                 # scores[idx] * .95 in fp32
            masked = s > thr
            n_surv = int((~masked).sum())
            m = min(100, n_surv)
            got, want = list(negative_dict[idx]), ref[idx].tolist()
            assert len(got) == 100, (idx, len(got))
            assert len(set(got)) == 100, f"row {idx}: duplicates"
            assert all(0 <= j < n for j in got)
            assert not masked[want[:m]].any()                           # sanity of the checker against the reference
            for j in range(m):
                assert not masked[got[j]], f"row {idx} pos {j}: masked column {got[j]} returned"
                assert s_int[got[j]] == s_int[want[j]], f"row {idx} pos {j}: score differs from the reference"
                if (s_int[~masked] == s_int[want[j]]).sum() == 1:
                    assert got[j] == want[j], f"row {idx} pos {j}"
                else:
                    stats["tied_positions"] += 1
            assert masked[got[m:]].all(), f"row {idx}: tail holds an unmasked column"
            assert masked[want[m:]].all()
            stats["rows"] += 1
            stats["short_rows"] += m < 100
        return stats
