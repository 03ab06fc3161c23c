"""Generate golden fixtures by running the REFERENCE's own importable code on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    python oracle/gen_golden.py

Imports ``model/abc_util.py`` (compute_contrastive_loss / compute_gathered_loss) and
``visualization/neg_mine.py`` (batch) from the reference, and compiles the hot-path function definitions of
``evaluate/eval_mscoco.py``, ``evaluate/eval_flickr30k.py`` and ``pretrain/create_negative_index.py`` out of those
files (their modules cannot be imported here), runs them on CPU and stores inputs + outputs as small files under
tests/golden/.  Only data is stored, never reference source.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REF = os.environ.get("ABC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _ref():
    sys.path.insert(0, REF)
    from model import abc_util  # noqa
    from visualization import neg_mine  # noqa
    return abc_util, neg_mine


class _Temp(torch.nn.Module):
    # stand-in with the contract of model/modeling_abc.py:36-43 (that file needs `timm`)
    def __init__(self, t):
        super().__init__()
        self.temp = torch.nn.Parameter(torch.tensor(t, dtype=torch.float32))

    def forward(self, x):
        return x / (self.temp.float())


def bf16_bits(x: torch.Tensor) -> np.ndarray:
    return x.to(torch.bfloat16).view(torch.int16).numpy().copy()


def _unit(n, d, g):
    return torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1)


def gen_loss(abc_util):
    cases = {}
    n = 0
    for B in (4, 32):
        for mult in (1, 8):
            for eps in (0.0, 0.1):
                for t in (0.07, 0.02):
                    g = torch.Generator().manual_seed(100 + n)
                    D = 64
                    q = _unit(B, D, g).requires_grad_(True)
                    p = _unit(B * mult, D, g)
                    # make positives correlated so accuracy is non-trivial
                    p[:B] = torch.nn.functional.normalize(q.detach() + 0.8 * p[:B], dim=-1)
                    p.requires_grad_(True)
                    temp = _Temp(t)
                    loss, acc = abc_util.compute_contrastive_loss(q, p, temp, label_smoothing=eps)
                    loss.backward()
                    pre = f"c{n}_"
                    cases.update({pre + "q": q.detach().numpy(), pre + "p": p.detach().numpy(),
                                  pre + "temp": np.float32(t), pre + "eps": np.float32(eps),
                                  pre + "loss": loss.detach().numpy(), pre + "acc": acc.numpy(),
                                  pre + "dq": q.grad.numpy(), pre + "dp": p.grad.numpy(),
                                  pre + "dtemp": temp.temp.grad.numpy()})
                    n += 1
    cases["n_cases"] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, "loss_small.npz"), **cases)

    # BASELINE config 4 on one rank: B=128, C=1024, D=4096, eps=0.1, temp=0.07; seed-defined inputs
    g = torch.Generator().manual_seed(4242)
    q = _unit(128, 4096, g)
    p = _unit(1024, 4096, g)
    p[:128] = torch.nn.functional.normalize(q + 0.9 * p[:128], dim=-1)
    q.requires_grad_(True)
    p.requires_grad_(True)
    temp = _Temp(0.07)
    loss, acc = abc_util.compute_contrastive_loss(q, p, temp, label_smoothing=0.1)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "loss_cfg4.npz"), seed=np.int64(4242), loss=loss.detach().numpy(),
                        acc=acc.numpy(), dtemp=temp.temp.grad.numpy(),
                        dq_rows=q.grad[:4].numpy(), dp_rows=p.grad[:4].numpy(),
                        dq_colsum=q.grad.sum(0).numpy(), dp_colsum=p.grad.sum(0).numpy(),
                        dq_abs=q.grad.abs().sum().numpy(), dp_abs=p.grad.abs().sum().numpy())


def _gather_worker(rank, world, port, B, mult, D, eps, t, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    abc_util, _ = _ref()
    g = torch.Generator().manual_seed(7000 + rank)
    q = _unit(B, D, g)
    c = _unit(B * mult, D, g)
    c[:B] = torch.nn.functional.normalize(q + 0.8 * c[:B], dim=-1)
    q.requires_grad_(True)
    c.requires_grad_(True)
    temp = _Temp(t)
    loss, acc, num = abc_util.compute_gathered_loss(q, c, temp, label_smoothing=eps)
    loss.backward()
    ret[rank] = dict(q=q.detach().numpy(), c=c.detach().numpy(), loss=loss.detach().numpy(), acc=acc.numpy(),
                     num=num.numpy(), dq=q.grad.numpy(), dc=c.grad.numpy(), dtemp=temp.temp.grad.numpy())
    dist.destroy_process_group()


def gen_gathered():
    out = {}
    port = 29611
    for world in (2, 4):
        mgr = mp.Manager()
        ret = mgr.dict()
        B, mult, D, eps, t = 4, 2, 64, 0.1, 0.07
        mp.spawn(_gather_worker, args=(world, port, B, mult, D, eps, t, ret), nprocs=world, join=True)
        port += 1
        for r in range(world):
            for k, v in ret[r].items():
                out[f"w{world}_r{r}_{k}"] = v
        out[f"w{world}_cfg"] = np.array([B, mult, D], dtype=np.int64)
        out[f"w{world}_eps"] = np.float32(eps)
        out[f"w{world}_temp"] = np.float32(t)
    np.savez_compressed(os.path.join(OUT, "gathered_gloo.npz"), **out)


def gen_neg_mine(neg_mine):
    from oracle.search_oracle import make_clustered
    out = {}
    # small case with stored bf16-exact inputs
    q, c = make_clustered(512, 512, 128, seed=31, n_centres=16)
    q, c = q.to(torch.bfloat16).float(), c.to(torch.bfloat16).float()
    out["s_q_bf16"] = bf16_bits(q)
    out["s_c_bf16"] = bf16_bits(c)
    for k in (1, 5, 10, 64, 100):
        with contextlib.redirect_stdout(io.StringIO()):
            sc, ix, frac, pos = neg_mine.batch(q, c, 0, 3000, k)
        out[f"s_k{k}_score"] = sc.numpy()
        out[f"s_k{k}_idx"] = ix.numpy().astype(np.int32)
    out["s_frac"] = frac.numpy()
    out["s_pos"] = pos.numpy()
    # two batches of the same data (idx offset path, neg_mine.py:23-24,32)
    with contextlib.redirect_stdout(io.StringIO()):
        sc, ix, frac, pos = neg_mine.batch(q, c, 300, 200, 10)
    out["s_b300_score"], out["s_b300_idx"] = sc.numpy(), ix.numpy().astype(np.int32)
    out["s_b300_frac"], out["s_b300_pos"] = frac.numpy(), pos.numpy()
    # larger seed-defined case: inputs regenerated by the test from the seed
    q, c = make_clustered(3000, 4000, 256, seed=77, n_centres=64)
    q, c = q.to(torch.bfloat16).float(), c.to(torch.bfloat16).float()
    with contextlib.redirect_stdout(io.StringIO()):
        sc, ix, frac, pos = neg_mine.batch(q, c, 0, 3000, 10)
    out["l_seed"] = np.int64(77)
    out["l_score"], out["l_idx"] = sc.numpy(), ix.numpy().astype(np.int32)
    out["l_frac"], out["l_pos"] = frac.numpy(), pos.numpy()
    out["l_q_sum"] = q.double().sum().numpy()
    out["l_c_sum"] = c.double().sum().numpy()
    np.savez_compressed(os.path.join(OUT, "neg_mine_batch.npz"), **out)


def gen_mean_pool(abc_util):
    """abc_util.get_mean_token_embed (model/abc_util.py:33-41) on seeded inputs: fp32 and bf16 hidden states,
    with and without an instruction mask; padded tails, one all-pad row."""
    g = torch.Generator().manual_seed(77)
    B, S, H = 6, 19, 64
    hidden = torch.randn(B, S, H, generator=g)
    ids = torch.randint(1, 1000, (B, S), generator=g)
    lens = [19, 12, 1, 7, 0, 15]
    for b, n in enumerate(lens):
        ids[b, n:] = 0
    instr = (torch.rand(B, S, generator=g) > 0.3).float()
    out = {"ids": ids.numpy(), "hidden": hidden.numpy(), "instr": instr.numpy(), "pad": np.int64(0)}
    out["f32_plain"] = abc_util.get_mean_token_embed(ids, hidden, 0).numpy()
    out["f32_instr"] = abc_util.get_mean_token_embed(ids, hidden, 0, instruction_mask=instr).numpy()
    hb = hidden.to(torch.bfloat16)
    out["bf16_plain_bits"] = bf16_bits(abc_util.get_mean_token_embed(ids, hb, 0))
    out["bf16_instr_bits"] = bf16_bits(abc_util.get_mean_token_embed(ids, hb, 0, instruction_mask=instr.to(torch.bfloat16)))
    np.savez_compressed(os.path.join(OUT, "mean_pool.npz"), **out)


# ----------------------------------------------------------------------------------------------------------------
# a1 / a2 / a3 glue: evaluate/eval_mscoco.py, evaluate/eval_flickr30k.py, pretrain/create_negative_index.py.
# These files cannot be imported as modules here (their module-level imports need transformers 4.44 / Python 3.12,
# SURVEY.md 8c), but the hot-path FUNCTIONS in them are plain torch.  Their definitions are compiled straight out of
# the reference files (ast: the named top-level `def`s only, nothing else of the module runs) and executed on CPU with
# `.cuda()` / `.cpu()` as the identity.  Only inputs and outputs are stored.
# ----------------------------------------------------------------------------------------------------------------
def _ref_functions(relpath, names, env):
    import ast
    path = os.path.join(REF, relpath)
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), (relpath, [n.name for n in body])
    ns = dict(env)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


@contextlib.contextmanager
def _cuda_is_identity():
    keep = torch.Tensor.cuda, torch.Tensor.cpu
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.cpu = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.Tensor.cpu = keep


def _grid(shape, g, amp=127):
    """Integers in [-amp, amp]: embeddings are grid / 128, so every product is a multiple of 2^-14 and every partial sum of
    a d <= 64 dot product is exact in fp32 (and the values themselves in bf16): CPU BLAS and any MFMA order agree bit for
    bit, and equal scores are TRUE ties."""
    return torch.randint(-amp, amp + 1, shape, generator=g, dtype=torch.int64)


def _exact_scores(a, b):
    return a @ b.t()                                   # int64, units of 2^-14


def _karpathy(n_img, g):
    """A Karpathy-shaped split file: test / train / val images, 5 captions each, some caption strings used by two images
    (the dict of get_topk_candidates collapses them, eval_mscoco.py:45-48)."""
    images, n = [], 0
    for i in range(n_img + 12):
        split = "test" if i < n_img else ("train" if i % 2 else "val")
        sents = []
        for j in range(5):
            raw = f"caption {n} of image {i}"
            if split == "test" and i % 9 == 4 and j == 2:
                raw = f"a shared caption {i // 18}"        # images 4 and 13, 22 and 31, ... share one string
            sents.append({"raw": raw, "sentid": n, "imgid": i})
            n += 1
        images.append({"split": split, "filepath": "val2014", "filename": f"COCO_val2014_{i:012d}.jpg",
                       "sentences": sents, "imgid": i})
    return {"images": images, "dataset": "synthetic"}


def gen_glue():
    import json
    import tqdm as _tqdm
    D, n_img = 64, 60
    fix = {}
    # ---- a1 / a2: eval_mscoco / eval_flickr run whole, on a synthetic split and a table-lookup embedding function
    for seed in range(1000, 1200):
        g = torch.Generator().manual_seed(seed)
        kj = _karpathy(n_img, g)
        test = [x for x in kj["images"] if x["split"] == "test"]
        caps = sorted({s["raw"] for x in test for s in x["sentences"]})
        img_grid = _grid((n_img, D), g)
        cap_grid = _grid((len(caps), D), g)
        # positives: caption = its image + noise, so recall is neither 0 nor 1
        cap_of = {c: i for i, c in enumerate(caps)}
        for i, x in enumerate(test):
            for s in x["sentences"]:
                r = cap_of[s["raw"]]
                cap_grid[r] = torch.clamp((img_grid[i] * (1 + (i + r) % 3)) // 16 + cap_grid[r] // 2, -127, 127)
        # no exact ties around any cut the evaluation makes (k = 1, 5, 10), in either direction: the reference's topk
        # order among ties is unspecified
        text_rows = torch.tensor([cap_of[s["raw"]] for x in test for s in x["sentences"]])
        S = _exact_scores(img_grid, cap_grid[text_rows])                  # i2t: duplicated strings = duplicated columns
        ok = True
        for M, ids in ((S, [s["raw"] for x in test for s in x["sentences"]]), (S.t().contiguous(), list(range(n_img)))):
            top = torch.sort(M, dim=1, descending=True, stable=True)
            for r in range(M.shape[0]):
                v, ix = top.values[r, :12].tolist(), top.indices[r, :12].tolist()
                for a in range(11):
                    if v[a] == v[a + 1] and ids[ix[a]] != ids[ix[a + 1]]:
                        ok = False
        if ok:
            break
    assert ok, "no tie-free seed found"
    fix["eval_seed"] = np.int64(seed)
    fix["eval_img_grid"] = img_grid.numpy().astype(np.int8)
    fix["eval_cap_grid"] = cap_grid.numpy().astype(np.int8)
    fix["eval_caps"] = np.array(caps)
    with open(os.path.join(OUT, "glue_karpathy.json"), "w") as f:
        json.dump(kj, f)

    img_emb = (img_grid.float() / 128).contiguous()
    cap_emb = (cap_grid.float() / 128).contiguous()

    def make_fxn(root, sub):
        img_row = {os.path.join(root, sub if sub is not None else x["filepath"], x["filename"]): i for i, x in enumerate(test)}

        def fxn(item, dtype="image"):
            return img_emb[img_row[item]].unsqueeze(0) if dtype == "image" else cap_emb[cap_of[item]].unsqueeze(0)
        return fxn

    env = {"torch": torch, "os": os, "json": json, "tqdm": lambda x, **k: x}
    json_path = os.path.join(OUT, "glue_karpathy.json")
    root = os.path.dirname(json_path)
    for name, rel, fn, var, sub in (("mscoco", "evaluate/eval_mscoco.py", "eval_mscoco", "MSCOCO_EVAL", None),
                                    ("flickr", "evaluate/eval_flickr30k.py", "eval_flickr", "FLICKR_EVAL", "flickr30k_images")):
        ns = _ref_functions(rel, ["intersect", "get_topk_candidates", fn], env)
        os.environ[var] = json_path
        buf = io.StringIO()
        with _cuda_is_identity(), contextlib.redirect_stdout(buf):
            ns[fn](make_fxn(root, sub))
        fix[f"{name}_lines"] = np.array(buf.getvalue().splitlines())
        # the dicts themselves (a1), ids as indices: image i / caption row of `caps`
        fxn = make_fxn(root, sub)
        paths = [os.path.join(root, sub if sub is not None else x["filepath"], x["filename"]) for x in test]
        images = [(p_, fxn(p_, dtype="image")) for p_ in paths]
        text = [(s["raw"], fxn(s["raw"], dtype="text")) for x in test for s in x["sentences"]]
        if name == "mscoco":
            with _cuda_is_identity():
                for k in (1, 3, 5, 10):
                    d_i2t = ns["get_topk_candidates"](images, text, k)
                    d_t2i = ns["get_topk_candidates"](text, images, k)
                    fix[f"i2t_k{k}"] = np.array([[cap_of[c] for c in d_i2t[p_]] for p_ in paths], dtype=np.int32)
                    fix[f"t2i_k{k}"] = np.array([[paths.index(p_) for p_ in d_t2i[c]] for c in caps], dtype=np.int32)
                    assert len(d_i2t) == n_img and len(d_t2i) == len(caps)      # duplicated strings collapsed
                dflt = ns["get_topk_candidates"](images, text)                  # default k = 3
                assert all(len(v) == 3 for v in dflt.values())

    # ---- a3: negatives_for_sample on row-aligned pairs, with the edge cases of SURVEY.md 8a
    ns = _ref_functions("pretrain/create_negative_index.py", ["negatives_for_sample"], {"torch": torch, "inf": float("inf")})
    g = torch.Generator().manual_seed(2024)
    N = 320
    c_grid = _grid((N, D), g)
    q_grid = torch.clamp(c_grid // 2 + _grid((N, D), g) // 3, -127, 127)        # row i of c is the positive of query i
    q_grid[5] = -c_grid[5]                               # negative positive score: the threshold keeps the positive
    q_grid[6] = torch.clamp(-c_grid[6] // 2 + _grid((1, D), g)[0] // 4, -127, 127)
    c_grid[40] = c_grid[7]                               # duplicated corpus rows: exact ties in every query
    c_grid[41] = c_grid[7]
    c_grid[200] = c_grid[199]
    q_grid[300] = 0                                      # all scores 0: the positive scores 0, threshold 0, nothing masked
    c_grid[310:316] = c_grid[309]                        # six copies of a positive: near-duplicates above 0.95 * s_pos
    q_grid[309] = c_grid[309]
    qe, ce = (q_grid.float() / 128).contiguous(), (c_grid.float() / 128).contiguous()
    negative_dict = {}
    with _cuda_is_identity(), torch.no_grad():
        for idx in range(N):
            ns["negatives_for_sample"](idx, negative_dict, qe, ce.clone())
    assert list(negative_dict) == list(range(N)) and all(len(v) == 100 for v in negative_dict.values())
    fix["mine_q_grid"] = q_grid.numpy().astype(np.int8)
    fix["mine_c_grid"] = c_grid.numpy().astype(np.int8)
    fix["mine_negatives"] = np.array([negative_dict[i] for i in range(N)], dtype=np.int32)
    fix["mine_json_sha256"] = np.array(__import__("hashlib").sha256(json.dumps(negative_dict).encode()).hexdigest())
    np.savez_compressed(os.path.join(OUT, "glue.npz"), **fix)


def _ref_classes(relpath, names, env):
    """The named top-level class definitions of a reference file that cannot be imported (see _ref_functions)."""
    import ast
    path = os.path.join(REF, relpath)
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), (relpath, [n.name for n in body])
    ns = dict(env)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def gen_head():
    """model/modeling_abc.py:9-43 -- Linear, MLP, Temperature -- compiled out of the file (the module needs timm /
    flash_attn) and run on CPU: a fresh MLP (identity / zero initialisation), a trained-looking one through the training
    forward's chain F.normalize(mlp_head(x).float(), dim=-1) (:151-154) with autograd gradients of x and of the four
    parameters, and Temperature.forward."""
    from torch import nn
    from torch.nn import init
    ns = _ref_classes("model/modeling_abc.py", ["Linear", "MLP", "Temperature"], {"torch": torch, "nn": nn, "init": init})
    MLP, Temperature = ns["MLP"], ns["Temperature"]
    out = {}
    g = torch.Generator().manual_seed(2025)
    H, Hd, B = 64, 72, 9
    fresh = MLP(H, Hd)
    out["fresh_w1"], out["fresh_b1"] = fresh.linear_layer1.weight.detach().numpy(), fresh.linear_layer1.bias.detach().numpy()
    out["fresh_w2"], out["fresh_b2"] = fresh.linear_layer2.weight.detach().numpy(), fresh.linear_layer2.bias.detach().numpy()
    assert sorted(fresh.state_dict()) == ["linear_layer1.bias", "linear_layer1.weight", "linear_layer2.bias", "linear_layer2.weight"]
    x = 0.5 * torch.randn(B, H, generator=g)
    with torch.no_grad():
        out["x"], out["fresh_y"] = x.numpy(), fresh(x).numpy()
    mlp = MLP(H, Hd)
    with torch.no_grad():
        for p in mlp.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    for k, v in mlp.state_dict().items():
        out["p_" + k] = v.detach().numpy().copy()
    xg = x.clone().requires_grad_(True)
    emb = torch.nn.functional.normalize(mlp(xg).float(), dim=-1)
    gout = torch.randn(B, H, generator=g) / H ** 0.5
    emb.backward(gout)
    out["emb"], out["gout"], out["dx"] = emb.detach().numpy(), gout.numpy(), xg.grad.numpy()
    out["dw1"], out["db1"] = mlp.linear_layer1.weight.grad.numpy(), mlp.linear_layer1.bias.grad.numpy()
    out["dw2"], out["db2"] = mlp.linear_layer2.weight.grad.numpy(), mlp.linear_layer2.bias.grad.numpy()
    with torch.no_grad():
        out["y"] = mlp(x).numpy()
    t = Temperature()
    out["temp_init"] = t.temp.detach().numpy()
    s = torch.randn(5, 7, generator=g)
    out["temp_in"], out["temp_out"] = s.numpy(), t(s).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "head.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    abc_util, neg_mine = _ref()
    torch.set_num_threads(4)
    gen_loss(abc_util)
    gen_neg_mine(neg_mine)
    gen_gathered()
    gen_mean_pool(abc_util)
    gen_glue()
    gen_head()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
