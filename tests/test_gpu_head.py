"""GPU parity of the embedding epilogue: mean pool against the reference's own
outputs (tests/golden/mean_pool.npz) and the oracle, the fused MLP head + normalise + quantise against
the oracle's restatement of model/modeling_abc.py:16-34,151-154."""
import os

import numpy as np
import pytest
import torch

import oracle
from vllm_amd import head, kernels, search
from vllm_amd._lib import AbcError

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf16(bits):
    return torch.from_numpy(bits.astype(np.int16)).view(torch.bfloat16)


def test_mean_pool_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "mean_pool.npz"))
    ids, hidden, instr = (torch.from_numpy(z[k]).to(DEV) for k in ("ids", "hidden", "instr"))
    pad = int(z["pad"])
    # fp32: sequential fp32 accumulation vs torch's reduction order -> 1e-6 relative to the row scale
    for key, im in (("f32_plain", None), ("f32_instr", instr)):
        got = head.get_mean_token_embed(ids, hidden, pad, im).cpu()
        ref = torch.from_numpy(z[key])
        assert got.dtype == torch.float32
        assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        assert float(got[4].abs().max()) == 0.0                                  # all-pad row
    # bf16: same rounding points as the reference's bf16 tensors -> at most 1 bf16 ulp from summation order
    hb = hidden.to(torch.bfloat16)
    for key, im in (("bf16_plain_bits", None), ("bf16_instr_bits", instr)):
        got = head.get_mean_token_embed(ids, hb, pad, im).cpu()
        ref = _bf16(z[key])
        assert got.dtype == torch.bfloat16
        same = (got.view(torch.int16) == ref.view(torch.int16)).float().mean()
        assert float(same) > 0.97
        assert float((got.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())


@pytest.mark.parametrize("B,S,H", [(5, 33, 896), (128, 64, 3584), (1, 1, 8)])
def test_mean_pool_shapes_vs_oracle(B, S, H):
    g = torch.Generator().manual_seed(B * 1000 + S)
    hidden = torch.randn(B, S, H, generator=g)
    ids = torch.randint(1, 50, (B, S), generator=g)
    for b in range(B):
        ids[b, int(torch.randint(0, S + 1, (1,), generator=g)):] = 0
    ref = oracle.get_mean_token_embed(ids, hidden, 0)
    got = head.get_mean_token_embed(ids.to(DEV), hidden.to(DEV), 0).cpu()
    assert float((got - ref).abs().max()) <= 2e-6 * max(float(ref.abs().max()), 1e-6)


@pytest.mark.parametrize("B,H,Hd", [(128, 3584, 4096), (7, 896, 4096), (64, 4096, 4096), (3, 64, 72)])
def test_embed_head_vs_oracle(B, H, Hd):
    g = torch.Generator().manual_seed(H + B)
    ref_mlp = oracle.OracleMLP(H, Hd)
    with torch.no_grad():                          # a trained-looking head, not the identity initialisation
        for p in ref_mlp.parameters():
            p.add_(0.02 * torch.randn(p.shape, generator=g))
    mlp = head.MLP(H, Hd)
    mlp.load_state_dict(ref_mlp.state_dict())
    mlp = mlp.to(DEV)
    x = 0.5 * torch.randn(B, H, generator=g)
    with torch.no_grad():
        ref_y = ref_mlp(x.double().float())
        ref64 = x.double() + torch.nn.functional.linear(
            torch.nn.functional.selu(torch.nn.functional.linear(x.double(), ref_mlp.linear_layer1.weight.double(),
                                                                ref_mlp.linear_layer1.bias.double())),
            ref_mlp.linear_layer2.weight.double(), ref_mlp.linear_layer2.bias.double())
        got_y = mlp(x.to(DEV)).cpu()
        # judged against fp64: the HIP result is as close to it as torch-CPU fp32 is (x4), and 1e-5 of the row scale
        scale = float(ref64.abs().max())
        err_ref = float((ref_y.double() - ref64).abs().max())
        err_got = float((got_y.double() - ref64).abs().max())
        assert err_got <= max(4 * err_ref, 1e-5 * scale), (err_got, err_ref)
        ref_n = torch.nn.functional.normalize(ref64, dim=-1)
        out, o16, o8, sc = head._head(mlp, x.to(DEV), normalize=True, want_bf16=True, want_i8=True)
        assert float((out.cpu().double() - ref_n).abs().max()) <= 1e-5 * float(ref_n.abs().max())
        assert float((out.norm(dim=-1) - 1).abs().max()) < 1e-5
        # the quantised forms are exactly what the library's own quantisers make of the fp32 result
        assert torch.equal(o16, kernels.cast_bf16(out))
        q8, s8 = kernels.quantize_i8(out)
        assert torch.equal(o8, q8) and torch.equal(sc, s8)
        qa, sa = oracle.quantize_i8(out.cpu())
        assert torch.equal(o8.cpu(), qa) and torch.equal(sc.cpu(), sa)


def test_identity_initialised_head_and_pipeline():
    """Fresh MLP (identity weights, zero bias, modeling_abc.py:9-14): y = x + selu(x).  Then the whole
    embed path (pool -> head -> normalise) against the oracle, feeding the search directly."""
    B, S, H = 40, 21, 256
    g = torch.Generator().manual_seed(9)
    mlp = head.MLP(H, H).to(DEV)
    x = torch.randn(B, H, generator=g)
    with torch.no_grad():
        y = mlp(x.to(DEV)).cpu()
    assert torch.allclose(y, x + torch.nn.functional.selu(x), rtol=1e-6, atol=1e-6)
    hidden = torch.randn(B, S, H, generator=g)
    ids = torch.randint(1, 9, (B, S), generator=g)
    ids[:, 15:] = 0
    instr = (torch.rand(B, S, generator=g) > 0.5).float()
    with torch.no_grad():
        ref = oracle.embed_ref(ids, hidden, 0, oracle.OracleMLP(H, H), instr)
        out, emb = head.embed_head(ids.to(DEV), hidden.to(DEV), 0, mlp, instr.to(DEV), compute="bf16")
    assert float((out.cpu() - ref).abs().max()) <= 2e-6
    assert emb.compute == "bf16" and emb.data.dtype == torch.bfloat16
    v, i, _ = search.topk(emb, emb, 1)
    assert i.flatten().tolist() == list(range(B))                     # every vector's nearest neighbour is itself


def test_head_refuses_training_and_cpu():
    mlp = head.MLP(64, 64).to(DEV)
    x = torch.randn(4, 64, device=DEV)
    with pytest.raises(AbcError):
        mlp(x)                                                        # grad enabled, parameters require grad
    with torch.no_grad(), pytest.raises(AbcError):
        mlp(x.cpu())
    with torch.no_grad(), pytest.raises(AbcError):
        head.get_mean_token_embed(torch.zeros(2, 3, dtype=torch.long), torch.zeros(2, 3, 8), 0)
