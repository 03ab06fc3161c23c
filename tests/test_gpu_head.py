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


def test_head_refuses_cpu_and_quantised_outputs_under_grad():
    mlp = head.MLP(64, 64).to(DEV)
    x = torch.randn(4, 64, device=DEV)
    with pytest.raises(AbcError):
        head._head(mlp, x, normalize=True, want_bf16=True)            # grad enabled: the corpus forms are inference outputs
    with torch.no_grad(), pytest.raises(AbcError):
        mlp(x.cpu())
    with torch.no_grad(), pytest.raises(AbcError):
        head.get_mean_token_embed(torch.zeros(2, 3, dtype=torch.long), torch.zeros(2, 3, 8), 0)


def _trained_pair(H, Hd, g):
    ref_mlp = oracle.OracleMLP(H, Hd)
    with torch.no_grad():
        for p in ref_mlp.parameters():
            p.add_(0.02 * torch.randn(p.shape, generator=g))
    mlp = head.MLP(H, Hd)
    mlp.load_state_dict(ref_mlp.state_dict())
    return ref_mlp, mlp.to(DEV)


def _judge(got, ref32, ref64, what):
    """The HIP gradient must be as close to the fp64 autograd result as torch-CPU fp32 autograd is (x4), or within 2e-6 of
    the gradient's scale (cancellation in the normalise backward makes element-wise rtol between two fp32 results
    meaningless, as for the loss: DESIGN.md section 2)."""
    scale = float(ref64.abs().max())
    e_got = float((got.double() - ref64).abs().max())
    e_ref = float((ref32.double() - ref64).abs().max())
    assert e_got <= max(4 * e_ref, 2e-6 * scale), (what, e_got, e_ref, scale)


@pytest.mark.parametrize("B,H,Hd,normalize", [(288, 3584, 3584, True), (37, 896, 1024, True), (5, 64, 72, True), (16, 128, 128, False)])
def test_head_backward_vs_autograd(B, H, Hd, normalize):
    """d(normalize(mlp_head(x))) for x and the four parameters against torch-CPU autograd of the oracle's restatement of
    model/modeling_abc.py:16-34,151-154 (fp32 and, as the judge, fp64).  288 x 3584 is one rank's 32 queries + 256
    candidates of config/qwen/QwenVL-8B.json."""
    g = torch.Generator().manual_seed(B + H)
    ref_mlp, mlp = _trained_pair(H, Hd, g)
    x = 0.5 * torch.randn(B, H, generator=g)
    gout = torch.randn(B, H, generator=g) / H ** 0.5

    def run_ref(dtype):
        m = oracle.OracleMLP(H, Hd).to(dtype)
        m.load_state_dict({k: v.to(dtype) for k, v in ref_mlp.state_dict().items()})
        xx = x.detach().clone().to(dtype).requires_grad_(True)
        y = m(xx)
        out = torch.nn.functional.normalize(y, dim=-1) if normalize else y
        out.backward(gout.to(dtype))
        return [xx.grad, m.linear_layer1.weight.grad, m.linear_layer1.bias.grad, m.linear_layer2.weight.grad,
                m.linear_layer2.bias.grad]

    r32, r64 = run_ref(torch.float32), run_ref(torch.float64)
    xd = x.detach().clone().to(DEV).requires_grad_(True)
    out = head._head(mlp, xd, normalize=normalize)[0]
    out.backward(gout.to(DEV))
    got = [xd.grad, mlp.linear_layer1.weight.grad, mlp.linear_layer1.bias.grad, mlp.linear_layer2.weight.grad,
           mlp.linear_layer2.bias.grad]
    for name, a, b, c in zip(("dx", "dW1", "db1", "dW2", "db2"), got, r32, r64):
        assert a is not None and tuple(a.shape) == tuple(c.shape), name
        _judge(a.cpu(), b, c, name)
    # parameters only (frozen input) and input only (frozen head): the unneeded GEMMs are skipped, results unchanged
    for p in mlp.parameters():
        p.grad = None
    head._head(mlp, x.to(DEV), normalize=normalize)[0].backward(gout.to(DEV))
    assert torch.equal(mlp.linear_layer2.weight.grad, got[3]) and torch.equal(mlp.linear_layer1.weight.grad, got[1])
    for p in mlp.parameters():
        p.requires_grad_(False)
    xd2 = x.detach().clone().to(DEV).requires_grad_(True)
    head._head(mlp, xd2, normalize=normalize)[0].backward(gout.to(DEV))
    assert torch.equal(xd2.grad, got[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mean_pool_backward_vs_autograd(dtype):
    """d hidden of get_mean_token_embed against autograd through the oracle's restatement of model/abc_util.py:33-41 (whose
    forward is pinned by the reference's own fixture): pads get exact zeros, the instruction mask scales, division by S."""
    B, S, H = 6, 19, 64
    g = torch.Generator().manual_seed(3)
    hidden = torch.randn(B, S, H, generator=g).to(dtype)
    ids = torch.randint(1, 50, (B, S), generator=g)
    for b, n in enumerate([19, 12, 1, 7, 0, 15]):
        ids[b, n:] = 0
    instr = (torch.rand(B, S, generator=g) > 0.3).float()
    gout = torch.randn(B, H, generator=g).to(dtype)
    for im in (None, instr):
        hr = hidden.clone().requires_grad_(True)
        oracle.get_mean_token_embed(ids, hr, 0, None if im is None else im.to(dtype)).backward(gout)
        hd = hidden.to(DEV).requires_grad_(True)
        out = head.get_mean_token_embed(ids.to(DEV), hd, 0, None if im is None else im.to(DEV))
        assert out.dtype == dtype
        out.backward(gout.to(DEV))
        got, ref = hd.grad.cpu(), hr.grad
        assert got.dtype == dtype and tuple(got.shape) == (B, S, H)
        if dtype == torch.float32:
            assert torch.equal(got, ref)
        else:
            assert float((got.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())
        assert float(got[4].float().abs().max()) == 0.0 and float(got[1, 12:].float().abs().max()) == 0.0


def test_training_forward_chain_end_to_end():
    """modeling_abc.py:148-157 as one graph on the GPU: hidden states -> mean pool -> head -> normalise ->
    compute_contrastive_loss, gradients back to the hidden states, the head and the temperature, against torch-CPU
    autograd over the oracle's restatements of the same chain."""
    from vllm_amd import loss as L
    B, S, H = 24, 11, 128
    g = torch.Generator().manual_seed(11)
    ref_mlp, mlp = _trained_pair(H, H, g)
    hq, hc = torch.randn(B, S, H, generator=g), torch.randn(B, S, H, generator=g)
    hc = hc + 0.7 * hq
    ids = torch.randint(1, 9, (B, S), generator=g)
    ids[:, 8:] = 0

    def chain(pool, mlp_, loss_fn, temp, hq_, hc_, dev):
        q = torch.nn.functional.normalize(mlp_(pool(ids.to(dev), hq_, 0)).float(), dim=-1) if mlp_ is ref_mlp else \
            head.embed_head(ids.to(dev), hq_, 0, mlp_)
        c = torch.nn.functional.normalize(mlp_(pool(ids.to(dev), hc_, 0)).float(), dim=-1) if mlp_ is ref_mlp else \
            head.embed_head(ids.to(dev), hc_, 0, mlp_)
        return loss_fn(q, c, temp, 0.1)

    t_ref = oracle.OracleTemperature(0.07)
    a, b = hq.clone().requires_grad_(True), hc.clone().requires_grad_(True)
    l_ref, acc_ref = chain(oracle.get_mean_token_embed, ref_mlp, oracle.compute_contrastive_loss, t_ref, a, b, "cpu")
    l_ref.backward()
    t_got = L.Temperature(0.07).to(DEV)
    ad, bd = hq.to(DEV).requires_grad_(True), hc.to(DEV).requires_grad_(True)
    l_got, acc_got = chain(None, mlp, L.compute_contrastive_loss, t_got, ad, bd, DEV)
    l_got.backward()
    assert abs(float(l_got) - float(l_ref)) <= 1e-5 * abs(float(l_ref)) and float(acc_got) == float(acc_ref)
    for name, x, y in (("d hidden_q", ad.grad.cpu(), a.grad), ("d hidden_c", bd.grad.cpu(), b.grad),
                       ("dW1", mlp.linear_layer1.weight.grad.cpu(), ref_mlp.linear_layer1.weight.grad),
                       ("dW2", mlp.linear_layer2.weight.grad.cpu(), ref_mlp.linear_layer2.weight.grad),
                       ("db2", mlp.linear_layer2.bias.grad.cpu(), ref_mlp.linear_layer2.bias.grad),
                       ("dtemp", t_got.temp.grad.cpu(), t_ref.temp.grad)):
        assert float((x - y).abs().max()) <= 2e-4 * float(y.abs().max()) + 1e-9, name


def test_head_vs_reference_classes_fixture(golden_dir):
    """The HIP head (forward, normalise, backward) and loss.Temperature against what model/modeling_abc.py:9-43's own MLP and
    Temperature returned on CPU (tests/golden/head.npz, oracle/gen_golden.py::gen_head): state-dict names load unchanged,
    fresh initialisation equal, forward within 1e-6 of the row scale, gradients within 2e-6 of their scale."""
    from vllm_amd import loss as L
    z = np.load(os.path.join(golden_dir, "head.npz"))
    t = lambda k: torch.from_numpy(np.asarray(z[k]))  # noqa: E731
    H, Hd = 64, 72
    fresh = head.MLP(H, Hd)
    assert torch.equal(fresh.linear_layer1.weight.detach(), t("fresh_w1")) and torch.equal(fresh.linear_layer2.weight.detach(), t("fresh_w2"))
    assert float(fresh.linear_layer1.bias.abs().max()) == 0.0 and float(fresh.linear_layer2.bias.abs().max()) == 0.0
    x = t("x")
    with torch.no_grad():
        y = fresh.to(DEV)(x.to(DEV)).cpu()
    assert float((y - t("fresh_y")).abs().max()) <= 1e-6 * float(t("fresh_y").abs().max())
    mlp = head.MLP(H, Hd)
    mlp.load_state_dict({k[2:]: t(k) for k in z.files if k.startswith("p_")})
    mlp = mlp.to(DEV)
    xd = x.clone().to(DEV).requires_grad_(True)
    emb = head._head(mlp, xd, normalize=True)[0]
    emb.backward(t("gout").to(DEV))
    close = lambda a, b, tol: float((a.cpu() - b).abs().max()) <= tol * float(b.abs().max())  # noqa: E731
    assert close(emb.detach(), t("emb"), 1e-6)
    for name, got in (("dx", xd.grad), ("dw1", mlp.linear_layer1.weight.grad), ("db1", mlp.linear_layer1.bias.grad),
                      ("dw2", mlp.linear_layer2.weight.grad), ("db2", mlp.linear_layer2.bias.grad)):
        assert close(got, t(name), 2e-6), name
    with torch.no_grad():
        assert close(mlp(x.to(DEV)), t("y"), 1e-6)
    temp = L.Temperature().to(DEV)
    assert torch.equal(temp.temp.detach().cpu(), t("temp_init"))
    assert torch.equal(temp(t("temp_in").to(DEV)).detach().cpu(), t("temp_out"))          # x / temp: one IEEE division
