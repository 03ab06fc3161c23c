"""CPU oracle: temperature-scaled, label-smoothed InfoNCE over all-gathered candidates.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Citations are relative to
/root/reference (snapshot 2025-03-01).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn.functional as F

__all__ = ["OracleTemperature", "compute_contrastive_loss", "compute_gathered_loss_from_blocks",
           "gathered_column_order", "infonce_closed_form", "get_mean_token_embed", "OracleMLP", "embed_ref"]


class OracleTemperature(torch.nn.Module):
    """Restates model/modeling_abc.py:36-43: scalar fp32 parameter, init 0.07, forward = x / temp."""

    def __init__(self, init: float = 0.07):
        super().__init__()
        self.temp = torch.nn.Parameter(torch.tensor(init, dtype=torch.float32))

    def forward(self, x):
        return x / (self.temp.float())


def compute_contrastive_loss(q_embeds, p_embeds, temperature, label_smoothing=0.0):
    """Restates model/abc_util.py:43-56."""
    bs = q_embeds.size(0)
    score = temperature(torch.matmul(q_embeds, p_embeds.t()))
    sim_targets = torch.arange(bs).to(score.device)
    loss = F.cross_entropy(score, sim_targets, label_smoothing=label_smoothing)
    _max_score, max_idxs = torch.max(score, 1)
    accuracy = (max_idxs == sim_targets).sum() / bs
    return loss, accuracy


def gathered_column_order(rank: int, world_size: int) -> List[int]:
    """Block order of c_global on ``rank`` after model/abc_util.py:19-22.

    all_gather fills slots 0..W-1 in rank order, slot[rank] is replaced by the local
    (grad-carrying) tensor, then slots 0 and rank are swapped.
    """
    order = list(range(world_size))
    order[0], order[rank] = order[rank], order[0]
    return order


def compute_gathered_loss_from_blocks(q_emb, c_blocks: List[torch.Tensor], rank: int, temperature,
                                      label_smoothing=0.0):
    """Single-process restatement of model/abc_util.py:5-31 given every rank's candidate block.

    ``c_blocks[r]`` is rank r's c_emb; only ``c_blocks[rank]`` keeps its autograd history
    (remote blocks arrive through the non-differentiable dist.all_gather, :19-21).
    """
    q_emb = q_emb.float()
    blocks = [b.float() if r == rank else b.detach().float() for r, b in enumerate(c_blocks)]
    order = gathered_column_order(rank, len(blocks))
    c_global = torch.cat([blocks[r] for r in order], dim=0)
    loss, acc = compute_contrastive_loss(q_emb, c_global, temperature, label_smoothing)
    num_cand = torch.tensor([c_global.size(0)])
    return loss, acc, num_cand


def infonce_closed_form(q: torch.Tensor, p: torch.Tensor, temp: float, eps: float, n_local: int
                        ) -> Tuple[torch.Tensor, ...]:
    """Closed form of a7, fp64, used to check the fused kernels.

    z=s/temp, logp=z-lse(z); loss=mean_i[(1-eps)(-logp_ii)+(eps/C)sum_j(-logp_ij)];
    g=(softmax(z)-[(1-eps)1{j=i}+eps/C])/B; dq=(g/temp)p; dp=(g/temp)^T q (first n_local rows
    only: the local block is slot 0 after the swap); dtemp=-sum(g*s)/temp^2.
    Returns (loss, lse[B], argmax[B], dq, dp_local, dtemp).
    """
    q, p = q.double(), p.double()
    B, C = q.shape[0], p.shape[0]
    s = q @ p.t()
    z = s / temp
    lse = torch.logsumexp(z, dim=1)
    logp = z - lse.unsqueeze(1)
    tgt = torch.full((B, C), eps / C, dtype=torch.float64)
    tgt[torch.arange(B), torch.arange(B)] += 1.0 - eps
    loss = -(tgt * logp).sum(1).mean()
    g = (logp.exp() - tgt) / B
    dq = (g / temp) @ p
    dp = (g / temp).t() @ q
    dtemp = -(g * s).sum() / (temp * temp)
    return loss, lse, z.argmax(1), dq, dp[:n_local], dtemp


# --------------------------------------------------------------------------------------
# f3: the embedding epilogue that feeds the path
# --------------------------------------------------------------------------------------
def get_mean_token_embed(input_ids, hidden_state, padding_token_id, instruction_mask=None):
    """Restates model/abc_util.py:33-41 verbatim (pinned by tests/golden/mean_pool.npz, which the
    reference function itself produced)."""
    if instruction_mask is not None:
        hidden_state = instruction_mask.unsqueeze(-1) * hidden_state
    mask = (input_ids != padding_token_id).unsqueeze(-1)
    masked_states = mask * hidden_state
    return torch.mean(masked_states, dim=1)


class OracleMLP(torch.nn.Module):
    """Restates model/modeling_abc.py:9-34 (Linear with identity/zero init, SELU, residual).  The module
    cannot be imported here (timm / flash_attn missing); pinned by tests/golden/head.npz, which the reference's own
    Linear / MLP classes (compiled out of the file by oracle/gen_golden.py::gen_head) produced."""

    def __init__(self, embed_size, hidden_size=None):
        super().__init__()
        hidden_size = hidden_size if hidden_size is not None else embed_size
        self.linear_layer1 = torch.nn.Linear(embed_size, hidden_size, dtype=torch.float32)
        self.linear_layer2 = torch.nn.Linear(hidden_size, embed_size, dtype=torch.float32)
        for lin in (self.linear_layer1, self.linear_layer2):
            torch.nn.init.eye_(lin.weight)
            torch.nn.init.zeros_(lin.bias)
        self.act = torch.nn.SELU()

    def forward(self, x):
        y = self.linear_layer1(x)
        y = self.act(y)
        y = self.linear_layer2(y)
        return x + y


def embed_ref(input_ids, hidden_state, padding_token_id, mlp, instruction_mask=None):
    """model/modeling_abc.py:172-179 after the backbone: pool -> mlp_head -> .float() -> F.normalize."""
    pooled = get_mean_token_embed(input_ids, hidden_state, padding_token_id, instruction_mask)
    return torch.nn.functional.normalize(mlp(pooled.float()).float(), dim=-1)
