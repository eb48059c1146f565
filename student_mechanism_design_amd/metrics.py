"""Batched reductions over a rollout record, on device.

Same formulas as the reference's offline toolkit (src/eval/metrics.py): `belief_cross_entropy`
(:294-306: clip to [1e-8, 1], renormalise, -log p[true node]) and the balance aggregates of
`MetricsTracker.get_aggregated_metrics` (:162-227: MrX win rate, episode length, time to catch,
survival time) — computed for all finished episodes of a [T, B] record without leaving the GPU.
"""
from typing import Dict

import torch


def belief_cross_entropy(belief: torch.Tensor, true_index: torch.Tensor) -> torch.Tensor:
    """belief [..., N] (padding columns allowed to be zero), true_index int[...] -> CE [...]."""
    b = belief.clamp(1e-8, 1.0)
    b = b / b.sum(-1, keepdim=True)
    return -torch.log(torch.gather(b, -1, true_index.long().unsqueeze(-1)).squeeze(-1))


def rollout_metrics(record: Dict[str, torch.Tensor], num_nodes: int) -> Dict[str, torch.Tensor]:
    """Aggregates over the episodes that END inside the record (winner: 1 Police, 2 MrX).
    Episode length = the env timestep at the finishing step + 1 (= number of steps played)."""
    done = (record["terminated"] | record["truncated"]).bool()
    winner = record["winner"]
    length = (record["t"] + 1).float()
    n = done.sum().clamp_min(1).float()
    mrx = done & (winner == 2)
    pol = done & (winner == 1)
    out = {
        "num_episodes": done.sum(),
        "mrx_wins": mrx.sum(), "police_wins": pol.sum(),
        "win_rate": mrx.sum().float() / n,                                    # MrX win rate, metrics.py:196
        "mean_episode_length": (length * done).sum() / n,
        "mean_time_to_catch": (length * pol).sum() / pol.sum().clamp_min(1).float(),
        "mean_survival_time": (length * mrx).sum() / mrx.sum().clamp_min(1).float(),
        "mean_budget_left": (record["budget"][..., 1:].float().mean(-1) * done).sum() / n,
    }
    if record.get("belief") is not None:
        ce = belief_cross_entropy(record["belief"][..., :num_nodes], record["pos"][..., 0])
        out["mean_belief_ce"] = ce.mean()
        out["belief_ce_std"] = ce.std(unbiased=False)                          # np.std, metrics.py:199
    return out
