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


def rollout_metrics(record: Dict[str, torch.Tensor], num_nodes: int, reveal_interval: int = 0) -> Dict[str, torch.Tensor]:
    """Aggregates over the episodes that END inside the record (winner: 1 Police, 2 MrX).
    Episode length = the env timestep at the finishing step + 1 (= number of steps played).

    Belief quality follows `MetricsTracker.record_step` (eval/metrics.py:138-141): the cross-entropy is taken AT
    REVEAL TIMES only — with `reveal_interval` = k, at the rows whose step ends with a reveal ((t + 1) % k == 0):
    the police's belief just before MrX shows himself, scored against where he stands.  `mean_belief_ce_all_steps`
    is the average over every recorded step (what round 1 reported under the reference's name).
    `mean_budget_before_final_step` is the police budget in the observation of the finishing step (the record
    holds observations BEFORE each step; the last move's debit is not part of it)."""
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
        "mean_budget_before_final_step": (record["budget"][..., 1:].float().mean(-1) * done).sum() / n,
    }
    if record.get("belief") is not None:
        ce = belief_cross_entropy(record["belief"][..., :num_nodes], record["pos"][..., 0])
        out["mean_belief_ce_all_steps"] = ce.mean()
        if reveal_interval and reveal_interval > 0:
            at_reveal = (record["t"] + 1) % int(reveal_interval) == 0
            k = at_reveal.sum().clamp_min(1).float()
            mean = (ce * at_reveal).sum() / k
            out["num_reveals"] = at_reveal.sum()
            out["mean_belief_ce"] = mean                                        # metrics.py:138-141,198
            out["belief_ce_std"] = (((ce - mean) ** 2 * at_reveal).sum() / k).sqrt()   # np.std, metrics.py:199
        else:
            out["mean_belief_ce"] = ce.mean()
            out["belief_ce_std"] = ce.std(unbiased=False)
    return out
