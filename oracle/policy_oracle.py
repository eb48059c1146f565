"""CPU restatement (numpy, float64) of the learned policy's action draw inside the fused rollout — TEST INFRASTRUCTURE ONLY
(only tests/ may import this; the product path never does).

What it restates: `MappoAgent.select_action` (/root/reference/src/agent/mappo_agent.py:87-142) as the engine performs it
for every (env, agent) of a recorded rollout, with the engine-defined random stream documented in DESIGN.md section 4:

  * observation (mappo_trainer.py:173,197): MrX's actor sees one-hot(MrX node), a police actor the multi-hot of all police
    nodes; probs = softmax(W2 relu(W1 obs + b1) + b2)                                   (mappo_agent.py:6-29)
  * masked, renormalised: p = probs * mask; if sum(p) <= 1e-8 -> uniform over the mask; else p / (sum + 1e-8);
    Categorical(p) renormalises                                                          (mappo_agent.py:112-134)
  * the draw (engine-defined): Gumbel-max over the affordable ELL entries of the agent's node,
        key_i = logit_i + g(x, column_i),   g = -log(-log(u)),   u = ((h >> 8) + 0.5) / 2^24,
        h = murmur3-finaliser(x ^ column_i * 0x9E3779B9 ^ 0x85EBCA6B),
    x = word (c & 3) of Philox4x32-7 block (global env id, c >> 2, ACT << 8 | agent) keyed with the stream key, c = the env's
    step counter before the step; the action is the entry with the largest key (logits 0 under the uniform fallback).

Because the device evaluates logits and Gumbel noise in float32 (fast log; u clamped below 1), the restatement reports the float64 keys and the
top-2 MARGIN per (env, agent): wherever the margin exceeds the float32 error the recorded action must be the arg-max.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def philox4x32_7(gid, ctr, purpose, idx, key_lo, key_hi):
    """Philox4x32 with 7 rounds (Salmon et al. 2011) on arrays: counter = (gid lo, gid hi, ctr, purpose << 8 | idx)."""
    gid = np.asarray(gid, dtype=np.uint64)
    c0 = gid & M32
    c1 = gid >> np.uint64(32)
    c2 = np.asarray(ctr, dtype=np.uint64) & M32
    c3 = (np.uint64(purpose) << np.uint64(8)) | np.asarray(idx, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint64(key_lo), np.uint64(key_hi)
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    for _ in range(7):
        p0 = m0 * c0
        p1 = m1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & M32
        hi1, lo1 = p1 >> np.uint64(32), p1 & M32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & M32, lo1, (hi0 ^ c3 ^ k1) & M32, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def gumbel_noise(x, column):
    """The engine's per-entry Gumbel noise from the agent's Philox word `x` and the entry's ELL column (float64 logs)."""
    h = (np.asarray(x, dtype=np.uint64) ^ ((np.asarray(column, dtype=np.uint64) * np.uint64(0x9E3779B9)) & M32) ^ np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    u = ((h >> np.uint64(8)).astype(np.float64) + 0.5) / 16777216.0
    return -np.log(-np.log(u))


def actor_logits(W1, b1, W2, b2, pos):
    """All N logits of every actor on the trainer's observations, float64.  W1 [A][H][N], b1 [A][H], W2 [A][N][H],
    b2 [A][N] (torch Linear layouts), pos int [R][A] -> [R][A][N]."""
    W1, b1, W2, b2 = (np.asarray(x, dtype=np.float64) for x in (W1, b1, W2, b2))
    R, A = pos.shape
    out = np.empty((R, A, W2.shape[1]), dtype=np.float64)
    for a in range(A):
        if a == 0:
            z = b1[0][None, :] + W1[0].T[pos[:, 0]]                                   # one-hot MrX node
        else:
            z = b1[a][None, :] + W1[a].T[pos[:, 1:]].sum(axis=1)                      # multi-hot police nodes
        out[:, a] = np.maximum(z, 0.0) @ W2[a].T + b2[a][None, :]
    return out


def policy_draws(pos, budget, step_count0, env_ids, ell_rows, weights, stream_key):
    """For every step s, env b, agent a of a recorded rollout (pos, budget int [T][B][A]: the observation BEFORE each step):

    ell_rows(b) -> uint32 [N][16] ELL table of env b's board (neighbour | weight << 16; padding weight 0xFFFF).
    Returns dict of [T][B][A] arrays: `action` (arg-max entry's node, -1 without a legal entry), `margin` (top-1 minus top-2
    key; +inf with fewer than two legal entries), `log_prob` (of `action`), `legal_mass` (softmax mass of the legal entries
    over all N logits), `fallback` (mass <= 1e-8: uniform over the mask), and `logp_of(node)` support via `keys`/`nodes`
    [T][B][A][16] (key -inf / node -1 on entries that are not legal)."""
    pos = np.asarray(pos, dtype=np.int64)
    budget = np.asarray(budget, dtype=np.int64)
    T, B, A = pos.shape
    key_lo, key_hi = int(stream_key) & 0xFFFFFFFF, (int(stream_key) >> 32) & 0xFFFFFFFF
    ell = np.stack([np.asarray(ell_rows(b), dtype=np.uint32) for b in range(B)])         # [B][N][16]
    N = ell.shape[1]
    out = {k: np.zeros((T, B, A), dtype=dt) for k, dt in (("action", np.int64), ("margin", np.float64), ("log_prob", np.float64),
                                                          ("legal_mass", np.float64), ("fallback", bool), ("count", np.int64))}
    out["keys"] = np.full((T, B, A, 16), -np.inf)
    out["nodes"] = np.full((T, B, A, 16), -1, dtype=np.int64)
    out["logp_entries"] = np.full((T, B, A, 16), -np.inf)
    cols = np.arange(16, dtype=np.uint64)
    bidx = np.arange(B)[:, None, None]
    for s in range(T):
        c = (np.asarray(step_count0, dtype=np.uint64) + np.uint64(s)) & M32               # [B]
        words = philox4x32_7(np.asarray(env_ids, dtype=np.uint64)[:, None], (c >> np.uint64(2))[:, None], 1,
                             np.arange(A, dtype=np.uint64)[None, :], key_lo, key_hi)     # [B][A][4]
        x = np.take_along_axis(words, (c & np.uint64(3)).astype(np.int64)[:, None, None].repeat(A, 1), axis=2)[..., 0]   # [B][A]
        ent = ell[bidx, pos[s][:, :, None], np.arange(16)[None, None, :]]                 # [B][A][16]
        nbr = (ent & np.uint32(0xFFFF)).astype(np.int64)
        wgt = (ent >> np.uint32(16)).astype(np.int64)
        legal = wgt <= budget[s][:, :, None]                                              # padding weight 0xFFFF > any budget
        logits = actor_logits(weights["W1"], weights["b1"], weights["W2"], weights["b2"], pos[s])      # [B][A][N]
        mx = logits.max(-1, keepdims=True)
        lse_all = mx[..., 0] + np.log(np.exp(logits - mx).sum(-1))
        l_ent = np.take_along_axis(logits, np.minimum(nbr, N - 1), axis=2)
        l_ent = np.where(legal, l_ent, -np.inf)
        cnt = legal.sum(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            lm = np.where(cnt > 0, l_ent.max(-1), 0.0)
            lse_legal = lm + np.log(np.exp(np.where(legal, l_ent - lm[..., None], -np.inf)).sum(-1))
            mass = np.where(cnt > 0, np.exp(lse_legal - lse_all), 0.0)
        fb = (mass <= 1e-8) & (cnt > 0)                                                   # mappo_agent.py:123-127
        l_use = np.where(fb[..., None], np.where(legal, 0.0, -np.inf), l_ent)
        with np.errstate(divide="ignore"):
            lse_use = np.where(fb, np.log(np.maximum(cnt, 1)), lse_legal)
        g = gumbel_noise(x[:, :, None], cols[None, None, :])
        keys = np.where(legal, l_use + g, -np.inf)
        order = np.sort(keys, axis=-1)
        best = keys.argmax(-1)
        out["action"][s] = np.where(cnt > 0, np.take_along_axis(nbr, best[..., None], axis=2)[..., 0], -1)
        with np.errstate(invalid="ignore"):
            out["margin"][s] = np.where(cnt > 1, order[..., -1] - order[..., -2], np.inf)
        with np.errstate(invalid="ignore"):
            lp_ent = np.where(legal, l_use - lse_use[..., None], -np.inf)
        out["log_prob"][s] = np.where(cnt > 0, np.take_along_axis(lp_ent, best[..., None], axis=2)[..., 0], 0.0)
        out["legal_mass"][s], out["fallback"][s], out["count"][s] = mass, fb, cnt
        out["keys"][s], out["nodes"][s], out["logp_entries"][s] = keys, np.where(legal, nbr, -1), lp_ent
    return out


def check_recorded_policy_rollout(rec_action, rec_log_prob, draws, margin_tol=1e-4, logp_tol=1e-4, mass_band=(1e-9, 1e-7)):
    """The recorded actions / log-probabilities of a policy rollout against `policy_draws`:
      * wherever the restatement's top-2 key margin exceeds `margin_tol` (and the legal mass is not within `mass_band` of the
        1e-8 underflow threshold, where float32 may decide the fallback differently), the recorded action IS the arg-max;
      * everywhere, the recorded action is a legal entry and its recorded log-probability equals the restatement's
        log-probability OF THAT ACTION to `logp_tol` (again outside the underflow band).
    Returns counts for the caller to assert on coverage."""
    act = np.asarray(rec_action, dtype=np.int64)
    lp = np.asarray(rec_log_prob, dtype=np.float64)
    cnt = draws["count"]
    none = cnt == 0
    assert (act[none] == -1).all() and (lp[none] == 0).all(), "an agent without a legal entry must record action -1, log-prob 0"
    band = (draws["legal_mass"] > mass_band[0]) & (draws["legal_mass"] < mass_band[1]) & ~none
    hit = (draws["nodes"] == act[..., None]) & (draws["nodes"] >= 0)
    assert (hit.sum(-1)[~none] == 1).all(), "a recorded action is not a legal neighbour of the recorded node"
    lp_of_action = np.where(hit, draws["logp_entries"], 0.0).sum(-1)
    ok = ~none & ~band
    err = np.abs(lp - lp_of_action)[ok]
    assert err.size == 0 or err.max() <= logp_tol, "log-prob differs by %g" % err.max()
    decided = ok & (draws["margin"] > margin_tol)
    wrong = decided & (act != draws["action"])
    assert not wrong.any(), "%d of %d decided draws differ from the restatement (first: %s)" % (
        wrong.sum(), decided.sum(), np.argwhere(wrong)[:3].tolist())
    return {"decided": int(decided.sum()), "undecided": int((ok & ~decided).sum()), "in_underflow_band": int(band.sum()),
            "fallbacks": int(draws["fallback"].sum()), "max_logp_err": float(err.max()) if err.size else 0.0,
            "agents": int(act.size)}
