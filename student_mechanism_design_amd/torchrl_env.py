"""Optional torchrl registration of the batched engine (level 2 of the boundary).

The reference wraps its PettingZoo env with `torchrl.envs.libs.pettingzoo.PettingZooWrapper`
(src/training/mappo_trainer.py:116, gnn_trainer.py:129) and then only uses `reset`, `step`, `step_mdp` and attribute
forwarding.  `make_torchrl_env` exposes `BatchedScotlandYardEnv` as a torchrl `EnvBase` with batch size [B] and the
same per-agent key layout (`td[agent]["observation"][key]`, `td[agent]["action"]`, `td["next"][agent]["reward" |
"terminated" | "truncated"]`), so that torchrl collectors / `step_mdp` work on it unchanged.

torchrl and tensordict are NOT importable in the build container nor on the GPU box, so this module cannot be tested
here: **parity unpinned** (DESIGN.md section 8).  Importing this module never fails; `make_torchrl_env` raises
ImportError with instructions when torchrl is absent.  Everything heavy stays in `env.py`; this file only translates
keys and specs.
"""
from typing import Optional

import torch


def torchrl_available() -> bool:
    try:
        import tensordict  # noqa: F401
        import torchrl  # noqa: F401
        return True
    except Exception:
        return False


def make_torchrl_env(env, hide_mrx_between_reveals: bool = False):
    """Wrap a `BatchedScotlandYardEnv` as a torchrl `EnvBase` (batch_size = [B], device = the engine's GPU)."""
    try:
        from tensordict import TensorDict
        from torchrl.envs import EnvBase
        import torchrl.data as trd
    except Exception as exc:  # pragma: no cover - torchrl is optional and absent offline
        raise ImportError("make_torchrl_env needs `torchrl` and `tensordict` (pip install torchrl); the engine itself "
                          "(student_mechanism_design_amd.BatchedScotlandYardEnv) does not") from exc

    # spec class names changed across torchrl releases: resolve both spellings
    Composite = getattr(trd, "Composite", None) or getattr(trd, "CompositeSpec")
    Unbounded = getattr(trd, "Unbounded", None) or getattr(trd, "UnboundedContinuousTensorSpec")
    Categorical = getattr(trd, "Categorical", None) or getattr(trd, "DiscreteTensorSpec")
    Binary = getattr(trd, "Binary", None) or getattr(trd, "BinaryDiscreteTensorSpec")

    B, A, N, P, dev = env.B, env.A, env.N, env.P, env.device
    agents = list(env.possible_agents)

    class ScotlandYardTorchRL(EnvBase):  # pragma: no cover - exercised only where torchrl exists
        batch_locked = True

        def __init__(self):
            super().__init__(device=dev, batch_size=torch.Size([B]))
            obs, act, rew, done = {}, {}, {}, {}
            for name in agents:
                obs[name] = Composite(observation=Composite(
                    MrX_pos=Categorical(N, shape=(B,), dtype=torch.int64, device=dev),
                    Polices_pos=Categorical(N, shape=(B, P), dtype=torch.int64, device=dev),
                    Currency=Unbounded(shape=(B, P), dtype=torch.int64, device=dev),
                    action_mask=Binary(N, shape=(B, N), dtype=torch.bool, device=dev),
                    agent_position=Categorical(N, shape=(B,), dtype=torch.int64, device=dev),
                    agent_budget=Unbounded(shape=(B, 1), dtype=torch.float32, device=dev),
                    belief_map=Unbounded(shape=(B, N), dtype=torch.float32, device=dev),
                    shape=(B,)), shape=(B,))
                act[name] = Composite(action=Categorical(N, shape=(B,), dtype=torch.int64, device=dev), shape=(B,))
                rew[name] = Composite(reward=Unbounded(shape=(B, 1), dtype=torch.float32, device=dev), shape=(B,))
                done[name] = Composite(terminated=Binary(1, shape=(B, 1), dtype=torch.bool, device=dev),
                                       truncated=Binary(1, shape=(B, 1), dtype=torch.bool, device=dev),
                                       done=Binary(1, shape=(B, 1), dtype=torch.bool, device=dev), shape=(B,))
            self.observation_spec = Composite(obs, shape=(B,))
            self.action_spec = Composite(act, shape=(B,))
            self.reward_spec = Composite(rew, shape=(B,))
            self.done_spec = Composite(done, shape=(B,))
            self.engine = env

        # attribute forwarding the reference relies on (gnn_trainer.py:133-135,207; utils.py:166-174)
        def __getattr__(self, name):
            try:
                return super().__getattr__(name)
            except AttributeError:
                return getattr(self.__dict__["engine"], name)

        def _obs(self):
            e = self.engine
            out = {}
            for i, name in enumerate(agents):
                o = {"MrX_pos": e.pos[:, 0].long(), "Polices_pos": e.pos[:, 1:].long(), "Currency": e.budget[:, 1:].long(),
                     "action_mask": e.action_mask[:, i], "agent_position": e.pos[:, i].long(),
                     "agent_budget": e.budget[:, i:i + 1].float(),
                     "belief_map": e.belief if e.belief is not None else torch.zeros((B, N), device=dev)}
                if hide_mrx_between_reveals and i > 0:
                    o["MrX_pos"] = torch.full_like(o["MrX_pos"], -1)
                out[name] = {"observation": o}
            return out

        def _reset(self, tensordict: Optional["TensorDict"] = None, **kwargs):
            mask = None
            if tensordict is not None and "_reset" in tensordict.keys():
                mask = tensordict["_reset"].reshape(B)
            self.engine.reset(seed=kwargs.get("seed"), env_mask=mask) if mask is not None else self.engine.reset(seed=kwargs.get("seed"))
            td = TensorDict(self._obs(), batch_size=[B], device=dev)
            for name in agents:
                for k in ("terminated", "truncated", "done"):
                    td[name, k] = torch.zeros((B, 1), dtype=torch.bool, device=dev)
            return td

        def _step(self, tensordict: "TensorDict"):
            actions = torch.stack([tensordict[name, "action"].reshape(B) for name in agents], dim=1).to(torch.int32)
            _, reward, term, trunc = self.engine.step(actions)
            td = TensorDict(self._obs(), batch_size=[B], device=dev)
            for i, name in enumerate(agents):
                td[name, "reward"] = reward[:, i:i + 1].float()
                td[name, "terminated"] = term.reshape(B, 1).clone()
                td[name, "truncated"] = trunc.reshape(B, 1).clone()
                td[name, "done"] = (term | trunc).reshape(B, 1)
            return td

        def _set_seed(self, seed: Optional[int]):
            if seed is not None:
                self.engine.reset(seed=int(seed))
            return seed

    return ScotlandYardTorchRL()
