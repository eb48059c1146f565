"""B = 1 facade with the reference's PettingZoo-style surface (level 1).

`CustomEnvironment` keeps the constructor and method signatures of
`src/environment/yard.py::CustomEnvironment` so per-env code (trainers, tests) can call it unchanged;
every transition runs on the GPU through `BatchedScotlandYardEnv` (one env, one wave).  It exists
for drop-in compatibility and for replaying golden traces through the public surface — throughput
comes from the batched class: a facade `reset()` builds a fresh one-env engine (board upload, ~12 small
allocations) and every `step()` reads the state back to the host (>= 6 blocking copies), i.e. it is
millisecond-scale per call by construction.  `as_tensordict` gives the nested key layout `src/training` reads
through torchrl's wrapper (level 2; torchrl itself is absent offline — parity unpinned).
"""
from types import SimpleNamespace

import numpy as np
import torch

from .env import BatchedScotlandYardEnv, WINNER_NAMES
from .graph import Board, sample_board

MAX_MONEY_LIMIT = 1000  # yard.py:11


class _NullLogger:
    def log(self, *a, **k):
        pass

    def log_scalar(self, *a, **k):
        pass


class CustomEnvironment:
    metadata = {"name": "scotland_yard_env"}
    DEFAULT_ACTION = -1

    def __init__(self, number_of_agents, agent_money, reward_weights, logger=None, epoch=0, graph_nodes=10,
                 graph_edges=None, vis_configs=None, device="cuda", seed=0, reveal_interval=0):
        """Same positional parameters as yard.py:18-28 (`number_of_agents` = number of police)."""
        self.reward_weights = reward_weights
        self.number_of_agents = int(number_of_agents)
        self.logger = logger if logger is not None else _NullLogger()
        self.agent_money = int(agent_money)
        self.vis_config = vis_configs
        self.graph_nodes, self.graph_edges = int(graph_nodes), graph_edges
        self.possible_agents = ["MrX"] + [f"Police{n}" for n in range(self.number_of_agents)]
        self.agents = list(self.possible_agents)
        self.current_winner = None
        self.epoch, self.episode = epoch, 0
        self._device, self._reveal = device, int(reveal_interval)
        self._rng = np.random.default_rng(seed)
        self._seed = int(seed)
        self._resets = 0          # every reset() draws new start nodes, like the reference (yard.py:112-116)
        self._env = None
        # yard.py:65-76: the first sample fixes the achievable edge count
        reference_board = sample_board(self.graph_nodes, self.graph_edges, rng=self._rng)
        self.actual_num_edges = reference_board.num_edges
        self.reset()

    # ------------------------------------------------------------------ reset / step
    def reset(self, episode=0, seed=None, options=None):
        """yard.py:80-142.  `options` may carry {"board": Board, "starts": [...]} to pin the episode
        (used by the golden replays); otherwise a new board is sampled like the reference does."""
        self.episode = episode
        options = options or {}
        board = options.get("board")
        if board is None:
            for attempt in range(100):
                board = sample_board(self.graph_nodes, self.graph_edges, rng=self._rng)
                if board.num_edges == self.actual_num_edges:
                    break
            else:
                raise RuntimeError(f"Failed to generate graph with {self.actual_num_edges} edges after 100 attempts.")
        self.board = board
        if self._env is not None:
            self._env.close()
        self._env = BatchedScotlandYardEnv(1, [board], self.number_of_agents, self.agent_money, self.reward_weights,
                                           auto_reset=False, reveal_interval=self._reveal, device=self._device,
                                           seed=self._seed + int(episode) + 1000003 * self._resets, waves_per_block=1)
        self._resets += 1
        if options.get("starts") is not None:
            self._env.reset_to(np.asarray(options["starts"], dtype=np.int32).reshape(1, -1))
        self.agents = list(self.possible_agents)
        self.current_winner = None
        self._sync()
        return self._get_graph_observations(), {a: {} for a in self.agents}

    def step(self, actions):
        """yard.py:144-269: dict agent -> node id / -1 / None.  Returns the reference's 5-tuple."""
        act = np.full((1, len(self.possible_agents)), -1, dtype=np.int32)
        for i, name in enumerate(self.possible_agents):
            a = actions.get(name, None)
            if a is not None:
                a = int(a.item()) if hasattr(a, "item") else int(a)
                act[0, i] = a if a != -1 else -1
        self._env.step(torch.as_tensor(act, device=self._env.device))
        self._sync()
        rew = self._env.reward[0].cpu().numpy()
        term, trunc = bool(self._env.terminated[0].item()), bool(self._env.truncated[0].item())
        self.current_winner = WINNER_NAMES[int(self._env.winner[0].item())]
        names = self.possible_agents
        rewards = {a: float(rew[i]) for i, a in enumerate(names)}
        terminations = {a: term for a in names}
        truncations = {a: trunc for a in names}
        observations = self._get_graph_observations()
        infos = {a: {} for a in names}
        if term or trunc:
            self.agents = []  # yard.py:260-266
        return observations, rewards, terminations, truncations, infos

    # ------------------------------------------------------------------ state mirrors
    def _sync(self):
        pos = self._env.pos[0].cpu().numpy()
        self.MrX_pos = [int(pos[0])]
        self.police_positions = [int(x) for x in pos[1:]]
        self.agents_money = [int(x) for x in self._env.budget[0].cpu().numpy()]
        self.timestep = int(self._env.t[0].item())
        vis = self._env.visits[0].cpu().numpy()
        self.node_visit_counts = {int(i): int(vis[i]) for i in np.nonzero(vis)[0]}

    def _get_adjacency_matrix(self):
        n = self.board.num_nodes
        adj = np.zeros((n, n))
        adj[self.board.edge_links[:, 0], self.board.edge_links[:, 1]] = 1
        adj[self.board.edge_links[:, 1], self.board.edge_links[:, 0]] = 1
        return adj

    def _get_edge_weight_matrix(self):
        n = self.board.num_nodes
        w = np.full((n, n), np.inf)
        np.fill_diagonal(w, 0)
        for (u, v), c in zip(self.board.edge_links, self.board.edges):
            w[u, v] = w[v, u] = c
        return w

    def _get_graph_observations(self):
        """yard.py:271-335 key layout; masks / positions / budgets come from the device state."""
        n, A = self.board.num_nodes, len(self.possible_agents)
        adjacency = self._get_adjacency_matrix()
        node_features = np.zeros((n, A))
        node_features[self.MrX_pos[0], 0] = 1
        for i, p in enumerate(self.police_positions):
            node_features[p, i + 1] = 1
        masks = self._env.action_mask[0].cpu().numpy()
        belief = None if self._env.belief is None else self._env.belief[0].cpu().numpy()
        obs = {}
        for idx, agent in enumerate(self.possible_agents):
            pos = self.MrX_pos[0] if idx == 0 else self.police_positions[idx - 1]
            obs[agent] = {
                "adjacency_matrix": adjacency, "node_features": node_features,
                "edge_index": self.board.edge_links.T, "edge_features": self.board.edges,
                "MrX_pos": self.MrX_pos[0], "Polices_pos": self.police_positions[:],
                "Currency": self.agents_money[1:], "action_mask": masks[idx].copy(),
                "agent_position": pos, "agent_budget": np.array([self.agents_money[idx]], dtype=np.float32),
            }
            if belief is not None and idx > 0:
                obs[agent]["belief_map"] = belief.copy()
        return obs

    def _get_possible_moves(self, pos, agent_idx):
        """yard.py:420-472: affordable distinct neighbours (ascending) and their cheapest edge weights."""
        row = self._env.pool.ell[0, int(pos)]
        nb, w = (row & 0xFFFF).astype(np.int32), (row >> 16).astype(np.int64)
        keep = (nb < self.board.num_nodes) & (w <= self.agents_money[agent_idx])
        return nb[keep], w[keep]

    def get_possible_moves(self, agent_idx):
        pos = self.MrX_pos[0] if agent_idx == 0 else self.police_positions[agent_idx - 1]
        return self._get_possible_moves(pos, agent_idx)[0]

    def get_distance(self, node1, node2):
        return float(self._env.pool.apsp[0, int(node1), int(node2)])

    def get_mrx_position(self):
        return self.MrX_pos

    def get_police_position(self, police_idx):
        return self.police_positions[police_idx]

    # ------------------------------------------------------------------ spaces (gymnasium optional)
    def action_space(self, agent):
        n = self.board.num_nodes
        try:
            from gymnasium.spaces import Discrete
            return Discrete(n)
        except Exception:
            return SimpleNamespace(n=n, shape=(), dtype=np.int64)

    def observation_space(self, agent):
        """Shapes of yard.py:500-554 as plain data when gymnasium is not installed."""
        n, P = self.board.num_nodes, self.number_of_agents
        return {"adjacency_matrix": (n, n), "node_features": (n, P + 1), "edge_index": (2, self.actual_num_edges),
                "edge_features": (self.actual_num_edges,), "MrX_pos": n, "Polices_pos": [n] * P,
                "Currency": [self.agent_money + 1] * P, "action_mask": (n,), "agent_position": n, "agent_budget": (1,)}

    # rendering is out of scope (disabled in every training config)
    def render(self):
        pass

    def initialize_render(self, reset=False):
        pass

    def close_render(self):
        pass

    def save_visualizations(self):
        pass

    def close(self):
        if self._env is not None:
            self._env.close()
            self._env = None


def as_tensordict(env: CustomEnvironment, observations, rewards=None, terminations=None, truncations=None):
    """Nested dict with the key paths `src/training` reads through torchrl's PettingZooWrapper:
    td[agent]["observation"][key] with a leading group dim of 1, one-hot `MrX_pos` [1,N] and
    `Polices_pos` [1,P,N] (mappo_trainer.py:173,197), td[agent]["reward"|"terminated"|"truncated"]."""
    n = env.board.num_nodes
    td = {}
    for agent, ob in observations.items():
        o = {}
        for k, v in ob.items():
            if k == "MrX_pos":
                t = torch.zeros(1, n)
                t[0, int(v)] = 1
            elif k == "Polices_pos":
                t = torch.zeros(1, len(v), n)
                for i, p in enumerate(v):
                    t[0, i, int(p)] = 1
            else:
                t = torch.as_tensor(np.asarray(v)).unsqueeze(0)
            o[k] = t
        td[agent] = {"observation": o}
        if rewards is not None:
            td[agent]["reward"] = torch.tensor([[rewards[agent]]], dtype=torch.float32)
            td[agent]["terminated"] = torch.tensor([[terminations[agent]]])
            td[agent]["truncated"] = torch.tensor([[truncations[agent]]])
    return td
