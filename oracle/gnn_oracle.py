"""CPU restatement (numpy, float64) of the reference's GNN Q-model — TEST INFRASTRUCTURE ONLY (only tests/ may import
this; the product path never does).

Follows /root/reference/src/agent/gnn_agent.py:230-257 (GNNModel: AntiSymmetricConv -> relu -> AntiSymmetricConv -> relu
-> Linear(F, 1)) and /root/reference/src/training/utils.py:151-209 (create_graph_data: node features = one-hot agent
columns; edge_index = board.edge_links.T, every stored edge once).  torch_geometric (requirements.txt:16, unpinned) is
NOT importable here, so the layer is restated from its published definition:

  AntiSymmetricConv (Gravina, Bacciu, Gallicchio: "Anti-Symmetric DGN", ICLR 2023; PyG defaults phi = GCNConv(F, F,
  bias=False), num_iters = 1, act = tanh):      x' = x + epsilon * tanh(x (W - W^T - gamma I)^T + GCN(x) + b)
  GCNConv (Kipf & Welling, ICLR 2017; PyG `gcn_norm`, flow source_to_target, add_self_loops=True):
      GCN(x)[v] = sum over edges (u -> v) and the self loop of  x[u] Theta^T / sqrt(deg(u) deg(v)),
      deg(i) = 1 + number of edges pointing INTO i.

Parity status: **unpinned against the library** (absent offline; the reference holds no fixture of this model); this
file is an INDEPENDENT formulation — the propagation matrix is built dense, edge by edge, straight from the edge list —
that the gather-table implementations (torch `policies.AntiSymmetricConvEll`, HIP `sy_gnn_q_act`) are held to.
"""
import numpy as np


def propagation_matrix(num_nodes, edge_links, directed=True):
    """A^ [N][N] float64 with A^[v][u] = 1 / sqrt(deg(u) deg(v)) per edge u -> v (parallel edges add), plus the diagonal."""
    n = int(num_nodes)
    A = np.zeros((n, n), dtype=np.float64)
    for u, v in np.asarray(edge_links, dtype=np.int64).reshape(-1, 2):
        A[v, u] += 1.0                      # message u -> v
        if not directed:
            A[u, v] += 1.0
    A += np.eye(n)                          # add_self_loops
    deg = A.sum(axis=1)                     # incoming edge weights incl. the self loop
    dinv = 1.0 / np.sqrt(deg)
    return dinv[:, None] * A * dinv[None, :]


def antisymmetric_conv(x, a_hat, W, theta, bias, epsilon=0.1, gamma=0.1):
    """x [R][N][F] float64 -> x'."""
    W = np.asarray(W, dtype=np.float64)
    anti = W - W.T - gamma * np.eye(W.shape[0])
    gcn = np.einsum("vu,ruf->rvf", a_hat, x @ np.asarray(theta, dtype=np.float64).T)
    return x + epsilon * np.tanh(x @ anti.T + gcn + np.asarray(bias, dtype=np.float64))


def gnn_q(x, a_hat, params):
    """GNNModel.forward: params = dict(conv1=(W, theta, bias), conv2=(...), out_w [F], out_b, epsilon, gamma)."""
    e, g = params.get("epsilon", 0.1), params.get("gamma", 0.1)
    h = np.maximum(antisymmetric_conv(np.asarray(x, dtype=np.float64), a_hat, *params["conv1"], epsilon=e, gamma=g), 0.0)
    h = np.maximum(antisymmetric_conv(h, a_hat, *params["conv2"], epsilon=e, gamma=g), 0.0)
    return h @ np.asarray(params["out_w"], dtype=np.float64).reshape(-1) + float(params["out_b"])


def node_features(pos, num_nodes, belief=None):
    """training/utils.py:176-200: column a = one-hot node of agent a; optional extra column = belief."""
    pos = np.asarray(pos, dtype=np.int64)
    R, A = pos.shape
    x = np.zeros((R, num_nodes, A + (belief is not None)), dtype=np.float64)
    for a in range(A):
        x[np.arange(R), pos[:, a], a] = 1.0
    if belief is not None:
        x[:, :, A] = np.asarray(belief, dtype=np.float64)[:, :num_nodes]
    return x


def greedy_actions(q_mrx, q_police, mask):
    """GNNAgent.select_action with epsilon = 0 (gnn_agent.py:62-74): valid_actions[np.argmax(q[valid_actions])], None -> -1.
    Returns (actions [R][A], margin [R][A] = best minus second-best valid Q; inf with fewer than two valid nodes)."""
    mask = np.asarray(mask, dtype=bool)
    R, A, N = mask.shape
    act = np.full((R, A), -1, dtype=np.int64)
    margin = np.full((R, A), np.inf)
    for r in range(R):
        for a in range(A):
            valid = np.where(mask[r, a])[0]
            if valid.size == 0:
                continue
            q = (q_mrx if a == 0 else q_police)[r][valid]
            act[r, a] = valid[np.argmax(q)]
            if valid.size > 1:
                s = np.sort(q)
                margin[r, a] = s[-1] - s[-2]
    return act, margin


def params_of(model):
    """A policies.GnnQModel's parameters as the dict `gnn_q` takes (float64 numpy)."""
    f = lambda t: t.detach().cpu().numpy().astype(np.float64)  # noqa: E731
    return {"conv1": (f(model.conv1.W), f(model.conv1.phi.weight), f(model.conv1.bias)),
            "conv2": (f(model.conv2.W), f(model.conv2.phi.weight), f(model.conv2.bias)),
            "out_w": f(model.out.weight), "out_b": float(model.out.bias.detach().cpu()),
            "epsilon": model.conv1.epsilon, "gamma": model.conv1.gamma}
