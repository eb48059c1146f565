/*
 * sy_env.h — C ABI of the MI355X-native batched Scotland-Yard environment engine.
 *
 * This is the drop-in boundary for the reference's env.step hot path.  The reference
 * (elte-collective-intelligence/student-mechanism-design) has no FFI of its own: its boundary is
 * the Python class CustomEnvironment (src/environment/yard.py).  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference/src); INTEGRATION.md shows the
 * ctypes binding a maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types in signatures.
 *   - every pointer documented "device" is device memory owned by the CALLER (e.g. torch tensors);
 *     it must stay alive until the work enqueued on `stream` has finished.
 *   - every call is asynchronous on the hipStream_t passed as `void *stream` (NULL = null stream);
 *     nothing here synchronises, allocates device memory, or copies (graph-capture safe), except
 *     sy_env_create / sy_env_destroy (host allocation only).
 *   - return value: 0 = ok, <0 = error (SY_ERR_*); sy_last_error() gives a thread-local message.
 *     Nothing throws across the ABI.  One host thread per handle.
 *   - invalid / unaffordable / blocked / -1 actions never fail: the agent stays (yard.py:168-229).
 *
 * Layouts (B envs, A = P+1 agents, N nodes, NS = node stride (multiple of 16, >= N), D = 16)
 *   pos, budget      int32  [B][A]      agent 0 = MrX, agent k+1 = Police k      (yard.py:125-126,117-119)
 *   t                int32  [B]         env timestep                              (yard.py:127,355)
 *   step_count       uint32 [B]         steps since the last full reset (RNG counter)
 *   visits           uint16 [B][NS]     police node_visit_counts                  (yard.py:244-245)
 *   belief           float  [B][NS]     police belief over MrX's node             (belief_module.py:69-111)
 *   mask             uint8  [B][A][NS]  action mask, 1 = legal                    (yard.py:297-327)
 *   reward           double [B][A]      (reward_calculator.py:26-266; float64 like the reference)
 *   terminated/truncated uint8 [B], winner int8 [B] (0 none, 1 Police, 2 MrX)     (reward_calculator.py:63-90)
 *   graph pool: ell uint32 [G][N][16]  = neighbour id | (edge weight << 16), rows sorted by neighbour
 *               id, padding entries = N | 0xFFFF0000;  apsp uint16 [G][N][N] weighted shortest paths
 *               (replaces pathfinding.py:34-137);  inv_deg float [G][NS] = 1/deg (0 if isolated);
 *               env_graph int32 [B]: graph of each env — all envs of one launch block (envs
 *               [k*waves_per_block, (k+1)*waves_per_block)) use the graph of the block's first env.
 */
#ifndef SY_ENV_H
#define SY_ENV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SY_ABI_VERSION 7
#define SY_ELL_WIDTH 16
#define SY_MAX_AGENTS 8
#define SY_MAX_NODES 1024
#define SY_NUM_WEIGHTS 11
#define SY_MRX_MONEY 1000 /* MAX_MONEY_LIMIT yard.py:11 */

#define SY_OK 0
#define SY_ERR_INVALID (-1)  /* bad argument / config */
#define SY_ERR_STATE (-2)    /* call order (e.g. step before bind) */
#define SY_ERR_HIP (-3)      /* a HIP runtime call failed */

/* bits of the engine's device status word (sy_env_bind_status / sy_env_status): a launch that hit one of these
 * still drains (every in-kernel spin is bounded), but its results must not be trusted */
#define SY_STATUS_BELIEF_WAIT_EXPIRED 1u /* a belief wave gave up waiting for its move wave's ring entry  */
#define SY_STATUS_RING_WAIT_EXPIRED 2u   /* a move wave gave up waiting for ring space (belief wave lost) */

typedef struct sy_env sy_env; /* opaque */

typedef struct sy_env_config {
    int32_t num_envs;          /* B, envs on THIS GPU                                         */
    int32_t num_nodes;         /* N  (graph_nodes, yard.py:25,52)                             */
    int32_t num_police;        /* P  (`number_of_agents`, yard.py:20,33)                      */
    int32_t agent_money;       /* police budget at reset, <= 65534 (yard.py:21,117-119)       */
    int32_t max_timestep;      /* truncation when t > max_timestep; reference: 250            */
    int32_t num_graphs;        /* G, graphs in the pool                                       */
    int32_t node_stride;       /* NS, multiple of 16, >= N                                    */
    int32_t reveal_interval;   /* MrX revealed to the belief when t_post % k == 0; 0 = never  */
    int32_t police_evidence;   /* 1: belief is zeroed on police-occupied nodes                */
    int32_t belief_init_onehot;/* 1: belief starts as a delta on MrX's start (else uniform)   */
    int32_t auto_reset;        /* 1: a finished episode restarts inside step/rollout          */
    int32_t waves_per_block;   /* envs per launch block (1..16, odd <= 7); 0 = engine default  */
    uint64_t env_id_offset;    /* global index of env 0 (rank * B): distinct RNG sub-streams  */
} sy_env_config;

/* device buffers of the live env state (all device, caller-owned; belief may be NULL = no belief) */
typedef struct sy_env_state {
    int32_t *pos;
    int32_t *budget;
    int32_t *t;
    uint32_t *step_count;
    uint16_t *visits;
    float *belief;
    uint8_t *mask;
    double *reward;
    uint8_t *terminated;
    uint8_t *truncated;
    int8_t *winner;
} sy_env_state;

/* Rollout record (all device).  `record` is one packed row of RW = sy_record_words(A) int32 words per
 * env and step, [T][B][RW], written as a single coalesced store:
 *   words [0, 2A)   reward, A float64 values          (this step's outcome)
 *   words [2A, 3A)  pos     (observation BEFORE the step)
 *   words [3A, 4A)  budget  (before the step)
 *   words [4A, 5A)  action taken (-1 = none)
 *   word  5A        t (before the step);  5A+1 terminated;  5A+2 truncated;  5A+3 winner
 * `mask` [T][B][A][NS] and `belief` [T][B][NS] hold the observation before the step; either may be NULL. */
typedef struct sy_rollout_buffers {
    int32_t *record;
    uint8_t *mask;
    float *belief;
    float *log_prob;   /* [T][B][A] log-probability of the recorded action under the in-kernel learned policy
                          (sy_env_set_policy); NULL or ignored for the uniform-random policy */
} sy_rollout_buffers;

/* The MAPPO networks' parameters, all device float32, laid out for the kernel: first actor layers transposed
 * w1t [A][N][H] (torch Linear weight [H][N] transposed), b1 [A][H]; second actor layers transposed w2t [A][H][N],
 * b2 [A][N]; critic first layer transposed c1t [N*A][H], cb1 [H], second layer c2 [H], cb2 [1] (critic may be NULL
 * when no value is requested). */
typedef struct sy_mappo_weights {
    const float *w1t, *b1, *w2t, *b2, *c1t, *cb1, *c2, *cb2;
    const float *w2;   /* [A][N][H] second actor layers in torch's own layout (a node's row contiguous): needed by
                          sy_env_set_policy only, may be NULL for sy_mappo_policy_act */
    const float *logit_bound; /* [A] an upper bound of every logit actor a can produce, over all observations:
                          max_n (b2[n] + sum_k max(w2[n][k], 0) * hmax[k]), hmax[k] = max(b1[k], 0) + hot * max_n max(w1[k][n], 0)
                          (hot = 1 for MrX's one-hot input, P for the police actors' multi-hot input).  Used by
                          sy_env_set_policy's underflow rule; NULL = the rule is not applied */
} sy_mappo_weights;

/* dwords per record row for A = num_police + 1 agents: 5A+4 rounded up to a multiple of 4 */
int sy_record_words(int32_t num_agents);

int sy_abi_version(void);
/* digest of the sources this library was built from (student_mechanism_design_amd/build.py::source_digest): stored
 * profiles carry it, so a benchmark can tell whether a counter file belongs to the library it runs */
const char *sy_build_id(void);
const char *sy_last_error(void);

/* replaces CustomEnvironment.__init__ (yard.py:18-78) for a batch of B envs */
int sy_env_create(const sy_env_config *cfg, sy_env **out);
int sy_env_destroy(sy_env *env);
/* LDS bytes and block count one engine launch uses (for DESIGN/bench reporting) */
int sy_env_launch_info(const sy_env *env, int32_t *waves_per_block, int32_t *blocks, int32_t *lds_bytes);

/* the kernel instance sy_env_rollout will launch for this handle as configured NOW (graph pool, policy, state
 * bound), e.g. "sy::rollout3_kernel<4,true,4,false,2>": the name rocprofv3 reports, so a benchmark can name the
 * kernel it measured without mirroring the launcher's rules.  `record` != 0: with a trajectory record. */
int sy_env_rollout_kernel_name(const sy_env *env, int32_t record, char *buf, int32_t buf_len);

/* board + shortest-path tables (replaces board.edge_links/edges yard.py:91-93 and Pathfinder.set_board
 * pathfinding.py:25-32); all device pointers.  max_degree is the widest ELL row of the pool: it sizes the neighbour
 * scan of the fused rollout (the half-wave scan needs rows of at most 12 neighbours at 4 police, 15 / 16 at 5 / 6 on
 * boards of 129..256 nodes; wider pools, and 0 = unknown, take the general paired scan — same results, slower) */
int sy_env_set_graph_pool(sy_env *env, const uint32_t *ell, const uint16_t *apsp, const float *inv_deg,
                          const int32_t *env_graph, int32_t max_degree /* widest ELL row of the pool; 0 = unknown */);
/* optional, boards of up to 256 nodes: the LDS layout of the fused rollout's belief filter (ParticleBeliefTracker.update's
 * diffusion step, belief_module.py:69-111) — gather_offsets uint16 [G][N][16]: for every node the scratch BYTE offsets
 * (entry * 8; padding = N * 8, the zero entry) of its neighbours in the order they are visited, a permutation of the node's
 * ELL row; node_slot uint16 [G][node_stride]: the entry a node's own value goes to (< node_stride + 16, never N).  Chosen
 * per board (student_mechanism_design_amd/graph.py::belief_layout) so that the 32 addresses a half-wave gathers per
 * instruction fall on 32 different LDS bank pairs; NULL, NULL = entry u for node u, ELL order.  Results are the same filter
 * (the sum over a node's neighbours in another order: float32 rounding). */
int sy_env_set_belief_layout(sy_env *env, const uint16_t *gather_offsets, const uint16_t *node_slot);
/* the 11 reward weights (host array, order = REWARD_WEIGHT_NAMES reward_net.py:5-17) and the device
 * tables exp_tab[d] = exp(-d), cov_tab[v] = exp(-log1p(v)) (reward_calculator.py:184-207) */
int sy_env_set_rewards(sy_env *env, const double *weights_host, const double *exp_tab, int32_t n_exp,
                       const double *cov_tab, int32_t n_cov);
int sy_env_bind_state(sy_env *env, const sy_env_state *state);

/* Failure reporting.  `status` is ONE device uint32 owned by the caller (zero it before binding); kernels OR
 * SY_STATUS_* bits into it when an internal wait ran out instead of carrying on silently.  sy_env_status copies
 * the word to the host and clears nothing; it is the one call here that synchronises (`stream`).  It returns
 * SY_ERR_HIP with a message when the word is non-zero, SY_OK when it is zero (or no word is bound). */
int sy_env_bind_status(sy_env *env, uint32_t *status);
int sy_env_status(sy_env *env, void *stream, uint32_t *status_host /* may be NULL */);

/* replaces CustomEnvironment.reset (yard.py:80-142): distinct uniform start nodes from the engine's
 * Philox stream, budgets [1000, money...], t = 0, visit counts cleared, belief re-initialised, masks.
 * env_sel: device uint8[B] (NULL = all envs, which also zeroes step_count). */
int sy_env_reset(sy_env *env, const uint8_t *env_sel, uint64_t seed, void *stream);
/* same with caller-given start nodes, device int32[B][A] (golden replays) */
int sy_env_reset_to(sy_env *env, const int32_t *starts, void *stream);

/* replaces CustomEnvironment.step (yard.py:144-269): actions device int32[B][A], node id or -1 */
int sy_env_step(sy_env *env, const int32_t *actions, void *stream);

/* sy_env_step that also fills one row of a rollout record (the trainer's per-step stores,
 * mappo_trainer.py:236-262): `row` holds the addresses of row s of the three buffers of
 * sy_rollout_buffers — the observation before the step (mask [B][A][NS], belief [B][NS]) and the packed
 * {reward, pos, budget, action, t, flags} row [B][RW]; null members are skipped */
int sy_env_step_record(sy_env *env, const int32_t *actions, const sy_rollout_buffers *row, void *stream);

/* replaces the rollout loop (mappo_trainer.py:161-287) with the uniform-random policy
 * (random_agent.py): T fused steps in ONE launch, records into `out` (may be NULL) */
int sy_env_rollout(sy_env *env, int32_t T, const sy_rollout_buffers *out, void *stream);

/* The rollout loop with the reference's own policy in it (mappo_trainer.py:161-287 with MappoAgent.select_action):
 * after this call sy_env_rollout samples every action from the MAPPO actors instead of uniformly — inside the
 * fused kernel, per (env, agent): hidden = relu(b1 + row lookups of w1t), a logit per affordable neighbour
 * (w2 row . hidden + b2), action ~ softmax over the affordable neighbours (= the reference's masked, renormalised
 * softmax), log-probability into sy_rollout_buffers.log_prob.  The reference's underflow rule (mappo_agent.py:123-134:
 * legal actions holding <= 1e-8 of the softmax mass -> uniform over the mask) is applied exactly when `logit_bound` is
 * given: a cheap bound rules it out on almost every step, otherwise the actor's N logits are evaluated.  An agent
 * without a legal action gets action -1 and log-probability 0 (the reference would draw an illegal node uniformly,
 * which its env ignores: the agent stays either way).
 * w = NULL restores the uniform-random policy.  hidden: a multiple of 4, at most 128 on boards of up to 256 nodes
 * (the pipeline kernel), at most 64 otherwise.  The weights must stay valid and unchanged while launches are in flight. */
int sy_env_set_policy(sy_env *env, const sy_mappo_weights *w, int32_t hidden);

/* replaces compute_action_mask (action_mask.py:30-84), batched over Q queries on dense float64
 * matrices (device; edge_weights / tolls may be NULL): mask uint8[Q][N] */
int sy_action_mask_dense(const double *adjacency, const double *edge_weights, const double *tolls,
                         int32_t num_nodes, const int32_t *current_node, const double *budget,
                         int32_t num_queries, uint8_t *mask, void *stream);

/* replaces ParticleBeliefTracker.update (belief_module.py:69-111) with its deterministic forward
 * filter, batched: belief float[Q][NS] in/out; hint int32[Q][H] (-1 padded, NULL = none);
 * reveal int32[Q] (-1 = none, NULL = none); ell/inv_deg of ONE graph */
int sy_belief_update(const uint32_t *ell, const float *inv_deg, int32_t num_nodes, int32_t node_stride,
                     float *belief, const int32_t *hint, int32_t hint_width, const int32_t *reveal,
                     int32_t num_queries, void *stream);

/* replaces the masked sampling of MappoAgent.select_action (agent/mappo_agent.py:112-142), batched over
 * num_rows = B * A (env, agent) rows, all device: probs float [rows][probs_row_stride] (the actor's softmax),
 * mask uint8 [rows][mask_row_stride] (the engine's action masks: stride NS).  p = probs * mask; sum <= 1e-8 ->
 * uniform over the mask (over all nodes if the mask is empty), else p / (sum + 1e-8); Categorical renormalises.
 * Outputs: action int32 [rows] (-1 for an empty mask when default_on_empty != 0 = yard.py's DEFAULT_ACTION),
 * log_prob float [rows] = log of the renormalised probability of the action, norm_probs float [rows][num_nodes]
 * (NULL = not wanted).  Draws come from the engine's Philox stream (seed; row, offset + *offset_dev): pass a fresh
 * offset per call, or keep a device-resident counter in offset_dev (NULL = none) so that a captured HIP graph
 * draws fresh numbers on every replay. */
int sy_masked_categorical_sample(const float *probs, int64_t probs_row_stride, const uint8_t *mask, int64_t mask_row_stride,
                                 int32_t num_rows, int32_t num_nodes, uint64_t seed, uint64_t offset,
                                 const uint64_t *offset_dev, int32_t default_on_empty, int32_t *action, float *log_prob,
                                 float *norm_probs, void *stream);

/* replaces MappoAgent.select_action for every (env, agent) in one launch (agent/mappo_agent.py:6-44,87-142):
 * actor MLPs on the trainer's observations (one-hot MrX node / multi-hot police nodes, mappo_trainer.py:173,197),
 * masked sampling as in sy_masked_categorical_sample (an empty mask gives action -1), and the central critic on
 * [mrx] + [police] * P.  pos int32 [B][A], mask uint8 [B][A][mask_row_stride]; outputs action int32 [B][A],
 * log_prob float [B][A], value float [B] (NULL = skip the critic), probs float [B][A][N] (NULL = not wanted: the
 * actor's softmax before masking).  hidden <= 128 (the reference's default, src/configs/agent/default.yaml:2). */
int sy_mappo_policy_act(const int32_t *pos, const uint8_t *mask, int64_t mask_row_stride, const sy_mappo_weights *w,
                        int32_t num_envs, int32_t num_police, int32_t num_nodes, int32_t hidden, uint64_t seed, uint64_t offset,
                        const uint64_t *offset_dev, int32_t *action, float *log_prob, float *value, float *probs, void *stream);

/* replaces GNNModel.forward + GNNAgent.select_action (agent/gnn_agent.py:230-257, :45-82) for every env and agent in one
 * launch: node features as training/utils.py:176-200 builds them (column a = one-hot node of agent a; optionally one
 * more column = the police belief over MrX's node, `belief` float [B][belief_row_stride]), two AntiSymmetricConv layers
 * (x' = x + eps tanh((W - W^T - gamma I) x + A^ Theta x + b), relu after each) and the Linear head -> one Q value per
 * node, from MrX's model for MrX and from the police model for every police agent (gnn_trainer.py:148-178); the
 * action is the masked arg-max (np.argmax: the first maximum in node order), with probability explore_eps a uniform
 * pick among the valid nodes instead (engine Philox stream: seed; row, offset + *offset_dev), -1 when no node is valid.
 * Propagation table per board (graph.py::gcn_tables, packed): gcn_table uint32 [G][N][K][2] = for every target node its K
 * = table_width (the pool's widest row, <= 16) entries {source node, float bits of 1 / sqrt(deg(src) deg(dst))}, rows
 * filled left to right, padding {0, 0.0f}; self_coef float [G][N]; env_graph int32 [B] (NULL = board 0).
 * models = sy_gnn_param_floats(F) pairs of floats, BOTH models interleaved ([p][0] MrX's, [p][1] the police's), FP =
 * sy_gnn_padded_features(F): per conv layer {Wa [FP][FP] = W - W^T - gamma I, Theta [FP][FP] (out, in), b [FP]} x 2,
 * w_out [FP], b_out, epsilon (zero padding).
 * Outputs: action int32 [B][A]; q_values float [B][2][N] (NULL = not wanted).  Boards of up to 256 nodes.
 * torch_geometric is not importable offline: pinned to a float64 restatement of the published layer, not to the library. */
int sy_gnn_padded_features(int32_t num_features);
int sy_gnn_param_floats(int32_t num_features);
int sy_gnn_q_act(const int32_t *pos, const float *belief, int64_t belief_row_stride, const uint8_t *mask, int64_t mask_row_stride,
                 const uint32_t *gcn_table, int32_t table_width, const float *self_coef, const int32_t *env_graph,
                 const float *models, int32_t num_envs, int32_t num_police, int32_t num_nodes, int32_t num_features,
                 float explore_eps, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, int32_t *action, float *q_values,
                 void *stream);

/* replaces the return / advantage lines of MappoAgent.ppo_update (agent/mappo_agent.py:247-258) for a whole
 * [T][B][A] rollout in ONE launch (the reference loops over a flat Python buffer), plus the GAE(gamma, lambda)
 * generalisation (the reference has no GAE; at lambda = 1 with a zero bootstrap GAE's returns equal mode 0's).
 * All pointers device.  Element (t, b, a) of `reward` is reward[t * reward_stride_t + b * reward_stride_b + a]
 * (strides in elements: the packed rollout record is addressed in place); `done_a` / `done_b` (done_b may be
 * NULL) are [T][B] flags of `done_bytes` (1 or 4) bytes each, done = either non-zero (the record's terminated /
 * truncated words; trainer rule training/utils.py:241-251); `value` (NULL = 0) is float with three strides
 * (value_stride_a = 0 broadcasts a central critic's [T][B]); `last_value` (GAE bootstrap, NULL = 0) float with
 * two strides.  Outputs `returns`, `adv` (NULL = skip): contiguous [T][B][A], float (compute_f64 = 0: the
 * reference's float32 arithmetic, bit for bit, rewards rounded to float first) or double (compute_f64 = 1).
 *   mode 0: R_t = r_t + (gamma * R_{t+1}) * (1 - d_t);  adv = R - V         (mappo_agent.py:248-256)
 *   mode 1: delta_t = (r_t + (gamma * V_{t+1}) * (1 - d_t)) - V_t;  A_t = delta_t + ((gamma * lambda) * (1 - d_t)) * A_{t+1};
 *           returns = A + V
 * The standardisation of :257-258 (global mean / std) stays a two-reduction torch expression. */
typedef struct sy_returns_args {
    int32_t T, B, A;
    int32_t mode;             /* 0 reference returns, 1 GAE */
    const void *reward;
    int32_t reward_f64;       /* reward element type: 1 double, 0 float */
    int64_t reward_stride_t, reward_stride_b;
    const void *done_a, *done_b;
    int32_t done_bytes;
    int64_t done_stride_t, done_stride_b;
    const float *value;
    int64_t value_stride_t, value_stride_b, value_stride_a;
    const float *last_value;
    int64_t last_value_stride_b, last_value_stride_a;
    double gamma, lambda;
    int32_t compute_f64;
    void *returns, *adv;
} sy_returns_args;
int sy_returns_advantages(const sy_returns_args *args, void *stream);

/* replaces the loss + backward of MappoAgent.ppo_update (agent/mappo_agent.py:260-293) over a rollout record, minibatch
 * by minibatch: the clipped surrogate (epsilon `clip`) of every agent's recorded action under its actor — the masked,
 * renormalised softmax of select_action (mappo_agent.py:112-134) over the affordable entries of the agent's ELL row —
 * averaged over num_rows * A, plus value_coef * the central critic's MSE against the team return, and the gradient of
 * that sum with respect to every parameter.  All pointers device.  Two entry points:
 *
 * sy_ppo_pack — once per update: the (shuffled) rows of the record into a compact IMAGE the gradient launches stream
 * (sy_ppo_image_bytes(A, num_rows) bytes: 16 B of agent nodes + 16 B per agent + 8 B per row).
 *   record   [R][record_words] packed rollout rows (sy_rollout_buffers.record): pos, budget, action are read in place
 *   log_prob [R][A] recorded log-probabilities, adv [R][A] (standardised) advantages, team_ret [R] critic targets
 *   rows     [num_rows] int32 record rows in image order (a permutation of the record, or part of one), or NULL =
 *            rows row0 .. row0 + num_rows - 1; record row r belongs to env r % num_envs (records are [T][B]);
 *            env_graph [num_envs] the board of every env
 *
 * sy_mappo_ppo_grad — once per minibatch = image rows row0 .. row0 + num_rows - 1 (row0_dev != NULL: row0 is read from
 * that device word instead, so that one captured graph serves every minibatch), one launch + a reduction launch.
 * Parameters and gradients share ONE layout, a slab of S = sy_ppo_slab_floats(N, H) floats per network, [A + 1][S]:
 *   actor a < A:  W1t[a] [N][H] (first layer transposed) | W2[a] [N][H] (torch's layout) | b1[a] [H] | b2[a] [N], padded to
 *                 DN = max(N, H) rounded up to 4 | 8 floats (grads: [0] = actor a's share of the actor loss)
 *   critic (A):   C1m [N][H] = first layer's MrX block transposed | C1p [N][H] = the SUM of its P police blocks transposed
 *                 (the critic's input repeats the police multi-hot P times, mappo_trainer.py:197-208; grads: the gradient
 *                 of EACH block) | cb1 [H] | c2 [H], padded to DN | 8 floats: [1] = cb2 (grads: [0] = the critic loss, MSE
 *                 without value_coef, [1] = d cb2)
 *   ell uint32 [G][N][16];  scratch: sy_ppo_scratch_floats(A, N, H) floats.
 * Optimiser (optional: adam_m / adam_v [A + 1][S] zero-initialised, adam_step one int32 starting at 0, all device): the
 * reduction launch also takes torch.optim.Adam's step (no weight decay) on `params` in place — the parameters then stay
 * resident in the layout the kernels read, and a minibatch is two launches (+ a one-thread tick of the step counter).
 * The police table C1p moves P steps (each of its P blocks takes one).
 * Limits: hidden a multiple of 4, at most 128; nodes < 65 536.  Sums are accumulated in float64 (LDS), in a different
 * order than a BLAS matmul: parity with the torch form is to float32 rounding. */
typedef struct sy_ppo_pack_args {
    const int32_t *record;
    int32_t record_words;
    const float *log_prob, *adv, *team_ret;
    const int32_t *rows;
    int32_t row0;
    int64_t num_rows;
    int32_t num_envs;
    const int32_t *env_graph;
    int32_t num_police;
    void *image;
    int64_t image_bytes;
    int64_t shuffle_domain;   /* rows == NULL and > 0: image row i <- record row row0 + pi(i), pi a pseudo-random permutation of */
    uint64_t shuffle_seed;    /* [0, shuffle_domain) keyed by the seed (a 4-round Feistel network, cycle-walked: no sort)       */
    int64_t chunk_rows;       /* > 0: `record` and `log_prob` are CHUNKS of chunk_rows rows each (the arenas of several ranks after */
    int64_t record_chunk_stride, log_prob_chunk_stride;   /* the one all-gather of an update), chunk c at c * stride elements;      */
                              /* adv / team_ret stay plain [R] arrays; 0: one contiguous array each                                */
} sy_ppo_pack_args;
int64_t sy_ppo_image_bytes(int32_t num_agents, int64_t num_rows);
int sy_ppo_pack(const sy_ppo_pack_args *args, void *stream);

typedef struct sy_ppo_args {
    const void *image;
    int64_t image_rows;       /* rows the image was packed with */
    int32_t row0;
    const int32_t *row0_dev;
    int32_t num_rows;
    const uint32_t *ell;
    int32_t num_police, num_nodes, hidden;
    float *params;
    float clip, value_coef;
    float *scratch;
    int64_t scratch_floats;
    float *grads;
    float *adam_m, *adam_v;
    int32_t *adam_step;
    float lr, beta1, beta2, eps;
} sy_ppo_args;
int32_t sy_ppo_slab_floats(int32_t num_nodes, int32_t hidden);
int64_t sy_ppo_scratch_floats(int32_t num_agents, int32_t num_nodes, int32_t hidden);
int sy_mappo_ppo_grad(const sy_ppo_args *args, void *stream);
/* the Adam step of sy_mappo_ppo_grad on its own, for data-parallel training: call sy_mappo_ppo_grad WITHOUT an optimiser
 * state, all-reduce (average) `grads` [A + 1][S] across the ranks, then take the step here (the optim.Adam lines of
 * agent/mappo_agent.py:80-83,265,293); same slab layout, same rule, adam_step advanced by one */
int sy_ppo_adam_step(float *params, const float *grads, float *adam_m, float *adam_v, int32_t *adam_step, int32_t num_police,
                     int32_t num_nodes, int32_t hidden, float lr, float beta1, float beta2, float eps, void *stream);

/* replaces Pathfinder.get_distance (pathfinding.py:34-137) for a whole pool: all-pairs weighted
 * shortest paths from the ELL table, apsp uint16 [G][N][N] (0xFFFF = unreachable); all device */
int sy_build_apsp(const uint32_t *ell, int32_t num_nodes, int32_t num_graphs, uint16_t *apsp, void *stream);

/* replaces ConnectedGraph.sample / _create_tree (graph_layout.py:9-80) for a pool of boards, on device:
 * random-Prim tree + extra edges under the degree cap (reference: 4), weights in {1..4}, own Philox
 * streams (statistical parity).  Outputs: ell uint32 [G][N][16], inv_deg float [G][NS], the edge list
 * in insertion order edge_links int32 [G][edge_capacity][2], edge_w int32 [G][edge_capacity], and
 * edges_out int32 [G] = edges realised (-1: a row exceeded the ELL width, redraw that board). */
int sy_sample_boards(int32_t num_nodes, int32_t node_stride, int32_t num_edges, int32_t max_edges_per_node, uint64_t seed,
                     int32_t num_graphs, uint32_t *ell, float *inv_deg, int32_t *edge_links, int32_t *edge_w,
                     int32_t *edges_out, int32_t edge_capacity, void *stream);

#ifdef __cplusplus
}
#endif
#endif
