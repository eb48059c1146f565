/*
 * sy_oracle.h — CPU ORACLE for the Scotland-Yard env hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's algorithm
 * (elte-collective-intelligence/student-mechanism-design, src/environment/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path
 * (student_mechanism_design_amd/) never does and fails loudly without its HIP library.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against golden traces
 * recorded from the unmodified reference (oracle/capture_goldens.py -> tests/golden/):
 * positions, budgets, masks, flags, winner, visit counts bit-exact; float64 rewards to 1e-12;
 * action-mask known answers (the reference's own test/test_action_mask.py cases + random dense
 * cases); the deterministic belief filter against the reference's seeded test and against
 * 4e5-particle Monte-Carlo runs of ParticleBeliefTracker (tolerance = Monte-Carlo error).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef SY_ORACLE_H
#define SY_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYO_MAX_AGENTS 8
#define SYO_NUM_WEIGHTS 11
#define SYO_INF 0x3fffffff
#define SYO_MRX_MONEY 1000 /* MAX_MONEY_LIMIT, yard.py:11,117 */

/* reward weight order = REWARD_WEIGHT_NAMES, src/reward_net.py:5-17 */
enum {
    SYO_W_POLICE_DISTANCE = 0, SYO_W_POLICE_GROUP, SYO_W_POLICE_POSITION, SYO_W_POLICE_TIME,
    SYO_W_MRX_CLOSEST, SYO_W_MRX_AVERAGE, SYO_W_MRX_POSITION, SYO_W_MRX_TIME,
    SYO_W_POLICE_COVERAGE, SYO_W_POLICE_PROXIMITY, SYO_W_POLICE_OVERLAP
};

typedef struct syo_graph {
    int32_t N, E;
    int32_t *wmin; /* [N*N] min weight of an edge u-v, -1 = no edge            */
    int32_t *dist; /* [N*N] weighted shortest path, SYO_INF = unreachable      */
    int32_t *deg;  /* [N]   number of distinct neighbours                      */
    int32_t *nbr_start, *nbr; /* CSR neighbour lists, ascending               */
} syo_graph;

/* graph from the reference's board arrays (edge_links int32[E][2], edges[E]); NULL on bad input
 * (self loops, ids out of range, negative weights). */
syo_graph *syo_graph_create(int32_t N, int32_t E, const int32_t *edge_links, const int32_t *edge_w);
void syo_graph_destroy(syo_graph *g);
const int32_t *syo_graph_dist(const syo_graph *g);
const int32_t *syo_graph_wmin(const syo_graph *g);

/* yard.py:420-472 _get_possible_moves: distinct affordable neighbours ascending + min weights */
int32_t syo_possible_moves(const syo_graph *g, int32_t pos, int64_t money, int32_t *nodes, int32_t *weights);

/* action_mask.py:30-84 compute_action_mask on dense float64 rows (weights / tolls may be NULL) */
void syo_action_mask_dense(const double *adjacency, const double *edge_weights, const double *tolls,
                           int32_t N, int32_t current_node, double budget, uint8_t *mask);

/* per-agent env masks (yard.py:297-317 via compute_action_mask == possible-move set) */
void syo_env_masks(const syo_graph *g, int32_t P, const int32_t *pos, const int32_t *money,
                   uint8_t *masks, int32_t mask_stride);

/* default tables: exp_tab[d] = exp(-d), cov_tab[v] = exp(-log1p(v)) (reward_calculator.py:184-207) */
void syo_default_tables(double *exp_tab, int32_t n_exp, double *cov_tab, int32_t n_cov);

/* one env transition: yard.py:144-269 + reward_calculator.py:26-266.  actions: -1 = no-op/None.
 * pos/money are [P+1] (index 0 = MrX), visits[N] = node_visit_counts (yard.py:244-245).
 * exp_tab/cov_tab may be NULL (libm is used).  Returns 1 if the episode ended. */
int32_t syo_step_one(const syo_graph *g, int32_t P, int32_t max_t, int32_t *pos, int32_t *money,
                     int32_t *t, int32_t *visits, const int32_t *actions, const double *weights,
                     const double *exp_tab, int32_t n_exp, const double *cov_tab, int32_t n_cov,
                     double *reward, uint8_t *terminated, uint8_t *truncated, int8_t *winner);

/* deterministic forward filter restating ParticleBeliefTracker.update (belief_module.py:69-111):
 * reveal>=0 -> one-hot; else b' = normalize((b.P) * lik), lik = 0.1+0.9*[j in hint] (1 w/o hint),
 * zero mass -> uniform.  zero_nodes (nullable) = build-defined hard evidence (police-occupied). */
void syo_belief_update(const syo_graph *g, double *belief, const int32_t *hint, int32_t n_hint,
                       int32_t reveal, const int32_t *zero_nodes, int32_t n_zero);

/* counter-based RNG shared with the device path (Philox4x32-7) */
void syo_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]);

/* ---- batched env with auto-reset + in-engine uniform-random policy (mirrors the device engine) ---- */
typedef struct syo_batch_config {
    int32_t B, N, P, money0, max_t, node_stride;
    int32_t reveal_interval, police_evidence, belief_init_onehot, auto_reset;
    uint64_t env_id_offset;
    int32_t threads; /* OpenMP threads for the batch loop (<=1: serial) */
} syo_batch_config;

typedef struct syo_batch_state {
    int32_t *pos;        /* [B][A]  */
    int32_t *money;      /* [B][A]  */
    int32_t *t;          /* [B]     */
    uint32_t *step_count;/* [B]     */
    int32_t *visits;     /* [B][NS] */
    double *belief;      /* [B][NS] (nullable) */
    uint8_t *mask;       /* [B][A][NS] */
    double *reward;      /* [B][A]  */
    uint8_t *terminated; /* [B] */
    uint8_t *truncated;  /* [B] */
    int8_t *winner;      /* [B] */
} syo_batch_state;

typedef struct syo_traj {
    int32_t *pos, *money, *t, *action; /* [T][B][A] x3, t [T][B], action [T][B][A] */
    uint8_t *mask;                     /* [T][B][A][NS] */
    double *belief;                    /* [T][B][NS]    */
    double *reward;                    /* [T][B][A]     */
    uint8_t *terminated, *truncated;   /* [T][B]        */
    int8_t *winner;                    /* [T][B]        */
} syo_traj;

/* graphs: array of G graph pointers; env_graph[B] picks one per env */
void syo_batch_reset(const syo_batch_config *c, syo_graph *const *graphs, const int32_t *env_graph,
                     syo_batch_state *s, const uint8_t *env_sel, uint64_t seed);
void syo_batch_reset_to(const syo_batch_config *c, syo_graph *const *graphs, const int32_t *env_graph,
                        syo_batch_state *s, const int32_t *starts);
void syo_batch_step(const syo_batch_config *c, syo_graph *const *graphs, const int32_t *env_graph,
                    syo_batch_state *s, const int32_t *actions, const double *weights,
                    const double *exp_tab, int32_t n_exp, const double *cov_tab, int32_t n_cov,
                    uint64_t seed);
void syo_batch_rollout(const syo_batch_config *c, syo_graph *const *graphs, const int32_t *env_graph,
                       syo_batch_state *s, int32_t T, const double *weights,
                       const double *exp_tab, int32_t n_exp, const double *cov_tab, int32_t n_cov,
                       uint64_t seed, syo_traj *traj);

/* agent/mappo_agent.py:247-258: reverse discounted reward sum with done masking over a flat buffer, float32 like
 * the reference's tensors:  R_i = r_i + (gamma * R_{i+1}) * (1 - done_i)  (that operation order), adv = R - V.
 * Batched over `cols` independent columns (element (t, c) at [t * cols + c]); values may be NULL (adv = R).
 * Pinned by tests/golden/ppo_returns_reference.npz (captured from the unmodified ppo_update). */
void syo_discounted_returns_f32(const float *reward, const uint8_t *done, const float *values, int32_t T, int32_t cols,
                                float gamma, float *returns, float *adv);
/* GAE(gamma, lambda), the generalisation the build adds (the reference has none):
 * delta_t = r_t + gamma * V_{t+1} * (1 - d_t) - V_t;  A_t = delta_t + gamma * lambda * (1 - d_t) * A_{t+1};
 * returns = A + V.  last_value [cols] bootstraps V_T (NULL = 0).  float64. */
void syo_gae_f64(const double *reward, const uint8_t *done, const double *values, const double *last_value, int32_t T,
                 int32_t cols, double gamma, double lam, double *adv, double *returns);

#ifdef __cplusplus
}
#endif
#endif
