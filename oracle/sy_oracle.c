/*
 * sy_oracle.c — CPU ORACLE (test infrastructure only; see sy_oracle.h for the rules).
 *
 * Plain-C restatement of the reference's env.step hot path.  Citations are file:line under
 * /root/reference/src/environment/.  Compile with -ffp-contract=off: reward arithmetic must
 * follow the reference's Python float64 operation order exactly (no fused multiply-add).
 */
#include "sy_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* graph                                                                                      */
/* ------------------------------------------------------------------------------------------ */

syo_graph *syo_graph_create(int32_t N, int32_t E, const int32_t *links, const int32_t *w) {
    if (N <= 0 || E < 0) return NULL;
    for (int32_t e = 0; e < E; ++e) {
        int32_t u = links[2 * e], v = links[2 * e + 1];
        if (u < 0 || v < 0 || u >= N || v >= N || u == v || w[e] < 0) return NULL;
    }
    syo_graph *g = (syo_graph *)calloc(1, sizeof(*g));
    g->N = N;
    g->E = E;
    g->wmin = (int32_t *)malloc(sizeof(int32_t) * (size_t)N * N);
    g->dist = (int32_t *)malloc(sizeof(int32_t) * (size_t)N * N);
    g->deg = (int32_t *)calloc((size_t)N, sizeof(int32_t));
    for (size_t i = 0; i < (size_t)N * N; ++i) {
        g->wmin[i] = -1;
        g->dist[i] = SYO_INF;
    }
    /* undirected board: both edge-list columns are scanned (yard.py:439-452, pathfinding.py:91-103);
     * parallel edges collapse to their minimum weight (yard.py:460-465). */
    for (int32_t e = 0; e < E; ++e) {
        int32_t u = links[2 * e], v = links[2 * e + 1];
        int32_t cur = g->wmin[(size_t)u * N + v];
        if (cur < 0 || w[e] < cur) {
            g->wmin[(size_t)u * N + v] = w[e];
            g->wmin[(size_t)v * N + u] = w[e];
        }
    }
    for (int32_t u = 0; u < N; ++u) {
        g->dist[(size_t)u * N + u] = 0;
        for (int32_t v = 0; v < N; ++v)
            if (g->wmin[(size_t)u * N + v] >= 0) {
                g->deg[u]++;
                g->dist[(size_t)u * N + v] = g->wmin[(size_t)u * N + v];
            }
    }
    /* neighbour lists (ascending), so the batched engine does not rescan dense rows every step */
    g->nbr_start = (int32_t *)calloc((size_t)N + 1, sizeof(int32_t));
    for (int32_t u = 0; u < N; ++u) g->nbr_start[u + 1] = g->nbr_start[u] + g->deg[u];
    g->nbr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->nbr_start[N] > 0 ? g->nbr_start[N] : 1));
    for (int32_t u = 0, k = 0; u < N; ++u)
        for (int32_t v = 0; v < N; ++v)
            if (g->wmin[(size_t)u * N + v] >= 0) g->nbr[k++] = v;
    /* pathfinding.py:34-137 is textbook Dijkstra on integer weights; all-pairs Floyd-Warshall gives
     * the same distances (exact integers). */
    for (int32_t k = 0; k < N; ++k)
        for (int32_t i = 0; i < N; ++i) {
            int32_t dik = g->dist[(size_t)i * N + k];
            if (dik >= SYO_INF) continue;
            for (int32_t j = 0; j < N; ++j) {
                int32_t nd = dik + g->dist[(size_t)k * N + j];
                if (nd < g->dist[(size_t)i * N + j]) g->dist[(size_t)i * N + j] = nd;
            }
        }
    return g;
}

void syo_graph_destroy(syo_graph *g) {
    if (!g) return;
    free(g->wmin);
    free(g->dist);
    free(g->deg);
    free(g->nbr_start);
    free(g->nbr);
    free(g);
}

const int32_t *syo_graph_dist(const syo_graph *g) { return g->dist; }
const int32_t *syo_graph_wmin(const syo_graph *g) { return g->wmin; }

/* yard.py:420-472 */
int32_t syo_possible_moves(const syo_graph *g, int32_t pos, int64_t money, int32_t *nodes, int32_t *weights) {
    int32_t k = 0;
    if (pos < 0 || pos >= g->N) return 0;
    for (int32_t q = g->nbr_start[pos]; q < g->nbr_start[pos + 1]; ++q) {
        const int32_t v = g->nbr[q];
        const int32_t w = g->wmin[(size_t)pos * g->N + v];
        if ((int64_t)w <= money) { /* `edges <= agent_money`, yard.py:443-444 */
            if (nodes) nodes[k] = v;
            if (weights) weights[k] = w;
            ++k;
        }
    }
    return k;
}

static int32_t move_cost(const syo_graph *g, int32_t pos, int32_t target, int64_t money) {
    /* membership test `action in possible_positions` (yard.py:168,218); returns -1 if not a member */
    if (target < 0 || target >= g->N) return -1;
    int32_t w = g->wmin[(size_t)pos * g->N + target];
    if (w >= 0 && (int64_t)w <= money) return w;
    return -1;
}

/* action_mask.py:30-84 */
void syo_action_mask_dense(const double *adjacency, const double *edge_weights, const double *tolls,
                           int32_t N, int32_t cur, double budget, uint8_t *mask) {
    for (int32_t n = 0; n < N; ++n) {
        mask[n] = 0;
        if (n == cur) continue;                               /* :66-67 */
        double a = adjacency[(size_t)cur * N + n];
        if (a == 0.0) continue;                               /* :68-69 */
        double w = edge_weights ? edge_weights[(size_t)cur * N + n] : a; /* :100-112 */
        double toll = tolls ? tolls[(size_t)cur * N + n] : 0.0;          /* :87-97  */
        double cost = w + toll;                               /* :72 */
        if (cost <= budget) mask[n] = 1;                      /* :74-76 */
    }
}

void syo_env_masks(const syo_graph *g, int32_t P, const int32_t *pos, const int32_t *money,
                   uint8_t *masks, int32_t stride) {
    /* yard.py:301-317: mask of agent a = affordable neighbours of its node with its own budget */
    for (int32_t a = 0; a <= P; ++a) {
        uint8_t *m = masks + (size_t)a * stride;
        memset(m, 0, (size_t)stride);
        for (int32_t q = g->nbr_start[pos[a]]; q < g->nbr_start[pos[a] + 1]; ++q) {
            const int32_t v = g->nbr[q];
            if (g->wmin[(size_t)pos[a] * g->N + v] <= money[a]) m[v] = 1;
        }
    }
}

void syo_default_tables(double *exp_tab, int32_t n_exp, double *cov_tab, int32_t n_cov) {
    for (int32_t d = 0; d < n_exp; ++d) exp_tab[d] = exp(-(double)d);
    for (int32_t v = 0; v < n_cov; ++v) cov_tab[v] = exp(-log1p((double)v)); /* reward_calculator.py:207 */
}

static inline double exp_neg(int32_t d, const double *tab, int32_t n) {
    if (d >= SYO_INF) return 0.0; /* exp(-inf) */
    if (tab && d < n) return tab[d];
    return exp(-(double)d);
}

/* ------------------------------------------------------------------------------------------ */
/* one transition: yard.py:144-269, reward_calculator.py:26-266                               */
/* ------------------------------------------------------------------------------------------ */

static int in_police(const int32_t *pos, int32_t P, int32_t node) {
    for (int32_t k = 1; k <= P; ++k)
        if (pos[k] == node) return 1;
    return 0;
}

int32_t syo_step_one(const syo_graph *g, int32_t P, int32_t max_t, int32_t *pos, int32_t *money,
                     int32_t *t, int32_t *visits, const int32_t *act, const double *wt,
                     const double *exp_tab, int32_t n_exp, const double *cov_tab, int32_t n_cov,
                     double *reward, uint8_t *terminated, uint8_t *truncated, int8_t *winner) {
    const int32_t N = g->N;
    /* --- MrX first, against the PRE-move police positions (yard.py:161-188).  None and -1 both
     *     leave him in place (-1 is never a member of possible_positions). */
    {
        int32_t tgt = pos[0];
        if (move_cost(g, pos[0], act[0], money[0]) >= 0) tgt = act[0];
        if (!in_police(pos, P, tgt)) pos[0] = tgt;
    }
    /* --- police strictly in index order (yard.py:191-243) */
    int no_money = 1;
    for (int32_t k = 1; k <= P; ++k) {
        if (act[k] == -1 || money[k] == 0) continue;          /* :210-215 (None is encoded as -1) */
        no_money = 0;                                          /* :216 */
        int32_t cost = move_cost(g, pos[k], act[k], money[k]);
        int32_t tgt = cost >= 0 ? act[k] : pos[k];             /* :218-229 */
        if (!in_police(pos, P, tgt) && tgt != pos[k]) {        /* :231 (list already updated) */
            pos[k] = tgt;
            money[k] -= cost;                                  /* :234-236 */
        }
    }
    for (int32_t k = 1; k <= P; ++k) visits[pos[k]] += 1;      /* :244-245 */

    /* --- outcome priority (reward_calculator.py:63-90) */
    *terminated = 0;
    *truncated = 0;
    *winner = 0;
    int ended = 0;
    if (in_police(pos, P, pos[0])) {
        reward[0] = -1.0;
        for (int32_t k = 1; k <= P; ++k) reward[k] = 1.0;
        *terminated = 1;
        *winner = 1;
        ended = 1;
    } else if (*t > max_t) { /* pre-increment timestep, :69 */
        reward[0] = 1.0;
        for (int32_t k = 1; k <= P; ++k) reward[k] = 0.0;
        *truncated = 1;
        *winner = 2;
        ended = 1;
    } else if (no_money) {
        reward[0] = 1.0;
        for (int32_t k = 1; k <= P; ++k) reward[k] = 0.0;
        *terminated = 1;
        *winner = 2;
        ended = 1;
    } else {
        /* --- shaped rewards (reward_calculator.py:94-266), Python float64 operation order */
        const double ts = (double)*t;
        double dsum = 0.0, closest = 0.0;
        for (int32_t k = 1; k <= P; ++k) {
            int32_t di = g->dist[(size_t)pos[0] * N + pos[k]];
            double d = di >= SYO_INF ? INFINITY : (double)di;
            dsum += d;
            if (k == 1 || d < closest) closest = d;
        }
        double avg = dsum / (double)P; /* np.mean, :134 */
        double cnt0 = (double)syo_possible_moves(g, pos[0], money[0], NULL, NULL); /* :139 */
        reward[0] = ((wt[SYO_W_MRX_CLOSEST] * (-1.0 / (closest + 1.0))
                      + wt[SYO_W_MRX_AVERAGE] * (-1.0 / (avg + 1.0)))
                     + wt[SYO_W_MRX_POSITION] * cnt0)
                    + (1.0 - wt[SYO_W_MRX_TIME]) * (0.1 * ts); /* :140-148 */
        for (int32_t i = 0; i < P; ++i) {
            int32_t pi = pos[i + 1];
            double e_mrx = exp_neg(g->dist[(size_t)pi * N + pos[0]], exp_tab, n_exp);
            double group = 0.0, overlap = 0.0, prox = 0.0;
            for (int32_t j = 0; j < P; ++j) {
                if (j == i) continue;
                int32_t dij = g->dist[(size_t)pi * N + pos[j + 1]];
                double e = exp_neg(dij, exp_tab, n_exp);
                group += e;                                    /* :185-189 */
                if (dij <= 1) overlap += 1.0;                  /* :192-196 */
                else prox += e;                                /* :198-202 */
            }
            /* quirk kept for parity: agent index i (not i+1) -> MrX's / previous police's budget (:190) */
            double cnt = (double)syo_possible_moves(g, pi, money[i], NULL, NULL);
            int32_t vc = visits[pi];
            double cov = (cov_tab && vc < n_cov) ? cov_tab[vc] : exp(-log1p((double)vc)); /* :204-207 */
            reward[i + 1] = (((((wt[SYO_W_POLICE_DISTANCE] * e_mrx
                                 + wt[SYO_W_POLICE_GROUP] * group)
                                + wt[SYO_W_POLICE_POSITION] * cnt)
                               + (1.0 - wt[SYO_W_POLICE_TIME]) * (0.05 * ts))
                              + wt[SYO_W_POLICE_PROXIMITY] * prox)
                             - wt[SYO_W_POLICE_OVERLAP] * overlap)
                            + wt[SYO_W_POLICE_COVERAGE] * cov; /* :214-229 */
        }
    }
    *t += 1; /* yard.py:355 */
    return ended;
}

/* ------------------------------------------------------------------------------------------ */
/* belief: deterministic restatement of belief_module.py:69-111                               */
/* ------------------------------------------------------------------------------------------ */

void syo_belief_update(const syo_graph *g, double *b, const int32_t *hint, int32_t n_hint,
                       int32_t reveal, const int32_t *zero_nodes, int32_t n_zero) {
    const int32_t N = g->N;
    if (reveal >= 0) { /* :86-88 -> reset(reveal): every particle on that node */
        for (int32_t j = 0; j < N; ++j) b[j] = 0.0;
        b[reveal] = 1.0;
        return;
    }
    double *nb = (double *)calloc((size_t)N, sizeof(double));
    /* :91-98: each particle hops to a uniformly chosen neighbour; stays only when isolated */
    for (int32_t j = 0; j < N; ++j) {
        double acc = g->deg[j] == 0 ? b[j] : 0.0;
        for (int32_t q = g->nbr_start[j]; q < g->nbr_start[j + 1]; ++q) { /* undirected: in-neighbours == neighbours */
            const int32_t i = g->nbr[q];
            acc += b[i] / (double)g->deg[i];
        }
        nb[j] = acc;
    }
    if (n_hint > 0) { /* :102-105: likelihood 0.1 + 0.9*hint_mask */
        for (int32_t j = 0; j < N; ++j) {
            int hit = 0;
            for (int32_t h = 0; h < n_hint; ++h)
                if (hint[h] == j) hit = 1;
            nb[j] *= hit ? 1.0 : 0.1;
        }
    }
    for (int32_t z = 0; z < n_zero; ++z)
        if (zero_nodes[z] >= 0 && zero_nodes[z] < N) nb[zero_nodes[z]] = 0.0;
    double s = 0.0;
    for (int32_t j = 0; j < N; ++j) s += nb[j];
    if (s == 0.0) { /* BeliefState.distribution fallback, :32-39 */
        for (int32_t j = 0; j < N; ++j) b[j] = 1.0 / (double)N;
    } else {
        for (int32_t j = 0; j < N; ++j) b[j] = nb[j] / s;
    }
    free(nb);
}

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-7 (Salmon et al. 2011) — the counter-based RNG the device engine also uses      */
/* ------------------------------------------------------------------------------------------ */

void syo_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 7; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define PURPOSE_ACT 1u
#define PURPOSE_RESET 2u

static inline uint32_t draw(uint64_t gid, uint32_t ctr, uint32_t purpose, uint32_t idx, uint64_t seed) {
    uint32_t o[4];
    syo_philox4x32((uint32_t)gid, (uint32_t)(gid >> 32), ctr, (purpose << 8) | idx,
                   (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return o[0];
}

/* action draws: one Philox block serves 4 consecutive steps (word = step_count & 3) */
static inline uint32_t draw_action(uint64_t gid, uint32_t step_count, uint32_t agent, uint64_t seed) {
    uint32_t o[4];
    syo_philox4x32((uint32_t)gid, (uint32_t)(gid >> 32), step_count >> 2, (PURPOSE_ACT << 8) | agent,
                   (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return o[step_count & 3u];
}

static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* ------------------------------------------------------------------------------------------ */
/* batched engine (spec shared with the device path; see DESIGN.md "Engine semantics")        */
/* ------------------------------------------------------------------------------------------ */

static void init_episode(const syo_batch_config *c, const syo_graph *g, syo_batch_state *s, int32_t e,
                         const int32_t *starts) {
    const int32_t A = c->P + 1, NS = c->node_stride;
    for (int32_t a = 0; a < A; ++a) {
        s->pos[(size_t)e * A + a] = starts[a];
        s->money[(size_t)e * A + a] = a == 0 ? SYO_MRX_MONEY : c->money0; /* yard.py:117-119 */
    }
    s->t[e] = 0;
    memset(s->visits + (size_t)e * NS, 0, sizeof(int32_t) * (size_t)NS); /* yard.py:85 */
    if (s->belief) {
        double *b = s->belief + (size_t)e * NS;
        for (int32_t j = 0; j < NS; ++j) b[j] = 0.0;
        if (c->belief_init_onehot) b[starts[0]] = 1.0;
        else
            for (int32_t j = 0; j < c->N; ++j) b[j] = 1.0 / (double)c->N; /* belief_module.py:53-55 in expectation */
    }
    syo_env_masks(g, c->P, s->pos + (size_t)e * A, s->money + (size_t)e * A,
                  s->mask + (size_t)e * A * NS, NS);
}

static void sample_starts(const syo_batch_config *c, int32_t e, uint32_t ctr, uint64_t seed, int32_t *starts) {
    /* distinct start nodes, uniform over ordered tuples of distinct nodes (yard.py:112-116
     * np.random.choice(N, A, replace=False); own RNG stream, engine-defined):
     *   boards with N >= 2 A^2 (collisions are rare): REJECTION of whole tuples — attempt j = 0, 1, ... takes word
     *   (j & 3) of Philox block (env, ctr, RESET << 8 | (j >> 2) << 3 | agent), node = mulhi(word, N), and the first
     *   attempt whose nodes are pairwise distinct wins (same distribution as drawing without replacement);
     *   smaller boards, or 128 failed attempts (probability < 4^-128): sequential draws without replacement from
     *   word 0 of block (env, ctr, RESET << 8 | agent). */
    const int32_t A = c->P + 1, N = c->N;
    const uint64_t gid = c->env_id_offset + (uint64_t)e;
    if (N >= 2 * A * A) {
        for (uint32_t j = 0; j < 128; ++j) {
            int ok = 1;
            for (int32_t i = 0; i < A; ++i) {
                uint32_t o[4];
                syo_philox4x32((uint32_t)gid, (uint32_t)(gid >> 32), ctr, (PURPOSE_RESET << 8) | ((j >> 2) << 3) | (uint32_t)i,
                               (uint32_t)seed, (uint32_t)(seed >> 32), o);
                starts[i] = (int32_t)mulhi32(o[j & 3u], (uint32_t)N);
                for (int32_t k = 0; k < i; ++k) ok = ok && starts[k] != starts[i];
            }
            if (ok) return;
        }
    }
    int32_t sorted[SYO_MAX_AGENTS];
    for (int32_t i = 0; i < A; ++i) {
        uint32_t x = draw(gid, ctr, PURPOSE_RESET, (uint32_t)i, seed);
        int32_t r = (int32_t)mulhi32(x, (uint32_t)(N - i));
        int32_t n = i;
        for (int32_t j = 0; j < n; ++j)
            if (r >= sorted[j]) ++r;
        starts[i] = r;
        int32_t j = n;
        while (j > 0 && sorted[j - 1] > r) { sorted[j] = sorted[j - 1]; --j; }
        sorted[j] = r;
    }
}

void syo_batch_reset(const syo_batch_config *c, syo_graph *const *graphs, const int32_t *env_graph,
                     syo_batch_state *s, const uint8_t *env_sel, uint64_t seed) {
    const int32_t A = c->P + 1;
    for (int32_t e = 0; e < c->B; ++e) {
        if (env_sel && !env_sel[e]) continue;
        if (!env_sel) s->step_count[e] = 0;
        int32_t starts[SYO_MAX_AGENTS];
        sample_starts(c, e, s->step_count[e], seed, starts);
        init_episode(c, graphs[env_graph[e]], s, e, starts);
        for (int32_t a = 0; a < A; ++a) s->reward[(size_t)e * A + a] = 0.0;
        s->terminated[e] = 0; s->truncated[e] = 0; s->winner[e] = 0;
    }
}

void syo_batch_reset_to(const syo_batch_config *c, syo_graph *const *graphs, const int32_t *env_graph,
                        syo_batch_state *s, const int32_t *starts) {
    const int32_t A = c->P + 1;
    for (int32_t e = 0; e < c->B; ++e) {
        s->step_count[e] = 0;
        init_episode(c, graphs[env_graph[e]], s, e, starts + (size_t)e * A);
        for (int32_t a = 0; a < A; ++a) s->reward[(size_t)e * A + a] = 0.0;
        s->terminated[e] = 0; s->truncated[e] = 0; s->winner[e] = 0;
    }
}

static void belief_env_step(const syo_batch_config *c, const syo_graph *g, double *b,
                            const int32_t *pos, int32_t t_post) {
    /* engine-defined schedule: MrX shows himself when the post-increment timestep is a multiple of
     * reveal_interval; otherwise diffusion, optionally with zero likelihood on police nodes */
    int reveal = (c->reveal_interval > 0 && t_post % c->reveal_interval == 0) ? pos[0] : -1;
    syo_belief_update(g, b, NULL, 0, reveal, c->police_evidence ? pos + 1 : NULL,
                      c->police_evidence ? c->P : 0);
}

static void step_env(const syo_batch_config *c, const syo_graph *g, syo_batch_state *s, int32_t e,
                     const int32_t *act, const double *wt, const double *exp_tab, int32_t n_exp,
                     const double *cov_tab, int32_t n_cov, uint64_t seed) {
    const int32_t A = c->P + 1, NS = c->node_stride;
    int32_t *pos = s->pos + (size_t)e * A, *money = s->money + (size_t)e * A;
    int32_t ended = syo_step_one(g, c->P, c->max_t, pos, money, &s->t[e], s->visits + (size_t)e * NS,
                                 act, wt, exp_tab, n_exp, cov_tab, n_cov,
                                 s->reward + (size_t)e * A, &s->terminated[e], &s->truncated[e], &s->winner[e]);
    s->step_count[e] += 1;
    if (ended && c->auto_reset) {
        int32_t starts[SYO_MAX_AGENTS];
        sample_starts(c, e, s->step_count[e], seed, starts);
        init_episode(c, g, s, e, starts);
    } else {
        if (s->belief) belief_env_step(c, g, s->belief + (size_t)e * NS, pos, s->t[e]);
        syo_env_masks(g, c->P, pos, money, s->mask + (size_t)e * A * NS, NS);
    }
}

void syo_batch_step(const syo_batch_config *c, syo_graph *const *graphs, const int32_t *env_graph,
                    syo_batch_state *s, const int32_t *actions, const double *wt,
                    const double *exp_tab, int32_t n_exp, const double *cov_tab, int32_t n_cov, uint64_t seed) {
    const int32_t A = c->P + 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(c->threads > 1 ? c->threads : 1)
#endif
    for (int32_t e = 0; e < c->B; ++e)
        step_env(c, graphs[env_graph[e]], s, e, actions + (size_t)e * A, wt, exp_tab, n_exp, cov_tab, n_cov, seed);
}

static void sample_actions(const syo_batch_config *c, const syo_graph *g, const syo_batch_state *s,
                           int32_t e, uint64_t seed, int32_t *act) {
    /* uniform over the agent's valid mask, -1 when empty (random_agent.py semantics) */
    const int32_t A = c->P + 1;
    int32_t nodes[4096];
    for (int32_t a = 0; a < A; ++a) {
        int32_t k = syo_possible_moves(g, s->pos[(size_t)e * A + a], s->money[(size_t)e * A + a], nodes, NULL);
        uint32_t x = draw_action(c->env_id_offset + (uint64_t)e, s->step_count[e], (uint32_t)a, seed);
        act[a] = k == 0 ? -1 : nodes[mulhi32(x, (uint32_t)k)];
    }
}

void syo_batch_rollout(const syo_batch_config *c, syo_graph *const *graphs, const int32_t *env_graph,
                       syo_batch_state *s, int32_t T, const double *wt,
                       const double *exp_tab, int32_t n_exp, const double *cov_tab, int32_t n_cov,
                       uint64_t seed, syo_traj *tr) {
    const int32_t A = c->P + 1, NS = c->node_stride, B = c->B;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(c->threads > 1 ? c->threads : 1)
#endif
    for (int32_t e = 0; e < B; ++e) {
        const syo_graph *g = graphs[env_graph[e]];
        for (int32_t st = 0; st < T; ++st) {
            size_t rec = (size_t)st * B + e;
            int32_t act[SYO_MAX_AGENTS];
            sample_actions(c, g, s, e, seed, act);
            if (tr) { /* record the pre-step observation and the chosen action */
                if (tr->pos) memcpy(tr->pos + rec * A, s->pos + (size_t)e * A, sizeof(int32_t) * A);
                if (tr->money) memcpy(tr->money + rec * A, s->money + (size_t)e * A, sizeof(int32_t) * A);
                if (tr->t) tr->t[rec] = s->t[e];
                if (tr->action) memcpy(tr->action + rec * A, act, sizeof(int32_t) * A);
                if (tr->mask) memcpy(tr->mask + rec * A * NS, s->mask + (size_t)e * A * NS, (size_t)A * NS);
                if (tr->belief && s->belief)
                    memcpy(tr->belief + rec * NS, s->belief + (size_t)e * NS, sizeof(double) * NS);
            }
            step_env(c, g, s, e, act, wt, exp_tab, n_exp, cov_tab, n_cov, seed);
            if (tr) {
                if (tr->reward) memcpy(tr->reward + rec * A, s->reward + (size_t)e * A, sizeof(double) * A);
                if (tr->terminated) tr->terminated[rec] = s->terminated[e];
                if (tr->truncated) tr->truncated[rec] = s->truncated[e];
                if (tr->winner) tr->winner[rec] = s->winner[e];
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * returns / advantages (agent/mappo_agent.py:247-258) and the GAE generalisation
 * ------------------------------------------------------------------------------------------------ */
void syo_discounted_returns_f32(const float *reward, const uint8_t *done, const float *values, int32_t T, int32_t cols,
                                float gamma, float *returns, float *adv) {
    for (int32_t c = 0; c < cols; ++c) {
        float run = 0.0f;                                        /* discounted_reward = 0          (:249) */
        for (int32_t t = T - 1; t >= 0; --t) {                   /* for i in reversed(range(...))  (:250) */
            const size_t i = (size_t)t * cols + c;
            const float nd = 1.0f - (done[i] ? 1.0f : 0.0f);
            const float gr = gamma * run;                        /* self.gamma * discounted_reward        */
            run = reward[i] + gr * nd;                           /* rewards_b[i] + ... * (1 - dones_b[i]) (:251-253) */
            returns[i] = run;                                    /* (:254) */
            if (adv) adv[i] = values ? run - values[i] : run;    /* advantages = returns - values (:256) */
        }
    }
}

void syo_gae_f64(const double *reward, const uint8_t *done, const double *values, const double *last_value, int32_t T,
                 int32_t cols, double gamma, double lam, double *adv, double *returns) {
    for (int32_t c = 0; c < cols; ++c) {
        double run = 0.0, nxt = last_value ? last_value[c] : 0.0;
        for (int32_t t = T - 1; t >= 0; --t) {
            const size_t i = (size_t)t * cols + c;
            const double nd = 1.0 - (done[i] ? 1.0 : 0.0);
            const double delta = (reward[i] + (gamma * nxt) * nd) - values[i];
            run = delta + ((gamma * lam) * nd) * run;
            adv[i] = run;
            if (returns) returns[i] = run + values[i];
            nxt = values[i];
        }
    }
}
