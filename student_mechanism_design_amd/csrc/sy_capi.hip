// sy_capi.hip — the C ABI declared in include/sy_env.h (no torch, no C++ types across the boundary).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "sy_kernels.h"

struct sy_env {
    sy_env_config cfg;
    sy::EngineParams p;
    bool has_graph, has_rewards, has_state;
    int wpb, blocks;
    size_t lds;
};

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "", long long x = 0, long long y = 0) {
    std::snprintf(g_err, sizeof(g_err), fmt, a, x, y);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return SY_ERR_HIP;
}

constexpr size_t kMaxLds = 160 * 1024;
}  // namespace

extern "C" {

#ifndef SY_BUILD_ID
#define SY_BUILD_ID "unstamped"
#endif
int sy_abi_version(void) { return SY_ABI_VERSION; }
const char* sy_build_id(void) { return SY_BUILD_ID; }
int sy_record_words(int32_t num_agents) { return (5 * num_agents + 4 + 3) & ~3; }
const char* sy_last_error(void) { return g_err; }

int sy_env_create(const sy_env_config* c, sy_env** out) {
    if (!c || !out) return fail(SY_ERR_INVALID, "sy_env_create: null argument%s");
    *out = nullptr;
    if (c->num_envs <= 0) return fail(SY_ERR_INVALID, "num_envs must be > 0%s");
    if (c->num_police < 1 || c->num_police + 1 > SY_MAX_AGENTS)
        return fail(SY_ERR_INVALID, "num_police must be in [1, %s%lld]", "", SY_MAX_AGENTS - 1);
    if (c->num_nodes < c->num_police + 1 || c->num_nodes > SY_MAX_NODES)
        return fail(SY_ERR_INVALID, "num_nodes must be in [num_police+1, %s%lld]", "", SY_MAX_NODES);
    if (c->node_stride < c->num_nodes || (c->node_stride & 15))
        return fail(SY_ERR_INVALID, "node_stride must be a multiple of 16 and >= num_nodes%s");
    if (c->num_graphs < 1) return fail(SY_ERR_INVALID, "num_graphs must be >= 1%s");
    if (c->agent_money < 0 || c->agent_money > 0xfffe)
        return fail(SY_ERR_INVALID, "agent_money must be in [0, 65534]%s");
    if (c->max_timestep < 0) return fail(SY_ERR_INVALID, "max_timestep must be >= 0%s");
    if (c->reveal_interval < 0) return fail(SY_ERR_INVALID, "reveal_interval must be >= 0%s");
    if (c->waves_per_block < 0 || c->waves_per_block > 16 || (c->waves_per_block > 7 && (c->waves_per_block & 1)))
        return fail(SY_ERR_INVALID, "waves_per_block must be in [0, 16], odd values at most 7%s");
    sy_env* e = new (std::nothrow) sy_env();
    if (!e) return fail(SY_ERR_INVALID, "out of host memory%s");
    std::memset(e, 0, sizeof(*e));
    e->cfg = *c;
    sy::EngineParams& p = e->p;
    p.B = c->num_envs;
    p.N = c->num_nodes;
    p.NS = c->node_stride;
    p.P = c->num_police;
    p.A = c->num_police + 1;
    p.G = c->num_graphs;
    p.money0 = c->agent_money;
    p.max_t = c->max_timestep;
    p.reveal_k = c->reveal_interval;
    p.police_ev = c->police_evidence ? 1 : 0;
    p.belief_onehot = c->belief_init_onehot ? 1 : 0;
    p.auto_reset = c->auto_reset ? 1 : 0;
    p.env_id_offset = c->env_id_offset;
    // per-episode LDS slice: mask rows, visit counters, belief scratch, hand-off ring, sync words
    p.wave_lds_bytes = p.A * p.NS + p.NS * 4 + (p.NS + 16) * 8 + SY_RING * SY_RING_ENTRY_BYTES + 16 + 256;
    p.rec_words = sy_record_words(p.A);
    p.scan_w = 16;
    p.max_deg = 16;
    // per-block LDS: board ELL (64 B/node) + belief gather offsets (32 B/node) + reward tables
    const size_t ell_bytes = (size_t)p.N * SY_ELL_WIDTH * 6 + (size_t)(4 * SY_LDS_TABLE + 4 + SY_LDS_AVGTAB + 16) * sizeof(double);
    // The paired rollout kernel runs one wave per episode (a move wave and a belief wave per two episodes),
    // so a 1024-thread block holds 16 episodes.  One such block fills a CU at 4 waves per SIMD: all waves of
    // the CU then have the same age and advance at the same rate, whereas two co-resident 8-episode blocks
    // finish 25 % apart (the instruction arbiter favours the older block) and leave the CU half empty
    // for the tail of the launch.  Odd block sizes use the unpaired kernel (1.5 waves per episode, 768-thread
    // blocks: at most 7).
    int wpb = c->waves_per_block ? c->waves_per_block : 16;
    while (wpb > 1 && ell_bytes + (size_t)wpb * p.wave_lds_bytes > kMaxLds) wpb -= (wpb > 8) ? 2 : 1;
    if (ell_bytes + (size_t)wpb * p.wave_lds_bytes > kMaxLds) {
        delete e;
        return fail(SY_ERR_INVALID, "board does not fit in LDS%s");
    }
    // kernel-side cursors are 32-bit byte offsets (one uniform 64-bit base + a per-lane offset): reject batches
    // whose per-step rows do not fit
    {
        const unsigned long long rowmax = 0xffffffffull;
        if ((unsigned long long)p.B * p.NS * 4ull > rowmax || (unsigned long long)p.B * p.A * p.NS > rowmax ||
            (unsigned long long)p.B * p.rec_words * 4ull > rowmax) {
            delete e;
            return fail(SY_ERR_INVALID, "num_envs * node_stride too large for one launch (a per-step row must stay below 4 GiB)%s");
        }
    }
    e->wpb = wpb;
    p.wpb = wpb;
    e->blocks = (p.B + wpb - 1) / wpb;
    e->lds = ell_bytes + (size_t)wpb * p.wave_lds_bytes;
    *out = e;
    return SY_OK;
}

int sy_env_destroy(sy_env* env) {
    delete env;
    return SY_OK;
}

int sy_env_launch_info(const sy_env* env, int32_t* wpb, int32_t* blocks, int32_t* lds_bytes) {
    if (!env) return fail(SY_ERR_INVALID, "null env%s");
    if (wpb) *wpb = env->wpb;
    if (blocks) *blocks = env->blocks;
    if (lds_bytes) *lds_bytes = (int32_t)env->lds;
    return SY_OK;
}

int sy_env_rollout_kernel_name(const sy_env* env, int32_t record, char* buf, int32_t buf_len) {
    if (!env || !buf || buf_len < 1) return fail(SY_ERR_INVALID, "sy_env_rollout_kernel_name: null argument%s");
    if (!env->has_graph) return fail(SY_ERR_STATE, "%s: call sy_env_set_graph_pool first", "sy_env_rollout_kernel_name");
    const sy::RolloutPlan pl = sy::plan_rollout(env->p, record != 0, env->wpb, env->lds);
    sy::rollout_plan_name(pl, buf, (size_t)buf_len);
    return SY_OK;
}

int sy_env_set_graph_pool(sy_env* env, const uint32_t* ell, const uint16_t* apsp, const float* inv_deg,
                          const int32_t* env_graph, int32_t max_degree) {
    if (!env || !ell || !apsp || !inv_deg || !env_graph) return fail(SY_ERR_INVALID, "sy_env_set_graph_pool: null argument%s");
    if ((reinterpret_cast<uintptr_t>(ell) & 15)) return fail(SY_ERR_INVALID, "ell must be 16-byte aligned%s");
    env->p.ell = ell;
    env->p.apsp = apsp;
    env->p.inv_deg = inv_deg;
    env->p.env_graph = env_graph;
    if (max_degree < 0 || max_degree > SY_ELL_WIDTH) return fail(SY_ERR_INVALID, "max_degree must be in [0, 16] (0 = unknown)%s");
    // ELL columns scanned per agent: 8, 12 or 16 (division-free lane mapping), or the pool's exact widest row
    // when that lets more agents share a scan pass and they are needed (7 agents fit one pass up to rows of 9,
    // 6 agents up to rows of 10); instances for at most 5 agents therefore only ever see 8 / 12 / 16
    {
        const int md = (max_degree <= 0 || max_degree > 16) ? 16 : (max_degree < 8 ? 8 : max_degree);
        env->p.max_deg = (max_degree <= 0 || max_degree > 16) ? 16 : max_degree;
        const int coarse = md <= 8 ? 8 : (md <= 12 ? 12 : 16);
        env->p.scan_w = (env->p.A > 64 / coarse && 64 / md > 64 / coarse) ? md : coarse;
    }
    env->has_graph = true;
    return SY_OK;
}

int sy_env_set_belief_layout(sy_env* env, const uint16_t* gather_offsets, const uint16_t* node_slot) {
    if (!env) return fail(SY_ERR_INVALID, "sy_env_set_belief_layout: null env%s");
    if ((gather_offsets == nullptr) != (node_slot == nullptr)) return fail(SY_ERR_INVALID, "sy_env_set_belief_layout: both tables or neither%s");
    if (gather_offsets && (reinterpret_cast<uintptr_t>(gather_offsets) & 7)) return fail(SY_ERR_INVALID, "sy_env_set_belief_layout: gather_offsets must be 8-byte aligned%s");
    if (gather_offsets && env->p.N > 256) return fail(SY_ERR_INVALID, "sy_env_set_belief_layout: boards of more than 256 nodes run on kernels without a layout%s");
    env->p.bel_gather = gather_offsets;
    env->p.bel_slot = node_slot;
    return SY_OK;
}

int sy_env_set_rewards(sy_env* env, const double* w, const double* exp_tab, int32_t n_exp, const double* cov_tab,
                       int32_t n_cov) {
    if (!env || !w || !exp_tab || !cov_tab) return fail(SY_ERR_INVALID, "sy_env_set_rewards: null argument%s");
    if (n_exp < 1 || n_cov < 1) return fail(SY_ERR_INVALID, "reward tables must not be empty%s");
    for (int i = 0; i < SY_NUM_WEIGHTS; ++i) env->p.w[i] = w[i];
    env->p.exp_tab = exp_tab;
    env->p.cov_tab = cov_tab;
    env->p.n_exp = n_exp;
    env->p.n_cov = n_cov;
    env->has_rewards = true;
    return SY_OK;
}

int sy_env_set_policy(sy_env* env, const sy_mappo_weights* w, int32_t hidden) {
    if (!env) return fail(SY_ERR_INVALID, "sy_env_set_policy: null env%s");
    if (!w) {
        env->p.pw1t = env->p.pb1 = env->p.pw2 = env->p.pb2 = nullptr;
        env->p.pbound = nullptr;
        env->p.pH = 0;
        env->p.pslice = 0;
        env->p.pcap = 0;
        return SY_OK;
    }
    if (!w->w1t || !w->b1 || !w->w2 || !w->b2) return fail(SY_ERR_INVALID, "sy_env_set_policy: w1t, b1, w2, b2 are required%s");
    if (!env->has_graph) return fail(SY_ERR_STATE, "%s: call sy_env_set_graph_pool first (the board decides the kernel instance)", "sy_env_set_policy");
    if ((env->wpb & 1) != 0) return fail(SY_ERR_INVALID, "sy_env_set_policy: needs an even waves_per_block%s");
    if (hidden < 4 || (hidden & 3)) return fail(SY_ERR_INVALID, "sy_env_set_policy: hidden must be a multiple of 4, at least 4%s");
    // the limits are those of the instance that will run (sy_dispatch.hip::plan_rollout — the launcher reads the same plan)
    const sy::RolloutPlan pl = sy::plan_rollout(env->p, true, env->wpb, env->lds, hidden);
    const int hmax = pl.family == 3 ? 128 : 64;
    if (!pl.pol)
        return fail(SY_ERR_INVALID, "sy_env_set_policy: no policy instance for this configuration (boards of more than 256 nodes, or "
                    "max_timestep >= 2^20 - 2, need agents that fit one scan pass)%s");
    if (hidden > hmax)
        return fail(SY_ERR_INVALID, "sy_env_set_policy: hidden must be at most %s%lld for this configuration (128 on the pipeline kernel: "
                    "boards of up to 256 nodes, max_timestep < 2^20 - 2; 64 otherwise)", "", hmax);
    if ((reinterpret_cast<uintptr_t>(w->w2) & 15) || (reinterpret_cast<uintptr_t>(w->w1t) & 15))
        return fail(SY_ERR_INVALID, "sy_env_set_policy: weights must be 16-byte aligned%s");
    const int pslice = pl.pslice;
    if (pl.lds > kMaxLds)
        return fail(SY_ERR_INVALID, "sy_env_set_policy: no LDS left for the policy scratch (use a smaller waves_per_block)%s");
    env->p.pw1t = w->w1t; env->p.pb1 = w->b1; env->p.pw2 = w->w2; env->p.pb2 = w->b2;
    env->p.pbound = w->logit_bound;
    env->p.pslice = pslice;
    env->p.pcap = pl.pcap;
    env->p.pH = hidden;
    return SY_OK;
}

int sy_env_bind_state(sy_env* env, const sy_env_state* s) {
    if (!env || !s) return fail(SY_ERR_INVALID, "sy_env_bind_state: null argument%s");
    if (!s->pos || !s->budget || !s->t || !s->step_count || !s->visits || !s->mask || !s->reward || !s->terminated ||
        !s->truncated || !s->winner)
        return fail(SY_ERR_INVALID, "sy_env_bind_state: only `belief` may be NULL%s");
    if ((reinterpret_cast<uintptr_t>(s->mask) & 15) || (reinterpret_cast<uintptr_t>(s->visits) & 15))
        return fail(SY_ERR_INVALID, "mask and visits must be 16-byte aligned%s");
    env->p.st = *s;
    env->has_state = true;
    return SY_OK;
}

int sy_env_bind_status(sy_env* env, uint32_t* status) {
    if (!env) return fail(SY_ERR_INVALID, "sy_env_bind_status: null env%s");
    if (reinterpret_cast<uintptr_t>(status) & 3) return fail(SY_ERR_INVALID, "status word must be 4-byte aligned%s");
    env->p.status = status;
    return SY_OK;
}

int sy_env_status(sy_env* env, void* stream, uint32_t* status_host) {
    if (!env) return fail(SY_ERR_INVALID, "sy_env_status: null env%s");
    uint32_t w = 0;
    if (env->p.status) {
        hipError_t e = hipMemcpyAsync(&w, env->p.status, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "sy_env_status");
    }
    if (status_host) *status_host = w;
    if (w == 0) return SY_OK;
    std::snprintf(g_err, sizeof(g_err), "engine status 0x%x:%s%s results of the affected launch are invalid", w,
                  (w & SY_STATUS_BELIEF_WAIT_EXPIRED) ? " a belief wave gave up waiting for its move wave;" : "",
                  (w & SY_STATUS_RING_WAIT_EXPIRED) ? " a move wave gave up waiting for ring space;" : "");
    return SY_ERR_HIP;
}

static int ready(const sy_env* env, const char* who) {
    if (!env) return fail(SY_ERR_INVALID, "%s: null env", who);
    if (!env->has_graph) return fail(SY_ERR_STATE, "%s: call sy_env_set_graph_pool first", who);
    if (!env->has_state) return fail(SY_ERR_STATE, "%s: call sy_env_bind_state first", who);
    return SY_OK;
}

int sy_env_reset(sy_env* env, const uint8_t* env_sel, uint64_t seed, void* stream) {
    int rc = ready(env, "sy_env_reset");
    if (rc) return rc;
    env->p.seed_lo = (uint32_t)seed;
    env->p.seed_hi = (uint32_t)(seed >> 32);
    hipError_t e = sy::launch_reset(env->p, env_sel, nullptr, env_sel ? 0 : 1, env->blocks, env->wpb, env->lds,
                                    (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_env_reset launch");
}

int sy_env_reset_to(sy_env* env, const int32_t* starts, void* stream) {
    int rc = ready(env, "sy_env_reset_to");
    if (rc) return rc;
    if (!starts) return fail(SY_ERR_INVALID, "sy_env_reset_to: null starts%s");
    hipError_t e = sy::launch_reset(env->p, nullptr, starts, 1, env->blocks, env->wpb, env->lds, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_env_reset_to launch");
}

int sy_env_step(sy_env* env, const int32_t* actions, void* stream) {
    int rc = ready(env, "sy_env_step");
    if (rc) return rc;
    if (!env->has_rewards) return fail(SY_ERR_STATE, "%s: call sy_env_set_rewards first", "sy_env_step");
    if (!actions) return fail(SY_ERR_INVALID, "sy_env_step: null actions%s");
    sy_rollout_buffers none;
    std::memset(&none, 0, sizeof(none));
    hipError_t e = sy::launch_step(env->p, actions, none, env->blocks, env->wpb, env->lds, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_env_step launch");
}

int sy_env_step_record(sy_env* env, const int32_t* actions, const sy_rollout_buffers* row, void* stream) {
    int rc = ready(env, "sy_env_step_record");
    if (rc) return rc;
    if (!env->has_rewards) return fail(SY_ERR_STATE, "%s: call sy_env_set_rewards first", "sy_env_step_record");
    if (!actions || !row) return fail(SY_ERR_INVALID, "sy_env_step_record: null argument%s");
    if ((reinterpret_cast<uintptr_t>(row->mask) & 15) || (reinterpret_cast<uintptr_t>(row->record) & 15))
        return fail(SY_ERR_INVALID, "sy_env_step_record: record / mask rows must be 16-byte aligned%s");
    hipError_t e = sy::launch_step(env->p, actions, *row, env->blocks, env->wpb, env->lds, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_env_step_record launch");
}

int sy_env_rollout(sy_env* env, int32_t T, const sy_rollout_buffers* out, void* stream) {
    int rc = ready(env, "sy_env_rollout");
    if (rc) return rc;
    if (!env->has_rewards) return fail(SY_ERR_STATE, "%s: call sy_env_set_rewards first", "sy_env_rollout");
    if (T < 1) return fail(SY_ERR_INVALID, "sy_env_rollout: T must be >= 1%s");
    sy_rollout_buffers o;
    if (out) o = *out;
    else std::memset(&o, 0, sizeof(o));
    if (out && !o.record) return fail(SY_ERR_INVALID, "sy_env_rollout: only `mask` and `belief` of the record may be NULL%s");
    if (!o.record && (o.mask || o.belief)) return fail(SY_ERR_INVALID, "sy_env_rollout: mask / belief need `record`%s");
    if (o.mask && (reinterpret_cast<uintptr_t>(o.mask) & 15)) return fail(SY_ERR_INVALID, "rollout mask must be 16-byte aligned%s");
    if (env->p.pw2) {       // learned policy in the kernel: recorded rollouts on single-pass boards
        if (!o.record) return fail(SY_ERR_INVALID, "sy_env_rollout: a policy rollout needs a record%s");
        // the instance is chosen per launch: a board pool bound after sy_env_set_policy may have moved it
        const sy::RolloutPlan pl = sy::plan_rollout(env->p, true, env->wpb, env->lds);
        if (!pl.pol || pl.pslice != env->p.pslice || env->p.pH > (pl.family == 3 ? 128 : 64) || pl.lds > kMaxLds)
            return fail(SY_ERR_STATE, "%s: the policy was set for another board pool; call sy_env_set_policy again", "sy_env_rollout");
    }
    hipError_t e = sy::launch_rollout(env->p, T, o, env->blocks, env->wpb, env->lds, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_env_rollout launch");
}

int sy_action_mask_dense(const double* adjacency, const double* edge_weights, const double* tolls, int32_t num_nodes,
                         const int32_t* current_node, const double* budget, int32_t num_queries, uint8_t* mask,
                         void* stream) {
    if (!adjacency || !current_node || !budget || !mask) return fail(SY_ERR_INVALID, "sy_action_mask_dense: null argument%s");
    if (num_nodes < 1 || num_queries < 0) return fail(SY_ERR_INVALID, "sy_action_mask_dense: bad sizes%s");
    if (num_queries == 0) return SY_OK;
    hipError_t e = sy::launch_action_mask_dense(adjacency, edge_weights, tolls, num_nodes, current_node, budget, num_queries,
                                                mask, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_action_mask_dense launch");
}

int sy_belief_update(const uint32_t* ell, const float* inv_deg, int32_t num_nodes, int32_t node_stride, float* belief,
                     const int32_t* hint, int32_t hint_width, const int32_t* reveal, int32_t num_queries, void* stream) {
    if (!ell || !inv_deg || !belief) return fail(SY_ERR_INVALID, "sy_belief_update: null argument%s");
    if (num_nodes < 1 || num_nodes > SY_MAX_NODES || node_stride < num_nodes || (node_stride & 15))
        return fail(SY_ERR_INVALID, "sy_belief_update: bad num_nodes / node_stride%s");
    if (hint && hint_width < 1) return fail(SY_ERR_INVALID, "sy_belief_update: hint_width must be >= 1%s");
    if (num_queries < 0) return fail(SY_ERR_INVALID, "sy_belief_update: bad num_queries%s");
    if (num_queries == 0) return SY_OK;
    hipError_t e = sy::launch_belief_update(ell, inv_deg, num_nodes, node_stride, belief, hint, hint ? hint_width : 0, reveal,
                                            num_queries, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_belief_update launch");
}

int sy_masked_categorical_sample(const float* probs, int64_t probs_row_stride, const uint8_t* mask, int64_t mask_row_stride,
                                 int32_t num_rows, int32_t num_nodes, uint64_t seed, uint64_t offset,
                                 const uint64_t* offset_dev, int32_t default_on_empty, int32_t* action, float* log_prob,
                                 float* norm_probs, void* stream) {
    if (!probs || !mask || !action || !log_prob) return fail(SY_ERR_INVALID, "sy_masked_categorical_sample: null argument%s");
    if (num_nodes < 1 || num_nodes > SY_MAX_NODES || probs_row_stride < num_nodes || mask_row_stride < num_nodes)
        return fail(SY_ERR_INVALID, "sy_masked_categorical_sample: bad num_nodes / strides%s");
    if (num_rows < 0) return fail(SY_ERR_INVALID, "sy_masked_categorical_sample: bad num_rows%s");
    if (num_rows == 0) return SY_OK;
    hipError_t e = sy::launch_masked_sample(probs, probs_row_stride, mask, mask_row_stride, num_rows, num_nodes, seed, offset,
                                            offset_dev, default_on_empty ? 1 : 0, action, log_prob, norm_probs,
                                            (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_masked_categorical_sample launch");
}

int sy_mappo_policy_act(const int32_t* pos, const uint8_t* mask, int64_t mask_row_stride, const sy_mappo_weights* w,
                        int32_t num_envs, int32_t num_police, int32_t num_nodes, int32_t hidden, uint64_t seed, uint64_t offset,
                        const uint64_t* offset_dev, int32_t* action, float* log_prob, float* value, float* probs, void* stream) {
    if (!pos || !mask || !w || !action || !log_prob) return fail(SY_ERR_INVALID, "sy_mappo_policy_act: null argument%s");
    if (!w->w1t || !w->b1 || !w->w2t || !w->b2) return fail(SY_ERR_INVALID, "sy_mappo_policy_act: null actor weights%s");
    if (value && (!w->c1t || !w->cb1 || !w->c2 || !w->cb2)) return fail(SY_ERR_INVALID, "sy_mappo_policy_act: null critic weights%s");
    if (num_police < 1 || num_police > SY_MAX_AGENTS - 1 || num_nodes < 1 || num_nodes > SY_MAX_NODES || mask_row_stride < num_nodes)
        return fail(SY_ERR_INVALID, "sy_mappo_policy_act: bad sizes%s");
    if (hidden < 1 || hidden > 128) return fail(SY_ERR_INVALID, "sy_mappo_policy_act: hidden size must be in [1, 128]%s");
    if (num_envs < 0) return fail(SY_ERR_INVALID, "sy_mappo_policy_act: bad num_envs%s");
    if (num_envs == 0) return SY_OK;
    hipError_t e = sy::launch_mappo_policy(pos, mask, mask_row_stride, w->w1t, w->b1, w->w2t, w->b2, w->c1t, w->cb1, w->c2, w->cb2,
                                           num_envs, num_police + 1, num_nodes, hidden, seed, offset, offset_dev, action, log_prob,
                                           value, probs, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_mappo_policy_act launch");
}

int sy_gnn_padded_features(int32_t num_features) { return sy::gnn_padded_features(num_features); }
int sy_gnn_param_floats(int32_t num_features) { return sy::gnn_param_floats(num_features); }

int sy_gnn_q_act(const int32_t* pos, const float* belief, int64_t belief_row_stride, const uint8_t* mask, int64_t mask_row_stride,
                 const uint32_t* gcn_table, int32_t table_width, const float* self_coef, const int32_t* env_graph, const float* models,
                 int32_t num_envs, int32_t num_police, int32_t num_nodes, int32_t num_features, float explore_eps, uint64_t seed,
                 uint64_t offset, const uint64_t* offset_dev, int32_t* action, float* q_values, void* stream) {
    if (!pos || !mask || !gcn_table || !self_coef || !models || !action) return fail(SY_ERR_INVALID, "sy_gnn_q_act: null argument%s");
    if (num_police < 1 || num_police > SY_MAX_AGENTS - 1 || num_nodes < 1 || num_nodes > 256 || mask_row_stride < num_nodes)
        return fail(SY_ERR_INVALID, "sy_gnn_q_act: bad sizes (boards of up to 256 nodes)%s");
    if (table_width < 1 || table_width > SY_ELL_WIDTH) return fail(SY_ERR_INVALID, "sy_gnn_q_act: table_width must be in [1, 16]%s");
    if ((reinterpret_cast<uintptr_t>(gcn_table) & 7) || (reinterpret_cast<uintptr_t>(models) & 7))
        return fail(SY_ERR_INVALID, "sy_gnn_q_act: gcn_table and models must be 8-byte aligned%s");
    const int A = num_police + 1;
    if (num_features != A && num_features != A + 1)
        return fail(SY_ERR_INVALID, "sy_gnn_q_act: num_features must be num_police + 1 (agent one-hots) or + 2 (with the belief column)%s");
    if (num_features == A + 1 && (!belief || belief_row_stride < num_nodes))
        return fail(SY_ERR_INVALID, "sy_gnn_q_act: the belief column needs `belief` rows of at least num_nodes floats%s");
    if (num_features > 9) return fail(SY_ERR_INVALID, "sy_gnn_q_act: at most 9 node features%s");
    if (!(explore_eps >= 0.0f && explore_eps <= 1.0f)) return fail(SY_ERR_INVALID, "sy_gnn_q_act: explore_eps must be in [0, 1]%s");
    if (num_envs < 0) return fail(SY_ERR_INVALID, "sy_gnn_q_act: bad num_envs%s");
    if (num_envs == 0) return SY_OK;
    hipError_t e = sy::launch_gnn_q_act(pos, num_features == A + 1 ? belief : nullptr, belief_row_stride, mask, mask_row_stride,
                                        gcn_table, table_width, self_coef, env_graph, models, num_envs, A, num_nodes, num_features,
                                        explore_eps, seed, offset, offset_dev, action, q_values, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_gnn_q_act launch");
}

int sy_returns_advantages(const sy_returns_args* a, void* stream) {
    if (!a || !a->reward || !a->done_a || !a->returns) return fail(SY_ERR_INVALID, "sy_returns_advantages: null argument%s");
    if (a->T < 1 || a->B < 1 || a->A < 1 || a->A > SY_MAX_AGENTS) return fail(SY_ERR_INVALID, "sy_returns_advantages: bad sizes%s");
    if (a->mode != 0 && a->mode != 1) return fail(SY_ERR_INVALID, "sy_returns_advantages: mode must be 0 or 1%s");
    if (a->done_bytes != 1 && a->done_bytes != 4) return fail(SY_ERR_INVALID, "sy_returns_advantages: done_bytes must be 1 or 4%s");
    if (a->mode == 1 && !a->value) return fail(SY_ERR_INVALID, "sy_returns_advantages: GAE needs values%s");
    if ((long long)a->B * a->A > 0x7fffffffLL) return fail(SY_ERR_INVALID, "sy_returns_advantages: too many columns%s");
    sy::ReturnsArgs r;
    r.T = a->T; r.B = a->B; r.A = a->A; r.mode = a->mode; r.reward_f64 = a->reward_f64 ? 1 : 0;
    r.done_bytes = a->done_bytes; r.compute_f64 = a->compute_f64 ? 1 : 0;
    r.reward = a->reward; r.rs_t = a->reward_stride_t; r.rs_b = a->reward_stride_b;
    r.done_a = a->done_a; r.done_b = a->done_b; r.ds_t = a->done_stride_t; r.ds_b = a->done_stride_b;
    r.value = a->value; r.vs_t = a->value_stride_t; r.vs_b = a->value_stride_b; r.vs_a = a->value_stride_a;
    r.last_value = a->last_value; r.lv_b = a->last_value_stride_b; r.lv_a = a->last_value_stride_a;
    r.gamma = a->gamma; r.lam = a->lambda; r.returns = a->returns; r.adv = a->adv;
    hipError_t e = sy::launch_returns(r, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_returns_advantages launch");
}

int32_t sy_ppo_slab_floats(int32_t num_nodes, int32_t hidden) { return sy::ppo_slab_floats(num_nodes, hidden); }

int64_t sy_ppo_scratch_floats(int32_t num_agents, int32_t num_nodes, int32_t hidden) {
    return (int64_t)sy::ppo_max_blocks_per_role() * (num_agents + 1) * sy::ppo_slab_floats(num_nodes, hidden);
}

int64_t sy_ppo_image_bytes(int32_t num_agents, int64_t num_rows) {
    return num_agents < 1 || num_rows < 0 ? 0 : (int64_t)sy::ppo_image_size(num_agents, num_rows);
}

int sy_ppo_pack(const sy_ppo_pack_args* a, void* stream) {
    if (!a || !a->record || !a->log_prob || !a->adv || !a->team_ret || !a->env_graph || !a->image)
        return fail(SY_ERR_INVALID, "sy_ppo_pack: null argument%s");
    const int A = a->num_police + 1;
    if (a->num_police < 1 || A > SY_MAX_AGENTS || a->num_rows < 1 || a->num_envs < 1 || a->row0 < 0)
        return fail(SY_ERR_INVALID, "sy_ppo_pack: bad sizes%s");
    if (a->record_words < sy_record_words(A)) return fail(SY_ERR_INVALID, "sy_ppo_pack: record_words too small for this many agents%s");
    if (a->image_bytes < sy_ppo_image_bytes(A, a->num_rows)) return fail(SY_ERR_INVALID, "sy_ppo_pack: image too small%s");
    if (reinterpret_cast<uintptr_t>(a->image) & 15) return fail(SY_ERR_INVALID, "sy_ppo_pack: image must be 16-byte aligned%s");
    sy::PpoPackArgs k;
    k.record = a->record; k.RW = a->record_words; k.log_prob = a->log_prob; k.adv = a->adv; k.team_ret = a->team_ret;
    k.rows = a->rows; k.row0 = a->row0; k.count = a->num_rows; k.B = a->num_envs; k.env_graph = a->env_graph; k.A = A;
    k.image = a->image;
    k.chunk_rows = a->chunk_rows; k.record_chunk_stride = a->record_chunk_stride; k.log_prob_chunk_stride = a->log_prob_chunk_stride;
    if (a->chunk_rows < 0 || (a->chunk_rows > 0 && (a->record_chunk_stride < a->chunk_rows * (int64_t)a->record_words ||
                                                   a->log_prob_chunk_stride < a->chunk_rows * (int64_t)A)))
        return fail(SY_ERR_INVALID, "sy_ppo_pack: chunk strides smaller than a chunk%s");
    k.shuffle_domain = 0; k.shuffle_hb = 0; k.shuffle_seed = a->shuffle_seed;
    if (a->shuffle_domain != 0) {
        if (a->rows) return fail(SY_ERR_INVALID, "sy_ppo_pack: give `rows` or a shuffle, not both%s");
        if (a->shuffle_domain < a->num_rows || a->shuffle_domain > 0x40000000LL) return fail(SY_ERR_INVALID, "sy_ppo_pack: shuffle_domain must be in [num_rows, 2^30]%s");
        int bits = 1;
        while ((1LL << bits) < a->shuffle_domain) ++bits;
        k.shuffle_hb = (bits + 1) / 2;
        k.shuffle_domain = a->shuffle_domain;
    }
    hipError_t e = sy::launch_ppo_pack(k, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_ppo_pack launch");
}

int sy_mappo_ppo_grad(const sy_ppo_args* a, void* stream) {
    if (!a || !a->image || !a->ell || !a->params || !a->scratch || !a->grads)
        return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: null argument%s");
    if (a->adam_m || a->adam_v || a->adam_step) {
        if (!a->adam_m || !a->adam_v || !a->adam_step) return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: adam_m, adam_v and adam_step go together%s");
        if (!(a->lr > 0.0f) || !(a->beta1 >= 0.0f && a->beta1 < 1.0f) || !(a->beta2 >= 0.0f && a->beta2 < 1.0f) || !(a->eps > 0.0f))
            return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: bad Adam constants%s");
    }
    const int A = a->num_police + 1;
    if (a->num_police < 1 || A > SY_MAX_AGENTS || a->num_nodes < 2 || a->num_nodes > SY_MAX_NODES || a->num_rows < 1 || a->row0 < 0 ||
        a->image_rows < 1)
        return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: bad sizes%s");
    if (!a->row0_dev && (int64_t)a->row0 + a->num_rows > a->image_rows) return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: the minibatch ends past the image%s");
    if (a->hidden < 4 || (a->hidden & 3) || a->hidden > 128) return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: hidden must be a multiple of 4 in [4, 128]%s");
    if (sy::ppo_parts(a->num_nodes, a->hidden) < 1 || 2 * (A + 1) * sy::ppo_parts(a->num_nodes, a->hidden) > SY_PPO_MAX_ROLES)
        return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: nodes x hidden too large (a gradient table is cut into too many row ranges)%s");
    if (a->scratch_floats < sy_ppo_scratch_floats(A, a->num_nodes, a->hidden)) return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: scratch too small%s");
    if ((reinterpret_cast<uintptr_t>(a->params) | reinterpret_cast<uintptr_t>(a->image)) & 15)
        return fail(SY_ERR_INVALID, "sy_mappo_ppo_grad: params and the image must be 16-byte aligned%s");
    sy::PpoArgs k;
    k.image = a->image; k.image_rows = a->image_rows; k.row0 = a->row0; k.row0_dev = a->row0_dev; k.mb = a->num_rows;
    k.ell = a->ell; k.A = A; k.N = a->num_nodes; k.H = a->hidden;
    k.params = a->params;
    k.clip = a->clip; k.value_coef = a->value_coef; k.partial = a->scratch; k.DN = 0; k.slab = 0; k.parts = 0; k.rpp = 0; k.nroles = 0; k.adam_step = nullptr;
    sy::PpoAdam ad;
    ad.params = a->adam_m ? a->params : nullptr; ad.m = a->adam_m; ad.v = a->adam_v; ad.step = a->adam_step;
    ad.lr = a->lr; ad.beta1 = a->beta1; ad.beta2 = a->beta2; ad.eps = a->eps;
    hipError_t e = sy::launch_ppo_grad(k, a->grads, ad, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_mappo_ppo_grad launch");
}

int sy_ppo_adam_step(float* params, const float* grads, float* adam_m, float* adam_v, int32_t* adam_step, int32_t num_police,
                     int32_t num_nodes, int32_t hidden, float lr, float beta1, float beta2, float eps, void* stream) {
    if (!params || !grads || !adam_m || !adam_v || !adam_step) return fail(SY_ERR_INVALID, "sy_ppo_adam_step: null argument%s");
    const int A = num_police + 1;
    if (num_police < 1 || A > SY_MAX_AGENTS || num_nodes < 2 || num_nodes > SY_MAX_NODES || hidden < 4 || (hidden & 3) || hidden > 128)
        return fail(SY_ERR_INVALID, "sy_ppo_adam_step: bad sizes%s");
    if (!(lr > 0.0f) || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps > 0.0f))
        return fail(SY_ERR_INVALID, "sy_ppo_adam_step: bad Adam constants%s");
    sy::PpoAdam ad;
    ad.params = params; ad.m = adam_m; ad.v = adam_v; ad.step = adam_step; ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps;
    hipError_t e = sy::launch_ppo_adam(grads, ad, A, num_nodes, hidden, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_ppo_adam_step launch");
}

int sy_build_apsp(const uint32_t* ell, int32_t num_nodes, int32_t num_graphs, uint16_t* apsp, void* stream) {
    if (!ell || !apsp) return fail(SY_ERR_INVALID, "sy_build_apsp: null argument%s");
    if (num_nodes < 1 || num_nodes > SY_MAX_NODES || num_graphs < 1) return fail(SY_ERR_INVALID, "sy_build_apsp: bad sizes%s");
    if ((reinterpret_cast<uintptr_t>(ell) & 15)) return fail(SY_ERR_INVALID, "ell must be 16-byte aligned%s");
    hipError_t e = sy::launch_apsp(ell, num_nodes, num_graphs, apsp, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_build_apsp launch");
}

int sy_sample_boards(int32_t num_nodes, int32_t node_stride, int32_t num_edges, int32_t max_edges_per_node, uint64_t seed,
                     int32_t num_graphs, uint32_t* ell, float* inv_deg, int32_t* edge_links, int32_t* edge_w,
                     int32_t* edges_out, int32_t edge_capacity, void* stream) {
    if (!ell || !inv_deg || !edge_links || !edge_w || !edges_out) return fail(SY_ERR_INVALID, "sy_sample_boards: null argument%s");
    if (num_nodes < 2 || num_nodes > SY_MAX_NODES || node_stride < num_nodes || (node_stride & 15) || num_graphs < 1)
        return fail(SY_ERR_INVALID, "sy_sample_boards: bad sizes%s");
    if (num_edges < num_nodes - 1) num_edges = num_nodes - 1;   // minimal case for a connected graph (graph_layout.py:20-21)
    if (edge_capacity < num_edges) return fail(SY_ERR_INVALID, "sy_sample_boards: edge_capacity < num_edges%s");
    if (max_edges_per_node < 1 || max_edges_per_node > SY_ELL_WIDTH) return fail(SY_ERR_INVALID, "sy_sample_boards: bad degree cap%s");
    hipError_t e = sy::launch_sample_boards(num_nodes, node_stride, num_edges, max_edges_per_node, seed, num_graphs, ell, inv_deg,
                                            edge_links, edge_w, edges_out, edge_capacity, (hipStream_t)stream);
    return e == hipSuccess ? SY_OK : hip_fail(e, "sy_sample_boards launch");
}

}  // extern "C"
