// sy_kernels.h — internal interface between the C ABI (sy_capi.hip) and the kernels (sy_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sy_env.h"

#define SY_LDS_TABLE 256   // exp(-d) / coverage / -1/(d+1) entries staged in LDS per launch block
#define SY_LDS_AVGTAB 512  // -1/(sum/P+1) entries staged in LDS per launch block
#define SY_RING 16         // move wave -> helper / belief wave ring depth (steps)
#define SY_RING_ENTRY_BYTES 64   // ring bytes per episode = SY_RING x this: 16 entries of 32 B (used) for the two older rollout
                                 // kernels, 8 entries of 128 B (8 agent slots x 16 B) for the rollout3 pipeline

namespace sy {

// Passed by value to every engine launch (kernel-argument segment, scalar loads).
struct EngineParams {
    int32_t B, N, NS, P, A, G;
    int32_t money0, max_t, reveal_k, police_ev, belief_onehot, auto_reset;
    int32_t wave_lds_bytes;       // private LDS slice per episode (mask rows, visits, belief scratch, ring, sync)
    int32_t scan_w;               // ELL columns scanned per agent: 8..16 (>= widest row of the pool)
    int32_t max_deg;              // widest ELL row of the pool (16 when the caller did not say)
    int32_t rec_words;            // dwords per packed trajectory record (sy_record_words)
    int32_t wpb;                  // episodes (move waves) per launch block
    uint32_t seed_lo, seed_hi;    // Philox key
    uint64_t env_id_offset;
    const uint32_t* ell;          // [G][N][16]
    const uint16_t* apsp;         // [G][N][N]
    const float* inv_deg;         // [G][NS]
    const uint16_t* bel_gather;   // [G][N][16] scratch byte offsets of every node's neighbour visits (sy_env_set_belief_layout) or nullptr
    const uint16_t* bel_slot;     // [G][NS]    scratch entry of every node's own value, or nullptr: entry = node
    const int32_t* env_graph;     // [B]
    double w[SY_NUM_WEIGHTS];
    const double* exp_tab;
    const double* cov_tab;
    int32_t n_exp, n_cov;
    sy_env_state st;
    // in-kernel learned policy (sy_env_set_policy); pw2 == nullptr: uniform-random
    const float* pw1t;            // [A][N][H]
    const float* pb1;             // [A][H]
    const float* pw2;             // [A][N][H]
    const float* pb2;             // [A][N]
    int32_t pH;
    int32_t pslice;               // per-episode LDS scratch of the pipeline's policy path: A * H * 4 + 128 bytes + its share of the
                                  // pair's entry list (sy_dispatch.hip::rollout_policy_slice)
    int32_t pcap;                 // affordable ELL entries an episode can have (sizes the pair's entry list of the policy path)
    const float* pbound;          // [A] upper bound of any logit of actor a (the underflow rule's cheap test); may be null
    uint32_t* status;             // device status word (sy_env_bind_status); nullptr = failures are not reported
};
#define SY_POLICY_SLICE 2304      // per-episode LDS scratch of the in-kernel policy: 8 hidden vectors of 64 floats + slots

// ---- the fused rollout: which kernel instance a configuration runs on (sy_dispatch.hip) ------------------------------
// One place decides it — the launcher, sy_env_set_policy's limits and sy_env_rollout_kernel_name all read this plan.
// ELL columns per agent of the half-wave neighbour scan (one episode per half wave: 32 lanes / agents)
__host__ __device__ constexpr int half_scan_gw(int P) { return 32 / (P + 1) > SY_ELL_WIDTH ? SY_ELL_WIDTH : 32 / (P + 1); }

struct RolloutPlan {
    int family;        // 3 = rollout3_kernel (move / helper pipeline), 2 = rollout2_kernel (paired move waves + belief
                       // waves), 1 = rollout_kernel (one episode per move wave, odd block sizes)
    int nr;            // belief slabs of 64 nodes the instance is compiled for: 1, 2, 4, 8, 16
    int pt;            // police count fixed at compile time (2, 4, 5, 6) or 0 = generic
    int hs;            // rollout3: columns per lane of the half-wave neighbour scan, 0 = paired scan
    bool rec, pol;     // trajectory recorded; actions from the MAPPO actors (sy_env_set_policy)
    int threads;       // block size
    size_t lds;        // dynamic LDS bytes of the launch (board + episode slices + policy scratch)
    int pslice;        // per-episode LDS scratch of the in-kernel policy on this family (0 without a policy)
    int pcap;          // ... of which the entry list holds this many entries per episode
};
// `policy_hidden` > 0 asks for the plan WITH a policy of that hidden size even if none is set yet (sy_env_set_policy
// validates its argument against the instance that would run); 0 = use p.pw2 / p.pH as they are.
RolloutPlan plan_rollout(const EngineParams& p, bool record, int wpb, size_t lds_base, int policy_hidden = 0);
void rollout_plan_name(const RolloutPlan& pl, char* buf, size_t n);     // e.g. "sy::rollout3_kernel<4,true,4,false,2>"
int rollout_policy_slice(int family, int A, int hidden, int entry_cap);

hipError_t launch_rollout(const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, int wpb, size_t lds,
                          hipStream_t stream);
hipError_t launch_step(const EngineParams& p, const int32_t* actions, const sy_rollout_buffers& rec, int blocks, int wpb, size_t lds,
                       hipStream_t stream);
// instance groups (one translation unit each); false = the planned instance is not in that unit
#define SY_DECL_GROUP(name) bool name(const RolloutPlan& pl, const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, hipStream_t stream)
SY_DECL_GROUP(launch_r3_a); SY_DECL_GROUP(launch_r3_b); SY_DECL_GROUP(launch_r3_c); SY_DECL_GROUP(launch_r3_d); SY_DECL_GROUP(launch_r3_p); SY_DECL_GROUP(launch_r3_q);
SY_DECL_GROUP(launch_r2_a); SY_DECL_GROUP(launch_r2_b); SY_DECL_GROUP(launch_r2_c);
SY_DECL_GROUP(launch_r1_a); SY_DECL_GROUP(launch_r1_b);
#undef SY_DECL_GROUP
hipError_t launch_reset(const EngineParams& p, const uint8_t* env_sel, const int32_t* starts, int zero_count, int blocks,
                        int wpb, size_t lds, hipStream_t stream);
hipError_t launch_action_mask_dense(const double* adj, const double* wts, const double* tolls, int N, const int32_t* cur,
                                    const double* budget, int Q, uint8_t* mask, hipStream_t stream);
hipError_t launch_belief_update(const uint32_t* ell, const float* inv_deg, int N, int NS, float* belief, const int32_t* hint,
                                int H, const int32_t* reveal, int Q, hipStream_t stream);

hipError_t launch_apsp(const uint32_t* ell, int N, int G, uint16_t* apsp, hipStream_t stream);

hipError_t launch_sample_boards(int N, int NS, int E_target, int max_deg_extra, uint64_t seed, int G, uint32_t* ell,
                                float* inv_deg, int32_t* edge_links, int32_t* edge_w, int32_t* num_edges, int E_cap,
                                hipStream_t stream);

hipError_t launch_masked_sample(const float* probs, long long probs_stride, const uint8_t* mask, long long mask_stride,
                                int rows, int N, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                int default_on_empty, int32_t* action, float* log_prob, float* norm_out, hipStream_t stream);

// returns / advantages over a [T][B][A] rollout (sy_returns_advantages)
struct ReturnsArgs {
    int32_t T, B, A, mode, reward_f64, done_bytes, compute_f64;
    const void* reward; long long rs_t, rs_b;
    const void* done_a; const void* done_b; long long ds_t, ds_b;
    const float* value; long long vs_t, vs_b, vs_a;
    const float* last_value; long long lv_b, lv_a;
    double gamma, lam;
    void* returns; void* adv;
};
hipError_t launch_returns(const ReturnsArgs& a, hipStream_t stream);

hipError_t launch_mappo_policy(const int32_t* pos, const uint8_t* mask, long long mask_row_stride, const float* w1t,
                               const float* b1, const float* w2t, const float* b2, const float* c1t, const float* cb1,
                               const float* c2, const float* cb2, int B, int A, int N, int H, uint64_t seed, uint64_t offset,
                               const uint64_t* offset_dev, int32_t* action, float* log_prob, float* value, float* probs_out,
                               hipStream_t stream);

// one PPO minibatch of the MAPPO networks: loss + gradient (sy_ppo.hip, sy_mappo_ppo_grad)
#define SY_PPO_MAX_ROLES 64              // 2 (A + 1) networks' tables x row ranges
#define SY_PPO_MAX_BLOCKS_PER_ROLE 96    // sizes the scratch of partial tables
struct PpoGrid { int32_t parts, rpp, nroles; uint16_t first[SY_PPO_MAX_ROLES + 1]; };
struct PpoPackArgs {     // rows of a rollout record -> the minibatch image (sy_ppo_pack)
    const int32_t* record; int32_t RW;
    const float* log_prob; const float* adv; const float* team_ret;
    const int32_t* rows; int32_t row0; long long count; int32_t B;
    const int32_t* env_graph;
    int32_t A;
    void* image;
    long long chunk_rows, record_chunk_stride, log_prob_chunk_stride;   // chunk_rows > 0: record / log_prob are chunks of that many rows, strides in elements
    long long shuffle_domain; int32_t shuffle_hb; uint64_t shuffle_seed;    // rows == nullptr, domain > 0: row0 + permutation(i) of [0, domain)
};
struct PpoArgs {
    const void* image; long long image_rows;       // the packed rows of an update (sy_ppo_pack)
    int32_t row0; const int32_t* row0_dev; int32_t mb;   // this minibatch: image rows row0 .. row0 + mb - 1
    const uint32_t* ell;
    int32_t A, N, H;
    const float* params;                            // [A + 1][slab]: every network in the slab layout (include/sy_env.h)
    float clip, value_coef;
    float* partial;
    int32_t* adam_step;
    int32_t DN, slab, parts, rpp;     // filled by the launcher (parts: row ranges a table is cut into; rpp: rows per part)
    int32_t nroles;
    uint16_t first[SY_PPO_MAX_ROLES + 1];    // first block of every role (a role = one table part of one network); [nroles] = grid size
};
size_t ppo_image_size(int A, long long rows);
hipError_t launch_ppo_pack(const PpoPackArgs& a, hipStream_t stream);
int ppo_slab_floats(int N, int H);
int ppo_parts(int N, int H);
int ppo_max_blocks_per_role();
struct PpoAdam {          // params == nullptr: no optimiser step
    float* params; float* m; float* v; int32_t* step;
    float lr, beta1, beta2, eps;
};
hipError_t launch_ppo_grad(PpoArgs a, float* grads, const PpoAdam& adam, hipStream_t stream);
hipError_t launch_ppo_adam(const float* grads, const PpoAdam& ad, int A, int N, int H, hipStream_t stream);

// the GNN Q-policy (sy_gnn.hip)
int gnn_padded_features(int F);
int gnn_param_floats(int F);
hipError_t launch_gnn_q_act(const int32_t* pos, const float* belief, long long belief_stride, const uint8_t* mask,
                            long long mask_row_stride, const uint32_t* tab, int K, const float* selfc, const int32_t* env_graph,
                            const float* models, int B, int A, int N, int F, float explore, uint64_t seed, uint64_t offset,
                            const uint64_t* offset_dev, int32_t* action, float* q_out, hipStream_t stream);

}  // namespace sy
