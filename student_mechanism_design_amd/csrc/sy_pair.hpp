// sy_pair.hpp — what the PAIRED rollout kernels share (two episodes per move wave: rollout2_kernel and the
// rollout3 pipeline): the lockstep belief filter of round 1, the single-pass paired scan with LDS result slots,
// the in-kernel learned policy on the paired scan, half-wave broadcasts, start sampling for a pair, the
// paired shaped reward and the diagnostic stamp macros.
#pragma once
#include "sy_device.hpp"

namespace sy {

// ---------------------------------------------------------------------------------------------
// Belief filter for the two episodes of a pair in lockstep (same board, same step): the scratch
// holds both episodes' b / deg interleaved (8 B per node), so one 8-byte LDS gather serves both
// and the sums are packed two-wide.  Per component the arithmetic and its order are exactly
// belief_step's.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NR>
__device__ __forceinline__ void belief_step_pair(v2f (&b)[NR], const float (&ideg)[NR], const int (&slab_w)[NR],
                                                 uint32_t c_off, const uint16_t* boff_s, int lane, int N, bool police_ev,
                                                 const int (&pol0)[SY_MAX_AGENTS - 1], const int (&pol1)[SY_MAX_AGENTS - 1],
                                                 int P, float uni) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        // slabs below N / 64 are full (a scalar test); NR may be rounded up past the last partial slab
        if (r < (N >> 6) || j < N) *lds_at<v2f>(c_off + (uint32_t)j * 8u) = b[r] * ideg[r];
    }
    if (lane == 0) *lds_at<v2f>(c_off + (uint32_t)N * 8u) = (v2f){0.0f, 0.0f};  // padding entries point here
    wave_lds_fence();
    auto ld = [c_off](uint32_t off) { return *lds_at<v2f>(c_off + off); };
    constexpr int GR = NR < 2 ? NR : 2;     // slabs pipelined together (register budget: 8 two-wide gathers each)
    v2f tot = {0.0f, 0.0f};
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += GR) {
        uint4 o[GR];
        int jr[GR];
#pragma unroll
        for (int q = 0; q < GR; ++q) {
            const int j = lane + 64 * (r0 + q);
            jr[q] = j < N ? j : N - 1;          // tail lanes read a valid row; their result is discarded
            o[q] = r0 + q < NR ? *reinterpret_cast<const uint4*>(boff_s + (jr[q] << 4)) : make_uint4(0, 0, 0, 0);
        }
        v2f g[GR][8];
#pragma unroll
        for (int q = 0; q < GR; ++q) {
            if (r0 + q < NR) {
                g[q][0] = ld(o[q].x & 0xffffu); g[q][1] = ld(o[q].x >> 16);
                g[q][2] = ld(o[q].y & 0xffffu); g[q][3] = ld(o[q].y >> 16);
                g[q][4] = ld(o[q].z & 0xffffu); g[q][5] = ld(o[q].z >> 16);
                g[q][6] = ld(o[q].w & 0xffffu); g[q][7] = ld(o[q].w >> 16);
            }
        }
#pragma unroll
        for (int q = 0; q < GR; ++q) {
            const int r = r0 + q;
            if (r < NR) {
                const int j = lane + 64 * r;
                v2f acc = ideg[r] == 0.0f ? b[r] : (v2f){0.0f, 0.0f};
                acc += ((g[q][0] + g[q][1]) + (g[q][2] + g[q][3])) + ((g[q][4] + g[q][5]) + (g[q][6] + g[q][7]));
                if (slab_w[r] > 2) {            // wave-uniform: some row of this slab has more than 8 neighbours
                    const uint4 o2 = *reinterpret_cast<const uint4*>(boff_s + (jr[q] << 4) + 8);
                    const v2f h0 = ld(o2.x & 0xffffu), h1 = ld(o2.x >> 16), h2 = ld(o2.y & 0xffffu), h3 = ld(o2.y >> 16);
                    const v2f h4 = ld(o2.z & 0xffffu), h5 = ld(o2.z >> 16), h6 = ld(o2.w & 0xffffu), h7 = ld(o2.w >> 16);
                    acc += ((h0 + h1) + (h2 + h3)) + ((h4 + h5) + (h6 + h7));
                }
                if (police_ev) {
#pragma unroll
                    for (int k = 0; k < SY_MAX_AGENTS - 1; ++k) {
                        if (k < P && j == pol0[k]) acc.x = 0.0f;
                        if (k < P && j == pol1[k]) acc.y = 0.0f;
                    }
                }
                acc = j < N ? acc : (v2f){0.0f, 0.0f};
                b[r] = acc;
                tot += acc;
            }
        }
    }
    const float t0 = wave_sum(tot.x), t1 = wave_sum(tot.y);
    // b * (1 / total), or the uniform distribution when the mass vanished: one fused multiply-add with
    // per-episode uniform (scale, offset) = (1/t, 0) or (0, 1/N); x * s + 0 rounds exactly like x * s
    const v2f scale = {t0 == 0.0f ? 0.0f : __builtin_amdgcn_rcpf(t0), t1 == 0.0f ? 0.0f : __builtin_amdgcn_rcpf(t1)};
    const v2f offs = {t0 == 0.0f ? uni : 0.0f, t1 == 0.0f ? uni : 0.0f};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        b[r] = __builtin_elementwise_fma(b[r], scale, offs);
        if (r >= (N >> 6)) {    // wave-uniform: only slabs from N / 64 on have lanes past the last node
            const bool in = lane + 64 * r < N;
            b[r].x = in ? b[r].x : 0.0f;
            b[r].y = in ? b[r].y : 0.0f;
        }
    }
    wave_lds_fence();
}

template <int NR, bool REC>
__device__ __forceinline__ void belief_pair_run(const EngineParams& p, const LdsMap& L, const EnvLds& E, const EnvLds& E1,
                                                int lane, int e, int g, int P, int T, sy_rollout_buffers out) {
    const int N = p.N, NS = p.NS, B = p.B;
    const bool live1 = e + 1 < B;
    v2f b[NR];
    float ideg[NR], b0[NR];
    int slab_w[NR];
    belief_load<NR>(b0, ideg, slab_w, p.st.belief + (size_t)e * NS, p.inv_deg + (size_t)g * NS, lane, N);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        b[r].x = b0[r];
        b[r].y = (live1 && j < N) ? p.st.belief[(size_t)(e + 1) * NS + j] : 0.0f;
    }
    const uint32_t c_off = lds_off(E.c_s);
    const uint32_t off_bel = ((uint32_t)e * (uint32_t)NS + (uint32_t)lane) * 4u;
    const bool rec_bel = REC && out.belief != nullptr;
    const bool onehot = p.belief_onehot != 0, pol_ev = p.police_ev != 0;
    const float uni = 1.0f / (float)N;
    for (int s = 0; s < T; ++s) {
        if (rec_bel) {
            float* row0 = at_bytes(out.belief, off_bel);
            float* row1 = row0 + NS;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (r < (NS >> 6) || lane + 64 * r < NS) {  // slabs below NS / 64 are full (a scalar test)
                    row0[64 * r] = b[r].x;
                    if (live1) row1[64 * r] = b[r].y;
                }
            }
        }
        // both ring entries of step s are published by one instruction of the pair's move wave
        {
            int spin = 0;
            for (; lds_peek(E.sync) <= s && spin < kSpinMax; ++spin) __builtin_amdgcn_s_sleep(2);
            if (spin == kSpinMax) report_status(SY_STATUS_BELIEF_WAIT_EXPIRED);
        }
        asm volatile("" ::: "memory");
        const int so = (s & (kRing - 1)) * 8;
        const int head0 = __builtin_amdgcn_readfirstlane(E.ring[so]), head1 = __builtin_amdgcn_readfirstlane(E1.ring[so]);
        int pol0[SY_MAX_AGENTS - 1], pol1[SY_MAX_AGENTS - 1];
#pragma unroll
        for (int k = 0; k < SY_MAX_AGENTS - 1; ++k) pol0[k] = pol1[k] = -1;
        if (pol_ev) {
#pragma unroll
            for (int k = 0; k < SY_MAX_AGENTS - 1; ++k) {
                pol0[k] = __builtin_amdgcn_readfirstlane(E.ring[so + 1 + k]);
                pol1[k] = __builtin_amdgcn_readfirstlane(E1.ring[so + 1 + k]);
            }
        }
        asm volatile("" ::: "memory");
        if (lane < 2) lds_poke(lane == 0 ? E.sync + 1 : E1.sync + 1, s + 1);   // entries copied: the slots may be reused
        const int node0 = head0 & 0xffff, flags0 = head0 >> 16, node1 = head1 & 0xffff, flags1 = head1 >> 16;
        // the filter runs in place for both; an episode that restarts or reveals is overwritten below
        if (((flags0 & 3) == 0) || ((flags1 & 3) == 0))
            belief_step_pair<NR>(b, ideg, slab_w, c_off, L.boff_s, lane, N, pol_ev, pol0, pol1, P, uni);
        if (flags0 & 3) {   // new episode -> prior, reveal -> delta (wave-uniform branches)
            const bool delta = (flags0 & 2) || onehot;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int j = lane + 64 * r;
                b[r].x = j < N ? (delta ? (j == node0 ? 1.0f : 0.0f) : uni) : 0.0f;
            }
        }
        if (flags1 & 3) {
            const bool delta = (flags1 & 2) || onehot;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int j = lane + 64 * r;
                b[r].y = j < N ? (delta ? (j == node1 ? 1.0f : 0.0f) : uni) : 0.0f;
            }
        }
        if (rec_bel) out.belief += (size_t)B * NS;
    }
    float* bel_out = kernarg_params()->st.belief;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        if (j < NS) {
            bel_out[(size_t)e * NS + j] = b[r].x;
            if (live1) bel_out[(size_t)(e + 1) * NS + j] = b[r].y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Single-pass form of scan_eval_pair (all A agents fit one pass — the common case).  Vector work is
// cut to the primitive compares: the ballots are combined as scalar masks; instead of clearing the
// whole mask rows, every scan lane clears the one byte it set on the previous step; the chosen
// entry and the position-reward count reach the agent lanes through a two-word LDS slot per agent
// (words 48.. of the record staging row, never stored) instead of ballot shifts and a bpermute.
struct PairScanLane {        // per-lane constants (LDS byte offsets) + the two carried "previous byte" offsets
    uint32_t row0, row1;     // mask row of my group's agent, episode 0 / 1
    uint32_t selw0, selw1;   // my group's slot, episode 0 / 1 (scan-lane role)
    uint32_t selr;           // slot of agent (lane & 7) of my half (agent-lane role)
    uint32_t prev0, prev1;
    uint32_t scratch;        // a word nobody reads (record staging row, word kDummyWord of my half)
    uint64_t on_m, lead_m;   // lanes scanning a real agent; the first lane of each such group
};
static constexpr int kSelWord = 48, kDummyWord = 47;
static constexpr uint64_t kAgentSlots = 0x000000ff000000ffull;   // lanes 0..7 of both halves

__device__ __forceinline__ PairScanLane make_pair_scan_lane(const EnvLds& E, const EnvLds& E1, const ScanMap& sm, int lane,
                                                            int A, int NS, int base = 0) {
    PairScanLane q;
    const int ag = base + sm.grp;              // the agent this lane's group scans in the pass starting at `base`
    const bool on = sm.live && ag < A;
    q.row0 = lds_off(E.mrow) + (uint32_t)(ag * NS);
    q.row1 = lds_off(E1.mrow) + (uint32_t)(ag * NS);
    q.selw0 = lds_off(E.rec_s) + (uint32_t)(kSelWord + 2 * (ag & 7)) * 4u;
    q.selw1 = lds_off(E1.rec_s) + (uint32_t)(kSelWord + 2 * (ag & 7)) * 4u;
    const uint32_t rec_h = lane >= 32 ? lds_off(E1.rec_s) : lds_off(E.rec_s);
    q.selr = rec_h + (uint32_t)(kSelWord + 2 * (lane & 7)) * 4u;
    q.prev0 = lds_off(E.rec_s) + kDummyWord * 4u;
    q.prev1 = lds_off(E1.rec_s) + kDummyWord * 4u;
    q.scratch = rec_h + kDummyWord * 4u;
    q.on_m = bal(on);
    q.lead_m = bal(on && sm.col == 0);
    return q;
}

// BEGIN / END: the first pass of a step writes the "no move" defaults, the last one reads the slots back
// (two passes when the agents do not fit one: e.g. P = 6 with rows wider than 9).
template <bool BEGIN = true, bool END = true>
__device__ __forceinline__ void scan_eval_pair1(PairScanLane& q, const ScanMap& sm, int gw, const ScanPairIn& g, int& act_v,
                                                int& cost_v, int& quirk_cnt) {
    SY_HOT(m_eval);
    if (BEGIN && lanes(kAgentSlots)) *lds_at<uint64_t>(q.selr) = 0x0000ffffull;   // "no move": action -1, cost 0
    *lds_at<uint8_t>(q.prev0) = 0;
    *lds_at<uint8_t>(q.prev1) = 0;
    const uint32_t fmask = (1u << gw) - 1u;
    const int w0 = (int)(g.ent0 >> 16), w1 = (int)(g.ent1 >> 16);
    const uint64_t bo0 = bal(w0 <= g.ma0) & q.on_m, bo1 = bal(w1 <= g.ma1) & q.on_m;
    const uint64_t bq0 = bal(w0 <= g.mq0) & q.on_m, bq1 = bal(w1 <= g.mq1) & q.on_m;
    const bool own0 = lanes(bo0), own1 = lanes(bo1);
    // lanes without an affordable entry write the scratch word instead of being masked off: a select is
    // cheaper than saving / restoring exec around every store
    const uint32_t n0 = own0 ? q.row0 + (g.ent0 & 0xffffu) : q.scratch, n1 = own1 ? q.row1 + (g.ent1 & 0xffffu) : q.scratch;
    *lds_at<uint8_t>(n0) = 1;
    *lds_at<uint8_t>(n1) = 1;
    q.prev0 = n0;      // the scratch byte is cleared like any other on the next step
    q.prev1 = n1;
    const uint32_t gf0 = (uint32_t)(bo0 >> sm.gsh) & fmask, gf1 = (uint32_t)(bo1 >> sm.gsh) & fmask;
    const int rr0 = (int)__umulhi(g.xa0, (uint32_t)__popc(gf0)), rr1 = (int)__umulhi(g.xa1, (uint32_t)__popc(gf1));
    const uint64_t ch0 = bal((int)__popc(gf0 & sm.lowmask) == rr0) & bo0, ch1 = bal((int)__popc(gf1 & sm.lowmask) == rr1) & bo1;
    *lds_at<int>(lanes(ch0) ? q.selw0 : q.scratch) = (int)g.ent0;
    *lds_at<int>(lanes(ch1) ? q.selw1 : q.scratch) = (int)g.ent1;
    if (lanes(q.lead_m)) {
        lds_at<int>(q.selw0)[1] = __popc((uint32_t)(bq0 >> sm.gsh) & fmask);
        lds_at<int>(q.selw1)[1] = __popc((uint32_t)(bq1 >> sm.gsh) & fmask);
    }
    wave_lds_fence();
    if (END) {
        const uint64_t r = *lds_at<uint64_t>(q.selr);
        act_v = (int)(int16_t)(uint32_t)r;              // 0xffff -> -1
        cost_v = (int)(((uint32_t)r) >> 16);
        quirk_cnt = (int)(r >> 32);
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// In-kernel learned policy (sy_env_set_policy): the rollout loop of mappo_trainer.py:161-287 with
// MappoAgent.select_action inside the fused kernel.  Only the logits of an agent's affordable
// neighbours are needed (softmax over the legal actions == the reference's masked, renormalised
// softmax), so per step and agent: hidden = relu(b1 + row lookups in w1t) (64 floats, lane = hidden
// unit, kept in LDS), one 64-term dot product per scan lane against that neighbour's row of w2, a
// Gumbel-max draw and a log-sum-exp over the group through three LDS slots.
// Per-episode LDS scratch (SY_POLICY_SLICE): [8 agents][64] hidden floats, then 8 x {max key, max logit,
// sum exp, log-prob of the winner}.
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t kPolSlots = 8 * 64 * 4;     // byte offset of the slots inside the policy scratch
__device__ __forceinline__ int f32_ordered(float f) {           // monotone float -> int map (for integer max)
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_f32(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

// hidden vectors of both episodes' next observation (mappo_trainer.py:173,197: one-hot MrX node for MrX's actor,
// multi-hot police nodes for the police actors); lane = hidden unit
__device__ __forceinline__ void policy_hidden_pair(const EngineParams& p, int P, int pos_n, int lane, uint32_t pol0,
                                                   uint32_t pol1) {
    const int H = p.pH, N = p.N;
    const bool hk = lane < H;
    const int k = hk ? lane : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t pb = h ? pol1 : pol0;
        int pj[SY_MAX_AGENTS];
#pragma unroll
        for (int j = 0; j < SY_MAX_AGENTS; ++j) pj[j] = j <= P ? rdlane(pos_n, 32 * h + j) : 0;
        float v = p.pb1[k] + p.pw1t[(size_t)pj[0] * H + k];
        if (hk) *lds_at<float>(pb + 4u * (uint32_t)lane) = v > 0.0f ? v : 0.0f;
#pragma unroll
        for (int a = 1; a < SY_MAX_AGENTS; ++a) {
            if (a <= P) {
                const float* w1a = p.pw1t + (size_t)a * N * H;
                float u = p.pb1[a * H + k];
#pragma unroll
                for (int j = 1; j < SY_MAX_AGENTS; ++j)
                    if (j <= P) u += w1a[(size_t)pj[j] * H + k];
                if (hk) *lds_at<float>(pb + 256u * (uint32_t)a + 4u * (uint32_t)lane) = u > 0.0f ? u : 0.0f;
            }
        }
    }
}

struct PolicyLane {          // per-lane constants of the policy scan
    uint32_t hs0, hs1;       // my group's agent's hidden vector, episode 0 / 1
    uint32_t sl0, sl1;       // my group's agent's slots, episode 0 / 1
    uint32_t slr;            // slots of agent (lane & 7) of my half (agent-lane role)
    const float* w2a;        // my group's agent's second layer [N][H]
    const float* b2a;
};
__device__ __forceinline__ PolicyLane make_policy_lane(const EngineParams& p, const ScanMap& sm, int lane, int A, uint32_t pol0,
                                                       uint32_t pol1) {
    PolicyLane q;
    const int ag = (sm.live && sm.grp < A) ? sm.grp : 0;
    q.hs0 = pol0 + 256u * (uint32_t)ag;
    q.hs1 = pol1 + 256u * (uint32_t)ag;
    q.sl0 = pol0 + kPolSlots + 16u * (uint32_t)ag;
    q.sl1 = pol1 + kPolSlots + 16u * (uint32_t)ag;
    q.slr = (lane >= 32 ? pol1 : pol0) + kPolSlots + 16u * (uint32_t)(lane & 7);
    q.w2a = p.pw2 + (size_t)ag * p.N * p.pH;
    q.b2a = p.pb2 + (size_t)ag * p.N;
    return q;
}

// scan_eval_pair1 with the learned policy choosing the action (single pass)
__device__ __forceinline__ void scan_eval_pair_policy(PairScanLane& q, const PolicyLane& pl, const ScanMap& sm, int gw, int H,
                                                      const ScanPairIn& g, int& act_v, int& cost_v, int& quirk_cnt,
                                                      float& logp_v) {
    if (lanes(kAgentSlots)) {
        *lds_at<uint64_t>(q.selr) = 0x0000ffffull;                                   // "no move": action -1, cost 0
        typedef int v4i __attribute__((ext_vector_type(4)));
        *lds_at<v4i>(pl.slr) = (v4i){(int)0x80000000, (int)0x80000000, 0, 0};         // max key, max logit, sum exp, log-prob
    }
    *lds_at<uint8_t>(q.prev0) = 0;
    *lds_at<uint8_t>(q.prev1) = 0;
    const uint32_t fmask = (1u << gw) - 1u;
    const int w0 = (int)(g.ent0 >> 16), w1 = (int)(g.ent1 >> 16);
    const uint64_t bo0 = bal(w0 <= g.ma0) & q.on_m, bo1 = bal(w1 <= g.ma1) & q.on_m;
    const uint64_t bq0 = bal(w0 <= g.mq0) & q.on_m, bq1 = bal(w1 <= g.mq1) & q.on_m;
    const bool own0 = lanes(bo0), own1 = lanes(bo1);
    const uint32_t nb0 = own0 ? (g.ent0 & 0xffffu) : 0u, nb1 = own1 ? (g.ent1 & 0xffffu) : 0u;
    const uint32_t n0 = own0 ? q.row0 + nb0 : q.scratch, n1 = own1 ? q.row1 + nb1 : q.scratch;
    *lds_at<uint8_t>(n0) = 1;
    *lds_at<uint8_t>(n1) = 1;
    q.prev0 = n0;
    q.prev1 = n1;
    // one logit per affordable entry: the neighbour's row of w2 (L2) against the agent's hidden vector (LDS)
    float l0 = pl.b2a[nb0], l1 = pl.b2a[nb1];
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* r0 = reinterpret_cast<const f4*>(pl.w2a + (size_t)nb0 * H);
        const f4* r1 = reinterpret_cast<const f4*>(pl.w2a + (size_t)nb1 * H);
        int c = 0;
        for (; c + 2 <= (H >> 2); c += 2) {          // two 16-byte chunks of both rows per round trip to L2
            const f4 a0 = r0[c], a1 = r1[c], b0 = r0[c + 1], b1 = r1[c + 1];
            const f4 h0 = *lds_at<f4>(pl.hs0 + 16u * (uint32_t)c), h1 = *lds_at<f4>(pl.hs1 + 16u * (uint32_t)c);
            l0 = fmaf(a0.x, h0.x, l0); l0 = fmaf(a0.y, h0.y, l0); l0 = fmaf(a0.z, h0.z, l0); l0 = fmaf(a0.w, h0.w, l0);
            l1 = fmaf(a1.x, h1.x, l1); l1 = fmaf(a1.y, h1.y, l1); l1 = fmaf(a1.z, h1.z, l1); l1 = fmaf(a1.w, h1.w, l1);
            const f4 g0 = *lds_at<f4>(pl.hs0 + 16u * (uint32_t)(c + 1)), g1 = *lds_at<f4>(pl.hs1 + 16u * (uint32_t)(c + 1));
            l0 = fmaf(b0.x, g0.x, l0); l0 = fmaf(b0.y, g0.y, l0); l0 = fmaf(b0.z, g0.z, l0); l0 = fmaf(b0.w, g0.w, l0);
            l1 = fmaf(b1.x, g1.x, l1); l1 = fmaf(b1.y, g1.y, l1); l1 = fmaf(b1.z, g1.z, l1); l1 = fmaf(b1.w, g1.w, l1);
        }
        for (; c < (H >> 2); ++c) {
            const f4 a0 = r0[c], a1 = r1[c];
            const f4 h0 = *lds_at<f4>(pl.hs0 + 16u * (uint32_t)c), h1 = *lds_at<f4>(pl.hs1 + 16u * (uint32_t)c);
            l0 = fmaf(a0.x, h0.x, l0); l0 = fmaf(a0.y, h0.y, l0); l0 = fmaf(a0.z, h0.z, l0); l0 = fmaf(a0.w, h0.w, l0);
            l1 = fmaf(a1.x, h1.x, l1); l1 = fmaf(a1.y, h1.y, l1); l1 = fmaf(a1.z, h1.z, l1); l1 = fmaf(a1.w, h1.w, l1);
        }
    }
    if (own0) atomicMax(lds_at_generic<int>(pl.sl0 + 4u), f32_ordered(l0));
    if (own1) atomicMax(lds_at_generic<int>(pl.sl1 + 4u), f32_ordered(l1));
    wave_lds_fence();
    const float L0 = ordered_f32(*lds_at<int>(pl.sl0 + 4u)), L1 = ordered_f32(*lds_at<int>(pl.sl1 + 4u));
    // Gumbel-max draw: a cheap per-lane hash of the agent's Philox word of this step
    auto gumbel = [&sm](uint32_t x) {
        uint32_t h = x ^ ((uint32_t)sm.col * 0x9E3779B9u) ^ 0x85EBCA6Bu;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const float u = fminf(((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f), 0x1.fffffep-1f);   // (h >> 8 = 2^24 - 1 rounds to 1.0f: -log(-log 1) = +inf)
        return -__logf(-__logf(u));
    };
    const float k0 = l0 + gumbel(g.xa0), k1 = l1 + gumbel(g.xa1);
    if (own0) {
        atomicAdd(lds_at_generic<float>(pl.sl0 + 8u), __expf(l0 - L0));
        atomicMax(lds_at_generic<int>(pl.sl0), f32_ordered(k0));
    }
    if (own1) {
        atomicAdd(lds_at_generic<float>(pl.sl1 + 8u), __expf(l1 - L1));
        atomicMax(lds_at_generic<int>(pl.sl1), f32_ordered(k1));
    }
    wave_lds_fence();
    const bool win0 = own0 && f32_ordered(k0) == *lds_at<int>(pl.sl0), win1 = own1 && f32_ordered(k1) == *lds_at<int>(pl.sl1);
    const float S0 = *lds_at<float>(pl.sl0 + 8u), S1 = *lds_at<float>(pl.sl1 + 8u);
    if (win0) {
        *lds_at<int>(q.selw0) = (int)g.ent0;
        *lds_at<float>(pl.sl0 + 12u) = (l0 - L0) - __logf(S0);
    }
    if (win1) {
        *lds_at<int>(q.selw1) = (int)g.ent1;
        *lds_at<float>(pl.sl1 + 12u) = (l1 - L1) - __logf(S1);
    }
    if (lanes(q.lead_m)) {
        lds_at<int>(q.selw0)[1] = __popc((uint32_t)(bq0 >> sm.gsh) & fmask);
        lds_at<int>(q.selw1)[1] = __popc((uint32_t)(bq1 >> sm.gsh) & fmask);
    }
    wave_lds_fence();
    const uint64_t r = *lds_at<uint64_t>(q.selr);
    act_v = (int)(int16_t)(uint32_t)r;              // 0xffff -> -1
    cost_v = (int)(((uint32_t)r) >> 16);
    quirk_cnt = (int)(r >> 32);
    logp_v = *lds_at<float>(pl.slr + 12u);
    wave_lds_fence();
}

// ---- the learned policy inside the pipeline (sy_env_set_policy) -----------------------------------
// The move wave evaluates the MAPPO actors for the next observation (mappo_agent.py:87-142 inside the rollout loop
// of mappo_trainer.py:161-287); everything that reads the board for the RECORD (visit counters, shortest paths,
// rewards) moves to the helper wave, so the move wave has the registers to keep a whole second-layer row per lane in
// flight.  Per step and agent: hidden = relu(b1 + row lookups in w1t) (lane = hidden unit; up to 128 units, kept in
// LDS), one H-term dot product per scan lane against its neighbour's row of w2 (two 16-byte pieces of both
// episodes' rows per round trip to L2), a Gumbel-max draw and a log-sum-exp over the agent's lane group through LDS
// slots: the softmax over the legal actions == the reference's masked, renormalised softmax.
// The reference's underflow rule (mappo_agent.py:123-134: if the legal actions hold <= 1e-8 of the softmax mass, the
// action is drawn uniformly over the mask) needs the mass of ALL nodes.  `bound[a]` (host, refreshed with the weights)
// is an upper bound of any logit of actor a over all observations; while
//     logsumexp(legal) > log(1e-8) + log(N) + bound[a]
// the legal mass provably exceeds 1e-8 and nothing else is evaluated; otherwise (rare) the wave evaluates actor a's
// N logits for that episode exactly and applies the reference's rule.
// Per-episode LDS scratch: [A][H] hidden floats, then 8 x {max key, max logit, sum exp, log-prob of the winner}.
struct PolLane3 {
    uint32_t hs0, hs1;       // my group's agent's hidden vector, episode 0 / 1
    uint32_t sl0, sl1;       // my group's agent's slots, episode 0 / 1
    uint32_t slr;            // slots of agent (lane & 7) of my half (agent-lane role)
    const float* w2a;        // my group's agent's second layer [N][H]
    const float* b2a;
    float thr;               // log(1e-8) + log(N) + bound[agent]   (+inf without a bound: never the exact path)
    int ag;
};
__device__ __forceinline__ PolLane3 make_pol_lane3(const EngineParams& p, const ScanMap& sm, int lane, int A, uint32_t pol0,
                                                   uint32_t pol1) {
    PolLane3 q;
    const int H = p.pH;
    q.ag = (sm.live && sm.grp < A) ? sm.grp : 0;
    q.hs0 = pol0 + (uint32_t)(q.ag * H) * 4u;
    q.hs1 = pol1 + (uint32_t)(q.ag * H) * 4u;
    const uint32_t slots = (uint32_t)(A * H) * 4u;
    q.sl0 = pol0 + slots + 16u * (uint32_t)q.ag;
    q.sl1 = pol1 + slots + 16u * (uint32_t)q.ag;
    q.slr = (lane >= 32 ? pol1 : pol0) + slots + 16u * (uint32_t)(lane & 7);
    q.w2a = p.pw2 + (size_t)q.ag * p.N * H;
    q.b2a = p.pb2 + (size_t)q.ag * p.N;
    q.thr = p.pbound ? (-18.420680744f + __logf((float)p.N) + p.pbound[q.ag]) : -3.0e38f;
    return q;
}
__device__ __forceinline__ void policy_hidden_pair3(const EngineParams& p, int P, int A, int pos_n, int lane, uint32_t pol0,
                                                    uint32_t pol1) {
    // All row lookups of a half are issued back to back (1 + P * P rows of w1t and the A biases: every load is
    // independent) and summed afterwards: one round trip to L2 per half instead of one per row.
    const int H = p.pH, N = p.N;
    for (int k0 = 0; k0 < H; k0 += 64) {      // (one pass up to 64 hidden units, two for 128)
        const bool hk = k0 + lane < H;
        const int k = hk ? k0 + lane : 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t pb = h ? pol1 : pol0;
            int pj[SY_MAX_AGENTS];
#pragma unroll
            for (int j = 0; j < SY_MAX_AGENTS; ++j) pj[j] = j <= P ? rdlane(pos_n, 32 * h + j) : 0;
            float bias[SY_MAX_AGENTS], row[SY_MAX_AGENTS][SY_MAX_AGENTS];
#pragma unroll
            for (int a = 0; a < SY_MAX_AGENTS; ++a) {
                bias[a] = a <= P ? p.pb1[a * H + k] : 0.0f;
#pragma unroll
                for (int j = 0; j < SY_MAX_AGENTS; ++j) row[a][j] = 0.0f;
            }
            row[0][0] = p.pw1t[(size_t)pj[0] * H + k];                         // MrX's actor: one-hot MrX node
#pragma unroll
            for (int a = 1; a < SY_MAX_AGENTS; ++a) {
                if (a <= P) {
                    const float* w1a = p.pw1t + (size_t)a * N * H;             // police actors: multi-hot police nodes
#pragma unroll
                    for (int j = 1; j < SY_MAX_AGENTS; ++j)
                        if (j <= P) row[a][j] = w1a[(size_t)pj[j] * H + k];
                }
            }
            {
                const float v = bias[0] + row[0][0];
                if (hk) *lds_at<float>(pb + 4u * (uint32_t)k) = v > 0.0f ? v : 0.0f;
            }
#pragma unroll
            for (int a = 1; a < SY_MAX_AGENTS; ++a) {
                if (a <= P) {
                    float u = bias[a];
#pragma unroll
                    for (int j = 1; j < SY_MAX_AGENTS; ++j)
                        if (j <= P) u += row[a][j];                            // same order as the sequential sum
                    if (hk) *lds_at<float>(pb + 4u * (uint32_t)(a * H + k)) = u > 0.0f ? u : 0.0f;
                }
            }
        }
    }
}
// The exact softmax mass of the legal actions of one (episode, agent): all N logits of the actor on the wave
// (node = lane + 64 r), float32 like the reference's tensors.  legal_lse = logsumexp of the legal logits.
#ifdef SY_POL_EXACT_INLINE
#define SY_EXACT_ATTR __forceinline__
#else
#define SY_EXACT_ATTR __noinline__     // a call keeps the cold path's registers out of the step loop (3.15 vs 2.97 G agent-steps/s)
#endif
template <int NR>
__device__ SY_EXACT_ATTR float exact_legal_mass(const float* w2a, const float* b2a, uint32_t hs, int H, int N, int lane, float legal_lse) {
    float l[NR];
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = lane + 64 * r;
        const int nn = n < N ? n : N - 1;
        float acc = b2a[nn];
        const float* row = w2a + (size_t)nn * H;
        for (int k = 0; k < H; k += 4) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 wv = *reinterpret_cast<const f4*>(row + k);
            const f4 hv = *lds_at<f4>(hs + 4u * (uint32_t)k);
            acc = fmaf(wv.x, hv.x, acc); acc = fmaf(wv.y, hv.y, acc); acc = fmaf(wv.z, hv.z, acc); acc = fmaf(wv.w, hv.w, acc);
        }
        l[r] = n < N ? acc : -3.0e38f;
        mx = l[r] > mx ? l[r] : mx;
    }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) {
        const float om = __shfl_xor(mx, o2, kWave);
        mx = om > mx ? om : mx;
    }
    float z = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) z += (lane + 64 * r < N) ? __expf(l[r] - mx) : 0.0f;
    z = wave_sum(z);
    return __expf(legal_lse - mx) / z;
}

// scan_eval_pair1 with the learned policy choosing the action (single pass).
template <int NR>
__device__ __forceinline__ void scan_eval_pair_policy3(PairScanLane& q, const PolLane3& pl, const ScanMap& sm, int gw, int H, int N,
                                                       int lane, const float* w2_all, const float* b2_all, const ScanPairIn& g,
                                                       int& act_v, int& cost_v, int& quirk_cnt, float& logp_v) {
    if (lanes(kAgentSlots)) {
        *lds_at<uint64_t>(q.selr) = 0x0000ffffull;                                   // "no move": action -1, cost 0
        typedef int v4i __attribute__((ext_vector_type(4)));
        *lds_at<v4i>(pl.slr) = (v4i){(int)0x80000000, (int)0x80000000, 0, 0};         // max key, max logit, sum exp, log-prob
    }
    *lds_at<uint8_t>(q.prev0) = 0;
    *lds_at<uint8_t>(q.prev1) = 0;
    const uint32_t fmask = (1u << gw) - 1u;
    const int w0 = (int)(g.ent0 >> 16), w1 = (int)(g.ent1 >> 16);
    const uint64_t bo0 = bal(w0 <= g.ma0) & q.on_m, bo1 = bal(w1 <= g.ma1) & q.on_m;
    const uint64_t bq0 = bal(w0 <= g.mq0) & q.on_m, bq1 = bal(w1 <= g.mq1) & q.on_m;
    const bool own0 = lanes(bo0), own1 = lanes(bo1);
    const uint32_t nb0 = own0 ? (g.ent0 & 0xffffu) : 0u, nb1 = own1 ? (g.ent1 & 0xffffu) : 0u;
    const uint32_t n0 = own0 ? q.row0 + nb0 : q.scratch, n1 = own1 ? q.row1 + nb1 : q.scratch;
    *lds_at<uint8_t>(n0) = 1;
    *lds_at<uint8_t>(n1) = 1;
    q.prev0 = n0;
    q.prev1 = n1;
    // one logit per affordable entry: the neighbour's row of w2 (L2) against the agent's hidden vector (LDS); a few
    // 16-byte pieces of both rows are in flight per round trip
    float l0 = pl.b2a[nb0], l1 = pl.b2a[nb1];
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* r0 = reinterpret_cast<const f4*>(pl.w2a + (size_t)nb0 * H);
        const f4* r1 = reinterpret_cast<const f4*>(pl.w2a + (size_t)nb1 * H);
        const int nq = H >> 2;
        int c = 0;
#ifndef SY_POL_BATCH
#define SY_POL_BATCH 2      // measured (tools/policy_rollout_bench.py, H = 64): 1 -> 2.7, 2 -> 3.2, 4 -> 2.4, 8 -> 2.1 G agent-steps/s:
#endif                      // more pieces in flight cost more registers than the round trips they save
        for (; c + SY_POL_BATCH <= nq; c += SY_POL_BATCH) {
            f4 a[SY_POL_BATCH], b[SY_POL_BATCH];
#pragma unroll
            for (int u = 0; u < SY_POL_BATCH; ++u) { a[u] = r0[c + u]; b[u] = r1[c + u]; }
#pragma unroll
            for (int u = 0; u < SY_POL_BATCH; ++u) {
                const f4 h0 = *lds_at<f4>(pl.hs0 + 16u * (uint32_t)(c + u)), h1 = *lds_at<f4>(pl.hs1 + 16u * (uint32_t)(c + u));
                l0 = fmaf(a[u].x, h0.x, l0); l0 = fmaf(a[u].y, h0.y, l0); l0 = fmaf(a[u].z, h0.z, l0); l0 = fmaf(a[u].w, h0.w, l0);
                l1 = fmaf(b[u].x, h1.x, l1); l1 = fmaf(b[u].y, h1.y, l1); l1 = fmaf(b[u].z, h1.z, l1); l1 = fmaf(b[u].w, h1.w, l1);
            }
        }
        for (; c < nq; ++c) {
            const f4 a0 = r0[c], a1 = r1[c];
            const f4 h0 = *lds_at<f4>(pl.hs0 + 16u * (uint32_t)c), h1 = *lds_at<f4>(pl.hs1 + 16u * (uint32_t)c);
            l0 = fmaf(a0.x, h0.x, l0); l0 = fmaf(a0.y, h0.y, l0); l0 = fmaf(a0.z, h0.z, l0); l0 = fmaf(a0.w, h0.w, l0);
            l1 = fmaf(a1.x, h1.x, l1); l1 = fmaf(a1.y, h1.y, l1); l1 = fmaf(a1.z, h1.z, l1); l1 = fmaf(a1.w, h1.w, l1);
        }
    }
    if (own0) atomicMax(lds_at_generic<int>(pl.sl0 + 4u), f32_ordered(l0));
    if (own1) atomicMax(lds_at_generic<int>(pl.sl1 + 4u), f32_ordered(l1));
    wave_lds_fence();
    float L0 = ordered_f32(*lds_at<int>(pl.sl0 + 4u)), L1 = ordered_f32(*lds_at<int>(pl.sl1 + 4u));
    if (own0) atomicAdd(lds_at_generic<float>(pl.sl0 + 8u), __expf(l0 - L0));
    if (own1) atomicAdd(lds_at_generic<float>(pl.sl1 + 8u), __expf(l1 - L1));
    wave_lds_fence();
    float S0 = *lds_at<float>(pl.sl0 + 8u), S1 = *lds_at<float>(pl.sl1 + 8u);
    // ---- the reference's underflow rule (mappo_agent.py:123-134), exact only where the cheap bound cannot rule it out
    {
        const bool lead = lanes(q.lead_m);
        const uint32_t gf0 = (uint32_t)(bo0 >> sm.gsh) & fmask, gf1 = (uint32_t)(bo1 >> sm.gsh) & fmask;
        const uint64_t sus0 = bal(lead && gf0 != 0u && !(L0 + __logf(S0) > pl.thr));
        const uint64_t sus1 = bal(lead && gf1 != 0u && !(L1 + __logf(S1) > pl.thr));
        uint64_t fb0 = 0ull, fb1 = 0ull;            // groups (leader-lane bits) that fall back to uniform over the mask
#ifdef SY_POL_NO_FALLBACK
        if (false) {
#else
        if ((sus0 | sus1) != 0ull) {                // rare: evaluate the suspicious actors exactly, one (episode, agent) at a time
#endif
            for (int h = 0; h < 2; ++h) {
                uint64_t todo = h ? sus1 : sus0;
                while (todo != 0ull) {
                    const int ll = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const int ag = rdlane(pl.ag, ll);
                    const uint32_t hs = (uint32_t)rdlane((int)(h ? pl.hs1 : pl.hs0), ll);
                    const float lse = __int_as_float(rdlane(__float_as_int(h ? L1 + __logf(S1) : L0 + __logf(S0)), ll));
                    const float* w2u = w2_all + (size_t)ag * N * H;       // actor `ag` (wave-uniform)
                    const float* b2u = b2_all + (size_t)ag * N;
                    const float mass = exact_legal_mass<NR>(w2u, b2u, hs, H, N, lane, lse);
                    if (mass <= 1e-8f) { if (h) fb1 |= 1ull << ll; else fb0 |= 1ull << ll; }
                }
            }
        }
        if ((fb0 | fb1) != 0ull) {                  // my group's leader bit -> my fallback flag
            const int lead_lane = lane - sm.col;    // the first lane of my group
            const bool f0 = ((fb0 >> lead_lane) & 1ull) != 0ull, f1 = ((fb1 >> lead_lane) & 1ull) != 0ull;
            l0 = f0 ? 0.0f : l0; L0 = f0 ? 0.0f : L0; S0 = f0 ? (float)__popc(gf0) : S0;
            l1 = f1 ? 0.0f : l1; L1 = f1 ? 0.0f : L1; S1 = f1 ? (float)__popc(gf1) : S1;
        }
    }
    // Gumbel-max draw: a cheap per-lane hash of the agent's Philox word of this step
    auto gumbel = [&sm](uint32_t x) {
        uint32_t h = x ^ ((uint32_t)sm.col * 0x9E3779B9u) ^ 0x85EBCA6Bu;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const float u = fminf(((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f), 0x1.fffffep-1f);   // (h >> 8 = 2^24 - 1 rounds to 1.0f: -log(-log 1) = +inf)
        return -__logf(-__logf(u));
    };
    const float k0 = l0 + gumbel(g.xa0), k1 = l1 + gumbel(g.xa1);
    if (own0) atomicMax(lds_at_generic<int>(pl.sl0), f32_ordered(k0));
    if (own1) atomicMax(lds_at_generic<int>(pl.sl1), f32_ordered(k1));
    wave_lds_fence();
    const bool win0 = own0 && f32_ordered(k0) == *lds_at<int>(pl.sl0), win1 = own1 && f32_ordered(k1) == *lds_at<int>(pl.sl1);
    if (win0) {
        *lds_at<int>(q.selw0) = (int)g.ent0;
        *lds_at<float>(pl.sl0 + 12u) = (l0 - L0) - __logf(S0);
    }
    if (win1) {
        *lds_at<int>(q.selw1) = (int)g.ent1;
        *lds_at<float>(pl.sl1 + 12u) = (l1 - L1) - __logf(S1);
    }
    if (lanes(q.lead_m)) {
        lds_at<int>(q.selw0)[1] = __popc((uint32_t)(bq0 >> sm.gsh) & fmask);
        lds_at<int>(q.selw1)[1] = __popc((uint32_t)(bq1 >> sm.gsh) & fmask);
    }
    wave_lds_fence();
    const uint64_t r = *lds_at<uint64_t>(q.selr);
    act_v = (int)(int16_t)(uint32_t)r;              // 0xffff -> -1
    cost_v = (int)(((uint32_t)r) >> 16);
    quirk_cnt = (int)(r >> 32);
    logp_v = *lds_at<float>(pl.slr + 12u);
    wave_lds_fence();
}

__device__ __forceinline__ int hbcast(int v, int src_local, bool upper) {   // v of local lane src_local of my half
    const int lo = rdlane(v, src_local), hi = rdlane(v, 32 + src_local);
    return upper ? hi : lo;
}
__device__ __forceinline__ bool hany(bool pred, bool upper) {               // pred on any lane of my half
    const uint64_t bm = __ballot(pred);
    return (upper ? (uint32_t)(bm >> 32) : (uint32_t)bm) != 0u;
}

// sample_starts (distinct start nodes, see above) for the halves named in `need` (bit 0 / bit 32): agent a of half h
// sits on lane 32 h + a; gid / ctr are replicated per half.  Tuple rejection: one Philox block per agent lane serves
// four attempts, distinctness is four DPP row shifts and scalar masks; the sequential fallback (small boards) draws on
// the vector unit and keeps the without-replacement bookkeeping on the scalar unit — its values are uniform per half.
__device__ __forceinline__ int sample_starts_pair(uint64_t need, int ln, int a, int A, int N, uint64_t gid, uint32_t ctr,
                                               uint32_t k0, uint32_t k1) {
    uint32_t o[4];
    int st = 0;
    if (N >= 2 * A * A) {
        uint64_t todo = half_any(need);
        for (uint32_t j = 0; j < 128u && todo != 0ull; ++j) {
            if ((j & 3u) == 0u) philox4(gid, ctr, kPurposeReset, ((j >> 2) << 3) | ((uint32_t)a & 7u), k0, k1, o);
            const uint32_t m = j & 3u;
            const uint32_t x = m == 0 ? o[0] : (m == 1 ? o[1] : (m == 2 ? o[2] : o[3]));
            const int r = (int)__umulhi(x, (uint32_t)N);
            const uint64_t good = todo & ~half_any(earlier_duplicates(r, A) & 0x000000ff000000ffull);
            st = lanes(good) ? r : st;
            todo &= ~good;
        }
        if (todo == 0ull) return st;
        need = todo;
    }
    philox4(gid, ctr, kPurposeReset, (uint32_t)a, k0, k1, o);
    const int xv = (int)o[0];
    for (int h = 0; h < 2; ++h) {
        if (((need >> (32 * h)) & 1ull) == 0ull) continue;
        int sorted[SY_MAX_AGENTS];
#pragma unroll
        for (int j = 0; j < SY_MAX_AGENTS; ++j) sorted[j] = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < SY_MAX_AGENTS; ++i) {
            if (i < A) {
                const uint32_t x = (uint32_t)rdlane(xv, 32 * h + i);
                int r = (int)__umulhi(x, (uint32_t)(N - i));
#pragma unroll
                for (int j = 0; j < SY_MAX_AGENTS; ++j)
                    if (j < i) r += (r >= sorted[j]) ? 1 : 0;
#pragma unroll
                for (int j = SY_MAX_AGENTS - 1; j >= 0; --j) {
                    const int prev = j == 0 ? -1 : sorted[j - 1];
                    sorted[j] = sorted[j] < r ? sorted[j] : (prev < r ? r : prev);
                }
                st = ln == 32 * h + i ? r : st;
            }
        }
    }
    return st;
}

// shaped_reward for a paired wave (reward_calculator.py:94-266): min / sum of the police-to-MrX
// distances by DPP butterflies over the 8 agent lanes of a row, the proximity filter (d > 1) folded
// into a second table, MrX's / police terms selected once at the end.
template <int CTRL>
__device__ __forceinline__ int dpp_perm(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ double shaped_reward3(const RewardTabs& tb, int a, int P, uint64_t POLM, int t_v, int qcnt, int vc,
                                                 int dm, const int (&dj)[SY_MAX_AGENTS - 1], const Coefs<true>& kc) {
    int mn = lanes(POLM) ? dm : 0x7fffffff, sum = dm;        // dm is 0 off the police lanes
    { const int o = dpp_perm<0xB1>(mn); mn = o < mn ? o : mn; }   // lane ^ 1
    sum += dpp_perm<0xB1>(sum);
    { const int o = dpp_perm<0x4E>(mn); mn = o < mn ? o : mn; }   // lane ^ 2
    sum += dpp_perm<0x4E>(sum);
    { const int o = dpp_perm<0x141>(mn); mn = o < mn ? o : mn; }  // lane -> 7 - lane (the other quad)
    sum += dpp_perm<0x141>(sum);
    int dor = dm | (vc < kLdsTab ? 0 : kLdsTab) | (sum < kAvgTab ? 0 : kLdsTab);   // mn <= dm-values <= sum
#pragma unroll
    for (int j = 1; j < SY_MAX_AGENTS; ++j) dor |= dj[j - 1];
    double xa, xb, group = 0.0, prox = 0.0, e_mrx, cov;
    int overlap = 0;
    if (bal(dor >= kLdsTab) == 0ull) {
        // fast path (wave-uniform): every lookup hits the LDS tables, all reads issued back to back
        SY_HOT(m_rew_fast);
        xa = lds_f64(tb.nrc_s + mn);     // lanes past the agents read out of range: LDS returns garbage or 0, unused
        xb = lds_f64(tb.nra_s + sum);
        e_mrx = lds_f64(tb.exp_s + dm);
        cov = lds_f64(tb.cov_s + vc);
        double ex[SY_MAX_AGENTS - 1], px[SY_MAX_AGENTS - 1];
        int idx[SY_MAX_AGENTS - 1];
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) {
            idx[j - 1] = j != a ? dj[j - 1] : kLdsTab;                  // own slot reads the 0.0 entries
            ex[j - 1] = j <= P ? lds_f64(tb.exp_s + idx[j - 1]) : 0.0;
            px[j - 1] = j <= P ? lds_f64(tb.px_s + idx[j - 1]) : 0.0;
        }
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) {
            if (j <= P) {
                group += ex[j - 1];                                  // x + 0.0 == x
                prox += px[j - 1];
                overlap += idx[j - 1] <= 1 ? 1 : 0;
            }
        }
    } else {
        xa = -1.0 / ((double)mn + 1.0);
        xb = -1.0 / ((double)sum / (double)P + 1.0);
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) {
            if (j <= P) {
                const int dij = dj[j - 1];
                const bool other = j != a;
                const double ex = other ? exp_neg_slow(tb, dij) : 0.0;
                group += ex;
                prox += dij > 1 ? ex : 0.0;
                overlap += (other && dij <= 1) ? 1 : 0;
            }
        }
        e_mrx = exp_neg_slow(tb, dm);
        cov = tb.cov_g[vc < tb.n_cov ? vc : tb.n_cov - 1];
    }
    const double ts = (double)t_v;
    const double x0 = a == 0 ? xa : e_mrx, x1 = a == 0 ? xb : group;
    const double base = ((kc.get(0) * x0 + kc.get(1) * x1) + kc.get(2) * (double)qcnt) + kc.get(3) * (kc.get(7) * ts);
    const double pol = ((base + kc.get(4) * prox) - kc.get(5) * (double)overlap) + kc.get(6) * cov;
    return a == 0 ? base : pol;
}

// Diagnostic phase timers (-DSY_STAMPS builds only; each stamp drains the LDS queue, so the build is
// for attribution, not for benchmarking).
#ifdef SY_STAMPS
#define SY_STAMP_DECL unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#define SY_STAMP(i) { const unsigned long long stamp_n = __builtin_amdgcn_s_memtime(); stamp_acc[i] += stamp_n - stamp_t; stamp_t = stamp_n; }
#define SY_STAMP_DUMP(T) if (blockIdx.x == 7 && threadIdx.x == 0) printf("stamps/step: C %llu G %llu V %llu B %llu F %llu D %llu R %llu E %llu S %llu\n", stamp_acc[0] / T, stamp_acc[1] / T, stamp_acc[2] / T, stamp_acc[3] / T, stamp_acc[4] / T, stamp_acc[5] / T, stamp_acc[6] / T, stamp_acc[7] / T, stamp_acc[8] / T);
#elif defined(SY_ENDTIMES)   // load-balance builds: per move wave (start, end) on the constant 100 MHz clock, left in the
                             // two padding words of the last record row (tools/endtimes.py reads them)
#define SY_STAMP_DECL const unsigned long long wave_t0 = __builtin_amdgcn_s_memrealtime();
#define SY_STAMP(i)
#define SY_STAMP_DUMP(T) if (REC && a0 == 0 && store_ok) { int* lastrow = out.record - (size_t)B * RW + (size_t)eh * RW; lastrow[RW - 2] = (int)(unsigned)wave_t0; lastrow[RW - 1] = (int)(unsigned)__builtin_amdgcn_s_memrealtime(); }
#elif defined(SY_PHASES)   // static census builds: tools/asm_phase_count.py reads the markers from the assembly
#define SY_STAMP_DECL
#define SY_STAMP(i) asm volatile("; ##PHASE P" #i);
#define SY_STAMP_DUMP(T) asm volatile("; ##PHASE epilogue");
#else
#define SY_STAMP_DECL
#define SY_STAMP(i)
#define SY_STAMP_DUMP(T)
#endif

}  // namespace sy
