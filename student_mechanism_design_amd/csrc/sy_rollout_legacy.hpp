// sy_rollout_legacy.hpp — round 1's fused rollouts: rollout_kernel (one episode per move wave, odd block sizes)
// and rollout2_kernel (paired move waves + belief waves: boards of more than 256 nodes, more than two scan
// passes, A/B baseline).  Instantiated by sy_rollout1.hip / sy_rollout2_*.hip.
#pragma once
#include "sy_pair.hpp"

namespace sy {

// The belief wave of the fused rollout: serves the episodes in LDS slots `slot` and `slot + 1`
// (a belief step is less than half a move step).  See rollout_kernel for the hand-off protocol.
template <int NR, bool REC>
__device__ __forceinline__ void belief_wave_run(const EngineParams& p, const LdsMap& L, const EnvLds& E, int lane, int slot,
                                                int e, int g, int wpb, int P, int A, int T, sy_rollout_buffers out) {
    const int N = p.N, NS = p.NS, B = p.B;
        // ================================ belief wave ================================
        // serves two episodes (slots `slot`, `slot+1`): a belief step is less than half a move step,
        // so 1.5 waves per episode keep the chip at 6 waves per SIMD with 80 VGPRs each.
        const bool live1 = (slot + 1 < wpb) && (e + 1 < B);
        const EnvLds E1 = env_lds(L.env_base, live1 ? slot + 1 : slot, p.wave_lds_bytes, A, NS);
        float b0[NR], b1[NR], ideg[NR];
        int slab_w[NR];
        belief_load<NR>(b0, ideg, slab_w, p.st.belief + (size_t)e * NS, p.inv_deg + (size_t)g * NS, lane, N);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = lane + 64 * r;
            b1[r] = (live1 && j < N) ? p.st.belief[(size_t)(e + 1) * NS + j] : 0.0f;
        }
        const uint32_t off_bel = ((uint32_t)e * (uint32_t)NS + (uint32_t)lane) * 4u;
        const bool rec_bel = REC && out.belief != nullptr;
        const bool onehot = p.belief_onehot != 0, pol_ev = p.police_ev != 0;
        for (int s = 0; s < T; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && !live1) break;
                float (&b)[NR] = h == 0 ? b0 : b1;
                const EnvLds& Eh = h == 0 ? E : E1;
                if (rec_bel) {
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (lane + 64 * r < NS) *at_bytes(out.belief, off_bel + (uint32_t)(h * NS) * 4u + 256u * r) = b[r];
                }
                {
                    int spin = 0;
                    for (; lds_peek(Eh.sync) <= s && spin < kSpinMax; ++spin) __builtin_amdgcn_s_sleep(2);
                    if (spin == kSpinMax) report_status(SY_STATUS_BELIEF_WAIT_EXPIRED);
                }
                asm volatile("" ::: "memory");
                const int4* ent = reinterpret_cast<const int4*>(Eh.ring + (s & (kRing - 1)) * 8);
                const int4 e0v = ent[0], e1v = ent[1];
                const int head = __builtin_amdgcn_readfirstlane(e0v.x);
                const int pol[SY_MAX_AGENTS - 1] = {
                    __builtin_amdgcn_readfirstlane(e0v.y), __builtin_amdgcn_readfirstlane(e0v.z),
                    __builtin_amdgcn_readfirstlane(e0v.w), __builtin_amdgcn_readfirstlane(e1v.x),
                    __builtin_amdgcn_readfirstlane(e1v.y), __builtin_amdgcn_readfirstlane(e1v.z),
                    __builtin_amdgcn_readfirstlane(e1v.w)};
                asm volatile("" ::: "memory");
                if (lane == 0) lds_poke(Eh.sync + 1, s + 1);   // entry copied to registers: the slot may be reused
                const int node = head & 0xffff, flags = head >> 16;
                if (flags & 1) belief_prior<NR>(b, lane, N, onehot, node);          // new episode
                else if (flags & 2) belief_prior<NR>(b, lane, N, true, node);       // reveal -> delta
                else belief_step<NR>(b, ideg, slab_w, Eh.c_s, L.boff_s, lane, N, pol_ev, pol, P);
            }
            if (rec_bel) out.belief += (size_t)B * NS;
        }
        float* bel_out = kernarg_params()->st.belief;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = lane + 64 * r;
            if (j < NS) {
                bel_out[(size_t)e * NS + j] = b0[r];
                if (live1) bel_out[(size_t)(e + 1) * NS + j] = b1[r];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// rollout_kernel: T fused env steps per launch with the uniform-random policy (sy_env_rollout).
// Waves [0, wpb) of a block are the move waves of its episodes; when the engine tracks a belief,
// waves [wpb, 2*wpb) are their belief waves.  Hand-off: after step s the move wave writes one ring
// entry {MrX node | flags, police nodes} and bumps `produced`; the belief wave records belief s,
// waits for entry s, applies prior / reveal / diffusion and bumps `consumed`.  LDS operations of a
// wave are performed in order, so data-then-counter needs no extra wait; all spins are bounded.
// ---------------------------------------------------------------------------------------------
template <int NR, bool REC, int PT>   // PT > 0: police count fixed at compile time (loops over police fully unrolled)
__global__ __launch_bounds__(768, SY_ROLLOUT_MIN_WAVES) void rollout_kernel(const EngineParams p, const int T, const sy_rollout_buffers out_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool has_belief = p.st.belief != nullptr;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = p.wpb;                          // move waves (= episodes) per block
    const bool belief_role = wid >= wpb;            // belief wave k serves episodes 2k and 2k+1 of the block
    const int slot = belief_role ? 2 * (wid - wpb) : wid;
    const int P = PT > 0 ? PT : p.P, A = P + 1;
    const int N = p.N, NS = p.NS, B = p.B;
    const int e0 = blockIdx.x * wpb;
    const int e = e0 + slot;
    const LdsMap L = lds_map(smem, N);
    const EnvLds E = env_lds(L.env_base, slot, p.wave_lds_bytes, A, NS);
    int g = __builtin_amdgcn_readfirstlane(p.env_graph[e0 < B ? e0 : B - 1]);
    g = g < 0 ? 0 : (g >= p.G ? p.G - 1 : g);
    stage_block<true>(p, L, g, N);
    if (!belief_role && lane == 0) {
        E.sync[0] = 0;
        E.sync[1] = 0;
    }
    __syncthreads();
    if (e >= B) return;
    sy_rollout_buffers out = out_arg;

    if (belief_role) {
        belief_wave_run<NR, REC>(p, L, E, lane, slot, e, g, wpb, P, A, T, out);
        return;
    }

    // ================================== move wave ==================================
    const uint16_t* __restrict__ ap = p.apsp + (size_t)g * N * N;
    const uint64_t gid = p.env_id_offset + (uint64_t)e;
    const int n16 = (A * NS) >> 4;
    const ScanMap sm = make_scan_map(lane, p.scan_w);
    Coefs<true> kc;
    kc.s = L.kc_s + (lane == 0 ? 0 : 8);
    RewardTabs tb;
    tb.exp_s = L.exp_s; tb.cov_s = L.cov_s; tb.nrc_s = L.nrc_s; tb.nra_s = L.nra_s;
    tb.exp_g = p.exp_tab; tb.cov_g = p.cov_tab; tb.n_exp = p.n_exp; tb.n_cov = p.n_cov;

    // ---- load the episode state: coalesced reads of the batched tensors
    int pos_v = lane < A ? p.st.pos[(size_t)e * A + lane] : 0;
    int mon_v = lane < A ? p.st.budget[(size_t)e * A + lane] : 0;
    int t = __builtin_amdgcn_readfirstlane(p.st.t[e]);                       // wave-uniform: keep in SGPRs
    uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.st.step_count[e]);
    for (int i = lane; i < (NS >> 3); i += kWave)
        reinterpret_cast<uint4*>(E.vis_s)[i] = reinterpret_cast<const uint4*>(p.st.visits + (size_t)e * NS)[i];
    int rev_ctr = p.reveal_k > 0 ? p.reveal_k - (t % p.reveal_k) : 0;   // steps until the next reveal
    // action draws: word (step_count & 3) of philox(env, step_count >> 2, ACT, lane) serves the step with that
    // counter; the action of the NEXT step is sampled inside each scan, so xw always covers the next counter.
    uint32_t xw[4];
    philox4(gid, sc >> 2, kPurposeAct, (uint32_t)lane, p.seed_lo, p.seed_hi, xw);
    auto draw_word = [&xw](uint32_t c) {
        const uint32_t m = c & 3u;
        return m == 0 ? xw[0] : (m == 1 ? xw[1] : (m == 2 ? xw[2] : xw[3]));
    };
    int qcnt, act_v, cost_v;
    scan_sample(L.ell_s, E.mrow, lane, A, NS, n16, p.scan_w, sm, pos_v, mon_v, draw_word(sc), act_v, cost_v, qcnt);

    // trajectory cursors: uniform base pointers advanced once per step + constant 32-bit lane offsets.
    // The per-step record (reward, pos, budget, action, t, flags) is assembled in LDS in its packed
    // layout and leaves as ONE coalesced store of RW dwords per episode and step.
    const int RW = p.rec_words;
    const uint32_t off_rec = ((uint32_t)e * (uint32_t)RW + (uint32_t)lane) * 4u;
    const uint32_t off_mask = (uint32_t)e * (uint32_t)(A * NS) + (uint32_t)lane * 16u;
    const size_t BA = (size_t)B * A;
    E.rec_s[lane] = 0;                            // padding words of the record row stay zero
    int* rec_rew = E.rec_s + 2 * lane;            // lane a: reward as two dwords
    int* rec_pos = E.rec_s + 2 * A + lane;        // lane a: pos / budget / action at +0, +A, +2A

    double rew = 0.0;
    int term = 0, trunc = 0, win = 0;

    for (int s = 0; s < T; ++s) {
        // Lane predicates are recomputed from this laundered copy every step: hoisted out of the loop
        // they would each pin an SGPR pair (and get spilled / reloaded by v_readlane).
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const bool is_pol = ln >= 1 && ln <= P;
        // ---- C. moves (yard.py:161-243); this step's action was sampled by the previous scan
        const int pos0_v = pos_v, mon0_v = mon_v;   // pre-step observation, recorded below
        const uint64_t skipm = __ballot(act_v == -1 || mon_v == 0);               // :210-215
        resolve_moves_fast<PT>(ln, P, is_pol, act_v >= 0 ? act_v : pos_v, skipm, cost_v, pos_v, mon_v);
        const uint64_t polm = ((1ull << P) - 1ull) << 1;
        const bool no_money = (skipm & polm) == polm;                             // :191,216
        // node_visit_counts (yard.py:244-245): police never share a node, so no conflicts
        int vc = 0;
        if (is_pol) {
            vc = (int)E.vis_s[pos_v] + 1;
            E.vis_s[pos_v] = (uint16_t)vc;
        }
        const int mrx = rdlane(pos_v, 0);
        // shortest-path lookups for the shaped rewards are issued now and consumed after the scan
        const uint32_t rowb = (uint32_t)(pos_v * N) * 2u;
        int dm = 0;
        int dj[SY_MAX_AGENTS - 1];
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) dj[j - 1] = 0;
        if (is_pol) {
            dm = (int)*at_bytes(ap + mrx, rowb);
#pragma unroll
            for (int j = 1; j < SY_MAX_AGENTS; ++j)
                if (j <= P) dj[j - 1] = (int)*at_bytes(ap + rdlane(pos_v, j), rowb);
        }

        // ---- B. record the pre-step observation and the action.  Issued after the loads above: vector
        //      memory returns in order, so the reward lookups never queue behind this step's stores.
        if (REC) {
            if (out.mask) {
                for (int base16 = 0; base16 < n16; base16 += kWave)   // wave-uniform trip count
                    if (base16 + ln < n16)
                        *reinterpret_cast<uint4*>(out.mask + off_mask + (uint32_t)base16 * 16u) =
                            reinterpret_cast<const uint4*>(E.mrow)[base16 + ln];
            }
        }

        // ---- F. post-move scan: masks for the next observation, position-reward counts, next action
        const uint32_t nxt = sc + 1u;
        if ((nxt & 3u) == 0u) philox4(gid, nxt >> 2, kPurposeAct, (uint32_t)ln, p.seed_lo, p.seed_hi, xw);
        const uint32_t x_next = draw_word(nxt);
        int act_n, cost_n;
        scan_sample(L.ell_s, E.mrow, ln, A, NS, n16, p.scan_w, sm, pos_v, mon_v, x_next, act_n, cost_n, qcnt);

        // ---- D. outcome priority (reward_calculator.py:63-90), flags shared by all agents
        const bool captured = __ballot(is_pol && pos_v == mrx) != 0ull;
        const bool timeout = t > p.max_t;  // pre-increment timestep
        term = (captured || (!timeout && no_money)) ? 1 : 0;
        trunc = (!captured && timeout) ? 1 : 0;
        win = captured ? 1 : ((timeout || no_money) ? 2 : 0);
        const bool ended = (term | trunc) != 0;
        if (ended) rew = captured ? (ln == 0 ? -1.0 : 1.0) : (ln == 0 ? 1.0 : 0.0);
        else rew = shaped_reward<true>(tb, ln, P, is_pol, t, qcnt, vc, dm, dj, kc);
        t += 1;   // yard.py:355
        sc += 1;
        if (REC) {
            if (ln < A) {
                rec_rew[0] = __double2loint(rew);
                rec_rew[1] = __double2hiint(rew);
                rec_pos[0] = pos0_v;
                rec_pos[A] = mon0_v;
                rec_pos[2 * A] = act_v;
            }
            if (ln < 4) E.rec_s[5 * A + ln] = ln == 0 ? t - 1 : (ln == 1 ? term : (ln == 2 ? trunc : win));
            wave_lds_fence();
            if (ln < RW) *at_bytes(out.record, off_rec) = E.rec_s[ln];
            out.record += (size_t)B * RW;
            if (out.mask) out.mask += BA * NS;
        }

        // ---- E. next episode (auto-reset) and the hand-off to the belief wave
        int flags = 0;
        if (ended && p.auto_reset) {
            const int st = sample_starts(ln, A, N, gid, sc, p.seed_lo, p.seed_hi);
            pos_v = ln < A ? st : 0;
            mon_v = ln == 0 ? SY_MRX_MONEY : (ln < A ? p.money0 : 0);   // yard.py:117-119
            t = 0;
            rev_ctr = p.reveal_k;
            for (int i = ln; i < (NS >> 3); i += kWave) reinterpret_cast<uint4*>(E.vis_s)[i] = make_uint4(0, 0, 0, 0);
            wave_lds_fence();
            scan_sample(L.ell_s, E.mrow, ln, A, NS, n16, p.scan_w, sm, pos_v, mon_v, x_next, act_n, cost_n, qcnt);
            flags = 1;
        } else if (p.reveal_k > 0 && --rev_ctr == 0) {   // post-increment timestep is a multiple of reveal_k
            rev_ctr = p.reveal_k;
            flags = 2;
        }
        if (has_belief) {
            {
                int spin = 0;
                for (; s - lds_peek(E.sync + 1) >= kRing && spin < kSpinMax; ++spin) __builtin_amdgcn_s_sleep(2);
                if (spin == kSpinMax) report_status(SY_STATUS_RING_WAIT_EXPIRED);
            }
            asm volatile("" ::: "memory");
            int* slot_p = E.ring + (s & (kRing - 1)) * 8;
            if (ln < 8) slot_p[ln] = ln == 0 ? (pos_v | (flags << 16)) : (ln <= P ? pos_v : -1);
            asm volatile("" ::: "memory");
            if (ln == 0) lds_poke(E.sync, s + 1);
        }
        act_v = act_n;
        cost_v = cost_n;
    }

    // ---- write the live state back (coalesced); state pointers re-read from the kernel arguments
    const KernargParams kq = kernarg_params();
    sy_env_state st;
    st.pos = kq->st.pos; st.budget = kq->st.budget; st.t = kq->st.t; st.step_count = kq->st.step_count;
    st.visits = kq->st.visits; st.belief = kq->st.belief; st.mask = kq->st.mask; st.reward = kq->st.reward;
    st.terminated = kq->st.terminated; st.truncated = kq->st.truncated; st.winner = kq->st.winner;
    if (lane < A) {
        st.pos[(size_t)e * A + lane] = pos_v;
        st.budget[(size_t)e * A + lane] = mon_v;
        st.reward[(size_t)e * A + lane] = rew;
    }
    if (lane == 0) {
        st.t[e] = t;
        st.step_count[e] = sc;
        st.terminated[e] = (uint8_t)term;
        st.truncated[e] = (uint8_t)trunc;
        st.winner[e] = (int8_t)win;
    }
    for (int i = lane; i < (NS >> 3); i += kWave)
        reinterpret_cast<uint4*>(st.visits + (size_t)e * NS)[i] = reinterpret_cast<const uint4*>(E.vis_s)[i];
    {
        uint4* dst = reinterpret_cast<uint4*>(st.mask + (size_t)e * A * NS);
        for (int i = lane; i < n16; i += kWave) dst[i] = reinterpret_cast<const uint4*>(E.mrow)[i];
    }
}

// ---------------------------------------------------------------------------------------------
// rollout2_kernel: the fused rollout with PAIRED move waves.  Most of a step touches only the A <= 8
// agent lanes, so one move wave carries two episodes: lanes 0-31 hold episode `e`, lanes 32-63 episode
// `e + 1` (agent a on lane h*32 + a).  What was wave-uniform per episode (timestep, flags, MrX's node,
// ...) becomes a value replicated across the 32 lanes of a half; "any lane of my half" tests read the
// matching 32 bits of one 64-bit ballot; broadcasts from an agent lane are two v_readlane + one select.
// Only the 64-lane ELL scan and the mask-row copies run once per episode.  Instruction count per
// episode drops by about a third; block = wpb/2 move waves + wpb/2 belief waves (wpb even).
template <int NR, bool REC, int PT, bool POL = false>   // POL: actions from the MAPPO actors (sy_env_set_policy)
__global__ __launch_bounds__(1024, 4) void rollout2_kernel(const EngineParams p, const int T, const sy_rollout_buffers out_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool has_belief = p.st.belief != nullptr;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = p.wpb;                          // episodes per block (even)
    const int nmove = wpb >> 1;                     // move waves: two episodes each
    const bool belief_role = wid >= nmove;
    const int slot = 2 * (belief_role ? wid - nmove : wid);
    const int P = PT > 0 ? PT : p.P, A = P + 1;
    const int N = p.N, NS = p.NS, B = p.B;
    const int e0 = blockIdx.x * wpb;
    const int e = e0 + slot;                        // first episode of this wave's pair
    const LdsMap L = lds_map(smem, N);
    const EnvLds E = env_lds(L.env_base, slot, p.wave_lds_bytes, A, NS);
    int g = __builtin_amdgcn_readfirstlane(p.env_graph[e0 < B ? e0 : B - 1]);
    g = g < 0 ? 0 : (g >= p.G ? p.G - 1 : g);
    stage_block<true, 3>(p, L, g, N);
    const EnvLds E1 = env_lds(L.env_base, slot + 1, p.wave_lds_bytes, A, NS);
    if (!belief_role && lane == 0) {
        E.sync[0] = 0; E.sync[1] = 0;
        E1.sync[0] = 0; E1.sync[1] = 0;
    }
    __syncthreads();
    if (e >= B) return;
    sy_rollout_buffers out = out_arg;
    if (belief_role) {
        belief_pair_run<NR, REC>(p, L, E, E1, lane, e, g, P, T, out);
        return;
    }

    // ================================== paired move wave ==================================
    const bool live1 = e + 1 < B;                   // a missing second episode shadows the first (its stores are masked)
    const bool upper0 = lane >= 32;
    const int a0 = lane & 31;
    const int eh = (upper0 && live1) ? e + 1 : e;   // this lane's episode
    const bool store_ok = !upper0 || live1;
    const bool all_store = live1;                   // wave-uniform: no lane of the wave is a shadow
    const uint16_t* __restrict__ ap = p.apsp + (size_t)g * N * N;
    const uint64_t gid = p.env_id_offset + (uint64_t)eh;
    const int n16 = (A * NS) >> 4;
    const ScanMap sm = make_scan_map<(PT == 0 || PT >= 5)>(lane, p.scan_w);
    Coefs<true> kc;
    kc.s = L.kc_s + (a0 == 0 ? 0 : 8);
    RewardTabs tb;
    tb.exp_s = L.exp_s; tb.cov_s = L.cov_s; tb.nrc_s = L.nrc_s; tb.nra_s = L.nra_s; tb.px_s = L.px_s;
    tb.exp_g = p.exp_tab; tb.cov_g = p.cov_tab; tb.n_exp = p.n_exp; tb.n_cov = p.n_cov;
    // per-lane views of the half's LDS slice
    uint16_t* const vis_h = upper0 ? E1.vis_s : E.vis_s;
    int* const rec_h = upper0 ? E1.rec_s : E.rec_s;
    int* const ring_h = upper0 ? E1.ring : E.ring;
    int* const sync_h = upper0 ? E1.sync : E.sync;
    uint8_t* const mrow_h = upper0 ? E1.mrow : E.mrow;
    const uint32_t xch_off = lds_off(rec_h) + kSelWord * 4u;   // 8 words: agent positions exchanged inside a step

    // ---- load both episodes' state
    int pos_v = a0 < A ? p.st.pos[(size_t)eh * A + a0] : 0;
    int mon_v = a0 < A ? p.st.budget[(size_t)eh * A + a0] : 0;
    int t_v = p.st.t[eh];
    uint32_t sc_v = p.st.step_count[eh];
    uint32_t* const vis32 = reinterpret_cast<uint32_t*>(vis_h);   // 32-bit counters: one returning LDS add per step
    for (int i = a0; i < NS; i += 32) vis32[i] = p.st.visits[(size_t)eh * NS + i];
    int rev_v = p.reveal_k > 0 ? p.reveal_k - (t_v % p.reveal_k) : 0;
    uint32_t xw[4];
    philox4(gid, sc_v >> 2, kPurposeAct, (uint32_t)a0, p.seed_lo, p.seed_hi, xw);
    auto draw_word = [&xw](uint32_t c) {
        const uint32_t m = c & 3u;
        return m == 0 ? xw[0] : (m == 1 ? xw[1] : (m == 2 ? xw[2] : xw[3]));
    };
    const int RW = p.rec_words;
    const size_t BA = (size_t)B * A;
    if (a0 < 32) rec_h[a0 + 32 * 0] = 0;
    rec_h[32 + a0] = 0;                              // padding words of the record row stay zero
    wave_lds_fence();
    int qcnt = 0, act_v = -1, cost_v = 0;
    const bool one_pass = A <= sm.per_pass;          // wave-uniform: every agent scanned in a single pass
    // two passes of the slot-based scan (instances that can have more than 5 agents only: registers)
    const bool two_pass = (PT == 0 || PT >= 5) && !one_pass && A <= 2 * sm.per_pass;
    PairScanLane psl = make_pair_scan_lane(E, E1, sm, lane, A, NS);
    PairScanLane psl2 = psl;
    if (two_pass) psl2 = make_pair_scan_lane(E, E1, sm, lane, A, NS, sm.per_pass);
    // in-kernel policy: per-episode scratch behind the episode slices
    const uint32_t pol0 = lds_off(L.env_base) + (uint32_t)wpb * (uint32_t)p.wave_lds_bytes + (uint32_t)slot * SY_POLICY_SLICE;
    const uint32_t pol1 = pol0 + SY_POLICY_SLICE;
    PolLane3 pll;
    float logp_v = 0.0f;
    if (POL) pll = make_pol_lane3(p, sm, lane, A, pol0, pol1);
    if (one_pass || two_pass) {
        for (int i = lane; i < n16; i += kWave) {
            reinterpret_cast<uint4*>(E.mrow)[i] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(E1.mrow)[i] = make_uint4(0, 0, 0, 0);
        }
        wave_lds_fence();
        const ScanPairIn g0 = scan_gather_pair(L.ell_s, A, sm, 0, pos_v, mon_v, draw_word(sc_v));
        if (POL) {           // (the launcher only picks this instance for single-pass boards)
            policy_hidden_pair3(p, P, A, pos_v, lane, pol0, pol1);
            wave_lds_fence();
            scan_eval_pair_policy3<NR>(psl, pll, sm, p.scan_w, p.pH, N, lane, p.pw2, p.pb2, g0, act_v, cost_v, qcnt, logp_v);
        } else if (one_pass) {
            scan_eval_pair1(psl, sm, p.scan_w, g0, act_v, cost_v, qcnt);
        } else {
            const ScanPairIn g1 = scan_gather_pair(L.ell_s, A, sm, sm.per_pass, pos_v, mon_v, draw_word(sc_v));
            scan_eval_pair1<true, false>(psl, sm, p.scan_w, g0, act_v, cost_v, qcnt);
            scan_eval_pair1<false, true>(psl2, sm, p.scan_w, g1, act_v, cost_v, qcnt);
        }
    } else {
        scan_sample_pair(L.ell_s, E.mrow, E1.mrow, lane, A, NS, n16, p.scan_w, sm, pos_v, mon_v, draw_word(sc_v), act_v,
                         cost_v, qcnt);
    }
    double rew = 0.0;
    int term_v = 0, trunc_v = 0, win_v = 0;
    const uint64_t POLM = (((1ull << P) - 1ull) << 1) * 0x0000000100000001ull;   // police lanes of both halves
    constexpr uint64_t kMrxLanes = 0x0000000100000001ull;

    SY_STAMP_DECL
    for (int s = 0; s < T; ++s) {
        int ln = lane;                               // laundered: lane predicates are recomputed every step
        asm volatile("" : "+v"(ln));
        const bool upper = ln >= 32;
        const int a = ln & 31;
        const bool is_pol = a >= 1 && a <= P;
        SY_STAMP(8)

        // ---- C. moves (yard.py:161-243), both episodes at once
        const int pos0_v = pos_v, mon0_v = mon_v;
        const int tgt_v = act_v >= 0 ? act_v : pos_v;
        const uint64_t SK = bal(act_v == -1) | bal(mon_v == 0);               // skipped agents (:210-215)
        {   // MrX vs PRE-move police (:180-188)
            const int t_lo = rdlane(tgt_v, 0), t_hi = rdlane(tgt_v, 32);
            const uint64_t hit = half_pick(bal(pos_v == t_lo), bal(pos_v == t_hi)) & POLM;
            pos_v = lanes(kMrxLanes & ~half_any(hit)) ? tgt_v : pos_v;
        }
        // any police pair that could interact this step (same target, or one moving onto the other's node)?
        uint64_t CF = 0;
        if (1 < P) CF |= pair_conflicts<1>(tgt_v, pos_v) & (POLM & (POLM << 1));
        if (2 < P) CF |= pair_conflicts<2>(tgt_v, pos_v) & (POLM & (POLM << 2));
        if (3 < P) CF |= pair_conflicts<3>(tgt_v, pos_v) & (POLM & (POLM << 3));
        if (4 < P) CF |= pair_conflicts<4>(tgt_v, pos_v) & (POLM & (POLM << 4));
        if (5 < P) CF |= pair_conflicts<5>(tgt_v, pos_v) & (POLM & (POLM << 5));
        if (6 < P) CF |= pair_conflicts<6>(tgt_v, pos_v) & (POLM & (POLM << 6));
        if (CF == 0ull) {                              // no police collision in either episode: order cannot matter
            SY_HOT(m_moves);
            const uint64_t mv = POLM & ~SK & bal(tgt_v != pos_v);
            pos_v = lanes(mv) ? tgt_v : pos_v;
            mon_v -= lanes(mv) ? cost_v : 0;                                  // :234-236
        } else {                                      // exact sequential order (:191-243), harmless for a clean half
            const bool skip_v = lanes(SK);
            for (int k = 1; k <= P; ++k) {
                const int tk = hbcast(tgt_v, k, upper);
                const bool occ = hany(is_pol && pos_v == tk, upper);          // own node included (:231)
                if (!occ && !skip_v && a == k) {
                    pos_v = tk;
                    mon_v -= cost_v;
                }
            }
        }
        const uint64_t NM = ~half_any(POLM & ~SK);                            // nobody could act (:191,216)
        // ---- outcome priority (reward_calculator.py:63-90): known as soon as the moves are
        const uint64_t CAP = half_any(half_pick(bal(pos_v == rdlane(pos_v, 0)), bal(pos_v == rdlane(pos_v, 32))) & POLM);
        const uint64_t TO = bal(t_v > p.max_t);                               // t_v is replicated over its half
        const uint64_t ENDED = CAP | TO | NM;
        const uint64_t NEED = p.auto_reset != 0 ? ENDED : 0ull;
        term_v = lanes(CAP | (NM & ~TO)) ? 1 : 0;
        trunc_v = lanes(TO & ~CAP) ? 1 : 0;
        win_v = lanes(CAP) ? 1 : (lanes(TO | NM) ? 2 : 0);
        // ---- the state the next step starts from: a finished episode restarts right here, so the one
        // scan below already serves the new episode (its masks, its first action)
        int pos_n = pos_v, mon_n = mon_v;
        if (NEED != 0ull) {
            const int st = sample_starts_pair(NEED, ln, a, A, N, gid, sc_v + 1u, p.seed_lo, p.seed_hi);
            const int m_init = a == 0 ? SY_MRX_MONEY : (a < A ? p.money0 : 0);     // yard.py:117-119
            pos_n = lanes(NEED) ? st : pos_v;
            mon_n = lanes(NEED) ? m_init : mon_v;
        }
        SY_STAMP(0)
        // next step's draw + the gather half of the scan, issued now (see scan_gather_pair)
        const uint32_t nxt_v = sc_v + 1u;
        if (bal((nxt_v & 3u) == 0u) != 0ull) {
            uint32_t nw[4];
            philox4(gid, nxt_v >> 2, kPurposeAct, (uint32_t)a, p.seed_lo, p.seed_hi, nw);
            if ((nxt_v & 3u) == 0u) { xw[0] = nw[0]; xw[1] = nw[1]; xw[2] = nw[2]; xw[3] = nw[3]; }
        }
        const uint32_t x_next = draw_word(nxt_v);
        const ScanPairIn sg = scan_gather_pair(L.ell_s, A, sm, 0, pos_n, mon_n, x_next);
        ScanPairIn sg2 = sg;
        if (two_pass) sg2 = scan_gather_pair(L.ell_s, A, sm, sm.per_pass, pos_n, mon_n, x_next);
        if (POL) policy_hidden_pair3(p, P, A, pos_n, ln, pol0, pol1);      // hidden vectors of the next observation
        SY_STAMP(1)
        int vc = 0;
        if (is_pol) {                                                         // :244-245
            vc = (int)atomicAdd(vis32 + pos_v, 1u) + 1;
        }
        // every agent's node to every lane of its half through LDS (the result-slot words of the record
        // staging row, free until the scan is evaluated): one round trip instead of P + 1 lane broadcasts
        if (lanes(kAgentSlots)) lds_at<int>(xch_off)[a] = pos_v;
        wave_lds_fence();
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i qa = *lds_at<v4i>(xch_off), qb = *lds_at<v4i>(xch_off + 16u);
        const int q[SY_MAX_AGENTS] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        const uint32_t rowb = (uint32_t)(pos_v * N) * 2u;
        int dm = 0;
        int dj[SY_MAX_AGENTS - 1];
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) dj[j - 1] = 0;
        if (is_pol) {
            dm = (int)*at_bytes(ap, rowb + (uint32_t)q[0] * 2u);
#pragma unroll
            for (int j = 1; j < SY_MAX_AGENTS; ++j)
                if (j <= P) dj[j - 1] = (int)*at_bytes(ap, rowb + (uint32_t)q[j] * 2u);
        }

        SY_STAMP(2)
        // ---- B. record the pre-step masks (after the loads, see rollout_kernel)
        if (REC && out.mask) {
            // three 16-byte LDS reads in flight per lane (unconditional: a read past the rows is harmless),
            // then the predicated stores; rows longer than 96 x 16 B take the tail loop
            uint4* md = reinterpret_cast<uint4*>(out.mask + (size_t)eh * (size_t)(A * NS)) + a;
            const uint4* mr = reinterpret_cast<const uint4*>(mrow_h) + a;
            const uint4 v0 = mr[0], v1 = mr[32], v2 = mr[64];
            if (all_store && n16 >= 64) {      // wave-uniform: both episodes live, the first two stores are full
                md[0] = v0;
                md[32] = v1;
                if (a + 64 < n16) md[64] = v2;
                for (int i = 96; a + i < n16; i += 32) md[i] = mr[i];
            } else if (store_ok) {
                if (a < n16) md[0] = v0;
                if (a + 32 < n16) md[32] = v1;
                if (a + 64 < n16) md[64] = v2;
                for (int i = 96; a + i < n16; i += 32) md[i] = mr[i];
            }
        }

        SY_STAMP(3)
        // ---- F. evaluate half of the scan: masks, position-reward counts, next action
        int act_n = -1, cost_n = 0;
        float logp_n = 0.0f;
        if (POL) {
            scan_eval_pair_policy3<NR>(psl, pll, sm, p.scan_w, p.pH, N, ln, p.pw2, p.pb2, sg, act_n, cost_n, qcnt, logp_n);
        } else if (one_pass) {
            scan_eval_pair1(psl, sm, p.scan_w, sg, act_n, cost_n, qcnt);
        } else if (two_pass) {
            scan_eval_pair1<true, false>(psl, sm, p.scan_w, sg, act_n, cost_n, qcnt);
            scan_eval_pair1<false, true>(psl2, sm, p.scan_w, sg2, act_n, cost_n, qcnt);
        } else scan_eval_pair(L.ell_s, E.mrow, E1.mrow, ln, A, NS, n16, p.scan_w, sm, sg, pos_n, mon_n, x_next, act_n, cost_n, qcnt);
        SY_STAMP(4)

        // ---- D. rewards
        const double shaped = shaped_reward3(tb, a, P, POLM, t_v, qcnt, vc, dm, dj, kc);
        rew = lanes(ENDED) ? (lanes(CAP) ? (a == 0 ? -1.0 : 1.0) : (a == 0 ? 1.0 : 0.0)) : shaped;
        const int t_rec = t_v;
        t_v = lanes(NEED) ? 0 : t_v + 1;   // yard.py:355; a restarted episode begins at 0
        sc_v += 1u;
        SY_STAMP(5)

        SY_STAMP(6)
        // ---- E. bookkeeping of a restart / reveal, and the hand-off to the belief wave
        int flags_v = 0;
        if (NEED != 0ull) {
            if (lanes(NEED)) {
                rev_v = p.reveal_k;
                for (int i = a; i < (NS >> 2); i += 32) reinterpret_cast<uint4*>(vis32)[i] = make_uint4(0, 0, 0, 0);
                flags_v = 1;
            }
        }
        if (!lanes(NEED) && p.reveal_k > 0) {
            rev_v -= 1;
            if (rev_v == 0) {        // post-increment timestep is a multiple of reveal_k
                rev_v = p.reveal_k;
                flags_v = 2;
            }
        }
        if (has_belief) {
            // back-pressure: every kRing/2 steps make sure the belief wave is at most kRing/2 entries behind,
            // so the ring can never be overrun in between (two fewer LDS round trips on the other steps)
            if ((s & (kRing / 2 - 1)) == 0) {
                int spin = 0;
                for (; spin < kSpinMax; ++spin) {
                    const int c0 = lds_peek(E.sync + 1), c1 = live1 ? lds_peek(E1.sync + 1) : s;
                    if (s - c0 <= kRing / 2 && s - c1 <= kRing / 2) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (spin == kSpinMax) report_status(SY_STATUS_RING_WAIT_EXPIRED);
            }
            asm volatile("" ::: "memory");
            int* slot_p = ring_h + (s & (kRing - 1)) * 8;
            if (a < 8) slot_p[a] = a == 0 ? (pos_n | (flags_v << 16)) : (a <= P ? pos_n : -1);
            asm volatile("" ::: "memory");
#ifdef SY_INJECT_LOST_HANDOFF   // fault-injection build (tests only): episode 0 stops publishing after step 2
            if (a == 0 && !(eh == 0 && s >= 2)) lds_poke(sync_h, s + 1);
#else
            if (a == 0) lds_poke(sync_h, s + 1);
#endif
        }
        if (REC) {
            // the packed row straight from the agent lanes: five narrow stores into one 128-byte line
            int* rdst = out.record + (size_t)eh * RW;
            if (store_ok) {
                if (a < A) {
                    *reinterpret_cast<double*>(rdst + 2 * a) = rew;
                    rdst[2 * A + a] = pos0_v;
                    rdst[3 * A + a] = mon0_v;
                    rdst[4 * A + a] = act_v;
                }
                if (a < RW - 5 * A) rdst[5 * A + a] = a == 0 ? t_rec : (a == 1 ? term_v : (a == 2 ? trunc_v : (a == 3 ? win_v : 0)));
            }
            out.record += (size_t)B * RW;
            if (out.mask) out.mask += BA * NS;
            if (POL && out.log_prob) {
                if (store_ok && a < A) out.log_prob[(size_t)eh * A + a] = logp_v;   // of the action executed this step
                out.log_prob += BA;
            }
        }
        if (POL) logp_v = logp_n;
        pos_v = pos_n;
        mon_v = mon_n;
        act_v = act_n;
        cost_v = cost_n;
        SY_STAMP(7)
    }
    SY_STAMP_DUMP(T)

    // ---- write the live state back; state pointers re-read from the kernel arguments
    const KernargParams kq = kernarg_params();
    sy_env_state st;
    st.pos = kq->st.pos; st.budget = kq->st.budget; st.t = kq->st.t; st.step_count = kq->st.step_count;
    st.visits = kq->st.visits; st.belief = kq->st.belief; st.mask = kq->st.mask; st.reward = kq->st.reward;
    st.terminated = kq->st.terminated; st.truncated = kq->st.truncated; st.winner = kq->st.winner;
    if (store_ok) {
        if (a0 < A) {
            st.pos[(size_t)eh * A + a0] = pos_v;
            st.budget[(size_t)eh * A + a0] = mon_v;
            st.reward[(size_t)eh * A + a0] = rew;
        }
        if (a0 == 0) {
            st.t[eh] = t_v;
            st.step_count[eh] = sc_v;
            st.terminated[eh] = (uint8_t)term_v;
            st.truncated[eh] = (uint8_t)trunc_v;
            st.winner[eh] = (int8_t)win_v;
        }
        for (int i = a0; i < NS; i += 32) st.visits[(size_t)eh * NS + i] = (uint16_t)vis32[i];
        uint4* dst = reinterpret_cast<uint4*>(st.mask + (size_t)eh * A * NS);
        for (int i = a0; i < n16; i += 32) dst[i] = reinterpret_cast<const uint4*>(mrow_h)[i];
    }
}

}  // namespace sy
