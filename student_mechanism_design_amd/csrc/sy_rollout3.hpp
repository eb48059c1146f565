// sy_rollout3.hpp — rollout3_kernel, the move / helper pipeline (the default fused rollout).  Instantiated per
// instance group by sy_rollout3_*.hip so that an edit here rebuilds in parallel.
#pragma once
#include <type_traits>

#include "sy_pair.hpp"

namespace sy {

// ---------------------------------------------------------------------------------------------
// rollout3_kernel: the fused rollout as a PIPELINE of specialised waves (default for even block sizes).
//
// Measured on MI355X (tools/regime_probe.sh, round 2): one paired move wave ALONE on its SIMD needs 0.46 ms
// for 256 steps and four waves per SIMD need 0.55 ms — the launch is bound by the serial instruction stream
// of a step (every instruction of a wave costs >= 4 issue cycles, every LDS / L2 round trip is exposed), not
// by issue slots or HBM.  At B = 4096 a CU holds only 16 episodes, so the way to go faster is a SHORTER
// per-step chain.  Only the state feedback loop is inherently serial:
//       action -> moves -> outcome -> (restart) -> neighbour scan -> next action.
// The visit counters, the shortest-path gathers, the float64 rewards, the trajectory record and the belief
// filter only consume states and feed nothing back.  So:
//   move wave   (two episodes, lanes 0-31 / 32-63 as in rollout2): runs that loop and everything that reads the
//               board with it (mask rows and their record copy, visit counters, shortest-path gathers, the
//               float64 rewards), and publishes one ring entry per episode and step,
//               E[k] = {observation before step k, action of step k, reward and outcome marks of step k - 1};
//   helper wave (the same two episodes): everything that only LEAVES the chip — for transition k it holds E[k]
//               and reads E[k+1]: stores the packed record row {reward, pos, budget, action, t, flags}, and
//               runs the belief filter (new episode -> prior, reveal -> delta, else one diffusion step) with
//               its record rows.  Splitting these stores off takes ~30 % of the instructions (and every store
//               stall) out of the move wave's serial stream; the helper may lag up to kRing / 2 steps and the
//               move wave never waits for it otherwise.
// Ring entry (128 B per episode): agent slot a = {pos | action << 16, budget, reward (float64)}; slot 0's
// second word is the meta word  t | flags << 25  (MrX's budget is the constant SY_MRX_MONEY), flags: bit 0
// terminated, 1 truncated, 2-3 winner, 4 new episode, 5 reveal.
// ---------------------------------------------------------------------------------------------
static constexpr int kMetaCntShift = 20, kMetaFlagShift = 25, kMetaTimeMask = (1 << kMetaCntShift) - 1;
static constexpr int kRing3 = 8, kEntry3 = 128;    // the pipeline's ring: 8 entries of 128 B in the slice's 1 KB ring area
static_assert(kRing3 * kEntry3 == SY_RING * SY_RING_ENTRY_BYTES, "ring area");
static constexpr int kFlagTerm = 1, kFlagWinShift = 2, kFlagRestart = 16, kFlagReveal = 32;   // (bit 1: truncated)

#ifdef SY_STAMPS3   // phase timers of the pipeline roles (attribution only; every stamp drains the LDS queue)
#define S3_DECL unsigned long long s3_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long s3_t = __builtin_amdgcn_s_memtime();
#define S3(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long s3_n = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); s3_acc[i] += s3_n - s3_t; s3_t = s3_n; __builtin_amdgcn_sched_barrier(0); }
#define S3_DUMP(who, T) if (blockIdx.x == 7 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) % 8 == 1) printf("%s stamps/step: %llu %llu %llu %llu %llu %llu %llu %llu\n", who, s3_acc[0] / T, s3_acc[1] / T, s3_acc[2] / T, s3_acc[3] / T, s3_acc[4] / T, s3_acc[5] / T, s3_acc[6] / T, s3_acc[7] / T);
#else
#define S3_DECL
#define S3(i)
#define S3_DUMP(who, T)
#endif

// ---- the pipeline's neighbour scan: one episode per HALF wave ------------------------------------------------------
// The paired scan (scan_eval_pair1) gives every lane one (agent, ELL column) of BOTH episodes: two full instruction
// streams per lane, and two passes when the agents do not fit (6 or 7 agents at rows of more than 10 / 9 neighbours).
// Here the columns of one episode live in its own half: GW = 32 / A columns per agent (6 at P = 4, 5 at P = 5, 4 at
// P = 6), lane h*32 + g*GW + c scans columns c, c + GW, ... c + (NC-1) GW of agent g of episode h.  Columns beyond the
// first are short extra streams, evaluated unconditionally (rows wider than GW are rare — 3 % of the visits on
// reference-shaped 200-node boards — but a quarter of the pair-steps has one: a branch costs more than it saves);
// NC is the launch's choice (the pool's widest row fits NC * GW).  Counts, the r-th legal neighbour in ascending node
// order (all columns ranked in one NC*GW-bit field) and the position-reward count are the quantities of scan_sample;
// results reach the agent lanes through the LDS slots of the paired scan.
#ifndef SY_COOP_BATCH
#define SY_COOP_BATCH 3      // rounds (of 8 entries) whose row pieces are in flight together (hidden 33..64: two 16-byte pieces per lane and round)
#endif
template <int GW, int NC>
struct HalfScan {
    static_assert(GW * NC <= 32, "one rank field per agent");
    static constexpr uint32_t kField = (1u << GW) - 1u;
    uint32_t row, selw, selr, scratch, low0, bsrc, bsrcq, ell_col;
    uint32_t prev[NC];
    int gsh, col, ag;
    uint64_t on_m, lead_m;
    struct In { uint32_t ent[NC]; uint32_t xa; int ma, mq; };
    struct Pol {                 // learned policy (sy_env_set_policy): my group's actor and its LDS scratch, my episode
        uint32_t hs, sl, slr;
        uint32_t pol0, dpol, list, logits;   // the pair's scratch: episode 0's base, distance to episode 1's, entry descriptors, entry logits
        const float* w2a;
        const float* b2a;
        float thr;               // log(1e-8) + log(N) + bound[agent]   (+inf without a bound: never the exact path)
    };

    __device__ __forceinline__ void init(const LdsMap& L, const EnvLds& E, const EnvLds& E1, int lane, int A, int NS) {
        const bool up = lane >= 32;
        const int li = lane & 31, grp = li / GW;
        col = li - grp * GW;
        const bool on = grp < A;
        ag = on ? grp : 0;
        const uint32_t rec_h = lds_off(up ? E1.rec_s : E.rec_s);
        row = lds_off(up ? E1.mrow : E.mrow) + (uint32_t)(ag * NS);
        selw = rec_h + (uint32_t)(kSelWord + 2 * ag) * 4u;
        selr = rec_h + (uint32_t)(kSelWord + 2 * (lane & 7)) * 4u;
        scratch = rec_h + kDummyWord * 4u;
#pragma unroll
        for (int k = 0; k < NC; ++k) prev[k] = scratch;
        low0 = (1u << col) - 1u;                                   // entries of my agent ranked before my first column
        gsh = (lane & 32) + ag * GW;
        bsrc = (uint32_t)((lane & 32) + ag) * 4u;
        bsrcq = (uint32_t)((lane & 32) + (ag > 0 ? ag - 1 : 0)) * 4u;   // the PREVIOUS agent's budget (reward_calculator.py:190)
        ell_col = lds_off(L.ell_s) + (uint32_t)col * 4u;
        on_m = bal(on);
        lead_m = bal(on && col == 0);
    }
    // gather half: the agent's node, budgets and draw by bpermute, then my columns of the ELL row
    __device__ __forceinline__ In gather(int pos_v, int mon_v, uint32_t x_v) const {
        In g;
        const int pa = bperm((int)bsrc, pos_v);
        g.ma = bperm((int)bsrc, mon_v);
        g.mq = bperm((int)bsrcq, mon_v);
        g.xa = (uint32_t)bperm((int)bsrc, (int)x_v);
        const uint32_t rowaddr = ell_col + ((uint32_t)pa << 6);
        g.ent[0] = *lds_at<uint32_t>(rowaddr);
#pragma unroll
        for (int k = 1; k < NC; ++k) {                              // a column past the ELL row reads as padding: never affordable
            const bool in_row = col + k * GW < kD;
            const uint32_t e = *lds_at<uint32_t>(rowaddr + (in_row ? (uint32_t)(k * GW) * 4u : 0u));
            g.ent[k] = in_row ? e : 0xffff0000u;
        }
        return g;
    }
    // evaluate half: mask bytes of the new state, the next action (uniform over the legal neighbours) and the counts
    __device__ __forceinline__ void eval(const In& g, int& act_v, int& cost_v, int& quirk_cnt) {
        SY_HOT(m_eval);
        if (lanes(kAgentSlots)) *lds_at<uint64_t>(selr) = 0x0000ffffull;   // "no move": action -1, cost 0, count 0
#pragma unroll
        for (int k = 0; k < NC; ++k) *lds_at<uint8_t>(prev[k]) = 0;
        uint64_t bo[NC];
        uint32_t gf = 0, qf = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int w = (int)(g.ent[k] >> 16);
            bo[k] = bal(w <= g.ma) & on_m;
            const uint64_t bq = bal(w <= g.mq) & on_m;
            gf |= ((uint32_t)(bo[k] >> gsh) & kField) << (k * GW);
            qf |= ((uint32_t)(bq >> gsh) & kField) << (k * GW);
        }
        const int rr = (int)__umulhi(g.xa, (uint32_t)__popc(gf));
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            // lanes without an affordable entry write the scratch word instead of being masked off
            const uint32_t n = lanes(bo[k]) ? row + (g.ent[k] & 0xffffu) : scratch;
            *lds_at<uint8_t>(n) = 1;
            prev[k] = n;
            const uint32_t low = k == 0 ? low0 : ((1u << (k * GW)) - 1u) | (low0 << (k * GW));   // ranked before column k of mine
            const uint64_t ch = bal((int)__popc(gf & low) == rr) & bo[k];
            *lds_at<int>(lanes(ch) ? selw : scratch) = (int)g.ent[k];
        }
        if (lanes(lead_m)) lds_at<int>(selw)[1] = __popc(qf);
        wave_lds_fence();
        const uint64_t r = *lds_at<uint64_t>(selr);
        act_v = (int)(int16_t)(uint32_t)r;              // 0xffff -> -1
        cost_v = (int)(((uint32_t)r) >> 16);
        quirk_cnt = (int)(r >> 32);
        wave_lds_fence();
    }

    // ---- the learned policy choosing the action (scan_eval_pair_policy3 in the half-wave layout) ----
    __device__ __forceinline__ Pol make_pol(const EngineParams& p, int lane, int A, uint32_t pol0, uint32_t pol1, uint32_t plist) const {
        Pol q;
        q.pol0 = pol0;
        q.dpol = pol1 - pol0;
        q.list = plist;
        q.logits = plist + (uint32_t)((4 * p.pcap + 15) & ~15);      // behind the 2-byte descriptors of both episodes
        const int H = p.pH;
        const uint32_t pb = lane >= 32 ? pol1 : pol0;
        const uint32_t slots = (uint32_t)(A * H) * 4u;
        q.hs = pb + (uint32_t)(ag * H) * 4u;
        q.sl = pb + slots + 16u * (uint32_t)ag;
        q.slr = pb + slots + 16u * (uint32_t)(lane & 7);
        q.w2a = p.pw2 + (size_t)ag * p.N * H;
        q.b2a = p.pb2 + (size_t)ag * p.N;
        q.thr = p.pbound ? (-18.420680744f + __logf((float)p.N) + p.pbound[ag]) : -3.0e38f;
        return q;
    }
    // one logit: the neighbour's row of w2 (L2) against the agent's hidden vector (LDS), the reference's summation order
    template <int BATCH>
    static __device__ __forceinline__ float logit_of(const Pol& pl, uint32_t nb, int H) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        float l = pl.b2a[nb];
        const f4* r0 = reinterpret_cast<const f4*>(pl.w2a + (size_t)nb * H);
        const int nq = H >> 2;
        int c = 0;
        for (; c + BATCH <= nq; c += BATCH) {
            f4 a[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) a[u] = r0[c + u];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const f4 h0 = *lds_at<f4>(pl.hs + 16u * (uint32_t)(c + u));
                l = fmaf(a[u].x, h0.x, l); l = fmaf(a[u].y, h0.y, l); l = fmaf(a[u].z, h0.z, l); l = fmaf(a[u].w, h0.w, l);
            }
        }
        for (; c < nq; ++c) {
            const f4 a0 = r0[c];
            const f4 h0 = *lds_at<f4>(pl.hs + 16u * (uint32_t)c);
            l = fmaf(a0.x, h0.x, l); l = fmaf(a0.y, h0.y, l); l = fmaf(a0.z, h0.z, l); l = fmaf(a0.w, h0.w, l);
        }
        return l;
    }
    template <int NR>
    __device__ __forceinline__ void eval_policy(const In& g, const Pol& pl, int H, int N, int lane, const float* w2_all,
                                                const float* b2_all, uint32_t x_own, int& act_v, int& cost_v, int& quirk_cnt,
                                                float& logp_v) {
        // agent slot (16 B): [0..7] the winning entry — (ordered Gumbel key) << 32 | node | column << 10 | edge cost << 16,
        // 0 = no affordable entry —, [8] max logit (ordered int), [12] sum of exp(logit - max)
        if (lanes(kAgentSlots)) {
            lds_at<int>(selr)[1] = 0;                                                     // position-reward count
            typedef int v4i __attribute__((ext_vector_type(4)));
            *lds_at<v4i>(pl.slr) = (v4i){0, 0, (int)0x80000000, 0};
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) *lds_at<uint8_t>(prev[k]) = 0;
        uint64_t bo[NC];
        uint32_t nb[NC];
        uint32_t gf = 0, qf = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int w = (int)(g.ent[k] >> 16);
            bo[k] = bal(w <= g.ma) & on_m;
            const uint64_t bq = bal(w <= g.mq) & on_m;
            gf |= ((uint32_t)(bo[k] >> gsh) & kField) << (k * GW);
            qf |= ((uint32_t)(bq >> gsh) & kField) << (k * GW);
            const bool own = lanes(bo[k]);
            nb[k] = own ? (g.ent[k] & 0xffffu) : 0u;
            const uint32_t n = own ? row + nb[k] : scratch;
            *lds_at<uint8_t>(n) = 1;
            prev[k] = n;
        }
        // ---- logits of the affordable entries, COOPERATIVELY.  Round 2 gave every scan lane its own 64-term dot product:
        // 16 requests of 16 bytes per lane from up to 64 different rows, four dependent L2 round trips per column.  Here
        // the entries of the pair are compacted into a list (2 bytes each: node | agent << 8 | half << 11) and every group
        // of 8 lanes takes one entry per round: its lanes read the entry's w2 row as 16-byte pieces (piece j, j + 8, ...:
        // whole rows, eight per instruction), FMAs against the agent's hidden vector in LDS, a DPP sum over the 8 lanes;
        // the loads of a few rounds are in flight together, so ~40 entries cost two round trips, not four to eight.
        float l[NC];
        {
            typedef float f4 __attribute__((ext_vector_type(4)));
            uint32_t eidx[NC];
            float bias[NC];                                                    // b2 of my entries: requested now, needed after the rounds
#pragma unroll
            for (int k = 0; k < NC; ++k) bias[k] = pl.b2a[nb[k]];
            uint32_t ne = 0;                                                   // wave-uniform
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                eidx[k] = ne + __builtin_amdgcn_mbcnt_hi((uint32_t)(bo[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bo[k], 0u));
                ne += (uint32_t)__popcll(bo[k]);
                if (lanes(bo[k])) *lds_at<uint16_t>(pl.list + 2u * eidx[k]) = (uint16_t)(nb[k] | ((uint32_t)ag << 8) | ((uint32_t)(lane & 32) << 6));
            }
            wave_lds_fence();
            const uint32_t q = (uint32_t)lane >> 3, j = (uint32_t)lane & 7u;   // 8 lanes per entry, 8 entries per round
            const uint32_t nq = (uint32_t)H >> 2;                              // 16-byte pieces per row (<= 32): lane j takes j, j + 8, ...
            auto rounds = [&](auto pieces, auto batch) {
                constexpr int PPL = decltype(pieces)::value;                   // pieces per lane
                constexpr int RB = decltype(batch)::value;                     // rounds whose loads are in flight together
                for (uint32_t r0 = 0; 8u * r0 < ne; r0 += RB) {
                    f4 w[RB][PPL];
                    uint32_t hsa[RB];
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const uint32_t e = 8u * (r0 + (uint32_t)u) + q;
                        const uint32_t d = e < ne ? (uint32_t)*lds_at<uint16_t>(pl.list + 2u * e) : 0u;
                        const uint32_t node = d & 255u, agu = (d >> 8) & 7u;
                        hsa[u] = pl.pol0 + (d >> 11) * pl.dpol + (agu * (uint32_t)H + 4u * j) * 4u;
                        const float* rp = w2_all + ((size_t)(agu * (uint32_t)N + node) * H + 4u * j);
#pragma unroll
                        for (int m = 0; m < PPL; ++m)
                            w[u][m] = (j + 8u * m < nq) ? *reinterpret_cast<const f4*>(rp + 32 * m) : (f4){0.0f, 0.0f, 0.0f, 0.0f};
                    }
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const uint32_t e = 8u * (r0 + (uint32_t)u) + q;
                        float part = 0.0f;
#pragma unroll
                        for (int m = 0; m < PPL; ++m) {
                            if (j + 8u * m < nq) {
                                const f4 h = *lds_at<f4>(hsa[u] + 128u * m);
                                part = fmaf(w[u][m].x, h.x, part); part = fmaf(w[u][m].y, h.y, part);
                                part = fmaf(w[u][m].z, h.z, part); part = fmaf(w[u][m].w, h.w, part);
                            }
                        }
                        part += dpp_mov<0xB1>(part);                          // my 8 lanes: the quad (lane ^ 1, lane ^ 2), then the other quad
                        part += dpp_mov<0x4E>(part);
                        part += dpp_mov<0x141>(part);                         // row_half_mirror: lane i <-> 7 - i
                        if (j == 0u && e < ne) *lds_at<float>(pl.logits + 4u * e) = part;
                    }
                }
            };
            const uint32_t ppl = (nq + 7u) >> 3;
            if (ppl <= 1u) rounds(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
            else if (ppl == 2u) rounds(std::integral_constant<int, 2>{}, std::integral_constant<int, SY_COOP_BATCH>{});
            else if (ppl == 3u) rounds(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
            else rounds(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < NC; ++k) l[k] = lanes(bo[k]) ? *lds_at<float>(pl.logits + 4u * eidx[k]) + bias[k] : 0.0f;
        }
        // Gumbel-max draw: a cheap per-entry hash (ELL column) of the agent's Philox word of this step
        auto gumbel = [](uint32_t x, uint32_t column) {
            uint32_t h = x ^ (column * 0x9E3779B9u) ^ 0x85EBCA6Bu;
            h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
            const float u = fminf(((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f), 0x1.fffffep-1f);   // (h >> 8 = 2^24 - 1 rounds to 1.0f: -log(-log 1) = +inf)
            return -__logf(-__logf(u));
        };
        // the winner travels WITH its key: one 64-bit LDS max per entry, payload in the low word (no second hand-off)
        auto entry_word = [&](float key, int k) {
            const uint32_t uk = (uint32_t)f32_ordered(key) ^ 0x80000000u;               // unsigned order of the keys
            const uint32_t pay = (g.ent[k] & 0x3ffu) | ((uint32_t)(k * GW + col) << 10) | (g.ent[k] & 0xffff0000u);
            return ((uint64_t)uk << 32) | pay;
        };
        // ---- phase 1: the group's largest logit and its winning key, together
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            if (lanes(bo[k])) {
                atomicMax(lds_at_generic<int>(pl.sl + 8u), f32_ordered(l[k]));
                atomicMax(lds_at_generic<unsigned long long>(pl.sl), (unsigned long long)entry_word(l[k] + gumbel(g.xa, (uint32_t)(k * GW + col)), k));
            }
        }
        wave_lds_fence();
        float Lm = ordered_f32(*lds_at<int>(pl.sl + 8u));
        // ---- phase 2: the sum of exp(logit - max), in 2^-26 fixed point: every term is in (0, 1] and there are at most 16,
        // so the sum fits 31 bits; an integer LDS add runs ~20x faster than ds_add_f32 on gfx950
        // (tools/probes/lds_atomic_probe.hip) and makes the sum — and with it the recorded log-probability — independent of
        // the order the lanes arrive in
        constexpr float kSumScale = 67108864.0f, kSumInv = 1.0f / 67108864.0f;
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (lanes(bo[k])) atomicAdd(lds_at_generic<unsigned int>(pl.sl + 12u), (unsigned int)(__expf(l[k] - Lm) * kSumScale + 0.5f));
        if (lanes(lead_m)) lds_at<int>(selw)[1] = __popc(qf);
        wave_lds_fence();
        // ---- the reference's underflow rule (mappo_agent.py:123-134), exact only where the cheap bound cannot rule it out
        {
            const float S = (float)*lds_at<unsigned int>(pl.sl + 12u) * kSumInv;
            const bool lead = lanes(lead_m);
            const uint64_t sus = bal(lead && gf != 0u && !(Lm + __logf(S) > pl.thr));
#ifndef SY_POL_NO_FALLBACK
            if (sus != 0ull) {                        // rare: evaluate the suspicious actors exactly, one (episode, agent) at a time
                uint64_t fb = 0ull, todo = sus;       // groups (leader-lane bits) that fall back to uniform over the mask
                while (todo != 0ull) {
                    const int ll = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const int agu = rdlane(ag, ll);
                    const uint32_t hsu = (uint32_t)rdlane((int)pl.hs, ll);
                    const float lse = __int_as_float(rdlane(__float_as_int(Lm + __logf(S)), ll));
                    const float mass = exact_legal_mass<NR>(w2_all + (size_t)agu * N * H, b2_all + (size_t)agu * N, hsu, H, N, lane, lse);
                    if (mass <= 1e-8f) fb |= 1ull << ll;
                }
                if (fb != 0ull) {                     // uniform over the mask: logits 0 -> keys = the Gumbel noise alone; redo the slot
                    const bool f = ((fb >> (lane - col)) & 1ull) != 0ull;            // my group's leader bit -> my fallback flag
                    if (f && lanes(lead_m)) {
                        typedef int v4i __attribute__((ext_vector_type(4)));
                        *lds_at<v4i>(pl.sl) = (v4i){0, 0, f32_ordered(0.0f), (int)((unsigned int)__popc(gf) << 26)};
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int k = 0; k < NC; ++k)
                        if (f && lanes(bo[k]))
                            atomicMax(lds_at_generic<unsigned long long>(pl.sl), (unsigned long long)entry_word(gumbel(g.xa, (uint32_t)(k * GW + col)), k));
                    wave_lds_fence();
                }
            }
#endif
        }
        // ---- the agent lanes read their slot: winner, max, sum -> action, cost, log-probability
        {
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i sv = *lds_at<v4i>(pl.slr);
            const uint32_t pay = (uint32_t)sv.x, uk = (uint32_t)sv.y;
            const bool any = (pay | uk) != 0u;
            const float keyw = ordered_f32((int)(uk ^ 0x80000000u));
            const float lw = keyw - gumbel(x_own, (pay >> 10) & 31u);                     // the winner's logit back from its key
            act_v = any ? (int)(pay & 0x3ffu) : -1;
            cost_v = any ? (int)(pay >> 16) : 0;
            logp_v = any ? (lw - ordered_f32(sv.z)) - __logf((float)(unsigned int)sv.w * kSumInv) : 0.0f;
            quirk_cnt = lds_at<int>(selr)[1];
        }
        wave_lds_fence();
    }
};

// ---- the move wave -----------------------------------------------------------------------------
template <int NR, bool REC, int PT, bool POL, int HS>   // HS > 0: half-wave neighbour scan with HS columns per lane (no row of the pool wider than HS * GW)
__device__ __forceinline__ void move_wave3(const EngineParams& p, const LdsMap& L, const EnvLds& E, const EnvLds& E1, int lane, int e,
                                           int g, int slot, int T, sy_rollout_buffers out) {
    const int P = PT > 0 ? PT : p.P, A = P + 1;
    const int N = p.N, NS = p.NS, B = p.B;
    const bool live1 = e + 1 < B;
    const bool upper0 = lane >= 32;
    const int a0 = lane & 31;
    const int eh = (upper0 && live1) ? e + 1 : e;
    const bool store_ok = !upper0 || live1;
    const uint16_t* __restrict__ ap = p.apsp + (size_t)g * N * N;
    const uint64_t gid = p.env_id_offset + (uint64_t)eh;
    const int n16 = (A * NS) >> 4;
    const ScanMap sm = make_scan_map<(PT == 0 || PT >= 5)>(lane, p.scan_w);
    const bool one_pass = A <= sm.per_pass;          // (the launcher only picks this kernel for one- or two-pass boards)
    Coefs<true> kc;
    kc.s = L.kc_s + (a0 == 0 ? 0 : 8);
    RewardTabs tb;
    tb.exp_s = L.exp_s; tb.cov_s = L.cov_s; tb.nrc_s = L.nrc_s; tb.nra_s = L.nra_s; tb.px_s = L.px_s;
    tb.exp_g = p.exp_tab; tb.cov_g = p.cov_tab; tb.n_exp = p.n_exp; tb.n_cov = p.n_cov;
    int* const sync_h = upper0 ? E1.sync : E.sync;
    uint8_t* const mrow_h = upper0 ? E1.mrow : E.mrow;
    int* const rec_h = upper0 ? E1.rec_s : E.rec_s;
    uint16_t* const vis_h = upper0 ? E1.vis_s : E.vis_s;
    const uint32_t ring_w = lds_off(upper0 ? E1.ring : E.ring) + (uint32_t)(a0 & 7) * 16u;   // my agent slot inside an entry
    const uint32_t xch_off = lds_off(rec_h) + kSelWord * 4u;   // 8 words: agent positions exchanged inside a step

    int pos_v = a0 < A ? p.st.pos[(size_t)eh * A + a0] : 0;
    int mon_v = a0 < A ? p.st.budget[(size_t)eh * A + a0] : 0;
    int t_v = p.st.t[eh];
    uint32_t sc_v = p.st.step_count[eh];
    uint32_t* const vis32 = reinterpret_cast<uint32_t*>(vis_h);   // 32-bit counters: one returning LDS add per step
    if (!POL)
        for (int i = a0; i < NS; i += 32) vis32[i] = p.st.visits[(size_t)eh * NS + i];
    // in-kernel policy: per-episode scratch behind the episode slices
    const uint32_t pol0 = lds_off(L.env_base) + (uint32_t)p.wpb * (uint32_t)p.wave_lds_bytes + (uint32_t)slot * (uint32_t)p.pslice;
    // (the pair's scratch = two episode slices: [vectors + slots of episode 0][... of episode 1][the pair's entry list])
    const uint32_t pol1 = pol0 + (uint32_t)(A * p.pH) * 4u + 128u;
    const uint32_t plist = pol1 + (uint32_t)(A * p.pH) * 4u + 128u;
    PolLane3 pll;
    float logp_v = 0.0f;
    if (POL) pll = make_pol_lane3(p, sm, lane, A, pol0, pol1);
    int rev_v = p.reveal_k > 0 ? p.reveal_k - (t_v % p.reveal_k) : 0;
    // the draws of a Philox block are used one per step: xw[0] is always the word of the coming step (the words are
    // shifted down after every step instead of being selected by the step count)
    uint32_t xw[4];
    philox4(gid, sc_v >> 2, kPurposeAct, (uint32_t)a0, p.seed_lo, p.seed_hi, xw);
#pragma unroll
    for (uint32_t r = 1; r <= 3; ++r)
        if ((sc_v & 3u) >= r) { xw[0] = xw[1]; xw[1] = xw[2]; xw[2] = xw[3]; }
    rec_h[a0] = 0;
    rec_h[32 + a0] = 0;
    for (int i = lane; i < n16; i += kWave) {
        reinterpret_cast<uint4*>(E.mrow)[i] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(E1.mrow)[i] = make_uint4(0, 0, 0, 0);
    }
    wave_lds_fence();
    // up to 5 agents, random policy (police count fixed at compile time): one episode per half wave in the scan
    constexpr bool HALF = HS > 0 && PT >= 1 && PT <= 7;
    constexpr int GWH = HALF ? half_scan_gw(PT) : kD;
    HalfScan<GWH, (HALF ? HS : 1)> hs;
    if (HALF) hs.init(L, E, E1, lane, A, NS);
    PairScanLane psl = make_pair_scan_lane(E, E1, sm, lane, A, NS);
    PairScanLane psl2 = psl;
    if (!one_pass) psl2 = make_pair_scan_lane(E, E1, sm, lane, A, NS, sm.per_pass);
    int act_v = -1, cost_v = 0, qcnt = 0;
    typename HalfScan<GWH, (HALF ? HS : 1)>::Pol hpl;
    if (HALF && POL) hpl = hs.make_pol(p, lane, A, pol0, pol1, plist);
    if (HALF) {
        const auto g0 = hs.gather(pos_v, mon_v, xw[0]);
        if (POL) {
            policy_hidden_pair3(p, P, A, pos_v, lane, pol0, pol1);
            wave_lds_fence();
            hs.template eval_policy<NR>(g0, hpl, p.pH, N, lane, p.pw2, p.pb2, xw[0], act_v, cost_v, qcnt, logp_v);
        } else {
            hs.eval(g0, act_v, cost_v, qcnt);
        }
    } else {
        const ScanPairIn g0 = scan_gather_pair(L.ell_s, A, sm, 0, pos_v, mon_v, xw[0]);
        if (POL) {           // (the launcher only picks this instance for single-pass boards)
            policy_hidden_pair3(p, P, A, pos_v, lane, pol0, pol1);
            wave_lds_fence();
            scan_eval_pair_policy3<NR>(psl, pll, sm, p.scan_w, p.pH, N, lane, p.pw2, p.pb2, g0, act_v, cost_v, qcnt, logp_v);
        } else if (one_pass) {
            scan_eval_pair1(psl, sm, p.scan_w, g0, act_v, cost_v, qcnt);
        } else {
            const ScanPairIn g1 = scan_gather_pair(L.ell_s, A, sm, sm.per_pass, pos_v, mon_v, xw[0]);
            scan_eval_pair1<true, false>(psl, sm, p.scan_w, g0, act_v, cost_v, qcnt);
            scan_eval_pair1<false, true>(psl2, sm, p.scan_w, g1, act_v, cost_v, qcnt);
        }
    }
    // E[k] = the observation before step k, the action of step k, and what step k - 1 produced: the reward and the
    // outcome marks — or, with the learned policy (the helper evaluates the rewards then), the position-reward counts
    // of the observation and the log-probability of the action
    auto publish = [&](int k, int pos, int act, int mon, double reward, int t_and_flags) {
        if (lanes(kAgentSlots)) {
            typedef int v4i __attribute__((ext_vector_type(4)));
            const int w0 = (pos & 0xffff) | (act << 16);
            if (POL) {
                const int w1 = (lane & 31) == 0 ? (t_and_flags | (qcnt << kMetaCntShift)) : (mon | (qcnt << 16));
                *lds_at<v4i>(ring_w + (uint32_t)(k & (kRing3 - 1)) * kEntry3) = (v4i){w0, w1, __float_as_int(logp_v), 0};
            } else {
                const int w1 = (lane & 31) == 0 ? t_and_flags : mon;
                *lds_at<v4i>(ring_w + (uint32_t)(k & (kRing3 - 1)) * kEntry3) =
                    (v4i){w0, w1, __double2loint(reward), __double2hiint(reward)};
            }
        }
        asm volatile("" ::: "memory");
#ifdef SY_INJECT_LOST_HANDOFF   // fault-injection build (tests only): episode 0 stops publishing after entry 2
        if ((lane & 31) == 0 && !(eh == 0 && k >= 3)) lds_poke(sync_h, k + 1);
#else
        if ((lane & 31) == 0) lds_poke(sync_h, k + 1);
#endif
    };
    publish(0, pos_v, act_v, mon_v, 0.0, t_v & kMetaTimeMask);

    // mask record cursor: a uniform base advanced once per step + constant 32-bit lane offsets
    const uint32_t off_mask = (uint32_t)eh * (uint32_t)(A * NS) + (uint32_t)a0 * 16u;
    const size_t mask_step = (size_t)B * A * NS;
    const int mc0 = a0 < n16 ? a0 : n16 - 1, mc1 = a0 + 32 < n16 ? a0 + 32 : n16 - 1, mc2 = a0 + 64 < n16 ? a0 + 64 : n16 - 1;
    const uint32_t mc_base = (uint32_t)eh * (uint32_t)(A * NS);
    const uint32_t mc_off0 = mc_base + (uint32_t)mc0 * 16u, mc_off1 = mc_base + (uint32_t)mc1 * 16u, mc_off2 = mc_base + (uint32_t)mc2 * 16u;
    const uint32_t mc_lds0 = lds_off(mrow_h) + (uint32_t)mc0 * 16u, mc_lds1 = lds_off(mrow_h) + (uint32_t)mc1 * 16u,
                   mc_lds2 = lds_off(mrow_h) + (uint32_t)mc2 * 16u;
    double rew = 0.0;
    int term_v = 0, trunc_v = 0, win_v = 0;
    const uint64_t POLM = (((1ull << P) - 1ull) << 1) * 0x0000000100000001ull;   // police lanes of both halves
    const uint32_t POL32 = (uint32_t)POLM;                                       // ... of one half


#ifdef SY_ENDTIMES3
    const unsigned long long et_t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t et_restarts = 0, et_conflicts = 0;
#endif
#ifndef SY_NO_PRIO_TURNS
    const int pslot = slot ^ 8;                                   // the move wave that shares my SIMD: wave w ^ 4 = pair slot ^ 8
    const bool partner_ok = pslot < p.wpb && (int)(blockIdx.x * p.wpb) + pslot < B;
    const int* partner_sync = reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(E.sync) + (ptrdiff_t)(pslot - slot) * p.wave_lds_bytes);
#endif
    S3_DECL
    for (int s = 0; s < T; ++s) {
        int ln = lane;                               // laundered: lane predicates are recomputed every step
        asm volatile("" : "+v"(ln));
        const bool upper = ln >= 32;
        const int a = ln & 31;
        const bool is_pol = a >= 1 && a <= P;
        S3(7)
#ifndef SY_NO_PRIO_TURNS
        // The two move waves that share a SIMD (waves w and w ^ 4 of the block) level each other: every 8 steps a wave looks
        // at its partner's published entry counter and takes the HIGHER issue priority if it is behind (ties: turns).  Left
        // alone the arbiter serves the older wave first — tools/endtimes.py: waves 0-3 of every block finished a launch at
        // 390 us, waves 4-7 at 450 us, and a launch takes as long as the starved half.  Levelled, both finish at 440 us and
        // the spread of all wave run times falls from 32 to 12 us: a launch ends 6-7 % earlier.  Move waves use priorities 3 / 2,
        // the helpers level each other the same way one class below (1 / 0: another 0.5 %).
        if ((s & 7) == 0) {
            const int other = partner_ok ? lds_peek(partner_sync) : s;
            const bool ahead = s > other || (s == other && ((((unsigned)s >> 3) ^ ((unsigned)slot >> 3)) & 1u) != 0u);
            if (ahead) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
        }
#endif

        // ---- moves (yard.py:161-243), both episodes at once
        const int tgt_v = act_v >= 0 ? act_v : pos_v;
        const uint64_t SK = bal(act_v == -1) | bal(mon_v == 0);               // skipped agents (:210-215)
        {   // MrX vs PRE-move police (:180-188)
            const int t_lo = rdlane(tgt_v, 0), t_hi = rdlane(tgt_v, 32);
            const uint32_t hit_lo = (uint32_t)bal(pos_v == t_lo) & POL32, hit_hi = (uint32_t)(bal(pos_v == t_hi) >> 32) & POL32;
            const uint64_t mrx_moves = (uint64_t)z31(hit_lo) | ((uint64_t)z31(hit_hi) << 32);   // lanes 0 / 32
            pos_v = lanes(mrx_moves) ? tgt_v : pos_v;
        }
        uint64_t CF = 0;   // any police pair that could interact this step?  INV: the officers of those pairs (both indices)
#ifdef SY_CONFLICT_ALL
#define SY_PAIR(D) if (D < P) CF |= pair_conflicts<D>(tgt_v, pos_v) & (POLM & (POLM << D));
#else
        uint64_t INV = 0;
#define SY_PAIR(D) if (D < P) { const uint64_t c = pair_conflicts<D>(tgt_v, pos_v) & (POLM & (POLM << D)); CF |= c; INV |= c | (c >> D); }
#endif
        SY_PAIR(1) SY_PAIR(2) SY_PAIR(3) SY_PAIR(4) SY_PAIR(5) SY_PAIR(6)
#undef SY_PAIR
#ifdef SY_ENDTIMES3
        et_conflicts += CF != 0ull ? 1u : 0u;
#endif
        if (CF == 0ull) {                              // no police collision in either episode: order cannot matter
            SY_HOT(m_moves);
            const uint64_t mv = POLM & ~SK & bal(tgt_v != pos_v);
            pos_v = lanes(mv) ? tgt_v : pos_v;
            mon_v -= lanes(mv) ? cost_v : 0;                                  // :234-236
        } else {                                      // exact sequential order (:191-243), harmless for a clean half
            const bool skip_v = lanes(SK);
#ifndef SY_CONFLICT_ALL
            // A third of the steps get here (tools/endtimes.py: the random policy keeps officers next to each other, and
            // a target that is another officer's node or target is an interaction).  Officers outside every flagged pair
            // share no node with anybody: they move as on the fast path, all at once; only the officers of flagged pairs —
            // usually two — take the sequential loop, in index order (an officer that already moved finds its own node
            // occupied and stays: the loop body is idempotent for it).
            {
                const uint64_t mv = POLM & ~SK & ~INV & bal(tgt_v != pos_v);
                pos_v = lanes(mv) ? tgt_v : pos_v;
                mon_v -= lanes(mv) ? cost_v : 0;
            }
            const uint32_t inv_any = (uint32_t)INV | (uint32_t)(INV >> 32);      // involved in either episode of the pair
#endif
            for (int k = 1; k <= P; ++k) {
#ifndef SY_CONFLICT_ALL
                if (((inv_any >> k) & 1u) == 0u) continue;
#endif
                const int tk = hbcast(tgt_v, k, upper);
                const bool occ = hany(is_pol && pos_v == tk, upper);          // own node included (:231)
                if (!occ && !skip_v && a == k) {
                    pos_v = tk;
                    mon_v -= cost_v;
                }
            }
        }
        const uint64_t NM = ~half_any8(POLM & ~SK);                           // nobody could act (:191,216)
        // ---- outcome priority (reward_calculator.py:63-90)
        const uint64_t CAP = half_any8(half_pick(bal(pos_v == rdlane(pos_v, 0)), bal(pos_v == rdlane(pos_v, 32))) & POLM);
        const uint64_t TO = bal(t_v > p.max_t);                               // t_v is replicated over its half
        const uint64_t ENDED = CAP | TO | NM;
        const uint64_t NEED = p.auto_reset != 0 ? ENDED : 0ull;
        term_v = lanes(CAP | (NM & ~TO)) ? 1 : 0;
        trunc_v = lanes(TO & ~CAP) ? 1 : 0;
        win_v = lanes(CAP) ? 1 : (lanes(TO | NM) ? 2 : 0);
        int flags_v = term_v | (trunc_v << 1) | (win_v << kFlagWinShift);
        S3(0)
        // ---- a finished episode restarts right here: the one scan below already serves the new episode
        const int pos_m = pos_v;           // post-move nodes: the visit counters and the shaped rewards use these
        // every agent's post-move node goes to LDS now (the result-slot words of the staging row, free until the scan is
        // evaluated): by the time the shortest-path gathers read them back, the write is long done
        if (!POL && lanes(kAgentSlots)) lds_at<int>(xch_off)[a] = pos_m;
        const int t_rew = t_v;             // pre-increment timestep of this step (reward_calculator.py:145,219)
#ifdef SY_ENDTIMES3
        et_restarts += NEED != 0ull ? 1u : 0u;
#endif
        if (NEED != 0ull) {
            const int st = sample_starts_pair(NEED, ln, a, A, N, gid, sc_v + 1u, p.seed_lo, p.seed_hi);
            const int m_init = a == 0 ? SY_MRX_MONEY : (a < A ? p.money0 : 0);     // yard.py:117-119
            pos_v = lanes(NEED) ? st : pos_v;
            mon_v = lanes(NEED) ? m_init : mon_v;
            flags_v |= lanes(NEED) ? kFlagRestart : 0;
            rev_v = lanes(NEED) ? p.reveal_k + 1 : rev_v;
        }
        t_v = lanes(NEED) ? 0 : t_v + 1;   // yard.py:355; a restarted episode begins at 0
        if (p.reveal_k > 0) {              // the post-increment timestep is a multiple of reveal_k (never on a restart)
            rev_v -= 1;
            const bool rv = rev_v == 0;
            rev_v = rv ? p.reveal_k : rev_v;
            flags_v |= (rv && !lanes(NEED)) ? kFlagReveal : 0;
        }
        S3(1)
        // ---- next step's draw and the gather half of the scan of the new state
        const uint32_t nxt_v = sc_v + 1u;
        if (bal((nxt_v & 3u) == 0u) != 0ull) {      // some episode starts a new block (the two of a pair may be out of phase)
            uint32_t nw[4];
            philox4(gid, nxt_v >> 2, kPurposeAct, (uint32_t)a, p.seed_lo, p.seed_hi, nw);
            const bool refill = (nxt_v & 3u) == 0u;
            xw[0] = refill ? nw[0] : xw[1]; xw[1] = refill ? nw[1] : xw[2]; xw[2] = refill ? nw[2] : xw[3]; xw[3] = refill ? nw[3] : xw[3];
        } else {
            xw[0] = xw[1]; xw[1] = xw[2]; xw[2] = xw[3];
        }
        sc_v = nxt_v;
        const uint32_t x_next = xw[0];
        typename HalfScan<GWH, (HALF ? HS : 1)>::In hg;
        ScanPairIn sg, sg2;
        if (HALF) {
            hg = hs.gather(pos_v, mon_v, x_next);
        } else {
            sg = scan_gather_pair(L.ell_s, A, sm, 0, pos_v, mon_v, x_next);
            sg2 = sg;
            if (!one_pass) sg2 = scan_gather_pair(L.ell_s, A, sm, sm.per_pass, pos_v, mon_v, x_next);
        }
        if (POL) policy_hidden_pair3(p, P, A, pos_v, ln, pol0, pol1);      // hidden vectors of the next observation
        S3(2)
        // ---- node_visit_counts (yard.py:244-245); then the LDS reads of this phase issued back to back — every agent's
        // post-move node (for the shortest-path gathers) and the mask rows of the observation before the step (LDS
        // operations of a wave are in order: these reads see the rows before the scan below rewrites them) — one
        // round trip instead of three; then the shortest-path loads, then the mask stores (stores queued ahead of
        // loads would delay them)
        int vc = 0;
        int dm = 0;
        int dj[SY_MAX_AGENTS - 1];
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) dj[j - 1] = 0;
        typedef int v4i __attribute__((ext_vector_type(4)));
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        v4i qa = {0, 0, 0, 0}, qb = {0, 0, 0, 0};
        if (!POL) {
            SY_HOT(m_visits);
            if (is_pol) vc = (int)atomicAdd(vis32 + pos_m, 1u);     // (the count before this visit: + 1 where it is used, so nothing waits here)
            if (NEED != 0ull) {                // a new episode starts from zero (yard.py:85)
                if (lanes(NEED))
                    for (int i = a; i < (NS >> 2); i += 32) reinterpret_cast<uint4*>(vis32)[i] = make_uint4(0, 0, 0, 0);
            }
            qa = *lds_at<v4i>(xch_off);
            qb = *lds_at<v4i>(xch_off + 16u);
        }
        // 16-byte pieces c, c + 32, c + 64 of my episode's rows (pieces past the end are clamped to the last one: a few
        // lanes then store the same bytes to the same address, which is cheaper than masking them off)
        const bool rec_mask = REC && out.mask;
        v4u v0, v1, v2;
        if (REC) { v0 = *lds_at<v4u>(mc_lds0); v1 = *lds_at<v4u>(mc_lds1); v2 = *lds_at<v4u>(mc_lds2); }
        if (!POL) {
            const int q[SY_MAX_AGENTS] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            const uint32_t rowb = (uint32_t)(pos_m * N) * 2u;
            if (is_pol) {
                dm = (int)*at_bytes(ap, rowb + (uint32_t)q[0] * 2u);
#pragma unroll
                for (int j = 1; j < SY_MAX_AGENTS; ++j)
                    if (j <= P) dj[j - 1] = (int)*at_bytes(ap, rowb + (uint32_t)q[j] * 2u);
            }
        }
        if (rec_mask) {
            if (store_ok) {
                SY_HOT(m_maskcopy);
                SY_STREAM_STORE(reinterpret_cast<v4u*>(out.mask + mc_off0), v0);
                SY_STREAM_STORE(reinterpret_cast<v4u*>(out.mask + mc_off1), v1);
                SY_STREAM_STORE(reinterpret_cast<v4u*>(out.mask + mc_off2), v2);
                if (n16 > 96) {
                    uint4* md = reinterpret_cast<uint4*>(out.mask + off_mask);
                    const uint4* mr = reinterpret_cast<const uint4*>(mrow_h) + a;
                    for (int i = 96; a + i < n16; i += 32) md[i] = mr[i];
                }
            }
#ifndef SY_DIAG_NO_ADVANCE     // timing-only build: every step overwrites step 0's rows (same instructions, no HBM stream)
            out.mask += mask_step;
#endif
        }
        S3(3)
        // ---- evaluate half of the scan: masks of the new state, position-reward counts, next action
        if (HALF) {
            if (POL) hs.template eval_policy<NR>(hg, hpl, p.pH, N, ln, p.pw2, p.pb2, x_next, act_v, cost_v, qcnt, logp_v);
            else hs.eval(hg, act_v, cost_v, qcnt);
        } else if (POL) {
            scan_eval_pair_policy3<NR>(psl, pll, sm, p.scan_w, p.pH, N, ln, p.pw2, p.pb2, sg, act_v, cost_v, qcnt, logp_v);
        } else if (one_pass) {
            scan_eval_pair1(psl, sm, p.scan_w, sg, act_v, cost_v, qcnt);
        } else {
            scan_eval_pair1<true, false>(psl, sm, p.scan_w, sg, act_v, cost_v, qcnt);
            scan_eval_pair1<false, true>(psl2, sm, p.scan_w, sg2, act_v, cost_v, qcnt);
        }
        S3(4)
        // ---- rewards of the step (reward_calculator.py:63-90 constants, :94-266 shaped)
        if (!POL) {
            SY_HOT(m_rewards);
            asm volatile("" : "+v"(vc));       // (keeps the "+ 1" — and with it the wait for the LDS add — down here)
            const double shaped = shaped_reward3(tb, a, P, POLM, t_rew, qcnt, is_pol ? vc + 1 : 0, dm, dj, kc);
            rew = lanes(ENDED) ? (lanes(CAP) ? (a == 0 ? -1.0 : 1.0) : (a == 0 ? 1.0 : 0.0)) : shaped;
        }
        S3(6)
        // ---- hand the new state and the step's outcome to the helper.  Back-pressure: every kRing3 / 2 entries make sure
        // the helper is at most kRing3 / 2 entries behind, so the ring cannot be overrun in between.
        const int k = s + 1;
        if ((k & (kRing3 / 2 - 1)) == 0) {
            int spin = 0;
            for (; spin < kSpinMax; ++spin) {
                if (k - lds_peek(E.sync + 1) <= kRing3 / 2) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (spin == kSpinMax) report_status(SY_STATUS_RING_WAIT_EXPIRED);
            asm volatile("" ::: "memory");
        }
        publish(k, pos_v, act_v, mon_v, rew, (t_v & kMetaTimeMask) | (flags_v << kMetaFlagShift));
        S3(5)
    }
    S3_DUMP("move  [moves+outcome, restart, rng+gather, visits+apsp+maskcopy, eval, publish, rewards, loophead]", T)
#ifdef SY_ENDTIMES3    // load-balance builds (tools/endtimes.py): this wave's start / end on the constant 100 MHz clock, left in the
                       // padding bytes [N, NS) of the last recorded mask row of the episode's last agent
    if (REC && out.mask && (lane & 31) == 0 && store_ok && NS - N >= 8) {
        uint32_t* tw = reinterpret_cast<uint32_t*>(out.mask - mask_step + (size_t)eh * A * NS + (size_t)(A - 1) * NS + N);
        tw[0] = (uint32_t)et_t0;
        tw[1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        uint32_t* tc = reinterpret_cast<uint32_t*>(out.mask - mask_step + (size_t)eh * A * NS + (size_t)(A - 2) * NS + N);
        tc[0] = et_restarts;      // steps of this wave on which an episode of the pair restarted
        tc[1] = et_conflicts;     // steps on which the police moves took the exact sequential order
    }
#endif

    // ---- write the live state back; state pointers re-read from the kernel arguments
    const KernargParams kq = kernarg_params();
    if (store_ok) {
        if (a0 < A) {
            kq->st.pos[(size_t)eh * A + a0] = pos_v;
            kq->st.budget[(size_t)eh * A + a0] = mon_v;
            if (!POL) kq->st.reward[(size_t)eh * A + a0] = rew;
        }
        if (a0 == 0) {
            kq->st.t[eh] = t_v;
            kq->st.step_count[eh] = sc_v;
            kq->st.terminated[eh] = (uint8_t)term_v;
            kq->st.truncated[eh] = (uint8_t)trunc_v;
            kq->st.winner[eh] = (int8_t)win_v;
        }
        if (!POL) {
            uint16_t* vis_out = kq->st.visits;
            for (int i = a0; i < NS; i += 32) vis_out[(size_t)eh * NS + i] = (uint16_t)vis32[i];
        }
        uint4* dst = reinterpret_cast<uint4*>(kq->st.mask + (size_t)eh * A * NS);
        for (int i = a0; i < n16; i += 32) dst[i] = reinterpret_cast<const uint4*>(mrow_h)[i];
    }
}

// ---- the helper's belief filter -----------------------------------------------------------------
// Both episodes of the pair in lockstep (they share the board).  Node-major lanes: lane L owns the NR consecutive
// nodes NR * L ... NR * L + NR - 1 (NR = 1, 2 or 4: boards of up to 256 nodes), so a lane's part of a belief row is
// one 16-byte piece: the record row of an episode is ONE global_store_dwordx4 per lane (two per pair and step
// instead of eight dword stores), the scaled vector c = b / deg goes to the LDS scratch as 16-byte writes.  The
// scratch holds both episodes interleaved (8 B per node): one 8-byte gather serves both, sums are packed two-wide,
// in belief_step_pair's order.  The LDS byte addresses of every node's first eight neighbour entries are kept in
// registers (they never change) instead of being unpacked on every step.
// Normalisation: without evidence the diffusion conserves the mass (every node of a connected board has
// neighbours), so the filter renormalises only every 8th step and when it leaves; with police evidence (mass is
// removed, possibly all of it -> uniform fallback) every step, as the reference filter does.
template <int NR, bool LAY>      // LAY: honour the pool's bank-aware scratch layout (the random-policy instances; the policy
                                 // instances, whose helper sits on the VGPR limit and is not LDS-bound, keep node order)
struct BeliefLanes {
    typedef float vNf __attribute__((ext_vector_type(NR == 1 ? 1 : NR)));
    v2f b[NR];
    float ideg[NR];
    uint32_t ga[NR][8];                                    // LDS addresses of the first eight neighbour entries
    int slab_w[NR];
    uint64_t in_m[NR];                                     // lanes whose node k exists
    uint32_t c_off, off_bel;
    uint32_t cw[LAY ? NR : 1];                             // LDS addresses of my nodes' own scratch entries (node order: of the first)
    int j0;
    bool mine;

    __device__ __forceinline__ void load(const EngineParams& p, const LdsMap& L, const EnvLds& E, int lane, int e, int g, bool live1) {
        const int N = p.N, NS = p.NS;
        c_off = lds_off(E.c_s);
        j0 = NR * lane;
        mine = j0 < NS;
        const uint16_t* const slot_of = (LAY && p.bel_gather) ? p.bel_slot + (size_t)g * NS : nullptr;   // (the layout's two tables go together)
        off_bel = ((uint32_t)e * (uint32_t)NS + (uint32_t)j0) * 4u;
        const float* r0 = p.st.belief + (size_t)e * NS + j0;
        const float* r1 = r0 + NS;
        const float* dg = p.inv_deg + (size_t)g * NS + j0;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int j = j0 + k;
            b[k].x = (mine && j < N) ? r0[k] : 0.0f;
            b[k].y = (mine && live1 && j < N) ? r1[k] : 0.0f;
            ideg[k] = (mine && j < N) ? dg[k] : 0.0f;
            const int jr = j < N ? j : N - 1;
            const uint32_t cwk = LAY ? c_off + 8u * (uint32_t)(j < N ? ((slot_of && mine) ? (int)slot_of[j] : j) : N)   // (lanes past the board write their 0 to the zero entry)
                                     : c_off + 8u * (uint32_t)j;
            if (LAY || k == 0) cw[LAY ? k : 0] = cwk;
            const uint4 o = *reinterpret_cast<const uint4*>(L.boff_s + (jr << 4));
            ga[k][0] = c_off + (o.x & 0xffffu); ga[k][1] = c_off + (o.x >> 16);
            ga[k][2] = c_off + (o.y & 0xffffu); ga[k][3] = c_off + (o.y >> 16);
            ga[k][4] = c_off + (o.z & 0xffffu); ga[k][5] = c_off + (o.z >> 16);
            ga[k][6] = c_off + (o.w & 0xffffu); ga[k][7] = c_off + (o.w >> 16);
            const int deg = ideg[k] > 0.0f ? (int)(1.0f / ideg[k] + 0.5f) : 0;
            // Two special cases folded into the gather addresses so that the step needs no per-node selects: lanes past
            // the board gather the zero entry only (their belief stays 0), and a node without neighbours (the mass on
            // it stays: belief_module.py keeps the particle) gathers ITSELF once with weight 1 — b * 1 + zeros == b.
            const uint32_t zero_e = c_off + (uint32_t)N * 8u;
            const bool inr = mine && j < N;
            if (!inr || deg == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) ga[k][q] = zero_e;
                if (inr) { ga[k][0] = cwk; ideg[k] = 1.0f; }
            }
            int need = (deg + 3) >> 2;
#pragma unroll
            for (int o2 = 32; o2 >= 1; o2 >>= 1) {
                const int other = __shfl_xor(need, o2, kWave);
                need = other > need ? other : need;
            }
            slab_w[k] = rdlane(need, 0);
            in_m[k] = bal(j < N);
        }
        if (lane == 0) *lds_at<v2f>(c_off + (uint32_t)N * 8u) = (v2f){0.0f, 0.0f};   // padding entries point here
    }
    // the belief before the step goes to the record: one 16-byte store per lane and episode
    __device__ __forceinline__ void record(float* row_base, int NS, bool live1) const {
        if (mine) {
            vNf v0, v1;
#pragma unroll
            for (int k = 0; k < NR; ++k) { v0[k] = b[k].x; v1[k] = b[k].y; }
            SY_STREAM_STORE(reinterpret_cast<vNf*>(at_bytes(row_base, off_bel)), v0);
            if (live1) SY_STREAM_STORE(reinterpret_cast<vNf*>(at_bytes(row_base, off_bel + (uint32_t)NS * 4u)), v1);
        }
    }
    // one transition: bf = 0 filter step, 1 new episode (prior), 2 reveal (delta on MrX's node)
    __device__ __forceinline__ void step(const LdsMap& L, int N, int P, int bf0, int bf1, int node0, int node1, bool onehot, bool pol_ev,
                                         const int (&pol0)[SY_MAX_AGENTS - 1], const int (&pol1)[SY_MAX_AGENTS - 1], float uni,
                                         bool norm_now) {
        if (bf0 == 0 || bf1 == 0) {
            SY_HOT(h_belstep);
            if (mine) {      // c = b / deg of my nodes into the interleaved scratch
                if constexpr (LAY) {      // each to its own entry (node order, or the pool's bank-aware layout: the 16 lanes of an LDS
                                          // store cycle then hit 16 different bank pairs)
#pragma unroll
                    for (int k = 0; k < NR; ++k) *lds_at<v2f>(cw[k]) = b[k] * ideg[k];
                } else if constexpr (NR == 1) {      // node order: NR * 8 contiguous bytes
                    *lds_at<v2f>(cw[0]) = b[0] * ideg[0];
                } else {
                    typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int k = 0; k + 1 < NR; k += 2) {
                        const v2f c0 = b[k] * ideg[k], c1 = b[k + 1] * ideg[k + 1];
                        *lds_at<v4f>(cw[0] + (uint32_t)k * 8u) = (v4f){c0.x, c0.y, c1.x, c1.y};
                    }
                }
            }
            wave_lds_fence();
            constexpr int GR = NR < 2 ? NR : 2;
            v2f tot = {0.0f, 0.0f};
#pragma unroll
            for (int r0 = 0; r0 < NR; r0 += GR) {
                v2f gq[GR][8];
#pragma unroll
                for (int q = 0; q < GR; ++q)
#pragma unroll
                    for (int k = 0; k < 8; ++k) gq[q][k] = *lds_at<v2f>(ga[r0 + q][k]);
#pragma unroll
                for (int q = 0; q < GR; ++q) {
                    const int r = r0 + q;
                    const int j = j0 + r;
                    v2f acc = {0.0f, 0.0f};
                    acc += ((gq[q][0] + gq[q][1]) + (gq[q][2] + gq[q][3])) + ((gq[q][4] + gq[q][5]) + (gq[q][6] + gq[q][7]));
                    if (slab_w[r] > 2) {            // wave-uniform: some node of this group has more than 8 neighbours
                        const int jr = j < N ? j : N - 1;
                        const uint4 o2 = *reinterpret_cast<const uint4*>(L.boff_s + (jr << 4) + 8);
                        const uint32_t co = c_off;
                        auto ld = [co](uint32_t off) { return *lds_at<v2f>(co + off); };
                        const v2f x0 = ld(o2.x & 0xffffu), x1 = ld(o2.x >> 16), x2 = ld(o2.y & 0xffffu), x3 = ld(o2.y >> 16);
                        const v2f x4 = ld(o2.z & 0xffffu), x5 = ld(o2.z >> 16), x6 = ld(o2.w & 0xffffu), x7 = ld(o2.w >> 16);
                        acc += ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
                        acc = lanes(in_m[r]) ? acc : (v2f){0.0f, 0.0f};       // (lanes past the board read a clamped row here)
                    }
                    if (pol_ev) {
#pragma unroll
                        for (int k = 0; k < SY_MAX_AGENTS - 1; ++k) {
                            if (k < P && j == pol0[k]) acc.x = 0.0f;
                            if (k < P && j == pol1[k]) acc.y = 0.0f;
                        }
                    }
                    b[r] = acc;
                    tot += acc;
                }
            }
            if (norm_now) {
                const float t0 = wave_sum(tot.x), t1 = wave_sum(tot.y);
                const v2f scale = {t0 == 0.0f ? 0.0f : __builtin_amdgcn_rcpf(t0), t1 == 0.0f ? 0.0f : __builtin_amdgcn_rcpf(t1)};
                const v2f offs = {t0 == 0.0f ? uni : 0.0f, t1 == 0.0f ? uni : 0.0f};
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    b[r] = __builtin_elementwise_fma(b[r], scale, offs);
                    b[r] = lanes(in_m[r]) ? b[r] : (v2f){0.0f, 0.0f};
                }
            }
            wave_lds_fence();
        }
        if (bf0 != 0) {   // new episode -> prior, reveal -> delta on MrX's node (wave-uniform branches)
            const bool delta = bf0 == 2 || onehot;
#pragma unroll
            for (int r = 0; r < NR; ++r) b[r].x = lanes(in_m[r]) ? (delta ? (j0 + r == node0 ? 1.0f : 0.0f) : uni) : 0.0f;
        }
        if (bf1 != 0) {
            const bool delta = bf1 == 2 || onehot;
#pragma unroll
            for (int r = 0; r < NR; ++r) b[r].y = lanes(in_m[r]) ? (delta ? (j0 + r == node1 ? 1.0f : 0.0f) : uni) : 0.0f;
        }
    }
    __device__ __forceinline__ void finish(float* bel_out, int NS, int e, bool live1, bool renorm) {
        if (renorm) {      // leave a normalised belief behind (the filter renormalises lazily)
            v2f tot = {0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < NR; ++r) tot += b[r];
            const float t0 = wave_sum(tot.x), t1 = wave_sum(tot.y);
            const v2f scale = {t0 == 0.0f ? 1.0f : 1.0f / t0, t1 == 0.0f ? 1.0f : 1.0f / t1};
#pragma unroll
            for (int r = 0; r < NR; ++r) b[r] = b[r] * scale;
        }
        if (mine) {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                bel_out[(size_t)e * NS + j0 + k] = b[k].x;
                if (live1) bel_out[(size_t)(e + 1) * NS + j0 + k] = b[k].y;
            }
        }
    }
};

// ---- the helper wave ---------------------------------------------------------------------------
// Everything that only leaves the chip: the belief filter with its record rows, and the packed record row of every
// transition {reward, pos, budget, action, t, flags} (the move wave hands over the reward it computed).  POL (the
// move wave evaluates the learned policy): the helper also counts visits, gathers the shortest paths and evaluates
// the rewards itself (entry k + 1 holds the post-move nodes and the position-reward counts), and records the
// log-probabilities.
template <int NR, bool REC, int PT, bool POL>
__device__ __forceinline__ void helper_wave3(const EngineParams& p, const LdsMap& L, const EnvLds& E, const EnvLds& E1, int lane, int e,
                                             int g, int T, sy_rollout_buffers out) {
    const int P = PT > 0 ? PT : p.P, A = P + 1;
    const int N = p.N, NS = p.NS, B = p.B;
    const bool live1 = e + 1 < B;
    const bool upper0 = lane >= 32;
    const int a0 = lane & 31;
    const int eh = (upper0 && live1) ? e + 1 : e;
    const bool store_ok = !upper0 || live1;
    const uint32_t ring_h = lds_off(upper0 ? E1.ring : E.ring);
    const uint32_t ring_r = ring_h + (uint32_t)(a0 & 7) * 16u;              // my agent slot inside an entry
    // POL: the reward side of the step
    const uint16_t* __restrict__ ap = p.apsp + (size_t)g * N * N;
    Coefs<true> kc;
    kc.s = L.kc_s + (a0 == 0 ? 0 : 8);
    RewardTabs tb;
    tb.exp_s = L.exp_s; tb.cov_s = L.cov_s; tb.nrc_s = L.nrc_s; tb.nra_s = L.nra_s; tb.px_s = L.px_s;
    tb.exp_g = p.exp_tab; tb.cov_g = p.cov_tab; tb.n_exp = p.n_exp; tb.n_cov = p.n_cov;
    uint32_t* const vis32 = reinterpret_cast<uint32_t*>(upper0 ? E1.vis_s : E.vis_s);
    if (POL)
        for (int i = a0; i < NS; i += 32) vis32[i] = p.st.visits[(size_t)eh * NS + i];
    const uint64_t POLM = (((1ull << P) - 1ull) << 1) * 0x0000000100000001ull;
    double rew = 0.0;
    float logp0_v = 0.0f;
    const bool has_belief = p.st.belief != nullptr;
    BeliefLanes<NR, !POL> bl;
    if (has_belief) bl.load(p, L, E, lane, e, g, live1);
    wave_lds_fence();
    const bool rec_bel = REC && has_belief && out.belief != nullptr;
    const bool onehot = p.belief_onehot != 0, pol_ev = p.police_ev != 0;
    const float uni = 1.0f / (float)N;
    const size_t bel_step = (size_t)B * NS;
    const int RW = p.rec_words;
    const uint32_t off_rec = (uint32_t)eh * (uint32_t)RW * 4u;              // byte offset of my episode's row inside a step

    typedef int v4i __attribute__((ext_vector_type(4)));
    auto wait_entry = [&](int k) {      // entries 0 .. k are published once produced > k
        if (lds_peek(E.sync) <= k) {    // (usually there already: the bounded wait stays off the common path)
            int spin = 0;
            for (; lds_peek(E.sync) <= k && spin < kSpinMax; ++spin) __builtin_amdgcn_s_sleep(1);
            if (spin == kSpinMax) report_status(SY_STATUS_BELIEF_WAIT_EXPIRED);
        }
        asm volatile("" ::: "memory");
    };
    // the outcome words of a record row {t, terminated, truncated, winner, 0 ...}: lane a takes (marks >> shift) & mask
    const int mw_shift = a0 == 2 ? 1 : (a0 == 3 ? kFlagWinShift : 0), mw_mask = (a0 == 1 || a0 == 2) ? 1 : (a0 == 3 ? 3 : 0);

    int pos0_v, act0_v, mon0_v, t0_v;
    {
        wait_entry(0);
        const v4i w = *lds_at<v4i>(ring_r);
        pos0_v = w.x & 0xffff;
        act0_v = w.x >> 16;
        mon0_v = (lane & 31) == 0 ? SY_MRX_MONEY : w.y;
        const int m_lo = rdlane(w.y, 0), m_hi = rdlane(w.y, 32);
        t0_v = (upper0 ? m_hi : m_lo) & kMetaTimeMask;
        if (POL) {
            mon0_v = (lane & 31) == 0 ? SY_MRX_MONEY : (w.y & 0xffff);
            logp0_v = __int_as_float(w.z);
        }
        asm volatile("" ::: "memory");
        if (lane < 2) lds_poke(lane == 0 ? E.sync + 1 : E1.sync + 1, 1);
    }

    S3_DECL
    for (int s = 0; s < T; ++s) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const bool upper = ln >= 32;
        const int a = ln & 31;
        S3(7)
#ifndef SY_NO_PRIO_TURNS   // the helper that is behind its SIMD partner helper goes first (priorities 1 / 0, below the move waves)
        if ((s & 7) == 0) {
            const int hslot = e - (int)(blockIdx.x * p.wpb);
            const int ps = hslot ^ 8;
            const bool ok = ps < p.wpb && (int)(blockIdx.x * p.wpb) + ps < B;
            const int other = ok ? lds_peek(reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(E.sync + 1) + (ptrdiff_t)(ps - hslot) * p.wave_lds_bytes)) : s;
            if (s > other || (s == other && (hslot & 8))) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
        }
#endif
#ifdef SY_DIAG_H_IDLE     // timing-only build: the helper just consumes the entries (the move wave's chain alone)
        {
            wait_entry(s + 1);
            const v4i w = *lds_at<v4i>(ring_r + (uint32_t)((s + 1) & (kRing3 - 1)) * kEntry3);
            asm volatile("" :: "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));
            asm volatile("" ::: "memory");
            if (lane < 2) lds_poke(lane == 0 ? E.sync + 1 : E1.sync + 1, s + 2);
            continue;
        }
#endif
        if (rec_bel) {      // the belief before the step goes to the record
            SY_HOT(h_belrec);
            bl.record(out.belief, NS, live1);
#ifndef SY_DIAG_NO_ADVANCE
            out.belief += bel_step;
#endif
        }
        S3(0)
        // ---- what step s produced: entry s + 1 (the next observation, the reward, the outcome marks)
        wait_entry(s + 1);
        S3(1)
        const uint32_t ent = (uint32_t)((s + 1) & (kRing3 - 1)) * kEntry3;
        const v4i w = *lds_at<v4i>(ring_r + ent);
        int q[SY_MAX_AGENTS];                 // POL: every agent's post-move node of my half (the entry's slots)
#pragma unroll
        for (int j = 0; j < SY_MAX_AGENTS; ++j) q[j] = POL ? (*lds_at<int>(ring_h + ent + 16u * (uint32_t)j) & 0xffff) : 0;
        asm volatile("" ::: "memory");
        if (lane < 2) lds_poke(lane == 0 ? E.sync + 1 : E1.sync + 1, s + 2);   // entry copied: the slot may be reused
        const int pos1_v = w.x & 0xffff, act1_v = w.x >> 16;                   // arithmetic shift: 0xffff -> -1
        const int m_lo = rdlane(w.y, 0), m_hi = rdlane(w.y, 32);
        const int meta_v = upper ? m_hi : m_lo;
        const int mon1_v = a == 0 ? SY_MRX_MONEY : (POL ? (w.y & 0xffff) : w.y);
        const int fl_v = meta_v >> kMetaFlagShift;
        const int f_lo = m_lo >> kMetaFlagShift, f_hi = m_hi >> kMetaFlagShift;     // wave-uniform copies
        S3(2)
        int rw_lo = w.z, rw_hi = w.w;         // the float64 reward of the transition (from the move wave, or evaluated here)
        int dm = 0, vc = 0;
        int dj[SY_MAX_AGENTS - 1];
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) dj[j - 1] = 0;
        bool do_shaped = false;
        if (POL) {
            const bool is_pol = a >= 1 && a <= P;
            const bool restart_v = (fl_v & kFlagRestart) != 0;
            do_shaped = ((f_lo & (kFlagTerm | 2)) == 0) || ((f_hi & (kFlagTerm | 2)) == 0);
            // shortest paths between the post-move nodes (reward_calculator.py:126-202): issued now, used after the belief
            if (do_shaped && is_pol) {
                const uint32_t rowb = (uint32_t)(pos1_v * N) * 2u;
                dm = (int)*at_bytes(ap, rowb + (uint32_t)q[0] * 2u);
#pragma unroll
                for (int j = 1; j < SY_MAX_AGENTS; ++j)
                    if (j <= P) dj[j - 1] = (int)*at_bytes(ap, rowb + (uint32_t)q[j] * 2u);
            }
            // node_visit_counts (yard.py:244-245): post-move police nodes; a new episode starts from zero (yard.py:85)
            if (((f_lo | f_hi) & kFlagRestart) != 0) {
                if (restart_v)
                    for (int i = a; i < (NS >> 2); i += 32) reinterpret_cast<uint4*>(vis32)[i] = make_uint4(0, 0, 0, 0);
            }
            if (is_pol && !restart_v) vc = (int)atomicAdd(vis32 + pos1_v, 1u) + 1;
        }
        // ---- belief: new episode -> prior, reveal -> delta on MrX's node, else one filter step
        if (has_belief) {
            const int node0 = rdlane(pos1_v, 0), node1 = rdlane(pos1_v, 32);
            // new episode -> 1, reveal -> 2, else 0: the two marks are adjacent bits and never set together
            static_assert(kFlagRestart == 16 && kFlagReveal == 32, "bf = (flags >> 4) & 3");
            const int bf0 = (f_lo >> 4) & 3, bf1 = (f_hi >> 4) & 3;
            int pol0[SY_MAX_AGENTS - 1], pol1[SY_MAX_AGENTS - 1];
#pragma unroll
            for (int k = 0; k < SY_MAX_AGENTS - 1; ++k) pol0[k] = pol1[k] = -1;
            if (pol_ev) {
#pragma unroll
                for (int k = 0; k < SY_MAX_AGENTS - 1; ++k) {
                    if (k < P) {
                        pol0[k] = rdlane(pos1_v, 1 + k);
                        pol1[k] = rdlane(pos1_v, 33 + k);
                    }
                }
            }
#ifdef SY_BELIEF_ALWAYS_NORM
            const bool norm_now = true;
#else
            const bool norm_now = pol_ev || ((s & 7) == 7);
#endif
            bl.step(L, N, P, bf0, bf1, node0, node1, onehot, pol_ev, pol0, pol1, uni, norm_now);
        }
        S3(6)
        if (POL) {      // rewards of the step (reward_calculator.py:63-90 constants, :94-266 shaped)
            const int qcnt = a == 0 ? ((meta_v >> kMetaCntShift) & 31) : (int)((uint32_t)w.y >> 16);
            const bool ended_v = (fl_v & (kFlagTerm | 2)) != 0;
            const bool cap_v = ((fl_v >> kFlagWinShift) & 3) == 1;
            double shaped = 0.0;
            if (do_shaped) shaped = shaped_reward3(tb, a, P, POLM, t0_v, qcnt, vc, dm, dj, kc);
            rew = ended_v ? (cap_v ? (a == 0 ? -1.0 : 1.0) : (a == 0 ? 1.0 : 0.0)) : shaped;
            rw_lo = __double2loint(rew);
            rw_hi = __double2hiint(rew);
        }
        // ---- the packed record row of the transition
        if (REC) {
            int* rdst = at_bytes(out.record, off_rec);
            if (store_ok) {
                SY_HOT(h_row);
                if (a < A) {
                    typedef int v2i __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<v2i*>(rdst + 2 * a) = (v2i){rw_lo, rw_hi};      // the float64 reward
                    rdst[2 * A + a] = pos0_v;
                    rdst[3 * A + a] = mon0_v;
                    rdst[4 * A + a] = act0_v;
                }
                if (a < RW - 5 * A)
                    rdst[5 * A + a] = a == 0 ? t0_v : ((fl_v >> mw_shift) & mw_mask);
            }
#ifndef SY_DIAG_NO_ADVANCE
            out.record += (size_t)B * RW;
#endif
            if (POL && out.log_prob) {
                if (store_ok && a < A) out.log_prob[(size_t)eh * A + a] = logp0_v;   // of the action executed in this step
                out.log_prob += (size_t)B * A;
            }
        }
        S3(5)
        if (POL) logp0_v = __int_as_float(w.z);
        pos0_v = pos1_v;
        act0_v = act1_v;
        mon0_v = mon1_v;
        t0_v = meta_v & kMetaTimeMask;
    }
    S3_DUMP("helper [belief store, wait entry, read+unpack, -, -, record store, belief, loophead]", T)
    if (has_belief) bl.finish(kernarg_params()->st.belief, NS, e, live1, !pol_ev);
    if (POL && store_ok) {      // the helper's share of the live state when it evaluates the rewards
        const KernargParams kq = kernarg_params();
        if (a0 < A) kq->st.reward[(size_t)eh * A + a0] = rew;
        uint16_t* vis_out = kq->st.visits;
        for (int i = a0; i < NS; i += 32) vis_out[(size_t)eh * NS + i] = (uint16_t)vis32[i];
    }
}

// Block = wpb episodes (even): wpb / 2 move waves, then wpb / 2 helper waves — one wave per episode, 16 episodes
// per 1024-thread block (one block per CU at B = 4096), 4 waves per SIMD.
template <int NR, bool REC, int PT, bool POL = false, int HS = 0>   // POL: actions from the MAPPO actors (sy_env_set_policy); HS: half-wave scan, columns per lane
__global__ __launch_bounds__(1024, 4) void rollout3_kernel(const EngineParams p, const int T, const sy_rollout_buffers out_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = p.wpb;                          // episodes per block (even)
    const int nmove = wpb >> 1;                     // move waves: two episodes each; as many helper waves
    const bool helper_role = wid >= nmove;
    const int slot = 2 * (helper_role ? wid - nmove : wid);
    const int A = (PT > 0 ? PT : p.P) + 1;
    const int N = p.N, NS = p.NS, B = p.B;
    const int e0 = blockIdx.x * wpb;
    const int e = e0 + slot;                        // first episode of this wave's pair
    const LdsMap L = lds_map(smem, N);
    const EnvLds E = env_lds(L.env_base, slot, p.wave_lds_bytes, A, NS);
    int g = __builtin_amdgcn_readfirstlane(p.env_graph[e0 < B ? e0 : B - 1]);
    g = g < 0 ? 0 : (g >= p.G ? p.G - 1 : g);
    stage_block<true, 3, !POL>(p, L, g, N);
    const EnvLds E1 = env_lds(L.env_base, slot + 1, p.wave_lds_bytes, A, NS);
    if (!helper_role && lane == 0) {
        E.sync[0] = 0; E.sync[1] = 0;
        E1.sync[0] = 0; E1.sync[1] = 0;
    }
    __syncthreads();
    if (e >= B) return;
#if defined(SY_ISA_ROLE) && SY_ISA_ROLE == 1      // register census builds (tools/isa_only.sh -DSY_ISA_ROLE=1|2): one role alone
    move_wave3<NR, REC, PT, POL, HS>(p, L, E, E1, lane, e, g, slot, T, out_arg);
#elif defined(SY_ISA_ROLE) && SY_ISA_ROLE == 2
    helper_wave3<NR, REC, PT, POL>(p, L, E, E1, lane, e, g, T, out_arg);
#else
    if (helper_role) helper_wave3<NR, REC, PT, POL>(p, L, E, E1, lane, e, g, T, out_arg);
    else move_wave3<NR, REC, PT, POL, HS>(p, L, E, E1, lane, e, g, slot, T, out_arg);
#endif
}

}  // namespace sy
