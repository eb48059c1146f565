// sy_device.hpp — device-side helpers shared by every engine kernel (wave primitives, Philox, the neighbour scans,
// move resolution, reward tables, the single-episode belief filter, the LDS carve-up, board staging).
// Included by each kernel family's translation unit; see sy_kernels.h for the file map.
//
// Execution model.  Agent state lives in lanes: lane a holds agent a's node / budget / action (lane 0 =
// MrX, lane k+1 = Police k).  A launch block holds `wpb` episodes that share one board: the board's ELL
// adjacency (16 packed entries per node) is staged once per block in LDS together with the reward
// lookup tables; every episode owns a private LDS slice (mask rows, visit counters, belief scratch,
// ring).  Kernels:
//   step_kernel          one transition with caller actions, one wavefront per episode
//   rollout3_kernel      the fused rollout (default): a two-stage pipeline per pair of episodes — a "move"
//                        wavefront carries TWO episodes (lanes 0-31 / 32-63, per-half predicates as scalar
//                        lane masks, DPP pair checks and reductions; up to 5 agents: the neighbour scan gives each
//                        episode its own half wave) and runs the state feedback loop plus
//                        everything that reads the board; a "helper" wavefront takes what only leaves the
//                        chip (record rows, the belief filter and its rows) from an LDS ring: 4 waves per
//                        SIMD at 16 episodes per CU.  Actions come from the uniform-random policy or (POL)
//                        from the MAPPO actors evaluated in the move wave (rewards then move to the helper).
//   rollout2_kernel      round 1's two-role kernel (move wave does everything but the belief): boards of more
//                        than 256 nodes / more than two scan passes, A/B baseline
//   rollout_kernel       the same with one episode per move wave (odd block sizes)
//   returns_kernel       returns / advantages / GAE of a whole [T][B][A] record in one launch
//   reset / belief_update / action_mask_dense / apsp / sample_boards        reset-side and standalone ops
//   masked_sample / mappo_policy      policy side: masked categorical sampling; the MAPPO networks with
//                        the second layers on the matrix cores (f32 MFMA) — the only GEMM-shaped work here.
// Membership tests ("is this node a neighbour", "is the target occupied", "is MrX caught") are wave
// ballots; the sequential move order of the reference (yard.py:161-243) is kept exactly whenever two
// officers could interact.
//
// Semantics follow the reference file:line cited at each phase (paths under
// /root/reference/src/environment/).  Compile with -ffp-contract=off: the float64 reward
// arithmetic keeps the reference's Python operation order (no fused multiply-add).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sy_kernels.h"

namespace sy {

#ifndef SY_ROLLOUT_MIN_WAVES
#define SY_ROLLOUT_MIN_WAVES 6   // waves per SIMD the rollout kernel is register-budgeted for (2 blocks of 12 waves per CU)
#endif

static constexpr int kWave = 64;
static constexpr int kD = SY_ELL_WIDTH;  // 16 ELL entries per node
static constexpr uint32_t kPurposeAct = 1u, kPurposeReset = 2u;
static constexpr int kPhiloxRounds = 7;
static constexpr int kLdsTab = SY_LDS_TABLE;   // entries of the exp / coverage / reciprocal tables in LDS
static constexpr int kAvgTab = SY_LDS_AVGTAB;  // entries of the -1/(sum/P+1) table in LDS
static constexpr int kRing = SY_RING;          // move wave -> belief wave ring depth (steps)
#ifndef SY_SPIN_MAX
#define SY_SPIN_MAX (1 << 20)
#endif
static constexpr int kSpinMax = SY_SPIN_MAX;   // every spin is bounded: a lost partner cannot hang the GPU

// The trajectory record is written once and read by nobody inside the launch: streaming (non-temporal) stores.
#ifdef SY_NO_STREAM_STORES
#define SY_STREAM_STORE(ptr, val) (*(ptr) = (val))
#else
#define SY_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif

#ifdef SY_ISA_ONLY   // way-points for tools/isa_hot.py (comments in the assembly listing of the ISA-only build)
#define SY_HOT(tag) asm volatile("; SYHOT " #tag)
#else
#define SY_HOT(tag)
#endif

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in order; this only stops the compiler from reordering
    // the cross-lane LDS hand-offs inside a wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int bperm(int byte_addr, int v) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }

// Explicit LDS addressing: byte offsets in registers, typed address-space-3 accesses (always ds_* instructions).
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T* lds_at(uint32_t off) {   // LDS byte offset -> typed LDS pointer
    return (__attribute__((address_space(3))) T*)(uintptr_t)off;
}
template <typename T>
__device__ __forceinline__ T* lds_at_generic(uint32_t off) { return (T*)lds_at<T>(off); }   // ... and back to a generic pointer
__device__ __forceinline__ uint32_t lds_off(const void* q) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)q; }

// Mask algebra for paired waves: predicates that are uniform per half are kept as 64-bit lane masks in
// SGPRs and combined with scalar instructions; only the primitive compares are vector work.
__device__ __forceinline__ uint64_t bal(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lanes(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
static constexpr uint64_t kLowHalf = 0x00000000ffffffffull, kHighHalf = 0xffffffff00000000ull;
__device__ __forceinline__ uint64_t half_any(uint64_t m) {      // each half all-ones iff any of its bits is set
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    const uint32_t mlo = lo ? ~0u : 0u, mhi = hi ? ~0u : 0u;
    return ((uint64_t)mhi << 32) | mlo;
}
// The same for masks of agent lanes (bits 0..7 of a half; any value below 2^31 works) in plain scalar arithmetic: the
// compare-and-select form above makes the compiler route the booleans through the vector unit.
__device__ __forceinline__ uint32_t nz31(uint32_t m) { return (0u - m) >> 31; }   // 1 iff m != 0   (m < 2^31)
__device__ __forceinline__ uint32_t z31(uint32_t m) { return (m - 1u) >> 31; }    // 1 iff m == 0   (m < 2^31)
__device__ __forceinline__ uint64_t half_any8(uint64_t m) {
    const uint32_t mlo = 0u - nz31((uint32_t)m), mhi = 0u - nz31((uint32_t)(m >> 32));
    return ((uint64_t)mhi << 32) | mlo;
}
__device__ __forceinline__ uint64_t half_pick(uint64_t lo_src, uint64_t hi_src) {
    return (lo_src & kLowHalf) | (hi_src & kHighHalf);
}
// lane i <- lane i - D inside its row of 16 (agents of a half sit on lanes 0..7 of a row); zero fill
template <int D>
__device__ __forceinline__ int dpp_shr(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xf, 0xf, true);
}
// Police pairs (k - D, k): does either one target the other's node or the same node?  Evaluated on lane k.
template <int D>
__device__ __forceinline__ uint64_t pair_conflicts(int tgt_v, int pos_v) {
    const int st = dpp_shr<D>(tgt_v), sp = dpp_shr<D>(pos_v);
    return bal(tgt_v == st) | bal(tgt_v == sp) | bal(pos_v == st);
}


template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Wave-wide float sum without LDS traffic: DPP butterflies inside each 16-lane row
// (quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8), then the four row totals via v_readlane.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x128>(v);
    const int iv = __float_as_int(v);
    return ((__int_as_float(rdlane(iv, 0)) + __int_as_float(rdlane(iv, 16))) + __int_as_float(rdlane(iv, 32))) +
           __int_as_float(rdlane(iv, 48));
}

// Philox4x32-7 (Salmon et al. 2011; 7 rounds is the paper's Crush-resistant minimum).
__device__ __forceinline__ void philox4(uint64_t gid, uint32_t ctr, uint32_t purpose, uint32_t idx, uint32_t k0,
                                        uint32_t k1, uint32_t (&o)[4]) {
    uint32_t c0 = (uint32_t)gid, c1 = (uint32_t)(gid >> 32), c2 = ctr, c3 = (purpose << 8) | idx;
#pragma unroll
    for (int r = 0; r < kPhiloxRounds; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// lane i <- lane i - D inside its row of 16 (agents of an episode sit on lanes 0..7 of a row); zero fill
template <int D>
__device__ __forceinline__ int dpp_row_shr(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xf, 0xf, true);
}
// lanes whose value equals the value of an earlier agent lane of the same row (lanes a with d <= a < A, any d)
__device__ __forceinline__ uint64_t earlier_duplicates(int r, int A) {
    uint64_t dup = 0;
    const uint64_t rows = 0x0001000100010001ull;          // lane 0 of every row of 16
    const uint64_t agents = ((1ull << A) - 1ull) * rows;
    dup |= __builtin_amdgcn_ballot_w64(r == dpp_row_shr<1>(r)) & agents & ~(((1ull << 1) - 1ull) * rows);
    if (A > 2) dup |= __builtin_amdgcn_ballot_w64(r == dpp_row_shr<2>(r)) & agents & ~(((1ull << 2) - 1ull) * rows);
    if (A > 3) dup |= __builtin_amdgcn_ballot_w64(r == dpp_row_shr<3>(r)) & agents & ~(((1ull << 3) - 1ull) * rows);
    if (A > 4) dup |= __builtin_amdgcn_ballot_w64(r == dpp_row_shr<4>(r)) & agents & ~(((1ull << 4) - 1ull) * rows);
    if (A > 5) dup |= __builtin_amdgcn_ballot_w64(r == dpp_row_shr<5>(r)) & agents & ~(((1ull << 5) - 1ull) * rows);
    if (A > 6) dup |= __builtin_amdgcn_ballot_w64(r == dpp_row_shr<6>(r)) & agents & ~(((1ull << 6) - 1ull) * rows);
    if (A > 7) dup |= __builtin_amdgcn_ballot_w64(r == dpp_row_shr<7>(r)) & agents & ~(((1ull << 7) - 1ull) * rows);
    return dup;
}

// Distinct start nodes, uniform over ordered tuples of distinct nodes (replaces np.random.choice(N, A,
// replace=False), yard.py:112-116; own RNG stream, engine-defined).  Boards with N >= 2 A^2 (collisions are rare):
// REJECTION of whole tuples — attempt j = 0, 1, ... takes word (j & 3) of Philox block (env, ctr, RESET << 8 |
// (j >> 2) << 3 | agent), node = mulhi(word, N); the first attempt whose nodes are pairwise distinct wins (the same
// distribution as drawing without replacement, ~A^2 / 2N retries).  Smaller boards, or 128 failed attempts: sequential
// draws without replacement from word 0 of block (env, ctr, RESET << 8 | agent).  Lane a returns agent a's start.
__device__ __noinline__ int sample_starts(int lane, int A, int N, uint64_t gid, uint32_t ctr, uint32_t k0, uint32_t k1) {
    uint32_t o[4];
    if (N >= 2 * A * A) {
        for (uint32_t j = 0; j < 128u; ++j) {
            if ((j & 3u) == 0u) philox4(gid, ctr, kPurposeReset, ((j >> 2) << 3) | ((uint32_t)lane & 7u), k0, k1, o);
            const uint32_t m = j & 3u;
            const uint32_t x = m == 0 ? o[0] : (m == 1 ? o[1] : (m == 2 ? o[2] : o[3]));
            const int r = (int)__umulhi(x, (uint32_t)N);
            if ((earlier_duplicates(r, A) & 0xffull) == 0ull) return r;
        }
    }
    philox4(gid, ctr, kPurposeReset, (uint32_t)lane, k0, k1, o);
    const uint32_t xv = o[0];
    int sorted[SY_MAX_AGENTS];
#pragma unroll
    for (int j = 0; j < SY_MAX_AGENTS; ++j) sorted[j] = 0x7fffffff;
    int mine = 0;
#pragma unroll
    for (int i = 0; i < SY_MAX_AGENTS; ++i) {
        if (i < A) {
            const uint32_t x = (uint32_t)rdlane((int)xv, i);
            int r = (int)__umulhi(x, (uint32_t)(N - i));
#pragma unroll
            for (int j = 0; j < SY_MAX_AGENTS; ++j)
                if (j < i) r += (r >= sorted[j]) ? 1 : 0;
#pragma unroll
            for (int j = SY_MAX_AGENTS - 1; j >= 0; --j) {
                const int prev = j == 0 ? -1 : sorted[j - 1];
                sorted[j] = sorted[j] < r ? sorted[j] : (prev < r ? r : prev);
            }
            if (lane == i) mine = r;
        }
    }
    return mine;
}

// Lane -> (agent slot, ELL column) mapping of the neighbour scans.  `gw` ELL columns per agent
// (8..16: the pool's widest row, at least 8), so 64 / gw agents are scanned per pass: at P = 4 and
// rows of at most 12 neighbours one pass covers all 5 agents, at P = 6 rows of at most 9 do.
struct ScanMap {
    int grp, col, sh, per_pass;
    int gsh;            // bit offset of this lane's group inside a pass ballot
    int ash;            // agent role: bit offset of agent (lane & 31)'s field inside a pass ballot
    uint32_t lowmask;   // bits of the group's field below this lane's column
    bool live;
};
template <bool ANY_WIDTH = true>
__device__ __forceinline__ ScanMap make_scan_map(int lane, int gw) {
    ScanMap m;
    if (ANY_WIDTH) {     // any width in 8..16 (the host picks 9 or 10 when that saves a scan pass for 6 or 7 agents)
        m.per_pass = 64 / gw;
        m.grp = lane / gw;
    } else {             // 8, 12 or 16: no division (instances for at most 5 agents never see another width)
        m.per_pass = gw == 8 ? 8 : (gw == 12 ? 5 : 4);
        m.grp = gw == 8 ? (lane >> 3) : (gw == 12 ? (lane * 43) >> 9 : (lane >> 4));
    }
    m.col = lane - m.grp * gw;
    m.live = m.grp < m.per_pass;
    m.sh = (lane % m.per_pass) * gw;   // bit offset of agent (lane)'s field inside a pass ballot
    m.gsh = m.live ? m.grp * gw : 0;
    m.ash = ((lane & 31) % m.per_pass) * gw;
    m.lowmask = (1u << m.col) - 1u;
    return m;
}

// Post-move scan (yard.py:297-317 masks == yard.py:420-472 node sets).  Rebuilds the wave's mask
// rows in LDS and returns, on lane a, agent a's "affordable entry" bit field and the
// |possible_moves| count the police position reward uses — which the reference evaluates with agent
// index i instead of i+1, i.e. the budget of the PREVIOUS agent (reward_calculator.py:190; kept for
// parity).  Padding entries carry weight 0xFFFF, above any budget the ABI admits, so "affordable"
// alone identifies real neighbours.
__device__ __forceinline__ void scan_masks(const uint32_t* ell_s, uint8_t* mrow, int lane, int A, int NS, int n16,
                                           int gw, const ScanMap& sm, int pos_v, int mon_v, uint32_t& aff_field,
                                           int& quirk_cnt) {
    for (int base16 = 0; base16 < n16; base16 += kWave)   // wave-uniform trip count
        if (base16 + lane < n16) reinterpret_cast<uint4*>(mrow)[base16 + lane] = make_uint4(0, 0, 0, 0);
    wave_lds_fence();
    aff_field = 0;
    quirk_cnt = 0;
    const uint32_t fmask = (1u << gw) - 1u;
    for (int base = 0; base < A; base += sm.per_pass) {
        const int a = base + sm.grp;
        const bool on = sm.live && a < A;
        const int src = on ? a : 0;
        const int pa = bperm(src << 2, pos_v);
        int ma = bperm(src << 2, mon_v);
        const int mq = bperm((src > 0 ? src - 1 : 0) << 2, mon_v);
        ma = on ? ma : -1;
        const uint32_t ent = ell_s[(pa << 4) | sm.col];
        const int w = (int)(ent >> 16);
        const bool own = w <= ma;
        const uint64_t bo = __ballot(own), bq = __ballot(on && w <= mq);
        if (own) mrow[a * NS + (int)(ent & 0xffffu)] = 1;
        if (lane >= base && lane < base + sm.per_pass) {
            aff_field = (uint32_t)(bo >> sm.sh) & fmask;
            quirk_cnt = __popc((uint32_t)(bq >> sm.sh) & fmask);
        }
    }
    wave_lds_fence();
}

// scan_masks plus the uniform-random policy (random_agent.py) for the NEXT step, decided inside the
// scan: a scan lane is chosen when it is affordable and its rank among its agent's affordable
// entries equals r = mulhi(draw, count) — i.e. the r-th legal neighbour in ascending node order.
// Lane a returns the sampled action (-1 if the mask is empty) and its edge cost.
__device__ __forceinline__ void scan_sample(const uint32_t* ell_s, uint8_t* mrow, int lane, int A, int NS, int n16,
                                            int gw, const ScanMap& sm, int pos_v, int mon_v, uint32_t x_v,
                                            int& act_v, int& cost_v, int& quirk_cnt, int lane_off = 0) {
    // lane_off: first agent lane of the scanned episode (0; 32 for the second episode of a paired wave).
    // All 64 lanes scan; only that episode's agent lanes receive results (the others keep theirs).
    for (int base16 = 0; base16 < n16; base16 += kWave)   // wave-uniform trip count
        if (base16 + lane < n16) reinterpret_cast<uint4*>(mrow)[base16 + lane] = make_uint4(0, 0, 0, 0);
    wave_lds_fence();
    const int al = lane - lane_off;
    if (al >= 0 && al < 32) {
        act_v = -1;
        cost_v = 0;
        quirk_cnt = 0;
    }
    const uint32_t fmask = (1u << gw) - 1u;
    for (int base = 0; base < A; base += sm.per_pass) {
        const int a = base + sm.grp;
        const bool on = sm.live && a < A;
        const int src = lane_off + (on ? a : 0);
        const int pa = bperm(src << 2, pos_v);
        int ma = bperm(src << 2, mon_v);
        const int mq = bperm((on && a > 0 ? src - 1 : src) << 2, mon_v);
        const uint32_t xa = (uint32_t)bperm(src << 2, (int)x_v);
        ma = on ? ma : -1;
        const uint32_t ent = ell_s[(pa << 4) | sm.col];
        const int w = (int)(ent >> 16);
        const bool own = w <= ma;
        const uint64_t bo = __ballot(own), bq = __ballot(on && w <= mq);
        if (own) mrow[a * NS + (int)(ent & 0xffffu)] = 1;
        const uint32_t gfield = (uint32_t)(bo >> sm.gsh) & fmask;           // this lane's agent's affordable entries
        const int rr = (int)__umulhi(xa, (uint32_t)__popc(gfield));
        const bool chosen = own && (int)__popc(gfield & sm.lowmask) == rr;
        const uint64_t bc = __ballot(chosen);
        const uint32_t cf = (uint32_t)(bc >> sm.ash) & fmask;               // agent lane's chosen column, one-hot
        const int from = sm.ash + (cf ? __ffs((int)cf) - 1 : 0);
        const uint32_t esel = (uint32_t)bperm(from << 2, (int)ent);
        if (al >= base && al < base + sm.per_pass) {
            act_v = cf ? (int)(esel & 0xffffu) : -1;
            cost_v = cf ? (int)(esel >> 16) : 0;
            quirk_cnt = __popc((uint32_t)(bq >> sm.ash) & fmask);
        }
    }
    wave_lds_fence();
}

// Both episodes of a paired wave scanned in lockstep (same work as two scan_sample calls, but the two
// independent dependency chains — bpermute -> ELL read -> ballots -> bpermute — overlap).  Split in
// two so the gather half (two dependent LDS round trips) can be issued right after the moves and
// overlap the visit / shortest-path / mask-copy phases; the evaluate half runs where the scan was.
struct ScanPairIn {   // first-pass operands of both episodes, one set per scan lane
    uint32_t ent0, ent1, xa0, xa1;
    int ma0, ma1, mq0, mq1;
};
__device__ __forceinline__ ScanPairIn scan_gather_pair(const uint32_t* ell_s, int A, const ScanMap& sm, int base, int pos_v,
                                                       int mon_v, uint32_t x_v) {
    ScanPairIn g;
    const int a = base + sm.grp;
    const bool on = sm.live && a < A;
    const int s0 = on ? a : 0, s1 = 32 + s0;
    const int q0 = (on && a > 0) ? s0 - 1 : s0, q1 = 32 + q0;
    const int pa0 = bperm(s0 << 2, pos_v), pa1 = bperm(s1 << 2, pos_v);
    g.ma0 = bperm(s0 << 2, mon_v);
    g.ma1 = bperm(s1 << 2, mon_v);
    g.mq0 = bperm(q0 << 2, mon_v);
    g.mq1 = bperm(q1 << 2, mon_v);
    g.xa0 = (uint32_t)bperm(s0 << 2, (int)x_v);
    g.xa1 = (uint32_t)bperm(s1 << 2, (int)x_v);
    g.ent0 = ell_s[(pa0 << 4) | sm.col];
    g.ent1 = ell_s[(pa1 << 4) | sm.col];
    return g;
}

__device__ __forceinline__ void scan_eval_pair(const uint32_t* ell_s, uint8_t* mrow0, uint8_t* mrow1, int lane, int A, int NS,
                                               int n16, int gw, const ScanMap& sm, ScanPairIn g, int pos_v, int mon_v,
                                               uint32_t x_v, int& act_v, int& cost_v, int& quirk_cnt) {
    for (int base16 = 0; base16 < n16; base16 += kWave)   // wave-uniform trip count
        if (base16 + lane < n16) {
            reinterpret_cast<uint4*>(mrow0)[base16 + lane] = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(mrow1)[base16 + lane] = make_uint4(0, 0, 0, 0);
        }
    wave_lds_fence();
    act_v = -1;
    cost_v = 0;
    quirk_cnt = 0;
    const uint32_t fmask = (1u << gw) - 1u;
    const bool upper = lane >= 32;
    const int al = lane & 31;
    for (int base = 0; base < A; base += sm.per_pass) {
        if (base > 0) g = scan_gather_pair(ell_s, A, sm, base, pos_v, mon_v, x_v);   // further passes: gather inline
        const int a = base + sm.grp;
        const bool on = sm.live && a < A;
        const int w0 = (int)(g.ent0 >> 16), w1 = (int)(g.ent1 >> 16);
        const bool own0 = on && w0 <= g.ma0, own1 = on && w1 <= g.ma1;
        const uint64_t bo0 = __ballot(own0), bo1 = __ballot(own1);
        const uint64_t bq0 = __ballot(on && w0 <= g.mq0), bq1 = __ballot(on && w1 <= g.mq1);
        if (own0) mrow0[a * NS + (int)(g.ent0 & 0xffffu)] = 1;
        if (own1) mrow1[a * NS + (int)(g.ent1 & 0xffffu)] = 1;
        const uint32_t gf0 = (uint32_t)(bo0 >> sm.gsh) & fmask, gf1 = (uint32_t)(bo1 >> sm.gsh) & fmask;
        const int rr0 = (int)__umulhi(g.xa0, (uint32_t)__popc(gf0)), rr1 = (int)__umulhi(g.xa1, (uint32_t)__popc(gf1));
        const bool ch0 = own0 && (int)__popc(gf0 & sm.lowmask) == rr0, ch1 = own1 && (int)__popc(gf1 & sm.lowmask) == rr1;
        const uint64_t bc0 = __ballot(ch0), bc1 = __ballot(ch1);
        // agent lanes: lower half takes episode 0's ballots, upper half episode 1's
        const uint64_t bc = upper ? bc1 : bc0, bq = upper ? bq1 : bq0;
        const uint32_t cf = (uint32_t)(bc >> sm.ash) & fmask;
        const int from = sm.ash + (cf ? __ffs((int)cf) - 1 : 0);
        const uint32_t e0s = (uint32_t)bperm(from << 2, (int)g.ent0), e1s = (uint32_t)bperm(from << 2, (int)g.ent1);
        const uint32_t esel = upper ? e1s : e0s;
        if (al >= base && al < base + sm.per_pass) {
            act_v = cf ? (int)(esel & 0xffffu) : -1;
            cost_v = cf ? (int)(esel >> 16) : 0;
            quirk_cnt = __popc((uint32_t)(bq >> sm.ash) & fmask);
        }
    }
    wave_lds_fence();
}

__device__ __forceinline__ void scan_sample_pair(const uint32_t* ell_s, uint8_t* mrow0, uint8_t* mrow1, int lane, int A,
                                                 int NS, int n16, int gw, const ScanMap& sm, int pos_v, int mon_v,
                                                 uint32_t x_v, int& act_v, int& cost_v, int& quirk_cnt) {
    const ScanPairIn g = scan_gather_pair(ell_s, A, sm, 0, pos_v, mon_v, x_v);
    scan_eval_pair(ell_s, mrow0, mrow1, lane, A, NS, n16, gw, sm, g, pos_v, mon_v, x_v, act_v, cost_v, quirk_cnt);
}

// Membership test `action in possible_positions` (yard.py:168,218) for caller-given actions:
// lane a gets ok (affordable neighbour) and the edge cost (yard.py:234-236).
__device__ __forceinline__ void scan_hits(const uint32_t* ell_s, int lane, int A, int gw, const ScanMap& sm, int pos_v,
                                          int mon_v, int act_v, bool& ok, int& cost) {
    ok = false;
    cost = 0;
    const uint32_t fmask = (1u << gw) - 1u;
    for (int base = 0; base < A; base += sm.per_pass) {
        const int a = base + sm.grp;
        const bool on = sm.live && a < A;
        const int src = on ? a : 0;
        const int pa = bperm(src << 2, pos_v);
        const int ma = bperm(src << 2, mon_v);
        const int aa = bperm(src << 2, act_v);
        const uint32_t ent = ell_s[(pa << 4) | sm.col];
        const int nbr = (int)(ent & 0xffffu), w = (int)(ent >> 16);
        const bool hit = on && (w <= ma) && (nbr == aa);
        const uint64_t bh = __ballot(hit);
        const uint32_t field = (uint32_t)(bh >> sm.sh) & fmask;
        const int from = sm.sh + (field ? __ffs((int)field) - 1 : 0);
        const int wsel = bperm(from << 2, w);
        if (lane >= base && lane < base + sm.per_pass) {
            ok = field != 0;
            cost = ok ? wsel : 0;
        }
    }
}

// Moves (yard.py:161-243): MrX first against the PRE-move police, then police strictly in index
// order, each seeing earlier moves.  tgt_v = wanted node (own node when the action is not a legal
// move), skipm = ballot of agents that are skipped (-1 / None / no money).
__device__ __forceinline__ void resolve_moves(int lane, int P, bool is_pol, int tgt_v, uint64_t skipm, int cost_v,
                                              int& pos_v, int& mon_v) {
    {
        const int tgt = rdlane(tgt_v, 0);
        const bool blocked = __ballot(is_pol && pos_v == tgt) != 0ull;      // :180-188
        if (!blocked && lane == 0) pos_v = tgt;
    }
    for (int k = 1; k <= P; ++k) {
        const int tgt = rdlane(tgt_v, k);
        const bool occ = __ballot(is_pol && pos_v == tgt) != 0ull;          // own node included (:231)
        if (!occ && !((skipm >> k) & 1ull) && lane == k) {
            pos_v = tgt;
            mon_v -= cost_v;                                                // :234-236
        }
    }
}

// Same result as resolve_moves, with a parallel fast path: when no police target coincides with
// another police officer's current node or target, the sequential order cannot matter and every
// non-skipped officer whose target differs from its own node simply moves.  P independent ballots
// instead of a chain of P dependent ones; conflicts (rare on a sparse board) take the exact loop.
template <int PT>
__device__ __forceinline__ void resolve_moves_fast(int lane, int P, bool is_pol, int tgt_v, uint64_t skipm, int cost_v,
                                                   int& pos_v, int& mon_v) {
    const int t0 = rdlane(tgt_v, 0);
    const bool blocked = __ballot(is_pol && pos_v == t0) != 0ull;            // MrX vs PRE-move police (:180-188)
    uint64_t conf = 0ull;
#pragma unroll
    for (int k = 1; k < SY_MAX_AGENTS; ++k) {
        if (k <= P) {
            const int tk = rdlane(tgt_v, k);
            conf |= __ballot(is_pol && lane != k && (pos_v == tk || tgt_v == tk));
        }
    }
    if (!blocked && lane == 0) pos_v = t0;
    if (conf == 0ull) {
        if (is_pol && !((skipm >> lane) & 1ull) && tgt_v != pos_v) {
            pos_v = tgt_v;
            mon_v -= cost_v;                                                  // :234-236
        }
    } else {
        for (int k = 1; k <= P; ++k) {
            const int tgt = rdlane(tgt_v, k);
            const bool occ = __ballot(is_pol && pos_v == tgt) != 0ull;        // own node included (:231)
            if (!occ && !((skipm >> k) & 1ull) && lane == k) {
                pos_v = tgt;
                mon_v -= cost_v;
            }
        }
    }
}

// Reward lookup tables: LDS copies for the fused rollout, global tables for the single step.
struct RewardTabs {
    const double* exp_s;   // [kLdsTab + 1] exp(-d); slot kLdsTab holds 0.0
    const double* cov_s;   // [kLdsTab]     exp(-log1p(v))
    const double* nrc_s;   // [kLdsTab]     -1/(d+1)
    const double* nra_s;   // [kAvgTab]     -1/(s/P+1)
    const double* px_s;    // [kLdsTab + 1] exp(-d) for d > 1, else 0.0 (the proximity term's filter folded into the table)
    const double* exp_g;   // global tables (any length)
    const double* cov_g;
    int n_exp, n_cov;
};

__device__ __forceinline__ double exp_neg_slow(const RewardTabs& tb, int d) { return d < tb.n_exp ? tb.exp_g[d] : 0.0; }
// explicit LDS-address-space read: keeps table lookups on ds_read_b64 (never merged into FLAT loads)
__device__ __forceinline__ double lds_f64(const double* p) {
    return *(const __attribute__((address_space(3))) double*)p;
}

// Per-lane reward coefficients: registers for the single step, a 2x8 LDS table (row 0 = MrX's lane,
// row 1 = police lanes) for the fused rollout, where registers are what limits waves per SIMD.
template <bool LDS_TAB>
struct Coefs {
    double r[8];
    const double* s;
    __device__ __forceinline__ double get(int i) const { return LDS_TAB ? lds_f64(s + i) : r[i]; }
};

// Shaped rewards (reward_calculator.py:94-266) in float64, reference operation order.
// Lane 0 = MrX (:126-148), lanes 1..P = police (:182-229).  dm = d(police, MrX), dj[j-1] = d(police, police j).
// kc[] are per-lane coefficients: lane 0 {w_closest, w_average, w_position, 1-w_time, -, -, -, 0.1},
// police {w_distance, w_group, w_position, 1-w_time, w_proximity, w_overlap, w_coverage, 0.05}.
template <bool LDS_TAB>
__device__ __forceinline__ double shaped_reward(const RewardTabs& tb, int lane, int P, bool is_pol, int t, int qcnt,
                                                int vc, int dm, const int (&dj)[SY_MAX_AGENTS - 1],
                                                const Coefs<LDS_TAB>& kc) {
    int mn = 0x7fffffff, sum = 0;
    for (int k = 1; k <= P; ++k) {
        const int dk = rdlane(dm, k);
        mn = dk < mn ? dk : mn;
        sum += dk;
    }
    // MrX terms: -1/(closest+1), -1/(mean+1)                                   (:140-144)
    double xa, xb;
    if (LDS_TAB && mn < kLdsTab && sum < kAvgTab) {
        xa = lds_f64(tb.nrc_s + mn);
        xb = lds_f64(tb.nra_s + sum);
    } else {
        xa = -1.0 / ((double)mn + 1.0);
        xb = -1.0 / ((double)sum / (double)P + 1.0);
    }
    // police terms: sums over the other police in index order                  (:185-202)
    double group = 0.0, prox = 0.0;
    int overlap = 0;
    int dor = dm | (vc < kLdsTab ? 0 : kLdsTab);   // any value >= 256 sets a bit above bit 7 of the OR
#pragma unroll
    for (int j = 1; j < SY_MAX_AGENTS; ++j) dor |= dj[j - 1];
    double e_mrx, cov;
    if (LDS_TAB && __ballot(dor >= kLdsTab) == 0ull) {
        // fast path (wave-uniform): every lookup hits the LDS tables
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) {
            if (j <= P) {
                const int dij = dj[j - 1];
                const bool other = j != lane;
                const double ex = lds_f64(tb.exp_s + (other ? dij : kLdsTab));   // slot kLdsTab = 0.0: x + 0.0 == x
                group += ex;
                prox += dij > 1 ? ex : 0.0;
                overlap += (other && dij <= 1) ? 1 : 0;
            }
        }
        e_mrx = lds_f64(tb.exp_s + dm);
        cov = lds_f64(tb.cov_s + vc);
    } else {
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) {
            if (j <= P) {
                const int dij = dj[j - 1];
                const bool other = j != lane;
                const double ex = other ? exp_neg_slow(tb, dij) : 0.0;
                group += ex;
                prox += dij > 1 ? ex : 0.0;
                overlap += (other && dij <= 1) ? 1 : 0;
            }
        }
        e_mrx = exp_neg_slow(tb, dm);
        cov = tb.cov_g[vc < tb.n_cov ? vc : tb.n_cov - 1];                 // :204-207
    }
    const double ts = (double)t;
    const double x0 = lane == 0 ? xa : e_mrx, x1 = lane == 0 ? xb : group;
    const double base = ((kc.get(0) * x0 + kc.get(1) * x1) + kc.get(2) * (double)qcnt) + kc.get(3) * (kc.get(7) * ts);   // :140-148 / :214-221
    const double pol = ((base + kc.get(4) * prox) - kc.get(5) * (double)overlap) + kc.get(6) * cov;                      // :222-228
    return lane == 0 ? base : pol;
}

// One diffusion + evidence step of the deterministic belief filter (belief_module.py:69-111 in
// expectation): b' = normalize((b.P) * lik), P[i][j] = adj/deg(i) (row e_i if isolated),
// zero mass -> uniform.  Belief lives in registers (NR slabs of 64 nodes); the scaled vector
// c = b/deg goes through the episode's LDS slice for the neighbour gathers.  boff_s holds, per
// node, 16 uint16 byte offsets (neighbour*4, padding -> the zero slot N*4); each slab gathers only as
// many 4-entry chunks as its widest node needs (rows are filled left to right).
template <int NR>
__device__ __forceinline__ void belief_step(float (&b)[NR], const float (&ideg)[NR], const int (&slab_w)[NR],
                                            float* c_s, const uint16_t* boff_s, int lane, int N, bool police_ev,
                                            const int (&pol)[SY_MAX_AGENTS - 1], int P) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        if (j < N) c_s[j] = b[r] * ideg[r];
    }
    if (lane == 0) c_s[N] = 0.0f;  // padding entries point here
    wave_lds_fence();
    const char* cb = reinterpret_cast<const char*>(c_s);
    auto ld = [cb](uint32_t off) { return *reinterpret_cast<const float*>(cb + off); };
    // Software-pipelined over the slabs: all offset rows first, then every slab's first eight
    // gathers (padding entries read the zero slot), then the sums — three LDS round trips per step
    // instead of three per slab.  Rows wider than 8 neighbours take the second pass below.
    constexpr int GR = NR < 4 ? NR : 4;     // slabs pipelined together (register budget: 8 gathers each)
    float tot = 0.0f;
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += GR) {
        uint4 o[GR];
        int jr[GR];
#pragma unroll
        for (int q = 0; q < GR; ++q) {
            const int j = lane + 64 * (r0 + q);
            jr[q] = j < N ? j : N - 1;          // tail lanes read a valid row; their result is discarded
            o[q] = *reinterpret_cast<const uint4*>(boff_s + (jr[q] << 4));
        }
        float g[GR][8];
#pragma unroll
        for (int q = 0; q < GR; ++q) {
            g[q][0] = ld(o[q].x & 0xffffu); g[q][1] = ld(o[q].x >> 16);
            g[q][2] = ld(o[q].y & 0xffffu); g[q][3] = ld(o[q].y >> 16);
            g[q][4] = ld(o[q].z & 0xffffu); g[q][5] = ld(o[q].z >> 16);
            g[q][6] = ld(o[q].w & 0xffffu); g[q][7] = ld(o[q].w >> 16);
        }
#pragma unroll
        for (int q = 0; q < GR; ++q) {
            const int r = r0 + q;
            const int j = lane + 64 * r;
            float acc = ideg[r] == 0.0f ? b[r] : 0.0f;
            acc += ((g[q][0] + g[q][1]) + (g[q][2] + g[q][3])) + ((g[q][4] + g[q][5]) + (g[q][6] + g[q][7]));
            if (slab_w[r] > 2) {            // wave-uniform: some row of this slab has more than 8 neighbours
                const uint4 o2 = *reinterpret_cast<const uint4*>(boff_s + (jr[q] << 4) + 8);
                const float h0 = ld(o2.x & 0xffffu), h1 = ld(o2.x >> 16), h2 = ld(o2.y & 0xffffu), h3 = ld(o2.y >> 16);
                const float h4 = ld(o2.z & 0xffffu), h5 = ld(o2.z >> 16), h6 = ld(o2.w & 0xffffu), h7 = ld(o2.w >> 16);
                acc += ((h0 + h1) + (h2 + h3)) + ((h4 + h5) + (h6 + h7));
            }
            if (police_ev) {
#pragma unroll
                for (int k = 0; k < SY_MAX_AGENTS - 1; ++k)
                    if (k < P && j == pol[k]) acc = 0.0f;
            }
            acc = j < N ? acc : 0.0f;
            b[r] = acc;
            tot += acc;
        }
    }
    tot = wave_sum(tot);
    const float uni = 1.0f / (float)N;
    const float inv = tot == 0.0f ? 0.0f : __builtin_amdgcn_rcpf(tot);   // 1 ulp; the filter's tolerance is 1e-5
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        b[r] = j < N ? (tot == 0.0f ? uni : b[r] * inv) : 0.0f;
    }
    wave_lds_fence();
}

template <int NR>
__device__ __forceinline__ void belief_prior(float (&b)[NR], int lane, int N, bool onehot, int m0) {
    const float uni = 1.0f / (float)N;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        b[r] = j < N ? (onehot ? (j == m0 ? 1.0f : 0.0f) : uni) : 0.0f;
    }
}

template <int NR>
__device__ __forceinline__ void belief_load(float (&b)[NR], float (&ideg)[NR], int (&slab_w)[NR], const float* bel_row,
                                            const float* ideg_row, int lane, int N) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        b[r] = j < N ? bel_row[j] : 0.0f;
        ideg[r] = j < N ? ideg_row[j] : 0.0f;
        // widest row of the slab, in 4-entry chunks (1/deg -> deg is exact for deg <= 16)
        const int deg = ideg[r] > 0.0f ? (int)(1.0f / ideg[r] + 0.5f) : 0;
        int need = (deg + 3) >> 2;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int other = __shfl_xor(need, o, kWave);
            need = other > need ? other : need;
        }
        slab_w[r] = rdlane(need, 0);
    }
}

__device__ __forceinline__ int lds_peek(const int* p) {   // every lane reads the same word: result is wave-uniform
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_poke(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The engine parameters are the first kernel argument.  Reading rarely used fields (the state
// pointers of the epilogue, the slow-path tables) through this laundered kernarg pointer keeps them
// out of SGPRs during the step loop: the loads stay where they are written.
typedef const __attribute__((address_space(4))) EngineParams* KernargParams;
__device__ __forceinline__ KernargParams kernarg_params() {
    KernargParams q = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

// A bounded spin that ran out is reported, not swallowed: the bit lands in the engine's status word (device memory
// bound with sy_env_bind_status, read by sy_env_status), the launch still drains.  The pointer is re-read from the
// kernel arguments on this cold path only.
__device__ __forceinline__ void report_status(uint32_t bit) {
    uint32_t* w = kernarg_params()->status;
    if (w != nullptr && (threadIdx.x & 63) == 0) atomicOr(w, bit);
}

template <typename T>
__device__ __forceinline__ T* at_bytes(T* base, uint32_t byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ const T* at_bytes(const T* base, uint32_t byte_off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// LDS carve-up shared by all engine kernels (host mirror: sy_capi.hip::lds_bytes_for).
struct LdsMap {
    uint32_t* ell_s;
    uint16_t* boff_s;
    double *exp_s, *cov_s, *nrc_s, *nra_s, *px_s, *kc_s;
    unsigned char* env_base;   // first per-episode slice
};
__device__ __forceinline__ LdsMap lds_map(unsigned char* smem, int N) {
    // fixed-size tables first: their LDS addresses are compile-time immediates (no SGPRs spent on them)
    LdsMap m;
    m.exp_s = reinterpret_cast<double*>(smem);
    m.cov_s = m.exp_s + (kLdsTab + 2);
    m.nrc_s = m.cov_s + kLdsTab;
    m.nra_s = m.nrc_s + kLdsTab;
    m.px_s = m.nra_s + kAvgTab;
    m.kc_s = m.px_s + (kLdsTab + 2);
    m.ell_s = reinterpret_cast<uint32_t*>(m.kc_s + 16);
    m.boff_s = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(m.ell_s) + (size_t)N * kD * 4);
    m.env_base = reinterpret_cast<unsigned char*>(m.ell_s) + (size_t)N * kD * 6;
    return m;
}

// Per-episode LDS slice: [sync 16 B][ring kRing x 64 B][record 256 B][belief scratch (NS+16)*8][visit counters NS*4][mask rows A*NS]
// — the fixed-size parts first, so they sit at immediate offsets from the slice base.
struct EnvLds {
    uint8_t* mrow;
    uint16_t* vis_s;
    float* c_s;
    int* ring;          // kRing entries of 8 dwords
    int* sync;          // [0] produced, [1] consumed
    int* rec_s;         // 64 dwords: one packed trajectory record being assembled
};
__device__ __forceinline__ EnvLds env_lds(unsigned char* base, int slot, int slice_bytes, int A, int NS) {
    EnvLds e;
    unsigned char* w = base + (size_t)slot * slice_bytes;
    e.sync = reinterpret_cast<int*>(w);
    e.ring = e.sync + 4;
    e.rec_s = e.ring + kRing * (SY_RING_ENTRY_BYTES / 4);
    e.c_s = reinterpret_cast<float*>(e.rec_s + 64);
    e.vis_s = reinterpret_cast<uint16_t*>(e.c_s + 2 * (NS + 16));   // 8 B per node: the paired kernel interleaves two episodes
    e.mrow = reinterpret_cast<uint8_t*>(e.vis_s + 2 * NS);   // NS*4 bytes: the paired kernel keeps 32-bit counters
    return e;
}

__device__ __forceinline__ void load_coeffs(const EngineParams& p, int lane, double (&kc)[8]) {
    // reward_calculator.py:140-148 for MrX on lane 0, :214-229 for police (weights order reward_net.py:5-17)
    kc[0] = lane == 0 ? p.w[4] : p.w[0];
    kc[1] = lane == 0 ? p.w[5] : p.w[1];
    kc[2] = lane == 0 ? p.w[6] : p.w[2];
    kc[3] = 1.0 - (lane == 0 ? p.w[7] : p.w[3]);
    kc[4] = p.w[9];
    kc[5] = p.w[10];
    kc[6] = p.w[8];
    kc[7] = lane == 0 ? 0.1 : 0.05;
}

// Stage the block's board: ELL rows (coalesced 16-byte loads), the belief gather offsets derived
// from them, and (TABLES) the reward lookup tables.
// LAYOUT: the kernel's belief filter honours the host's bank-aware scratch layout (sy_env_set_belief_layout): the gather
// offsets then come from the pool's table instead of the ELL order.
template <bool TABLES, int CSHIFT = 2, bool LAYOUT = false>
__device__ __forceinline__ void stage_block(const EngineParams& p, const LdsMap& L, int g, int N) {
    const uint4* src = reinterpret_cast<const uint4*>(p.ell + (size_t)g * N * kD);
    uint4* dst = reinterpret_cast<uint4*>(L.ell_s);
    const uint2* lay = (LAYOUT && p.bel_gather) ? reinterpret_cast<const uint2*>(p.bel_gather + (size_t)g * N * kD) : nullptr;
    for (int i = threadIdx.x; i < N * 4; i += blockDim.x) {
        const uint4 v = src[i];
        dst[i] = v;
        uint2 o;   // byte offsets of the neighbours' belief-scratch entries (4 B each, 8 B in the paired kernel)
        o.x = ((v.x & 0xffffu) << CSHIFT) | ((v.y & 0xffffu) << (16 + CSHIFT));
        o.y = ((v.z & 0xffffu) << CSHIFT) | ((v.w & 0xffffu) << (16 + CSHIFT));
        if (LAYOUT && lay) o = lay[i];
        reinterpret_cast<uint2*>(L.boff_s)[i] = o;
    }
    if (TABLES) {
        for (int i = threadIdx.x; i < kLdsTab; i += blockDim.x) {
            L.exp_s[i] = i < p.n_exp ? p.exp_tab[i] : 0.0;
            L.cov_s[i] = p.cov_tab[i < p.n_cov ? i : p.n_cov - 1];
            L.nrc_s[i] = -1.0 / ((double)i + 1.0);
            L.px_s[i] = (i > 1 && i < p.n_exp) ? p.exp_tab[i] : 0.0;
        }
        if (threadIdx.x == 0) { L.exp_s[kLdsTab] = 0.0; L.px_s[kLdsTab] = 0.0; }
        for (int i = threadIdx.x; i < kAvgTab; i += blockDim.x) L.nra_s[i] = -1.0 / ((double)i / (double)p.P + 1.0);
        if (threadIdx.x < 2) {
            double kc[8];
            load_coeffs(p, (int)threadIdx.x, kc);
#pragma unroll
            for (int i = 0; i < 8; ++i) L.kc_s[threadIdx.x * 8 + i] = kc[i];
        }
    }
}

}  // namespace sy
