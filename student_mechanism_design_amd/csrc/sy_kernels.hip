// sy_kernels.hip — hand-written CDNA4 (gfx950) kernels of the batched Scotland-Yard engine.
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
template <bool TABLES, int CSHIFT = 2>
__device__ __forceinline__ void stage_block(const EngineParams& p, const LdsMap& L, int g, int N) {
    const uint4* src = reinterpret_cast<const uint4*>(p.ell + (size_t)g * N * kD);
    uint4* dst = reinterpret_cast<uint4*>(L.ell_s);
    for (int i = threadIdx.x; i < N * 4; i += blockDim.x) {
        const uint4 v = src[i];
        dst[i] = v;
        uint2 o;   // byte offsets of the neighbours' belief-scratch entries (4 B each, 8 B in the paired kernel)
        o.x = ((v.x & 0xffffu) << CSHIFT) | ((v.y & 0xffffu) << (16 + CSHIFT));
        o.y = ((v.z & 0xffffu) << CSHIFT) | ((v.w & 0xffffu) << (16 + CSHIFT));
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

// ---------------------------------------------------------------------------------------------
// step_kernel: one env transition with caller-given actions (sy_env_step), one wave per episode.
// ---------------------------------------------------------------------------------------------
// `rec` (sy_env_step_record): the row of a rollout record this transition fills — the observation before
// the step (masks, belief), the packed {reward, pos, budget, action, t, flags} row — so a policy-driven
// collector needs no copy kernels.  All three pointers may be null.
template <int NR>
__global__ __launch_bounds__(1024) void step_kernel(const EngineParams p, const int32_t* __restrict__ actions,
                                                    const sy_rollout_buffers rec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    const int N = p.N, NS = p.NS, A = p.A, P = p.P, B = p.B;
    const int e0 = blockIdx.x * wpb;
    const int e = e0 + wid;
    const LdsMap L = lds_map(smem, N);
    const EnvLds E = env_lds(L.env_base, wid, p.wave_lds_bytes, A, NS);
    int g = __builtin_amdgcn_readfirstlane(p.env_graph[e0 < B ? e0 : B - 1]);
    g = g < 0 ? 0 : (g >= p.G ? p.G - 1 : g);
    stage_block<false>(p, L, g, N);
    __syncthreads();
    if (e >= B) return;

    const uint16_t* __restrict__ ap = p.apsp + (size_t)g * N * N;
    const bool has_belief = p.st.belief != nullptr;
    const bool is_pol = lane >= 1 && lane <= P;
    const int n16 = (A * NS) >> 4;
    const ScanMap sm = make_scan_map(lane, p.scan_w);
    Coefs<false> kc;
    load_coeffs(p, lane, kc.r);
    kc.s = nullptr;
    RewardTabs tb;
    tb.exp_s = tb.cov_s = tb.nrc_s = tb.nra_s = nullptr;
    tb.exp_g = p.exp_tab; tb.cov_g = p.cov_tab; tb.n_exp = p.n_exp; tb.n_cov = p.n_cov;

    int pos_v = lane < A ? p.st.pos[(size_t)e * A + lane] : 0;
    int mon_v = lane < A ? p.st.budget[(size_t)e * A + lane] : 0;
    int t = __builtin_amdgcn_readfirstlane(p.st.t[e]);                       // wave-uniform: keep in SGPRs
    uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.st.step_count[e]);
    const int act_v = lane < A ? actions[(size_t)e * A + lane] : -1;
    float b[NR], ideg[NR];
    int slab_w[NR];
    if (has_belief) belief_load<NR>(b, ideg, slab_w, p.st.belief + (size_t)e * NS, p.inv_deg + (size_t)g * NS, lane, N);

    const int pos0_v = pos_v, mon0_v = mon_v, t0 = t;
    if (rec.mask) {        // the pre-step masks are the state's
        const uint4* src = reinterpret_cast<const uint4*>(p.st.mask + (size_t)e * A * NS);
        uint4* dst = reinterpret_cast<uint4*>(rec.mask + (size_t)e * A * NS);
        for (int i = lane; i < n16; i += kWave) dst[i] = src[i];
    }
    if (rec.belief && has_belief) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (lane + 64 * r < NS) rec.belief[(size_t)e * NS + lane + 64 * r] = b[r];
    }
    bool ok_v;
    int cost_v;
    scan_hits(L.ell_s, lane, A, p.scan_w, sm, pos_v, mon_v, act_v, ok_v, cost_v);
    const int tgt_v = ok_v ? act_v : pos_v;                                   // yard.py:168-178, :218-229
    const uint64_t skipm = __ballot(act_v == -1 || mon_v == 0);               // :210-215
    resolve_moves(lane, P, is_pol, tgt_v, skipm, cost_v, pos_v, mon_v);
    const uint64_t polm = ((1ull << P) - 1ull) << 1;
    const bool no_money = (skipm & polm) == polm;                             // :191,216
    int vc = 0;
    if (is_pol) {                                                             // :244-245
        uint16_t* vp = p.st.visits + (size_t)e * NS + pos_v;
        vc = (int)*vp + 1;
        *vp = (uint16_t)vc;
    }
    const int mrx = rdlane(pos_v, 0);
    const int row = pos_v * N;
    int dm = 0;
    int dj[SY_MAX_AGENTS - 1];
#pragma unroll
    for (int j = 1; j < SY_MAX_AGENTS; ++j) dj[j - 1] = 0;
    if (is_pol) {
        dm = (int)ap[row + mrx];
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j)
            if (j <= P) dj[j - 1] = (int)ap[row + rdlane(pos_v, j)];
    }
    uint32_t aff;
    int qcnt;
    scan_masks(L.ell_s, E.mrow, lane, A, NS, n16, p.scan_w, sm, pos_v, mon_v, aff, qcnt);
    // outcome priority (reward_calculator.py:63-90), flags shared by all agents
    const bool captured = __ballot(is_pol && pos_v == mrx) != 0ull;
    const bool timeout = t > p.max_t;  // pre-increment timestep
    const int term = (captured || (!timeout && no_money)) ? 1 : 0;
    const int trunc = (!captured && timeout) ? 1 : 0;
    const int win = captured ? 1 : ((timeout || no_money) ? 2 : 0);
    const bool ended = (term | trunc) != 0;
    double rew;
    if (ended) rew = captured ? (lane == 0 ? -1.0 : 1.0) : (lane == 0 ? 1.0 : 0.0);
    else rew = shaped_reward<false>(tb, lane, P, is_pol, t, qcnt, vc, dm, dj, kc);
    t += 1;   // yard.py:355
    sc += 1;
    if (lane < A) p.st.reward[(size_t)e * A + lane] = rew;
    if (lane == 0) {
        p.st.terminated[e] = (uint8_t)term;
        p.st.truncated[e] = (uint8_t)trunc;
        p.st.winner[e] = (int8_t)win;
    }
    if (rec.record) {      // the packed row, same layout as the fused rollout's
        const int RW = p.rec_words;
        int* rdst = rec.record + (size_t)e * RW;
        if (lane < A) {
            *reinterpret_cast<double*>(rdst + 2 * lane) = rew;
            rdst[2 * A + lane] = pos0_v;
            rdst[3 * A + lane] = mon0_v;
            rdst[4 * A + lane] = act_v;
        }
        if (lane < RW - 5 * A) rdst[5 * A + lane] = lane == 0 ? t0 : (lane == 1 ? term : (lane == 2 ? trunc : (lane == 3 ? win : 0)));
    }
    if (ended && p.auto_reset) {
        const int st = sample_starts(lane, A, N, p.env_id_offset + (uint64_t)e, sc, p.seed_lo, p.seed_hi);
        pos_v = lane < A ? st : 0;
        mon_v = lane == 0 ? SY_MRX_MONEY : (lane < A ? p.money0 : 0);   // yard.py:117-119
        t = 0;
        for (int i = lane; i < (NS >> 3); i += kWave)
            reinterpret_cast<uint4*>(p.st.visits + (size_t)e * NS)[i] = make_uint4(0, 0, 0, 0);
        if (has_belief) belief_prior<NR>(b, lane, N, p.belief_onehot != 0, rdlane(pos_v, 0));
        scan_masks(L.ell_s, E.mrow, lane, A, NS, n16, p.scan_w, sm, pos_v, mon_v, aff, qcnt);
    } else if (has_belief) {
        if (p.reveal_k > 0 && (t % p.reveal_k) == 0) {   // post-increment timestep is a multiple of reveal_k
            belief_prior<NR>(b, lane, N, true, mrx);
        } else {
            int pol[SY_MAX_AGENTS - 1];
#pragma unroll
            for (int k = 0; k < SY_MAX_AGENTS - 1; ++k) pol[k] = k < P ? rdlane(pos_v, k + 1) : -1;
            belief_step<NR>(b, ideg, slab_w, E.c_s, L.boff_s, lane, N, p.police_ev != 0, pol, P);
        }
    }
    if (lane < A) {
        p.st.pos[(size_t)e * A + lane] = pos_v;
        p.st.budget[(size_t)e * A + lane] = mon_v;
    }
    if (lane == 0) {
        p.st.t[e] = t;
        p.st.step_count[e] = sc;
    }
    {
        uint4* dst = reinterpret_cast<uint4*>(p.st.mask + (size_t)e * A * NS);
        for (int i = lane; i < n16; i += kWave) dst[i] = reinterpret_cast<const uint4*>(E.mrow)[i];
    }
    if (has_belief) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = lane + 64 * r;
            if (j < NS) p.st.belief[(size_t)e * NS + j] = b[r];
        }
    }
}

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
// rollout2_kernel: the fused rollout with PAIRED move waves.  Most of a step touches only the A <= 8
// agent lanes, so one move wave carries two episodes: lanes 0-31 hold episode `e`, lanes 32-63 episode
// `e + 1` (agent a on lane h*32 + a).  What was wave-uniform per episode (timestep, flags, MrX's node,
// ...) becomes a value replicated across the 32 lanes of a half; "any lane of my half" tests read the
// matching 32 bits of one 64-bit ballot; broadcasts from an agent lane are two v_readlane + one select.
// Only the 64-lane ELL scan and the mask-row copies run once per episode.  Instruction count per
// episode drops by about a third; block = wpb/2 move waves + wpb/2 belief waves (wpb even).
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
        const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
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
        const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
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
    __device__ __forceinline__ Pol make_pol(const EngineParams& p, int lane, int A, uint32_t pol0, uint32_t pol1) const {
        Pol q;
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
                                                const float* b2_all, int& act_v, int& cost_v, int& quirk_cnt, float& logp_v) {
        if (lanes(kAgentSlots)) {
            *lds_at<uint64_t>(selr) = 0x0000ffffull;                                     // "no move": action -1, cost 0
            typedef int v4i __attribute__((ext_vector_type(4)));
            *lds_at<v4i>(pl.slr) = (v4i){(int)0x80000000, (int)0x80000000, 0, 0};         // max key, max logit, sum exp, log-prob
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
        // logits of the affordable entries: the first column of every lane, further columns only on the steps where
        // some agent of the pair stands on a row that wide (wave-uniform)
#ifndef SY_POL_BATCH_HALF
#define SY_POL_BATCH_HALF 4
#endif
        float l[NC];
        l[0] = logit_of<SY_POL_BATCH_HALF>(pl, nb[0], H);
#pragma unroll
        for (int k = 1; k < NC; ++k) {
            l[k] = 0.0f;
            if (bo[k] != 0ull) l[k] = logit_of<SY_POL_BATCH_HALF>(pl, nb[k], H);
        }
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (lanes(bo[k])) atomicMax(lds_at_generic<int>(pl.sl + 4u), f32_ordered(l[k]));
        wave_lds_fence();
        float Lm = ordered_f32(*lds_at<int>(pl.sl + 4u));
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (lanes(bo[k])) atomicAdd(lds_at_generic<float>(pl.sl + 8u), __expf(l[k] - Lm));
        wave_lds_fence();
        float S = *lds_at<float>(pl.sl + 8u);
        // ---- the reference's underflow rule (mappo_agent.py:123-134), exact only where the cheap bound cannot rule it out
        {
            const bool lead = lanes(lead_m);
            const uint64_t sus = bal(lead && gf != 0u && !(Lm + __logf(S) > pl.thr));
            uint64_t fb = 0ull;                       // groups (leader-lane bits) that fall back to uniform over the mask
#ifndef SY_POL_NO_FALLBACK
            if (sus != 0ull) {                        // rare: evaluate the suspicious actors exactly, one (episode, agent) at a time
                uint64_t todo = sus;
                while (todo != 0ull) {
                    const int ll = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const int agu = rdlane(ag, ll);
                    const uint32_t hsu = (uint32_t)rdlane((int)pl.hs, ll);
                    const float lse = __int_as_float(rdlane(__float_as_int(Lm + __logf(S)), ll));
                    const float mass = exact_legal_mass<NR>(w2_all + (size_t)agu * N * H, b2_all + (size_t)agu * N, hsu, H, N, lane, lse);
                    if (mass <= 1e-8f) fb |= 1ull << ll;
                }
            }
#endif
            if (fb != 0ull) {                         // my group's leader bit -> my fallback flag
                const bool f = ((fb >> (lane - col)) & 1ull) != 0ull;
#pragma unroll
                for (int k = 0; k < NC; ++k) l[k] = f ? 0.0f : l[k];
                Lm = f ? 0.0f : Lm;
                S = f ? (float)__popc(gf) : S;
            }
        }
        // Gumbel-max draw: a cheap per-entry hash (ELL column) of the agent's Philox word of this step
        auto gumbel = [](uint32_t x, uint32_t column) {
            uint32_t h = x ^ (column * 0x9E3779B9u) ^ 0x85EBCA6Bu;
            h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
            const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
            return -__logf(-__logf(u));
        };
        float key[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            key[k] = l[k] + gumbel(g.xa, (uint32_t)(k * GW + col));
            if (lanes(bo[k])) atomicMax(lds_at_generic<int>(pl.sl), f32_ordered(key[k]));
        }
        wave_lds_fence();
        const int kmax = *lds_at<int>(pl.sl);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            if (lanes(bo[k]) && f32_ordered(key[k]) == kmax) {
                *lds_at<int>(selw) = (int)g.ent[k];
                *lds_at<float>(pl.sl + 12u) = (l[k] - Lm) - __logf(S);
            }
        }
        if (lanes(lead_m)) lds_at<int>(selw)[1] = __popc(qf);
        wave_lds_fence();
        const uint64_t r = *lds_at<uint64_t>(selr);
        act_v = (int)(int16_t)(uint32_t)r;              // 0xffff -> -1
        cost_v = (int)(((uint32_t)r) >> 16);
        quirk_cnt = (int)(r >> 32);
        logp_v = *lds_at<float>(pl.slr + 12u);
        wave_lds_fence();
    }
};
// columns per lane the half-wave scan needs for a pool whose widest row has max_deg entries (0: use the paired scan)
__host__ __device__ constexpr int half_scan_gw(int P) { return 32 / (P + 1) > kD ? kD : 32 / (P + 1); }

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
    const uint32_t pol1 = pol0 + (uint32_t)p.pslice;
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
    constexpr bool HALF = HS > 0 && PT >= 1 && PT <= 6;
    constexpr int GWH = HALF ? half_scan_gw(PT) : kD;
    HalfScan<GWH, (HALF ? HS : 1)> hs;
    if (HALF) hs.init(L, E, E1, lane, A, NS);
    PairScanLane psl = make_pair_scan_lane(E, E1, sm, lane, A, NS);
    PairScanLane psl2 = psl;
    if (!one_pass) psl2 = make_pair_scan_lane(E, E1, sm, lane, A, NS, sm.per_pass);
    int act_v = -1, cost_v = 0, qcnt = 0;
    typename HalfScan<GWH, (HALF ? HS : 1)>::Pol hpl;
    if (HALF && POL) hpl = hs.make_pol(p, lane, A, pol0, pol1);
    if (HALF) {
        const auto g0 = hs.gather(pos_v, mon_v, xw[0]);
        if (POL) {
            policy_hidden_pair3(p, P, A, pos_v, lane, pol0, pol1);
            wave_lds_fence();
            hs.template eval_policy<NR>(g0, hpl, p.pH, N, lane, p.pw2, p.pb2, act_v, cost_v, qcnt, logp_v);
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


    S3_DECL
    for (int s = 0; s < T; ++s) {
        int ln = lane;                               // laundered: lane predicates are recomputed every step
        asm volatile("" : "+v"(ln));
        const bool upper = ln >= 32;
        const int a = ln & 31;
        const bool is_pol = a >= 1 && a <= P;
        S3(7)

        // ---- moves (yard.py:161-243), both episodes at once
        const int tgt_v = act_v >= 0 ? act_v : pos_v;
        const uint64_t SK = bal(act_v == -1) | bal(mon_v == 0);               // skipped agents (:210-215)
        {   // MrX vs PRE-move police (:180-188)
            const int t_lo = rdlane(tgt_v, 0), t_hi = rdlane(tgt_v, 32);
            const uint32_t hit_lo = (uint32_t)bal(pos_v == t_lo) & POL32, hit_hi = (uint32_t)(bal(pos_v == t_hi) >> 32) & POL32;
            const uint64_t mrx_moves = (uint64_t)z31(hit_lo) | ((uint64_t)z31(hit_hi) << 32);   // lanes 0 / 32
            pos_v = lanes(mrx_moves) ? tgt_v : pos_v;
        }
        uint64_t CF = 0;   // any police pair that could interact this step?
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
            out.mask += mask_step;
        }
        S3(3)
        // ---- evaluate half of the scan: masks of the new state, position-reward counts, next action
        if (HALF) {
            if (POL) hs.template eval_policy<NR>(hg, hpl, p.pH, N, ln, p.pw2, p.pb2, act_v, cost_v, qcnt, logp_v);
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
template <int NR>
struct BeliefLanes {
    typedef float vNf __attribute__((ext_vector_type(NR == 1 ? 1 : NR)));
    v2f b[NR];
    float ideg[NR];
    uint32_t ga[NR][8];                                    // LDS addresses of the first eight neighbour entries
    int slab_w[NR];
    uint64_t in_m[NR];                                     // lanes whose node k exists
    uint32_t c_off, c_mine, off_bel;
    int j0;
    bool mine;

    __device__ __forceinline__ void load(const EngineParams& p, const LdsMap& L, const EnvLds& E, int lane, int e, int g, bool live1) {
        const int N = p.N, NS = p.NS;
        c_off = lds_off(E.c_s);
        j0 = NR * lane;
        mine = j0 < NS;
        c_mine = c_off + (uint32_t)j0 * 8u;
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
                if (inr) { ga[k][0] = c_off + (uint32_t)j * 8u; ideg[k] = 1.0f; }
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
            if (mine) {      // c = b / deg of my nodes: NR * 8 contiguous bytes of the interleaved scratch
                if (NR == 1) {
                    *lds_at<v2f>(c_mine) = b[0] * ideg[0];
                } else {
                    typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int k = 0; k + 1 < NR; k += 2) {
                        const v2f c0 = b[k] * ideg[k], c1 = b[k + 1] * ideg[k + 1];
                        *lds_at<v4f>(c_mine + (uint32_t)k * 8u) = (v4f){c0.x, c0.y, c1.x, c1.y};
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
    BeliefLanes<NR> bl;
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
            out.belief += bel_step;
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
            out.record += (size_t)B * RW;
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
    stage_block<true, 3>(p, L, g, N);
    const EnvLds E1 = env_lds(L.env_base, slot + 1, p.wave_lds_bytes, A, NS);
    if (!helper_role && lane == 0) {
        E.sync[0] = 0; E.sync[1] = 0;
        E1.sync[0] = 0; E1.sync[1] = 0;
    }
    __syncthreads();
    if (e >= B) return;
    if (helper_role) helper_wave3<NR, REC, PT, POL>(p, L, E, E1, lane, e, g, T, out_arg);
    else move_wave3<NR, REC, PT, POL, HS>(p, L, E, E1, lane, e, g, slot, T, out_arg);
}

// ---------------------------------------------------------------------------------------------
// reset (yard.py:80-142): new start nodes, budgets, counters, belief, masks.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void reset_kernel(const EngineParams p, const uint8_t* __restrict__ env_sel,
                                                     const int32_t* __restrict__ starts, const int zero_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    const int N = p.N, NS = p.NS, A = p.A, B = p.B;
    const int e0 = blockIdx.x * wpb;
    const int e = e0 + wid;
    const LdsMap L = lds_map(smem, N);
    const EnvLds E = env_lds(L.env_base, wid, p.wave_lds_bytes, A, NS);
    int g = __builtin_amdgcn_readfirstlane(p.env_graph[e0 < B ? e0 : B - 1]);
    g = g < 0 ? 0 : (g >= p.G ? p.G - 1 : g);
    stage_block<false>(p, L, g, N);
    __syncthreads();
    if (e >= B) return;
    if (env_sel && !__builtin_amdgcn_readfirstlane((int)env_sel[e])) return;
    uint32_t sc = zero_count ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)p.st.step_count[e]);
    int pos_v;
    if (starts) {
        int sv = lane < A ? starts[(size_t)e * A + lane] : 0;
        pos_v = sv < 0 ? 0 : (sv >= N ? N - 1 : sv);
    } else {
        const int st = sample_starts(lane, A, N, p.env_id_offset + (uint64_t)e, sc, p.seed_lo, p.seed_hi);
        pos_v = lane < A ? st : 0;
    }
    const int mon_v = lane == 0 ? SY_MRX_MONEY : (lane < A ? p.money0 : 0);
    const ScanMap sm = make_scan_map(lane, p.scan_w);
    uint32_t aff;
    int qcnt;
    scan_masks(L.ell_s, E.mrow, lane, A, NS, (A * NS) >> 4, p.scan_w, sm, pos_v, mon_v, aff, qcnt);
    if (lane < A) {
        p.st.pos[(size_t)e * A + lane] = pos_v;
        p.st.budget[(size_t)e * A + lane] = mon_v;
        p.st.reward[(size_t)e * A + lane] = 0.0;
    }
    if (lane == 0) {
        p.st.t[e] = 0;
        p.st.step_count[e] = sc;
        p.st.terminated[e] = 0;
        p.st.truncated[e] = 0;
        p.st.winner[e] = 0;
    }
    for (int i = lane; i < (NS >> 3); i += kWave)
        reinterpret_cast<uint4*>(p.st.visits + (size_t)e * NS)[i] = make_uint4(0, 0, 0, 0);
    {
        uint4* dst = reinterpret_cast<uint4*>(p.st.mask + (size_t)e * A * NS);
        for (int i = lane; i < ((A * NS) >> 4); i += kWave) dst[i] = reinterpret_cast<const uint4*>(E.mrow)[i];
    }
    if (p.st.belief) {
        const int m0 = rdlane(pos_v, 0);
        const float uni = 1.0f / (float)N;
        for (int j = lane; j < NS; j += kWave)
            p.st.belief[(size_t)e * NS + j] = j < N ? (p.belief_onehot ? (j == m0 ? 1.0f : 0.0f) : uni) : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// compute_action_mask on dense float64 matrices (action_mask.py:30-84), one thread per (query,node)
// ---------------------------------------------------------------------------------------------
__global__ void action_mask_dense_kernel(const double* __restrict__ adj, const double* __restrict__ wts,
                                         const double* __restrict__ tolls, int N, const int32_t* __restrict__ cur,
                                         const double* __restrict__ budget, int Q, uint8_t* __restrict__ mask) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)Q * N) return;
    const int q = (int)(i / N), n = (int)(i % N);
    const int c = cur[q];
    uint8_t m = 0;
    if (c >= 0 && c < N && n != c) {                                 // :66-67
        const double a = adj[(size_t)c * N + n];
        if (a != 0.0) {                                              // :68-69
            const double w = wts ? wts[(size_t)c * N + n] : a;       // :100-112
            const double toll = tolls ? tolls[(size_t)c * N + n] : 0.0;  // :87-97
            m = (w + toll <= budget[q]) ? 1 : 0;                     // :72-76
        }
    }
    mask[i] = m;
}

// ---------------------------------------------------------------------------------------------
// stand-alone belief update (ParticleBeliefTracker.update, belief_module.py:69-111), one wave per
// belief vector; hint lists give the soft likelihood 0.1 + 0.9*[j in hint]
// ---------------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(256) void belief_update_kernel(const uint32_t* __restrict__ ell,
                                                            const float* __restrict__ inv_deg, int N, int NS,
                                                            float* __restrict__ belief, const int32_t* __restrict__ hint,
                                                            int H, const int32_t* __restrict__ reveal, int Q) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    uint32_t* ell_s = reinterpret_cast<uint32_t*>(smem);
    float* c_s = reinterpret_cast<float*>(smem + (size_t)N * kD * 4) + (size_t)wid * (NS + 16);
    {
        const uint4* src = reinterpret_cast<const uint4*>(ell);
        uint4* dst = reinterpret_cast<uint4*>(ell_s);
        for (int i = threadIdx.x; i < N * 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int q = blockIdx.x * wpb + wid;
    if (q >= Q) return;
    float b[NR], ideg[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        b[r] = j < N ? belief[(size_t)q * NS + j] : 0.0f;
        ideg[r] = j < N ? inv_deg[j] : 0.0f;
    }
    const int rv = reveal ? reveal[q] : -1;
    if (rv >= 0) {   // :86-88 every particle on the revealed node
#pragma unroll
        for (int r = 0; r < NR; ++r) b[r] = (lane + 64 * r == rv) ? 1.0f : 0.0f;
    } else {
        // diffusion without normalisation, then the hint likelihood, then normalise
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = lane + 64 * r;
            if (j < N) c_s[j] = b[r] * ideg[r];
        }
        if (lane == 0) c_s[N] = 0.0f;
        wave_lds_fence();
        bool any_hint = false;
        if (hint)
            for (int h = 0; h < H; ++h) any_hint = any_hint || (hint[(size_t)q * H + h] >= 0);
        float tot = 0.0f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = lane + 64 * r;
            float acc = 0.0f;
            if (j < N) {
                const uint4* row = reinterpret_cast<const uint4*>(ell_s + j * kD);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint4 v = row[k];
                    acc += c_s[v.x & 0xffffu];
                    acc += c_s[v.y & 0xffffu];
                    acc += c_s[v.z & 0xffffu];
                    acc += c_s[v.w & 0xffffu];
                }
                if (ideg[r] == 0.0f) acc += b[r];
                if (any_hint) {
                    bool hit = false;
                    for (int h = 0; h < H; ++h) hit = hit || (hint[(size_t)q * H + h] == j);
                    acc *= hit ? 1.0f : 0.1f;                       // :102-105
                }
            }
            b[r] = acc;
            tot += acc;
        }
        tot = wave_sum(tot);
        const float uni = 1.0f / (float)N;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = lane + 64 * r;
            b[r] = j < N ? (tot == 0.0f ? uni : b[r] / tot) : 0.0f;  // :32-39
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        if (j < NS) belief[(size_t)q * NS + j] = b[r];
    }
}

// ---------------------------------------------------------------------------------------------
// All-pairs weighted shortest paths of a board pool (replaces per-query Dijkstra, pathfinding.py:34-137,
// and the host Floyd-Warshall for large pools): one wave per (board, source) runs Bellman-Ford over
// the ELL rows with the distance vector in LDS; integer weights -> exact.  Unreachable = 0xFFFF.
// ---------------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(256) void apsp_kernel(const uint32_t* __restrict__ ell, int N, uint16_t* __restrict__ apsp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
    const int g = blockIdx.y;
    uint32_t* ell_s = reinterpret_cast<uint32_t*>(smem);
    int* dist_s = reinterpret_cast<int*>(smem + (size_t)N * kD * 4) + (size_t)wid * (N + 16);
    {
        const uint4* src = reinterpret_cast<const uint4*>(ell + (size_t)g * N * kD);
        uint4* dst = reinterpret_cast<uint4*>(ell_s);
        for (int i = threadIdx.x; i < N * 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int source = blockIdx.x * wpb + wid;
    if (source >= N) return;
    constexpr int kInf = 0x3fffffff;
    int d[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) d[r] = (lane + 64 * r == source) ? 0 : kInf;
    for (int it = 0; it < N; ++it) {          // at most N-1 relaxation rounds; stops when nothing changes
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (lane + 64 * r < N) dist_s[lane + 64 * r] = d[r];
        if (lane == 0) dist_s[N] = kInf;      // padding entries point here
        wave_lds_fence();
        bool changed = false;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = lane + 64 * r;
            const int jj = j < N ? j : N - 1;
            const uint4* row = reinterpret_cast<const uint4*>(ell_s + (jj << 4));
            int nd = d[r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 v = row[q];
                nd = min(nd, dist_s[v.x & 0xffffu] + (int)(v.x >> 16));
                nd = min(nd, dist_s[v.y & 0xffffu] + (int)(v.y >> 16));
                nd = min(nd, dist_s[v.z & 0xffffu] + (int)(v.z >> 16));
                nd = min(nd, dist_s[v.w & 0xffffu] + (int)(v.w >> 16));
            }
            changed = changed || (j < N && nd < d[r]);
            d[r] = nd;
        }
        wave_lds_fence();
        if (__ballot(changed) == 0ull) break;
    }
    uint16_t* out = apsp + ((size_t)g * N + source) * N;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        if (j < N) out[j] = d[r] < 0xFFFF ? (uint16_t)d[r] : (uint16_t)0xFFFF;
    }
}

// ---------------------------------------------------------------------------------------------
// Board sampler (reset side; replaces ConnectedGraph.sample / _create_tree, graph_layout.py:9-80, for a
// whole pool): one wave per board.
//   tree   — random-Prim == random node order + uniform parent among the earlier nodes (an edge drawn
//            uniformly from visited x unvisited is exactly that), built lane-parallel;
//   extras — the reference walks a shuffled list of all non-edges and adds a pair when both degrees are
//            below the cap.  Equivalent rejection sampling: every round each lane proposes a uniform
//            pair, the first valid proposal in lane order is accepted (later lanes are discarded because
//            their validity may have changed) — one accepted edge per round, failures 64 at a time;
//   weights uniform in {1..4} (randint(1, 5)).  Own Philox streams: parity is statistical.
// Outputs the ELL rows (sorted by neighbour), 1/deg, and the edge list in insertion order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sample_boards_kernel(int N, int NS, int E_target, int max_deg_extra, uint32_t seed_lo,
                                                           uint32_t seed_hi, int G, int max_rounds,
                                                           uint32_t* __restrict__ ell, float* __restrict__ inv_deg,
                                                           int32_t* __restrict__ edge_links, int32_t* __restrict__ edge_w,
                                                           int32_t* __restrict__ num_edges, int E_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x;
    uint16_t* nbr = reinterpret_cast<uint16_t*>(smem);                 // [N][16]
    uint8_t* wgt = reinterpret_cast<uint8_t*>(nbr + (size_t)N * kD);   // [N][16]
    int* deg = reinterpret_cast<int*>(wgt + (size_t)N * kD);           // [N]
    uint16_t* perm = reinterpret_cast<uint16_t*>(deg + N);             // [N]
    int* flag = reinterpret_cast<int*>(perm + ((N + 1) & ~1));         // [2]: overflow, edge count
    int32_t* el = edge_links + (size_t)g * E_cap * 2;
    int32_t* ew = edge_w + (size_t)g * E_cap;
    for (int i = lane; i < N; i += kWave) {
        deg[i] = 0;
        perm[i] = (uint16_t)i;
    }
    if (lane == 0) { flag[0] = 0; flag[1] = 0; }
    wave_lds_fence();
    // random node order (Fisher-Yates on one lane; N <= 1024)
    if (lane == 0) {
        for (int i = N - 1; i > 0; --i) {
            uint32_t o[4];
            philox4((uint64_t)g, (uint32_t)i, 3u, 0u, seed_lo, seed_hi, o);
            const int j = (int)__umulhi(o[0], (uint32_t)(i + 1));
            const uint16_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
        }
    }
    wave_lds_fence();
    // spanning tree: child perm[i], parent perm[uniform(0..i-1)]
    for (int i = 1 + lane; i < N; i += kWave) {
        uint32_t o[4];
        philox4((uint64_t)g, (uint32_t)i, 4u, 0u, seed_lo, seed_hi, o);
        const int u = perm[__umulhi(o[0], (uint32_t)i)], v = perm[i];
        const int w = 1 + (int)__umulhi(o[1], 4u);
        const int su = atomicAdd(&deg[u], 1), sv = atomicAdd(&deg[v], 1);
        if (su < kD && sv < kD) {
            nbr[u * kD + su] = (uint16_t)v; wgt[u * kD + su] = (uint8_t)w;
            nbr[v * kD + sv] = (uint16_t)u; wgt[v * kD + sv] = (uint8_t)w;
        } else {
            flag[0] = 1;   // a row would exceed the ELL width: the host redraws this board
        }
        if (i - 1 < E_cap) { el[2 * (i - 1)] = u; el[2 * (i - 1) + 1] = v; ew[i - 1] = w; }   // (visited, new) like :66-70
    }
    wave_lds_fence();
    int edges = N - 1;
    // extra edges under the degree cap
    for (int round = 0; round < max_rounds && edges < E_target; ++round) {
        uint32_t o[4];
        philox4((uint64_t)g, (uint32_t)round, 5u, (uint32_t)lane, seed_lo, seed_hi, o);
        int a = (int)__umulhi(o[0], (uint32_t)N), b = (int)__umulhi(o[1], (uint32_t)(N - 1));
        b += b >= a ? 1 : 0;                       // uniform unordered pair of distinct nodes
        const int i = a < b ? a : b, j = a < b ? b : a;
        bool ok = deg[i] < max_deg_extra && deg[j] < max_deg_extra;   // :38-43
        if (ok) {
            const int di = deg[i];
            for (int q = 0; q < di; ++q) ok = ok && nbr[i * kD + q] != (uint16_t)j;   // not yet an edge (:28)
        }
        const uint64_t bm = __ballot(ok);
        if (bm != 0ull) {
            const int win = __ffsll((long long)bm) - 1;
            if (lane == win) {
                const int w = 1 + (int)__umulhi(o[2], 4u);
                const int si = deg[i]++, sj = deg[j]++;
                nbr[i * kD + si] = (uint16_t)j; wgt[i * kD + si] = (uint8_t)w;
                nbr[j * kD + sj] = (uint16_t)i; wgt[j * kD + sj] = (uint8_t)w;
                if (edges < E_cap) { el[2 * edges] = i; el[2 * edges + 1] = j; ew[edges] = w; }
            }
            ++edges;
            wave_lds_fence();
        }
    }
    wave_lds_fence();
    // sort every row by neighbour id and emit the packed ELL row + 1/deg
    for (int u = lane; u < NS; u += kWave) {
        if (u < N) {
            const int d = deg[u] < kD ? deg[u] : kD;
            for (int x = 1; x < d; ++x) {   // insertion sort, rows have <= 16 entries
                const uint16_t kn = nbr[u * kD + x];
                const uint8_t kw = wgt[u * kD + x];
                int y = x - 1;
                while (y >= 0 && nbr[u * kD + y] > kn) {
                    nbr[u * kD + y + 1] = nbr[u * kD + y];
                    wgt[u * kD + y + 1] = wgt[u * kD + y];
                    --y;
                }
                nbr[u * kD + y + 1] = kn;
                wgt[u * kD + y + 1] = kw;
            }
            uint32_t* row = ell + ((size_t)g * N + u) * kD;
            for (int x = 0; x < kD; ++x)
                row[x] = x < d ? ((uint32_t)nbr[u * kD + x] | ((uint32_t)wgt[u * kD + x] << 16)) : ((uint32_t)N | 0xFFFF0000u);
            inv_deg[(size_t)g * NS + u] = d > 0 ? 1.0f / (float)d : 0.0f;
        } else {
            inv_deg[(size_t)g * NS + u] = 0.0f;
        }
    }
    if (lane == 0) num_edges[g] = flag[0] ? -1 : edges;
}

// ---------------------------------------------------------------------------------------------
// masked_sample_kernel: MappoAgent.select_action's masked sampling (agent/mappo_agent.py:112-142),
// one wave per (env, agent) row.  p = probs * mask; if its sum is <= 1e-8 the row falls back to
// uniform over the mask (or over all nodes when the mask is empty), else p / (sum + 1e-8);
// Categorical(probs = p) renormalises (norm = p / sum p), samples, and reports log norm[a].
// Engine-defined draw: u = 24 bits of word 0 of Philox(seed; row, offset) in [0, 1); the action is
// the first index whose inclusive prefix sum of norm (float32, node order) exceeds u.  Elements are
// read coalesced (lane + 64 r) and the prefix is built per 64-node slab with DPP scans.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {   // inclusive prefix sum over the 64 lanes
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));   // row_shr:8
    // rows of 16 are scanned; add the totals of the rows below
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(rdlane(iv, 15)), r1 = __int_as_float(rdlane(iv, 31)), r2 = __int_as_float(rdlane(iv, 47));
    const int row = lane >> 4;
    return v + (row == 0 ? 0.0f : (row == 1 ? r0 : (row == 2 ? r0 + r1 : (r0 + r1) + r2)));
}

// The sampling core shared by masked_sample_kernel and mappo_policy_kernel: pr[] = the actor's
// probabilities of nodes lane + 64 r (0 past N), mr = the row's mask bytes.
template <int NR>
__device__ __forceinline__ void masked_sample_row(const float (&pr)[NR], const uint8_t* __restrict__ mr, int lane, int N,
                                                  uint64_t stream_row, uint32_t seed_lo, uint32_t seed_hi, uint64_t offset,
                                                  int default_on_empty, int32_t* action_out, float* logp_out,
                                                  float* norm_row) {
    float pv[NR], mv[NR];
    float s_loc = 0.0f, m_loc = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        mv[r] = (j < N && mr[j]) ? 1.0f : 0.0f;
        pv[r] = j < N ? pr[r] * mv[r] : 0.0f;
        s_loc += pv[r];
        m_loc += mv[r];
    }
    const float s = wave_sum(s_loc), msum = wave_sum(m_loc);
    // mappo_agent.py:120-129
    const bool degenerate = s <= 1e-8f;
    const bool empty = !(msum > 1e-8f);
    const float inv_s = 1.0f / (s + 1e-8f), inv_m = empty ? 0.0f : 1.0f / msum, uni = 1.0f / (float)N;
    float t_loc = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        pv[r] = degenerate ? (empty ? (j < N ? uni : 0.0f) : mv[r] * inv_m) : pv[r] * inv_s;
        t_loc += pv[r];
    }
    const float total = wave_sum(t_loc);
    const float inv_t = 1.0f / total;
    uint32_t o[4];
    philox4(stream_row, (uint32_t)offset, 3u, (uint32_t)(offset >> 32) & 0xffu, seed_lo, seed_hi, o);
    const float u = (float)(o[0] >> 8) * (1.0f / 16777216.0f);
    float base = 0.0f;
    int found = 0x7fffffff;
    float p_found = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        const float nv = pv[r] * inv_t;                       // Categorical's renormalised probability
        if (norm_row && j < N) norm_row[j] = nv;
        const float incl = base + wave_incl_scan(nv, lane);
        if (nv > 0.0f && incl > u && j < found) { found = j; p_found = nv; }
        base = __int_as_float(rdlane(__float_as_int(incl), 63));
    }
    // first hit in node order = smallest j over the lanes; rounding may leave u above the last prefix:
    // then the last node with positive probability is taken
    int last_pos = -1;
    float p_last = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        if (pv[r] > 0.0f && j > last_pos) { last_pos = j; p_last = pv[r] * inv_t; }
    }
    int best = found, bl = last_pos;
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) {
        const int ob = __shfl_xor(best, o2, kWave), ol = __shfl_xor(bl, o2, kWave);
        best = ob < best ? ob : best;
        bl = ol > bl ? ol : bl;
    }
    const int a = best != 0x7fffffff ? best : bl;
    const float pa_mine = (found == a) ? p_found : ((last_pos == a && best == 0x7fffffff) ? p_last : 0.0f);
    const float pa = wave_sum(pa_mine);                          // exactly one lane holds it
    if (lane == 0) {
        *action_out = (empty && default_on_empty) ? -1 : a;
        *logp_out = logf(pa);
    }
}

template <int NR>
__global__ __launch_bounds__(256) void masked_sample_kernel(const float* __restrict__ probs, long long probs_stride,
                                                            const uint8_t* __restrict__ mask, long long mask_stride, int rows,
                                                            int N, uint32_t seed_lo, uint32_t seed_hi, uint64_t offset_imm,
                                                            const uint64_t* __restrict__ offset_dev,
                                                            int default_on_empty, int32_t* __restrict__ action,
                                                            float* __restrict__ log_prob, float* __restrict__ norm_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* prow = probs + (size_t)row * probs_stride;
    float pr[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) pr[r] = lane + 64 * r < N ? prow[lane + 64 * r] : 0.0f;
    // a device-resident offset lets a captured HIP graph advance the stream between replays
    const uint64_t offset = offset_imm + (offset_dev ? *offset_dev : 0ull);
    masked_sample_row<NR>(pr, mask + (size_t)row * mask_stride, lane, N, (uint64_t)row, seed_lo, seed_hi, offset,
                          default_on_empty, action + row, log_prob + row, norm_out ? norm_out + (size_t)row * N : nullptr);
}

// ---------------------------------------------------------------------------------------------
// mappo_policy_kernel: MappoAgent.select_action for every (env, agent) in ONE launch — the actor MLPs
// (AgentPolicy, agent/mappo_agent.py:6-29: Linear -> ReLU -> Linear -> softmax on the trainer's
// observations, mappo_trainer.py:173,197: one-hot MrX node for MrX, multi-hot police nodes for the
// police), the masked sampling (:112-142), and the central critic (CentralCritic, :32-44, on
// [mrx] + [police] * P).  A one-hot input makes the first layer a row lookup in its transposed
// weight; the second layer of a block's 16 envs is a 16 x H x N product on the matrix cores
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate); softmax and sampling run one wave per env.
// grid.y = agent (A = the critic's blocks).
// ---------------------------------------------------------------------------------------------
struct MappoWeights {
    const float* w1t;   // [A][N][H]   first actor layers, transposed
    const float* b1;    // [A][H]
    const float* w2t;   // [A][H][N]   second actor layers, transposed
    const float* b2;    // [A][N]
    const float* c1t;   // [N * A][H]  first critic layer, transposed
    const float* cb1;   // [H]
    const float* c2;    // [H]         second critic layer
    const float* cb2;   // [1]
};

template <int NR>
__global__ __launch_bounds__(1024) void mappo_policy_kernel(const int32_t* __restrict__ pos, const uint8_t* __restrict__ mask,
                                                            long long mask_row_stride, const MappoWeights w, int B, int A, int N,
                                                            int H, uint32_t seed_lo, uint32_t seed_hi, uint64_t offset_imm,
                                                            const uint64_t* __restrict__ offset_dev, int32_t* __restrict__ action,
                                                            float* __restrict__ log_prob, float* __restrict__ value,
                                                            float* __restrict__ probs_out) {
    // block = 16 waves = 16 envs of one agent (grid.y = agent; y == A: the critic's blocks)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kHS = 65;                                      // hidden row stride (floats): conflict-free A-operand reads
    const int LS = NR * 64 + 1;                                  // logits row stride
    float* hs = reinterpret_cast<float*>(smem);                  // [16][kHS]   hidden activations of the block's 16 rows
    float* ls = hs + 16 * kHS;                                   // [16][LS]    their logits
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int a = blockIdx.y;
    const int b = blockIdx.x * 16 + wid;
    const bool live = b < B;
    const int P = A - 1;
    const int32_t* prow = pos + (size_t)(live ? b : 0) * A;
    if (a == A) {
        // ---- central critic: h = relu(cb1 + C1t[mrx] + sum_k sum_j C1t[N (1 + k) + police_j]), value = c2 . h + cb2
        if (!value || !live) return;
        float h = lane < H ? w.cb1[lane] : 0.0f;
        if (lane < H) {
            h += w.c1t[(size_t)prow[0] * H + lane];
            for (int k = 0; k < P; ++k)
                for (int j = 0; j < P; ++j) h += w.c1t[((size_t)N * (1 + k) + prow[1 + j]) * H + lane];
        }
        h = h > 0.0f ? h : 0.0f;
        const float v = wave_sum(lane < H ? h * w.c2[lane] : 0.0f) + w.cb2[0];
        if (lane == 0) value[b] = v;
        return;
    }
    // ---- phase 1: actor a's first layer by row lookups (lane k holds hidden unit k of this wave's env)
    {
        float h = lane < H ? w.b1[(size_t)a * H + lane] : 0.0f;
        if (lane < H && live) {
            const float* w1a = w.w1t + (size_t)a * N * H;
            if (a == 0) h += w1a[(size_t)prow[0] * H + lane];
            else
                for (int j = 0; j < P; ++j) h += w1a[(size_t)prow[1 + j] * H + lane];
        }
        hs[wid * kHS + lane] = (lane < H && h > 0.0f) ? h : 0.0f;
    }
    __syncthreads();
    // ---- phase 2: logits[16 envs][N] = hs[16][H] x W2t[H][N] + b2 on the matrix cores (f32 in, f32 accumulate):
    // one 16 x 16 output tile per wave and pass, H / 4 v_mfma_f32_16x16x4_f32 each; A from LDS, B straight from L2
    {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int col = lane & 15, kq = lane >> 4;
        const float* w2a = w.w2t + (size_t)a * H * N;
        const int tiles = (N + 15) >> 4;
        for (int t = wid; t < tiles; t += 16) {
            const int n = 16 * t + col;
            const float bias = n < N ? w.b2[(size_t)a * N + n] : 0.0f;
            f32x4 acc = {bias, bias, bias, bias};
            for (int k0 = 0; k0 < H; k0 += 4) {
                const int k = k0 + kq;
                const float av = hs[col * kHS + k];                               // A[i = lane & 15][k = lane >> 4]  (0 past H)
                const float bv = (n < N && k < H) ? w2a[(size_t)k * N + n] : 0.0f;   // B[k = lane >> 4][j = lane & 15]
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
            if (n < N) {
#pragma unroll
                for (int v = 0; v < 4; ++v) ls[(4 * kq + v) * LS + n] = acc[v];     // D[i = 4 (lane >> 4) + v][j = lane & 15]
            }
        }
    }
    __syncthreads();
    if (!live) return;
    // ---- phase 3: this wave's env: softmax over the N nodes, masked sampling
    float acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = lane + 64 * r < N ? ls[wid * LS + lane + 64 * r] : 0.0f;
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < NR; ++r) mx = (lane + 64 * r < N && acc[r] > mx) ? acc[r] : mx;
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) {
        const float om = __shfl_xor(mx, o2, kWave);
        mx = om > mx ? om : mx;
    }
    float pr[NR], se = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        pr[r] = lane + 64 * r < N ? __expf(acc[r] - mx) : 0.0f;
        se += pr[r];
    }
    const float inv = 1.0f / wave_sum(se);
#pragma unroll
    for (int r = 0; r < NR; ++r) pr[r] *= inv;
    const size_t row = (size_t)b * A + a;
    if (probs_out) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (lane + 64 * r < N) probs_out[row * N + lane + 64 * r] = pr[r];
    }
    const uint64_t offset = offset_imm + (offset_dev ? *offset_dev : 0ull);
    masked_sample_row<NR>(pr, mask + row * mask_row_stride, lane, N, (uint64_t)row, seed_lo, seed_hi, offset, 1, action + row,
                          log_prob + row, nullptr);
}

// ---------------------------------------------------------------------------------------------
// returns_kernel: the return / advantage lines of MappoAgent.ppo_update (agent/mappo_agent.py:247-258) for a whole
// [T][B][A] rollout in ONE launch, and their GAE(gamma, lambda) generalisation.  One lane per (env, agent) column
// walks the time axis backwards; the recurrence is sequential, the loads are not: U rows are fetched ahead of the
// arithmetic (the record's reward / terminated / truncated words of one env-step share a 128-byte line).
//   mode 0 (the reference):  R_t = r_t + (gamma * R_{t+1}) * (1 - d_t)   in exactly that operation order,
//                            adv_t = R_t - V_t   (V = 0 when no values are given)
//   mode 1 (GAE):            delta_t = (r_t + (gamma * V_{t+1}) * (1 - d_t)) - V_t
//                            A_t = delta_t + ((gamma * lambda) * (1 - d_t)) * A_{t+1},   R_t = A_t + V_t
// CT = float reproduces the reference's float32 tensors bit for bit; CT = double keeps the engine's float64 rewards.
// ---------------------------------------------------------------------------------------------
template <typename CT, typename RT, typename DT>
__global__ __launch_bounds__(256) void returns_kernel(const ReturnsArgs a) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= a.B * a.A) return;
    const int b = col / a.A, ag = col - b * a.A;
    const RT* rp = reinterpret_cast<const RT*>(a.reward) + (long long)b * a.rs_b + ag;
    const DT* da = reinterpret_cast<const DT*>(a.done_a) + (long long)b * a.ds_b;
    const DT* db = a.done_b ? reinterpret_cast<const DT*>(a.done_b) + (long long)b * a.ds_b : nullptr;
    const float* vp = a.value ? a.value + (long long)b * a.vs_b + (long long)ag * a.vs_a : nullptr;
    const long long BA = (long long)a.B * a.A;
    CT* ret = reinterpret_cast<CT*>(a.returns) + col;
    CT* adv = a.adv ? reinterpret_cast<CT*>(a.adv) + col : nullptr;
    const CT gamma = (CT)a.gamma, gl = (CT)a.gamma * (CT)a.lam;
    CT run = (CT)0;
    CT nxt = (a.mode == 1 && a.last_value) ? (CT)a.last_value[(long long)b * a.lv_b + (long long)ag * a.lv_a] : (CT)0;
    constexpr int U = 16;
    for (int t1 = a.T; t1 > 0; t1 -= U) {
        CT r[U], nd[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t1 - 1 - u;
            const bool in = t >= 0;
            const int tt = in ? t : 0;
            r[u] = (CT)rp[(long long)tt * a.rs_t];
            bool d = da[(long long)tt * a.ds_t] != 0;
            if (db) d = d || db[(long long)tt * a.ds_t] != 0;
            nd[u] = (CT)1 - (d ? (CT)1 : (CT)0);
            v[u] = vp ? (CT)vp[(long long)tt * a.vs_t] : (CT)0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t1 - 1 - u;
            if (t < 0) continue;
            CT R, A;
            if (a.mode == 0) {
                const CT gr = gamma * run;
                run = r[u] + gr * nd[u];
                R = run;
                A = run - v[u];
            } else {
                const CT delta = (r[u] + (gamma * nxt) * nd[u]) - v[u];
                run = delta + (gl * nd[u]) * run;
                A = run;
                R = run + v[u];
                nxt = v[u];
            }
            ret[(long long)t * BA] = R;
            if (adv) adv[(long long)t * BA] = A;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers (called from the C ABI, sy_capi.hip)
// ---------------------------------------------------------------------------------------------
// The half-wave neighbour scan (random policy): `cols` columns per scan lane cover the pool's widest row.  Instances:
// 2 columns for up to 5 agents (any board size up to 256 nodes); boards of 129..256 nodes also 2 / 3 columns at 6
// agents and 3 / 4 at 7 agents.  Returns false when no instance fits (the caller takes the paired scan).
template <int NR, int PT, bool REC, int COLS>
static void launch_half_instance(const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, int threads3, size_t lds,
                                 hipStream_t stream) {
    hipLaunchKernelGGL((rollout3_kernel<NR, REC, PT, false, COLS>), dim3(blocks), dim3(threads3), lds, stream, p, T, out);
}
template <int NR, int PT>
static bool launch_half_scan(const EngineParams& p, int T, const sy_rollout_buffers& out, int cols, int blocks, int threads3, size_t lds,
                             hipStream_t stream) {
    if (cols <= 0) return false;
    if constexpr (NR <= 4 && PT >= 1 && PT <= 4) {
        if (cols > 2) return false;
        if (out.record) launch_half_instance<NR, PT, true, 2>(p, T, out, blocks, threads3, lds, stream);
        else launch_half_instance<NR, PT, false, 2>(p, T, out, blocks, threads3, lds, stream);
        return true;
    } else if constexpr (NR == 4 && (PT == 5 || PT == 6)) {
        constexpr int C0 = PT == 5 ? 2 : 3, C1 = C0 + 1;
        if (cols > C1) return false;
        if (cols <= C0) {
            if (out.record) launch_half_instance<NR, PT, true, C0>(p, T, out, blocks, threads3, lds, stream);
            else launch_half_instance<NR, PT, false, C0>(p, T, out, blocks, threads3, lds, stream);
        } else {
            if (out.record) launch_half_instance<NR, PT, true, C1>(p, T, out, blocks, threads3, lds, stream);
            else launch_half_instance<NR, PT, false, C1>(p, T, out, blocks, threads3, lds, stream);
        }
        return true;
    } else {
        return false;
    }
}

template <int NR>
static hipError_t launch_engine_nr(const EngineParams& p, const int32_t* actions, int T, const sy_rollout_buffers& out,
                                   bool ext, int blocks, int wpb, size_t lds, hipStream_t stream) {
#ifdef SY_ISA_ONLY   // tools/isa_only.sh: only the headline instance, for a quick look at its ISA (not a usable library)
#ifndef SY_ISA_PT
#define SY_ISA_PT 4
#define SY_ISA_HS 2
#endif
    if (NR == 4) hipLaunchKernelGGL((rollout3_kernel<4, true, SY_ISA_PT, false, SY_ISA_HS>), dim3(blocks), dim3(64 * wpb), lds, stream, p, T, out);
    return hipGetLastError();
#else
    if (ext) {
        hipLaunchKernelGGL((step_kernel<NR>), dim3(blocks), dim3(wpb * 64), lds, stream, p, actions, out);
    } else {
        const bool paired = (wpb & 1) == 0;            // paired move waves need an even number of episodes per block
        // the move / helper pipeline (rollout3) covers boards whose agents fit one or two scan passes
        const int per_pass = 64 / p.scan_w;
#ifdef SY_NO_PIPELINE
        const bool pipelined = false, pol_pipeline = false;
        const int threads3 = 0;
#else
        // ... of up to 256 nodes (node-major belief lanes)
        const int threads3 = 64 * wpb;
#ifdef SY_POL_ROLLOUT2
        const bool pol_pipeline = false;
#else
        const bool pol_pipeline = true;
#endif
        const bool pipelined = NR <= 4 && p.A <= 2 * per_pass && p.max_t < (1 << 20) - 2;
#endif
        const int threads = paired ? 64 * (wpb / 2 + (p.st.belief ? wpb / 2 : 0))
                                   : 64 * (wpb + (p.st.belief ? (wpb + 1) / 2 : 0));   // move waves + belief waves
        // half-wave scan (random policy, up to 7 agents): columns per lane that cover the pool's widest row; 0 = paired scan.
        // Instances: 2 columns for up to 5 agents (any board size), and for boards of 129..256 nodes 2 / 3 columns at
        // 6 agents, 3 / 4 at 7 agents.
        const int hs_gw = half_scan_gw(p.P);
        const int hs_need = (p.max_deg + hs_gw - 1) / hs_gw;
        const int hs_cols = (p.A <= 7 && hs_need * hs_gw <= 32) ? hs_need : 0;
#define SY_LAUNCH_ROLLOUT(PT_)                                                                                            \
    do {                                                                                                                  \
        if (paired && pipelined && pol_pipeline && p.pw2 != nullptr && p.A <= per_pass) {                                 \
            if (PT_ == 4 && hs_cols > 0 && hs_cols <= 2)    /* 4 police: the half-wave scan, one logit stream per lane */ \
                hipLaunchKernelGGL((rollout3_kernel<(NR <= 4 ? NR : 1), true, 4, true, 2>), dim3(blocks),                  \
                                   dim3(threads3), lds + (size_t)wpb * p.pslice, stream, p, T, out);                       \
            else                                                                                                          \
            hipLaunchKernelGGL((rollout3_kernel<(NR <= 4 ? NR : 1), true, (PT_ == 4 ? 4 : 0), true>), dim3(blocks),        \
                               dim3(threads3), lds + (size_t)wpb * p.pslice, stream, p, T, out);                           \
        } else if (paired && p.pw2 != nullptr) {                                                                          \
            hipLaunchKernelGGL((rollout2_kernel<NR, true, (PT_ == 4 ? 4 : 0), true>), dim3(blocks), dim3(threads),        \
                               lds + (size_t)wpb * SY_POLICY_SLICE, stream, p, T, out);                                    \
        } else if (paired && pipelined && launch_half_scan<NR, PT_>(p, T, out, hs_cols, blocks, threads3, lds, stream)) {  \
            /* launched with the half-wave neighbour scan */                                                              \
        } else if (paired && pipelined) {                                                                                 \
            if (out.record)                                                                                               \
                hipLaunchKernelGGL((rollout3_kernel<(NR <= 4 ? NR : 1), true, PT_>), dim3(blocks), dim3(threads3), lds, stream, p, T, out);\
            else                                                                                                          \
                hipLaunchKernelGGL((rollout3_kernel<(NR <= 4 ? NR : 1), false, PT_>), dim3(blocks), dim3(threads3), lds, stream, p, T, out);\
        } else if (paired) {                                                                                              \
            if (out.record)                                                                                               \
                hipLaunchKernelGGL((rollout2_kernel<NR, true, PT_>), dim3(blocks), dim3(threads), lds, stream, p, T, out); \
            else                                                                                                          \
                hipLaunchKernelGGL((rollout2_kernel<NR, false, PT_>), dim3(blocks), dim3(threads), lds, stream, p, T, out);\
        } else if (out.record)                                                                                            \
            hipLaunchKernelGGL((rollout_kernel<NR, true, PT_>), dim3(blocks), dim3(threads), lds, stream, p, T, out);     \
        else                                                                                                              \
            hipLaunchKernelGGL((rollout_kernel<NR, false, PT_>), dim3(blocks), dim3(threads), lds, stream, p, T, out);    \
    } while (0)
        switch (p.P) {   // the BASELINE.json police counts get fully unrolled instances
            case 2: SY_LAUNCH_ROLLOUT(2); break;
            case 4: SY_LAUNCH_ROLLOUT(4); break;
            case 5: SY_LAUNCH_ROLLOUT(5); break;
            case 6: SY_LAUNCH_ROLLOUT(6); break;
            default: SY_LAUNCH_ROLLOUT(0); break;
        }
#undef SY_LAUNCH_ROLLOUT
    }
    return hipGetLastError();
#endif
}

hipError_t launch_engine(const EngineParams& p, const int32_t* actions, int T, const sy_rollout_buffers& out, bool ext,
                         int blocks, int wpb, size_t lds, hipStream_t stream) {
    const int nr = (p.N + 63) / 64;
    if (nr <= 1) return launch_engine_nr<1>(p, actions, T, out, ext, blocks, wpb, lds, stream);
    if (nr <= 2) return launch_engine_nr<2>(p, actions, T, out, ext, blocks, wpb, lds, stream);
    if (nr <= 4) return launch_engine_nr<4>(p, actions, T, out, ext, blocks, wpb, lds, stream);
    if (nr <= 8) return launch_engine_nr<8>(p, actions, T, out, ext, blocks, wpb, lds, stream);
    return launch_engine_nr<16>(p, actions, T, out, ext, blocks, wpb, lds, stream);
}

hipError_t launch_reset(const EngineParams& p, const uint8_t* env_sel, const int32_t* starts, int zero_count, int blocks,
                        int wpb, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL(reset_kernel, dim3(blocks), dim3(wpb * 64), lds, stream, p, env_sel, starts, zero_count);
    return hipGetLastError();
}

hipError_t launch_action_mask_dense(const double* adj, const double* wts, const double* tolls, int N, const int32_t* cur,
                                    const double* budget, int Q, uint8_t* mask, hipStream_t stream) {
    const long long total = (long long)Q * N;
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads);
    hipLaunchKernelGGL(action_mask_dense_kernel, dim3(blocks), dim3(threads), 0, stream, adj, wts, tolls, N, cur, budget, Q,
                       mask);
    return hipGetLastError();
}

template <int NR>
static hipError_t launch_belief_nr(const uint32_t* ell, const float* inv_deg, int N, int NS, float* belief,
                                   const int32_t* hint, int H, const int32_t* reveal, int Q, hipStream_t stream) {
    const int wpb = 4;
    const size_t lds = (size_t)N * kD * 4 + (size_t)wpb * (NS + 16) * 4;
    hipLaunchKernelGGL((belief_update_kernel<NR>), dim3((Q + wpb - 1) / wpb), dim3(wpb * 64), lds, stream, ell, inv_deg, N,
                       NS, belief, hint, H, reveal, Q);
    return hipGetLastError();
}

hipError_t launch_belief_update(const uint32_t* ell, const float* inv_deg, int N, int NS, float* belief, const int32_t* hint,
                                int H, const int32_t* reveal, int Q, hipStream_t stream) {
    const int nr = (N + 63) / 64;
    if (nr <= 1) return launch_belief_nr<1>(ell, inv_deg, N, NS, belief, hint, H, reveal, Q, stream);
    if (nr <= 2) return launch_belief_nr<2>(ell, inv_deg, N, NS, belief, hint, H, reveal, Q, stream);
    if (nr <= 4) return launch_belief_nr<4>(ell, inv_deg, N, NS, belief, hint, H, reveal, Q, stream);
    if (nr <= 8) return launch_belief_nr<8>(ell, inv_deg, N, NS, belief, hint, H, reveal, Q, stream);
    return launch_belief_nr<16>(ell, inv_deg, N, NS, belief, hint, H, reveal, Q, stream);
}

template <int NR>
static hipError_t launch_apsp_nr(const uint32_t* ell, int N, int G, uint16_t* apsp, hipStream_t stream) {
    const int wpb = 4;
    const size_t lds = (size_t)N * kD * 4 + (size_t)wpb * (N + 16) * 4;
    hipLaunchKernelGGL((apsp_kernel<NR>), dim3((N + wpb - 1) / wpb, G), dim3(wpb * 64), lds, stream, ell, N, apsp);
    return hipGetLastError();
}

hipError_t launch_apsp(const uint32_t* ell, int N, int G, uint16_t* apsp, hipStream_t stream) {
    const int nr = (N + 63) / 64;
    if (nr <= 1) return launch_apsp_nr<1>(ell, N, G, apsp, stream);
    if (nr <= 2) return launch_apsp_nr<2>(ell, N, G, apsp, stream);
    if (nr <= 4) return launch_apsp_nr<4>(ell, N, G, apsp, stream);
    if (nr <= 8) return launch_apsp_nr<8>(ell, N, G, apsp, stream);
    return launch_apsp_nr<16>(ell, N, G, apsp, stream);
}

hipError_t launch_sample_boards(int N, int NS, int E_target, int max_deg_extra, uint64_t seed, int G, uint32_t* ell,
                                float* inv_deg, int32_t* edge_links, int32_t* edge_w, int32_t* num_edges, int E_cap,
                                hipStream_t stream) {
    const size_t lds = (size_t)N * kD * 3 + (size_t)N * 4 + (size_t)((N + 1) & ~1) * 2 + 16;
    const int max_rounds = 64 * N + 4096;
    hipLaunchKernelGGL(sample_boards_kernel, dim3(G), dim3(64), lds, stream, N, NS, E_target, max_deg_extra, (uint32_t)seed,
                       (uint32_t)(seed >> 32), G, max_rounds, ell, inv_deg, edge_links, edge_w, num_edges, E_cap);
    return hipGetLastError();
}

hipError_t launch_masked_sample(const float* probs, long long probs_stride, const uint8_t* mask, long long mask_stride,
                                int rows, int N, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                int default_on_empty, int32_t* action, float* log_prob, float* norm_out, hipStream_t stream) {
    const int wpb = 4, blocks = (rows + wpb - 1) / wpb;
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    const int nr = (N + 63) / 64;
#define SY_LAUNCH_MS(NR_) hipLaunchKernelGGL((masked_sample_kernel<NR_>), dim3(blocks), dim3(wpb * 64), 0, stream, probs,    \
                                             probs_stride, mask, mask_stride, rows, N, lo, hi, offset, offset_dev,            \
                                             default_on_empty,                                                                \
                                             action, log_prob, norm_out)
    if (nr <= 1) SY_LAUNCH_MS(1);
    else if (nr <= 2) SY_LAUNCH_MS(2);
    else if (nr <= 4) SY_LAUNCH_MS(4);
    else if (nr <= 8) SY_LAUNCH_MS(8);
    else SY_LAUNCH_MS(16);
#undef SY_LAUNCH_MS
    return hipGetLastError();
}

hipError_t launch_mappo_policy(const int32_t* pos, const uint8_t* mask, long long mask_row_stride, const float* w1t,
                               const float* b1, const float* w2t, const float* b2, const float* c1t, const float* cb1,
                               const float* c2, const float* cb2, int B, int A, int N, int H, uint64_t seed, uint64_t offset,
                               const uint64_t* offset_dev, int32_t* action, float* log_prob, float* value, float* probs_out,
                               hipStream_t stream) {
    MappoWeights w{w1t, b1, w2t, b2, c1t, cb1, c2, cb2};
    const int wpb = 16;
    const dim3 grid((B + wpb - 1) / wpb, A + (value ? 1 : 0));
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    const int nr = (N + 63) / 64;
    const int nrp = nr <= 1 ? 1 : (nr <= 2 ? 2 : (nr <= 4 ? 4 : (nr <= 8 ? 8 : 16)));
    const size_t lds = (size_t)(16 * 65 + 16 * (nrp * 64 + 1)) * sizeof(float);   // hidden rows + logits rows of 16 envs
#define SY_LAUNCH_MP(NR_) hipLaunchKernelGGL((mappo_policy_kernel<NR_>), grid, dim3(wpb * 64), lds, stream, pos, mask,        \
                                             mask_row_stride, w, B, A, N, H, lo, hi, offset, offset_dev, action, log_prob,   \
                                             value, probs_out)
    if (nr <= 1) SY_LAUNCH_MP(1);
    else if (nr <= 2) SY_LAUNCH_MP(2);
    else if (nr <= 4) SY_LAUNCH_MP(4);
    else if (nr <= 8) SY_LAUNCH_MP(8);
    else SY_LAUNCH_MP(16);
#undef SY_LAUNCH_MP
    return hipGetLastError();
}

template <typename CT>
static hipError_t launch_returns_ct(const ReturnsArgs& a, hipStream_t stream) {
    const int threads = 256, blocks = (a.B * a.A + threads - 1) / threads;
#define SY_LAUNCH_RET(RT_, DT_) hipLaunchKernelGGL((returns_kernel<CT, RT_, DT_>), dim3(blocks), dim3(threads), 0, stream, a)
    if (a.reward_f64) { if (a.done_bytes == 4) SY_LAUNCH_RET(double, int32_t); else SY_LAUNCH_RET(double, uint8_t); }
    else { if (a.done_bytes == 4) SY_LAUNCH_RET(float, int32_t); else SY_LAUNCH_RET(float, uint8_t); }
#undef SY_LAUNCH_RET
    return hipGetLastError();
}

hipError_t launch_returns(const ReturnsArgs& a, hipStream_t stream) {
    return a.compute_f64 ? launch_returns_ct<double>(a, stream) : launch_returns_ct<float>(a, stream);
}

}  // namespace sy
