// sy_kernels.hip — hand-written CDNA4 (gfx950) kernels of the batched Scotland-Yard engine.
//
// Execution model: ONE 64-lane wavefront per live episode.  A launch block holds `wpb` episodes
// (waves) that share one board: the board's ELL adjacency (16 packed entries per node) is staged
// once per block in LDS; every wave owns a private LDS slice for its belief scratch vector, its
// action-mask rows and its visit counters.  Agent state lives in lanes: lane a holds agent a's
// node / budget / action (lane 0 = MrX, lane k+1 = Police k).  Membership tests ("is this node a
// neighbour", "is the target occupied", "is MrX caught") are wave ballots; the sequential move
// order of the reference (yard.py:161-243) is kept by a wave-uniform loop over v_readlane.
// No MFMA: this is gather / index work bounded by HBM traffic and LDS issue.
//
// Semantics follow the reference file:line cited at each phase (paths under
// /root/reference/src/environment/).  Compile with -ffp-contract=off: the float64 reward
// arithmetic keeps the reference's Python operation order (no fused multiply-add).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sy_kernels.h"

namespace sy {

static constexpr int kWave = 64;
static constexpr int kD = SY_ELL_WIDTH;  // 16 ELL entries per node
static constexpr uint32_t kPurposeAct = 1u, kPurposeReset = 2u;
static constexpr int kPhiloxRounds = 7;
static constexpr int kLdsTab = SY_LDS_TABLE;  // exp / coverage table entries staged in LDS

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in order; this only stops the compiler from reordering
    // the cross-lane LDS hand-offs inside a wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int bperm(int byte_addr, int v) { return __builtin_amdgcn_ds_bpermute(byte_addr, v); }

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

// Philox4x32-7 (Salmon et al. 2011; 7 rounds is the paper's Crush-resistant minimum), word 0.
__device__ __forceinline__ uint32_t philox_draw(uint64_t gid, uint32_t ctr, uint32_t purpose, uint32_t idx,
                                                uint32_t k0, uint32_t k1) {
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
    return c0;
}

// Distinct start nodes, uniform without replacement (replaces np.random.choice(N, A, replace=False),
// yard.py:112-116).  All values are wave-uniform; lane a returns agent a's start.
__device__ __noinline__ int sample_starts(int lane, int A, int N, uint64_t gid, uint32_t ctr, uint32_t k0, uint32_t k1) {
    const uint32_t xv = philox_draw(gid, ctr, kPurposeReset, (uint32_t)lane, k0, k1);
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

// Post-move scan (yard.py:297-317 masks == yard.py:420-472 node sets): 4 agents x 16 ELL entries per
// pass.  Rebuilds the wave's mask rows in LDS and returns, on lane a, agent a's 16-bit "affordable
// entry" field and the |possible_moves| count the police position reward uses — which the reference
// evaluates with agent index i instead of i+1, i.e. the budget of the PREVIOUS agent
// (reward_calculator.py:190; kept for parity).  Padding entries carry weight 0xFFFF, above any
// budget the ABI admits, so "affordable" alone identifies real neighbours.
__device__ __forceinline__ void scan_masks(const uint32_t* ell_s, uint8_t* mrow, int lane, int A, int NS, int n16,
                                           int pos_v, int mon_v, uint32_t& aff_field, int& quirk_cnt) {
    for (int i = lane; i < n16; i += kWave) reinterpret_cast<uint4*>(mrow)[i] = make_uint4(0, 0, 0, 0);
    wave_lds_fence();
    aff_field = 0;
    quirk_cnt = 0;
    const int d = lane & 15, grp = lane >> 4, sh = (lane & 3) << 4;
    for (int base = 0; base < A; base += 4) {
        const int a = base + grp;
        const int src = a < A ? a : A - 1;
        const int pa = bperm(src << 2, pos_v);
        int ma = bperm(src << 2, mon_v);
        const int mq = bperm((src > 0 ? src - 1 : 0) << 2, mon_v);
        ma = a < A ? ma : -1;
        const uint32_t ent = ell_s[(pa << 4) | d];
        const int w = (int)(ent >> 16);
        const bool own = w <= ma;
        const uint64_t bo = __ballot(own), bq = __ballot(w <= mq && a < A);
        if (own) mrow[a * NS + (int)(ent & 0xffffu)] = 1;
        if ((lane >> 2) == (base >> 2)) {
            aff_field = (uint32_t)(bo >> sh) & 0xffffu;
            quirk_cnt = __popc((uint32_t)(bq >> sh) & 0xffffu);
        }
    }
    wave_lds_fence();
}

// Membership test `action in possible_positions` (yard.py:168,218) for caller-given actions:
// lane a gets ok (affordable neighbour) and the edge cost (yard.py:234-236).
__device__ __forceinline__ void scan_hits(const uint32_t* ell_s, int lane, int A, int pos_v, int mon_v, int act_v,
                                          bool& ok, int& cost) {
    ok = false;
    cost = 0;
    const int d = lane & 15, grp = lane >> 4, sh = (lane & 3) << 4;
    for (int base = 0; base < A; base += 4) {
        const int a = base + grp;
        const int src = a < A ? a : A - 1;
        const int pa = bperm(src << 2, pos_v);
        const int ma = bperm(src << 2, mon_v);
        const int aa = bperm(src << 2, act_v);
        const uint32_t ent = ell_s[(pa << 4) | d];
        const int nbr = (int)(ent & 0xffffu), w = (int)(ent >> 16);
        const bool hit = (a < A) && (w <= ma) && (nbr == aa);
        const uint64_t bh = __ballot(hit);
        const uint32_t field = (uint32_t)(bh >> sh) & 0xffffu;
        const int from = sh + (field ? __ffs((int)field) - 1 : 0);
        const int wsel = bperm(from << 2, w);
        if ((lane >> 2) == (base >> 2)) {
            ok = field != 0;
            cost = ok ? wsel : 0;
        }
    }
}

// One diffusion + evidence step of the deterministic belief filter (belief_module.py:69-111 in
// expectation): b' = normalize((b.P) * lik), P[i][j] = adj/deg(i) (row e_i if isolated),
// reveal -> delta, zero mass -> uniform.  Belief lives in registers (NR slabs of 64 nodes); the
// scaled vector c = b/deg goes through the wave's LDS slice for the neighbour gathers.  Each slab
// gathers only as many ELL columns as its widest node has neighbours (rows are filled left to right).
template <int NR>
__device__ __forceinline__ void belief_step(float (&b)[NR], const float (&ideg)[NR], const int (&slab_w)[NR],
                                            float* c_s, const uint32_t* ell_s, int lane, int N, bool police_ev,
                                            int pos_v, int P) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        if (j < N) c_s[j] = b[r] * ideg[r];
    }
    if (lane == 0) c_s[N] = 0.0f;  // ELL padding entries point here
    wave_lds_fence();
    float tot = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        const int jj = j < N ? j : N - 1;   // tail lanes read a valid row; their result is discarded
        const uint4* row = reinterpret_cast<const uint4*>(ell_s + (jj << 4));
        float acc = ideg[r] == 0.0f ? b[r] : 0.0f;
        const int wq = slab_w[r];           // wave-uniform number of 4-entry chunks in this slab
        for (int q = 0; q < wq; ++q) {
            const uint4 v = row[q];
            acc += c_s[v.x & 0xffffu];
            acc += c_s[v.y & 0xffffu];
            acc += c_s[v.z & 0xffffu];
            acc += c_s[v.w & 0xffffu];
        }
        if (police_ev)
            for (int k = 1; k <= P; ++k)
                if (j == rdlane(pos_v, k)) acc = 0.0f;
        acc = j < N ? acc : 0.0f;
        b[r] = acc;
        tot += acc;
    }
    tot = wave_sum(tot);
    const float uni = 1.0f / (float)N;
    const float inv = tot == 0.0f ? 0.0f : 1.0f / tot;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        b[r] = j < N ? (tot == 0.0f ? uni : b[r] * inv) : 0.0f;
    }
    wave_lds_fence();
}

template <typename T>
__device__ __forceinline__ T* at_bytes(T* base, uint32_t byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}

// ---------------------------------------------------------------------------------------------
// The engine kernel: T fused env steps per launch.
//   EXT = true : actions come from the caller (sy_env_step; T == 1).
//   EXT = false: uniform-random policy inside the kernel (sy_env_rollout), trajectory recorded.
// LDS: [board ELL N*64 B][exp table 256 f64][coverage table 256 f64][per wave: belief scratch,
// mask rows, visit counters].
// ---------------------------------------------------------------------------------------------
template <int NR, bool EXT, bool REC>
__global__ __launch_bounds__(1024) void engine_kernel(const EngineParams p, const int32_t* __restrict__ actions,
                                                      const int T, const sy_rollout_buffers out_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int N = p.N, NS = p.NS, A = p.A, P = p.P, B = p.B;
    const int e0 = blockIdx.x * wpb;
    const int e = e0 + wid;

    uint32_t* ell_s = reinterpret_cast<uint32_t*>(smem);
    double* exp_s = reinterpret_cast<double*>(smem + (size_t)N * kD * 4);
    double* cov_s = exp_s + kLdsTab;
    unsigned char* wbase = reinterpret_cast<unsigned char*>(cov_s + kLdsTab) + (size_t)wid * p.wave_lds_bytes;
    float* c_s = reinterpret_cast<float*>(wbase);
    uint8_t* mrow = wbase + (size_t)(NS + 16) * 4;
    uint16_t* vis_s = reinterpret_cast<uint16_t*>(mrow + (size_t)A * NS);

    // ---- stage the block's board (ELL rows) and the reward tables in LDS: coalesced 16-byte loads
    int g = p.env_graph[e0 < B ? e0 : B - 1];
    g = g < 0 ? 0 : (g >= p.G ? p.G - 1 : g);
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.ell + (size_t)g * N * kD);
        uint4* dst = reinterpret_cast<uint4*>(ell_s);
        for (int i = threadIdx.x; i < N * 4; i += blockDim.x) dst[i] = src[i];
        for (int i = threadIdx.x; i < kLdsTab; i += blockDim.x) {
            exp_s[i] = i < p.n_exp ? p.exp_tab[i] : 0.0;
            cov_s[i] = p.cov_tab[i < p.n_cov ? i : p.n_cov - 1];
        }
    }
    __syncthreads();
    if (e >= B) return;

    const uint16_t* __restrict__ ap = p.apsp + (size_t)g * N * N;
    const uint64_t gid = p.env_id_offset + (uint64_t)e;
    const bool has_belief = p.st.belief != nullptr;
    const bool is_pol = lane >= 1 && lane <= P;
    const int n16 = (A * NS) >> 4;

    // per-lane reward coefficients (reward_calculator.py:140-148 for MrX on lane 0, :214-229 for police)
    const double k0 = lane == 0 ? p.w[4] : p.w[0];
    const double k1 = lane == 0 ? p.w[5] : p.w[1];
    const double k2 = lane == 0 ? p.w[6] : p.w[2];
    const double k3 = 1.0 - (lane == 0 ? p.w[7] : p.w[3]);
    const double k4 = p.w[9], k5 = p.w[10], k6 = p.w[8];

    // ---- load the episode state: coalesced reads of the batched tensors
    int pos_v = lane < A ? p.st.pos[(size_t)e * A + lane] : 0;
    int mon_v = lane < A ? p.st.budget[(size_t)e * A + lane] : 0;
    int t = p.st.t[e];
    uint32_t sc = p.st.step_count[e];
    for (int i = lane; i < (NS >> 3); i += kWave)
        reinterpret_cast<uint4*>(vis_s)[i] = reinterpret_cast<const uint4*>(p.st.visits + (size_t)e * NS)[i];
    float b[NR], ideg[NR];
    int slab_w[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int j = lane + 64 * r;
        b[r] = (has_belief && j < N) ? p.st.belief[(size_t)e * NS + j] : 0.0f;
        ideg[r] = (has_belief && j < N) ? p.inv_deg[(size_t)g * NS + j] : 0.0f;
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
    int rev_ctr = p.reveal_k > 0 ? p.reveal_k - (t % p.reveal_k) : 0;   // steps until the next reveal
    uint32_t aff = 0;
    int qcnt = 0;
    if (!EXT) scan_masks(ell_s, mrow, lane, A, NS, n16, pos_v, mon_v, aff, qcnt);  // rebuild the pre-step masks

    // ---- trajectory cursors: uniform base pointers advanced once per step + constant 32-bit lane offsets
    sy_rollout_buffers out = out_arg;
    const uint32_t off_small = (uint32_t)(e * A + lane) * 4u;
    const uint32_t off_env = (uint32_t)e;
    const uint32_t off_mask = (uint32_t)e * (uint32_t)(A * NS) + (uint32_t)lane * 16u;
    const uint32_t off_bel = ((uint32_t)e * (uint32_t)NS + (uint32_t)lane) * 4u;
    const size_t BA = (size_t)B * A;

    double rew = 0.0;
    int term = 0, trunc = 0, win = 0;

    for (int s = 0; s < T; ++s) {
        // ---- A. actions
        int act_v = -1, cost_v = 0;
        bool ok_v = false;
        if (EXT) {
            act_v = lane < A ? actions[(size_t)e * A + lane] : -1;
            scan_hits(ell_s, lane, A, pos_v, mon_v, act_v, ok_v, cost_v);
        } else {
            // uniform over the agent's valid mask, -1 when it is empty (random_agent.py)
            const int k = __popc(aff);
            const uint32_t x = philox_draw(gid, sc, kPurposeAct, (uint32_t)lane, p.seed_lo, p.seed_hi);
            const int r = (int)__umulhi(x, (uint32_t)k);
            uint32_t f = aff;
            for (int i = 0; i < r; ++i) f &= f - 1;
            const int bit = f ? __ffs((int)f) - 1 : 0;
            const uint32_t ent = ell_s[(pos_v << 4) | bit];
            ok_v = (lane < A) && (k > 0);
            act_v = ok_v ? (int)(ent & 0xffffu) : -1;
            cost_v = ok_v ? (int)(ent >> 16) : 0;
            // ---- B. record the pre-step observation and the action
            if (REC && lane < A) {
                *at_bytes(out.pos, off_small) = pos_v;
                *at_bytes(out.budget, off_small) = mon_v;
                *at_bytes(out.action, off_small) = act_v;
            }
            if (REC && lane == 0) out.t[off_env] = t;
            if (REC && out.mask) {
                for (int i = lane; i < n16; i += kWave)
                    *reinterpret_cast<uint4*>(out.mask + off_mask + (uint32_t)(i - lane) * 16u) =
                        reinterpret_cast<const uint4*>(mrow)[i];
            }
            if (REC && out.belief && has_belief) {
#pragma unroll
                for (int r2 = 0; r2 < NR; ++r2)
                    if (lane + 64 * r2 < NS) *at_bytes(out.belief, off_bel + 256u * r2) = b[r2];
            }
        }

        // ---- C. moves.  MrX first against the PRE-move police (yard.py:161-188) ...
        const int tgt_v = ok_v ? act_v : pos_v;                                   // :168-178, :218-229
        const uint64_t skipm = __ballot(act_v == -1 || mon_v == 0);               // :210-215
        {
            const int tgt = rdlane(tgt_v, 0);
            const bool blocked = __ballot(is_pol && pos_v == tgt) != 0ull;
            if (!blocked && lane == 0) pos_v = tgt;
        }
        // ... then police strictly in index order, each seeing earlier moves (yard.py:191-243)
        for (int k = 1; k <= P; ++k) {
            const int tgt = rdlane(tgt_v, k);
            const bool occ = __ballot(is_pol && pos_v == tgt) != 0ull;            // own node included (:231)
            if (!occ && !((skipm >> k) & 1ull) && lane == k) {
                pos_v = tgt;
                mon_v -= cost_v;                                                  // :234-236
            }
        }
        const uint64_t polm = ((1ull << P) - 1ull) << 1;
        const bool no_money = (skipm & polm) == polm;                             // :191,216
        // node_visit_counts (yard.py:244-245): police never share a node, so no conflicts
        int vc = 0;
        if (is_pol) {
            vc = (int)vis_s[pos_v] + 1;
            vis_s[pos_v] = (uint16_t)vc;
        }
        const int mrx = rdlane(pos_v, 0);
        // shortest-path lookups for the shaped rewards are issued now and consumed after the scan
        const int row = pos_v * N;
        int dm = 0x7fffffff;
        int dj[SY_MAX_AGENTS - 1];
#pragma unroll
        for (int j = 1; j < SY_MAX_AGENTS; ++j) dj[j - 1] = 0;
        if (is_pol) {
            dm = (int)ap[row + mrx];
#pragma unroll
            for (int j = 1; j < SY_MAX_AGENTS; ++j)
                if (j <= P) dj[j - 1] = (int)ap[row + rdlane(pos_v, j)];
        }

        // ---- F. post-move scan: masks for the next observation + position-reward counts
        scan_masks(ell_s, mrow, lane, A, NS, n16, pos_v, mon_v, aff, qcnt);

        // ---- D. outcome priority (reward_calculator.py:63-90), flags shared by all agents
        const bool captured = __ballot(is_pol && pos_v == mrx) != 0ull;
        const bool timeout = t > p.max_t;  // pre-increment timestep
        term = (captured || (!timeout && no_money)) ? 1 : 0;
        trunc = (!captured && timeout) ? 1 : 0;
        win = captured ? 1 : ((timeout || no_money) ? 2 : 0);
        const bool ended = (term | trunc) != 0;
        if (ended) {
            rew = captured ? (lane == 0 ? -1.0 : 1.0) : (lane == 0 ? 1.0 : 0.0);
        } else {
            // shaped rewards (reward_calculator.py:94-266) in float64, reference operation order
            int mn = 0x7fffffff, sum = 0;
            for (int k = 1; k <= P; ++k) {
                const int dk = rdlane(dm, k);
                mn = dk < mn ? dk : mn;
                sum += dk;
            }
            const double ts = (double)t;
            const double cntd = (double)qcnt;
            if (lane == 0) {
                const double closest = (double)mn, avg = (double)sum / (double)P;
                rew = ((k0 * (-1.0 / (closest + 1.0)) + k1 * (-1.0 / (avg + 1.0))) + k2 * cntd) + k3 * (0.1 * ts);
            } else if (is_pol) {
                double group = 0.0, overlap = 0.0, prox = 0.0;
#pragma unroll
                for (int j = 1; j < SY_MAX_AGENTS; ++j) {
                    if (j <= P && j != lane) {
                        const int dij = dj[j - 1];
                        double ex;
                        if (EXT) ex = dij < p.n_exp ? p.exp_tab[dij] : 0.0;
                        else ex = dij < kLdsTab ? exp_s[dij] : (dij < p.n_exp ? p.exp_tab[dij] : 0.0);
                        group += ex;
                        if (dij <= 1) overlap += 1.0;
                        else prox += ex;
                    }
                }
                double e_mrx, cov;
                if (EXT) {
                    e_mrx = dm < p.n_exp ? p.exp_tab[dm] : 0.0;
                    cov = p.cov_tab[vc < p.n_cov ? vc : p.n_cov - 1];
                } else {
                    e_mrx = dm < kLdsTab ? exp_s[dm] : (dm < p.n_exp ? p.exp_tab[dm] : 0.0);
                    cov = vc < kLdsTab ? cov_s[vc] : p.cov_tab[vc < p.n_cov ? vc : p.n_cov - 1];
                }
                rew = (((((k0 * e_mrx + k1 * group) + k2 * cntd) + k3 * (0.05 * ts)) + k4 * prox) - k5 * overlap) + k6 * cov;
            }
        }
        t += 1;   // yard.py:355
        sc += 1;
        if (REC) {
            if (lane < A) *at_bytes(out.reward, off_small * 2u) = rew;
            if (lane == 0) {
                out.terminated[off_env] = (uint8_t)term;
                out.truncated[off_env] = (uint8_t)trunc;
                out.winner[off_env] = (int8_t)win;
            }
            out.pos += BA; out.budget += BA; out.action += BA; out.reward += BA;
            out.t += B; out.terminated += B; out.truncated += B; out.winner += B;
            if (out.mask) out.mask += BA * NS;
            if (out.belief) out.belief += (size_t)B * NS;
        }

        // ---- E. next episode (auto-reset) or belief update for the new positions
        if (ended && p.auto_reset) {
            const int st = sample_starts(lane, A, N, gid, sc, p.seed_lo, p.seed_hi);
            pos_v = lane < A ? st : 0;
            mon_v = lane == 0 ? SY_MRX_MONEY : (lane < A ? p.money0 : 0);   // yard.py:117-119
            t = 0;
            rev_ctr = p.reveal_k;
            for (int i = lane; i < (NS >> 3); i += kWave) reinterpret_cast<uint4*>(vis_s)[i] = make_uint4(0, 0, 0, 0);
            if (has_belief) {
                const int m0 = rdlane(pos_v, 0);
                const float uni = 1.0f / (float)N;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int j = lane + 64 * r;
                    b[r] = j < N ? (p.belief_onehot ? (j == m0 ? 1.0f : 0.0f) : uni) : 0.0f;
                }
            }
            wave_lds_fence();
            scan_masks(ell_s, mrow, lane, A, NS, n16, pos_v, mon_v, aff, qcnt);
        } else {
            bool reveal = false;
            if (p.reveal_k > 0 && --rev_ctr == 0) {   // post-increment timestep is a multiple of reveal_k
                reveal = true;
                rev_ctr = p.reveal_k;
            }
            if (has_belief) {
                if (reveal) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) b[r] = (lane + 64 * r == mrx) ? 1.0f : 0.0f;
                } else {
                    belief_step<NR>(b, ideg, slab_w, c_s, ell_s, lane, N, p.police_ev != 0, pos_v, P);
                }
            }
        }
    }

    // ---- write the live state back (coalesced)
    if (lane < A) {
        p.st.pos[(size_t)e * A + lane] = pos_v;
        p.st.budget[(size_t)e * A + lane] = mon_v;
        p.st.reward[(size_t)e * A + lane] = rew;
    }
    if (lane == 0) {
        p.st.t[e] = t;
        p.st.step_count[e] = sc;
        p.st.terminated[e] = (uint8_t)term;
        p.st.truncated[e] = (uint8_t)trunc;
        p.st.winner[e] = (int8_t)win;
    }
    for (int i = lane; i < (NS >> 3); i += kWave)
        reinterpret_cast<uint4*>(p.st.visits + (size_t)e * NS)[i] = reinterpret_cast<const uint4*>(vis_s)[i];
    {
        uint4* dst = reinterpret_cast<uint4*>(p.st.mask + (size_t)e * A * NS);
        for (int i = lane; i < n16; i += kWave) dst[i] = reinterpret_cast<const uint4*>(mrow)[i];
    }
    if (has_belief) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = lane + 64 * r;
            if (j < NS) p.st.belief[(size_t)e * NS + j] = b[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// reset (yard.py:80-142): new start nodes, budgets, counters, belief, masks.
// ---------------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(1024) void reset_kernel(const EngineParams p, const uint8_t* __restrict__ env_sel,
                                                     const int32_t* __restrict__ starts, const int zero_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int N = p.N, NS = p.NS, A = p.A, B = p.B;
    const int e0 = blockIdx.x * wpb;
    const int e = e0 + wid;
    uint32_t* ell_s = reinterpret_cast<uint32_t*>(smem);
    unsigned char* wbase = smem + (size_t)N * kD * 4 + 2 * kLdsTab * sizeof(double) + (size_t)wid * p.wave_lds_bytes;
    uint8_t* mrow = wbase + (size_t)(NS + 16) * 4;
    int g = p.env_graph[e0 < B ? e0 : B - 1];
    g = g < 0 ? 0 : (g >= p.G ? p.G - 1 : g);
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.ell + (size_t)g * N * kD);
        uint4* dst = reinterpret_cast<uint4*>(ell_s);
        for (int i = threadIdx.x; i < N * 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    if (e >= B) return;
    if (env_sel && !env_sel[e]) return;
    uint32_t sc = zero_count ? 0u : p.st.step_count[e];
    int pos_v;
    if (starts) {
        int sv = lane < A ? starts[(size_t)e * A + lane] : 0;
        pos_v = sv < 0 ? 0 : (sv >= N ? N - 1 : sv);
    } else {
        const int st = sample_starts(lane, A, N, p.env_id_offset + (uint64_t)e, sc, p.seed_lo, p.seed_hi);
        pos_v = lane < A ? st : 0;
    }
    const int mon_v = lane == 0 ? SY_MRX_MONEY : (lane < A ? p.money0 : 0);
    uint32_t aff;
    int qcnt;
    scan_masks(ell_s, mrow, lane, A, NS, (A * NS) >> 4, pos_v, mon_v, aff, qcnt);
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
        for (int i = lane; i < ((A * NS) >> 4); i += kWave) dst[i] = reinterpret_cast<const uint4*>(mrow)[i];
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
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
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
// host-side launchers (called from the C ABI, sy_capi.hip)
// ---------------------------------------------------------------------------------------------
template <int NR>
static hipError_t launch_engine_nr(const EngineParams& p, const int32_t* actions, int T, const sy_rollout_buffers& out,
                                   bool ext, int blocks, int threads, size_t lds, hipStream_t stream) {
    if (ext) {
        hipLaunchKernelGGL((engine_kernel<NR, true, false>), dim3(blocks), dim3(threads), lds, stream, p, actions, T, out);
    } else if (out.pos) {
        hipLaunchKernelGGL((engine_kernel<NR, false, true>), dim3(blocks), dim3(threads), lds, stream, p, actions, T, out);
    } else {
        hipLaunchKernelGGL((engine_kernel<NR, false, false>), dim3(blocks), dim3(threads), lds, stream, p, actions, T, out);
    }
    return hipGetLastError();
}

hipError_t launch_engine(const EngineParams& p, const int32_t* actions, int T, const sy_rollout_buffers& out, bool ext,
                         int blocks, int threads, size_t lds, hipStream_t stream) {
    const int nr = (p.N + 63) / 64;
    if (nr <= 1) return launch_engine_nr<1>(p, actions, T, out, ext, blocks, threads, lds, stream);
    if (nr <= 2) return launch_engine_nr<2>(p, actions, T, out, ext, blocks, threads, lds, stream);
    if (nr <= 4) return launch_engine_nr<4>(p, actions, T, out, ext, blocks, threads, lds, stream);
    if (nr <= 8) return launch_engine_nr<8>(p, actions, T, out, ext, blocks, threads, lds, stream);
    return launch_engine_nr<16>(p, actions, T, out, ext, blocks, threads, lds, stream);
}

hipError_t launch_reset(const EngineParams& p, const uint8_t* env_sel, const int32_t* starts, int zero_count, int blocks,
                        int threads, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL((reset_kernel<1>), dim3(blocks), dim3(threads), lds, stream, p, env_sel, starts, zero_count);
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

}  // namespace sy
