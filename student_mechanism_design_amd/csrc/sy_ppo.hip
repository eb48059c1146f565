// sy_ppo.hip — one PPO minibatch of the MAPPO networks: loss AND gradient in one launch (sy_mappo_ppo_grad).
//
// What it replaces: the loss + backward of MappoAgent.ppo_update (/root/reference/src/agent/mappo_agent.py:247-293) as
// student_mechanism_design_amd/update.py::MappoUpdater._losses states it batched over a rollout record — the clipped
// surrogate (epsilon `clip`) of every agent's own action under its actor, on the masked, renormalised softmax of
// select_action (mappo_agent.py:112-134), and the central critic's MSE against the team return.  The torch form builds
// [A, mb, N] logits (131 MB per 32 768-row minibatch), gathers <= 16 of the 200 columns and runs ~100 kernels forward and
// backward: 1.9 ms per minibatch on MI355X.  Here every (row, agent) evaluates only what the loss reads:
//
//   h      = relu(b1 + sum of the W1t rows of the observation's nodes)           (one-hot MrX node / multi-hot police nodes)
//   l_e    = b2[n_e] + W2[n_e] . h     for the AFFORDABLE entries e of the agent's ELL row only (<= 16, ~3 on average)
//   new_lp = l_act - logsumexp_e l_e;  ratio, clipped surrogate;  d l_e = G (1[e = act] - p_e)
//   dW2[n_e] += d l_e h;  db2[n_e] += d l_e;  dh = sum_e d l_e W2[n_e];  dz = dh [z > 0];  dW1t[node] += dz;  db1 += dz
//
// Grid: a block owns ONE of a network's two [N][H] gradient tables (A actors + the critic: 2 (A + 1) roles; tables larger
// than the LDS are cut into row ranges, one role each) and keeps it in LDS as float64 — 102 KB at N = 200, H = 64 —, so
// every update is an LDS add; both blocks of a network run the forward pass (it is cheap: the adds are the cost).  Blocks
// write their tables to a partial buffer, a second kernel sums the partials (no global atomics).
// Lane layout: every 16-lane group of a wave works on its own row (four rows per wave in flight), lane j of the group
// holding 16-byte piece j of the hidden vector (pieces j, j + 16 for hidden > 64): a table row is one 256-byte request of
// the group, a dot product 4 FMAs + 4 DPP steps inside the group, the softmax over the affordable entries a DPP reduction
// with entry j's logit on lane j.  The loop is a chain of dependent lookups (row index -> record words -> ELL row + W1t
// rows -> W2 rows): the first two hops are prefetched one iteration ahead, the W2 rows of four entries are in flight
// together, and 16 waves per CU overlap the rest.  Summation order differs from a BLAS matmul: parity with the torch form
// is to float32 rounding (tests: 1e-4 relative on every gradient).
#include <type_traits>

#include "sy_device.hpp"

namespace sy {

typedef float ppo_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row16_sum(float v) {     // every lane of a 16-lane group gets the group's sum
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x128>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x128>(v));
    return v;
}
// The gradient tables are float64 in LDS.  On gfx950 ds_add_f32 runs ~20x slower than ds_add_u32 / ds_add_f64
// (tools/probes/lds_atomic_probe.hip: 12 363 vs 606 / 634 ticks for 16 bytes per lane from 16 waves) — the first version
// of this kernel, float32 tables, spent 2.7 of its 3.3 ms in them.  float64 adds run at the integer rate and make the
// sums independent of the order the waves arrive in (to float32 rounding of the final value).
__device__ __forceinline__ void lds_add(double* p, float v) {
#if defined(SY_PPO_DIAG_NO_ADD)          // timing-only diagnostic (wrong results): what do the LDS atomics cost?
    asm volatile("" ::"v"(p), "v"(v));
#else
    __hip_atomic_fetch_add(p, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
__device__ __forceinline__ void lds_add4(double* p, ppo_f4 v) {
    lds_add(p, v.x); lds_add(p + 1, v.y); lds_add(p + 2, v.z); lds_add(p + 3, v.w);
}
__device__ __forceinline__ float dot4(ppo_f4 a, ppo_f4 b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ ppo_f4 relu4(ppo_f4 z) {
    ppo_f4 h;
    h.x = fmaxf(z.x, 0.0f); h.y = fmaxf(z.y, 0.0f); h.z = fmaxf(z.z, 0.0f); h.w = fmaxf(z.w, 0.0f);
    return h;
}
__device__ __forceinline__ ppo_f4 gate4(ppo_f4 z, ppo_f4 d) {    // d where z > 0
    ppo_f4 o;
    o.x = z.x > 0.0f ? d.x : 0.0f; o.y = z.y > 0.0f ? d.y : 0.0f; o.z = z.z > 0.0f ? d.z : 0.0f; o.w = z.w > 0.0f ? d.w : 0.0f;
    return o;
}

// What a group needs of its row before any table lookup, read from the minibatch IMAGE (ppo_pack_kernel below): the
// shuffled rows of an update, packed once — 16 bytes of agent nodes, 16 bytes per (row, agent), 8 bytes of critic target +
// board — so that the 2 (A + 1) roles stream consecutive rows instead of each gathering ~5 scattered 128-byte lines per row
// from a record larger than the L2s (250 MB of random reads per 32 768-row launch: a 50 us floor under the first version).
struct PpoRow {
    int posv;       // lane j < A: node of agent j before the step
    int bud, act;   // the actor's agent: budget before the step, recorded action
    float olp, adv; // recorded log-probability, advantage           (critic: adv = the team return)
    int g;          // board of the row's env
    bool on;        // the group has a row (the tail of a minibatch may not fill all four groups)
};
struct PpoImage {
    const uint16_t* posq;   // [rows][8] agent nodes
    const int4* agent;      // [A][rows] {budget, action, log-prob bits, advantage bits}
    const int2* tail;       // [rows] {team return bits, board}
};
__host__ __device__ inline size_t ppo_image_bytes(int A, long long rows) { return (size_t)rows * (16 + 16 * (size_t)A + 8); }
__host__ __device__ inline PpoImage ppo_image(const void* base, int A, long long rows) {
    PpoImage im;
    const char* b = reinterpret_cast<const char*>(base);
    im.posq = reinterpret_cast<const uint16_t*>(b);
    im.agent = reinterpret_cast<const int4*>(b + (size_t)rows * 16);
    im.tail = reinterpret_cast<const int2*>(b + (size_t)rows * (16 + 16 * (size_t)A));
    return im;
}
template <bool ACTOR>
__device__ __forceinline__ PpoRow ppo_fetch_row(const PpoArgs& p, const PpoImage& im, int row0, int i, int a, int j) {
    PpoRow w;
    w.on = i < p.mb;
    const size_t r = (size_t)row0 + (size_t)(w.on ? i : p.mb - 1);
    w.posv = (int)im.posq[r * 8 + (j & 7)];
    const int2 t = im.tail[r];
    if (ACTOR) {
        const int4 q = im.agent[(size_t)a * p.image_rows + r];
        w.bud = q.x; w.act = q.y; w.olp = __int_as_float(q.z); w.adv = __int_as_float(q.w);
    } else {
        w.bud = 0; w.act = 0; w.olp = 0.0f;
        w.adv = __int_as_float(t.x);
    }
    w.g = t.y;
    return w;
}

// A pseudo-random permutation of [0, R) without a sort: a 4-round balanced Feistel network on 2 * hb bits (2^(2 hb) >= R)
// keyed by the seed, cycle-walked back into range — a bijection, so every row is visited exactly once per update.
// (torch.randperm on the device is a key sort: ~120 us for 262 144 rows, as much as a minibatch.)
__device__ __forceinline__ uint32_t ppo_shuffle(uint32_t x, uint32_t R, int hb, uint32_t k0, uint32_t k1) {
    const uint32_t mask = (1u << hb) - 1u;
    do {
        uint32_t l = x >> hb, r = x & mask;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t f = (r + (k & 1 ? k1 : k0) + (uint32_t)k * 0x9E3779B9u) * 0x85EBCA6Bu;
            f ^= f >> 15; f *= 0xC2B2AE35u; f ^= f >> 13;
            const uint32_t nl = r;
            r = l ^ (f & mask);
            l = nl;
        }
        x = (l << hb) | r;
    } while (x >= R);
    return x;
}

// Pack the rows of an update: image row i <- record row rows[i] (rows == nullptr: row0 + i, or the shuffle of i).
__global__ __launch_bounds__(256) void ppo_pack_kernel(const PpoPackArgs p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.count) return;
    const int A = p.A;
    long long r = p.rows ? (long long)p.rows[i] : (long long)p.row0 + i;
    if (!p.rows && p.shuffle_domain > 0)
        r = (long long)p.row0 + ppo_shuffle((uint32_t)i, (uint32_t)p.shuffle_domain, p.shuffle_hb, (uint32_t)p.shuffle_seed, (uint32_t)(p.shuffle_seed >> 32));
    // chunked records (the arenas of several ranks after the all-gather, one chunk per rank): row r = chunk r / chunk_rows
    size_t rec_off = (size_t)r * p.RW, lp_off = (size_t)r * A;
    if (p.chunk_rows > 0) {
        const long long ch = r / p.chunk_rows, rr = r - ch * p.chunk_rows;
        rec_off = (size_t)ch * p.record_chunk_stride + (size_t)rr * p.RW;
        lp_off = (size_t)ch * p.log_prob_chunk_stride + (size_t)rr * A;
    }
    const int32_t* const rec = p.record + rec_off;
    const float* const lp = p.log_prob + lp_off;
    char* const base = reinterpret_cast<char*>(p.image);
    uint16_t* const posq = reinterpret_cast<uint16_t*>(base) + (size_t)i * 8;
    int4* const agent = reinterpret_cast<int4*>(base + (size_t)p.count * 16);
    int2* const tail = reinterpret_cast<int2*>(base + (size_t)p.count * (16 + 16 * (size_t)A));
    uint16_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = k < A ? (uint16_t)rec[2 * A + k] : (uint16_t)0;
    *reinterpret_cast<uint4*>(posq) = *reinterpret_cast<const uint4*>(q);
    for (int k = 0; k < A; ++k)
        agent[(size_t)k * p.count + i] = make_int4(rec[3 * A + k], rec[4 * A + k], __float_as_int(lp[k]),
                                                   __float_as_int(p.adv[(size_t)r * A + k]));
    tail[i] = make_int2(__float_as_int(p.team_ret[r]), p.env_graph[r % p.B]);
}

#ifndef SY_PPO_RPG
#define SY_PPO_RPG 1      // rows per 16-lane group and pass of the actor roles (2: two interleaved chains per group, see below)
#endif
#ifdef SY_PPO_DIAG_TIMES
#define PPO_STAMP(k) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[k] += (unsigned)(tn - tl); tl = tn; }
#else
#define PPO_STAMP(k)
#endif
template <int KP, bool WL>   // KP: 16-byte pieces of a hidden vector per lane, 1 (hidden <= 64) or 2 (<= 128);
                             // WL: the network's second-layer table (critic: its police block) is staged in LDS
__global__ __launch_bounds__(1024) void ppo_grad_kernel(const PpoArgs p) {
    extern __shared__ double acc[];
    // role index y = ((network * 2) + table) * parts + part: this block accumulates rows [n0, n1) of ONE table of one
    // network.  Roles cost differently (the critic's police table takes P adds per row behind a short forward pass; MrX's
    // actor has every entry affordable): the launcher gives each role a share of the grid in proportion (p.first[y] = its
    // first block), so that all roles end together.
#ifdef SY_PPO_DIAG_TIMES        // timing-only diagnostic: every block's start / end on the constant 100 MHz clock, in the tail of the scratch
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned ph[5] = {0, 0, 0, 0, 0};
    unsigned long long tl = __builtin_amdgcn_s_memtime();
#endif
    int y = 0;
    while (y + 1 < p.nroles && (int)blockIdx.x >= p.first[y + 1]) ++y;
    const int bx = (int)blockIdx.x - p.first[y], nbx = p.first[y + 1] - p.first[y];     // my index among the role's blocks
#ifdef SY_PPO_DIAG_ONLY          // timing-only diagnostic: one role's blocks work, the others leave at once
    if (y != SY_PPO_DIAG_ONLY) return;
#endif
    const int N = p.N, H = p.H, NH = N * H, A = p.A, P = A - 1;
    const int role = y / (2 * p.parts);                               // network: actor a < A, or the critic (A)
    const int tab = (y / p.parts) & 1, part = y % p.parts;
    const int n0 = part * p.rpp, n1 = min(N, n0 + p.rpp);
    double* const gT = acc;                              // table 0: d W1t[a] (critic: d C1m, the first layer's MrX block); table 1: d W2[a]
                                                         // (critic: d C1p, every police block's gradient); rows n0 .. n1 - 1, [H] each
    double* const gC = acc + (size_t)p.rpp * H;          // [H]   d b1[a] (table 0, part 0)            critic: d cb1
    double* const gD = gC + H;                           // [DN]  d b2[a] ([N]; table 1, part 0)       critic: d c2 ([H]; table 0, part 0)
    double* const gE = gD + p.DN;                        // [8]   0: loss sum (actor: table 1 part 0; critic: table 0 part 0), 1: d cb2
    // WL: W2[a] (critic: c1p) as float32 [N][H] behind the sums.  Every row of the minibatch reads 4-8 rows of it: from L2
    // that is most of the kernel's 1.4 GB of L1 <- L2 traffic per 32 768-row launch (11 TB/s: the kernel's bound before).
    // The block's passes are handed out by ticket (an LDS counter), not by a fixed stride per wave: the SIMD arbiter serves the
    // OLDEST ready wave first, so with equal shares waves 0-3 of a block left their loop at 74 us, waves 12-15 at 105-110
    // (-DSY_PPO_DIAG_TIMES), and the block waited for its youngest waves; with tickets the fast waves take more passes.
    int* const ticket = reinterpret_cast<int*>(gE + 8);
    float* const wl = reinterpret_cast<float*>(gE + 10);
    const bool smalls = part == 0 && tab == (role < A ? 1 : 0);   // this block also owns the loss (+ b2 / critic head) sums
    const bool own_b1 = part == 0 && tab == 0;                    // ... the first layer's bias sums
    {
        const int tot = p.rpp * H + H + p.DN + 8;
        for (int k = threadIdx.x; k < tot; k += blockDim.x) acc[k] = 0.0;
        if (threadIdx.x == 0) *ticket = 2 * (int)(blockDim.x >> 6);     // (every wave starts with passes `wave` and W + wave)
        if (WL) {
            const ppo_f4* const src = reinterpret_cast<const ppo_f4*>(p.params + (size_t)role * p.slab + NH);
            for (int k = threadIdx.x; k < (NH >> 2); k += blockDim.x) reinterpret_cast<ppo_f4*>(wl)[k] = src[k];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int u = lane >> 4, j = lane & 15;          // group u of the wave works on its own row; j: piece of the hidden vector
    const int nq = H >> 2;
    int pk[KP];
    bool pv[KP];
#pragma unroll
    for (int m = 0; m < KP; ++m) {
        pv[m] = j + 16 * m < nq;
        pk[m] = pv[m] ? 4 * (j + 16 * m) : 0;
    }
    const ppo_f4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int stride = nbx * W * 4;
    // pass c of the block (c = k * W + w: what wave w did in its k-th pass under the fixed stride) covers the rows from here on
    auto pass_row = [&](int c, int rpg) { return (bx * W + c % W) * 4 * rpg + (c / W) * stride * rpg + u; };
    auto take_ticket = [&]() {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(t);
    };
    const PpoImage im = ppo_image(p.image, A, p.image_rows);
    const int row0 = p.row0_dev ? *p.row0_dev : p.row0;        // (a device word: a captured graph replays on every minibatch)
    if (p.adam_step && blockIdx.x == 0 && threadIdx.x == 0) *p.adam_step += 1;   // Adam's step count: read by the reduction launch
    float loss = 0.0f;
    if (role < A) {
        // ---------------------------------------------------------------- actor `role`
        const int a = role;
        const float* const th = p.params + (size_t)a * p.slab;      // actor a: W1t [N][H] | W2 [N][H] | b1 [H] | b2 [N]
        const float* const w1 = th;
        const float* const w2 = WL ? wl : th + NH;
        const float* const b2 = th + 2 * NH + H;
#ifdef SY_PPO_DIAG_NOZ
        const int nit = 0;
#else
        const int nit = a == 0 ? 1 : P;                      // nodes of the observation: MrX's / all police
#endif
        const float inv = 1.0f / ((float)p.mb * (float)A);
        const float lo = 1.0f - p.clip, hi = 1.0f + p.clip;
        ppo_f4 b1p[KP], gb1[KP];
#pragma unroll
        for (int m = 0; m < KP; ++m) {
            b1p[m] = pv[m] ? *reinterpret_cast<const ppo_f4*>(th + 2 * NH + pk[m]) : zero4;
            gb1[m] = zero4;
        }
        // RPG rows per group and pass, their independent chains of lookups interleaved statement by statement.  A pass is a chain
        // of dependent LDS / L2 round trips (broadcast -> row -> DPP sum -> ... -> LDS add) and a SIMD is ~1/3 busy at four
        // waves, so two rows per group looked like the way to overlap two chains — measured 138 vs 133 us per 32 768 rows
        // (-DSY_PPO_RPG=2: the per-row cost does not move; the CU's one LDS pipe, which every broadcast, staged row and
        // float64 add goes through, is what the 16 waves queue for).  Kept as a parameter, default 1.
        constexpr int RPG = KP == 1 ? SY_PPO_RPG : 1;          // (hidden > 64: two pieces per lane already fill the registers)
        int ia = pass_row(wave, RPG), ia1 = pass_row(W + wave, RPG);      // this pass, the next (the one after comes by ticket)
        // Software pipeline over passes: the record words of a row are fetched TWO passes ahead (nx2), its ELL row and the W1t rows
        // of its observation ONE pass ahead (entN, rowsN -> zN at the end of the pass): the waves of a block spent a third of
        // their time in s_waitcnt vmcnt for this pass's lookups (SQ_WAIT_ANY 51 % of wave cycles, SQ_WAIT_INST_LDS 16 %), and
        // loads return in order, so everything a pass itself needs from memory (only b2 of its entries) is requested first.
        constexpr int MAXIT = KP == 1 ? 4 : 1;        // W1t rows requested a pass ahead (registers); further rows of the observation are
                                                      // read when the pass ends
        PpoRow nx[RPG], nx2[RPG];
        uint32_t entN[RPG];
        ppo_f4 zN[RPG][KP];
        auto issue_lookups = [&](const PpoRow& w, uint32_t& e, ppo_f4 (&rows)[MAXIT][KP]) {
            const int node_a = __shfl(w.posv, a, 16);
            // the agent's ELL row: lane j holds entry j; affordable entries are compacted to lanes 0 .. n - 1 of the group later
            e = p.ell[((size_t)w.g * N + node_a) * SY_ELL_WIDTH + j];
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                if (it < nit) {                                  // (uniform: the W1t rows of the observation's nodes, all in flight)
                    const int node = __shfl(w.posv, a == 0 ? 0 : 1 + it, 16);
#pragma unroll
                    for (int m = 0; m < KP; ++m) rows[it][m] = pv[m] ? *reinterpret_cast<const ppo_f4*>(w1 + (size_t)node * H + pk[m]) : zero4;
                }
            }
        };
        auto finish_lookups = [&](const ppo_f4 (&rows)[MAXIT][KP], ppo_f4 (&zz)[KP]) {
#pragma unroll
            for (int m = 0; m < KP; ++m) zz[m] = b1p[m];
#pragma unroll
            for (int it = 0; it < MAXIT; ++it)
                if (it < nit) {
#pragma unroll
                    for (int m = 0; m < KP; ++m) zz[m] += rows[it][m];
                }
        };
        auto late_lookups = [&](const PpoRow& w, ppo_f4 (&zz)[KP]) {         // rows MAXIT .. nit - 1 of the observation
            for (int it = MAXIT; it < nit; ++it) {
                const int node = __shfl(w.posv, a == 0 ? 0 : 1 + it, 16);
#pragma unroll
                for (int m = 0; m < KP; ++m)
                    if (pv[m]) zz[m] += *reinterpret_cast<const ppo_f4*>(w1 + (size_t)node * H + pk[m]);
            }
        };
#pragma unroll
        for (int s = 0; s < RPG; ++s) {
            nx[s] = ppo_fetch_row<true>(p, im, row0, ia + 4 * s, a, j);
            ppo_f4 rows0[MAXIT][KP];
            issue_lookups(nx[s], entN[s], rows0);
            finish_lookups(rows0, zN[s]);
            late_lookups(nx[s], zN[s]);
            nx2[s] = ppo_fetch_row<true>(p, im, row0, ia1 + 4 * s, a, j);
        }
        for (; __builtin_amdgcn_readfirstlane(ia - u) < p.mb;) {
            PPO_STAMP(4)
            PpoRow rw[RPG];
            uint32_t ent[RPG];
            ppo_f4 z[RPG][KP], h[RPG][KP];
#pragma unroll
            for (int s = 0; s < RPG; ++s) {
                rw[s] = nx[s];
                ent[s] = entN[s];
#pragma unroll
                for (int m = 0; m < KP; ++m) {
                    z[s][m] = zN[s][m];
                    h[s][m] = relu4(z[s][m]);
                }
            }
            PPO_STAMP(0)
            int n[RPG], nbj[RPG];
            float b2j[RPG];
            int nm = 0;
#pragma unroll
            for (int s = 0; s < RPG; ++s) {
                const bool legal = rw[s].on && (int)(ent[s] >> 16) <= rw[s].bud;
                const uint32_t L = (uint32_t)(bal(legal) >> (16 * u)) & 0xffffu;
                n[s] = __popc(L);
                const int below = __popc(L & ((1u << j) - 1u));
                const int dstl = legal ? below : n[s] + (j - below);            // a permutation of the group's lanes
                const uint32_t cent = (uint32_t)__builtin_amdgcn_ds_permute((16 * u + dstl) << 2, (int)ent[s]);
                nbj[s] = (int)(cent & 0xffffu);                                  // lane j < n: node of affordable entry j
                b2j[s] = j < n[s] ? b2[nbj[s]] : 0.0f;                           // (requested now, needed after the logits)
                nm = max(nm, n[s]);
            }
            // after this pass's own request (b2 above): the lookups of the NEXT pass, then the words of the pass after it
            ppo_f4 rowsN[RPG][MAXIT][KP];
#pragma unroll
            for (int s = 0; s < RPG; ++s) {
                nx[s] = nx2[s];
                issue_lookups(nx[s], entN[s], rowsN[s]);
            }
            const int ia2 = pass_row(take_ticket(), RPG);
#pragma unroll
            for (int s = 0; s < RPG; ++s) nx2[s] = ppo_fetch_row<true>(p, im, row0, ia2 + 4 * s, a, j);
#ifdef SY_PPO_DIAG_NOLOGIT
            const int nmax = 0;
#else
            const int nmax = __builtin_amdgcn_readfirstlane(max(max(__shfl(nm, 0), __shfl(nm, 16)), max(__shfl(nm, 32), __shfl(nm, 48))));
#endif
            // logits: entry e of every group per step; lane j keeps the logit of entry j.  Rounds of FOUR entries' broadcasts and
            // rows in flight, then rounds of TWO for what is left: a row has ~4 affordable entries, the widest of a wave's
            // rows ~6, and every slot of a round costs its instructions whether a group still has an entry or not (rounds of
            // 4 + 4: 134 us per 32 768 rows; one round of 8: 154).
            float lgj[RPG];
#pragma unroll
            for (int s = 0; s < RPG; ++s) lgj[s] = -3.0e38f;
            auto logits_round = [&](auto cs, int e0) {
                constexpr int CS = decltype(cs)::value;
                ppo_f4 wq[RPG][CS][KP];
#pragma unroll
                for (int s = 0; s < RPG; ++s)
#pragma unroll
                    for (int t = 0; t < CS; ++t) {
                        const int e = e0 + t;
                        int nb = __shfl(nbj[s], e < n[s] ? e : 0, 16);
                        nb = n[s] > 0 ? nb : 0;
#pragma unroll
                        for (int m = 0; m < KP; ++m)
                            wq[s][t][m] = pv[m] ? *reinterpret_cast<const ppo_f4*>(w2 + (size_t)nb * H + pk[m]) : zero4;
                    }
#pragma unroll
                for (int s = 0; s < RPG; ++s)
#pragma unroll
                    for (int t = 0; t < CS; ++t) {
                        const int e = e0 + t;
                        float d = 0.0f;
#pragma unroll
                        for (int m = 0; m < KP; ++m) d += dot4(wq[s][t][m], h[s][m]);
                        d = row16_sum(d);
                        lgj[s] = (j == e && e < n[s]) ? d : lgj[s];
                    }
            };
            if (nmax > 0) logits_round(std::integral_constant<int, 4>{}, 0);
            for (int e0 = 4; e0 < nmax; e0 += 2) logits_round(std::integral_constant<int, 2>{}, e0);
            PPO_STAMP(1)
            bool mine[RPG];
            float G[RPG], dlj[RPG];
#pragma unroll
            for (int s = 0; s < RPG; ++s) {
                mine[s] = j < n[s];
                if (mine[s]) lgj[s] += b2j[s];
                const float mx = row16_max(mine[s] ? lgj[s] : -3.0e38f);
                const float ex = mine[s] ? __expf(lgj[s] - mx) : 0.0f;
                const float se = row16_sum(ex);
                const bool hit = mine[s] && nbj[s] == rw[s].act;
                const float cm = row16_sum(hit ? 1.0f : 0.0f);
                const float la = row16_sum(hit ? lgj[s] : 0.0f);
                // clipped surrogate (mappo_agent.py:284-291) and its derivative with respect to the new log-probability
                const bool valid = rw[s].act >= 0 && cm > 0.0f;       // an agent without a legal action: ratio 1, no gradient
                const float lse = valid ? mx + __logf(se) : 0.0f;
                const float icm = valid ? 1.0f / cm : 0.0f;
                const float new_lp = valid ? la * icm - lse : 0.0f;
                const float ratio = __expf(new_lp - (valid ? rw[s].olp : 0.0f));
                const float s1 = ratio * rw[s].adv, s2 = fminf(fmaxf(ratio, lo), hi) * rw[s].adv;
                if (rw[s].on && j == 0) loss -= fminf(s1, s2) * inv;
                const bool within = ratio >= lo && ratio <= hi;
                G[s] = (rw[s].on && valid && (within || s1 < s2)) ? -inv * rw[s].adv * ratio : 0.0f;   // (a clipped sample has no gradient)
                dlj[s] = (mine[s] && G[s] != 0.0f) ? G[s] * ((hit ? icm : 0.0f) - __expf(lgj[s] - lse)) : 0.0f;   // d loss / d logit of entry j
            }
            PPO_STAMP(2)
            bool anyg = false;
#pragma unroll
            for (int s = 0; s < RPG; ++s) anyg = anyg | (G[s] != 0.0f);
            if (bal(anyg) != 0ull) {
#pragma unroll
                for (int s = 0; s < RPG; ++s)
                    if (smalls && mine[s] && G[s] != 0.0f) lds_add(gD + nbj[s], dlj[s]);
                if (tab == 1) {
                    // ---- the W2 table: d W2[n_e] += d l_e h — no second-layer rows, no hidden-layer gradient
                    auto w2_round = [&](auto cs, int e0) {
                        constexpr int CS = decltype(cs)::value;
                        int nbs[RPG][CS];
                        float dls[RPG][CS];
#pragma unroll
                        for (int s = 0; s < RPG; ++s)
#pragma unroll
                            for (int t = 0; t < CS; ++t) {
                                const int e = e0 + t;
                                const int src = e < n[s] ? e : 0;
                                nbs[s][t] = __shfl(nbj[s], src, 16);
                                const float dl = __shfl(dlj[s], src, 16);
                                dls[s][t] = e < n[s] ? dl : 0.0f;
                            }
#pragma unroll
                        for (int s = 0; s < RPG; ++s)
#pragma unroll
                            for (int t = 0; t < CS; ++t) {
                                if (dls[s][t] != 0.0f && nbs[s][t] >= n0 && nbs[s][t] < n1) {
#pragma unroll
                                    for (int m = 0; m < KP; ++m)
                                        if (pv[m]) lds_add4(gT + (size_t)(nbs[s][t] - n0) * H + pk[m], dls[s][t] * h[s][m]);
                                }
                            }
                    };
                    if (nmax > 0) w2_round(std::integral_constant<int, 4>{}, 0);
                    for (int e0 = 4; e0 < nmax; e0 += 2) w2_round(std::integral_constant<int, 2>{}, e0);
                } else {
                    // ---- the W1t table: dh = sum_e d l_e W2[n_e], dz = dh [z > 0], d W1t[node] += dz for the observation's nodes
                    ppo_f4 dh[RPG][KP];
#pragma unroll
                    for (int s = 0; s < RPG; ++s)
#pragma unroll
                        for (int m = 0; m < KP; ++m) dh[s][m] = zero4;
                    auto dh_round = [&](auto cs, int e0) {
                        constexpr int CS = decltype(cs)::value;
                        ppo_f4 wq[RPG][CS][KP];
                        float dls[RPG][CS];
#pragma unroll
                        for (int s = 0; s < RPG; ++s)
#pragma unroll
                            for (int t = 0; t < CS; ++t) {
                                const int e = e0 + t;
                                const int src = e < n[s] ? e : 0;
                                int nb = __shfl(nbj[s], src, 16);
                                nb = n[s] > 0 ? nb : 0;
                                const float dl = __shfl(dlj[s], src, 16);
                                dls[s][t] = e < n[s] ? dl : 0.0f;
#pragma unroll
                                for (int m = 0; m < KP; ++m)
                                    wq[s][t][m] = pv[m] ? *reinterpret_cast<const ppo_f4*>(w2 + (size_t)nb * H + pk[m]) : zero4;
                            }
#pragma unroll
                        for (int s = 0; s < RPG; ++s)
#pragma unroll
                            for (int t = 0; t < CS; ++t)
#pragma unroll
                                for (int m = 0; m < KP; ++m) dh[s][m] += dls[s][t] * wq[s][t][m];
                    };
                    if (nmax > 0) dh_round(std::integral_constant<int, 4>{}, 0);
                    for (int e0 = 4; e0 < nmax; e0 += 2) dh_round(std::integral_constant<int, 2>{}, e0);
                    ppo_f4 dz[RPG][KP];
#pragma unroll
                    for (int s = 0; s < RPG; ++s)
#pragma unroll
                        for (int m = 0; m < KP; ++m) {
                            dz[s][m] = gate4(z[s][m], dh[s][m]);
                            gb1[m] += dz[s][m];
                        }
                    for (int it = 0; it < nit; ++it) {
#pragma unroll
                        for (int s = 0; s < RPG; ++s) {
                            const int node = __shfl(rw[s].posv, a == 0 ? 0 : 1 + it, 16);
                            if (G[s] != 0.0f && node >= n0 && node < n1) {
#pragma unroll
                                for (int m = 0; m < KP; ++m)
                                    if (pv[m]) lds_add4(gT + (size_t)(node - n0) * H + pk[m], dz[s][m]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < RPG; ++s) { finish_lookups(rowsN[s], zN[s]); late_lookups(nx[s], zN[s]); }   // (requested a pass ago)
            PPO_STAMP(3)
            ia = ia1;
            ia1 = ia2;
        }
        if (own_b1) {
#pragma unroll
            for (int m = 0; m < KP; ++m)
                if (pv[m]) lds_add4(gC + pk[m], gb1[m]);
        }
    } else {
        // ---------------------------------------------------------------- the central critic (mappo_agent.py:32-44, :260-265)
        const float inv = 1.0f / (float)p.mb;
        const float* const th = p.params + (size_t)A * p.slab;      // critic: C1m [N][H] | C1p [N][H] | cb1 [H] | c2 [H] (DN) | 8: [1] = cb2
        ppo_f4 cb1p[KP], c2p[KP], gcb1[KP], gc2[KP];
#pragma unroll
        for (int m = 0; m < KP; ++m) {
            cb1p[m] = pv[m] ? *reinterpret_cast<const ppo_f4*>(th + 2 * NH + pk[m]) : zero4;
            c2p[m] = pv[m] ? *reinterpret_cast<const ppo_f4*>(th + 2 * NH + H + pk[m]) : zero4;
            gcb1[m] = zero4;
            gc2[m] = zero4;
        }
        const float cb2 = th[2 * NH + H + p.DN + 1];
        float gcb2 = 0.0f;
        int i = pass_row(wave, 1), i1 = pass_row(W + wave, 1);
        PpoRow nx = ppo_fetch_row<false>(p, im, row0, i, 0, j);
        for (; __builtin_amdgcn_readfirstlane(i - u) < p.mb;) {
            const PpoRow rw = nx;
            ppo_f4 z[KP], hc[KP];
#pragma unroll
            for (int m = 0; m < KP; ++m) z[m] = cb1p[m];
            for (int it = 0; it < A; ++it) {                         // item 0: MrX's node on c1m; items 1..P: police nodes on c1p
                const int node = __shfl(rw.posv, it, 16);
                const float* const tab = it == 0 ? th : (WL ? wl : th + NH);
#pragma unroll
                for (int m = 0; m < KP; ++m)
                    if (pv[m]) z[m] += *reinterpret_cast<const ppo_f4*>(tab + (size_t)node * H + pk[m]);
            }
            nx = ppo_fetch_row<false>(p, im, row0, i1, 0, j);      // (after this iteration's lookups: see the actors)
            float d = 0.0f;
#pragma unroll
            for (int m = 0; m < KP; ++m) {
                hc[m] = relu4(z[m]);
                d += dot4(c2p[m], hc[m]);
            }
            const float err = rw.on ? row16_sum(d) + cb2 - rw.adv : 0.0f;
            if (j == 0) {
                loss += err * err * inv;
                gcb2 += p.value_coef * 2.0f * err * inv;
            }
            const float dv = p.value_coef * 2.0f * err * inv;
            ppo_f4 dz[KP];
#pragma unroll
            for (int m = 0; m < KP; ++m) {
                gc2[m] += dv * hc[m];
                dz[m] = gate4(z[m], dv * c2p[m]);
                gcb1[m] += dz[m];
            }
            for (int it = tab; it < (tab == 0 ? 1 : A); ++it) {      // table 0: MrX's node; table 1: the police nodes
                const int node = __shfl(rw.posv, it, 16);
                if (rw.on && node >= n0 && node < n1) {
#pragma unroll
                    for (int m = 0; m < KP; ++m)
                        if (pv[m]) lds_add4(gT + (size_t)(node - n0) * H + pk[m], dz[m]);
                }
            }
            i = i1;
            i1 = pass_row(take_ticket(), 1);
        }
        if (smalls) {
#pragma unroll
            for (int m = 0; m < KP; ++m) {
                if (pv[m]) {
                    lds_add4(gC + pk[m], gcb1[m]);
                    lds_add4(gD + pk[m], gc2[m]);
                }
            }
            if (j == 0) lds_add(gE + 1, gcb2);
        }
    }
    if (smalls && j == 0) lds_add(gE, loss);
#ifdef SY_PPO_DIAG_TIMES      // when did every wave leave its row loop?
    if (lane == 0) {
        unsigned long long* tw = reinterpret_cast<unsigned long long*>(p.partial + (size_t)(SY_PPO_MAX_BLOCKS_PER_ROLE - 1) * (A + 1) * p.slab);
        tw[3 * 256 + 4 * 256 + 16 * blockIdx.x + wave] = __builtin_amdgcn_s_memrealtime() - t_begin;
    }
#endif
    __syncthreads();
    // this block's share of the network's slab [d table 0 (N*H) | d table 1 (N*H) | H | DN | 8]: every region has one owner
    float* const dst = p.partial + ((size_t)bx * (A + 1) + role) * p.slab;
    {
        float* const dt = dst + (size_t)tab * NH + (size_t)n0 * H;
        const int cnt = (n1 - n0) * H;
        for (int k = threadIdx.x; k < cnt; k += blockDim.x) dt[k] = (float)gT[k];
    }
    if (role < A ? own_b1 : smalls)
        for (int k = threadIdx.x; k < H; k += blockDim.x) dst[2 * NH + k] = (float)gC[k];
    if (smalls)
        for (int k = threadIdx.x; k < p.DN + 8; k += blockDim.x) dst[2 * NH + H + k] = (float)gD[k];
#ifdef SY_PPO_DIAG_TIMES
    if (threadIdx.x == 0) {
        unsigned long long* tw = reinterpret_cast<unsigned long long*>(p.partial + (size_t)(SY_PPO_MAX_BLOCKS_PER_ROLE - 1) * (A + 1) * p.slab);
        tw[3 * blockIdx.x] = t_begin;
        tw[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        tw[3 * blockIdx.x + 2] = (unsigned long long)y;
        unsigned* pw = reinterpret_cast<unsigned*>(tw + 3 * 256) + 8 * blockIdx.x;
        for (int k = 0; k < 5; ++k) pw[k] = ph[k];
    }
#endif
}

// grads[t] = sum over the blocks of a role of their partial tables; with an optimiser state the Adam step of
// torch.optim.Adam (no weight decay, no amsgrad) follows in the same thread: exp_avg = b1 m + (1 - b1) g,
// exp_avg_sq = b2 v + (1 - b2) g^2, param -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
// The slab has slots that are not parameters (the loss sums, padding): they are left alone.  The critic's police table is
// the SUM of its P blocks, every one of which takes the step: the sum moves P steps.
__global__ __launch_bounds__(256) void ppo_reduce_kernel(const float* __restrict__ partial, const PpoGrid gr, int total, float* __restrict__ out,
                                                         const PpoAdam ad, int A, int N, int H, int DN, int slab) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int role = t / slab, off = t - role * slab, NH = N * H;
    // the role that owns this word of the network's slab (ppo_grad_kernel's epilogue), and how many blocks it had
    int tab, part = 0;
    if (off < 2 * NH) {
        tab = off >= NH ? 1 : 0;
        part = ((off - tab * NH) / H) / gr.rpp;
    } else {
        tab = (role < A && off >= 2 * NH + H) ? 1 : 0;       // actor: b1 sums with table 0, b2 + loss with table 1; critic: all with table 0
    }
    const int y = (role * 2 + tab) * gr.parts + part;
    const int nb = gr.first[y + 1] - gr.first[y];
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = 0;
    for (; b + 4 <= nb; b += 4) {
        s0 += partial[(size_t)b * total + t];
        s1 += partial[(size_t)(b + 1) * total + t];
        s2 += partial[(size_t)(b + 2) * total + t];
        s3 += partial[(size_t)(b + 3) * total + t];
    }
    for (; b < nb; ++b) s0 += partial[(size_t)b * total + t];
    const float g = (s0 + s1) + (s2 + s3);
    out[t] = g;
    if (ad.params == nullptr) return;
    const int oE = 2 * NH + H + DN;
    const bool is_param = role < A ? off < 2 * NH + H + N : (off < 2 * NH + 2 * H || off == oE + 1);
    if (!is_param) return;
    const int step = *ad.step;                          // (advanced by the gradient launch)
    const float bc1 = 1.0f - powf(ad.beta1, (float)step), bc2 = 1.0f - powf(ad.beta2, (float)step);
    const float m = ad.beta1 * ad.m[t] + (1.0f - ad.beta1) * g;
    const float v = ad.beta2 * ad.v[t] + (1.0f - ad.beta2) * g * g;
    ad.m[t] = m;
    ad.v[t] = v;
    const float denom = sqrtf(v) / sqrtf(bc2) + ad.eps;
    const float scale = (role == A && off >= NH && off < 2 * NH) ? (float)(A - 1) : 1.0f;
    ad.params[t] -= scale * (ad.lr / bc1) * (m / denom);
}

// The Adam step alone, on a gradient slab that is already summed (data-parallel training: the ranks' slabs are all-reduced
// between the gradient launch and this one).  Same rule as the fused form above.
__global__ __launch_bounds__(256) void ppo_adam_kernel(const float* __restrict__ grads, int total, const PpoAdam ad, int A, int N, int H,
                                                       int DN, int slab) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int role = t / slab, off = t - role * slab, NH = N * H, oE = 2 * NH + H + DN;
    const bool is_param = role < A ? off < 2 * NH + H + N : (off < 2 * NH + 2 * H || off == oE + 1);
    if (!is_param) return;
    const float g = grads[t];
    const int step = *ad.step;                          // (advanced by ppo_adam_tick_kernel, the launch before)
    const float bc1 = 1.0f - powf(ad.beta1, (float)step), bc2 = 1.0f - powf(ad.beta2, (float)step);
    const float m = ad.beta1 * ad.m[t] + (1.0f - ad.beta1) * g;
    const float v = ad.beta2 * ad.v[t] + (1.0f - ad.beta2) * g * g;
    ad.m[t] = m;
    ad.v[t] = v;
    const float denom = sqrtf(v) / sqrtf(bc2) + ad.eps;
    const float scale = (role == A && off >= NH && off < 2 * NH) ? (float)(A - 1) : 1.0f;
    ad.params[t] -= scale * (ad.lr / bc1) * (m / denom);
}
__global__ void ppo_adam_tick_kernel(int32_t* step) { *step += 1; }

// ---- launchers
hipError_t launch_ppo_adam(const float* grads, const PpoAdam& ad, int A, int N, int H, hipStream_t stream) {
    const int DN = ((N > H ? N : H) + 3) & ~3, slab = ppo_slab_floats(N, H), total = (A + 1) * slab;
    hipLaunchKernelGGL(ppo_adam_tick_kernel, dim3(1), dim3(1), 0, stream, ad.step);
    hipLaunchKernelGGL(ppo_adam_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, grads, total, ad, A, N, H, DN, slab);
    return hipGetLastError();
}
int ppo_slab_floats(int N, int H) {
    const int dn = ((N > H ? N : H) + 3) & ~3;
    return 2 * N * H + H + dn + 8;
}
// rows of one [N][H] float64 table a block can hold beside the small vectors (one block per CU: 160 KB of LDS), with
// (`staged`) or without the float32 copy of the network's second-layer table
static int ppo_rows_per_part(int N, int H, bool staged) {
    const int dn = ((N > H ? N : H) + 3) & ~3;
    long long room = 160 * 1024 - (long long)(H + dn + 10) * 8 - (staged ? (long long)N * H * 4 : 0);
    if (room < 0) room = 0;
    long long rpp = room / ((long long)H * 8);
    if (rpp > N) rpp = N;
    return (int)rpp;
}
int ppo_parts(int N, int H) {
    const int rpp = ppo_rows_per_part(N, H, false);
    return rpp < 1 ? 0 : (N + rpp - 1) / rpp;
}
// stage the weights when that does not cut the gradient table into more row ranges
static bool ppo_staged(int N, int H) {
    const int rpp = ppo_rows_per_part(N, H, true);
    return rpp >= 1 && (N + rpp - 1) / rpp == ppo_parts(N, H);
}
// The grid: every role's share of ~one block per CU, in proportion to what a pass over 64 rows costs it INSIDE a full launch
// (block start / end stamps, -DSY_PPO_DIAG_TIMES, 200 nodes / hidden 64 / 4 police: MrX's W1t table 3.5 us, its W2 table
// 5.4 — every row of MrX has a gradient and ~4 affordable entries —, a police actor's tables 5.1 / 4.7, the critic's MrX
// block 1.7, its police block 4.5 = P adds per row).  With equal shares the launch waited for MrX's W2 table (148 us);
// with these every role ends within 106-118 us (136 us per call; re-tuned after the passes went to tickets).  Time roles
// INSIDE a full launch: a build that runs one role alone leaves the other tables unwritten, the reduction then produces a
// garbage gradient, and after one Adam step NaN parameters make every launch skip its backward pass.
static void ppo_grid(int A, int N, int H, int mb, PpoGrid& g) {
    g.parts = ppo_parts(N, H);
    g.rpp = (N + g.parts - 1) / g.parts;
    g.nroles = 2 * (A + 1) * g.parts;
    const int P = A - 1;
    double w[SY_PPO_MAX_ROLES], tot = 0.0;
    for (int y = 0; y < g.nroles; ++y) {
        const int net = y / (2 * g.parts), tab = (y / g.parts) & 1;
        w[y] = net == 0 ? (tab == 0 ? 0.60 : 1.04) : (net < A ? (tab == 0 ? 0.96 : 0.87) : (tab == 0 ? 0.37 : 0.216 * P));
        tot += w[y];
    }
#ifdef SY_PPO_DIAG_BUDGET          // timing-only diagnostic: fewer blocks than CUs (is a pass slower because the chip is full?)
    const int budget = SY_PPO_DIAG_BUDGET;
#else
    const int budget = 256 > g.nroles ? 256 : g.nroles;                 // one 1024-thread block per CU: 256 CUs
#endif
    const int need = (mb + 64 * SY_PPO_RPG - 1) / (64 * SY_PPO_RPG);       // a 16-wave block takes 64 (x rows per group) rows per pass
    int used = 0;
    g.first[0] = 0;
    for (int y = 0; y < g.nroles; ++y) {
        int nb = (int)(budget * w[y] / tot);
        if (nb < 1) nb = 1;
        if (nb > need) nb = need;
        if (nb > SY_PPO_MAX_BLOCKS_PER_ROLE) nb = SY_PPO_MAX_BLOCKS_PER_ROLE;
        used += nb;
        g.first[y + 1] = used;
    }
}
int ppo_max_blocks_per_role() { return SY_PPO_MAX_BLOCKS_PER_ROLE; }

size_t ppo_image_size(int A, long long rows) { return ppo_image_bytes(A, rows); }

hipError_t launch_ppo_pack(const PpoPackArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(ppo_pack_kernel, dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_ppo_grad(PpoArgs a, float* grads, const PpoAdam& adam, hipStream_t stream) {
    a.adam_step = adam.params ? adam.step : nullptr;
    a.DN = ((a.N > a.H ? a.N : a.H) + 3) & ~3;
    a.slab = ppo_slab_floats(a.N, a.H);
    if (ppo_parts(a.N, a.H) < 1 || 2 * (a.A + 1) * ppo_parts(a.N, a.H) > SY_PPO_MAX_ROLES) return hipErrorInvalidValue;
    PpoGrid g;
    ppo_grid(a.A, a.N, a.H, a.mb, g);
    a.parts = g.parts; a.rpp = g.rpp; a.nroles = g.nroles;
    for (int y = 0; y <= g.nroles; ++y) a.first[y] = g.first[y];
    const bool staged = ppo_staged(a.N, a.H);
    const size_t lds = ((size_t)a.rpp * a.H + a.H + a.DN + 10) * sizeof(double) + (staged ? (size_t)a.N * a.H * sizeof(float) : 0);
    const dim3 grid(g.first[g.nroles]);
    if (a.H <= 64) {
        if (staged) hipLaunchKernelGGL((ppo_grad_kernel<1, true>), grid, dim3(1024), lds, stream, a);
        else hipLaunchKernelGGL((ppo_grad_kernel<1, false>), grid, dim3(1024), lds, stream, a);
    } else {
        if (staged) hipLaunchKernelGGL((ppo_grad_kernel<2, true>), grid, dim3(1024), lds, stream, a);
        else hipLaunchKernelGGL((ppo_grad_kernel<2, false>), grid, dim3(1024), lds, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int total = (a.A + 1) * a.slab;
    hipLaunchKernelGGL(ppo_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, a.partial, g, total, grads, adam, a.A, a.N, a.H,
                       a.DN, a.slab);
    return hipGetLastError();
}

}  // namespace sy
