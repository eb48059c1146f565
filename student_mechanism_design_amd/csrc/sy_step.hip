// sy_step.hip — step_kernel (sy_env_step / sy_env_step_record) and reset_kernel (sy_env_reset / sy_env_reset_to).
#include "sy_device.hpp"

namespace sy {

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

// ---- launchers
template <int NR>
static hipError_t launch_step_nr(const EngineParams& p, const int32_t* actions, const sy_rollout_buffers& rec, int blocks, int wpb,
                                 size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL((step_kernel<NR>), dim3(blocks), dim3(wpb * 64), lds, stream, p, actions, rec);
    return hipGetLastError();
}

hipError_t launch_step(const EngineParams& p, const int32_t* actions, const sy_rollout_buffers& rec, int blocks, int wpb, size_t lds,
                       hipStream_t stream) {
    const int nr = (p.N + 63) / 64;
    if (nr <= 1) return launch_step_nr<1>(p, actions, rec, blocks, wpb, lds, stream);
    if (nr <= 2) return launch_step_nr<2>(p, actions, rec, blocks, wpb, lds, stream);
    if (nr <= 4) return launch_step_nr<4>(p, actions, rec, blocks, wpb, lds, stream);
    if (nr <= 8) return launch_step_nr<8>(p, actions, rec, blocks, wpb, lds, stream);
    return launch_step_nr<16>(p, actions, rec, blocks, wpb, lds, stream);
}

hipError_t launch_reset(const EngineParams& p, const uint8_t* env_sel, const int32_t* starts, int zero_count, int blocks,
                        int wpb, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL(reset_kernel, dim3(blocks), dim3(wpb * 64), lds, stream, p, env_sel, starts, zero_count);
    return hipGetLastError();
}

}  // namespace sy
