// sy_dispatch.hip — which kernel instance a fused rollout runs on, and its launch.  Host code only: the instances
// themselves live in sy_rollout3_*.hip / sy_rollout2_*.hip / sy_rollout1_*.hip (one translation unit per group, built in
// parallel); this file is the single owner of the selection rules.
#include <cstdio>

#include "sy_kernels.h"

namespace sy {

int rollout_policy_slice(int family, int A, int hidden, int entry_cap) {
    // rollout3: A hidden vectors + 8 slots of 16 B + the episode's share of the pair's compacted entry list (node | agent
    // | half in 2 bytes, the logit in 4: 6 bytes per affordable ELL entry, at most `entry_cap` per episode);
    // rollout2: the fixed round-1 slice (8 vectors of 64 floats + slots)
    return family == 3 ? A * hidden * 4 + 128 + ((6 * entry_cap + 15) & ~15) + 16 : SY_POLICY_SLICE;
}

RolloutPlan plan_rollout(const EngineParams& p, bool record, int wpb, size_t lds_base, int policy_hidden) {
    RolloutPlan pl;
    const int nr = (p.N + 63) / 64;
    pl.nr = nr <= 1 ? 1 : (nr <= 2 ? 2 : (nr <= 4 ? 4 : (nr <= 8 ? 8 : 16)));
    pl.pt = (p.P == 2 || p.P == 4 || p.P == 5 || p.P == 6) ? p.P : 0;   // the BASELINE.json police counts get unrolled instances
    pl.hs = 0;
    pl.rec = record;
    const bool pol = policy_hidden > 0 || p.pw2 != nullptr;
    const int H = policy_hidden > 0 ? policy_hidden : p.pH;
    pl.pol = pol;
    pl.pslice = 0;
    pl.pcap = 0;
    const bool paired = (wpb & 1) == 0;              // paired move waves need an even number of episodes per block
    const int per_pass = 64 / p.scan_w;              // agents one pass of the paired scan covers
    const bool has_belief = p.st.belief != nullptr;
#ifdef SY_NO_PIPELINE
    const bool pipelined = false;
#else
    // the move / helper pipeline: boards of up to 256 nodes (node-major belief lanes) whose agents fit one or two
    // scan passes; the ring's meta word holds the timestep in 20 bits
    const bool pipelined = paired && pl.nr <= 4 && p.A <= 2 * per_pass && p.max_t < (1 << 20) - 2;
#endif
#ifdef SY_POL_ROLLOUT2
    const bool pol_pipeline = false;
#else
    const bool pol_pipeline = pipelined;
#endif
    // half-wave scan: columns per lane that cover the pool's widest row (0 = no instance, the paired scan)
    const int hs_gw = half_scan_gw(p.P);
    const int hs_need = (p.max_deg + hs_gw - 1) / hs_gw;
    const int hs_cols = (p.A <= 7 && hs_need * hs_gw <= 32) ? hs_need : 0;
    const int threads3 = 64 * wpb;
    const int threads12 = paired ? 64 * (wpb / 2 + (has_belief ? wpb / 2 : 0))
                                 : 64 * (wpb + (has_belief ? (wpb + 1) / 2 : 0));   // move waves + belief waves
    // learned policy in the move wave: the half-wave scan (no passes: any row width) for the police counts with a
    // compile-time instance — the smallest column count that covers the pool's widest row —, the single-pass paired
    // scan for the other counts
    int pol_hs = 0;
    if (pol && pol_pipeline) {
        const int P = p.P;
        const int need = hs_need;
        if (pl.nr == 4) {
            if (P == 2 && need <= 2) pol_hs = 2;
            else if (P == 4 && need <= 3) pol_hs = need <= 2 ? 2 : 3;
            else if (P == 5 && need <= 4) pol_hs = need <= 2 ? 2 : 4;
            else if (P == 6 && need <= 4) pol_hs = need <= 3 ? 3 : 4;
            else if (P == 7 && need <= 4) pol_hs = 4;
        } else {
            if (P == 2 && need <= 2) pol_hs = 2;
            else if (P == 4 && need <= 3) pol_hs = 3;
            else if ((P == 5 || P == 6 || P == 7) && need <= 4) pol_hs = 4;
        }
    }
    if (pol && pol_pipeline && pol_hs > 0) {
        pl.family = 3;
        pl.rec = true;
        pl.pt = p.P;
        pl.hs = pol_hs;
    } else if (pol && pol_pipeline && p.A <= per_pass) {
        pl.family = 3;
        pl.rec = true;
        pl.pt = 0;
    } else if (pol && paired && p.A <= per_pass) {        // round 1's kernel: single-pass boards, hidden <= 64
        pl.family = 2;
        pl.rec = true;
        pl.pt = pl.pt == 4 ? 4 : 0;
    } else if (pipelined) {
        pl.family = 3;
        pl.pol = false;
        if (pl.pt >= 1 && pl.pt <= 4) {
            if (hs_cols > 0 && hs_cols <= 2) pl.hs = 2;
        } else if (pl.nr == 4 && (pl.pt == 5 || pl.pt == 6)) {
            const int c0 = pl.pt == 5 ? 2 : 3;
            if (hs_cols > 0 && hs_cols <= c0) pl.hs = c0;
            else if (hs_cols == c0 + 1) pl.hs = c0 + 1;
        }
    } else {
        pl.family = paired ? 2 : 1;
        pl.pol = false;      // no policy instance (odd block sizes; boards of more than 256 nodes whose agents need two scan
                             // passes): sy_env_set_policy / sy_env_rollout refuse these configurations
    }
    if (pl.pol) {
        // affordable entries an episode can have: every agent's row, as wide as the scan covers
        const int cols = pl.hs > 0 ? pl.hs * half_scan_gw(p.P) : SY_ELL_WIDTH;
        pl.pcap = p.A * (cols < SY_ELL_WIDTH ? cols : SY_ELL_WIDTH);
        pl.pslice = rollout_policy_slice(pl.family, p.A, H, pl.pcap);
    }
    pl.threads = pl.family == 3 ? threads3 : threads12;
    pl.lds = lds_base + (size_t)wpb * pl.pslice;
    return pl;
}

void rollout_plan_name(const RolloutPlan& pl, char* buf, size_t n) {
    const char* r = pl.rec ? "true" : "false";
    const char* q = pl.pol ? "true" : "false";
    if (pl.family == 3) std::snprintf(buf, n, "sy::rollout3_kernel<%d,%s,%d,%s,%d>", pl.nr, r, pl.pt, q, pl.hs);
    else if (pl.family == 2) std::snprintf(buf, n, "sy::rollout2_kernel<%d,%s,%d,%s>", pl.nr, r, pl.pt, q);
    else std::snprintf(buf, n, "sy::rollout_kernel<%d,%s,%d>", pl.nr, r, pl.pt);
}

hipError_t launch_rollout(const EngineParams& p, int T, const sy_rollout_buffers& out, int blocks, int wpb, size_t lds,
                          hipStream_t stream) {
    const RolloutPlan pl = plan_rollout(p, out.record != nullptr, wpb, lds);
    bool ok = false;
    if (pl.family == 3) {
        if (pl.pol) ok = pl.nr == 4 ? launch_r3_p(pl, p, T, out, blocks, stream) : launch_r3_q(pl, p, T, out, blocks, stream);
        else if (pl.nr == 4) ok = pl.rec ? launch_r3_a(pl, p, T, out, blocks, stream) : launch_r3_b(pl, p, T, out, blocks, stream);
        else ok = pl.nr == 1 ? launch_r3_c(pl, p, T, out, blocks, stream) : launch_r3_d(pl, p, T, out, blocks, stream);
    } else if (pl.family == 2) {
        ok = pl.nr <= 2 ? launch_r2_a(pl, p, T, out, blocks, stream)
                        : (pl.nr == 4 ? launch_r2_b(pl, p, T, out, blocks, stream) : launch_r2_c(pl, p, T, out, blocks, stream));
    } else {
        ok = pl.nr <= 4 ? launch_r1_a(pl, p, T, out, blocks, stream) : launch_r1_b(pl, p, T, out, blocks, stream);
    }
    if (!ok) return hipErrorInvalidDeviceFunction;     // a plan without an instance: a bug in plan_rollout, never silent
    return hipGetLastError();
}

}  // namespace sy
