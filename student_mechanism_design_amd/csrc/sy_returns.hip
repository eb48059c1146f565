// sy_returns.hip — returns / advantages / GAE of a whole [T][B][A] record in one launch.
#include "sy_device.hpp"

namespace sy {

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


// ---- launchers
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
