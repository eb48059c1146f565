// sy_gnn.hip — the GNN Q-policy of the reference (src/agent/gnn_agent.py:230-257: two AntiSymmetricConv layers + a
// Linear head giving one Q value per node) for every env of a batch in ONE launch, with the epsilon-greedy masked
// arg-max of GNNAgent.select_action (gnn_agent.py:45-82) for every agent.
//
// This is gather work on a sparse board, not a matmul: F = number of agents (+ 1 with the belief column) is 3..9, rows
// have at most 16 sources.  One wave per env; lane L owns the nodes L, L + 64, ...: their feature vectors live in
// registers, the transformed features Phi x go through a per-wave LDS scratch for the neighbour gathers (the belief
// filter's pattern).  Both models of the trainer (MrX's and the police's, gnn_trainer.py:148-178) are evaluated on the
// same node features; every police agent takes the masked arg-max of the police model's Q map.
//
//   AntiSymmetricConv (Gravina et al., ICLR 2023; torch_geometric defaults the reference uses: phi = GCNConv(F, F,
//   bias = False), num_iters = 1, epsilon = gamma = 0.1, act = tanh):
//       x' = x + epsilon * tanh((W - W^T - gamma I) x + A^ (Theta x) + b),    A^ = D^-1/2 (A + I) D^-1/2
//   model: x1 = relu(conv1(x)); x2 = relu(conv2(x1)); q = w_out . x2 + b_out
// The propagation tables (sources + coefficients per target node, K = the widest row of the pool) come from
// graph.py::gcn_tables — by default the DIRECTED edge list exactly as the reference hands it to the model
// (training/utils.py:170); a lane keeps its nodes' rows in registers for both layers of both models.
// torch_geometric is not importable offline: parity is pinned to an independent float64 restatement of the published
// formula (oracle/gnn_oracle.py), "unpinned vs the library".
#include "sy_device.hpp"

namespace sy {

typedef float v2f __attribute__((ext_vector_type(2)));

// Packed parameters, float32, BOTH models interleaved (element p of MrX's model at [p][0], of the police model at [p][1]:
// the two models run in lockstep as the two halves of packed-f32 operations); FP = padded feature count (rows / columns
// past F are zero):  per conv layer: Wa [FP][FP] = W - W^T - gamma I (row-major: out feature, in feature), Th [FP][FP] =
// GCNConv weight (out, in), b [FP];  then w_out [FP], b_out, epsilon.
template <int FP>
struct GnnLayout {
    static constexpr int kLayer = 2 * FP * FP + FP;
    static constexpr int kOutW = 2 * kLayer, kOutB = kOutW + FP, kEps = kOutB + 1, kTotal = kEps + 1;
    static constexpr int kRow = (2 * FP + 3) & ~3;      // floats per node in the LDS scratch: both models' Theta x, 16-byte pieces
};

__device__ __forceinline__ v2f fast_tanh2(v2f h) {      // tanh(x) = 1 - 2 / (exp(2x) + 1); |error| < 3e-7 (the test's bound is 1e-5)
    v2f r;
    r.x = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * h.x) + 1.0f);
    r.y = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * h.y) + 1.0f);
    return r;
}

// one AntiSymmetricConv + relu for both models; tab[r][k] = {source node, coefficient bits} of my node r (registers)
template <int NR, int FP, int KW>
__device__ __forceinline__ void gnn_conv(v2f (&x)[NR][FP], const v2f* __restrict__ prm, v2f eps, uint32_t scr,
                                         const uint2 (&tab)[NR][KW], const float (&selfc)[NR], int lane, int N) {
    typedef GnnLayout<FP> LY;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v2f* Wa = prm;
    const v2f* Th = prm + FP * FP;
    const v2f* bb = prm + 2 * FP * FP;
    v2f t[NR][FP];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = lane + 64 * r;
        float row[LY::kRow];
#pragma unroll
        for (int q = 0; q < LY::kRow; ++q) row[q] = 0.0f;
#pragma unroll
        for (int o = 0; o < FP; ++o) {
            v2f acc = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < FP; ++i) acc = __builtin_elementwise_fma(Th[o * FP + i], x[r][i], acc);     // Theta x  (GCNConv's linear map)
            t[r][o] = acc;
            row[2 * o] = acc.x;
            row[2 * o + 1] = acc.y;
        }
        if (n < N) {
#pragma unroll
            for (int q = 0; q < LY::kRow; q += 4)
                *lds_at<v4f>(scr + (uint32_t)(n * LY::kRow + q) * 4u) = (v4f){row[q], row[q + 1], row[q + 2], row[q + 3]};
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        v2f h[FP];
        const v2f sc = {selfc[r], selfc[r]};
#pragma unroll
        for (int o = 0; o < FP; ++o) {
            v2f acc = bb[o];
#pragma unroll
            for (int i = 0; i < FP; ++i) acc = __builtin_elementwise_fma(Wa[o * FP + i], x[r][i], acc);     // (W - W^T - gamma I) x + b
            h[o] = __builtin_elementwise_fma(sc, t[r][o], acc);                                              // the self loop of A^
        }
#pragma unroll
        for (int k = 0; k < KW; ++k) {            // padding entries: source 0 with coefficient 0
            const uint32_t base = scr + tab[r][k].x * (uint32_t)(LY::kRow * 4);
            const float cf = __uint_as_float(tab[r][k].y);
            const v2f c = {cf, cf};
            float g[LY::kRow];
#pragma unroll
            for (int q = 0; q < LY::kRow; q += 4) {
                const v4f v = *lds_at<v4f>(base + (uint32_t)q * 4u);
                g[q] = v.x; g[q + 1] = v.y; g[q + 2] = v.z; g[q + 3] = v.w;
            }
#pragma unroll
            for (int o = 0; o < FP; ++o) h[o] = __builtin_elementwise_fma(c, (v2f){g[2 * o], g[2 * o + 1]}, h[o]);
        }
#pragma unroll
        for (int o = 0; o < FP; ++o) {
            const v2f v = x[r][o] + eps * fast_tanh2(h[o]);
            x[r][o] = (v2f){v.x > 0.0f ? v.x : 0.0f, v.y > 0.0f ? v.y : 0.0f};                              // the model's relu after each conv
        }
    }
    wave_lds_fence();
}

template <int NR, int FP, int KW>
__global__ __launch_bounds__(256) void gnn_q_act_kernel(const int32_t* __restrict__ pos, const float* __restrict__ belief,
                                                        long long belief_stride, const uint8_t* __restrict__ mask,
                                                        long long mask_row_stride, const uint2* __restrict__ tab_all, int K,
                                                        const float* __restrict__ self_all, const int32_t* __restrict__ env_graph,
                                                        const v2f* __restrict__ prm, int B, int A, int N, int F, float explore,
                                                        uint32_t seed_lo, uint32_t seed_hi, uint64_t offset_imm,
                                                        const uint64_t* __restrict__ offset_dev, int32_t* __restrict__ action,
                                                        float* __restrict__ q_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef GnnLayout<FP> LY;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int b = blockIdx.x * wpb + wid;
    if (b >= B) return;
    const uint32_t scr = lds_off(smem) + (uint32_t)wid * (uint32_t)(N * LY::kRow) * 4u;
    const int g = env_graph ? env_graph[b] : 0;
    // my nodes' table rows, once, into registers (both layers of both models use them): K entries of 8 bytes per node
    uint2 tab[NR][KW];
    float selfc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = lane + 64 * r;
        const int nn = n < N ? n : N - 1;
        const uint2* row = tab_all + ((size_t)g * N + nn) * K;
#pragma unroll
        for (int k = 0; k < KW; ++k) tab[r][k] = k < K ? row[k] : make_uint2(0u, 0u);
        selfc[r] = self_all[(size_t)g * N + nn];
    }
    // node features (training/utils.py:176-200): column a = one-hot node of agent a; the optional last column = belief
    int pa[SY_MAX_AGENTS];
#pragma unroll
    for (int a = 0; a < SY_MAX_AGENTS; ++a) pa[a] = a < A ? pos[(size_t)b * A + a] : -1;
    v2f x[NR][FP];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = lane + 64 * r;
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            float v = 0.0f;
            if (f < SY_MAX_AGENTS && f < A) v = (n == pa[f < SY_MAX_AGENTS ? f : 0]) ? 1.0f : 0.0f;
            else if (f == A && belief != nullptr && f < F && n < N) v = belief[(size_t)b * belief_stride + n];
            x[r][f] = (v2f){v, v};
        }
    }
    const uint64_t offset = offset_imm + (offset_dev ? *offset_dev : 0ull);
    const v2f eps = prm[LY::kEps];
    gnn_conv<NR, FP, KW>(x, prm, eps, scr, tab, selfc, lane, N);
    gnn_conv<NR, FP, KW>(x, prm + LY::kLayer, eps, scr, tab, selfc, lane, N);
    v2f q[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        v2f acc = prm[LY::kOutB];
#pragma unroll
        for (int f = 0; f < FP; ++f) acc = __builtin_elementwise_fma(prm[LY::kOutW + f], x[r][f], acc);
        q[r] = acc;
        const int n = lane + 64 * r;
        if (q_out && n < N) {
            q_out[((size_t)b * 2 + 0) * N + n] = acc.x;
            q_out[((size_t)b * 2 + 1) * N + n] = acc.y;
        }
    }
    // epsilon-greedy masked arg-max per agent (gnn_agent.py:62-74): np.argmax -> the FIRST maximum in node order;
    // MrX reads its own model's Q map, every police agent the police model's
    for (int a = 0; a < A; ++a) {
        const uint8_t* mr = mask + ((size_t)b * A + a) * mask_row_stride;
        float best = -3.0e38f;
        int bidx = 0x7fffffff, cnt = 0;
        bool ok[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int n = lane + 64 * r;
            ok[r] = n < N && mr[n] != 0;
            cnt += ok[r] ? 1 : 0;
            const float qv = a == 0 ? q[r].x : q[r].y;
            if (ok[r] && (qv > best)) { best = qv; bidx = n; }
        }
#pragma unroll
        for (int o2 = 32; o2 >= 1; o2 >>= 1) {
            const float ob = __shfl_xor(best, o2, kWave);
            const int oi = __shfl_xor(bidx, o2, kWave);
            cnt += __shfl_xor(cnt, o2, kWave);
            const bool take = (ob > best) || (ob == best && oi < bidx);
            best = take ? ob : best;
            bidx = take ? oi : bidx;
        }
        int act = cnt > 0 ? bidx : -1;                                              // no valid action -> None (-1)
        if (explore > 0.0f && cnt > 0) {
            uint32_t w[4];
            philox4((uint64_t)b * (uint64_t)A + (uint64_t)a, (uint32_t)offset, 3u, (uint32_t)(offset >> 32) & 0xffu, seed_lo, seed_hi, w);
            const float u = (float)(w[0] >> 8) * (1.0f / 16777216.0f);
            if (u <= explore) {                                                     // np.random.choice(valid_actions)
                const int pick = (int)__umulhi(w[1], (uint32_t)cnt);
                int seen = 0, chosen = -1;
#pragma unroll
                for (int r = 0; r < NR; ++r) {                                      // rank of my valid nodes in node order
                    const uint64_t bm = __ballot(ok[r]);
                    const int before = seen + __popcll(bm & ((1ull << lane) - 1ull));
                    if (ok[r] && before == pick) chosen = lane + 64 * r;
                    seen += __popcll(bm);
                }
#pragma unroll
                for (int o2 = 32; o2 >= 1; o2 >>= 1) {
                    const int oc = __shfl_xor(chosen, o2, kWave);
                    chosen = oc > chosen ? oc : chosen;
                }
                act = chosen;
            }
        }
        if (lane == 0) action[(size_t)b * A + a] = act;
    }
}

template <int FP, int KW>
static hipError_t launch_gnn_fk(const int32_t* pos, const float* belief, long long belief_stride, const uint8_t* mask,
                                long long mask_row_stride, const uint32_t* tab, int K, const float* selfc, const int32_t* env_graph,
                                const float* models, int B, int A, int N, int F, float explore, uint64_t seed, uint64_t offset,
                                const uint64_t* offset_dev, int32_t* action, float* q_out, hipStream_t stream) {
    const int wpb = 4, blocks = (B + wpb - 1) / wpb;
    const size_t lds = (size_t)wpb * N * GnnLayout<FP>::kRow * sizeof(float);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    const int nr = (N + 63) / 64;
#define SY_LAUNCH_GNN(NR_) hipLaunchKernelGGL((gnn_q_act_kernel<NR_, FP, KW>), dim3(blocks), dim3(wpb * 64), lds, stream, pos, belief,   \
                                              belief_stride, mask, mask_row_stride, reinterpret_cast<const uint2*>(tab), K, selfc,       \
                                              env_graph, reinterpret_cast<const v2f*>(models), B, A, N, F, explore, lo, hi, offset,      \
                                              offset_dev, action, q_out)
    if (nr <= 1) SY_LAUNCH_GNN(1);
    else if (nr <= 2) SY_LAUNCH_GNN(2);
    else if (nr <= 4) SY_LAUNCH_GNN(4);
    else return hipErrorInvalidValue;
#undef SY_LAUNCH_GNN
    return hipGetLastError();
}

int gnn_padded_features(int F) { return F <= 6 ? 6 : 9; }
int gnn_param_floats(int F) { return F <= 6 ? GnnLayout<6>::kTotal : GnnLayout<9>::kTotal; }

hipError_t launch_gnn_q_act(const int32_t* pos, const float* belief, long long belief_stride, const uint8_t* mask,
                            long long mask_row_stride, const uint32_t* tab, int K, const float* selfc, const int32_t* env_graph,
                            const float* models, int B, int A, int N, int F, float explore, uint64_t seed, uint64_t offset,
                            const uint64_t* offset_dev, int32_t* action, float* q_out, hipStream_t stream) {
#define SY_GNN_ARGS pos, belief, belief_stride, mask, mask_row_stride, tab, K, selfc, env_graph, models, B, A, N, F, explore, seed, offset, \
                    offset_dev, action, q_out, stream
    if (F <= 6) {
        if (K <= 4) return launch_gnn_fk<6, 4>(SY_GNN_ARGS);
        if (K <= 8) return launch_gnn_fk<6, 8>(SY_GNN_ARGS);
        return launch_gnn_fk<6, 16>(SY_GNN_ARGS);
    }
    if (K <= 8) return launch_gnn_fk<9, 8>(SY_GNN_ARGS);
    return launch_gnn_fk<9, 16>(SY_GNN_ARGS);
#undef SY_GNN_ARGS
}

}  // namespace sy
