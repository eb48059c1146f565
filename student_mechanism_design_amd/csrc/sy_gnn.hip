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
// The propagation tables (sources + coefficients per target node) come from graph.py::gcn_tables — by default the
// DIRECTED edge list exactly as the reference hands it to the model (training/utils.py:170).
// torch_geometric is not importable offline: parity is pinned to an independent float64 restatement of the published
// formula (oracle/gnn_oracle.py), "unpinned vs the library".
#include "sy_device.hpp"

namespace sy {

// packed parameters of one model, float32, FP = padded feature count (rows / columns past F are zero):
//   per conv layer: Wa [FP][FP] = W - W^T - gamma I (row-major: out feature, in feature), Th [FP][FP] = GCNConv weight
//   (out, in), b [FP];  then w_out [FP], b_out, epsilon
template <int FP>
struct GnnLayout {
    static constexpr int kLayer = 2 * FP * FP + FP;
    static constexpr int kOutW = 2 * kLayer, kOutB = kOutW + FP, kEps = kOutB + 1, kTotal = kEps + 1;
};

template <int NR, int FP>
__device__ __forceinline__ void gnn_conv(float (&x)[NR][FP], const float* __restrict__ prm, float eps, uint32_t scr,
                                         const int16_t* __restrict__ nbr, const float* __restrict__ coef,
                                         const float* __restrict__ selfc, int lane, int N) {
    const float* Wa = prm;
    const float* Th = prm + FP * FP;
    const float* bb = prm + 2 * FP * FP;
    float t[NR][FP];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = lane + 64 * r;
#pragma unroll
        for (int o = 0; o < FP; ++o) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < FP; ++i) acc = fmaf(Th[o * FP + i], x[r][i], acc);     // Theta x  (GCNConv's linear map)
            t[r][o] = acc;
            if (n < N) *lds_at<float>(scr + (uint32_t)(n * FP + o) * 4u) = acc;
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = lane + 64 * r;
        const int nn = n < N ? n : N - 1;
        float h[FP];
        const float sc = selfc[nn];
#pragma unroll
        for (int o = 0; o < FP; ++o) {
            float acc = bb[o];
#pragma unroll
            for (int i = 0; i < FP; ++i) acc = fmaf(Wa[o * FP + i], x[r][i], acc);     // (W - W^T - gamma I) x + b
            h[o] = fmaf(sc, t[r][o], acc);                                              // the self loop of A^
        }
        const int16_t* nb = nbr + (size_t)nn * 16;
        const float* cf = coef + (size_t)nn * 16;
        for (int k = 0; k < 16; ++k) {
            const int u = nb[k];
            if (u < 0) break;                                                           // rows are filled left to right
            const float c = cf[k];
#pragma unroll
            for (int o = 0; o < FP; ++o) h[o] = fmaf(c, *lds_at<float>(scr + (uint32_t)(u * FP + o) * 4u), h[o]);
        }
#pragma unroll
        for (int o = 0; o < FP; ++o) {
            const float v = x[r][o] + eps * tanhf(h[o]);
            x[r][o] = v > 0.0f ? v : 0.0f;                                              // the model's relu after each conv
        }
    }
    wave_lds_fence();
}

template <int NR, int FP>
__global__ __launch_bounds__(256) void gnn_q_act_kernel(const int32_t* __restrict__ pos, const float* __restrict__ belief,
                                                        long long belief_stride, const uint8_t* __restrict__ mask,
                                                        long long mask_row_stride, const int16_t* __restrict__ nbr_all,
                                                        const float* __restrict__ coef_all, const float* __restrict__ self_all,
                                                        const int32_t* __restrict__ env_graph, const float* __restrict__ prm_mrx,
                                                        const float* __restrict__ prm_pol, int B, int A, int N, int F,
                                                        float explore, uint32_t seed_lo, uint32_t seed_hi, uint64_t offset_imm,
                                                        const uint64_t* __restrict__ offset_dev, int32_t* __restrict__ action,
                                                        float* __restrict__ q_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int b = blockIdx.x * wpb + wid;
    if (b >= B) return;
    const uint32_t scr = lds_off(smem) + (uint32_t)wid * (uint32_t)(N * FP) * 4u;
    const int g = env_graph ? env_graph[b] : 0;
    const int16_t* nbr = nbr_all + (size_t)g * N * 16;
    const float* coef = coef_all + (size_t)g * N * 16;
    const float* selfc = self_all + (size_t)g * N;
    // node features (training/utils.py:176-200): column a = one-hot node of agent a; the optional last column = belief
    int pa[SY_MAX_AGENTS];
#pragma unroll
    for (int a = 0; a < SY_MAX_AGENTS; ++a) pa[a] = a < A ? pos[(size_t)b * A + a] : -1;
    float x0[NR][FP];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = lane + 64 * r;
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            float v = 0.0f;
            if (f < SY_MAX_AGENTS && f < A) v = (n == pa[f < SY_MAX_AGENTS ? f : 0]) ? 1.0f : 0.0f;
            else if (f == A && belief != nullptr && f < F && n < N) v = belief[(size_t)b * belief_stride + n];
            x0[r][f] = v;
        }
    }
    const uint64_t offset = offset_imm + (offset_dev ? *offset_dev : 0ull);
    typedef GnnLayout<FP> LY;
    for (int m = 0; m < 2; ++m) {                         // model 0: MrX's, model 1: the police's
        const float* prm = m == 0 ? prm_mrx : prm_pol;
        if (m == 1 && A < 2) break;
        float x[NR][FP];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int f = 0; f < FP; ++f) x[r][f] = x0[r][f];
        const float eps = prm[LY::kEps];
        gnn_conv<NR, FP>(x, prm, eps, scr, nbr, coef, selfc, lane, N);
        gnn_conv<NR, FP>(x, prm + LY::kLayer, eps, scr, nbr, coef, selfc, lane, N);
        float q[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float acc = prm[LY::kOutB];
#pragma unroll
            for (int f = 0; f < FP; ++f) acc = fmaf(prm[LY::kOutW + f], x[r][f], acc);
            q[r] = acc;
            const int n = lane + 64 * r;
            if (q_out && n < N) q_out[((size_t)b * 2 + m) * N + n] = acc;
        }
        // epsilon-greedy masked arg-max per agent (gnn_agent.py:62-74): np.argmax -> the FIRST maximum in node order
        const int a_lo = m == 0 ? 0 : 1, a_hi = m == 0 ? 1 : A;
        for (int a = a_lo; a < a_hi; ++a) {
            const uint8_t* mr = mask + ((size_t)b * A + a) * mask_row_stride;
            float best = -3.0e38f;
            int bidx = 0x7fffffff, cnt = 0;
            bool ok[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int n = lane + 64 * r;
                ok[r] = n < N && mr[n] != 0;
                cnt += ok[r] ? 1 : 0;
                if (ok[r] && (q[r] > best)) { best = q[r]; bidx = n; }
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
}

template <int FP>
static hipError_t launch_gnn_fp(const int32_t* pos, const float* belief, long long belief_stride, const uint8_t* mask,
                                long long mask_row_stride, const int16_t* nbr, const float* coef, const float* selfc,
                                const int32_t* env_graph, const float* prm_mrx, const float* prm_pol, int B, int A, int N, int F,
                                float explore, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int32_t* action,
                                float* q_out, hipStream_t stream) {
    const int wpb = 4, blocks = (B + wpb - 1) / wpb;
    const size_t lds = (size_t)wpb * N * FP * sizeof(float);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    const int nr = (N + 63) / 64;
#define SY_LAUNCH_GNN(NR_) hipLaunchKernelGGL((gnn_q_act_kernel<NR_, FP>), dim3(blocks), dim3(wpb * 64), lds, stream, pos, belief,  \
                                              belief_stride, mask, mask_row_stride, nbr, coef, selfc, env_graph, prm_mrx, prm_pol, \
                                              B, A, N, F, explore, lo, hi, offset, offset_dev, action, q_out)
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
                            long long mask_row_stride, const int16_t* nbr, const float* coef, const float* selfc,
                            const int32_t* env_graph, const float* prm_mrx, const float* prm_pol, int B, int A, int N, int F,
                            float explore, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int32_t* action,
                            float* q_out, hipStream_t stream) {
    if (F <= 6)
        return launch_gnn_fp<6>(pos, belief, belief_stride, mask, mask_row_stride, nbr, coef, selfc, env_graph, prm_mrx, prm_pol, B, A,
                                N, F, explore, seed, offset, offset_dev, action, q_out, stream);
    return launch_gnn_fp<9>(pos, belief, belief_stride, mask, mask_row_stride, nbr, coef, selfc, env_graph, prm_mrx, prm_pol, B, A, N,
                            F, explore, seed, offset, offset_dev, action, q_out, stream);
}

}  // namespace sy
