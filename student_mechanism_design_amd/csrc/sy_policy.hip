// sy_policy.hip — policy-side kernels: masked categorical sampling and the MAPPO networks (f32 MFMA second layer).
#include "sy_device.hpp"

namespace sy {

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
    constexpr int kHS = 129;                                     // hidden row stride (floats): conflict-free A-operand reads; H <= 128
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
        float part = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {                           // hidden units lane, lane + 64 (H <= 128)
            const int k = lane + 64 * u;
            if (k < H) {
                float h = w.cb1[k];
                h += w.c1t[(size_t)prow[0] * H + k];
                for (int kk = 0; kk < P; ++kk)
                    for (int j = 0; j < P; ++j) h += w.c1t[((size_t)N * (1 + kk) + prow[1 + j]) * H + k];
                h = h > 0.0f ? h : 0.0f;
                part += h * w.c2[k];
            }
        }
        const float v = wave_sum(part) + w.cb2[0];
        if (lane == 0) value[b] = v;
        return;
    }
    // ---- phase 1: actor a's first layer by row lookups (lane k holds hidden units k and k + 64 of this wave's env)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int k = lane + 64 * u;
        float h = k < H ? w.b1[(size_t)a * H + k] : 0.0f;
        if (k < H && live) {
            const float* w1a = w.w1t + (size_t)a * N * H;
            if (a == 0) h += w1a[(size_t)prow[0] * H + k];
            else
                for (int j = 0; j < P; ++j) h += w1a[(size_t)prow[1 + j] * H + k];
        }
        hs[wid * kHS + k] = (k < H && h > 0.0f) ? h : 0.0f;
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

// ---- launchers
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
    const size_t lds = (size_t)(16 * 129 + 16 * (nrp * 64 + 1)) * sizeof(float);   // hidden rows + logits rows of 16 envs
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

}  // namespace sy
