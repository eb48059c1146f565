// sy_aux.hip — standalone ops: compute_action_mask on dense matrices, the batched belief update, APSP, the board sampler.
#include "sy_device.hpp"

namespace sy {

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

// ---- launchers
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

}  // namespace sy
