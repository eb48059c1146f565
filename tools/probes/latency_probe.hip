// Dependent-chain latencies of the instruction kinds the rollout's move wave is made of, on gfx950.
// Each pattern is a chain of N dependent operations timed with s_memtime by one wave, (a) alone on its CU and
// (b) with 16 waves per CU all running the same chain (the rollout's occupancy: 4 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/latency_probe.hip -o tools/_diag/latency_probe && tools/_diag/latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

enum { P_VALU = 0, P_F64, P_SALU, P_CMP_SALU_CND, P_READLANE, P_LDS, P_BPERM, P_DPP, P_MULLO, P_LDSATOM, P_L2, P_READFIRST, P_BALLOT_SCC, P_MIX, P_LDS128, P_ST1, P_ST4, P_DSW, P_DSW128, P_ADD8, P_COUNT };
static const char* kNames[] = {"v_add_u32 chain", "v_fma_f64 chain", "s_add_u32 chain", "v_cmp -> s_and -> v_cndmask", "v_readlane -> v_add(sgpr)",
                               "ds_read_b32 pointer chase", "ds_bpermute chain", "v_mov_dpp row_shr chain", "v_mul_lo_u32 chain", "ds_add_rtn_u32 chain",
                               "global_load pointer chase (L2)", "v_readfirstlane -> s_add -> v_mov", "v_cmp -> s_cmp(scc) -> s_cselect -> v_add",
                               "v_add,s_add independent interleave", "ds_read_b128 then use",
                               "8 v_add + global_store_dword", "8 v_add + global_store_dwordx4", "8 v_add + ds_write_b32", "8 v_add + ds_write_b128",
                               "8 v_add (reference for the four above)"};

__global__ __launch_bounds__(1024) void probe(int pattern, int iters, unsigned* chase, unsigned long long* out) {
    __shared__ unsigned lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned)(((i * 4 + 256) & 16383));   // byte address of the next word
    __syncthreads();
    unsigned v = lane, w = 1;
    unsigned s = blockIdx.x;
    double d = 1.0 + lane;
    unsigned addr = (threadIdx.x * 4) & 16383;
    unsigned long long gp = (unsigned long long)chase;
    if (pattern == P_L2) v = lane * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        switch (pattern) {
        case P_VALU: asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(v) : "v"(w)); break;
        case P_F64: asm volatile(REP64("v_fma_f64 %0, %0, %0, %0\n") : "+v"(d)); break;
        case P_SALU: asm volatile(REP64("s_add_u32 %0, %0, 3\n") : "+s"(s) : : "scc"); break;
        case P_CMP_SALU_CND:
            asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %1, %0, vcc\n") : "+v"(v) : "v"(w) : "vcc"); break;
        case P_READLANE:
            asm volatile(REP64("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0\n") : "+v"(v) : : "s20"); break;
        case P_LDS: asm volatile(REP64("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(addr) : : "memory"); break;
        case P_BPERM: asm volatile(REP64("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(v) : "v"(addr) : "memory"); break;
        case P_DPP: asm volatile(REP64("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(v)); break;
        case P_MULLO: asm volatile(REP64("v_mul_lo_u32 %0, %0, %1\n") : "+v"(v) : "v"(w)); break;
        case P_LDSATOM: asm volatile(REP64("ds_add_rtn_u32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(v) : "v"(addr) : "memory"); break;
        case P_L2: {
            unsigned lo = (unsigned)gp, hi = (unsigned)(gp >> 32);
            asm volatile(REP64("global_load_dword %0, %2, %1\n s_waitcnt vmcnt(0)\n") : "+v"(v) : "s"(gp), "v"(v) : "memory");
            (void)lo; (void)hi;
            break; }
        case P_READFIRST:
            asm volatile(REP64("v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 1\n v_mov_b32 %0, s20\n") : "+v"(v) : : "s20", "scc"); break;
        case P_BALLOT_SCC:
            asm volatile(REP64("v_cmp_lt_u32 vcc, %0, %1\n s_cmp_lg_u64 vcc, 0\n s_cselect_b32 s20, 1, 2\n v_add_u32 %0, s20, %0\n") : "+v"(v) : "v"(w) : "vcc", "scc", "s20"); break;
        case P_MIX:
            asm volatile(REP64("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 3\n") : "+v"(v), "+s"(s) : "v"(w) : "scc"); break;
        case P_LDS128:
            asm volatile(REP64("ds_read_b128 v[100:103], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3ff0, v100\n") : "+v"(addr) : : "memory", "v100", "v101", "v102", "v103"); break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (v == 0xdeadbeef && s == 77 && d == 3.0 && addr == 12345) out[0] = v + s;   // keep results live
}

// marginal cost of a store / an LDS write inside a dependent VALU chain (8 v_add per group)
__global__ __launch_bounds__(1024) void probe2(int pattern, int iters, unsigned long long* out, unsigned* sink) {
    __shared__ unsigned lds[4096];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = 0;
    __syncthreads();
    unsigned v = lane, w = 1;
    const unsigned addr = (threadIdx.x * 4) & 16383, addr16 = (threadIdx.x * 16) & 16383;
    const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned* vp1 = sink + gtid;
    unsigned* vp4 = sink + gtid * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        switch (pattern) {
        case P_ST1:
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n global_store_dword %2, %0, off\n")
                         : "+v"(v) : "v"(w), "v"(vp1) : "memory"); break;
        case P_ST4:
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n global_store_dwordx4 %2, v[100:103], off\n")
                         : "+v"(v) : "v"(w), "v"(vp4) : "memory", "v100", "v101", "v102", "v103"); break;
        case P_DSW:
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n ds_write_b32 %2, %0\n")
                         : "+v"(v) : "v"(w), "v"(addr) : "memory"); break;
        case P_DSW128:
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n ds_write_b128 %2, v[100:103]\n")
                         : "+v"(v) : "v"(w), "v"(addr16) : "memory", "v100", "v101", "v102", "v103"); break;
        case P_ADD8:
            asm volatile(REP16("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n")
                         : "+v"(v) : "v"(w)); break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (v == 0xdeadbeef) out[0] = v + lds[lane];
}

int main() {
    unsigned* chase; unsigned long long* out; unsigned* sink;
    hipMalloc(&sink, 256 * 1024 * 16);
    const int NCH = 1 << 16;
    hipMalloc(&chase, NCH * 4); hipMalloc(&out, 256 * 16 * 8);
    std::vector<unsigned> h(NCH);
    for (int i = 0; i < NCH; ++i) h[i] = (unsigned)(((i * 4 + 4096 + 64) % (NCH * 4)) & ~3u);    // byte offset of the next word
    hipMemcpy(chase, h.data(), NCH * 4, hipMemcpyHostToDevice);
    std::vector<unsigned long long> o(256 * 16);
    const int iters = 16, chain = iters * 64;
    double wall_clock_mhz = 100.0;
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0);
    printf("s_memtime unit: shader/ref clock as reported below; wall clock rate attr %d kHz\n", clk);
    for (int pat = 0; pat < P_COUNT; ++pat) {
        double res[2];
        for (int mode = 0; mode < 2; ++mode) {
            const int blocks = mode == 0 ? 1 : 256, threads = mode == 0 ? 64 : 1024;
            if (pat >= P_ST1) {
                probe2<<<blocks, threads>>>(pat, iters, out, sink);
                probe2<<<blocks, threads>>>(pat, iters, out, sink);
            } else {
                probe<<<blocks, threads>>>(pat, iters, chase, out);   // warm
                probe<<<blocks, threads>>>(pat, iters, chase, out);
            }
            hipDeviceSynchronize();
            hipMemcpy(o.data(), out, blocks * (threads / 64) * 8, hipMemcpyDeviceToHost);
            double sum = 0; int n = blocks * (threads / 64);
            for (int i = 0; i < n; ++i) sum += (double)o[i];
            res[mode] = sum / n / (pat >= P_ST1 ? iters * 16 : chain);
        }
        printf("%-48s alone %7.2f   16 waves/CU %7.2f   (s_memtime ticks per link)\n", kNames[pat], res[0], res[1]);
    }
    // tick calibration: a kernel of known wall duration
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); probe<<<1, 64>>>(P_VALU, 4096, chase, out); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(o.data(), out, 8, hipMemcpyDeviceToHost);
    printf("calibration: %llu ticks in %.3f ms -> %.1f MHz tick rate\n", o[0], ms, (double)o[0] / ms / 1e3);
    (void)wall_clock_mhz;
    return 0;
}
