// Issue cost of the 32-bit integer multiplies Philox is made of (gfx950): independent instructions of one kind, one wave
// alone on its CU and 16 waves per CU.   hipcc --offload-arch=gfx950 -O3 tools/probes/intmul_probe.hip -o tools/_diag/intmul_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
__global__ __launch_bounds__(1024) void probe(int pattern, int iters, unsigned long long* out) {
    const int lane = threadIdx.x & 63;
    unsigned a = lane + 3, b = lane * 7 + 1, c = 5, d = 9, k = 0xD2511F53u;
    unsigned long long q0 = 1, q1 = 2;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        switch (pattern) {
        case 0: asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k)); break;
        case 1: asm volatile(REP16("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k)); break;
        case 2: asm volatile(REP16("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k)); break;
        case 3: asm volatile(REP16("v_mad_u64_u32 %0, vcc, %2, %4, 0\n v_mad_u64_u32 %1, vcc, %3, %4, 0\n v_mad_u64_u32 %0, vcc, %2, %4, 0\n v_mad_u64_u32 %1, vcc, %3, %4, 0\n") : "+v"(q0), "+v"(q1), "+v"(a), "+v"(b) : "v"(k) : "vcc"); break;
        case 4: asm volatile(REP16("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k)); break;
        case 5: asm volatile(REP16("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k)); break;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (a + b + c + d + (unsigned)q0 + (unsigned)q1 == 0x12345) out[0] = 1;
}
int main() {
    unsigned long long* out; hipMalloc(&out, 256 * 16 * 8);
    std::vector<unsigned long long> o(256 * 16);
    const char* names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_xor_b32"};
    for (int pat = 0; pat < 6; ++pat)
        for (int mode = 0; mode < 2; ++mode) {
            const int blocks = mode == 0 ? 1 : 256, threads = mode == 0 ? 64 : 1024, iters = 64;
            probe<<<blocks, threads>>>(pat, iters, out); probe<<<blocks, threads>>>(pat, iters, out);
            hipDeviceSynchronize();
            hipMemcpy(o.data(), out, blocks * (threads / 64) * 8, hipMemcpyDeviceToHost);
            double sum = 0; int n = blocks * (threads / 64);
            for (int i = 0; i < n; ++i) sum += (double)o[i];
            printf("%-16s %s: %.2f ticks per instruction (independent, one wave's view)\n", names[pat], mode ? "16 waves/CU" : "alone      ", sum / n / (iters * 64));
        }
    return 0;
}
