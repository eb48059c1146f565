// LDS atomic throughput probe (gfx950): cycles per wave-instruction of ds_add_f32 / ds_add_u32 / ds_add_f64 / plain RMW,
// 16 waves per CU all issuing, conflict-free addresses (lane i -> word i of a per-wave 256-word window) and random rows.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe lds_atomic_probe.hip && ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, long long* cyc, int iters, int spread) {
    extern __shared__ float acc[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) acc[i] = 0.0f;
    __syncthreads();
    uint32_t rng = threadIdx.x * 2654435761u + blockIdx.x;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        rng = rng * 1664525u + 1013904223u;
        // a group of 16 lanes hits one 64-float row: row chosen per group per iteration
        const uint32_t g = (__shfl((int)rng, lane & 48) >> 8) % (uint32_t)spread;
        float* p = acc + g * 64 + 4 * (lane & 15);
        const float v = (float)(it & 7);
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) __hip_atomic_fetch_add(p + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) __hip_atomic_fetch_add((int*)p + c, (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 2; ++c) __hip_atomic_fetch_add((double*)p + c, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 3) {      // non-atomic read-modify-write (racy: timing only)
            float4 x = *(float4*)p;
            x.x += v; x.y += v; x.z += v; x.w += v;
            *(float4*)p = x;
        } else if (MODE == 4) {
#pragma unroll
            for (int c = 0; c < 2; ++c) __hip_atomic_fetch_add((unsigned long long*)p + c, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 5) {      // lane-linear: 64 lanes -> 64 consecutive words (no bank conflict at all)
            float* q = acc + g * 64 + lane;
            __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 6) {
            int* q = (int*)acc + g * 64 + lane;
            __hip_atomic_fetch_add(q, (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 1024 + threadIdx.x] = acc[threadIdx.x];
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    const char* names[] = {"ds_add_f32 x4 (16 B per lane)", "ds_add_u32 x4", "ds_add_f64 x2", "plain ds_read_b128 + ds_write_b128", "ds_add_u64 x2",
                           "ds_add_f32 x1 lane-linear", "ds_add_u32 x1 lane-linear"};
    for (int spread : {400, 8}) {
        for (int mode = 0; mode < 7; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(1024), 131072, 0, out, cyc, iters, spread); break;
                    case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(1024), 131072, 0, out, cyc, iters, spread); break;
                    case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(1024), 131072, 0, out, cyc, iters, spread); break;
                    case 3: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(1024), 131072, 0, out, cyc, iters, spread); break;
                    case 4: hipLaunchKernelGGL(probe<4>, dim3(256), dim3(1024), 131072, 0, out, cyc, iters, spread); break;
                    case 5: hipLaunchKernelGGL(probe<5>, dim3(256), dim3(1024), 131072, 0, out, cyc, iters, spread); break;
                    case 6: hipLaunchKernelGGL(probe<6>, dim3(256), dim3(1024), 131072, 0, out, cyc, iters, spread); break;
                }
                hipDeviceSynchronize();
            }
            long long h[256];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
            // clock64 = s_memtime (100 MHz constant clock on gfx950?) -> report raw ticks per wave-iteration across 16 waves
            printf("rows %3d  %-40s %8.1f ticks per iteration of 16 waves (block total %.0f)\n", spread, names[mode], s / 256 / iters, s / 256);
        }
    }
    return 0;
}
