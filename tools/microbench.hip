// microbench.hip — latency probes used to size the chain kernels (not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N 256
__device__ inline uint64_t now() { return __builtin_amdgcn_s_memtime(); }
// pin `a` so the timed region cannot move across the clock reads
#define PIN(a) do { asm volatile("" : "+v"(a) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int CTRL, int RM, bool B>
__device__ inline double dppf(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, RM, 0xF, B);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, RM, 0xF, B);
    return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v) {
    v += dppf<0x111, 0xF, true>(v); v += dppf<0x112, 0xF, true>(v); v += dppf<0x114, 0xF, true>(v);
    v += dppf<0x118, 0xF, true>(v); v += dppf<0x142, 0xA, false>(v); v += dppf<0x143, 0xC, false>(v);
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__global__ void probe(double* out, uint64_t* t, double seed, double* gbuf) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = seed + lane, b = 1.0000001, c = 0.5;
    uint64_t t0, t1;
    // 1. dependent v_add_f64 chain
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = a + b;
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[0] = t1 - t0;
    // 2. dependent fma chain
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int i = 0; i < N; ++i) a = fma(a, b, c);
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[1] = t1 - t0;
    // 3. 8 independent fma chains (throughput)
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = a + k;
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = fma(x[k], b, c);
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(x[k]));
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[2] = t1 - t0;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += x[k];
    // 4. wave_sum x 16 dependent
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int i = 0; i < 16; ++i) a = wave_sum(a) * 0.015625 + lane;
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[3] = t1 - t0;
    // 5. LDS write -> read round trip (same wave), 32 dependent
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int i = 0; i < 32; ++i) { lds[threadIdx.x] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); a = lds[threadIdx.x ^ 1] + 1.0; }
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[4] = t1 - t0;
    // 6. readlane x 64 (32 doubles) dependent on a
    PIN(a); t0 = now(); PIN(a);
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) { int lo = __builtin_amdgcn_readlane(__double2loint(a), i), hi = __builtin_amdgcn_readlane(__double2hiint(a), i); acc += __hiloint2double(hi, lo); }
    asm volatile("" : "+v"(acc));
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[5] = t1 - t0;
    a += acc;
    // 7. barrier x 32 (workgroup), LDS-only fence
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int i = 0; i < 32; ++i) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[6] = t1 - t0;
    // 8. 16 coalesced global_store_dwordx2 per lane x 8 rounds (issue cost only)
    __attribute__((address_space(1))) double* g = (__attribute__((address_space(1))) double*)gbuf + (size_t)wave * 64 * 16 * 8 + lane;
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) g[(r * 16 + k) * 64] = a + k;
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[7] = t1 - t0;
    // 9. ldexp/frexp chain
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int i = 0; i < 64; ++i) { int e = __builtin_amdgcn_frexp_exp(a); a = ldexp(a, -e) + 1.5; }
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[8] = t1 - t0;
    // 10. barrier + LDS exchange like the chain kernel: write partial, barrier, read 4 partials
    PIN(a); t0 = now(); PIN(a);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        lds[(i & 1) * 512 + threadIdx.x] = a;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        a = lds[(i & 1) * 512 + lane] + lds[(i & 1) * 512 + 64 + lane] + lds[(i & 1) * 512 + 128 + lane] + lds[(i & 1) * 512 + 192 + lane];
    }
    PIN(a); t1 = now(); PIN(a); if (threadIdx.x == 0) t[9] = t1 - t0;
    out[threadIdx.x] = a;
}
int main() {
    double *out, *g; uint64_t* t;
    hipMalloc(&out, 4096 * 8); hipMalloc(&t, 64 * 8); hipMalloc(&g, 1 << 24);
    for (int threads : {64, 256, 320, 384, 512}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, out, t, 1.0, g); hipDeviceSynchronize(); }
        uint64_t h[16]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        printf("threads=%d: dep add %.1f cyc/op | dep fma %.1f | 8-indep fma %.1f cyc/op | wave_sum %.0f | lds rt %.0f | readlane dbl %.1f | barrier %.0f | store dwordx2 %.1f cyc/instr | frexp+ldexp+add %.0f | exch(barrier+4reads) %.0f\n",
               threads, h[0] / 256.0, h[1] / 256.0, h[2] / 256.0, h[3] / 16.0, h[4] / 32.0, h[5] / 32.0, h[6] / 32.0, h[7] / 128.0, h[8] / 64.0, h[9] / 16.0);
    }
    return 0;
}
