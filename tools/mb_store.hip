// mb_store.hip — per-CU store / LDS-DMA throughput probes (tooling, not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef double v2f64 __attribute__((ext_vector_type(2)));
#define GAS __attribute__((address_space(1)))
#define LAS __attribute__((address_space(3)))
__device__ inline uint64_t now() { return __builtin_amdgcn_s_memtime(); }

// mode 0: 8 x dwordx4 per thread per iteration (32 KB per 256-thread block), back to back
// mode 1: same, 24 dependent-free FMAs between consecutive stores
// mode 2: 16 x dwordx2
// mode 3: 8 x dwordx4 with half the lanes masked off
template <int MODE>
__global__ __launch_bounds__(256) void k_store(double* buf, uint64_t* t, int iters) {
    GAS v2f64* dst = (GAS v2f64*)buf + (size_t)blockIdx.x * iters * 2048 + threadIdx.x;
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
    uint64_t t0 = now();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                dst[k * 256] = v2f64{a[k], a[k] + 1.0};
                if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 24; ++q) a[(k + q) & 7] = fma(a[(k + q) & 7], 1.0000001, 0.5);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if (MODE == 2) {
            GAS double* d1 = (GAS double*)dst;
#pragma unroll
            for (int k = 0; k < 16; ++k) d1[k * 256 + threadIdx.x] = a[k & 7];
        } else {
            if (threadIdx.x & 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dst[k * 256] = v2f64{a[k], a[k] + 1.0};
            }
        }
        dst += 2048;
    }
    uint64_t t1 = now();
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
    if (a[0] == 12345.678) buf[0] = a[1];
}

// LDS-DMA: `nw` loader waves each stream 32 KB/iteration slices HBM -> LDS, counted vmcnt
__global__ __launch_bounds__(256) void k_dma(const double* buf, uint64_t* t, int iters, int nw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= nw) return;
    const int per = 32 / nw;  // dwordx4 wave-instructions per column per loader wave
    const GAS char* g = (const GAS char*)buf + (size_t)blockIdx.x * iters * 32768 + (size_t)wave * per * 1024 + lane * 16;
    uint64_t t0 = now();
    for (int it = 0; it < iters; ++it) {
        LAS unsigned char* l = (LAS unsigned char*)ring + (it & 3) * 32768 + wave * per * 1024;
        for (int q = 0; q < per; ++q)
            __builtin_amdgcn_global_load_lds((const GAS void*)(g + q * 1024), (LAS void*)(l + q * 1024), 16, 0, 0);
        g += 32768;
        // keep two iterations in flight
        if (per == 32) __builtin_amdgcn_s_waitcnt((63 & 0xF) | ((63 >> 4) << 14) | 0x0F70);
        else if (per == 16) __builtin_amdgcn_s_waitcnt((32 & 0xF) | ((32 >> 4) << 14) | 0x0F70);
        else __builtin_amdgcn_s_waitcnt((16 & 0xF) | ((16 >> 4) << 14) | 0x0F70);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    uint64_t t1 = now();
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

// plain register loads: 4 waves x 8 dwordx4, 2 iterations in flight via unrolled double buffer
__global__ __launch_bounds__(256) void k_load(const double* buf, uint64_t* t, double* out, int iters) {
    const GAS v2f64* src = (const GAS v2f64*)buf + (size_t)blockIdx.x * iters * 2048 + threadIdx.x;
    v2f64 a[8], b[8];
    double acc = 0;
    uint64_t t0 = now();
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = src[k * 256];
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = src[2048 + k * 256];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += a[k].x + a[k].y;
        src += 4096;
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = src[k * 256];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += b[k].x + b[k].y;
    }
    uint64_t t1 = now();
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int iters = 2000;
    for (int blocks : {1, 256}) {
        double* buf; uint64_t* t; double* out;
        size_t bytes = (size_t)blocks * (iters + 4) * 32768;
        hipMalloc(&buf, bytes); hipMalloc(&t, blocks * 8); hipMalloc(&out, blocks * 256 * 8);
        hipMemset(buf, 0, bytes);
        std::vector<uint64_t> h(blocks);
        auto report = [&](const char* name) {
            hipDeviceSynchronize();
            hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("blocks %3d %-28s %8.1f cycles per 32 KB column\n", blocks, name, s / blocks / iters);
        };
        k_store<0><<<blocks, 256>>>(buf, t, iters); report("store 8 x dwordx4 b2b");
        k_store<1><<<blocks, 256>>>(buf, t, iters); report("store 8 x dwordx4 + 24 fma");
        k_store<2><<<blocks, 256>>>(buf, t, iters); report("store 16 x dwordx2");
        k_store<3><<<blocks, 256>>>(buf, t, iters); report("store 8 x dwordx4 half lanes");
        hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        for (int nw : {1, 2, 4}) { k_dma<<<blocks, 256, 131072>>>(buf, t, iters, nw); char nm[64]; snprintf(nm, 64, "lds-dma %d loader wave(s)", nw); report(nm); }
        k_load<<<blocks, 256>>>(buf, t, out, iters); report("register loads 4 waves");
        hipFree(buf); hipFree(t); hipFree(out);
    }
    return 0;
}
