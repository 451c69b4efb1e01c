// mb_waves.hip — does a second (third, fourth) wave on a SIMD buy issue throughput for the lean step's instruction mix?
// One workgroup of 256 / 512 / 768 / 1024 threads on one CU (1 - 4 waves per SIMD), every wave runs the same stream and
// times itself.  Tooling, not product code.   build: hipcc --offload-arch=gfx950 -O3 tools/mb_waves.hip -o tools/mb_waves.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
__device__ inline uint64_t now() {
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void probe(uint64_t* out, double* gbuf, double seed) {
    __shared__ double lds[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a0 = seed + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, y = 0.0;
    double b = 1.0000001, c = 0.5, u = 0.25 + lane, sc = 1.0;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    uint64_t t0, t1;
    int slot = 0;
    auto rec = [&](uint64_t d) { if (lane == 0) out[wave * 16 + slot] = d; ++slot; };
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
    __syncthreads();
    t0 = now(); t1 = now(); rec(t1 - t0);
    // 1: 128 independent v_fma_f64
    __syncthreads(); t0 = now();
    asm volatile(REP16("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    t1 = now(); rec(t1 - t0);
    // 2: 128 independent v_add_u32
    __syncthreads(); t0 = now();
    asm volatile(REP16(REP4("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n") "") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(lane));
    t1 = now(); rec(t1 - t0);
    // 3: 128 states pattern: 32 x {fma, fmac_dpp, fmac, mul}
    __syncthreads(); t0 = now();
    asm volatile(REP16("v_fma_f64 %0, %5, %1, %6\n v_fma_f64 %2, %5, %3, %6\n v_fmac_f64_dpp %0, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %7, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_e32 %4, %5, %1\n v_fmac_f64_e32 %4, %5, %3\n v_mul_f64 %1, %5, %0\n v_mul_f64 %3, %5, %2\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(y) : "v"(b), "v"(c), "v"(u), "v"(sc));
    t1 = now(); rec(t1 - t0);
    // 4: mix of the pipelined step: 16 x {4 DP state ops, 2 v_mov_b32_dpp, 1 v_add_f64, 1 s_add}  (128 instr)
    uint32_t s = __builtin_amdgcn_readfirstlane(wave);
    __syncthreads(); t0 = now();
    asm volatile(REP16("v_fma_f64 %0, %5, %1, %6\n v_fmac_f64_dpp %0, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f64_e32 %4, %5, %1\n v_mov_b32_dpp %10, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f64 %1, %5, %0\n v_add_f64 %2, %2, %3\n s_add_u32 %11, %11, 3\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(y) : "v"(b), "v"(c), "v"(u), "v"(sc), "v"(i0), "v"(i1), "s"(s) : "scc");
    t1 = now(); rec(t1 - t0);
    // 5: 16 v_mfma_f64_16x16x4 (independent accumulators) — matrix pipe alone
    v4f64 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    __syncthreads(); t0 = now();
    asm volatile(REP4("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n")
                 : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(b), "v"(c));
    asm volatile("s_nop 7\n s_nop 7\n v_add_f64 %0, %0, %1" : "+v"(a4) : "v"(m3.x));
    t1 = now(); rec(t1 - t0);
    // 6: even waves 16 MFMA f64, odd waves 128 v_fma_f64: do they overlap?
    __syncthreads(); t0 = now();
    if (wave & 1) {
        asm volatile(REP16("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else {
        asm volatile(REP4("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n")
                     : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(b), "v"(c));
        asm volatile("s_nop 7\n s_nop 7\n v_add_f64 %0, %0, %1" : "+v"(a4) : "v"(m3.x));
    }
    t1 = now(); rec(t1 - t0);
    // 7: 32 ds_read_b128 pair-table pattern + 96 v_fma_f64 interleaved (1 : 3)
    {
        uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds + (lane & 1) * 16 + (wave & 7) * 512;
        v2f64 r0, r1;
        __syncthreads(); t0 = now();
        asm volatile(REP16("ds_read_b128 %8, %10\n v_fma_f64 %0, %0, %11, %12\n v_fma_f64 %1, %1, %11, %12\n v_fma_f64 %2, %2, %11, %12\n ds_read_b128 %9, %10 offset:32\n v_fma_f64 %3, %3, %11, %12\n v_fma_f64 %4, %4, %11, %12\n v_fma_f64 %5, %5, %11, %12\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(r0), "=&v"(r1) : "v"(addr), "v"(b), "v"(c) : "memory");
        t1 = now(); rec(t1 - t0);
        a6 += r0.x + r1.x;
    }
    // 8: 16 stores (1 KB) + 112 v_fma_f64 interleaved (1 : 7)
    {
        v2f64 val = {a0, a1};
        double* p = gbuf + (size_t)wave * 4096 + lane * 2;
        __syncthreads(); t0 = now();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("global_store_dwordx4 %0, %1, off offset:0" :: "v"(p + k * 128), "v"(val) : "memory");
            asm volatile("v_fma_f64 %0, %0, %7, %8\n v_fma_f64 %1, %1, %7, %8\n v_fma_f64 %2, %2, %7, %8\n v_fma_f64 %3, %3, %7, %8\n v_fma_f64 %4, %4, %7, %8\n v_fma_f64 %5, %5, %7, %8\n v_fma_f64 %6, %6, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6) : "v"(b), "v"(c));
        }
        uint64_t tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
        rec(tm - t0);
        t1 = now(); rec(t1 - t0);
    }
    // 10: 16 x (ds_write, barrier, ds_read, add): the exchange with this many waves
    {
        uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds + lane * 8;
        const int nw = blockDim.x >> 6;
        double r = a0;
        __syncthreads(); t0 = now();
#pragma unroll
        for (int k = 0; k < 16; ++k)
            asm volatile("ds_write_b64 %1, %0\n s_waitcnt lgkmcnt(0)\n s_barrier\n ds_read_b64 %0, %2\n s_waitcnt lgkmcnt(0)\n v_add_f64 %0, %0, 1.0"
                         : "+v"(r) : "v"(addr + wave * 512 + (k & 1) * 8192), "v"(addr + ((wave + 1) % nw) * 512 + (k & 1) * 8192) : "memory");
        t1 = now(); rec(t1 - t0);
        a4 += r;
    }
    gbuf[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + y + i0 + i1 + i2 + i3 + m0.x + m1.x + m2.x;
}
int main() {
    uint64_t* d_out; double* d_buf;
    hipMalloc(&d_out, 16 * 16 * sizeof(uint64_t));
    hipMalloc(&d_buf, 1 << 22);
    const char* names[] = {"empty", "128 independent v_fma_f64", "128 independent v_add_u32", "32 states (128 DP instr)", "mix: 16 x (4 DP state, 2 mov_dpp, add_f64, s_add)", "16 MFMA f64 16x16x4 + use",
                           "even waves 16 MFMA / odd waves 128 fma", "32 ds_read_b128 + 96 fma", "16 stores + 112 fma: issue", "  ... acknowledged", "16 x (ds_write, barrier, ds_read, add)"};
    for (int threads = 256; threads <= 1024; threads += 256) {
        hipMemset(d_out, 0, 256 * sizeof(uint64_t));
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, d_out, d_buf, 1.0);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(256);
        hipMemcpy(h.data(), d_out, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost);
        printf("---- %d waves per SIMD (%d threads): cycles per wave (waves 0, 1, 2, 3 and the slowest of all); empty bracket subtracted\n", threads / 256, threads);
        for (int i = 1; i < (int)(sizeof(names) / sizeof(names[0])); ++i) {
            printf("%-52s", names[i]);
            long long mx = 0;
            for (int w = 0; w < threads / 64; ++w) { long long v = (long long)h[w * 16 + i] - (long long)h[w * 16]; if (v > mx) mx = v; if (w < 4) printf(" %7lld", v); }
            printf("   max %7lld\n", mx);
        }
    }
    return 0;
}
