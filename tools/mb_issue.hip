// mb_issue.hip — what the instructions of the lean step cost ONE wave per SIMD (4 waves of a 256-thread workgroup on
// one CU, the lone-chain configuration): issue cost of independent streams, latency of dependent ones, the state
// pattern at different interleavings, stores and LDS reads of the step, the barrier.  Tooling, not product code.
// build: hipcc --offload-arch=gfx950 -O3 tools/mb_issue.hip -o tools/mb_issue.bin      run: tools/mb_issue.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__device__ inline uint64_t now() {
    uint64_t t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
typedef double v2f64 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void probe(uint64_t* out, double* gbuf, double seed) {
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a0 = seed + lane, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001, c = 0.5, u = 0.25 + lane, sc = 1.0;
    uint64_t t0, t1;
    int slot = 0;
    auto rec = [&](uint64_t d) { if (lane == 0) out[wave * 64 + slot] = d; ++slot; };
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();

    // 0: empty
    t0 = now(); t1 = now(); rec(t1 - t0);
    // 1: 64 independent v_fma_f64 (8 chains)
    t0 = now();
    asm volatile(REP8("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    t1 = now(); rec(t1 - t0);
    // 2: 64 dependent v_fma_f64
    t0 = now();
    asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
    t1 = now(); rec(t1 - t0);
    // 3: 64 independent v_fmac_f64_dpp row_newbcast (8 accumulators)
    t0 = now();
    asm volatile(REP8("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u), "v"(sc));
    t1 = now(); rec(t1 - t0);
    // 4: 64 independent v_mul_f64
    t0 = now();
    asm volatile(REP8("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    t1 = now(); rec(t1 - t0);
    // 5: the state, sequential: 16 x {fma, fmac_dpp, fmac (independent), mul}
    double y = 0.0;
    t0 = now();
    asm volatile(REP16("v_fma_f64 %0, %3, %1, %4\n v_fmac_f64_dpp %0, %5, %6 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_e32 %2, %3, %1\n v_mul_f64 %1, %3, %0\n")
                 : "+v"(a0), "+v"(a1), "+v"(y) : "v"(b), "v"(c), "v"(u), "v"(sc));
    t1 = now(); rec(t1 - t0);
    // 6: the state, two interleaved: 8 x {fma a, fma b, dpp a, dpp b, fmac, fmac, mul a, mul b}
    t0 = now();
    asm volatile(REP8("v_fma_f64 %0, %5, %1, %6\n v_fma_f64 %2, %5, %3, %6\n v_fmac_f64_dpp %0, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %7, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_e32 %4, %5, %1\n v_fmac_f64_e32 %4, %5, %3\n v_mul_f64 %1, %5, %0\n v_mul_f64 %3, %5, %2\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(y) : "v"(b), "v"(c), "v"(u), "v"(sc));
    t1 = now(); rec(t1 - t0);
    // 7: the state, four interleaved: 4 x {4 fma, 4 dpp, 4 fmac, 4 mul}
    t0 = now();
    asm volatile(REP4("v_fma_f64 %0, %9, %1, %10\n v_fma_f64 %2, %9, %3, %10\n v_fma_f64 %4, %9, %5, %10\n v_fma_f64 %6, %9, %7, %10\n"
                      "v_fmac_f64_dpp %0, %11, %12 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %11, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %4, %11, %12 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %11, %12 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                      "v_fmac_f64_e32 %8, %9, %1\n v_fmac_f64_e32 %8, %9, %3\n v_fmac_f64_e32 %8, %9, %5\n v_fmac_f64_e32 %8, %9, %7\n"
                      "v_mul_f64 %1, %9, %0\n v_mul_f64 %3, %9, %2\n v_mul_f64 %5, %9, %4\n v_mul_f64 %7, %9, %6\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(y) : "v"(b), "v"(c), "v"(u), "v"(sc));
    t1 = now(); rec(t1 - t0);
    // 8: 64 independent v_mov_b32_dpp (row_shr:1)
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    t0 = now();
    asm volatile(REP16("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
    t1 = now(); rec(t1 - t0);
    // 9: 16 dependent DPP reduction levels {mov lo, mov hi, add}
    int lo = 0, hi = 0;
    t0 = now();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int l2 = __builtin_amdgcn_update_dpp(0, __double2loint(a0), 0x111, 0xF, 0xF, true), h2 = __builtin_amdgcn_update_dpp(0, __double2hiint(a0), 0x111, 0xF, 0xF, true);
        a0 += __hiloint2double(h2, l2);
        asm volatile("" : "+v"(a0));
    }
    t1 = now(); rec(t1 - t0);
    // 10: 64 SALU (s_add_u32 dependent)
    uint32_t s = __builtin_amdgcn_readfirstlane(wave);
    t0 = now();
    asm volatile(REP64("s_add_u32 %0, %0, 3\n") : "+s"(s) :: "scc");
    t1 = now(); rec(t1 - t0);
    // 11: 16 v_readlane_b32 + dependent use
    t0 = now();
    asm volatile(REP16("v_readlane_b32 %1, %0, 63\n s_nop 3\n v_add_u32 %0, %1, %0\n") : "+v"(i0), "=&s"(s));
    t1 = now(); rec(t1 - t0);
    // 12: 32 global_store_dwordx4 (1 KB each), back to back, every wave its own 32 KB
    {
        v2f64 val = {a0, a1};
        double* p = gbuf + (size_t)wave * 4096 + lane * 2;
        t0 = now();
#pragma unroll
        for (int k = 0; k < 32; ++k) asm volatile("global_store_dwordx4 %0, %1, off offset:0" :: "v"(p + k * 128), "v"(val) : "memory");
        uint64_t tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
        rec(tm - t0);           // issue time of 32 stores
        t1 = now(); rec(t1 - t0);   // until all are acknowledged
    }
    // 14: 8 stores spaced by 10 independent fma each (the step's pattern)
    {
        v2f64 val = {a0, a1};
        double* p = gbuf + (size_t)wave * 4096 + lane * 2;
        t0 = now();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("global_store_dwordx4 %0, %1, off offset:0" :: "v"(p + k * 128), "v"(val) : "memory");
            asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
        uint64_t tm;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm) :: "memory");
        rec(tm - t0);           // 8 stores + 80 fma issued
        t1 = now(); rec(t1 - t0);
    }
    // 16: 32 ds_read_b128 (two distinct addresses per wave: the pair table pattern), results unused until the end
    {
        uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds + (lane & 1) * 16 + wave * 512;
        v2f64 r0, r1, r2, r3;
        t0 = now();
        asm volatile(REP8("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:32\n ds_read_b128 %2, %4 offset:64\n ds_read_b128 %3, %4 offset:96\n")
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
        uint64_t tm;
        asm volatile("s_memtime %0" : "=s"(tm) :: "memory");
        t1 = now(); rec(tm - t0); rec(t1 - t0);
        a2 += r0.x + r1.x + r2.x + r3.x;
    }
    // 18: ds_read_b64 per-lane (512 B), 32 of them
    {
        uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds + lane * 8 + wave * 512;
        double r0, r1, r2, r3;
        t0 = now();
        asm volatile(REP8("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:2048\n ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:6144\n")
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
        uint64_t tm;
        asm volatile("s_memtime %0" : "=s"(tm) :: "memory");
        t1 = now(); rec(tm - t0); rec(t1 - t0);
        a3 += r0 + r1 + r2 + r3;
    }
    // 20: 16 x {ds_write_b64, s_barrier, ds_read_b64 dependent}: the exchange of the plain step
    {
        uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds + lane * 8;
        double r = a0;
        __syncthreads();
        t0 = now();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("ds_write_b64 %1, %0\n s_waitcnt lgkmcnt(0)\n s_barrier\n ds_read_b64 %0, %2\n s_waitcnt lgkmcnt(0)\n v_add_f64 %0, %0, 1.0"
                         : "+v"(r) : "v"(addr + wave * 512 + (k & 1) * 2048), "v"(addr + ((wave + 1) & 3) * 512 + (k & 1) * 2048) : "memory");
        }
        t1 = now(); rec(t1 - t0);
        a4 += r;
    }
    // 21: 16 bare s_barrier
    __syncthreads();
    t0 = now();
    asm volatile(REP16("s_barrier\n") ::: "memory");
    t1 = now(); rec(t1 - t0);
    // 22: 16 x {v_frexp_exp, v_sub, v_ldexp_f64 dependent}
    {
        int e;
        t0 = now();
        asm volatile(REP16("v_frexp_exp_i32_f64 %1, %0\n v_sub_u32 %1, 3, %1\n v_ldexp_f64 %0, %0, %1\n") : "+v"(a5), "=&v"(e));
        t1 = now(); rec(t1 - t0);
    }
    // 23: 64 independent v_ldexp_f64
    {
        int e = 1;
        t0 = now();
        asm volatile(REP8("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));
        t1 = now(); rec(t1 - t0);
    }
    // 24: 64 independent v_add_f64
    t0 = now();
    asm volatile(REP8("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    t1 = now(); rec(t1 - t0);
    // 25: 64 v_add_u32 independent (32-bit VALU)
    t0 = now();
    asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(lane));
    t1 = now(); rec(t1 - t0);
    gbuf[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + y + i0 + i1 + i2 + i3 + lo + hi + s;
}

int main() {
    uint64_t* d_out; double* d_buf;
    hipMalloc(&d_out, 256 * sizeof(uint64_t));
    hipMalloc(&d_buf, 1 << 22);
    hipMemset(d_out, 0, 256 * sizeof(uint64_t));
    const char* names[] = {"empty", "64 independent v_fma_f64", "64 dependent v_fma_f64", "64 independent v_fmac_f64_dpp", "64 independent v_mul_f64",
                           "16 states sequential (64 instr)", "16 states, two interleaved (64 instr)", "16 states, four interleaved (64 instr)", "64 independent v_mov_b32_dpp",
                           "16 dependent DPP levels (nop, 2 mov_dpp, add)", "64 dependent s_add_u32", "16 x (readlane, nop 3, use)", "32 stores back to back: issue", "  ... until acknowledged",
                           "8 x (store + 10 fma): issue", "  ... until acknowledged", "32 ds_read_b128 (pair table): issue", "  ... landed", "32 ds_read_b64 per lane: issue", "  ... landed",
                           "16 x (ds_write, barrier, ds_read, add)", "16 bare s_barrier", "16 dependent (frexp_exp, sub, ldexp)", "64 independent v_ldexp_f64", "64 independent v_add_f64", "64 independent v_add_u32"};
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d_out, d_buf, 1.0);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(256);
    hipMemcpy(h.data(), d_out, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    printf("cycles (s_memtime ticks), four waves of one workgroup each timing itself; the empty bracket is subtracted\n");
    printf("%-48s %8s %8s %8s %8s\n", "", "wave 0", "wave 1", "wave 2", "wave 3");
    for (int i = 0; i < (int)(sizeof(names) / sizeof(names[0])); ++i) {
        printf("%-48s", names[i]);
        for (int w = 0; w < 4; ++w) printf(" %8lld", (long long)h[w * 64 + i] - (i ? (long long)h[w * 64] : 0));
        printf("\n");
    }
    return 0;
}
