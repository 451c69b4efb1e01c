// pg_sampler.hip — MI355X-native HaplotypeSampler (include/pangenie_sampler.h).
//
// The reference (src/haplotypesampler.cpp) runs `size` passes of an integer min-plus Viterbi over the H
// panel paths, each pass masking the paths earlier passes picked at a column and penalising the alleles
// they covered.  Per column the update is
//     cell_i = min( prev_i            (same path, if prev_i is unmasked; cost 0),
//                   min_{j != i} prev_j + cost(recombination) )  + emission(allele of path i)
// i.e. elementwise work on H values plus ONE reduction: the smallest and second smallest entry of the
// previous column (the minimum "over all j but i" is the first unless i holds it).  Integer work with a
// sequential dependency over the columns AND over the passes: the only parallelism is over the paths of a
// column (one workgroup) and over contigs (one workgroup each), so the design goal is the shortest
// possible dependent chain per column.  No MFMA, no GEMM shape.
//
// Fast path (every realistic input; the host checks the two bounds that make it exact):
//   ks_expand         all CUs, per pass: emission cost of every (column, path) cell as ONE byte (0xFF =
//                     masked by an earlier pass / beyond H), stored thread-major so that the sequential
//                     kernel reads one coalesced dword per lane and column
//   ks_forward_fast   one workgroup per contig, 4 paths per lane.  Values are kept RELATIVE to the
//                     previous column's minimum (all unmasked cells lie within recombination cost + 50 of
//                     it), so (value, index) packs into one u32 key and "smaller value, then smaller index"
//                     (the reference's tie rule, src/haplotypesampler.cpp:79-107) is v_min_u32: the two
//                     reductions per column are 6 DPP steps each, one LDS exchange + one barrier when the
//                     workgroup has more than one wave.  The cost bytes are prefetched 16 columns ahead.
//                     The backtrace is 1 bit per cell ("stayed on the path") + the two minima ids per column.
//   ks_backtrack_fast one wave per contig, 64 columns at a time: a ballot finds the next column at which
//                     the traced path switched; writes the sampled path and applies
//                     SamplingEmissions::penalize to the alleles on it
// General path (saturating arithmetic, any cost range, H up to 65534; PG_SAMPLER_KERNEL=general forces
// it): ks_forward / ks_backtrack with u64 keys and u16 backtrace ids.
// Costs are formed on the HOST exactly as the reference forms them (float / long double + truncation).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/pangenie_sampler.h"

#define KS_UMAX 0xFFFFFFFFu
#define KS_KMAX 0xFFFFFFFFFFFFFFFFull
#define KS_MAXPPT 64    // paths per thread of the general kernel (H <= 65536 at 1024 threads)
#define KS_MAXPASS 64   // passes whose picks are kept in LDS for the masks
#define KS_BT_BLOCK 32  // columns per backtrack block of the general kernel

namespace {

void set_err(char* err, size_t errlen, const char* fmt, ...) {
    if (!err || errlen == 0) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, errlen, fmt, ap);
    va_end(ap);
}

struct SamplerDev {
    uint32_t V, P, T, penalty;    // T = threads of the fast forward kernel (4 paths each)
    const uint32_t* allele_off;   // [V+1]
    const uint16_t* allele_id;    // [sumA]
    const uint16_t* path_allele;  // [V*P]
    const uint32_t* tcost;        // [V+48] cost of a recombination between columns c-1 and c
    uint16_t* ecost;              // [sumA] emission costs, penalised pass by pass
    uint32_t* paths;              // [size*V] sampled paths
    uint32_t* best;               // [size]
    // general path
    uint16_t* bt;                 // [V*P] backtrace ids of the pass in flight (0xFFFF = none)
    uint32_t* last_col;           // [P] last column of the pass in flight
    // fast path
    uint8_t* slot;                // [V*P] index of the cell's allele in its column's allele list (0xFF = not listed, 0xFE = masked)
    unsigned long long* ecell;    // [(V+32)*T] half word k of (c, t) = KS_BIAS + cost of path k*T + t at column c
    uint32_t* stay;               // [ceil((V-1)/16)*2*T] "cell continued its own path" bits, 16 columns per half word
    uint32_t* minima;             // [V] id of the smallest entry of column c-1
    uint32_t* last;               // [1] best path of the last column
};

// ------------------------------------------------------------------------------------------------
//  general path
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const int lo = __shfl_xor((int)(uint32_t)v, m), hi = __shfl_xor((int)(uint32_t)(v >> 32), m);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
// merge two sorted pairs of distinct keys into the two smallest
__device__ __forceinline__ void top2_merge(unsigned long long& a1, unsigned long long& a2, unsigned long long b1, unsigned long long b2) {
    const unsigned long long f = a1 < b1 ? a1 : b1;
    const unsigned long long s = a1 < b1 ? (a2 < b1 ? a2 : b1) : (a1 < b2 ? a1 : b2);
    a1 = f; a2 = s;
}

// one pass of the Viterbi: grid = contigs, block = T threads
template <int T, int PPT>
__global__ __launch_bounds__(T) void ks_forward(const SamplerDev* devs, uint32_t pass) {
    __shared__ unsigned long long s_k1[T / 64], s_k2[T / 64];
    __shared__ uint16_t s_aid[2][256];
    __shared__ uint16_t s_ec[2][256];
    __shared__ uint32_t s_pick[2][KS_MAXPASS];
    const SamplerDev d = devs[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t V = d.V, P = d.P;
    uint32_t prev[PPT];  // the column, in registers: path tid + k*T in prev[k]
    unsigned long long prev_ok = 0ull;  // bit k: path tid + k*T was unmasked in the previous column
    // stage column c's alleles / emission costs / earlier picks into LDS buffer c & 1
    auto stage = [&](uint32_t c) {
        if (c >= V) return;
        const uint32_t a0 = d.allele_off[c], A = d.allele_off[c + 1] - a0;
        for (uint32_t q = tid; q < A && q < 256u; q += T) { s_aid[c & 1u][q] = d.allele_id[a0 + q]; s_ec[c & 1u][q] = d.ecost[a0 + q]; }
        for (uint32_t q = tid; q < pass && q < KS_MAXPASS; q += T) s_pick[c & 1u][q] = d.paths[(size_t)q * V + c];
    };
    stage(0);
    __syncthreads();
    for (uint32_t c = 0; c < V; ++c) {
        const uint32_t b = c & 1u;
        const uint32_t a0 = d.allele_off[c], A = d.allele_off[c + 1] - a0;
        // ---- smallest and second smallest unmasked entry of the previous column
        unsigned long long k1 = KS_KMAX, k2 = KS_KMAX;
        if (c > 0) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const uint32_t i = tid + k * T;
                if (i < P && ((prev_ok >> k) & 1ull) && prev[k] != KS_UMAX) {  // a saturated entry is never a minimum (strict <)
                    const unsigned long long key = ((unsigned long long)prev[k] << 32) | i;
                    if (key < k1) { k2 = k1; k1 = key; }
                    else if (key < k2) k2 = key;
                }
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const unsigned long long o1 = shfl_xor_u64(k1, m), o2 = shfl_xor_u64(k2, m);
                top2_merge(k1, k2, o1, o2);
            }
            if (lane == 0) { s_k1[wave] = k1; s_k2[wave] = k2; }
        }
        stage(c + 1);  // next column's tables into the other buffer (read after the next barrier)
        __syncthreads();
        uint32_t first_val = KS_UMAX, first_id = KS_UMAX, second_val = KS_UMAX, second_id = KS_UMAX, tcost = 0;
        if (c > 0) {
            k1 = s_k1[0]; k2 = s_k2[0];
            for (int w = 1; w < T / 64; ++w) top2_merge(k1, k2, s_k1[w], s_k2[w]);
            if (k1 != KS_KMAX) { first_val = (uint32_t)(k1 >> 32); first_id = (uint32_t)k1; }
            if (k2 != KS_KMAX) { second_val = (uint32_t)(k2 >> 32); second_id = (uint32_t)k2; }
            tcost = d.tcost[c];
        }
        // ---- the column (reference src/haplotypesampler.cpp:242-283)
        unsigned long long cur_ok = 0ull;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const uint32_t i = tid + k * T;
            if (i >= P) continue;
            bool ok = true;  // SampledPaths::mask_indexes: false if an earlier pass picked path i here
            for (uint32_t q = 0; q < pass && q < KS_MAXPASS; ++q) ok = ok && (s_pick[b][q] != i);
            for (uint32_t q = KS_MAXPASS; q < pass; ++q) ok = ok && (d.paths[(size_t)q * V + c] != i);
            uint32_t cell = KS_UMAX;
            uint16_t back = 0xFFFFu;
            if (ok) {
                uint32_t previous_cell = 0;
                if (c > 0) {
                    const uint32_t hv = (i == first_id) ? second_val : first_val;
                    const uint32_t hid = (i == first_id) ? second_id : first_id;
                    previous_cell = hv + tcost;
                    if (previous_cell < hv) previous_cell = KS_UMAX;
                    back = (uint16_t)hid;  // (0xFFFFFFFF -> 0xFFFF: none)
                    if ((prev_ok >> k) & 1ull) {
                        const uint32_t same = prev[k];
                        if (same < previous_cell) { previous_cell = same; back = (uint16_t)i; }
                    }
                }
                const uint16_t allele = d.path_allele[(size_t)c * P + i];
                uint32_t e = 0;
                if (A <= 256u) {
                    for (uint32_t q = 0; q < A; ++q) if (s_aid[b][q] == allele) { e = s_ec[b][q]; break; }
                } else {
                    for (uint32_t q = 0; q < A; ++q) if (d.allele_id[a0 + q] == allele) { e = d.ecost[a0 + q]; break; }
                }
                cell = previous_cell + e;
                if (cell < previous_cell) cell = KS_UMAX;
                cur_ok |= 1ull << k;
            }
            prev[k] = cell;
            d.bt[(size_t)c * P + i] = back;
        }
        prev_ok = cur_ok;
        // (the barrier of the next iteration separates this column's reads of buffer b from the staging
        // of column c+2 into it; s_k1/s_k2 are rewritten only after that barrier too)
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const uint32_t i = tid + k * T;
        if (i < P) d.last_col[i] = prev[k];
    }
}

// SamplingEmissions::penalize on the allele the picked path carries (src/samplingemissions.cpp:39-45)
__device__ __forceinline__ void penalize(const SamplerDev& d, uint32_t c, uint32_t path) {
    const uint32_t a0 = d.allele_off[c], A = d.allele_off[c + 1] - a0;
    const uint16_t allele = d.path_allele[(size_t)c * d.P + path];
    for (uint32_t q = 0; q < A; ++q)
        if (d.allele_id[a0 + q] == allele) {
            uint16_t e = (uint16_t)(d.ecost[a0 + q] + d.penalty);  // unsigned short arithmetic, as in the reference
            if (e > 25u) e = 25u;
            d.ecost[a0 + q] = e;
            break;
        }
}

// backtrace of one pass: grid = contigs, block = 256 threads
__global__ __launch_bounds__(256) void ks_backtrack(const SamplerDev* devs, uint32_t pass, uint32_t lds_cols) {
    extern __shared__ uint16_t s_bt[];  // [lds_cols][P] when a block fits, else unused
    __shared__ uint32_t s_best;
    const SamplerDev d = devs[blockIdx.x];
    const uint32_t tid = threadIdx.x, V = d.V, P = d.P;
    if (tid == 0) {
        // best value in the last column: the FIRST minimum, masked entries included (they hold UINT_MAX)
        uint32_t bi = 0, bv = d.last_col[0];
        for (uint32_t i = 1; i < P; ++i) if (d.last_col[i] < bv) { bv = d.last_col[i]; bi = i; }
        d.best[pass] = bv;
        s_best = bi;
    }
    __syncthreads();
    uint32_t best = s_best;
    const bool lds = lds_cols > 0 && (size_t)lds_cols * P * 2 <= 64u * 1024u;
    int64_t hi = (int64_t)V - 1;
    while (hi >= 0) {
        const int64_t lo = (lds && hi + 1 > (int64_t)lds_cols) ? hi + 1 - lds_cols : 0;
        if (lds) {  // stage columns lo..hi (coalesced), then chase through LDS
            const size_t n = (size_t)(hi - lo + 1) * P;
            const uint16_t* src = d.bt + (size_t)lo * P;
            for (size_t q = tid; q < n; q += 256) s_bt[q] = src[q];
            __syncthreads();
        }
        if (tid == 0) {
            for (int64_t c = hi; c >= lo; --c) {
                d.paths[(size_t)pass * V + c] = best;
                penalize(d, (uint32_t)c, best);
                if (c > 0) best = lds ? s_bt[(size_t)(c - lo) * P + best] : d.bt[(size_t)c * P + best];
            }
            s_best = best;
        }
        __syncthreads();
        best = s_best;
        hi = lo - 1;
    }
}

// ------------------------------------------------------------------------------------------------
//  fast path
// ------------------------------------------------------------------------------------------------
#define KS_AS1 __attribute__((address_space(1)))  // loads / stores through these compile to global_* (vmcnt only), not flat_*
#define KS_BIAS 0x2000u    // added to every cost word: keeps the relative values positive
#define KS_MASKED 0x8000u  // cost word (before the bias) of a masked cell: above every live value
#define KS_TMAX 8191u      // largest recombination cost the fast path takes (KS_BIAS - 1)

// slot (index into the column's allele list) of every cell, once per run: grid-stride over (column, path)
__global__ __launch_bounds__(256) void ks_slots(const SamplerDev* devs) {
    const SamplerDev d = devs[blockIdx.y];
    const uint32_t V = d.V, P = d.P;
    const size_t n = (size_t)V * P;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (size_t)gridDim.x * 256) {
        const uint32_t c = (uint32_t)(g / P);
        const uint32_t a0 = d.allele_off[c], A = d.allele_off[c + 1] - a0;
        const uint16_t allele = d.path_allele[g];
        uint32_t slot = 0xFFu;  // not listed (malformed input): cost 0, nothing to penalise
        for (uint32_t q = 0; q < A && q < 254u; ++q) if (d.allele_id[a0 + q] == allele) { slot = q; break; }
        d.slot[g] = (uint8_t)slot;
    }
}

// after a pass: SamplingEmissions::penalize on the allele the picked path carries (reference
// src/haplotypesampler.cpp:162-164, src/samplingemissions.cpp:39-45), and the picked cell is masked for
// all later passes (SampledPaths::mask_indexes) by overwriting its slot byte.  One thread per column.
__global__ __launch_bounds__(256) void ks_apply(const SamplerDev* devs, uint32_t pass) {
    const SamplerDev d = devs[blockIdx.y];
    const uint32_t V = d.V, P = d.P;
    for (uint32_t c = blockIdx.x * 256u + threadIdx.x; c < V; c += gridDim.x * 256u) {
        const uint32_t path = d.paths[(size_t)pass * V + c];
        const size_t cell = (size_t)c * P + path;
        const uint32_t slot = d.slot[cell];
        if (slot < 0xFEu) {
            const uint32_t a = d.allele_off[c] + slot;
            uint16_t e = (uint16_t)(d.ecost[a] + d.penalty);  // unsigned short arithmetic, as in the reference
            if (e > 25u) e = 25u;
            d.ecost[a] = e;
        }
        d.slot[cell] = 0xFEu;
    }
}

// cost word of every cell of this pass: grid-stride over (column, thread of the forward kernel)
__global__ __launch_bounds__(256) void ks_expand(const SamplerDev* devs) {
    const SamplerDev d = devs[blockIdx.y];
    const uint32_t V = d.V, P = d.P, T = d.T;
    const size_t n = (size_t)V * T;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (size_t)gridDim.x * 256) {
        const uint32_t c = (uint32_t)(g / T), t = (uint32_t)(g % T);
        const uint32_t a0 = d.allele_off[c];
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = (uint32_t)k * T + t;
            uint32_t e = KS_MASKED;  // beyond H, or picked by an earlier pass
            if (i < P) {
                const uint32_t slot = d.slot[(size_t)c * P + i];
                if (slot < 0xFEu) e = d.ecost[a0 + slot];
                else if (slot == 0xFFu) e = 0u;
            }
            w[k] = e + KS_BIAS;
        }
        d.ecell[g] = (unsigned long long)(w[0] | (w[1] << 16)) | ((unsigned long long)(w[2] | (w[3] << 16)) << 32);
    }
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_min(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)KS_UMAX, (int)v, CTRL, ROWMASK, 0xF, false);
    return o < v ? o : v;
}
// minimum over the wave, uniform: row_shr 1/2/4/8 gather each row of 16 into its last lane, row_bcast15 /
// row_bcast31 carry the row results to lane 63
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
    v = dpp_min<0x111, 0xF>(v);
    v = dpp_min<0x112, 0xF>(v);
    v = dpp_min<0x114, 0xF>(v);
    v = dpp_min<0x118, 0xF>(v);
    v = dpp_min<0x142, 0xA>(v);
    v = dpp_min<0x143, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// the same over one row of 16 lanes (for values every row holds alike)
__device__ __forceinline__ uint32_t row_min(uint32_t v) {
    v = dpp_min<0x111, 0xF>(v);
    v = dpp_min<0x112, 0xF>(v);
    v = dpp_min<0x114, 0xF>(v);
    v = dpp_min<0x118, 0xF>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}
template <int NW>
__device__ __forceinline__ uint32_t block_min(uint32_t v, uint32_t* s_x, uint32_t wave, uint32_t lane) {
    const uint32_t fw = wave_min(v);
    if (NW == 1) return fw;
    if (lane == 0) s_x[wave] = fw;
    __syncthreads();
    return row_min(s_x[lane & (NW - 1)]);  // NW <= 16: every row of 16 lanes holds all the waves' minima
}

// One column of the relative-value DP (reference src/haplotypesampler.cpp:223-284).
//   rho[q] = value of path q*T + tid at the previous column, relative to THAT column's predecessor minimum,
//            minus the previous recombination cost, plus KS_BIAS (what the cost words carry); a masked cell
//            holds >= 0x8001.  All live values of a column lie within (recombination cost + 50) of its
//            minimum, so (rho << 16 | path id) is a u32 key whose minimum is the reference's "smallest value,
//            then smallest index".
//   With a recombination cost t >= 1 the minimum itself always continues its own path (same = min < second +
//   t), so the second smallest entry of the reference's helper never decides anything: ONE reduction.
//   d = same - t < 0  <=>  the cell continues its own path (strict, src/haplotypesampler.cpp:272).
template <int NW, int PPL>
__device__ __forceinline__ void fast_column(uint32_t (&rho)[4], const uint32_t (&idx)[4], uint32_t& acc, uint32_t& t_prev, unsigned long long w, uint32_t t,
                                            uint32_t* s_x, uint32_t wave, uint32_t lane, uint32_t (&hist)[4], uint32_t& first_id) {
    uint32_t kmin = KS_UMAX;
#pragma unroll
    for (int q = 0; q < PPL; ++q) kmin = min(kmin, (rho[q] << 16) | idx[q]);
    const uint32_t F = block_min<NW>(kmin, s_x, wave, lane);
    const uint32_t rmin = F >> 16;
    first_id = F & 0xFFFFu;
    acc += rmin + t_prev - KS_BIAS;  // absolute minimum of the previous column
    t_prev = t;
    const uint32_t mt = rmin + t;
    const uint32_t wx = (uint32_t)w, wy = (uint32_t)(w >> 32);
    const uint32_t e[4] = {wx & 0xFFFFu, wx >> 16, wy & 0xFFFFu, wy >> 16};
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int32_t dd = (int32_t)(rho[q] - mt);
        hist[q] = (hist[q] << 1) | ((uint32_t)dd >> 31);
        rho[q] = (uint32_t)min(dd, 0) + e[q];
    }
}

// Backtrace layout of the fast path: columns 1.. are grouped in blocks of 16 (column c -> block (c-1)/16,
// bit 15 - (c-1)%16); per block and thread two dwords hold the stay bits of its four paths
// (q0 | q1 << 16, q2 | q3 << 16); minima[c] = id of the smallest entry of column c-1.
// PPL = paths per lane (1, 2 or 4; fewer than 4 only with one wave): the cost words always carry four.
template <int NW, int PPL>
__global__ __launch_bounds__(NW * 64) void ks_forward_fast(const SamplerDev* devs, uint32_t pass) {
    constexpr uint32_t T = NW * 64;
    __shared__ uint32_t s_x[2][NW];
    const SamplerDev d = devs[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t V = d.V;
    const unsigned long long KS_AS1* ecell = (const unsigned long long KS_AS1*)d.ecell + tid;
    const uint32_t KS_AS1* tcost = (const uint32_t KS_AS1*)d.tcost;
    uint32_t KS_AS1* stay = (uint32_t KS_AS1*)d.stay + tid;
    uint32_t KS_AS1* minima = (uint32_t KS_AS1*)d.minima;
    const uint32_t idx[4] = {tid, T + tid, 2 * T + tid, 3 * T + tid};
    uint32_t rho[4];
    uint32_t acc = 0, t_prev = 0;
    {
        const unsigned long long w = ecell[0];  // column 0: the emission cost alone
        const uint32_t wx = (uint32_t)w, wy = (uint32_t)(w >> 32);
        rho[0] = wx & 0xFFFFu; rho[1] = wx >> 16; rho[2] = wy & 0xFFFFu; rho[3] = wy >> 16;
    }
    const uint32_t nfull = (V - 1) / 16;  // blocks of 16 columns after column 0
    // cost words of the block in flight and of the next one, in two register buffers that swap roles (no
    // copies): the loads of block b+1 are issued before block b is processed, a whole block (16 dependent
    // columns) ahead of their use.  ecell / tcost are padded by 32 columns, so the prefetch never needs a
    // bounds check.
    unsigned long long bufA[16], bufB[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bufA[j] = ecell[(size_t)(1u + j) * T];
    uint32_t tcA = tcost[1u + (lane & 15u)], tcB = 0;
    uint32_t par = 0;
    uint32_t hist[4] = {0u, 0u, 0u, 0u};
    auto do_block = [&](const unsigned long long (&use)[16], unsigned long long (&load)[16], const uint32_t& tc_use, uint32_t& tc_load, uint32_t blk) {
        const uint32_t c0 = 1u + 16u * blk;
#pragma unroll
        for (int j = 0; j < 16; ++j) load[j] = ecell[(size_t)(c0 + 16u + j) * T];
        tc_load = tcost[c0 + 16u + (lane & 15u)];
        // nothing of this block before its prefetches are in flight: the first use of `use` / tc_use then waits
        // with these 17 loads behind it instead of draining the previous block's stores
        __builtin_amdgcn_sched_barrier(0);
        uint32_t mins_v = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tc_use, j);
            uint32_t first_id;
            fast_column<NW, PPL>(rho, idx, acc, t_prev, use[j], t, s_x[par], wave, lane, hist, first_id);
            par ^= 1u;
            mins_v = lane == (uint32_t)j ? first_id : mins_v;
        }
        stay[(size_t)(blk * 2u) * T] = (hist[0] & 0xFFFFu) | (hist[1] << 16);
        stay[(size_t)(blk * 2u + 1u) * T] = (hist[2] & 0xFFFFu) | (hist[3] << 16);
        if (tid < 16u) minima[c0 + tid] = mins_v;
    };
    {
        uint32_t blk = 0;
        for (; blk + 2u <= nfull; blk += 2u) {
            do_block(bufA, bufB, tcA, tcB, blk);
            do_block(bufB, bufA, tcB, tcA, blk + 1u);
        }
        if (blk < nfull) do_block(bufA, bufB, tcA, tcB, blk);
    }
    {   // the last, partial block
        const uint32_t c0 = 1u + 16u * nfull;
        for (uint32_t c = c0; c < V; ++c) {
            uint32_t first_id;
            fast_column<NW, PPL>(rho, idx, acc, t_prev, ecell[(size_t)c * T], tcost[c], s_x[par], wave, lane, hist, first_id);
            par ^= 1u;
            if (tid == 0) minima[c] = first_id;
        }
        if (c0 < V) {
            const uint32_t sh = 16u - (V - c0);  // column c0 + j at bit 15 - j, as in a full block
            stay[(size_t)(nfull * 2u) * T] = ((hist[0] << sh) & 0xFFFFu) | ((hist[1] << sh) << 16);
            stay[(size_t)(nfull * 2u + 1u) * T] = ((hist[2] << sh) & 0xFFFFu) | ((hist[3] << sh) << 16);
        }
    }
    // best value in the last column = its first minimum (at least one cell is unmasked)
    uint32_t kmin = KS_UMAX;
#pragma unroll
    for (int q = 0; q < PPL; ++q) kmin = min(kmin, (rho[q] << 16) | idx[q]);
    const uint32_t F = block_min<NW>(kmin, s_x[par], wave, lane);
    if (tid == 0) { d.best[pass] = acc + (F >> 16) + t_prev - KS_BIAS; d.last[0] = F & 0xFFFFu; }
}

// grid = contigs, block = one wave.  The traced path stays on `b` until a column whose stay bit is 0 (there
// it came from the minimum of the column before); each lane inspects one block of 16 columns, so one step
// covers 1024 columns.
__global__ __launch_bounds__(64) void ks_backtrack_fast(const SamplerDev* devs, uint32_t pass) {
    const SamplerDev d = devs[blockIdx.x];
    const uint32_t lane = threadIdx.x, V = d.V, T = d.T;
    auto assign = [&](uint32_t lo, uint32_t hi, uint32_t path) {  // columns lo..hi carry `path` (penalties: ks_apply)
        for (uint32_t c = lo + lane; c <= hi; c += 64u) d.paths[(size_t)pass * V + c] = path;
    };
    uint32_t b = d.last[0], cur = V - 1;  // the state at column cur is b
    while (cur >= 1u) {
        const uint32_t blk_cur = (cur - 1u) / 16u, j_cur = (cur - 1u) % 16u;
        const bool valid = lane <= blk_cur;
        const uint32_t q = b / T, t = b % T;
        uint32_t word = 0xFFFFu;
        if (valid) {
            const uint32_t dw = d.stay[((size_t)(blk_cur - lane) * 2u + (q >> 1)) * T + t];
            word = (q & 1u) ? dw >> 16 : dw & 0xFFFFu;
        }
        if (lane == 0) word |= (1u << (15u - j_cur)) - 1u;  // columns above cur are behind us
        const uint32_t zero = ~word & 0xFFFFu;
        const unsigned long long sw = __ballot(valid && zero != 0u);
        if (sw == 0ull) {
            const uint32_t lo = blk_cur >= 63u ? (blk_cur - 63u) * 16u + 1u : 1u;
            assign(lo, cur, b);
            cur = lo - 1u;
            continue;
        }
        const uint32_t ls = (uint32_t)__ffsll((long long)sw) - 1u;
        const uint32_t wz = (uint32_t)__builtin_amdgcn_readlane((int)zero, ls);
        const uint32_t cs = (blk_cur - ls) * 16u + 1u + (15u - ((uint32_t)__ffs((int)wz) - 1u));  // the highest column at which the path switched
        assign(cs, cur, b);
        b = d.minima[cs];
        cur = cs - 1u;
    }
    assign(0u, 0u, b);
}

__global__ void ks_minima(const uint32_t* column, const uint8_t* mask, uint32_t n, uint32_t* out4) {
    __shared__ unsigned long long s_k1[4], s_k2[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long k1 = KS_KMAX, k2 = KS_KMAX;
    for (uint32_t i = tid; i < n; i += 256)
        if (mask[i] && column[i] != KS_UMAX) {
            const unsigned long long key = ((unsigned long long)column[i] << 32) | i;
            if (key < k1) { k2 = k1; k1 = key; }
            else if (key < k2) k2 = key;
        }
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned long long o1 = shfl_xor_u64(k1, m), o2 = shfl_xor_u64(k2, m);
        top2_merge(k1, k2, o1, o2);
    }
    if (lane == 0) { s_k1[wave] = k1; s_k2[wave] = k2; }
    __syncthreads();
    if (tid == 0) {
        k1 = s_k1[0]; k2 = s_k2[0];
        for (int w = 1; w < 4; ++w) top2_merge(k1, k2, s_k1[w], s_k2[w]);
        out4[0] = k1 == KS_KMAX ? KS_UMAX : (uint32_t)k1;
        out4[1] = k2 == KS_KMAX ? KS_UMAX : (uint32_t)k2;
        out4[2] = k1 == KS_KMAX ? KS_UMAX : (uint32_t)(k1 >> 32);
        out4[3] = k2 == KS_KMAX ? KS_UMAX : (uint32_t)(k2 >> 32);
    }
}

thread_local double g_last_ms[3] = {0.0, 0.0, 0.0};
thread_local int g_last_kernel = 0;

unsigned on_slot(const pg_contig_batch* b, uint32_t slot, uint32_t k) {
    const uint32_t off = b->allele_kmer_off[slot];
    if (k < off || k >= off + 32u) return 0;
    return (b->allele_kmer_mask[slot] >> (k - off)) & 1u;
}

int check_panel(const pg_contig_batch* b, uint32_t size, char* err, size_t errlen) {
    const uint32_t V = b->n_variants, P = b->n_paths;
    if (P < 2) { set_err(err, errlen, "HaplotypeSampler needs at least two paths"); return PG_ERR_INVALID; }
    if (P > 65534u) { set_err(err, errlen, "at most 65534 paths (reference README.md:260)"); return PG_ERR_UNSUPPORTED; }
    if (size >= P) { set_err(err, errlen, "more passes than paths"); return PG_ERR_INVALID; }
    if (!b->variant_pos || !b->kmer_off || !b->allele_off || !b->allele_id || !b->allele_flags || !b->allele_kmer_off ||
        !b->allele_kmer_mask || !b->path_allele || (b->kmer_off[V] > 0 && !b->kmer_count)) {
        set_err(err, errlen, "batch has null arrays");
        return PG_ERR_INVALID;
    }
    for (uint32_t v = 0; v < V; ++v)
        if (b->allele_off[v + 1] <= b->allele_off[v] || b->kmer_off[v + 1] < b->kmer_off[v]) { set_err(err, errlen, "malformed offsets at variant %u", v); return PG_ERR_INVALID; }
    return PG_OK;
}

}  // namespace

extern "C" int pg_sampler_emission_costs(const pg_contig_batch* b, uint16_t* cost) {
    if (!b || !cost) return PG_ERR_INVALID;
    for (uint32_t v = 0; v < b->n_variants; ++v) {
        const uint32_t k0 = b->kmer_off[v], K = b->kmer_off[v + 1] - k0;
        for (uint32_t s = b->allele_off[v]; s < b->allele_off[v + 1]; ++s) {
            if (b->allele_flags[s] & 1) { cost[s] = 50; continue; }  // undefined allele (src/samplingemissions.cpp:18-21)
            /* total = KmerPath::nr_kmers (popcount of the window, src/kmerpath.cpp:50-55); present = k-mers of the
             * variant with a read count >= 3 that lie on the allele (src/multiallelicuniquekmers.cpp:155-162) */
            unsigned short total = (unsigned short)__builtin_popcount(b->allele_kmer_mask[s]), present = 0;
            for (uint32_t k = 0; k < K; ++k)
                if (b->kmer_count[k0 + k] >= 3 && on_slot(b, s, k)) present += 1;
            const float fraction = total > 0 ? present / (float)total : 1.0f;
            // the reference's `log10(fraction)` is the FLOAT overload (<cmath>, using namespace std): log10f
            if (fraction > 0.0) cost[s] = (unsigned short)(-10.0 * log10f(fraction));
            else cost[s] = 25;
        }
    }
    return PG_OK;
}

extern "C" uint32_t pg_sampler_transition_cost(uint64_t from_pos, uint64_t to_pos, double recombrate, uint32_t nr_paths,
                                               long double effective_N) {
    const long double distance = (to_pos - from_pos) * 0.000004L * ((long double)recombrate) * effective_N;
    // the reference's exp()/log10() here are the C double functions (no `using namespace std` in
    // src/samplingtransitions.cpp), the products around them long double
    const long double recomb_prob = (1.0L - exp((double)(-distance / (long double)nr_paths))) * (1.0L / (long double)nr_paths);
    const double cost = -10.0 * log10((double)recomb_prob);
    if (!(cost < 4294967295.0)) return KS_UMAX;  // coincident positions: undefined in the reference, saturates here
    if (cost < 0.0) return 0;
    return (unsigned int)cost;
}

#define HIP_TRY(call)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            set_err(err, errlen, "%s failed: %s", #call, hipGetErrorString(e_));          \
            rc = PG_ERR_DEVICE;                                                           \
            goto done;                                                                    \
        }                                                                                 \
    } while (0)

extern "C" int pg_sampler_column_minima(const uint32_t* column, const uint8_t* mask, uint32_t n, int device, uint32_t out4[4],
                                        char* err, size_t errlen) {
    if (!column || !mask || !out4 || n == 0) { set_err(err, errlen, "bad argument"); return PG_ERR_INVALID; }
    int ndev = 0, rc = PG_OK;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(err, errlen, "no HIP device available (no CPU fallback)"); return PG_ERR_DEVICE; }
    uint32_t *d_col = nullptr, *d_out = nullptr;
    uint8_t* d_mask = nullptr;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc((void**)&d_col, (size_t)n * 4));
    HIP_TRY(hipMalloc((void**)&d_mask, n));
    HIP_TRY(hipMalloc((void**)&d_out, 16));
    HIP_TRY(hipMemcpy(d_col, column, (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_mask, mask, n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ks_minima, dim3(1), dim3(256), 0, nullptr, d_col, d_mask, n, d_out);
    HIP_TRY(hipMemcpy(out4, d_out, 16, hipMemcpyDeviceToHost));
done:
    if (d_col) hipFree(d_col);
    if (d_mask) hipFree(d_mask);
    if (d_out) hipFree(d_out);
    return rc;
}

namespace {
// what pg_sampler_then_job keeps of a sampler run: the arena stays allocated (the caller frees it), and for every
// contig with variants the device arrays the panel update reads
struct SamplerKeep {
    unsigned char* arena = nullptr;
    struct Contig { uint32_t g; const uint32_t* allele_off; const uint16_t* allele_id; const uint16_t* path_allele; const uint32_t* paths; };
    std::vector<Contig> contigs;
};
int sampler_run_core(const pg_contig_batch* panels, uint32_t n_contigs, uint32_t size, double recombrate,
                     long double effective_N, uint16_t allele_penalty, int device, uint32_t* const* sampled_paths,
                     uint32_t* const* best_scores, SamplerKeep* keep, char* err, size_t errlen);
}  // namespace

extern "C" int pg_sampler_run_batch(const pg_contig_batch* panels, uint32_t n_contigs, uint32_t size, double recombrate,
                                    long double effective_N, uint16_t allele_penalty, int device, uint32_t* const* sampled_paths,
                                    uint32_t* const* best_scores, char* err, size_t errlen) {
    if (!panels || !sampled_paths) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    return sampler_run_core(panels, n_contigs, size, recombrate, effective_N, allele_penalty, device, sampled_paths, best_scores, nullptr, err, errlen);
}

namespace {
int sampler_run_core(const pg_contig_batch* panels, uint32_t n_contigs, uint32_t size, double recombrate,
                     long double effective_N, uint16_t allele_penalty, int device, uint32_t* const* sampled_paths,
                     uint32_t* const* best_scores, SamplerKeep* keep, char* err, size_t errlen) {
    if (size < 1 || n_contigs == 0) return PG_OK;  // reference src/haplotypesampler.cpp:28
    // contigs without variants have nothing to sample
    std::vector<uint32_t> live;
    for (uint32_t g = 0; g < n_contigs; ++g) {
        if (panels[g].n_variants == 0) continue;
        if (!keep && !sampled_paths[g]) { set_err(err, errlen, "null output for contig %u", g); return PG_ERR_INVALID; }
        const int rc0 = check_panel(&panels[g], size, err, errlen);
        if (rc0 != PG_OK) return rc0;
        live.push_back(g);
    }
    if (live.empty()) return PG_OK;
    int ndev = 0, rc = PG_OK;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err(err, errlen, "no HIP device available (no CPU fallback)"); return PG_ERR_DEVICE; }
    if (device < 0 || device >= ndev) { set_err(err, errlen, "bad device %d", device); return PG_ERR_INVALID; }
    const uint32_t n = (uint32_t)live.size();
    // host-side costs, and the bounds under which the relative-value kernel is exact: every recombination
    // cost in 1..KS_TMAX (with two or more paths the reference's formula gives >= 3; the largest real ones are
    // a few hundred), 50 (V + 1) + the largest cost fits 32 bits (no saturation anywhere), <= 4096 paths,
    // <= 254 alleles per column
    std::vector<std::vector<uint16_t>> ecost(n);
    std::vector<std::vector<uint32_t>> tcost(n);
    uint32_t maxP = 0, maxV = 0;
    bool fast = true;
    for (uint32_t j = 0; j < n; ++j) {
        const pg_contig_batch* b = &panels[live[j]];
        const uint32_t V = b->n_variants, P = b->n_paths;
        ecost[j].resize(b->allele_off[V]);
        pg_sampler_emission_costs(b, ecost[j].data());
        tcost[j].assign(V, 0);
        uint32_t tmax = 0, tmin = 1;
        for (uint32_t c = 1; c < V; ++c) {
            tcost[j][c] = pg_sampler_transition_cost(b->variant_pos[c - 1], b->variant_pos[c], recombrate, P, effective_N);
            if (tcost[j][c] > tmax) tmax = tcost[j][c];
            if (tcost[j][c] < tmin) tmin = tcost[j][c];
        }
        uint32_t maxA = 0;
        for (uint32_t v = 0; v < V; ++v) if (b->allele_off[v + 1] - b->allele_off[v] > maxA) maxA = b->allele_off[v + 1] - b->allele_off[v];
        if (tmin < 1u || tmax > KS_TMAX || 50.0 * ((double)V + 1.0) + tmax >= 4294967295.0 || P > 4096u || maxA > 254u) fast = false;
        if (P > maxP) maxP = P;
        if (V > maxV) maxV = V;
    }
    if (const char* e = getenv("PG_SAMPLER_KERNEL")) {
        if (!strcmp(e, "general")) fast = false;
        else if (!strcmp(e, "fast") && !fast) { set_err(err, errlen, "PG_SAMPLER_KERNEL=fast: the panel is outside the fast kernel's bounds"); return PG_ERR_UNSUPPORTED; }
    }
    uint32_t NW = 1;
    while (NW * 256u < maxP) NW *= 2;  // 4 paths per lane
    const uint32_t T = NW * 64u;

    std::vector<SamplerDev> devs(n);
    unsigned char* arena = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 255) / 256 * 256; size_t o = off; off += bytes ? bytes : 8; return o; };
    struct Offs { size_t aoff, aid, pa, tc, ec, paths, best, bt, last_col, ecell, stay, minima, last, slot; };
    std::vector<Offs> offs(n);
    for (uint32_t j = 0; j < n; ++j) {
        const pg_contig_batch* b = &panels[live[j]];
        const size_t V = b->n_variants, P = b->n_paths, sumA = b->allele_off[V];
        Offs& o = offs[j];
        o.aoff = take((V + 1) * 4); o.aid = take(sumA * 2); o.pa = take(V * P * 2); o.tc = take((V + 48) * 4); o.ec = take(sumA * 2);
        o.paths = take((size_t)size * V * 4); o.best = take((size_t)size * 4);
        if (fast) { o.slot = take(V * P); o.ecell = take((V + 32) * T * 8); o.stay = take(((V + 14) / 16 + 1) * 2 * T * 4); o.minima = take(V * 4); o.last = take(4); o.bt = o.last_col = 0; }
        else { o.bt = take(V * P * 2); o.last_col = take(P * 4); o.ecell = o.stay = o.minima = o.last = o.slot = 0; }
    }
    const size_t o_devs = take(sizeof(SamplerDev) * n);
    double ms[3] = {0.0, 0.0, 0.0};
    HIP_TRY(hipSetDevice(device));
    {
        hipError_t he = hipMalloc((void**)&arena, off);
        if (he != hipSuccess) {   // cached arenas of the one-shot call may be what stands in the way
            pg_hmm_release_cache();
            hipSetDevice(device);
            he = hipMalloc((void**)&arena, off);
        }
        if (he != hipSuccess) { set_err(err, errlen, "hipMalloc(%zu bytes) failed: %s", off, hipGetErrorString(he)); rc = PG_ERR_NOMEM; goto done; }
    }
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    for (uint32_t j = 0; j < n; ++j) {
        const pg_contig_batch* b = &panels[live[j]];
        const size_t V = b->n_variants, P = b->n_paths, sumA = b->allele_off[V];
        const Offs& o = offs[j];
        SamplerDev& d = devs[j];
        memset(&d, 0, sizeof(d));
        d.V = (uint32_t)V; d.P = (uint32_t)P; d.T = T; d.penalty = allele_penalty;
        d.allele_off = (const uint32_t*)(arena + o.aoff); d.allele_id = (const uint16_t*)(arena + o.aid);
        d.path_allele = (const uint16_t*)(arena + o.pa); d.tcost = (const uint32_t*)(arena + o.tc);
        d.ecost = (uint16_t*)(arena + o.ec); d.paths = (uint32_t*)(arena + o.paths); d.best = (uint32_t*)(arena + o.best);
        if (fast) {
            d.slot = (uint8_t*)(arena + o.slot); d.ecell = (unsigned long long*)(arena + o.ecell); d.stay = (uint32_t*)(arena + o.stay);
            d.minima = (uint32_t*)(arena + o.minima); d.last = (uint32_t*)(arena + o.last);
        } else {
            d.bt = (uint16_t*)(arena + o.bt); d.last_col = (uint32_t*)(arena + o.last_col);
        }
        HIP_TRY(hipMemcpy(arena + o.aoff, b->allele_off, (V + 1) * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(arena + o.aid, b->allele_id, sumA * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(arena + o.pa, b->path_allele, V * P * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(arena + o.tc, tcost[j].data(), V * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(arena + o.ec, ecost[j].data(), sumA * 2, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(arena + o_devs, devs.data(), sizeof(SamplerDev) * n, hipMemcpyHostToDevice));
    {
        const SamplerDev* dd = (const SamplerDev*)(arena + o_devs);
        // general path: backtrack blocks through LDS when KS_BT_BLOCK columns of u16 ids fit into 64 KB
        const uint32_t lds_cols = ((size_t)KS_BT_BLOCK * maxP * 2 <= 64u * 1024u) ? KS_BT_BLOCK : 0u;
        const size_t lds_bytes = (size_t)lds_cols * maxP * 2;
        const uint32_t ex_blocks = 2048u;
        for (uint32_t pass = 0; pass < size; ++pass) {
            HIP_TRY(hipEventRecord(ev[0], nullptr));
            if (fast && pass == 0) hipLaunchKernelGGL(ks_slots, dim3(ex_blocks, n), dim3(256), 0, nullptr, dd);
            if (fast) hipLaunchKernelGGL(ks_expand, dim3(ex_blocks, n), dim3(256), 0, nullptr, dd);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(ev[1], nullptr));
            if (fast) {
                if (maxP <= 64u) hipLaunchKernelGGL((ks_forward_fast<1, 1>), dim3(n), dim3(64), 0, nullptr, dd, pass);
                else if (maxP <= 128u) hipLaunchKernelGGL((ks_forward_fast<1, 2>), dim3(n), dim3(64), 0, nullptr, dd, pass);
                else switch (NW) {
                    case 1: hipLaunchKernelGGL((ks_forward_fast<1, 4>), dim3(n), dim3(64), 0, nullptr, dd, pass); break;
                    case 2: hipLaunchKernelGGL((ks_forward_fast<2, 4>), dim3(n), dim3(128), 0, nullptr, dd, pass); break;
                    case 4: hipLaunchKernelGGL((ks_forward_fast<4, 4>), dim3(n), dim3(256), 0, nullptr, dd, pass); break;
                    case 8: hipLaunchKernelGGL((ks_forward_fast<8, 4>), dim3(n), dim3(512), 0, nullptr, dd, pass); break;
                    default: hipLaunchKernelGGL((ks_forward_fast<16, 4>), dim3(n), dim3(1024), 0, nullptr, dd, pass); break;
                }
            } else if (maxP <= 256u) hipLaunchKernelGGL((ks_forward<256, 1>), dim3(n), dim3(256), 0, nullptr, dd, pass);
            else if (maxP <= 1024u) hipLaunchKernelGGL((ks_forward<1024, 1>), dim3(n), dim3(1024), 0, nullptr, dd, pass);
            else if (maxP <= 4096u) hipLaunchKernelGGL((ks_forward<1024, 4>), dim3(n), dim3(1024), 0, nullptr, dd, pass);
            else if (maxP <= 16384u) hipLaunchKernelGGL((ks_forward<1024, 16>), dim3(n), dim3(1024), 0, nullptr, dd, pass);
            else hipLaunchKernelGGL((ks_forward<1024, KS_MAXPPT>), dim3(n), dim3(1024), 0, nullptr, dd, pass);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(ev[2], nullptr));
            if (fast) {
                hipLaunchKernelGGL(ks_backtrack_fast, dim3(n), dim3(64), 0, nullptr, dd, pass);
                hipLaunchKernelGGL(ks_apply, dim3(256, n), dim3(256), 0, nullptr, dd, pass);
            } else hipLaunchKernelGGL(ks_backtrack, dim3(n), dim3(256), lds_bytes, nullptr, dd, pass, lds_cols);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(ev[3], nullptr));
            HIP_TRY(hipEventSynchronize(ev[3]));
            float a = 0.f, f = 0.f, k = 0.f;
            HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
            HIP_TRY(hipEventElapsedTime(&f, ev[1], ev[2]));
            HIP_TRY(hipEventElapsedTime(&k, ev[2], ev[3]));
            ms[0] += a; ms[1] += f; ms[2] += k;
        }
    }
    for (uint32_t j = 0; j < n; ++j) {
        const size_t V = panels[live[j]].n_variants;
        if (sampled_paths && sampled_paths[live[j]]) HIP_TRY(hipMemcpy(sampled_paths[live[j]], arena + offs[j].paths, (size_t)size * V * 4, hipMemcpyDeviceToHost));
        if (best_scores && best_scores[live[j]]) HIP_TRY(hipMemcpy(best_scores[live[j]], arena + offs[j].best, (size_t)size * 4, hipMemcpyDeviceToHost));
        if (keep) keep->contigs.push_back({live[j], devs[j].allele_off, devs[j].allele_id, devs[j].path_allele, devs[j].paths});
    }
    g_last_ms[0] = ms[0]; g_last_ms[1] = ms[1]; g_last_ms[2] = ms[2];
    g_last_kernel = fast ? (int)NW : 0;
    if (keep) { keep->arena = arena; arena = nullptr; }
done:
    for (auto& e : ev) if (e) hipEventDestroy(e);
    if (arena) hipFree(arena);
    return rc;
}
}  // namespace

extern "C" int pg_sampler_run(const pg_contig_batch* b, uint32_t size, double recombrate, long double effective_N,
                              uint16_t allele_penalty, int device, uint32_t* sampled_paths, uint32_t* best_scores,
                              char* err, size_t errlen) {
    if (!b || !sampled_paths) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    uint32_t* sp[1] = {sampled_paths};
    uint32_t* bs[1] = {best_scores};
    return pg_sampler_run_batch(b, 1, size, recombrate, effective_N, allele_penalty, device, sp, bs, err, errlen);
}

// ------------------------------------------------------------------------------------------------
//  Sampler -> panel update -> genotyping job, the panel staying on the device.
//
//  The reference's constructor ends with UniqueKmers::update_paths on every variant (src/haplotypesampler.cpp:44,
//  :296-309; src/biallelicuniquekmers.cpp:223-260, src/multiallelicuniquekmers.cpp:195-232): the panel keeps, at
//  variant v, path j = sampled_paths[j][v] (+ the reference path 0 when add_reference), the alleles those paths carry,
//  and the k-mers that lie on at least one kept allele — re-indexed in their old order, every allele's window starting
//  at its first k-mer (src/kmerpath.cpp:13-17).  Here two kernels do that on the flat arrays, one wave per variant:
//    ku_count  new path -> allele row (written at once: its place needs no scan), kept alleles / k-mers counted
//    ku_write  compacted allele and k-mer arrays at the offsets the host formed from the counts
//  Only the two counts per variant travel to the host (it plans the job's arena from them) and the offsets back.
// ------------------------------------------------------------------------------------------------
namespace {

struct UpdateDev {
    uint32_t V, P, S, size;            // S = kept paths = size (+ 1 with the reference path)
    // old panel (device)
    const uint32_t* kmer_off; const uint16_t* kmer_count; const uint32_t* allele_off; const uint16_t* allele_id;
    const uint8_t* allele_flags; const uint16_t* allele_koff; const uint32_t* allele_kmask; const uint16_t* path_allele;
    const uint32_t* paths;             // [size * V] sampled path of pass s at variant v
    // new panel (device)
    uint32_t* counts;                  // [2 V] kept alleles, kept k-mers of every variant
    const uint32_t* new_koff; const uint32_t* new_aoff;   // [V + 1] (ku_write)
    uint16_t* new_kcount; uint16_t* new_aid; uint8_t* new_aflags; uint16_t* new_akoff; uint32_t* new_akmask; uint16_t* new_pa;
    uint32_t* err;
};
#define KU_MAX_PATHS 1024u   // kept paths staged in LDS per variant
#define KU_MAX_KMERS 2048u   // k-mers of a variant whose new index is staged in LDS
#define KU_MAX_ALLELES 1024u

// which alleles / k-mers of variant v stay (shared by both kernels): kept alleles as a bitmap in LDS, kept k-mers as
// flags; returns through LDS: row[S] new path alleles, aKeep[A] / kKeep[K] 0 / 1
__device__ void ku_mark(const UpdateDev& d, uint32_t v, uint32_t lane, uint16_t* row, uint8_t* aKeep, uint8_t* kKeep) {
    const uint32_t a0 = d.allele_off[v], A = d.allele_off[v + 1] - a0, k0 = d.kmer_off[v], K = d.kmer_off[v + 1] - k0;
    for (uint32_t j = lane; j < d.S; j += 64u) {
        const uint32_t path = j < d.size ? d.paths[(size_t)j * d.V + v] : 0u;   // (the reference path, src/haplotypesampler.cpp:44)
        row[j] = d.path_allele[(size_t)v * d.P + path];
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t a = lane; a < A; a += 64u) {
        const uint16_t id = d.allele_id[a0 + a];
        uint8_t keep = 0;
        for (uint32_t j = 0; j < d.S; ++j) keep |= row[j] == id ? 1 : 0;
        aKeep[a] = keep;
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = lane; k < K; k += 64u) {
        uint8_t keep = 0;
        for (uint32_t a = 0; a < A; ++a) {
            if (!aKeep[a]) continue;
            const uint32_t off = d.allele_koff[a0 + a];
            if (k >= off && k < off + 32u && ((d.allele_kmask[a0 + a] >> (k - off)) & 1u)) keep = 1;
        }
        kKeep[k] = keep;
    }
    __builtin_amdgcn_wave_barrier();
}

struct KuShared {
    uint16_t row[4][KU_MAX_PATHS];
    uint8_t aKeep[4][KU_MAX_ALLELES];
    uint8_t kKeep[4][KU_MAX_KMERS];
    uint16_t kNew[4][KU_MAX_KMERS];
};

__global__ __launch_bounds__(256) void ku_count(UpdateDev d) {
    __shared__ KuShared sh;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u, v = blockIdx.x * 4u + w;
    if (v >= d.V) return;
    const uint32_t A = d.allele_off[v + 1] - d.allele_off[v], K = d.kmer_off[v + 1] - d.kmer_off[v];
    if (A > KU_MAX_ALLELES || K > KU_MAX_KMERS) { if (lane == 0) atomicOr(d.err, 1u); return; }
    ku_mark(d, v, lane, sh.row[w], sh.aKeep[w], sh.kKeep[w]);
    for (uint32_t j = lane; j < d.S; j += 64u) d.new_pa[(size_t)v * d.S + j] = sh.row[w][j];
    uint32_t na = 0, nk = 0;
    for (uint32_t a = lane; a < A; a += 64u) na += sh.aKeep[w][a];
    for (uint32_t k = lane; k < K; k += 64u) nk += sh.kKeep[w][k];
    for (int m = 32; m >= 1; m >>= 1) { na += __shfl_xor((int)na, m); nk += __shfl_xor((int)nk, m); }
    if (lane == 0) { d.counts[2 * v] = na; d.counts[2 * v + 1] = nk; }
}

__global__ __launch_bounds__(256) void ku_write(UpdateDev d) {
    __shared__ KuShared sh;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u, v = blockIdx.x * 4u + w;
    if (v >= d.V) return;
    const uint32_t a0 = d.allele_off[v], A = d.allele_off[v + 1] - a0, k0 = d.kmer_off[v], K = d.kmer_off[v + 1] - k0;
    if (A > KU_MAX_ALLELES || K > KU_MAX_KMERS) return;
    ku_mark(d, v, lane, sh.row[w], sh.aKeep[w], sh.kKeep[w]);
    // new index of every kept k-mer = its rank among the kept ones (old order), and the compacted counts
    const uint32_t nk0 = d.new_koff[v];
    uint32_t base = 0;
    for (uint32_t kb = 0; kb < K; kb += 64u) {
        const uint32_t k = kb + lane;
        const bool keep = k < K && sh.kKeep[w][k];
        const unsigned long long m = __ballot(keep);
        const uint32_t rank = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (keep) { sh.kNew[w][k] = (uint16_t)rank; d.new_kcount[nk0 + rank] = d.kmer_count[k0 + k]; }
        base += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    // kept alleles, in their old (= ascending id) order
    const uint32_t na0 = d.new_aoff[v];
    base = 0;
    for (uint32_t ab = 0; ab < A; ab += 64u) {
        const uint32_t a = ab + lane;
        const bool keep = a < A && sh.aKeep[w][a];
        const unsigned long long m = __ballot(keep);
        const uint32_t rank = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (keep) {
            const uint32_t off = d.allele_koff[a0 + a], mask = d.allele_kmask[a0 + a];
            uint32_t first = 0xFFFFFFFFu, bits = 0;
            for (uint32_t b = 0; b < 32u; ++b)   // (bits beyond the variant's k-mers do not exist)
                if (((mask >> b) & 1u) && off + b < K) { const uint32_t n = sh.kNew[w][off + b]; first = n < first ? n : first; }
            if (first != 0xFFFFFFFFu)
                for (uint32_t b = 0; b < 32u; ++b)
                    if (((mask >> b) & 1u) && off + b < K) {
                        const uint32_t sft = sh.kNew[w][off + b] - first;
                        if (sft >= 32u) atomicOr(d.err, 2u); else bits |= 1u << sft;
                    }
            d.new_aid[na0 + rank] = d.allele_id[a0 + a];
            d.new_aflags[na0 + rank] = d.allele_flags[a0 + a];
            d.new_akoff[na0 + rank] = (uint16_t)(first == 0xFFFFFFFFu ? 0u : first);
            d.new_akmask[na0 + rank] = bits;
        }
        base += (uint32_t)__popcll(m);
    }
}

}  // namespace

extern "C" int pg_sampler_then_job(const pg_contig_batch* panels, uint32_t n_contigs, uint32_t size, int add_reference,
                                   double sampling_recombrate, long double sampling_effective_N, uint16_t allele_penalty,
                                   const pg_table* table, const pg_hmm_params* params, int device,
                                   uint32_t* const* sampled_paths, uint32_t* const* best_scores,
                                   pg_job** out_job, char* err, size_t errlen) {
    if (!panels || !table || !params || !out_job || n_contigs == 0) { set_err(err, errlen, "null argument"); return PG_ERR_INVALID; }
    *out_job = nullptr;
    if (size < 1) { set_err(err, errlen, "pg_sampler_then_job: at least one pass"); return PG_ERR_INVALID; }
    const uint32_t S = size + (add_reference ? 1u : 0u);
    if (S > KU_MAX_PATHS) { set_err(err, errlen, "pg_sampler_then_job: at most %u kept paths", KU_MAX_PATHS); return PG_ERR_UNSUPPORTED; }
    SamplerKeep keep;
    int rc = sampler_run_core(panels, n_contigs, size, sampling_recombrate, sampling_effective_N, allele_penalty, device,
                              sampled_paths, best_scores, &keep, err, errlen);
    if (rc != PG_OK) return rc;
    // the rest of the old panel (what the sampler itself did not need on the device) and room for the new one
    struct Stage { size_t koff, kcnt, aflag, akoff, akmask, counts, nkoff, naoff, nkcnt, naid, naflag, nakoff, nakmask, npa; };
    std::vector<Stage> st(n_contigs);
    std::vector<int> keep_of(n_contigs, -1);
    for (size_t j = 0; j < keep.contigs.size(); ++j) keep_of[keep.contigs[j].g] = (int)j;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 255) / 256 * 256; size_t o = off; off += bytes ? bytes : 8; return o; };
    for (uint32_t g = 0; g < n_contigs; ++g) {
        const size_t V = panels[g].n_variants;
        if (V == 0) continue;
        const size_t sumK = panels[g].kmer_off[V], sumA = panels[g].allele_off[V];
        Stage& t = st[g];
        t.koff = take((V + 1) * 4); t.kcnt = take(sumK * 2); t.aflag = take(sumA); t.akoff = take(sumA * 2); t.akmask = take(sumA * 4);
        t.counts = take(V * 8); t.nkoff = take((V + 1) * 4); t.naoff = take((V + 1) * 4);
        t.nkcnt = take(sumK * 2); t.naid = take(sumA * 2); t.naflag = take(sumA); t.nakoff = take(sumA * 2); t.nakmask = take(sumA * 4);
        t.npa = take(V * S * 2);
    }
    const size_t o_err = take(4);
    unsigned char* stage = nullptr;
    std::vector<std::vector<uint32_t>> nkoff(n_contigs), naoff(n_contigs);
    std::vector<pg_contig_batch> nb(n_contigs);
    uint32_t dev_err = 0;
    HIP_TRY(hipSetDevice(device));
    {
        hipError_t he = hipMalloc((void**)&stage, off);
        if (he != hipSuccess) {   // cached arenas of the one-shot call may be what stands in the way
            pg_hmm_release_cache();
            hipSetDevice(device);
            he = hipMalloc((void**)&stage, off);
        }
        if (he != hipSuccess) { set_err(err, errlen, "hipMalloc(%zu bytes) failed: %s", off, hipGetErrorString(he)); rc = PG_ERR_NOMEM; goto done; }
    }
    HIP_TRY(hipMemset(stage + o_err, 0, 4));
    for (int phase = 0; phase < 2; ++phase) {
        for (uint32_t g = 0; g < n_contigs; ++g) {
            const pg_contig_batch& b = panels[g];
            const size_t V = b.n_variants;
            if (V == 0) continue;
            const size_t sumK = b.kmer_off[V], sumA = b.allele_off[V];
            const Stage& t = st[g];
            const SamplerKeep::Contig& kc = keep.contigs[(size_t)keep_of[g]];
            UpdateDev d;
            memset(&d, 0, sizeof(d));
            d.V = (uint32_t)V; d.P = b.n_paths; d.S = S; d.size = size;
            d.kmer_off = (const uint32_t*)(stage + t.koff); d.kmer_count = (const uint16_t*)(stage + t.kcnt);
            d.allele_off = kc.allele_off; d.allele_id = kc.allele_id; d.path_allele = kc.path_allele; d.paths = kc.paths;
            d.allele_flags = stage + t.aflag; d.allele_koff = (const uint16_t*)(stage + t.akoff); d.allele_kmask = (const uint32_t*)(stage + t.akmask);
            d.counts = (uint32_t*)(stage + t.counts); d.new_koff = (const uint32_t*)(stage + t.nkoff); d.new_aoff = (const uint32_t*)(stage + t.naoff);
            d.new_kcount = (uint16_t*)(stage + t.nkcnt); d.new_aid = (uint16_t*)(stage + t.naid); d.new_aflags = stage + t.naflag;
            d.new_akoff = (uint16_t*)(stage + t.nakoff); d.new_akmask = (uint32_t*)(stage + t.nakmask); d.new_pa = (uint16_t*)(stage + t.npa);
            d.err = (uint32_t*)(stage + o_err);
            if (phase == 0) {
                HIP_TRY(hipMemcpyAsync(stage + t.koff, b.kmer_off, (V + 1) * 4, hipMemcpyHostToDevice, nullptr));
                if (sumK) HIP_TRY(hipMemcpyAsync(stage + t.kcnt, b.kmer_count, sumK * 2, hipMemcpyHostToDevice, nullptr));
                HIP_TRY(hipMemcpyAsync(stage + t.aflag, b.allele_flags, sumA, hipMemcpyHostToDevice, nullptr));
                HIP_TRY(hipMemcpyAsync(stage + t.akoff, b.allele_kmer_off, sumA * 2, hipMemcpyHostToDevice, nullptr));
                HIP_TRY(hipMemcpyAsync(stage + t.akmask, b.allele_kmer_mask, sumA * 4, hipMemcpyHostToDevice, nullptr));
                hipLaunchKernelGGL(ku_count, dim3((uint32_t)((V + 3) / 4)), dim3(256), 0, nullptr, d);
                HIP_TRY(hipGetLastError());
            } else {
                hipLaunchKernelGGL(ku_write, dim3((uint32_t)((V + 3) / 4)), dim3(256), 0, nullptr, d);
                HIP_TRY(hipGetLastError());
            }
        }
        HIP_TRY(hipStreamSynchronize(nullptr));
        if (phase == 0) {
            HIP_TRY(hipMemcpy(&dev_err, stage + o_err, 4, hipMemcpyDeviceToHost));
            if (dev_err & 1u) { set_err(err, errlen, "pg_sampler_then_job: a variant has more than %u alleles or %u k-mers", KU_MAX_ALLELES, KU_MAX_KMERS); rc = PG_ERR_UNSUPPORTED; goto done; }
            // the counts -> offsets (host: the job's arena is planned from them), and back
            for (uint32_t g = 0; g < n_contigs; ++g) {
                const size_t V = panels[g].n_variants;
                nkoff[g].assign(V + 1, 0); naoff[g].assign(V + 1, 0);
                if (V == 0) continue;
                std::vector<uint32_t> counts(2 * V);
                HIP_TRY(hipMemcpy(counts.data(), stage + st[g].counts, V * 8, hipMemcpyDeviceToHost));
                for (size_t v = 0; v < V; ++v) { naoff[g][v + 1] = naoff[g][v] + counts[2 * v]; nkoff[g][v + 1] = nkoff[g][v] + counts[2 * v + 1]; }
                HIP_TRY(hipMemcpy(stage + st[g].nkoff, nkoff[g].data(), (V + 1) * 4, hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(stage + st[g].naoff, naoff[g].data(), (V + 1) * 4, hipMemcpyHostToDevice));
            }
        }
    }
    HIP_TRY(hipMemcpy(&dev_err, stage + o_err, 4, hipMemcpyDeviceToHost));
    if (dev_err & 2u) { set_err(err, errlen, "pg_sampler_then_job: an allele's k-mers span more than 32 positions after the update"); rc = PG_ERR_INVALID; goto done; }
    // the job over the updated panel: offsets, positions and coverage from the host, the six big arrays from the device
    for (uint32_t g = 0; g < n_contigs; ++g) {
        const pg_contig_batch& b = panels[g];
        pg_contig_batch& q = nb[g];
        q = b;
        if (b.n_variants == 0) continue;
        const Stage& t = st[g];
        q.n_paths = S;
        q.kmer_off = nkoff[g].data(); q.allele_off = naoff[g].data();
        q.kmer_count = (const uint16_t*)(stage + t.nkcnt); q.allele_id = (const uint16_t*)(stage + t.naid); q.allele_flags = stage + t.naflag;
        q.allele_kmer_off = (const uint16_t*)(stage + t.nakoff); q.allele_kmer_mask = (const uint32_t*)(stage + t.nakmask);
        q.path_allele = (const uint16_t*)(stage + t.npa);
    }
    rc = pg_job_new(device, n_contigs, nb.data(), table, params, out_job, err, errlen);
done:
    if (stage) hipFree(stage);
    if (keep.arena) hipFree(keep.arena);
    return rc;
}

extern "C" int pg_sampler_last_ms(double out3[3], int* kernel) {
    if (!out3) return PG_ERR_INVALID;
    out3[0] = g_last_ms[0]; out3[1] = g_last_ms[1]; out3[2] = g_last_ms[2];
    if (kernel) *kernel = g_last_kernel;
    return PG_OK;
}
