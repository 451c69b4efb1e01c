// pg_viterbi.hip — Viterbi phasing on the MI355X: HMM::compute_viterbi_path / compute_viterbi_column
// (reference src/hmm.cpp:112-173, :408-511; `run_phasing`), behind pg_hmm_genotype_contig / pg_job_run.
//
// The reference scans, for every state (i, j) of a column, all H^2 states of the previous column:
//     cell(i,j) = max_{(k,l)} prev(k,l) t(|{k != i}| + |{l != j}|)  * e(a_i, a_j),      column := column / sum(column)
// with `>=` in the scan (the LAST maximum wins, :468) and t = {p^2, pq, q^2} (p >= q).  Because t0 >= t1 >= t2 the
// maximum is one of four candidates — t0 prev(i,j), t1 max of row i, t1 max of column j, t2 max of the column as a
// whole — and the index the scan ends on is the largest index among the candidates of equal value (the last state of
// all when every product is 0).  So a column costs O(H^2) instead of O(H^4), the recursion is sequential over the
// columns, and the only parallelism is over the H^2 states of a column (one workgroup per chain) and over chains.
//
// Work of one column step (k_viterbi): row maxima by DPP reductions inside a wave (a row of states = L = 16 / 32 / 64
// neighbouring lanes), ONE LDS exchange + ONE workgroup barrier for the column / global maxima (the previous column
// is a symmetric matrix — emission table, update and start are symmetric — so the maxima of column j are those of
// row j), four compare-selects per state, one 2-byte backpointer store per state.  Nothing is loaded on the critical
// path: the column records (emission table + allele of every path, written by k_prep) and the transition
// probabilities are staged 64 columns at a time into LDS, one block ahead.
//
// Scaling.  The reference divides every column by its sum.  A uniform scale changes no comparison, so columns are
// instead multiplied by the exact power of two that brings their maximum into [1/2, 1): no sum, no division, and
// every comparison is the one exact arithmetic would make on the unnormalised values.  The reference's uniform
// fall-back (sum == 0, :484-491) is "all entries 0" here: the next step then sees a constant column.
//
// Backtrace (k_vit_backtrack): one wave per chain, 64 columns per step — a ballot finds the next column at which the
// best path leaves its state (recombinations are rare), so the walk takes about C/64 dependent loads, not C.
//
// Precision.  The decisions of the reference hang on differences fp64 cannot hold: once exp(-d/H) drops below 1e-16 of
// q (distance / H > 37: every default-constructed HMM with a handful of paths), "stay" beats "switch" by a relative
// 1e-18 that the reference's 80-bit products resolve and fp64 ties.  So the column is kept in DOUBLE-DOUBLE (hi + lo,
// 106 bits: every comparison is the one exact arithmetic makes), and the transition probabilities {p^2, pq, q^2} come
// from the host, formed in long double exactly as the reference forms them (pg_shim.cpp: viterbi_transitions) and
// shipped as exact (hi, lo) pairs — including the reference's own p == q once exp(-d/H) < 2^-64 q.  Emission
// probabilities are the fp64 products of k_prep: one rounding per allele pair, the same for every state that
// carries the pair.  What is left are decisions the reference itself takes on its rounding noise (5e-20 relative).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_device.h"
#include "pg_devmath.h"

#define DEVI __device__ __forceinline__
#define GAS __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 has no address-space-qualified copy)

namespace {

template <int CTRL, int ROWMASK>
DEVI double dpp_max(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    // lanes without a source (row edges, rows outside ROWMASK) keep their own value
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
    return fmax(x, __hiloint2double(ohi, olo));
}
template <int LANE>
DEVI double readlane_f64(double x) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), LANE), __builtin_amdgcn_readlane(__double2loint(x), LANE));
}
DEVI uint32_t last_bit32(uint32_t m) { return 31u - (uint32_t)__builtin_clz(m | 1u); }
DEVI uint32_t last_bit64(unsigned long long m) { return 63u - (uint32_t)__builtin_clzll(m | 1ull); }

// ---- double-double: value = hi + lo, |lo| <= ulp(hi) / 2 (so the order of two values is the order of (hi, lo))
struct dd { double hi, lo; };
DEVI dd dd_mul(dd a, dd b) {
    const double p = a.hi * b.hi;
    double e = fma(a.hi, b.hi, -p);
    e = fma(a.hi, b.lo, e);
    e = fma(a.lo, b.hi, e);
    const double s = p + e;
    return {s, e - (s - p)};
}
DEVI dd dd_mul_d(dd a, double b) {
    const double p = a.hi * b;
    double e = fma(a.hi, b, -p);
    e = fma(a.lo, b, e);
    const double s = p + e;
    return {s, e - (s - p)};
}
DEVI bool dd_gt(dd a, dd b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }
DEVI bool dd_eq(dd a, dd b) { return a.hi == b.hi && a.lo == b.lo; }

// maximum of x over every group of L neighbouring lanes (values are >= 0, or -1 / -inf on lanes that do not take part)
template <int L>
DEVI double group_max(double x, uint32_t grp) {
    double r = x;
    r = dpp_max<0x111, 0xF>(r);  // row_shr:1,2,4,8: lane 15 of every row of 16 holds the row's maximum
    r = dpp_max<0x112, 0xF>(r);
    r = dpp_max<0x114, 0xF>(r);
    r = dpp_max<0x118, 0xF>(r);
    if (L >= 32) r = dpp_max<0x142, 0xA>(r);  // row_bcast15: lanes 31 / 63 hold the maxima of the two halves
    if (L == 64) r = dpp_max<0x143, 0xC>(r);  // row_bcast31: lane 63 holds the wave's
    if (L == 64) return readlane_f64<63>(r);
    if (L == 32) {
        const double m0 = readlane_f64<31>(r), m1 = readlane_f64<63>(r);
        return grp ? m1 : m0;
    }
    const double m0 = readlane_f64<15>(r), m1 = readlane_f64<31>(r), m2 = readlane_f64<47>(r), m3 = readlane_f64<63>(r);
    return grp == 0 ? m0 : (grp == 1 ? m1 : (grp == 2 ? m2 : m3));
}
DEVI double wave_max(double x) { return group_max<64>(x, 0u); }
// the LAST lane of this lane's group for which `hit` holds
template <int L>
DEVI uint32_t group_last(bool hit, uint32_t grp) {
    const unsigned long long b = __ballot(hit);
    if (L == 64) return last_bit64(b);
    if (L == 32) {
        const uint32_t l0 = last_bit32((uint32_t)b), l1 = last_bit32((uint32_t)(b >> 32));
        return grp ? l1 : l0;
    }
    const uint32_t l0 = last_bit32((uint32_t)b & 0xFFFFu), l1 = last_bit32((uint32_t)(b >> 16) & 0xFFFFu),
                   l2 = last_bit32((uint32_t)(b >> 32) & 0xFFFFu), l3 = last_bit32((uint32_t)(b >> 48) & 0xFFFFu);
    return grp == 0 ? l0 : (grp == 1 ? l1 : (grp == 2 ? l2 : l3));
}
// maximum of the double-double x over every group of L lanes and the last lane that holds it: the largest hi, then
// the largest lo among the lanes with that hi
template <int L>
DEVI void group_max_last(dd x, uint32_t grp, dd& m, uint32_t& last) {
    m.hi = group_max<L>(x.hi, grp);
    m.lo = group_max<L>(x.hi == m.hi ? x.lo : -__builtin_inf(), grp);
    last = group_last<L>(x.hi == m.hi && x.lo == m.lo, grp);
}

// ------------------------------------------------------------------------------------------
//  k_viterbi<L> : the forward recursion of one chain with HP = L padded paths
// ------------------------------------------------------------------------------------------
template <int L>
struct VitCfg {
    static constexpr int NW = L == 16 ? 4 : 8;          // waves
    static constexpr int T = 64 * NW;
    static constexpr int G = 64 / L;                    // rows of states per wave and pass
    static constexpr int RPP = NW * G;                  // rows per pass of the workgroup
    static constexpr int SPL = (L + RPP - 1) / RPP;     // states per lane (1 / 2 / 8)
    static constexpr int RB = (PG_REC_ALLELES + L + 63) & ~63;  // = pg_rec_bytes(L)
    static constexpr int BC = 32;                       // columns per staged block
    static constexpr int UNITS = BC * RB / 16;          // 16-byte units of a block of column records
    static constexpr int UPT = (UNITS + T - 1) / T;
};
template <int L>
struct VitShared {
    alignas(16) unsigned char rec[2][VitCfg<L>::BC * VitCfg<L>::RB];  // column records, BC columns per block, two blocks
    alignas(16) double tq[2][VitCfg<L>::BC * 8];                      // their transition probabilities {t0, t1, t2} as (hi, lo)
    double rmh[2][64], rml[2][64];                         // row maxima of the previous column (by step parity)
    uint32_t rl[2][64];                                    // ... and the last second-path index that holds them
};

template <int L>
__global__ __launch_bounds__((VitCfg<L>::T)) void k_viterbi(const DevContig* __restrict__ contigs) {
    using Cfg = VitCfg<L>;
    constexpr int SPL = Cfg::SPL, RB = Cfg::RB, T = Cfg::T, UPT = Cfg::UPT, UNITS = Cfg::UNITS;
    constexpr uint32_t BC = Cfg::BC, BSH = 5;  // block of column c: c >> BSH
    static_assert((1u << BSH) == BC, "block size");
    const DevContig& dc = contigs[blockIdx.x];
    if (!dc.vit_back || dc.HP != (uint32_t)L) return;  // chains of another width: their own launch
    const uint32_t C = *dc.n_cols, H = dc.H;
    if (C == 0) return;  // reference src/hmm.cpp:114
    __shared__ VitShared<L> sh;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t grp = lane / L, p2 = lane % L, n = H * H;
    uint32_t row[SPL];
    bool real[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) { row[s] = (uint32_t)s * Cfg::RPP + wave * Cfg::G + grp; real[s] = row[s] < H && p2 < H; }
    const GAS unsigned char* vrec = (const GAS unsigned char*)dc.vrec;
    const GAS uint32_t* colv = (const GAS uint32_t*)dc.col_variant;
    const GAS double* tqg = (const GAS double*)dc.vit_tq;
    const GAS unsigned char* wide = (const GAS unsigned char*)dc.wide;
    GAS uint16_t* back = (GAS uint16_t*)dc.vit_back;

    // ---- block staging: the records of columns [BC b, BC b + BC) are loaded into registers while block b - 1 is
    //      being worked on, and written to LDS at the block boundary
    u32x4 pre[UPT];
    u32x4 pretq = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < UPT; ++u) pre[u] = pretq;
    auto issue = [&](uint32_t blk) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const uint32_t unit = tid + (uint32_t)u * T;
            const uint32_t col = blk * BC + unit / (RB / 16), off = unit % (RB / 16);
            if (unit < (uint32_t)UNITS && col < C) pre[u] = *(const GAS u32x4*)(vrec + (size_t)colv[col] * RB + off * 16u);
        }
        if (tid < BC * 4u) {  // BC columns x 64 bytes
            const uint32_t col = blk * BC + (tid >> 2);
            if (col < C) pretq = *(const GAS u32x4*)(tqg + (size_t)col * 8 + (tid & 3u) * 2u);
        }
    };
    auto commit = [&](uint32_t blk) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const uint32_t unit = tid + (uint32_t)u * T;
            if (unit < (uint32_t)UNITS) *(u32x4*)(sh.rec[blk & 1u] + unit * 16u) = pre[u];
        }
        if (tid < BC * 4u) *(u32x4*)((unsigned char*)sh.tq[blk & 1u] + tid * 16u) = pretq;
    };
    // emission probability of this lane's states at column c (EmissionProbabilityComputer::get_emission_probability,
    // reference src/emissionprobabilitycomputer.cpp:31-34): table of the column's record, indexed by the local
    // alleles of the two paths; lanes outside the matrix hit the table's zero row
    auto fetch_e = [&](uint32_t c, double (&e)[SPL]) {
        const unsigned char* rec = sh.rec[(c >> BSH) & 1u] + (c & (BC - 1u)) * RB;
        const uint32_t flags = rec[PG_REC_FLAGS];
        uint32_t lb = rec[PG_REC_ALLELES + p2];
        if (!(flags & PG_REC_FLAG_WIDE)) {
            lb = lb < (uint32_t)PG_AMAX ? lb : (uint32_t)PG_AMAX;
            const double* E = (const double*)(rec + PG_REC_E);
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                uint32_t la = rec[PG_REC_ALLELES + row[s]];
                la = la < (uint32_t)PG_AMAX ? la : (uint32_t)PG_AMAX;
                e[s] = E[la * PG_ESTRIDE + lb];
            }
        } else {  // more than PG_AMAX alleles on the selected paths: the table lives in the side buffer (rare)
            const uint32_t S = (uint32_t)rec[PG_REC_NLOCAL] + 1u;
            const GAS double* Ew = (const GAS double*)(wide + (size_t)(*(const uint32_t*)(rec + PG_REC_WIDE_IDX)) * 16u);
            lb = lb < S - 1u ? lb : S - 1u;
#pragma unroll
            for (int s = 0; s < SPL; ++s) {
                uint32_t la = rec[PG_REC_ALLELES + row[s]];
                la = la < S - 1u ? la : S - 1u;
                e[s] = Ew[la * S + lb];
            }
        }
    };

    issue(0);
    commit(0);
    __syncthreads();
    if (C > BC) issue(1);

    dd cur[SPL];  // the previous column (scaled); hi = -1 outside the matrix
    {
        double e[SPL];
        fetch_e(0, e);  // first column: previous_cell = 1 (reference src/hmm.cpp:475-477)
#pragma unroll
        for (int s = 0; s < SPL; ++s) cur[s] = {real[s] ? e[s] : -1.0, 0.0};
    }
    for (uint32_t c = 1; c <= C; ++c) {
        if ((c & (BC - 1u)) == 0u && c < C) {
            commit(c >> BSH);
            __syncthreads();
            if (((c >> BSH) + 1u) * BC < C) issue((c >> BSH) + 1u);
        }
        double e[SPL];
        if (c < C) fetch_e(c, e);
        const uint32_t par = c & 1u;
        // ---- maxima of the previous column: rows inside the wave, the rest through LDS
        dd m[SPL];
        uint32_t ml[SPL];
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
            group_max_last<L>(cur[s], grp, m[s], ml[s]);
            if (p2 == 0u && row[s] < H) { sh.rmh[par][row[s]] = m[s].hi; sh.rml[par][row[s]] = m[s].lo; sh.rl[par][row[s]] = ml[s]; }
        }
        __syncthreads();
        const uint32_t pc = p2 < H ? p2 : 0u, pl = lane < H ? lane : 0u;
        dd colm = {p2 < H ? sh.rmh[par][pc] : -1.0, sh.rml[par][pc]};  // column p2 of a symmetric matrix = row p2
        uint32_t coll = sh.rl[par][pc];                                // ... its last maximum sits in row rl[p2]
        const dd x = {lane < H ? sh.rmh[par][pl] : -1.0, sh.rml[par][pl]};
        const uint32_t xl = sh.rl[par][pl];
        dd gmax;
        gmax.hi = wave_max(x.hi);
        gmax.lo = wave_max(x.hi == gmax.hi ? x.lo : -__builtin_inf());
        const uint32_t ga = last_bit64(__ballot(x.hi == gmax.hi && x.lo == gmax.lo));  // last row that holds the column's maximum
        uint32_t gidx = ga * H + (uint32_t)__builtin_amdgcn_readlane((int)xl, (int)__builtin_amdgcn_readfirstlane((int)ga));
        if (!(gmax.hi > 0.0)) {
            // the previous column was all 0: the reference sets it to the constant 1/n (src/hmm.cpp:484-491)
#pragma unroll
            for (int s = 0; s < SPL; ++s) { cur[s] = {real[s] ? 1.0 : -1.0, 0.0}; m[s] = {1.0, 0.0}; ml[s] = H - 1u; }
            colm = {p2 < H ? 1.0 : -1.0, 0.0}; coll = H - 1u;
            gmax = {1.0, 0.0}; gidx = n - 1u;
        }
        if (c == C) {  // best state of the last column: the last maximum (reference src/hmm.cpp:131-141)
            if (tid == 0) *dc.vit_best = gidx;
            break;
        }
        const double* tq = sh.tq[(c >> BSH) & 1u] + (c & (BC - 1u)) * 8u;
        const dd t0 = {tq[0], tq[1]}, t1 = {tq[2], tq[3]}, t2 = {tq[4], tq[5]};
        int ex;
        (void)frexp(gmax.hi, &ex);
        const double scale = ldexp(1.0, -ex);  // exact: the new column's largest entry lands below 1
        const dd gv = dd_mul(gmax, t2), cv = dd_mul(colm, t1);
        const uint32_t ci = coll * H + p2;
        GAS uint16_t* bk = back + (size_t)c * n;
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
            const uint32_t i = row[s] * H + p2;
            dd v = dd_mul(cur[s], t0);
            uint32_t idx = i;
            auto take = [&](dd w, uint32_t wi) {
                const bool b = dd_gt(w, v) || (dd_eq(w, v) && wi >= idx);
                v.hi = b ? w.hi : v.hi;
                v.lo = b ? w.lo : v.lo;
                idx = b ? wi : idx;
            };
            take(dd_mul(m[s], t1), row[s] * H + ml[s]);
            take(cv, ci);
            take(gv, gidx);
            if (v.hi == 0.0) idx = n - 1u;  // every product is 0: the reference's scan ends on the last state
            if (real[s]) bk[i] = (uint16_t)idx;
            const dd nv = dd_mul_d(v, e[s]);
            cur[s] = {real[s] ? nv.hi * scale : -1.0, real[s] ? nv.lo * scale : 0.0};
        }
    }
}

// ------------------------------------------------------------------------------------------
//  k_vit_backtrack : one wave per chain (reference src/hmm.cpp:144-172)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_vit_backtrack(const DevContig* __restrict__ contigs) {
    const DevContig& dc = contigs[blockIdx.x];
    if (!dc.vit_back) return;
    const uint32_t C = *dc.n_cols, H = dc.H, n = H * H;
    if (C == 0) return;
    const uint32_t lane = threadIdx.x;
    const GAS uint16_t* back = (const GAS uint16_t*)dc.vit_back;
    uint32_t s = *dc.vit_best;
    if (s >= n) s = n - 1u;
    int c = (int)C - 1;
    while (true) {
        // lane k looks at column c - k: did the path through state s arrive there from state s?
        const int cc = c - (int)lane;
        const bool valid = cc >= 1;
        uint32_t b = 0xFFFFFFFFu;
        if (valid) b = back[(size_t)cc * n + s];
        const unsigned long long stay = __ballot(valid && b == s);
        const uint32_t r = stay == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~stay);
        // columns c, c-1, ..., c-r are in state s
        if (lane <= r && cc >= 0) {
            const uint32_t v = dc.col_variant[cc];
            dc.hap1[v] = dc.path_allele[(size_t)v * H + s / H];  // ColumnIndexer::get_path_ids_at + get_allele (:146-148)
            dc.hap2[v] = dc.path_allele[(size_t)v * H + s % H];
        }
        if (r == 64u) { c -= 64; continue; }
        const int cr = c - (int)r;
        if (cr <= 0) break;
        s = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)r);
        if (s >= n) s = n - 1u;
        c = cr - 1;
    }
}

}  // namespace

// hp_bits: 1 / 2 / 4 = the job has phasing chains with 16 / 32 / 64 padded paths
extern "C" void pgk_launch_viterbi(const DevContig* d_contigs, uint32_t n, uint32_t max_v, uint32_t hp_bits, hipStream_t s) {
    if (n == 0 || max_v == 0) return;
    if (hp_bits & 1u) k_viterbi<16><<<n, VitCfg<16>::T, 0, s>>>(d_contigs);
    if (hp_bits & 2u) k_viterbi<32><<<n, VitCfg<32>::T, 0, s>>>(d_contigs);
    if (hp_bits & 4u) k_viterbi<64><<<n, VitCfg<64>::T, 0, s>>>(d_contigs);
    k_vit_backtrack<<<n, 64, 0, s>>>(d_contigs);
}
