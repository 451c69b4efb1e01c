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
#include "pg_experiments.h"
#define GAS __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 has no address-space-qualified copy)

namespace {

DEVI uint32_t last_bit64(unsigned long long m) { return 63u - (uint32_t)__builtin_clzll(m | 1ull); }
constexpr double kNone = -1.7976931348623157e308;  // below every lo part: a lane that does not take part in a reduction

// ---- DPP steps of the reductions (gfx9 DPP: row_shr inside rows of 16 lanes, row_bcast15 / row_bcast31 across rows)
// x = max(x, the lane's DPP source); lanes without a source keep x.  ZERO: such lanes are fed 0 instead of a copy of
// themselves (two moves less) — only for values >= 0.
template <int CTRL, int ROWMASK, bool ZERO>
DEVI double dpp_max(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int olo = ZERO ? __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, true) : __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
    const int ohi = ZERO ? __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, true) : __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
    return fmax(x, __hiloint2double(ohi, olo));
}
template <int CTRL>
DEVI uint32_t dpp_max_u32(uint32_t x) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
    return o > x ? o : x;
}
// lane 15 of every row of 16 to all lanes of the row (one v_mov_b64_dpp row_newbcast)
DEVI double row_bcast15_f64(double v) {
    return __longlong_as_double(__builtin_amdgcn_update_dpp(__double_as_longlong(v), __double_as_longlong(v), 0x15F, 0xF, 0xF, true));
}
DEVI uint32_t row_bcast15_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x15F, 0xF, 0xF, true); }
DEVI double readlane63_f64(double x) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
// maximum over every row of 16 lanes, in all lanes of the row
template <bool ZERO>
DEVI double row16_max(double x) {
    double r = x;
    r = dpp_max<0x111, 0xF, ZERO>(r);  // row_shr:1,2,4,8: lane 15 holds the row's maximum
    r = dpp_max<0x112, 0xF, ZERO>(r);
    r = dpp_max<0x114, 0xF, ZERO>(r);
    r = dpp_max<0x118, 0xF, ZERO>(r);
    return row_bcast15_f64(r);
}
DEVI uint32_t row16_max_u32(uint32_t x) {
    uint32_t r = x;
    r = dpp_max_u32<0x111>(r);
    r = dpp_max_u32<0x112>(r);
    r = dpp_max_u32<0x114>(r);
    r = dpp_max_u32<0x118>(r);
    return row_bcast15_u32(r);
}
// maximum over the wave, uniform
template <bool ZERO>
DEVI double wave_max(double x) {
    double r = x;
    r = dpp_max<0x111, 0xF, ZERO>(r);
    r = dpp_max<0x112, 0xF, ZERO>(r);
    r = dpp_max<0x114, 0xF, ZERO>(r);
    r = dpp_max<0x118, 0xF, ZERO>(r);
    r = dpp_max<0x142, 0xA, false>(r);  // row_bcast15: lanes 31 / 63 hold the maxima of the two halves
    r = dpp_max<0x143, 0xC, false>(r);  // row_bcast31: lane 63 holds the wave's
    return readlane63_f64(r);
}

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
// (bitwise, not short-circuit: selects, no branches)
DEVI bool dd_gt(dd a, dd b) { return (a.hi > b.hi) | ((a.hi == b.hi) & (a.lo > b.lo)); }
DEVI bool dd_ge(dd a, dd b) { return (a.hi > b.hi) | ((a.hi == b.hi) & (a.lo >= b.lo)); }
DEVI bool dd_eq(dd a, dd b) { return (a.hi == b.hi) & (a.lo == b.lo); }
// the larger of two (value, state index) candidates; of equal values the larger index — what the reference's
// scan with `>=` ends on
struct Cand { dd v; uint32_t i; };
DEVI Cand better(Cand a, Cand b) {
    const bool t = dd_gt(b.v, a.v) | (dd_eq(b.v, a.v) & (b.i >= a.i));
    return {{t ? b.v.hi : a.v.hi, t ? b.v.lo : a.v.lo}, t ? b.i : a.i};
}

// ------------------------------------------------------------------------------------------
//  k_viterbi<K> : the forward recursion of one chain with HP = 16 K padded paths.
//  A row of the state matrix (first path fixed) lies in one DPP row of 16 lanes, K neighbouring second paths per lane;
//  a wave holds 4 rows, the workgroup 16 (K = 1) or 32 rows per pass, R passes cover the matrix.
// ------------------------------------------------------------------------------------------
#ifndef PG_VIT_NW
#define PG_VIT_NW 8
#endif
#ifndef PG_VIT_FASTG
#define PG_VIT_FASTG 1
#endif
#ifndef PG_VIT_FASTR
#define PG_VIT_FASTR 0
#endif
template <int K>
struct VitCfg {
    static constexpr int HP = 16 * K;
    static constexpr int NW = K == 1 ? 4 : PG_VIT_NW;   // waves
    static constexpr int T = 64 * NW;
    static constexpr int RPP = NW * 4;                  // rows per pass of the workgroup
    static constexpr int R = HP / RPP;                  // passes: 1 / 1 / 2
    static constexpr int RB = (PG_REC_ALLELES + HP + 63) & ~63;  // = pg_rec_bytes(HP)
    static constexpr int BC = 32;                       // columns per staged block
    static constexpr int UNITS = BC * RB / 16;          // 16-byte units of a block of column records
    static constexpr int UPT = (UNITS + T - 1) / T;
    static_assert(R * RPP == HP, "rows");
};
template <int K>
struct VitShared {
    alignas(16) unsigned char rec[2][VitCfg<K>::BC * VitCfg<K>::RB];  // column records, BC columns per block, two blocks
    alignas(16) double tq[2][VitCfg<K>::BC * 8];                      // their transition probabilities {t0, t1, t2} as (hi, lo)
    alignas(16) double rmh[2][64], rml[2][64];                        // row maxima of the previous column (by step parity)
    alignas(16) uint32_t rl[2][64];                                   // ... and the last second-path index that holds them
};

template <int K>
__global__ __launch_bounds__((VitCfg<K>::T)) void k_viterbi(const DevContig* __restrict__ contigs) {
    using Cfg = VitCfg<K>;
    constexpr int R = Cfg::R, RB = Cfg::RB, T = Cfg::T, UPT = Cfg::UPT, UNITS = Cfg::UNITS, HP = Cfg::HP;
    constexpr uint32_t BC = Cfg::BC, BSH = 5;  // block of column c: c >> BSH
    static_assert((1u << BSH) == BC, "block size");
    const DevContig& dc = contigs[blockIdx.x];
    if (!dc.vit_back || dc.HP != (uint32_t)HP) return;  // chains of another width: their own launch
    const uint32_t C = *dc.n_cols, H = dc.H;
    if (C == 0) return;  // reference src/hmm.cpp:114
    __shared__ VitShared<K> sh;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t grp = lane >> 4, col0 = (lane & 15u) * K, n = H * H;
    uint32_t row[R];
    bool real[R][K];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        row[r] = (uint32_t)r * Cfg::RPP + wave * 4u + grp;
#pragma unroll
        for (int k = 0; k < K; ++k) real[r][k] = row[r] < H && col0 + k < H;
    }
    const GAS unsigned char* vrec = (const GAS unsigned char*)dc.vrec;
    const GAS uint32_t* colv = (const GAS uint32_t*)dc.col_variant;
    const GAS double* tqg = (const GAS double*)dc.vit_tq;
    const GAS unsigned char* wide = (const GAS unsigned char*)dc.wide;
    GAS uint16_t* back = (GAS uint16_t*)dc.vit_back;

    // ---- block staging: the records of columns [BC b, BC b + BC) are loaded into registers while block b - 1 is
    //      being worked on, and written to LDS one column before the block starts.  A record is found through the column's
    //      variant (col_variant): those indices are fetched ONE BLOCK EARLIER still (round 6: the step used to wait for
    //      UPT dependent pairs of loads, index then record, one pair after the other, at every block boundary).
    u32x4 pre[UPT];
    u32x4 pretq = {0u, 0u, 0u, 0u};
    uint32_t pvar[UPT];  // variants of the columns this thread's units of the next block to issue belong to
#pragma unroll
    for (int u = 0; u < UPT; ++u) { pre[u] = pretq; pvar[u] = 0u; }
    auto issue_variants = [&](uint32_t blk) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const uint32_t unit = tid + (uint32_t)u * T;
            const uint32_t col = blk * BC + unit / (RB / 16);
            if (unit < (uint32_t)UNITS && col < C) pvar[u] = colv[col];
        }
    };
    auto issue = [&](uint32_t blk) {  // (pvar holds block blk's variants)
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const uint32_t unit = tid + (uint32_t)u * T;
            const uint32_t col = blk * BC + unit / (RB / 16), off = unit % (RB / 16);
            if (unit < (uint32_t)UNITS && col < C) pre[u] = *(const GAS u32x4*)(vrec + (size_t)pvar[u] * RB + off * 16u);
        }
        if (tid < BC * 4u) {  // BC columns x 64 bytes
            const uint32_t col = blk * BC + (tid >> 2);
            if (col < C) pretq = *(const GAS u32x4*)(tqg + (size_t)col * 8 + (tid & 3u) * 2u);
        }
        issue_variants(blk + 1u);
    };
    auto commit = [&](uint32_t blk) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const uint32_t unit = tid + (uint32_t)u * T;
            if (unit < (uint32_t)UNITS) *(u32x4*)(sh.rec[blk & 1u] + unit * 16u) = pre[u];
        }
        if (tid < BC * 4u) *(u32x4*)((unsigned char*)sh.tq[blk & 1u] + tid * 16u) = pretq;
    };
    // Emission probability of this lane's states at column c (EmissionProbabilityComputer::get_emission_probability,
    // reference src/emissionprobabilitycomputer.cpp:31-34): table of the column's record, indexed by the local alleles
    // of the two paths; lanes outside the matrix hit the table's zero row.  Two rounds of LDS reads, issued one column
    // ahead and far apart in the step, so that neither is waited for: (1) the alleles, (2) the table entries.
    struct Alleles { uint32_t la[R], lb[K]; };
    struct WideInfo { uint32_t flags, nlocal, widx; };
    auto fetch_alleles = [&](uint32_t c, Alleles& a) {
        const unsigned char* rec = sh.rec[(c >> BSH) & 1u] + (c & (BC - 1u)) * RB;
#pragma unroll
        for (int r = 0; r < R; ++r) a.la[r] = rec[PG_REC_ALLELES + row[r]];
#pragma unroll
        for (int k = 0; k < K; ++k) a.lb[k] = rec[PG_REC_ALLELES + col0 + k];
    };
    auto fetch_e = [&](uint32_t c, const Alleles& a, double (&e)[R][K], WideInfo& w) {
        const unsigned char* rec = sh.rec[(c >> BSH) & 1u] + (c & (BC - 1u)) * RB;
        const double* E = (const double*)(rec + PG_REC_E);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t la = a.la[r] < (uint32_t)PG_AMAX ? a.la[r] : (uint32_t)PG_AMAX;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t lb = a.lb[k] < (uint32_t)PG_AMAX ? a.lb[k] : (uint32_t)PG_AMAX;
                e[r][k] = E[la * PG_ESTRIDE + lb];
            }
        }
        w.flags = rec[PG_REC_FLAGS];
        w.nlocal = rec[PG_REC_NLOCAL];
        w.widx = *(const uint32_t*)(rec + PG_REC_WIDE_IDX);
    };
    // more than PG_AMAX alleles on the selected paths: the table lives in the side buffer (rare).  The loads are
    // waited for right here, so that the common path carries no pending load to its next wait.
    auto fetch_e_wide = [&](const Alleles& a, const WideInfo& w, double (&e)[R][K]) {
        const uint32_t S = w.nlocal + 1u;
        const GAS double* Ew = (const GAS double*)(wide + (size_t)w.widx * 16u);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t la = a.la[r] < S - 1u ? a.la[r] : S - 1u;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t lb = a.lb[k] < S - 1u ? a.lb[k] : S - 1u;
                e[r][k] = Ew[la * S + lb];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" : "+v"(e[r][k]));
    };

    issue_variants(0);
    issue(0);
    commit(0);
    __syncthreads();
    if (C > BC) issue(1);

    dd cur[R][K];       // the previous column (scaled); hi = -1 outside the matrix
    double e[R][K];     // emission probabilities of the column in work
    {
        Alleles a;
        WideInfo w;
        fetch_alleles(0, a);
        fetch_e(0, a, e, w);  // first column: previous_cell = 1 (reference src/hmm.cpp:475-477)
        if (w.flags & PG_REC_FLAG_WIDE) fetch_e_wide(a, w, e);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < K; ++k) cur[r][k] = {real[r][k] ? e[r][k] : -1.0, 0.0};
        if (C > 1u) {
            fetch_alleles(1, a);
            fetch_e(1, a, e, w);
            if (w.flags & PG_REC_FLAG_WIDE) fetch_e_wide(a, w, e);
        }
    }
    VitTimeline tl;
    tl.init();
    for (uint32_t c = 1; c <= C; ++c) {
        tl.template mark<0>(cur[0][0].hi);
        const uint32_t cn = c + 1u;  // the column whose emissions are fetched during this step
        if ((cn & (BC - 1u)) == 0u && cn < C) {
            commit(cn >> BSH);
            __syncthreads();
            if (((cn >> BSH) + 1u) * BC < C) issue((cn >> BSH) + 1u);
        }
        Alleles an;
        fetch_alleles(cn, an);  // (beyond the last column: stale bytes of the LDS block, never used)
        // this step's transition probabilities: read now, used after the exchange
        const double* tq = sh.tq[(c >> BSH) & 1u] + (c & (BC - 1u)) * 8u;
        const dd t0 = {tq[0], tq[1]}, t1 = {tq[2], tq[3]}, t2 = {tq[4], tq[5]};
        __builtin_amdgcn_sched_barrier(0);  // (the reads above are consumed far below: nothing waits for them here)
        const uint32_t par = c & 1u;
        // ---- maxima of the previous column: inside the lane, inside the row of 16 lanes, the rest through LDS
        dd m[R];
        uint32_t ml[R];
        bool rowfast[R];  // (wave-uniform)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dd lm = cur[r][0];
            uint32_t lk = 0;
#pragma unroll
            for (int k = 1; k < K; ++k) {
                const bool t = dd_ge(cur[r][k], lm);  // >=: the last one wins
                lm.hi = t ? cur[r][k].hi : lm.hi;
                lm.lo = t ? cur[r][k].lo : lm.lo;
                lk = t ? (uint32_t)k : lk;
            }
            m[r].hi = row16_max<true>(lm.hi);   // (an all-phantom row gives 0 instead of -1: never read)
            if (r == R - 1) tl.template mark<1>(m[r].hi);
            // The maximum of a row is (hi, lo, last index) in lexicographic order.  Round 6: when no row of the wave has two lanes
            // that hold the largest hi — a ballot and three bit operations per lane — the one lane that holds it IS the row's
            // maximum: it writes its (hi, lo, index) to the exchange itself, and the row's other lanes read lo and index back with
            // the column maxima behind the barrier.  Otherwise (ties at a row's maximum: exact duplicates among the second paths):
            // two more DPP passes, over lo among the holders of hi and over the index among the holders of (hi, lo).
            bool fast = false;
            const bool holder = lm.hi == m[r].hi;
            if (PG_VIT_FASTR && !(kVitExp & 3u)) {
                const unsigned long long hb = __ballot(holder);
                const uint32_t rowbits = (uint32_t)(hb >> (lane & 48u)) & 0xFFFFu;
                fast = __ballot((rowbits & (rowbits - 1u)) != 0u) == 0ull;
            }
            if (fast) {
                if (holder & (row[r] < H)) { sh.rmh[par][row[r]] = lm.hi; sh.rml[par][row[r]] = lm.lo; sh.rl[par][row[r]] = col0 + lk; }
                rowfast[r] = true;
                if (r == R - 1) { tl.template mark<2>(lm.lo); tl.template mark_u<3>(lk); }
            } else {
                m[r].lo = (kVitExp & 1u) ? lm.lo : row16_max<false>(holder ? lm.lo : kNone);
                if (r == R - 1) tl.template mark<2>(m[r].lo);
                ml[r] = (kVitExp & 2u) ? col0 + lk : row16_max_u32((holder & (lm.lo == m[r].lo)) ? col0 + lk : 0u);
                if (r == R - 1) tl.template mark_u<3>(ml[r]);
                if (((lane & 15u) == 0u) & (row[r] < H)) { sh.rmh[par][row[r]] = m[r].hi; sh.rml[par][row[r]] = m[r].lo; sh.rl[par][row[r]] = ml[r]; }
                rowfast[r] = false;
            }
        }
        tl.template mark_u<4>(ml[R - 1]);
        __syncthreads();
        tl.template mark_u<5>(ml[R - 1]);
        dd colm[K];      // column j of a symmetric matrix = row j
        uint32_t coll[K];  // ... its last maximum sits in row rl[j]
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t j = col0 + k < H ? col0 + k : 0u;
            colm[k] = {col0 + k < H ? sh.rmh[par][j] : -1.0, sh.rml[par][j]};
            coll[k] = sh.rl[par][j];
        }
        const uint32_t pl = lane < H ? lane : 0u;
        const dd x = {lane < H ? sh.rmh[par][pl] : -1.0, sh.rml[par][pl]};
        const uint32_t xl = sh.rl[par][pl];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (rowfast[r]) {  // (rows outside the matrix: values never used)
                const uint32_t rr = row[r] < H ? row[r] : 0u;
                m[r].lo = sh.rml[par][rr];
                ml[r] = sh.rl[par][rr];
            }
        double en[R][K];
        WideInfo wn;
        if (!(kVitExp & 16u)) fetch_e(cn, an, en, wn);  // (second round of the emission prefetch, in the shadow of the reads above)
        else {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < K; ++k) en[r][k] = e[r][k];
            wn.flags = 0; wn.nlocal = 0; wn.widx = 0;
        }
        tl.template mark<6>(x.hi);
        dd gmax;
        gmax.hi = wave_max<true>(x.hi);
        tl.template mark<7>(gmax.hi);
        // (hi, lo) of the column's maximum and the last row that holds it.  The matrix is symmetric, so at least two rows hold the
        // largest hi (unless it sits on the diagonal) — with the SAME lo.  Round 6: if every row that holds the largest hi has
        // the lo of the last one of them — one ballot, two lane reads, one compare —, that row is the answer; otherwise a second
        // reduction pass over lo among the holders.
        const unsigned long long gh = __ballot(x.hi == gmax.hi);
        const uint32_t glast = (uint32_t)__builtin_amdgcn_readfirstlane((int)last_bit64(gh));
        gmax.lo = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x.lo), (int)glast), __builtin_amdgcn_readlane(__double2loint(x.lo), (int)glast));
        uint32_t ga = glast;
        if (!PG_VIT_FASTG || (kVitExp & 4u) == 0u) {
            if (!PG_VIT_FASTG || (__ballot((x.hi == gmax.hi) & (x.lo != gmax.lo)) != 0ull)) {
                gmax.lo = wave_max<false>(x.hi == gmax.hi ? x.lo : kNone);
                ga = last_bit64(__ballot((x.hi == gmax.hi) & (x.lo == gmax.lo)));  // last row that holds the column's maximum
            }
        }
        uint32_t gidx = ga * H + (uint32_t)__builtin_amdgcn_readlane((int)xl, (int)__builtin_amdgcn_readfirstlane((int)ga));
        if (!(gmax.hi > 0.0)) {
            // the previous column was all 0: the reference sets it to the constant 1/n (src/hmm.cpp:484-491)
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < K; ++k) cur[r][k] = {real[r][k] ? 1.0 : -1.0, 0.0};
                m[r] = {1.0, 0.0}; ml[r] = H - 1u;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) { colm[k] = {col0 + k < H ? 1.0 : -1.0, 0.0}; coll[k] = H - 1u; }
            gmax = {1.0, 0.0}; gidx = n - 1u;
        }
        tl.template mark_u<8>(gidx);
        if (c == C) {  // best state of the last column: the last maximum (reference src/hmm.cpp:131-141)
            if (tid == 0) *dc.vit_best = gidx;
            break;
        }
        int ex;
        (void)frexp(gmax.hi, &ex);
        const double scale = ldexp(1.0, -ex);  // exact: the new column's largest entry lands below 1
        const Cand g = {dd_mul(gmax, t2), gidx};
        Cand cc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) cc[k] = {dd_mul(colm[k], t1), coll[k] * H + col0 + k};
        GAS uint16_t* bk = back + (size_t)c * H * HP;
        tl.template mark<9>(cc[K - 1].v.hi);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const Cand rg = better({dd_mul(m[r], t1), row[r] * H + ml[r]}, g);  // row or anywhere: the same for the whole row
            uint32_t idx[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                Cand b = {dd_mul(cur[r][k], t0), row[r] * H + col0 + k};
                b = better(b, rg);
                b = better(b, cc[k]);
                if (b.v.hi == 0.0) b.i = n - 1u;  // every product is 0: the reference's scan ends on the last state
                idx[k] = b.i;
                const dd nv = dd_mul_d(b.v, e[r][k]);
                cur[r][k] = {real[r][k] ? nv.hi * scale : -1.0, real[r][k] ? nv.lo * scale : 0.0};
            }
            // K backpointers of one row, neighbouring second paths: one aligned store (row stride HP)
            if (row[r] < H && col0 < H && !(kVitExp & 8u)) {
                GAS uint16_t* o = bk + row[r] * HP + col0;
                if (K == 1) o[0] = (uint16_t)idx[0];
                else if (K == 2) *(GAS uint32_t*)o = idx[0] | (idx[K > 1 ? 1 : 0] << 16);
                else {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 w = {idx[0] | (idx[K > 1 ? 1 : 0] << 16), idx[K > 2 ? 2 : 0] | (idx[K > 3 ? 3 : 0] << 16)};
                    *(GAS u32x2*)o = w;
                }
            }
        }
        tl.template mark<10>(cur[R - 1][K - 1].hi);
        if ((wn.flags & PG_REC_FLAG_WIDE) && cn < C) fetch_e_wide(an, wn, en);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < K; ++k) e[r][k] = en[r][k];
        tl.template mark<11>(e[R - 1][K - 1]);
        tl.template fold<11>();
    }
    if (kVitTimeline && tid == 0) tl.write(dc.prof);
}

// ------------------------------------------------------------------------------------------
//  k_vit_backtrack : one wave per chain (reference src/hmm.cpp:144-172)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_vit_backtrack(const DevContig* __restrict__ contigs) {
    const DevContig& dc = contigs[blockIdx.x];
    if (!dc.vit_back) return;
    const uint32_t C = *dc.n_cols, H = dc.H, HP = dc.HP, n = H * H;
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
        if (valid) b = back[((size_t)cc * H + s / H) * HP + s % H];  // (rows of the backtrace are HP entries apart)
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
    if (hp_bits & 1u) k_viterbi<1><<<n, VitCfg<1>::T, 0, s>>>(d_contigs);
    if (hp_bits & 2u) k_viterbi<2><<<n, VitCfg<2>::T, 0, s>>>(d_contigs);
    if (hp_bits & 4u) k_viterbi<4><<<n, VitCfg<4>::T, 0, s>>>(d_contigs);
    k_vit_backtrack<<<n, 64, 0, s>>>(d_contigs);
}
