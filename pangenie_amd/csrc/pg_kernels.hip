// pg_kernels.hip — hand-written CDNA4 (gfx950) kernels of the PanGenie genotyping hot path.
//
//   k_prep      EmissionProbabilityComputer + ColumnIndexer flags      (one wave / variant)
//               reference src/emissionprobabilitycomputer.cpp:9-53, src/columnindexer.cpp:12-31,
//               src/probabilitytable.cpp:47-85 (on-the-fly entries), src/copynumber.cpp:22-41
//   k_compact   kept variants -> column list                           (one workgroup / contig)
//   k_records   column records + Li-Stephens constants on device       (one wave / column)
//               reference src/transitionprobabilitycomputer.cpp:8-19
//   k_sweep     forward / backward half-chains, meet in the middle    (two workgroups / chain)
//               reference src/hmm.cpp:76-110, 175-273, 275-405
//               <..., 1> first halves, columns stored; <..., 2> second halves with the posterior
//               partials formed in the sweep (fused mode); <..., 3> second halves as store-only
//               chunks (chunked mode)
//   k_bins      posterior partials -> genotype bins (fused mode)       reference src/hmm.cpp:364-368
//   k_post      stored column pairs of a chunk -> genotype bins (chunked mode; idle CUs)
//
// No MFMA: the transition operator is rank-structured (A = r I + q 1 1^T), so a column
// update is elementwise + row/column sums.  The recursion is bound by HBM (16 H^2 bytes
// per column: one write + one read of the forward column) once enough chains run; a single
// chain is bound by the per-column latency of one workgroup.
//
// Thread mapping of the chain kernels (HP = padded #paths, R rows per thread, T = HP*HP/R):
//   j  = tid % HP            column (second path) owned by the thread
//   rg = tid / HP            row group; the thread holds rows i = rg*R .. rg*R+R-1 of column j
// For HP >= 64 a wave spans 64 columns of ONE row group, so everything indexed by the row
// (allele of path i, u_i) is wave-uniform and lives in SGPRs; column sums are in-lane adds
// plus one LDS exchange per column (one LDS-only workgroup barrier per column).  Columns are
// symmetric, so row sums == column sums and only column sums are computed.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "pg_device.h"
#include "pg_devmath.h"

#define DEVI __device__ __forceinline__
typedef double v2f64 __attribute__((ext_vector_type(2)));  // native vector type (address-space qualifiable)
#include "pg_experiments.h"   // masks of the timing experiments: all zero in the product build

// ------------------------------------------------------------------------------------------
//  wave-level helpers (wave = 64 lanes)
// ------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK, bool BOUND>
DEVI double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, BOUND);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, BOUND);
    return __hiloint2double(hi, lo);
}

DEVI double readlane_f64(double v, int src_lane /*uniform*/) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, result broadcast to every lane.  4 row_shr steps inside the
// 16-lane DPP rows, then row_bcast:15 / row_bcast:31 across rows (gfx9 DPP).
DEVI double wave_sum(double v) {
    v += dpp_f64<0x111, 0xF, true>(v);   // row_shr:1
    v += dpp_f64<0x112, 0xF, true>(v);   // row_shr:2
    v += dpp_f64<0x114, 0xF, true>(v);   // row_shr:4
    v += dpp_f64<0x118, 0xF, true>(v);   // row_shr:8
    v += dpp_f64<0x142, 0xA, false>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_f64<0x143, 0xC, false>(v);  // row_bcast:31 -> rows 2,3
    return readlane_f64(v, 63);
}

// Sum over the first N lanes only (N = 16: one DPP row; N = 32: two rows), broadcast to every lane.
template <int N>
DEVI double lanes_sum(double v) {
    static_assert(N == 16 || N == 32 || N == 64, "DPP row multiples");
    if constexpr (N == 64) return wave_sum(v);
    v += dpp_f64<0x111, 0xF, true>(v);
    v += dpp_f64<0x112, 0xF, true>(v);
    v += dpp_f64<0x114, 0xF, true>(v);
    v += dpp_f64<0x118, 0xF, true>(v);
    if constexpr (N == 32) v += dpp_f64<0x142, 0xA, false>(v);
    return readlane_f64(v, N - 1);
}

DEVI void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// wave-level sync for data exchanged through LDS only: waits for the LDS queue, not for outstanding global loads and
// stores (wave_sync() drains vmcnt too — in k_prep that put every global round trip of a variant in series)
DEVI void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
}

DEVI int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

DEVI uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= (uint32_t)__shfl_xor((int)v, off);
    return v;
}

// ------------------------------------------------------------------------------------------
//  ProbabilityTable lookups as (mantissa, exponent) pairs
// ------------------------------------------------------------------------------------------
// Smallest exponent the reference's 80-bit long double can hold (denormal minimum 2^-16445):
// anything below is an exact 0 there, and exact zeros matter (all_zeros rule, uniform
// fallbacks), so the wider (mantissa, exponent) representation flushes at the same point.
#define PG_LD_MIN_EXP (-16444)

DEVI void from_log2(double L, double& m, int& e) {
    if (L == -INFINITY || L < (double)(PG_LD_MIN_EXP - 1)) { m = 0.0; e = 0; return; }
    if (!(L == L) || L == INFINITY) { m = L; e = 0; return; }
    double fl = floor(L);
    e = (int)fl + 1;
    m = exp2(L - fl - 1.0);  // [0.5, 1)
}

DEVI void split(double p, double& m, int& e) {
    if (p == 0.0 || !(p == p) || isinf(p)) { m = p; e = 0; return; }
    int ee;
    m = frexp(p, &ee);
    e = ee;
}

// compute_probability on the fly (reference src/probabilitytable.cpp:55-65,75-85) in fp64,
// range-safe through log2-domain evaluation.
__device__ __noinline__ void cn_on_the_fly(double reg, uint32_t cov, uint32_t count, double m[3], int e[3]) {
    const double err = cov < 10 ? 0.99 : (cov < 20 ? 0.95 : (cov < 40 ? 0.9 : 0.8));
    const double LOG2E = 1.4426950408889634074;
    const double cnt = (double)count;
    double L0 = cnt * log2(1.0 - err) + log2(err);  // geometric: (1-p)^count * p
    double lg = lgamma(cnt + 1.0);                  // sum_{i<=count} log(i)
    double mean1 = (double)cov * 0.5, mean2 = (double)cov;
    double L1 = (-mean1 + cnt * log(mean1) - lg) * LOG2E;
    double L2 = (-mean2 + cnt * log(mean2) - lg) * LOG2E;
    from_log2(L0, m[0], e[0]);
    from_log2(L1, m[1], e[1]);
    from_log2(L2, m[2], e[2]);
    if (reg > 0) {  // CopyNumber(cn0,cn1,cn2,reg), reference src/copynumber.cpp:22-41
        double p0 = ldexp(m[0], e[0]), p1 = ldexp(m[1], e[1]), p2 = ldexp(m[2], e[2]);
        double sum = p0 + p1 + p2 + 3.0 * reg;
        double P0 = (p0 + reg) / sum, P1 = (p1 + reg) / sum;
        double P2 = 1.0 - P0 - P1;
        split(P0, m[0], e[0]);
        split(P1, m[1], e[1]);
        split(P2, m[2], e[2]);
    }
}

DEVI void cn_lookup(const DevTable& t, uint32_t cov, uint32_t count, double m[3], int e[3]) {
    if (cov >= t.cov_min && cov < t.cov_max && count < t.count_max) {
        size_t idx = ((size_t)(cov - t.cov_min) * t.count_max + count) * 3;
        m[0] = t.mant[idx]; m[1] = t.mant[idx + 1]; m[2] = t.mant[idx + 2];
        e[0] = t.expo[idx]; e[1] = t.expo[idx + 1]; e[2] = t.expo[idx + 2];
    } else {
        cn_on_the_fly(t.reg, cov, count, m, e);
    }
}

// scale * (a + b [+ c]) for (mantissa, exponent) operands, range-safe
DEVI void mix2(double ma, int ea, double mb, int eb, double scale, double& m, int& e) {
    int emax = ea > eb ? ea : eb;
    if (ma == 0.0) emax = eb;
    if (mb == 0.0) emax = ea;
    double s = (ldexp(ma, ea - emax) + ldexp(mb, eb - emax)) * scale;
    split(s, m, e);
    e += emax;
}
DEVI void mix3(const double* mm, const int* ee, double scale, double& m, int& e) {
    int emax = -(1 << 30);
    bool any = false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (mm[i] != 0.0) { emax = (any && emax > ee[i]) ? emax : ee[i]; any = true; }
    if (!any) emax = 0;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) s += ldexp(mm[i], ee[i] - emax);
    s *= scale;
    split(s, m, e);
    e += emax;
}

// kmer k on allele slot <=> 0 <= k-off < 32 && mask>>(k-off)&1  (reference src/kmerpath.cpp:33-48)
DEVI uint32_t kmer_on(uint32_t off, uint32_t mask, uint32_t k) {
    uint32_t d = k - off;
    return (k >= off && d < 32u) ? ((mask >> d) & 1u) : 0u;
}

// Product over all k-mers of variant v for allele-slot pair (s1,s2), one pair per lane.
// Called by a full wave; lds_m/lds_e are this wave's staging buffers [64][3].
// reference src/emissionprobabilitycomputer.cpp:36-53
DEVI void emission_pair_products(const DevContig& dc, const DevTable& tab, uint32_t v, uint32_t lane,
                                 bool active, uint32_t s1, uint32_t s2, double* lds_m, int* lds_e,
                                 double& pm, int& pe) {
    const uint32_t a0 = dc.allele_off[v];
    const uint32_t k0 = dc.kmer_off[v], K = dc.kmer_off[v + 1] - k0;
    const uint32_t cov = dc.cov[v];
    uint32_t off1 = 0, mask1 = 0, off2 = 0, mask2 = 0;
    bool u1 = false, u2 = false;
    if (active) {
        off1 = dc.allele_koff[a0 + s1]; mask1 = dc.allele_kmask[a0 + s1];
        off2 = dc.allele_koff[a0 + s2]; mask2 = dc.allele_kmask[a0 + s2];
        u1 = dc.allele_flags[a0 + s1] & 1; u2 = dc.allele_flags[a0 + s2] & 1;
    }
    pm = 1.0; pe = 0;
    for (uint32_t kb = 0; kb < K; kb += 64) {
        const uint32_t kk = kb + lane;
        if (kk < K) {
            double m[3]; int e[3];
            cn_lookup(tab, cov, dc.kmer_count[k0 + kk], m, e);
#pragma unroll
            for (int i = 0; i < 3; ++i) { lds_m[lane * 3 + i] = m[i]; lds_e[lane * 3 + i] = e[i]; }
        }
        wave_sync_lds();
        const uint32_t n = (K - kb) < 64u ? (K - kb) : 64u;
        if (active) {
            for (uint32_t q = 0; q < n; ++q) {
                const uint32_t k = kb + q;
                const uint32_t c = kmer_on(off1, mask1, k) + kmer_on(off2, mask2, k);
                double fm; int fe;
                if (u1 && u2) {
                    mix3(lds_m + q * 3, lds_e + q * 3, 1.0 / 3.0, fm, fe);
                } else if (u1 || u2) {
                    const uint32_t c2 = c + 1 > 2 ? 2 : c + 1;  // reference asserts c < 2 here
                    mix2(lds_m[q * 3 + c], lds_e[q * 3 + c], lds_m[q * 3 + c2], lds_e[q * 3 + c2], 0.5, fm, fe);
                } else {
                    fm = lds_m[q * 3 + c]; fe = lds_e[q * 3 + c];
                }
                pm *= fm; pe += fe;
            }
            double mm; int ee;
            split(pm, mm, ee);
            pm = mm; pe += ee;
            if (pe < PG_LD_MIN_EXP) { pm = 0.0; pe = 0; }  // underflows to 0 in the reference too
        }
        wave_sync_lds();
    }
}

DEVI uint32_t tri_local(uint32_t a, uint32_t b) {  // a <= b < PG_AMAX
    return a * PG_AMAX - a * (a - 1) / 2 + (b - a);
}
DEVI uint32_t tri_n(uint32_t a, uint32_t b, uint32_t n) {  // a <= b < n
    return a * n - a * (a - 1) / 2 + (b - a);
}

DEVI void decode_pair(uint32_t idx, uint32_t A, uint32_t& s1, uint32_t& s2) {
    uint32_t a = 0, rem = idx;
    while (a < A && rem >= A - a) { rem -= A - a; ++a; }
    s1 = a; s2 = a + rem;
}

// presence bitmap of the allele slots of one variant (256 bits, one per wave, in LDS)
DEVI bool slot_present(const uint32_t* pres, uint32_t s) { return (pres[s >> 5] >> (s & 31u)) & 1u; }
DEVI uint32_t local_index(const uint32_t* pres, uint32_t s) {  // number of present slots below s
    uint32_t n = 0;
    const uint32_t w = s >> 5;
    for (uint32_t q = 0; q < w; ++q) n += __popc(pres[q]);
    return n + __popc(pres[w] & ((1u << (s & 31u)) - 1u));
}
DEVI int slot_of(const DevContig& dc, uint32_t a0, uint32_t A, uint16_t a) {
    int s = -1;
    for (uint32_t q = 0; q < A; ++q)
        if (dc.allele_id[a0 + q] == a) s = (int)q;
    return s;
}

// ------------------------------------------------------------------------------------------
//  k_prep : one wave per variant
// ------------------------------------------------------------------------------------------
#ifndef PG_VREP
#define PG_VREP 8   // units (what one block of the one-variant-per-wave grid did) a block of k_prep / k_prep_bi walks: fewer, longer blocks
                    // (measured on 4096 chains of 8000 variants: 15.1 -> 13.2 ms; k_records and k_bins lose with the same change)
#endif
template <bool SPLIT>
DEVI void prep_unit(const DevContig& dc, const DevTable& tab, uint32_t unit) {
    constexpr int NLP = PG_AMAX * (PG_AMAX + 1) / 2;  // local pairs of a narrow column
    __shared__ double s_m[4][64 * 3];
    __shared__ int s_e[4][64 * 3];
    __shared__ double s_E[4][PG_ETAB];
    __shared__ double s_pm[4][NLP];
    __shared__ int s_pe[4][NLP];
    __shared__ uint32_t s_pres[4][8];
    __shared__ uint16_t s_aid[4][64];   // the variant's allele ids / flags (objects with <= 64 alleles: every real one)
    __shared__ uint8_t s_afl[4][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t v = unit * 4 + wave;
    // mixed chains (DevContig::prep_fast == 2): this kernel walks the LIST of the objects that are neither k_prep_bi's nor
    // k_prep_m4's — a grid over all variants whose waves mostly leave at once cost more than the objects it is here for
    if (dc.prep_w) {
        if (v >= dc.n_prep_w) return;
        v = dc.prep_w[v];
    }
    if (v >= dc.V) return;  // whole wave leaves; the kernel only uses wave-level sync

    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0;
    const uint32_t H = dc.H, HP = dc.HP;
    // the split path (pg_split.h): the list is the kernel's own, kept columns are the index's; the record goes to the
    // column's sample record instead of vrec
    uint32_t col = PG_COL_NONE;
    if constexpr (SPLIT) {
        col = dc.col_of[v];
        if (col == PG_COL_NONE) return;
    } else {
        if (dc.prep_fast == 2u && A == 2u && dc.kmer_off[v + 1] - dc.kmer_off[v] <= 32u) return;   // k_prep_bi's object
    }
    if (A > PG_MAX_ALLELES_PER_VARIANT || A == 0) {
        if (lane == 0 && !SPLIT) { atomicOr(dc.err, PG_DEVERR_TOO_MANY_ALLELES); dc.kept[v] = 0; }
        return;
    }
    // ---- ColumnIndexer rule: kept iff a selected path carries a defined non-ref allele
    //      (reference src/columnindexer.cpp:24-31); also which allele slots are present.
    uint32_t* pres = s_pres[wave];
    if (lane < 8) pres[lane] = 0;
    const bool staged = A <= 64u;  // allele list in LDS: one coalesced load instead of A dependent loads per path
    if (staged && lane < A) { s_aid[wave][lane] = dc.allele_id[a0 + lane]; s_afl[wave][lane] = dc.allele_flags[a0 + lane]; }
    wave_sync_lds();
    auto slot_lookup = [&](uint16_t a) -> int {
        if (staged) {
            int s = -1;
            for (uint32_t q = 0; q < A; ++q)
                if (s_aid[wave][q] == a) s = (int)q;
            return s;
        }
        return slot_of(dc, a0, A, a);
    };
    bool nonref = false, bad = false;
    for (uint32_t p = lane; p < H; p += 64) {
        const uint16_t a = dc.path_allele[(size_t)v * H + p];
        const int s = slot_lookup(a);
        if (s < 0) bad = true;
        else {
            atomicOr(&pres[(uint32_t)s >> 5], 1u << ((uint32_t)s & 31u));
            const uint8_t fl = staged ? s_afl[wave][s] : dc.allele_flags[a0 + s];
            if (a != 0 && !(fl & 1)) nonref = true;
        }
    }
    wave_sync_lds();
    const bool kept = SPLIT || __any(nonref) != 0;
    if (__any(bad) != 0) {
        if (lane == 0 && !SPLIT) { atomicOr(dc.err, PG_DEVERR_ALLELE_NOT_FOUND); dc.kept[v] = 0; }
        return;
    }
    uint32_t n_local = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) n_local += __popc(pres[q]);
    if constexpr (!SPLIT) {
        for (uint32_t q = lane; q < A; q += 64) dc.allele_present[a0 + q] = slot_present(pres, q) ? 1 : 0;
        if (lane == 0) dc.kept[v] = kept ? 1 : 0;
    }
    if (!kept) return;
    // more than PG_AMAX alleles on the selected paths: a WIDE column, its tables go to the side
    // buffer (pg_shim.cpp reserves an entry for every variant that could be wide; such a column is genotyped by k_post in
    // chunked mode — which the shim forces for every chain but k_sweep_small16x's — or by k_sweep_small16x + k_bins_wide)
    const bool wide = n_local > PG_AMAX;
    const uint32_t woff = wide && dc.wide_idx ? dc.wide_idx[v] : PG_WIDE_NONE;
    if (wide && (woff == PG_WIDE_NONE || n_local > PG_WIDE_MAX)) {
        if (lane == 0) atomicOr(dc.err, PG_DEVERR_TOO_MANY_LOCAL);
        return;
    }

    unsigned char* rec = SPLIT ? nullptr : dc.vrec + (size_t)v * dc.RB;
    unsigned char* srec = SPLIT ? (unsigned char*)dc.frec + (size_t)col * (dc.split == 1u ? PG_SREC1_BYTES : PG_SREC2_BYTES) : nullptr;
    // local (dense) allele index of every selected path; phantom paths of the padding get 255
    const uint32_t pspan = HP < 128u ? 128u : HP;
    for (uint32_t p0 = 0; p0 < pspan; p0 += 64) {
        const uint32_t p = p0 + lane;
        unsigned char val = PG_PHANTOM;
        if (p < H) {
            const int s = slot_lookup(dc.path_allele[(size_t)v * H + p]);
            val = (unsigned char)local_index(pres, (uint32_t)s);
        }
        if constexpr (SPLIT) {
            // a WIDE column's record carries its sixteen raw local alleles in its first sixteen bytes, the rest is zero (pg_small16x.h)
            if (n_local > PG_AMAX && p0 == 0) {
                if (lane < 16u) srec[lane] = val;
                if (lane >= 2u && lane < 15u) ((double*)srec)[lane] = 0.0;
            }
        } else {
        if (p < HP) rec[PG_REC_ALLELES + p] = val;
        const unsigned long long b1 = __ballot(val == 1);
        if (lane == 0 && p0 < 128u) ((unsigned long long*)(rec + PG_REC_BITS1))[p0 >> 6] = b1;
        }
    }

    // ---- emission products over ALL allele pairs of the object (a1<=a2; table is symmetric),
    //      64 pairs (one per lane) at a time.  Only pairs of alleles present on the selected paths
    //      enter the column's table, but the all_zeros rule looks at every pair of the object
    //      (emissionprobabilitycomputer.cpp:24).
    const uint32_t P = A * (A + 1) / 2;
    if (wide) {
        // two passes over the pairs: (1) all_zeros and the largest exponent X among the present pairs,
        // (2) the products again, straight into the wide entry — scaled by 2^-X for the recursion and as
        // (mantissa, exponent) for the bins (rare columns: the recomputation is cheaper than staging
        // thousands of pairs)
        const uint32_t S = n_local + 1u;
        unsigned char* ent = dc.wide + (size_t)woff * 16u;
        double* Ew = (double*)ent;
        double* Pm = (double*)(ent + PG_WIDE_OFF_PM(S));
        int* Pe = (int*)(ent + PG_WIDE_OFF_PE(S));
        uint16_t* slots = (uint16_t*)(ent + PG_WIDE_OFF_SLOT(S));
        for (uint32_t q = lane; q < S * S; q += 64) { Ew[q] = 0.0; Pm[q] = 0.0; Pe[q] = 0; }  // incl. the phantom row/column
        bool any_nz = false;
        int Xw = -(1 << 30);
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && (Xw == -(1 << 30) || !any_nz)) Xw = 0;
            for (uint32_t base = 0; base < P; base += 64) {
                const uint32_t idx = base + lane;
                const bool active = idx < P;
                uint32_t s1 = 0, s2 = 0;
                if (active) decode_pair(idx, A, s1, s2);
                double pm; int pe;
                emission_pair_products(dc, tab, v, lane, active, s1, s2, s_m[wave], s_e[wave], pm, pe);
                const bool both = active && slot_present(pres, s1) && slot_present(pres, s2);
                if (pass == 0) {
                    any_nz = any_nz || (__any(active && pm > 0.0) != 0);
                    const int xm = wave_max_i32((both && pm > 0.0) ? pe : -(1 << 30));
                    Xw = xm > Xw ? xm : Xw;
                } else if (both) {
                    const uint32_t la = local_index(pres, s1), lb = local_index(pres, s2);
                    double val, mv = pm;
                    int ev = pe;
                    if (!any_nz) { val = 1.0; mv = 0.5; ev = 1; }       // all_zeros: emissionprobabilitycomputer.cpp:31-34
                    else val = (pm > 0.0) ? ldexp(pm, pe - Xw) : pm;
                    Ew[la * S + lb] = val; Ew[lb * S + la] = val;
                    Pm[la * S + lb] = mv; Pm[lb * S + la] = mv;
                    Pe[la * S + lb] = ev; Pe[lb * S + la] = ev;
                }
            }
        }
        if constexpr (SPLIT) {
            if (lane == 0) {
                uint32_t l = 0;
                for (uint32_t sl = 0; sl < A; ++sl)
                    if (slot_present(pres, sl)) slots[l++] = (uint16_t)sl;
                const uint32_t fl = (any_nz ? 0u : PG_SREC_FLAG_ALLZERO) | PG_SREC_FLAG_WIDE;
                ((unsigned long long*)srec)[15] = (unsigned long long)(uint32_t)Xw | ((unsigned long long)fl << 32);
            }
            return;
        }
        if (lane < PG_ETAB) ((double*)(rec + PG_REC_E))[lane] = 0.0;
        if (lane < 4) ((double*)rec)[lane] = 0.0;
        if (lane == 0) {
            *(uint32_t*)(rec + PG_REC_VARIANT) = v;
            *(int32_t*)(rec + PG_REC_EXP) = Xw;
            rec[PG_REC_NLOCAL] = (unsigned char)n_local;
            rec[PG_REC_FLAGS] = (unsigned char)((any_nz ? 0 : PG_REC_FLAG_ALLZERO) | PG_REC_FLAG_WIDE);
            rec[PG_REC_FLAGS + 1] = 0; rec[PG_REC_FLAGS + 2] = 0;
            *(uint32_t*)(rec + PG_REC_WIDE_IDX) = woff;
            uint16_t* ls = (uint16_t*)(rec + PG_REC_LOCAL_SLOT);
            for (uint32_t l = 0; l < 8; ++l) ls[l] = 0;
            uint32_t l = 0;
            for (uint32_t sl = 0; sl < A; ++sl)
                if (slot_present(pres, sl)) slots[l++] = (uint16_t)sl;
            *(uint32_t*)(rec + PG_REC_AUX) = dc.aux_idx ? dc.aux_idx[v] : PG_WIDE_NONE;
        }
        return;
    }
    if (lane < (uint32_t)NLP) { s_pm[wave][lane] = 0.0; s_pe[wave][lane] = 0; }
    wave_sync_lds();
    bool any_nonzero = false;
    if (P <= 32u) {
        // Few pairs (<= 7 alleles: every BASELINE shape): G lanes per pair share its k-mers (lane g takes k-mers
        // g, g + G, ...) and the partial products are folded by xor shuffles — K / G sequential factors instead of K
        // on 3 of 64 lanes.  Same factors as emission_pair_products (emissionprobabilitycomputer.cpp:36-53); only the
        // order of the multiplications differs.
        const uint32_t G = P <= 4u ? 16u : (P <= 8u ? 8u : (P <= 16u ? 4u : 2u));
        const uint32_t idx = lane / G, g = lane % G;
        const bool active = idx < P;
        uint32_t s1 = 0, s2 = 0;
        if (active) decode_pair(idx, A, s1, s2);
        const uint32_t k0 = dc.kmer_off[v], K = dc.kmer_off[v + 1] - k0;
        const uint32_t cov = dc.cov[v];
        uint32_t off1 = 0, mask1 = 0, off2 = 0, mask2 = 0;
        bool u1 = false, u2 = false;
        if (active) {
            off1 = dc.allele_koff[a0 + s1]; mask1 = dc.allele_kmask[a0 + s1];
            off2 = dc.allele_koff[a0 + s2]; mask2 = dc.allele_kmask[a0 + s2];
            u1 = s_afl[wave][s1] & 1; u2 = s_afl[wave][s2] & 1;  // (P <= 32 implies A <= 7: staged)
        }
        double pm = 1.0;
        int pe = 0;
        double* lds_m = s_m[wave];
        int* lds_e = s_e[wave];
        for (uint32_t kb = 0; kb < K; kb += 64) {
            const uint32_t kk = kb + lane;
            if (kk < K) {
                double m[3]; int e[3];
                cn_lookup(tab, cov, dc.kmer_count[k0 + kk], m, e);
#pragma unroll
                for (int i = 0; i < 3; ++i) { lds_m[lane * 3 + i] = m[i]; lds_e[lane * 3 + i] = e[i]; }
            }
            wave_sync_lds();
            const uint32_t n = (K - kb) < 64u ? (K - kb) : 64u;
            if (active) {
                for (uint32_t q = g; q < n; q += G) {
                    const uint32_t k = kb + q;
                    const uint32_t c = kmer_on(off1, mask1, k) + kmer_on(off2, mask2, k);
                    double fm; int fe;
                    if (u1 && u2) {
                        mix3(lds_m + q * 3, lds_e + q * 3, 1.0 / 3.0, fm, fe);
                    } else if (u1 || u2) {
                        const uint32_t c2 = c + 1 > 2 ? 2 : c + 1;  // reference asserts c < 2 here
                        mix2(lds_m[q * 3 + c], lds_e[q * 3 + c], lds_m[q * 3 + c2], lds_e[q * 3 + c2], 0.5, fm, fe);
                    } else {
                        fm = lds_m[q * 3 + c]; fe = lds_e[q * 3 + c];
                    }
                    pm *= fm; pe += fe;
                }
                double mm; int ee;
                split(pm, mm, ee);
                pm = mm; pe += ee;
            }
            wave_sync_lds();
        }
        for (uint32_t m = 1; m < G; m <<= 1) {  // fold the G partial products of every pair
            pm *= __shfl_xor(pm, (int)m);
            pe += __shfl_xor(pe, (int)m);
        }
        {
            double mm; int ee;
            split(pm, mm, ee);
            pm = mm; pe += ee;
            if (pe < PG_LD_MIN_EXP) { pm = 0.0; pe = 0; }  // underflows to 0 in the reference too
        }
        any_nonzero = __any(active && pm > 0.0) != 0;
        if (active && g == 0 && slot_present(pres, s1) && slot_present(pres, s2)) {
            const uint32_t la = local_index(pres, s1), lb = local_index(pres, s2);
            s_pm[wave][tri_local(la, lb)] = pm;  // s1 <= s2  =>  la <= lb
            s_pe[wave][tri_local(la, lb)] = pe;
        }
    } else
    for (uint32_t base = 0; base < P; base += 64) {
        const uint32_t idx = base + lane;
        const bool active = idx < P;
        uint32_t s1 = 0, s2 = 0;
        if (active) decode_pair(idx, A, s1, s2);
        double pm; int pe;
        emission_pair_products(dc, tab, v, lane, active, s1, s2, s_m[wave], s_e[wave], pm, pe);
        any_nonzero = any_nonzero || (__any(active && pm > 0.0) != 0);
        if (active && slot_present(pres, s1) && slot_present(pres, s2)) {
            const uint32_t la = local_index(pres, s1), lb = local_index(pres, s2);
            s_pm[wave][tri_local(la, lb)] = pm;  // s1 <= s2  =>  la <= lb
            s_pe[wave][tri_local(la, lb)] = pe;
        }
    }
    wave_sync_lds();
    const bool all_zeros = !any_nonzero;
    // this lane's local pair (la <= lb < n_local), if any
    uint32_t la = 0, lb = 0;
    bool mine = false;
    if (lane < (uint32_t)NLP) {
        uint32_t rem = lane;
        while (la < PG_AMAX && rem >= PG_AMAX - la) { rem -= PG_AMAX - la; ++la; }
        lb = la + rem;
        mine = lb < n_local;
    }
    const double pm = mine ? s_pm[wave][lane] : 0.0;
    const int pe = mine ? s_pe[wave][lane] : 0;
    int X = wave_max_i32((mine && pm > 0.0) ? pe : -(1 << 30));
    if (X == -(1 << 30) || all_zeros) X = 0;

    if constexpr (SPLIT) {
        // lane = local pair in tri_local order (above): its scaled entry, X and the flags straight into the column's sample record
        const double val = !mine ? 0.0 : (all_zeros ? 1.0 : ((pm > 0.0) ? ldexp(pm, pe - X) : pm));
        const bool precise = __any(mine && !all_zeros && pm > 0.0 && pe - X < -1021) != 0;
        const uint32_t fl = (all_zeros ? PG_SREC_FLAG_ALLZERO : 0u) | (precise ? PG_SREC_FLAG_PRECISE : 0u);
        if (dc.split == 1u) {
            // (a biallelic object with more than 32 k-mers in an all-biallelic chain: k_sweep_small16's 64-byte record)
            const double E00 = readlane_f64(val, 0), E01 = readlane_f64(val, 1), E11 = readlane_f64(val, PG_AMAX);
            if (lane == 0) {
                const v2f64* ir = (const v2f64*)(dc.ix_rec + (size_t)col * PG_IXREC_BYTES);
                const unsigned long long bits = (unsigned long long)__double_as_longlong(ir[2].x);
                const unsigned long long packed = (bits & 0xFFFFull) | ((unsigned long long)fl << 16) | ((unsigned long long)(uint32_t)X << 32);
                v2f64* dst = (v2f64*)srec;
                dst[0] = ir[0]; dst[1] = ir[1];
                dst[2] = v2f64{E00, E01};
                dst[3] = v2f64{E11, __longlong_as_double((long long)packed)};
            }
        } else if (lane < 16u) {
            double out = val;
            if (lane == 15u) out = __longlong_as_double((long long)((unsigned long long)(uint32_t)X | ((unsigned long long)fl << 32)));
            ((double*)srec)[lane] = out;
        }
        if (precise && lane < 16u) {
            ((double*)(dc.cprec + (size_t)col * PG_CPREC_BYTES))[lane] = !mine ? 0.0 : (all_zeros ? 0.5 : pm);
            ((int*)(dc.cprec + (size_t)col * PG_CPREC_BYTES + 128u))[lane] = !mine ? 0 : (all_zeros ? 1 : pe);
        }
        return;
    }
    if (lane < PG_ETAB) s_E[wave][lane] = 0.0;
    wave_sync_lds();
    if (mine) {
        double val;
        if (all_zeros) val = 1.0;                           // emissionprobabilitycomputer.cpp:31-34
        else val = (pm > 0.0) ? ldexp(pm, pe - X) : pm;     // 0 (or NaN) stays
        s_E[wave][la * PG_ESTRIDE + lb] = val;
        s_E[wave][lb * PG_ESTRIDE + la] = val;
        // the unscaled product as (mantissa, exponent): what a finished posterior bin is multiplied with
        const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
        unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
        const uint32_t pi = tri_n(la, lb, pn);  // la <= lb < n_local <= pair_n
        ((double*)vp)[pi] = all_zeros ? 0.5 : pm;
        ((int*)(vp + (size_t)NP * 8u))[pi] = all_zeros ? 1 : pe;
    }
    wave_sync_lds();
    if (lane < PG_ETAB) ((double*)(rec + PG_REC_E))[lane] = s_E[wave][lane];
    if (lane < 4) ((double*)rec)[lane] = 0.0;  // transition constants are filled by k_records
    if (lane == 0) {
        *(uint32_t*)(rec + PG_REC_VARIANT) = v;
        *(int32_t*)(rec + PG_REC_EXP) = X;
        rec[PG_REC_NLOCAL] = (unsigned char)n_local;
        rec[PG_REC_FLAGS] = all_zeros ? PG_REC_FLAG_ALLZERO : 0;
        rec[PG_REC_FLAGS + 1] = 0; rec[PG_REC_FLAGS + 2] = 0;
        *(uint32_t*)(rec + PG_REC_WIDE_IDX) = PG_WIDE_NONE;
        uint16_t* ls = (uint16_t*)(rec + PG_REC_LOCAL_SLOT);
        uint32_t l = 0;
        for (uint32_t s = 0; s < A && l < 8; ++s)
            if (slot_present(pres, s)) ls[l++] = (uint16_t)s;
        for (; l < 8; ++l) ls[l] = 0;
        *(uint32_t*)(rec + PG_REC_AUX) = dc.aux_idx ? dc.aux_idx[v] : PG_WIDE_NONE;   // (the last two of the eight slots: PG_AMAX = 5 are used)
    }
}
__global__ __launch_bounds__(256) void k_prep(const DevContig* __restrict__ contigs, DevTable tab) {
    const DevContig& dc = contigs[blockIdx.y];
    if (dc.prep_fast == 1u || dc.split) return;  // k_prep_bi's chain (2: its two-allele objects only, see prep_unit); the split path's
#pragma unroll 1
    for (uint32_t r = 0; r < (uint32_t)PG_VREP; ++r) {
        prep_unit<false>(dc, tab, blockIdx.x * (uint32_t)PG_VREP + r);
        wave_sync_lds();   // (a wave's LDS slices are its own: the next unit's writes stay behind this unit's reads)
    }
}
// ... of the split path (pg_split.h): the objects of a split chain that are neither k_prep_s_bi's nor k_prep_s_m4's (the list prep_w)
__global__ __launch_bounds__(256) void k_prep_s_w(const DevContig* __restrict__ contigs, DevTable tab) {
    const DevContig& dc = contigs[blockIdx.y];
    if (!dc.split || !dc.prep_w) return;
#pragma unroll 1
    for (uint32_t r = 0; r < (uint32_t)PG_VREP; ++r) {
        prep_unit<true>(dc, tab, blockIdx.x * (uint32_t)PG_VREP + r);
        wave_sync_lds();
    }
}




// ------------------------------------------------------------------------------------------
//  k_prep_bi : the same for chains of biallelic objects with <= 32 k-mers and <= 64 selected paths (every BASELINE
//  shape but the multiallelic one) — FOUR variants per wave, one per DPP row of 16 lanes.  k_prep spends a whole wave,
//  its LDS staging and half a dozen wave syncs on 3 allele pairs x 20 k-mers; here a lane takes 4 paths and 2 k-mers,
//  the three pair products are folded by DPP row shifts, nothing goes through LDS.  Same factors as k_prep
//  (emissionprobabilitycomputer.cpp:36-53), multiplied in a different order: the (mantissa, exponent) products can
//  differ in the last bit.
// ------------------------------------------------------------------------------------------
template <int CTRL>
DEVI void dpp_mul_step(double& m, int& e) {
    // lanes without a source multiply by 1 * 2^0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(m), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(m), CTRL, 0xF, 0xF, false);
    const int oe = __builtin_amdgcn_update_dpp(0, e, CTRL, 0xF, 0xF, false);
    m *= __hiloint2double(hi, lo);
    e += oe;
}
DEVI double row_last_f64(double v) {  // lane 15 of the row of 16 to all its lanes
    return __longlong_as_double(__builtin_amdgcn_update_dpp(__double_as_longlong(v), __double_as_longlong(v), 0x15F, 0xF, 0xF, true));
}
DEVI int row_last_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x15F, 0xF, 0xF, true); }

DEVI void prep_bi_unit(const DevContig& dc, const DevTable& tab, uint32_t unit) {
    const uint32_t lane = threadIdx.x & 63u, grp = lane >> 4, l = lane & 15u;
    // mixed chains (DevContig::prep_fast == 2) hand this kernel the LIST of its objects (prep_b): no rows idling along on
    // the objects of the other kernels
    const uint32_t nobj = dc.prep_b ? dc.n_prep_b : dc.V;
    const uint32_t pos = unit * 16u + (threadIdx.x >> 6) * 4u + grp;
    if (unit * 16u + (threadIdx.x >> 6) * 4u >= nobj) return;  // the whole wave is beyond the contig
    bool live = pos < nobj;   // (a row beyond the contig idles along: the ballots below are wave-wide)
    const uint32_t v = dc.prep_b ? dc.prep_b[live ? pos : nobj - 1u] : pos;
    const uint32_t vv = live ? v : dc.V - 1u;
    const uint32_t H = dc.H, HP = dc.HP;
    const uint32_t a0 = dc.allele_off[vv];  // two alleles
    // DevContig::prep_fast == 2: a chain with other objects too — this kernel takes its two-allele objects with <= 32
    // k-mers, k_prep the rest (a row whose object is not this kernel's idles along like one beyond the contig)
    const bool mine = dc.prep_fast != 2u || (dc.allele_off[vv + 1] - a0 == 2u && dc.kmer_off[vv + 1] - dc.kmer_off[vv] <= 32u);
    live = live && mine;
    const uint32_t a1 = mine ? a0 + 1u : a0;
    const uint16_t id0 = dc.allele_id[a0], id1 = dc.allele_id[a1];
    const bool u0 = dc.allele_flags[a0] & 1, u1 = dc.allele_flags[a1] & 1;
    // ---- ColumnIndexer rule and the allele of every selected path (reference src/columnindexer.cpp:24-31): lane l
    //      takes paths l, 16 + l, 32 + l, 48 + l
    uint32_t slot[4];
    bool bad = false, nonref = false, has0 = false, has1 = false;
    unsigned long long ones = 0;  // bit p: selected path p carries allele slot 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t p = 16u * (uint32_t)i + l;
        const bool inb = live && p < H;
        const uint16_t a = inb ? dc.path_allele[(size_t)vv * H + p] : (uint16_t)0;
        const bool s1 = inb && a == id1, s0 = inb && !s1 && a == id0;  // (the last matching slot, as slot_of)
        slot[i] = s1 ? 1u : 0u;
        bad = bad || (inb && !s1 && !s0);
        has0 = has0 || s0;
        has1 = has1 || s1;
        nonref = nonref || ((s1 || s0) && a != 0 && !(s1 ? u1 : u0));
        const unsigned long long b = __ballot(s1);
        ones |= ((b >> (16u * grp)) & 0xFFFFull) << (16u * (uint32_t)i);
    }
    const uint32_t sh16 = 16u * grp;
    const bool any_bad = ((__ballot(bad) >> sh16) & 0xFFFFull) != 0, kept = ((__ballot(nonref) >> sh16) & 0xFFFFull) != 0;
    has0 = ((__ballot(has0) >> sh16) & 0xFFFFull) != 0;
    has1 = ((__ballot(has1) >> sh16) & 0xFFFFull) != 0;
    if (!live) return;
    if (any_bad) {
        if (l == 0) { atomicOr(dc.err, PG_DEVERR_ALLELE_NOT_FOUND); dc.kept[v] = 0; }
        return;
    }
    if (l == 0) { dc.allele_present[a0] = has0 ? 1 : 0; dc.kept[v] = kept ? 1 : 0; }
    if (l == 1) dc.allele_present[a0 + 1] = has1 ? 1 : 0;
    if (!kept) return;
    const uint32_t n_local = (has0 ? 1u : 0u) + (has1 ? 1u : 0u);
    const uint32_t loc1 = has0 ? 1u : 0u;  // local (dense) index of slot 1; slot 0 is local 0
    // The record is put together in LDS and leaves as whole 16-byte pieces (below): written field by field straight to
    // global memory, a record's lines reached HBM several times over (PMC: 1040 bytes written per 384-byte record).
    __shared__ unsigned char s_rec[4][4][448] __attribute__((aligned(16)));   // [wave][variant of the wave][RB <= 448: H <= 64]
    unsigned char* rec = s_rec[threadIdx.x >> 6][grp];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t p = 16u * (uint32_t)i + l;
        if (p < HP) rec[PG_REC_ALLELES + p] = (unsigned char)(p < H ? (slot[i] ? loc1 : 0u) : (uint32_t)PG_PHANTOM);
    }
    for (uint32_t pad = PG_REC_ALLELES + HP + l; pad < dc.RB; pad += 16u) rec[pad] = 0;   // (the record's tail up to RB)
    // ---- emission products of the three allele pairs (0,0), (0,1), (1,1): lane l takes k-mers l and 16 + l
    const uint32_t k0 = dc.kmer_off[v], K = dc.kmer_off[v + 1] - k0, cov = dc.cov[v];
    const uint32_t off0 = dc.allele_koff[a0], mask0 = dc.allele_kmask[a0], off1 = dc.allele_koff[a0 + 1], mask1 = dc.allele_kmask[a0 + 1];
    double pm[3] = {1.0, 1.0, 1.0};
    int pe[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t k = 16u * (uint32_t)j + l;
        if (k < K) {
            double m[3]; int e[3];
            cn_lookup(tab, cov, dc.kmer_count[k0 + k], m, e);
            const uint32_t on0 = kmer_on(off0, mask0, k), on1 = kmer_on(off1, mask1, k);
            auto factor = [&](uint32_t c, bool ua, bool ub, double& fm, int& fe) {
                const double mc = c == 0 ? m[0] : (c == 1 ? m[1] : m[2]);
                const int ec = c == 0 ? e[0] : (c == 1 ? e[1] : e[2]);
                if (ua && ub) mix3(m, e, 1.0 / 3.0, fm, fe);
                else if (ua || ub) {
                    const double m2 = c == 0 ? m[1] : m[2];   // c + 1, capped at 2 (the reference asserts c < 2 here)
                    const int e2 = c == 0 ? e[1] : e[2];
                    mix2(mc, ec, m2, e2, 0.5, fm, fe);
                } else { fm = mc; fe = ec; }
            };
            double fm; int fe;
            factor(2u * on0, u0, u0, fm, fe); pm[0] *= fm; pe[0] += fe;
            factor(on0 + on1, u0, u1, fm, fe); pm[1] *= fm; pe[1] += fe;
            factor(2u * on1, u1, u1, fm, fe); pm[2] *= fm; pe[2] += fe;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        dpp_mul_step<0x111>(pm[q], pe[q]);  // row_shr:1,2,4,8: lane 15 holds the product over the row
        dpp_mul_step<0x112>(pm[q], pe[q]);
        dpp_mul_step<0x114>(pm[q], pe[q]);
        dpp_mul_step<0x118>(pm[q], pe[q]);
        pm[q] = row_last_f64(pm[q]);
        pe[q] = row_last_i32(pe[q]);
        double mm; int ee;
        split(pm[q], mm, ee);
        pm[q] = mm; pe[q] += ee;
        if (pe[q] < PG_LD_MIN_EXP) { pm[q] = 0.0; pe[q] = 0; }  // underflows to 0 in the reference too
    }
    const bool all_zeros = !(pm[0] > 0.0 || pm[1] > 0.0 || pm[2] > 0.0);  // over ALL pairs of the object (emissionprobabilitycomputer.cpp:24)
    // local pairs (la <= lb < n_local) -> object pairs: both alleles present: the three pairs; one: its own pair
    const int q00 = has0 ? 0 : 2;
    const double m00 = pm[q00], m01 = pm[1], m11 = pm[2];
    const int e00 = pe[q00], e01 = pe[1], e11 = pe[2];
    const bool two = n_local == 2u;
    int X = -(1 << 30);
    if (m00 > 0.0) X = e00;
    if (two && m01 > 0.0 && e01 > X) X = e01;
    if (two && m11 > 0.0 && e11 > X) X = e11;
    if (X == -(1 << 30) || all_zeros) X = 0;
    auto scaled = [&](double m, int e) { return all_zeros ? 1.0 : (m > 0.0 ? ldexp(m, e - X) : m); };  // (0 or NaN stays)
    const double E00 = scaled(m00, e00), E01 = two ? scaled(m01, e01) : 0.0, E11 = two ? scaled(m11, e11) : 0.0;
    // the 6 x 6 table of the record: entries l, 16 + l, 32 + l
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t t = 16u * (uint32_t)i + l;
        if (t < (uint32_t)PG_ETAB) {
            const double val = t == 0 ? E00 : ((t == 1 || t == (uint32_t)PG_ESTRIDE) ? E01 : (t == (uint32_t)PG_ESTRIDE + 1u ? E11 : 0.0));
            ((double*)(rec + PG_REC_E))[t] = val;
        }
    }
    // the unscaled products as (mantissa, exponent): what a finished posterior bin is multiplied with
    const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
    unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
    if (l < 3u && (l == 0u || two)) {
        const uint32_t la = l == 2u ? 1u : 0u, lb = l == 0u ? 0u : 1u;
        const uint32_t pi = tri_n(la, lb, pn);
        const double mv = l == 0u ? m00 : (l == 1u ? m01 : m11);
        const int ev = l == 0u ? e00 : (l == 1u ? e01 : e11);
        ((double*)vp)[pi] = all_zeros ? 0.5 : mv;
        ((int*)(vp + (size_t)NP * 8u))[pi] = all_zeros ? 1 : ev;
    }
    if (l < 4u) ((double*)rec)[l] = 0.0;  // transition constants are filled by k_records
    if (l < 8u) ((uint16_t*)(rec + PG_REC_LOCAL_SLOT))[l] = (uint16_t)((l == 0u && !has0) || (l == 1u && two) ? 1u : 0u);
    if (l == 8u) {
        *(uint32_t*)(rec + PG_REC_VARIANT) = v;
        *(int32_t*)(rec + PG_REC_EXP) = X;
        rec[PG_REC_NLOCAL] = (unsigned char)n_local;
        rec[PG_REC_FLAGS] = all_zeros ? PG_REC_FLAG_ALLZERO : 0;
        rec[PG_REC_FLAGS + 1] = 0; rec[PG_REC_FLAGS + 2] = 0;
        *(uint32_t*)(rec + PG_REC_WIDE_IDX) = PG_WIDE_NONE;
        *(uint32_t*)(rec + PG_REC_AUX) = PG_WIDE_NONE;   // (a biallelic object has no aux slot: what pg_device.h says a reader finds; behind the slot entries 6, 7 above)
    }
    if (l == 9u) {
        ((unsigned long long*)(rec + PG_REC_BITS1))[0] = has0 ? ones : 0ull;  // bit p: path p carries LOCAL allele 1
        ((unsigned long long*)(rec + PG_REC_BITS1))[1] = 0ull;
    }
    wave_sync_lds();
    {   // copy-out: only the 16-byte pieces of the record that are not zero by construction — header (bytes 32 .. 63), row bits
        // (64 .. 79), E'00 E'01 (80 .. 95), E'10 E'11 (128 .. 143), the path alleles (PG_REC_ALLELES ...): 6 of 24 pieces at 16
        // paths, ONE store per object.  The rest of the record is zero since the job was built (pg_shim.cpp zeroes the variant
        // records once) and no kernel ever writes anything else there for THIS object (the kernel that prepares an object is a
        // function of the index alone).  The kernel waits for its memory 70 % of its cycles (profiles/r05_cohort_h16m_summary.txt,
        // round 4: 27 GB written per launch for 12.6 GB of records).
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2* src = (const f64x2*)rec;
        f64x2* dst = (f64x2*)(dc.vrec + (size_t)v * dc.RB);
        const uint32_t n_al = HP / 16u;   // pieces of path alleles
        if (l < 5u + n_al) {
            const uint32_t p = l == 0u ? 2u : l == 1u ? 3u : l == 2u ? 4u : l == 3u ? 5u : l == 4u ? 8u : (uint32_t)(PG_REC_ALLELES / 16) + (l - 5u);
            dst[p] = src[p];
        }
    }
    wave_sync_lds();   // (the slot is rewritten by the wave's next unit)
}
__global__ __launch_bounds__(256) void k_prep_bi(const DevContig* __restrict__ contigs, DevTable tab) {
    const DevContig& dc = contigs[blockIdx.y];
    if (!dc.prep_fast || dc.split) return;
#pragma unroll 1
    for (uint32_t r = 0; r < (uint32_t)PG_VREP; ++r) prep_bi_unit(dc, tab, blockIdx.x * (uint32_t)PG_VREP + r);
}


// ------------------------------------------------------------------------------------------
//  unit-level entry: full A x A emission products of one variant as (mantissa, exponent)
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
//  k_prep_m4 : FOUR multiallelic objects per wave (a DPP row of 16 lanes each) — the objects with 3 .. PG_AMAX alleles and at
//  most 64 k-mers of mixed chains with at most 64 paths (DevContig::prep_fast == 2), taken from the chain's list prep_m4
//  (HPRC-style panels and the 15 + 1 sampled paths have a fifth of their objects such: one wave per object — k_prep — spent
//  ~1100 instructions on each, 19 of the 31 ms of `cohort_h16m`'s preparation).  Lane l of the row takes paths l, 16 + l,
//  32 + l, 48 + l, the k-mers l, 16 + l, ... (their three copy-number factors go to LDS once) and the allele PAIR l (at most
//  15): its product over the k-mers is a sequential loop in the lane — no cross-lane fold.  Same factors, rules (all_zeros
//  over every pair of the object, the largest exponent over the present pairs) and record layout as prep_unit; only the
//  order of the multiplications differs.  reference src/emissionprobabilitycomputer.cpp:9-53, src/columnindexer.cpp:24-31
// ------------------------------------------------------------------------------------------
DEVI uint32_t row16_ballot(bool p, uint32_t grp) { return (uint32_t)((__ballot(p) >> (16u * grp)) & 0xFFFFull); }
template <bool SPLIT>
DEVI void prep_m4_unit(const DevContig& dc, const DevTable& tab, uint32_t unit) {
    // (row strides padded: at 192 doubles / ints the four rows of a wave fell on the same LDS banks — every factor read a
    //  four-way conflict, 40 % of the kernel's busy cycles)
    __shared__ double s_m[4][4][64 * 3 + 4];
    __shared__ int s_e[4][4][64 * 3 + 8];
    __shared__ unsigned char s_rec[4][4][448] __attribute__((aligned(16)));   // [wave][object of the wave][RB <= 448: H <= 64]
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u, grp = lane >> 4, l = lane & 15u;
    const uint32_t base = unit * 16u + wv * 4u;
    if (base >= dc.n_prep_m4) return;   // the whole wave is beyond the list
    bool live = base + grp < dc.n_prep_m4;   // (a row beyond the list idles along: the ballots below are wave-wide)
    const uint32_t v = dc.prep_m4[live ? base + grp : dc.n_prep_m4 - 1u];
    const uint32_t H = dc.H, HP = dc.HP;
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0;   // 3 .. PG_AMAX
    uint32_t ids[PG_AMAX];
    bool und[PG_AMAX];
#pragma unroll
    for (int q = 0; q < PG_AMAX; ++q) {
        const bool in = (uint32_t)q < A;
        ids[q] = in ? (uint32_t)dc.allele_id[a0 + q] : 0xFFFFFFFFu;
        und[q] = in && (dc.allele_flags[a0 + q] & 1);
    }
    // ---- ColumnIndexer rule and the allele slot of every selected path
    int slot[4];
    uint32_t mine = 0;   // slots this lane's paths carry
    bool bad = false, nonref = false;
    uint32_t pres = 0;
    uint32_t col = PG_COL_NONE;   // (SPLIT) the object's column
    if constexpr (SPLIT) {
        // the split path (pg_split.h): kept columns and present alleles are the INDEX's — nothing to scan
        col = live ? dc.col_of[v] : PG_COL_NONE;
        live = live && col != PG_COL_NONE;
#pragma unroll
        for (int q = 0; q < PG_AMAX; ++q) if ((uint32_t)q < A && dc.allele_present[a0 + q]) pres |= 1u << q;
        if (!live) return;
    } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t p = 16u * (uint32_t)i + l;
        const bool inb = live && p < H;
        const uint32_t a = inb ? (uint32_t)dc.path_allele[(size_t)v * H + p] : 0u;
        int sl = -1;
#pragma unroll
        for (int q = 0; q < PG_AMAX; ++q) if (ids[q] == a) sl = q;   // (the last matching slot, as slot_of)
        slot[i] = sl;
        bad = bad || (inb && sl < 0);
        if (inb && sl >= 0) {
            mine |= 1u << sl;
            bool u = false;
#pragma unroll
            for (int q = 0; q < PG_AMAX; ++q) u = u || (sl == q && und[q]);
            nonref = nonref || (a != 0u && !u);
        }
    }
    const bool any_bad = row16_ballot(bad, grp) != 0u, kept = row16_ballot(nonref, grp) != 0u;
#pragma unroll
    for (int q = 0; q < PG_AMAX; ++q) pres |= (row16_ballot((mine >> q) & 1u, grp) != 0u ? 1u : 0u) << q;
    if (!live) return;
    if (any_bad) {
        if (l == 0) { atomicOr(dc.err, PG_DEVERR_ALLELE_NOT_FOUND); dc.kept[v] = 0; }
        return;
    }
    if (l < A) dc.allele_present[a0 + l] = (pres >> l) & 1u;
    if (l == 0) dc.kept[v] = kept ? 1 : 0;
    if (!kept) return;
    }
    const uint32_t n_local = __popc(pres);
    auto local_of = [&](uint32_t sl) { return (uint32_t)__popc(pres & ((1u << sl) - 1u)); };
    unsigned char* rec = s_rec[wv][grp];
    unsigned long long ones = 0;   // bit p: selected path p carries LOCAL allele 1
    if constexpr (!SPLIT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t p = 16u * (uint32_t)i + l;
        const uint32_t loc = (p < H && slot[i] >= 0) ? local_of((uint32_t)slot[i]) : (uint32_t)PG_PHANTOM;
        if (p < HP) rec[PG_REC_ALLELES + p] = (unsigned char)loc;
        ones |= (unsigned long long)row16_ballot(loc == 1u, grp) << (16u * (uint32_t)i);
    }
    for (uint32_t pad = PG_REC_ALLELES + HP + l; pad < dc.RB; pad += 16u) rec[pad] = 0;   // (the record's tail up to RB)
    }
    // ---- the copy-number factors of the object's k-mers -> LDS (lane l: k-mers l, 16 + l, 32 + l, 48 + l)
    const uint32_t k0 = dc.kmer_off[v], K = dc.kmer_off[v + 1] - k0, cov = dc.cov[v];
    double* lm = s_m[wv][grp];
    int* le = s_e[wv][grp];
    // (as many rounds as the wave's largest object needs.  A BALLOT, not a shuffle butterfly: rows of the wave have left — objects
    //  that are no column — and a butterfly through their lanes loses the other rows' values: round 5's first version staged
    //  nothing for an object whose wave began with such a row)
    for (uint32_t i = 0; i < 4u; ++i) {
        if (!__any(16u * i < K)) break;
        const uint32_t k = 16u * i + l;
        if (k < K) {
            double m[3]; int e[3];
            cn_lookup(tab, cov, dc.kmer_count[k0 + k], m, e);
#pragma unroll
            for (int c = 0; c < 3; ++c) { lm[k * 3 + c] = m[c]; le[k * 3 + c] = e[c]; }
        }
    }
    if constexpr (!SPLIT) {
    if (l < 4u) ((double*)rec)[l] = 0.0;  // transition constants are filled by k_records
    for (uint32_t t = l; t < (uint32_t)PG_ETAB; t += 16u) ((double*)(rec + PG_REC_E))[t] = 0.0;
    }
    wave_sync_lds();
    // ---- the product of allele pair l over the k-mers
    const uint32_t P = A * (A + 1u) / 2u;
    const bool active = l < P;
    uint32_t s1 = 0, s2 = 0;
    if (active) decode_pair(l, A, s1, s2);
    uint32_t off1 = 0, mask1 = 0, off2 = 0, mask2 = 0;
    bool u1 = false, u2 = false;
    if (active) {
        off1 = dc.allele_koff[a0 + s1]; mask1 = dc.allele_kmask[a0 + s1];
        off2 = dc.allele_koff[a0 + s2]; mask2 = dc.allele_kmask[a0 + s2];
        u1 = dc.allele_flags[a0 + s1] & 1; u2 = dc.allele_flags[a0 + s2] & 1;
    }
    double pm = 1.0;
    int pe = 0;
    if (active) {
        // (k-mer k lies on an allele <=> bit k - off of its mask, 0 <= k - off < 32: as a 64-bit word, K <= 64 — two shifts and an
        //  add per k-mer where kmer_on took a dozen instructions)
        unsigned long long M1 = off1 < 64u ? (unsigned long long)mask1 << off1 : 0ull, M2 = off2 < 64u ? (unsigned long long)mask2 << off2 : 0ull;
        for (uint32_t k = 0; k < K; ++k, M1 >>= 1, M2 >>= 1) {
            const uint32_t c = (uint32_t)(M1 & 1ull) + (uint32_t)(M2 & 1ull);
            double fm; int fe;
            if (u1 && u2) mix3(lm + k * 3, le + k * 3, 1.0 / 3.0, fm, fe);
            else if (u1 || u2) {
                const uint32_t c2 = c + 1 > 2 ? 2 : c + 1;  // reference asserts c < 2 here
                mix2(lm[k * 3 + c], le[k * 3 + c], lm[k * 3 + c2], le[k * 3 + c2], 0.5, fm, fe);
            } else { fm = lm[k * 3 + c]; fe = le[k * 3 + c]; }
            pm *= fm; pe += fe;
        }
        double mm; int ee;
        split(pm, mm, ee);
        pm = mm; pe += ee;
        if (pe < PG_LD_MIN_EXP) { pm = 0.0; pe = 0; }  // underflows to 0 in the reference too
    }
    const bool all_zeros = row16_ballot(active && pm > 0.0, grp) == 0u;  // over ALL pairs of the object (emissionprobabilitycomputer.cpp:24)
    const bool both = active && ((pres >> s1) & 1u) && ((pres >> s2) & 1u);
    // X = the largest exponent among the present pairs with a non-zero product (a max over the row's 16 lanes)
    int X = (both && pm > 0.0) ? pe : -(1 << 30);
    {
        int o;
        o = __builtin_amdgcn_update_dpp(X, X, 0x128, 0xF, 0xF, false); X = o > X ? o : X;   // row_ror:8
        o = __builtin_amdgcn_update_dpp(X, X, 0x124, 0xF, 0xF, false); X = o > X ? o : X;   // row_ror:4
        o = __builtin_amdgcn_update_dpp(X, X, 0x122, 0xF, 0xF, false); X = o > X ? o : X;   // row_ror:2
        o = __builtin_amdgcn_update_dpp(X, X, 0x121, 0xF, 0xF, false); X = o > X ? o : X;   // row_ror:1
    }
    if (X == -(1 << 30) || all_zeros) X = 0;
    if constexpr (SPLIT) {
        // the column's sample record (pg_device.h): the fifteen scaled entries in tri_local order, then {X, flags} — put together
        // in the row's LDS slice (entry of local pair (la, lb) at tri_local(la, lb)), stored by the row's sixteen lanes as
        // 128 contiguous bytes; a column with a present pair below fp64's normal range under X also keeps its
        // (mantissa, exponent) products in cprec
        double* ent = (double*)rec;   // [16] entries, then [16] mantissas, [16] exponents (as ints)
        ent[l] = 0.0; ent[16 + l] = 0.0; ((int*)(ent + 32))[l] = 0;
        const bool lowp = both && !all_zeros && pm > 0.0 && pe - X < -1021;
        const bool precise = row16_ballot(lowp, grp) != 0u;
        wave_sync_lds();
        if (both) {
            const uint32_t la = local_of(s1), lb = local_of(s2);   // s1 <= s2  =>  la <= lb
            const uint32_t ti = tri_local(la, lb);
            ent[ti] = all_zeros ? 1.0 : ((pm > 0.0) ? ldexp(pm, pe - X) : pm);
            ent[16 + ti] = all_zeros ? 0.5 : pm;
            ((int*)(ent + 32))[ti] = all_zeros ? 1 : pe;
        }
        wave_sync_lds();
        const uint32_t flags = (all_zeros ? PG_SREC_FLAG_ALLZERO : 0u) | (precise ? PG_SREC_FLAG_PRECISE : 0u);
        double out = ent[l];
        if (l == 15u) out = __longlong_as_double((long long)((unsigned long long)(uint32_t)X | ((unsigned long long)flags << 32)));
        ((double*)((unsigned char*)dc.frec + (size_t)col * PG_SREC2_BYTES))[l] = out;
        if (precise) {
            ((double*)(dc.cprec + (size_t)col * PG_CPREC_BYTES))[l] = ent[16 + l];
            ((int*)(dc.cprec + (size_t)col * PG_CPREC_BYTES + 128u))[l] = ((int*)(ent + 32))[l];
        }
        wave_sync_lds();   // (the slots are rewritten by the wave's next unit)
        return;
    }
    if (both) {
        const uint32_t la = local_of(s1), lb = local_of(s2);   // s1 <= s2  =>  la <= lb
        double val;
        if (all_zeros) val = 1.0;                           // emissionprobabilitycomputer.cpp:31-34
        else val = (pm > 0.0) ? ldexp(pm, pe - X) : pm;     // 0 (or NaN) stays
        ((double*)(rec + PG_REC_E))[la * PG_ESTRIDE + lb] = val;
        ((double*)(rec + PG_REC_E))[lb * PG_ESTRIDE + la] = val;
        // the unscaled product as (mantissa, exponent): what a finished posterior bin is multiplied with
        const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
        unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
        const uint32_t pi = tri_n(la, lb, pn);
        ((double*)vp)[pi] = all_zeros ? 0.5 : pm;
        ((int*)(vp + (size_t)NP * 8u))[pi] = all_zeros ? 1 : pe;
    }
    if (l < 8u) {   // local allele l -> allele slot: the l-th present slot
        uint32_t sl = 0, seen = 0;
#pragma unroll
        for (int q = 0; q < PG_AMAX; ++q) { if (((pres >> q) & 1u) && seen == l) sl = (uint32_t)q; seen += (pres >> q) & 1u; }
        if (l < 6u) ((uint16_t*)(rec + PG_REC_LOCAL_SLOT))[l] = (uint16_t)(l < n_local ? sl : 0u);
    }
    if (l == 8u) {
        *(uint32_t*)(rec + PG_REC_VARIANT) = v;
        *(int32_t*)(rec + PG_REC_EXP) = X;
        rec[PG_REC_NLOCAL] = (unsigned char)n_local;
        rec[PG_REC_FLAGS] = all_zeros ? PG_REC_FLAG_ALLZERO : 0;
        rec[PG_REC_FLAGS + 1] = 0; rec[PG_REC_FLAGS + 2] = 0;
        *(uint32_t*)(rec + PG_REC_WIDE_IDX) = PG_WIDE_NONE;
        *(uint32_t*)(rec + PG_REC_AUX) = dc.aux_idx ? dc.aux_idx[v] : PG_WIDE_NONE;
    }
    if (l == 9u) {
        ((unsigned long long*)(rec + PG_REC_BITS1))[0] = ones;
        ((unsigned long long*)(rec + PG_REC_BITS1))[1] = 0ull;
    }
    wave_sync_lds();
    {   // copy-out: the object's 16 lanes move RB / 16 pieces of 16 bytes, consecutive lanes consecutive pieces
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const f64x2* src = (const f64x2*)rec;
        f64x2* dst = (f64x2*)(dc.vrec + (size_t)v * dc.RB);
        for (uint32_t p = l; p < dc.RB / 16u; p += 16u) dst[p] = src[p];
    }
    wave_sync_lds();   // (the slots are rewritten by the wave's next unit)
}
__global__ __launch_bounds__(256) void k_prep_m4(const DevContig* __restrict__ contigs, DevTable tab) {
    const DevContig& dc = contigs[blockIdx.y];
    if (!dc.prep_m4 || dc.split) return;
#pragma unroll 1
    for (uint32_t r = 0; r < (uint32_t)PG_VREP; ++r) prep_m4_unit<false>(dc, tab, blockIdx.x * (uint32_t)PG_VREP + r);
}
// ... of the split path: the same products of the chain's 3 .. PG_AMAX-allele objects, into the column's sample record (pg_split.h)
__global__ __launch_bounds__(256) void k_prep_s_m4(const DevContig* __restrict__ contigs, DevTable tab) {
    const DevContig& dc = contigs[blockIdx.y];
    if (!dc.prep_m4 || dc.split != 2u) return;
#pragma unroll 1
    for (uint32_t r = 0; r < (uint32_t)PG_VREP; ++r) prep_m4_unit<true>(dc, tab, blockIdx.x * (uint32_t)PG_VREP + r);
}

__global__ __launch_bounds__(64) void k_emission_single(const DevContig* __restrict__ contigs, DevTable tab,
                                                        uint32_t v, double* out_m, int* out_e) {
    __shared__ double s_m[64 * 3];
    __shared__ int s_e[64 * 3];
    const DevContig& dc = contigs[0];
    const uint32_t lane = threadIdx.x;
    const uint32_t A = dc.allele_off[v + 1] - dc.allele_off[v];
    const uint32_t P = A * (A + 1) / 2;
    for (uint32_t base = 0; base < P; base += 64) {
        const bool active = base + lane < P;
        uint32_t s1 = 0, s2 = 0;
        if (active) decode_pair(base + lane, A, s1, s2);
        double pm; int pe;
        emission_pair_products(dc, tab, v, lane, active, s1, s2, s_m, s_e, pm, pe);
        if (active) {
            out_m[s1 * A + s2] = pm; out_e[s1 * A + s2] = pe;
            out_m[s2 * A + s1] = pm; out_e[s2 * A + s1] = pe;
        }
    }
}

// ------------------------------------------------------------------------------------------
//  k_compact : kept[] -> col_variant[], n_cols       (one 1024-thread workgroup per contig)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_compact(const DevContig* __restrict__ contigs) {
    __shared__ uint32_t s_cnt[1024];
    const DevContig& dc = contigs[blockIdx.x];
    const uint32_t tid = threadIdx.x, V = dc.V;
    const uint32_t chunk = (V + 1023u) / 1024u;
    const uint32_t lo = tid * chunk < V ? tid * chunk : V;
    const uint32_t hi = lo + chunk < V ? lo + chunk : V;
    uint32_t cnt = 0;
    for (uint32_t v = lo; v < hi; ++v) cnt += dc.kept[v] ? 1u : 0u;
    s_cnt[tid] = cnt;
    __syncthreads();
    // inclusive Hillis-Steele scan
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t add = tid >= off ? s_cnt[tid - off] : 0u;
        __syncthreads();
        s_cnt[tid] += add;
        __syncthreads();
    }
    uint32_t pos = s_cnt[tid] - cnt;
    uint32_t* col_of = const_cast<uint32_t*>(dc.col_of);
    for (uint32_t v = lo; v < hi; ++v) {
        if (col_of) col_of[v] = dc.kept[v] ? pos : PG_COL_NONE;
        if (dc.kept[v]) dc.col_variant[pos++] = v;
    }
    if (tid == 1023) *dc.n_cols = s_cnt[1023];
}

// ------------------------------------------------------------------------------------------
//  Li-Stephens constants (reference src/transitionprobabilitycomputer.cpp:14-18).
//  With r = exp(-d/H) and q = (1-exp(-d/H))/H = -expm1(-d/H)/H :
//     t0 x + t1 (R_i+C_j-2x) + t2 (S-R_i-C_j+x) = r^2 x + q r (R_i+C_j) + q^2 S
//  (1 - exp(-x) as the reference's 80-bit arithmetic rounds it: pg_devmath.h)
// ------------------------------------------------------------------------------------------
DEVI void transition_consts(double d, uint32_t H, int uniform, double& c0, double& c1, double& c2, double& kappa) {
    if (uniform) { c0 = 0.0; c1 = 0.0; c2 = 1.0; }
    else {
        const double x = d / (double)H;
        const double r = exp(-x);
        const double q = one_minus_exp_neg_like_reference(x) / (double)H;
        c0 = r * r; c1 = q * r; c2 = q * q;
    }
    const double h = (double)H;
    kappa = c0 + 2.0 * h * c1 + h * h * c2;  // sum of one transition row-pair: sum(A w A^T) = kappa * sum(w)
}

__global__ __launch_bounds__(64) void k_transition_single(double d, uint32_t H, int uniform, double* out3) {
    if (threadIdx.x == 0) {
        transition_probs_f64(d, H, uniform, out3[0], out3[1], out3[2]);
    }
}

// ------------------------------------------------------------------------------------------
//  k_records : gather variant records into column order, fill transition constants
// ------------------------------------------------------------------------------------------
// Chains whose sweeps read only the compact records (triangle chains on k_sweep_lean / k_sweep_lean2, at least two
// columns) get nothing else: one THREAD per column forms the transition constants and gathers the 64-byte record;
// k_bins reads the variant record itself.  (The column-order copy of the full 448-byte records, one wave per column
// with 64 lanes computing the same exp(), was 4.6 ms of the cohort's 80.)
// The same for lean chains of chunked jobs: their sweeps run on k_sweep_lean and k_post reads the variant record.
// And for the 16-path chains whose two phases both run on k_sweep_small16 (DevContig::small == 2).
DEVI bool compact_records_only(const DevContig& dc, uint32_t C) {
    return ((dc.tri == 2u || dc.small == 2u || dc.smallx == 2u) && C >= 2u) || (dc.lean && dc.tri == 0u && dc.chunk_cols > 0u);
}

DEVI void records_unit(const DevContig& dc, uint32_t unit) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t C = *dc.n_cols;
    if (dc.split) return;   // the split path (pg_split.h): its records are written in column order by the emission kernels
    if (compact_records_only(dc, C) && !dc.smallx) {
        const uint32_t c = unit * 256u + threadIdx.x;
        if (c >= C) return;
        const uint32_t v = dc.col_variant[c];
        const uint64_t* src = (const uint64_t*)(dc.vrec + (size_t)v * dc.RB);
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, kappa = 0.0;
        if (c > 0) {
            const double d = (double)(dc.pos[v] - dc.pos[dc.col_variant[c - 1]]) * dc.dist_scale;
            transition_consts(d, dc.H, dc.uniform, c0, c1, c2, kappa);
        }
        const uint64_t* E = src + PG_REC_E / 8;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        f64x2* out = (f64x2*)(dc.frec + (size_t)c * 8);
        out[0] = f64x2{c0, c1};
        out[1] = f64x2{c2, kappa};
        out[2] = f64x2{__longlong_as_double((long long)E[0]), __longlong_as_double((long long)E[1])};
        out[3] = f64x2{__longlong_as_double((long long)E[PG_ESTRIDE + 1]), __longlong_as_double((long long)src[PG_REC_BITS1 / 8])};
        return;
    }
    // Full records: a wave takes 64 consecutive columns.  First every LANE forms the transition constants of ONE of them
    // (one wave per column had all 64 lanes evaluate the same exp(): 17.7 ms for 31.5 M columns of 16-path chains), then
    // the wave copies the 64 records into column order, the four constants of a column coming out of its LDS slice.
    __shared__ double s_c[4][64][4];
    __shared__ uint32_t s_v[4][64];
    const uint32_t cbase = (unit * 4u + wave) * 64u;
    if (cbase >= C) return;
    {
        const uint32_t c = cbase + lane;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, kappa = 0.0;
        uint32_t v = 0;
        if (c < C) {
            v = dc.col_variant[c];
            if (c > 0) {
                const double d = (double)(dc.pos[v] - dc.pos[dc.col_variant[c - 1]]) * dc.dist_scale;
                transition_consts(d, dc.H, dc.uniform, c0, c1, c2, kappa);
            }
        }
        s_c[wave][lane][0] = c0; s_c[wave][lane][1] = c1; s_c[wave][lane][2] = c2; s_c[wave][lane][3] = kappa;
        s_v[wave][lane] = v;
        // a wide column of a chain whose bins k_bins_wide forms: onto the chain's list (any order: every column is its own work item)
        if (c < C && dc.wcols && (dc.vrec[(size_t)v * dc.RB + PG_REC_FLAGS] & PG_REC_FLAG_WIDE)) dc.wcols[atomicAdd(dc.n_wcols, 1u)] = c;
    }
    wave_sync_lds();
    // the copy: the wave's n records are n * RB/16 16-byte pieces, contiguous on the destination side (1 KB per wave
    // instruction); piece p belongs to column p / (RB/16) (exact by a 32-bit reciprocal: p < 5632, 24 <= RB/16 <= 88)
    const uint32_t n = C - cbase < 64u ? C - cbase : 64u;
    const uint32_t ppr = dc.RB / 16u, inv = (uint32_t)((0x100000000ull + ppr - 1u) / ppr);
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2* dst = (f64x2*)(dc.colrec + (size_t)cbase * dc.RB);
    if (!compact_records_only(dc, C))   // (x chains of fused jobs: only the 320-byte records below)
    for (uint32_t p = lane; p < n * ppr; p += 64u) {
        const uint32_t q = __umulhi(p, inv), w = p - q * ppr;
        const uint32_t v = s_v[wave][q];
        f64x2 val = ((const f64x2*)(dc.vrec + (size_t)v * dc.RB))[w];
        if (w < 2u) val = f64x2{s_c[wave][q][2u * w], s_c[wave][q][2u * w + 1u]};
        dst[p] = val;
    }
    if (dc.smallx) {
        // the 192-byte records of k_sweep_small16x (pg_small16x.h): pieces 0..7 = the fifteen entries E(a, b), a <= b, of the 6 x 6
        // table in tri_local order (a WIDE column: piece 0 = its sixteen raw local alleles, the rest zero), 8 = header {nlocal |
        // flags << 8, wide entry, aux slot, 0}, 9-10 = constants, 11 = table-row offset (a + 1) * 48 of every path's allele
        // (0: phantom, and every path of a wide column); the wave's n records are n * 12 pieces, contiguous on the destination side
        f64x2* xd = (f64x2*)((unsigned char*)dc.frec + (size_t)cbase * 192u);
        for (uint32_t p = lane; p < n * 12u; p += 64u) {
            const uint32_t q = p / 12u, w = p - q * 12u;
            const unsigned char* src = dc.vrec + (size_t)s_v[wave][q] * dc.RB;
            const uint32_t nlf = (uint32_t)src[PG_REC_NLOCAL] | ((uint32_t)src[PG_REC_FLAGS] << 8);
            const bool widec = (nlf & 0x200u) != 0u;
            f64x2 val = f64x2{0.0, 0.0};
            if (w < 8u) {
                if (widec) { if (w == 0u) val = *(const f64x2*)(src + PG_REC_ALLELES); }
                else {
                    auto entry = [&](uint32_t e) {   // table entry e in tri_local order (15: padding)
                        if (e >= 15u) return 0.0;
                        const uint32_t a = (e >= 5u ? 1u : 0u) + (e >= 9u ? 1u : 0u) + (e >= 12u ? 1u : 0u) + (e >= 14u ? 1u : 0u);
                        const uint32_t b = e - (a * (uint32_t)PG_AMAX - a * (a - 1u) / 2u) + a;
                        return ((const double*)(src + PG_REC_E))[a * PG_ESTRIDE + b];
                    };
                    val = f64x2{entry(2u * w), entry(2u * w + 1u)};
                }
            } else if (w == 9u) val = f64x2{s_c[wave][q][0], s_c[wave][q][1]};
            else if (w == 10u) val = f64x2{s_c[wave][q][2], s_c[wave][q][3]};
            else {
                uint4 u;
                if (w == 8u) u = uint4{nlf, *(const uint32_t*)(src + PG_REC_WIDE_IDX), *(const uint32_t*)(src + PG_REC_AUX), 0u};
                else {
                    u = *(const uint4*)(src + PG_REC_ALLELES);
                    auto f = [&](uint32_t x) {   // four allele bytes -> four row offsets
                        uint32_t r = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const uint32_t a = (x >> (8 * b)) & 0xFFu;
                            r |= ((a < (uint32_t)PG_AMAX && !widec) ? (a + 1u) * (uint32_t)(PG_ESTRIDE * 8) : 0u) << (8 * b);
                        }
                        return r;
                    };
                    u = uint4{f(u.x), f(u.y), f(u.z), f(u.w)};
                }
                val = *(const f64x2*)&u;
            }
            xd[p] = val;
        }
    }
    if (dc.lean || dc.small) {
        // compact records of the lean sweeps: {c0, c1, c2, kappa, E'00, E'01, E'11, bits1}, eight columns per wave instruction
        for (uint32_t p = lane; p < n * 8u; p += 64u) {
            const uint32_t q = p >> 3, f = p & 7u;
            const uint64_t* src = (const uint64_t*)(dc.vrec + (size_t)s_v[wave][q] * dc.RB);
            uint64_t val;
            if (f < 4u) val = (uint64_t)__double_as_longlong(s_c[wave][q][f]);
            else if (f == 4u) val = src[PG_REC_E / 8];
            else if (f == 5u) val = src[PG_REC_E / 8 + 1];
            else if (f == 6u) val = src[PG_REC_E / 8 + PG_ESTRIDE + 1];
            else val = src[PG_REC_BITS1 / 8];
            ((uint64_t*)dc.frec)[(size_t)cbase * 8 + p] = val;
        }
    }
}

__global__ __launch_bounds__(256) void k_records(const DevContig* __restrict__ contigs) {
    const DevContig& dc = contigs[blockIdx.y];
    records_unit(dc, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
//  chain kernels (meet in the middle)
//
//  One chain = one (contig, path subset).  The recursion is strictly sequential over its C
//  columns, so the only exact parallelism inside a chain is the two directions: k_sweep runs
//  the forward half-chain and the backward half-chain as two workgroups at once.
//    phase 1 : forward  computes columns 0 .. mid-1   and stores v'_t   into slot t   (t <  mid)
//              backward computes columns C-1 .. mid   and stores beta'_t into slot t  (t >= mid)
//    phase 2 : forward  continues mid .. C-1, gets beta'_t from slot t, emits posterior partials
//              backward continues mid-1 .. 0, gets v'_t    from slot t, emits posterior partials
//    phase 3 : (instead of 2, few chains) the same continuation in chunks that only store their
//              columns into a scratch; k_post forms the posteriors of a finished chunk
//  In the fused mode every column is written once and read once (the 16*H^2 algorithmic bytes), by
//  workgroups on different CUs; the wall time of a chain is C/2 + C/2 column steps instead of 2C.
//  Hand-overs (at `mid`, between chunks) go through stored columns, i.e. through kernel
//  boundaries — no inter-workgroup flags, nothing placement dependent.
//
//  Workgroup = T compute threads + LOADER waves (the last one or two waves; none at HP = 128).
//  VMEM counters are per wave, so the split keeps every wait off the recursion's critical path:
//  compute waves only STORE (columns, partials); the loader waves only issue LDS-DMA (column
//  records, partner columns: HBM -> LDS without registers) and count its completion.  The
//  per-column workgroup barrier orders LDS only (no vmcnt wait).  All global pointers are
//  address_space(1) so that accesses are global_* (FLAT ops would also tick lgkmcnt and put the
//  HBM latency back on the LDS waits).
//
//  Scaling.  The reference normalises every column by its sum (a division on the critical
//  path).  Here a column is rescaled by the exact power of two 2^-e, e = exponent(sum), and
//  the mantissa m = sum * 2^-e in [0.5,1) goes to a side array; the true
//  alpha_hat_c * fsum_c is the stored column divided by m (k_bins does that division, off the
//  chain).  Same for the backward column.  Results are the reference's values up to fp64
//  rounding; no drift because every step renormalises to within a factor of 2.
// ------------------------------------------------------------------------------------------
// (In-kernel cycle counters and the masks of the timing experiments — kExp, kLeanExp, ... — live in pg_experiments.h: all
// zero / false in the product build.)
#define GAS __attribute__((address_space(1)))
typedef GAS double gdouble;
typedef GAS const double gcdouble;
typedef GAS const unsigned long long gcu64;
typedef GAS unsigned char gu8;
typedef double v2f64 __attribute__((ext_vector_type(2)));  // native vector type (address-space qualifiable)
typedef GAS v2f64 gdouble2;
typedef GAS const v2f64 gcdouble2;

#ifndef PG_PARK_ROWS
#define PG_PARK_ROWS 16
#endif
template <int HP, int R, bool LD = (HP < 128)>
struct ChainCfg {
    static constexpr int T = HP * HP / R;      // compute threads
    // HP = 128 keeps its 8 compute waves at 2 waves/SIMD (256 VGPRs): a 9th wave would cut the
    // register budget to 168 and spill, so there wave 0 does the loader's work inline.  So do the store-only phases at
    // HP = 32 (sweep_has_loader): their loader wave moved nothing but the column records, and without it a half-chain is two
    // waves instead of three — eight half-chains per CU instead of five, in a kernel whose waves wait three quarters of their cycles.
    static constexpr bool LOADER = LD;
    // loader waves: a wave can have at most 63 counted transfers in flight and a 32 KB column is 32
    // of them, so two columns deep (what hides the ~1.1 us DMA latency) needs two waves at HP = 64
    static constexpr int NLOAD = LOADER ? (HP >= 64 ? 2 : 1) : 0;
    static constexpr int TT = T + 64 * NLOAD;
    static constexpr int NRG = HP / R;
    static constexpr int NW = T / 64;          // compute waves
    static constexpr bool UNI = HP >= 64;
    static constexpr int RB = (PG_REC_ALLELES + HP + 63) & ~63;
    static constexpr int WORDS = RB / 8;
    // phase 2 without the LDS ring (HP = 128): a thread holds 32 rows of its own column AND of the prefetched partner
    // column; with the rest of the step that is ~22 doubles more than the 256 registers two waves per SIMD leave, and
    // the compiler spilled them to scratch inside the state loop (43 KB each way per 128 KB column, and every reload a
    // trip to memory).  The last PARK rows of the thread's OWN column live in LDS instead (read and rewritten by the
    // thread itself: no barrier involved), PARK/2 16-byte slots per thread in dynamic LDS.
    static constexpr int PARK = (!LOADER && R > 16) ? PG_PARK_ROWS : 0;
    // columns of fused jobs carry data in their first DevContig::live rows and lanes only (see there)
    static constexpr bool SHORT = HP == 32;
    static_assert(T % 64 == 0 && TT <= 1024, "bad workgroup size");
    static_assert(WORDS <= 64, "record must fit one wave-wide 8-byte load");
    static_assert(64 % R == 0 || R % 64 == 0, "row groups must not straddle 64-column blocks");
};

template <int HP, int PHASE>
constexpr bool sweep_has_loader() { return HP < 128 && !(HP == 32 && PHASE != 2); }

template <int HP, int R>
struct ChainShared {
    using Cfg = ChainCfg<HP, R>;
    unsigned char rec[8][Cfg::RB] __attribute__((aligned(16)));  // ring of 8 column records
    double psum[2][Cfg::NRG][HP];  // per row group partial column sums, double buffered by column parity
    double u[Cfg::NW][Cfg::UNI ? 64 : HP] __attribute__((aligned(16)));  // per-wave copy of the u vector
};

// workgroup barrier that orders LDS traffic only (global stores/loads stay in flight)
DEVI void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
DEVI void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// x = m * 2^e with m in [0.5,1)  (x > 0, finite)
DEVI int exponent_of(double x) { return __builtin_amdgcn_frexp_exp(x); }

// general emission lookup: E[a_i][a_j].  Narrow columns: the (PG_AMAX+1)^2 table inside the LDS
// record, phantom paths (255) hit its zero row/column PG_AMAX.  Wide columns (more than PG_AMAX
// alleles on the selected paths): the PG_WIDE_MAX^2 table of the variant's wide entry in global
// memory (L2-resident for the duration of the column; rare, so no staging).
struct EmSrc {
    const double* wide;  // nullptr: narrow
    uint32_t ajr;        // this thread's column allele, raw (PG_PHANTOM for padding paths)
    uint32_t wn;         // wide: n_local (row/column n_local of the table is zero; row stride n_local + 1)
};
DEVI double emission_narrow(const unsigned char* rec, uint32_t i, const EmSrc& es) {
    const uint32_t air = rec[PG_REC_ALLELES + i];
    const uint32_t ai = air > PG_AMAX ? PG_AMAX : air, aj = es.ajr > PG_AMAX ? PG_AMAX : es.ajr;
    return ((const double*)(rec + PG_REC_E))[ai * PG_ESTRIDE + aj];
}
DEVI double emission_wide(const unsigned char* rec, uint32_t i, const EmSrc& es) {
    const uint32_t air = rec[PG_REC_ALLELES + i];
    const uint32_t ai = air > es.wn ? es.wn : air, aj = es.ajr > es.wn ? es.wn : es.ajr;
    return ((gcdouble*)es.wide)[ai * (es.wn + 1u) + aj];
}
DEVI double emission_at(const unsigned char* rec, uint32_t i, const EmSrc& es) {
    return es.wide ? emission_wide(rec, i, es) : emission_narrow(rec, i, es);
}
DEVI uint32_t col_allele(const unsigned char* rec, uint32_t j) {
    uint32_t aj = rec[PG_REC_ALLELES + j];
    return aj > PG_AMAX ? PG_AMAX : aj;
}

// biallelic fast path (<= 2 local alleles, no phantom paths): e(i,j) = bit_i ? eB : eA with
// eA = E[0][a_j], eB = E[1][a_j] per lane and the row bits uniform per wave (UNI).
struct FastE {
    double eA, eB;
    uint32_t rowbits;
};
template <bool UNI>
DEVI FastE fast_setup(const unsigned char* rec, uint32_t j, uint32_t i0) {
    const double* E = (const double*)(rec + PG_REC_E);
    const double E00 = E[0], E01 = E[1], E11 = E[PG_ESTRIDE + 1];
    const unsigned long long* bits = (const unsigned long long*)(rec + PG_REC_BITS1);
    const uint32_t aj = (uint32_t)(bits[j >> 6] >> (j & 63u)) & 1u;
    uint32_t rbits = (uint32_t)(bits[i0 >> 6] >> (i0 & 63u));
    if (UNI) rbits = __builtin_amdgcn_readfirstlane(rbits);
    FastE f;
    f.eA = aj ? E01 : E00;
    f.eB = aj ? E11 : E01;
    f.rowbits = rbits;
    return f;
}

// Everything a recursion step needs from a column record, decoded into registers one column
// ahead of use so that the LDS latency of the record never sits on the chain.
struct RecInfo {
    double c0, c1, c2, kappa;
    FastE fe;
    EmSrc em;
    uint32_t nl;
    bool fast;
};
template <bool UNI>
DEVI RecInfo decode_record(const unsigned char* rec, uint32_t j, uint32_t i0, bool full, const uint8_t* wide_base) {
    RecInfo r;
    r.c0 = *(const double*)(rec + PG_REC_C0);
    r.c1 = *(const double*)(rec + PG_REC_C1);
    r.c2 = *(const double*)(rec + PG_REC_C2);
    r.kappa = *(const double*)(rec + PG_REC_KAPPA);
    r.nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[PG_REC_NLOCAL]);  // same record for every lane: keep it scalar
    r.fe = fast_setup<UNI>(rec, j, i0);
    r.em.ajr = rec[PG_REC_ALLELES + j];
    r.em.wide = nullptr;
    r.em.wn = r.nl;
    if (r.nl > PG_AMAX) {  // wide column (scalar branch: nl is uniform)
        const uint32_t woff = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t*)(rec + PG_REC_WIDE_IDX));
        r.em.wide = (const double*)(wide_base + (size_t)woff * 16u);
    }
    r.fast = full && r.nl <= 2;
    return r;
}

// Phase-2 partner columns reach the compute waves through an LDS ring filled by the loader wave
// with LDS-DMA (global_load_lds_dwordx4: HBM -> LDS, no VGPRs), two columns ahead of use.
#define LAS __attribute__((address_space(3)))
// Partner-column ring: PG_RING_SLOTS column slots in dynamic LDS, prefetch distance SLOTS-1 columns.
// 4 slots (128 KB at HP = 64) give every DMA two full steps to land — what a lone workgroup per CU
// needs; 2 slots (64 KB) let two workgroups share a CU when hundreds of chains are resident.
// DevContig::tri == 2: phase 2 of the chain runs on k_sweep_lean2 (the default for triangle chains); == 1 (PG_LEAN2=0):
// on the general kernel's triangle ring (cross-check of the two phase-2 implementations)
#ifndef PG_RING_SLOTS
#define PG_RING_SLOTS 4
#endif
static constexpr int kRingSlots = PG_RING_SLOTS, kRingDist = PG_RING_SLOTS - 1;
static_assert(kRingSlots == 2 || kRingSlots == 4, "ring slots: power of two");
// loader wave `part` of NPARTS moves its share of the column (1 KB per wave-instruction)
template <int HP, int NPARTS>
DEVI void dma_column(const gdouble* cols, int64_t c, int64_t C, LAS unsigned char* ring, uint32_t lane, uint32_t part) {
    if (c < 0 || c >= C) return;
    constexpr uint32_t COLB = HP * HP * 8u, SHARE = COLB / (NPARTS > 0 ? NPARTS : 1);
    static_assert(SHARE % 1024u == 0, "column share must be a whole number of wave transfers");
    const GAS char* g = (const GAS char*)(cols + (size_t)c * HP * HP) + part * SHARE + lane * 16u;
    LAS unsigned char* l = ring + (uint32_t)(c & (kRingSlots - 1)) * COLB + part * SHARE;  // wave-uniform base; lane l lands at +16*l
#pragma unroll 4
    for (uint32_t q = 0; q < SHARE / 1024u; ++q)
        __builtin_amdgcn_global_load_lds((const GAS void*)(g + q * 1024u), (LAS void*)(l + q * 1024u), 16, 0, 0);
}
// ... its first NT transfers only, and of those the lanes whose column lies below `live` (ChainCfg::SHORT: rows and lanes
// from `live` on are neither stored nor fetched; their part of the ring is zeroed once by the compute waves)
template <int HP, int NPARTS, int NT>
DEVI void dma_column_live(const gdouble* cols, int64_t c, int64_t C, LAS unsigned char* ring, uint32_t lane, uint32_t part, uint32_t live) {
    if (c < 0 || c >= C) return;
    constexpr uint32_t COLB = HP * HP * 8u, SHARE = COLB / (NPARTS > 0 ? NPARTS : 1);
    static_assert((uint32_t)NT <= SHARE / 1024u && HP <= 64, "a transfer is 64 lanes: whole rows of pairs");
    const GAS char* g = (const GAS char*)(cols + (size_t)c * HP * HP) + part * SHARE + lane * 16u;
    LAS unsigned char* l = ring + (uint32_t)(c & (kRingSlots - 1)) * COLB + part * SHARE;
    if ((lane % (uint32_t)HP) < live) {
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)NT; ++q)
            __builtin_amdgcn_global_load_lds((const GAS void*)(g + q * 1024u), (LAS void*)(l + q * 1024u), 16, 0, 0);
    }
}
// column record -> its LDS slot by DMA as well (lanes < RB/16 move 16 B each): the loader wave then
// has no register-returning loads at all and every completion is an explicit counted vmcnt
template <int RB>
DEVI void dma_record(gcu64* colrec, int64_t c, int64_t C, LAS unsigned char* slots, uint32_t lane) {
    if (c < 0 || c >= C) return;
    if (lane < (uint32_t)(RB / 16)) {
        const GAS char* g = (const GAS char*)colrec + (size_t)c * RB + lane * 16u;
        __builtin_amdgcn_global_load_lds((const GAS void*)g, (LAS void*)(slots + (uint32_t)(c & 7) * RB), 16, 0, 0);
    }
}
DEVI void wait_vmem_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0) only
// wait until at most N vector-memory operations are outstanding (loads complete in order)
template <int N>
DEVI void wait_vmem_keep() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);
}
template <int HP, int R>
DEVI void ring_read(const unsigned char* ring, int64_t c, const uint32_t i0, const uint32_t j, double (&v)[R]) {
    if (kExp & 32u) {
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = 1.0;
        return;
    }
    const v2f64* slot = (const v2f64*)(ring + (size_t)(c & (kRingSlots - 1)) * (HP * HP * 8u)) + (size_t)(i0 >> 1) * HP + j;
#pragma unroll
    for (int k = 0; k < R; k += 2) { const v2f64 t = slot[(size_t)(k >> 1) * HP]; v[k] = t.x; v[k + 1] = t.y; }
}

// ---- triangle ring (DevContig::tri, HP = 64): the stored half of a column is 1152 16-byte units (row pair q holds
// lanes 8 (q >> 2) .. 63: whole 128-byte lines, see k_sweep_lean's put_pair), enumerated row pair by row pair.  The
// stored column and the ring keep them COMPACT — one contiguous 18 KB run instead of 32 KB — so 8 ring slots fit
// where 4 did and every LDS-DMA transfer is a full 1 KB: 9
// transfers per loader and column instead of 16, seven columns in flight instead of three.  Phase 2 of many
// resident chains is bound by the bytes a CU has in flight, not by HBM bandwidth.
#ifndef PG_TRI_SLOTS
#define PG_TRI_SLOTS 8
#endif
static constexpr int kTriSlots = PG_TRI_SLOTS, kTriDist = PG_TRI_SLOTS - 1;  // (3 slots: two workgroups share a CU's LDS)
static constexpr int kTriKeep0 = (kTriDist - 1 < 2 ? kTriDist - 1 : 2) * (1 + 9), kTriKeep1 = (kTriDist - 1) * 9;
static_assert(kTriKeep0 < 64 && kTriKeep1 < 64 && kTriSlots >= 2, "vmcnt is 6 bits");
static constexpr uint32_t kTriUnits = 1152u, kTriSlotB = kTriUnits * 16u;  // 18432
static constexpr uint32_t kTriRingB = (uint32_t)kTriSlots * kTriSlotB + 16u;  // + one unit of zeros (what lies below the diagonal)
DEVI uint32_t tri_unit_of(uint32_t q /*row pair*/, uint32_t lane /* >= 8 (q >> 2) */) {
    const uint32_t g = q >> 2;
    return 256u * g - 16u * g * (g - 1u) + (q & 3u) * (64u - 8u * g) + (lane - 8u * g);
}
// loader `part` (of 2) moves transfers 9 part .. 9 part + 8 of the column's 18
DEVI void dma_column_tri(const gdouble* cols, size_t stride, int64_t c, int64_t C, LAS unsigned char* ring, uint32_t part, uint32_t lane) {
    if (c < 0 || c >= C) return;
    const GAS char* g = (const GAS char*)(cols + (size_t)c * stride) + part * 9u * 1024u + lane * 16u;  // the stored half is compact
    LAS unsigned char* l = ring + (uint32_t)((uint64_t)c % (uint32_t)kTriSlots) * kTriSlotB + part * 9u * 1024u;
#pragma unroll
    for (uint32_t n = 0; n < 9u; ++n)
        __builtin_amdgcn_global_load_lds((const GAS void*)(g + n * 1024u), (LAS void*)(l + n * 1024u), 16, 0, 0);
}
// this thread's eight units of a column: LDS byte offsets inside a slot; units below the diagonal read the zero unit
// Triangle STORE of one row pair (rows i0 + 2q, i0 + 2q + 1 of column j; i0 a multiple of 16) without per-pair registers — the
// general kernel has none to spare (k_sweep_lean_tri keeps factors and units in registers).  With d = j - i0: the pair's elements
// take the factor 0 below the diagonal, 1/2 on it, 1 above; the pair is not stored at all when its unit lies in a 128-byte line
// left of the stored part (j < 8 (row pair / 4)); its place in the compact column is base[g] + (q & 3) (64 - 8 g), g = row pair / 4.
struct TriStore {
    int d;                 // j - i0
    uint32_t base0, base1; // unit of row pair i0 / 2 (q = 0) and i0 / 2 + 4 (q = 4) for this lane
    uint32_t g0;           // (i0 / 2) / 4, wave-uniform
};
DEVI TriStore tri_store_setup(uint32_t i0, uint32_t j) {
    TriStore t;
    t.d = (int)j - (int)i0;
    t.g0 = i0 >> 3;
    const uint32_t g0 = t.g0, g1 = g0 + 1u;
    t.base0 = 256u * g0 - 16u * g0 * (g0 - 1u) + (j - 8u * g0);
    t.base1 = 256u * g1 - 16u * g1 * (g1 - 1u) + (j - 8u * g1);
    return t;
}
DEVI void tri_store_pair(const TriStore& t, gdouble2* col, int Q /* a constant once the caller's loop is unrolled */, double a, double b) {
    const int lo = Q < 4 ? 0 : 8;   // (r0 & ~7) - i0
    int d = t.d;
    uint32_t base = Q < 4 ? t.base0 : t.base1;
    asm volatile("" : "+v"(d), "+v"(base));   // (formed HERE, every time: hoisted out of the column loop the sixteen factors and eight units are forty registers)
    if (d < lo) return;               // the unit's line is not stored
    const double fa = d < 2 * Q ? 0.0 : (d == 2 * Q ? 0.5 : 1.0);
    const double fb = d <= 2 * Q ? 0.0 : (d == 2 * Q + 1 ? 0.5 : 1.0);
    const uint32_t g = t.g0 + (Q < 4 ? 0u : 1u);
    const uint32_t unit = base + (uint32_t)(Q & 3) * (64u - 8u * g);
    col[unit] = v2f64{a * fa, b * fb};
}
template <int R>
DEVI void tri_read_setup(uint32_t i0, uint32_t j, uint32_t (&loff)[R / 2], uint32_t& valid) {
    valid = 0;
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
        const uint32_t rp = (i0 >> 1) + (uint32_t)q;
        const bool ok = j >= 8u * (rp >> 2);
        loff[q] = ok ? tri_unit_of(rp, j) * 16u : 0u;
        valid |= (ok ? 1u : 0u) << q;
    }
}
template <int R>
DEVI void ring_read_tri(const unsigned char* ring, int64_t c, const uint32_t (&loff)[R / 2], uint32_t valid, double (&v)[R]) {
    const uint32_t slot = (uint32_t)((uint64_t)c % (uint32_t)kTriSlots) * kTriSlotB;
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
        const uint32_t off = ((valid >> q) & 1u) ? slot + loff[q] : (uint32_t)kTriSlots * kTriSlotB;
        const v2f64 t = *(const v2f64*)(ring + off);
        v[2 * q] = t.x; v[2 * q + 1] = t.y;
    }
}

// per-thread coordinates of a compute thread
struct ThreadPos {
    uint32_t tid, lane, wave, j, rg, i0, rb;
    uint32_t pe, pT;   // posterior partials: this thread's entry (PG_NO_ENTRY: none) and the entries per column and slot pair
};
#define PG_NO_ENTRY 0xFFFFFFFFu
// HP = 32: lanes l and l + 32 of a wave hold rows of the SAME column — their partials are added in registers (sum of the two
// halves in every lane) and only the lower half's lanes below DevContig::live write one: T / 2 entries per slot pair, of
// which k_bins fetches the real paths' only (2 KB -> 544 B per column and slot pair at 17 paths)
DEVI double fold32(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// Column-sum exchange.  Before the barrier every thread parks the sum of its R rows of its column;
// after it every thread adds the NRG partials of its column.  By symmetry of the column (v_ij = v_ji)
// the row sums equal the column sums, so the same vector serves u_i and u_j.  The TOTAL is reduced
// after the barrier, redundantly in every wave, from the column sums (each wave holds all of them
// across its lanes): that DPP reduction runs while the u vector makes its LDS round trip, instead
// of sitting in front of the barrier behind the last arithmetic of the step.
//   Cj   : sum of this thread's column
//   Crow : sum of the column whose index is (row block of this wave) + lane  (== Cj for HP <= 64)
//   Call : this lane's column sums added over all 64-column blocks (its wave sum is the total)
template <int HP, int R>
DEVI void read_colsums(const ChainShared<HP, R>& sh, uint32_t pb, const ThreadPos& p, double& Cj, double& Crow, double& Call) {
    using Cfg = ChainCfg<HP, R>;
    if (kExp & 64u) { Cj = Crow = Call = 1.0 / HP; return; }
    if constexpr (HP <= 64) {
        Cj = sh.psum[pb][0][p.j];
#pragma unroll
        for (int g = 1; g < Cfg::NRG; ++g) Cj += sh.psum[pb][g][p.j];
        Crow = Call = Cj;
    } else {
        static_assert(HP == 128, "two 64-column blocks");
        double ca = sh.psum[pb][0][p.lane], cb = sh.psum[pb][0][64 + p.lane];
#pragma unroll
        for (int g = 1; g < Cfg::NRG; ++g) { ca += sh.psum[pb][g][p.lane]; cb += sh.psum[pb][g][64 + p.lane]; }
        Cj = p.j >= 64u ? cb : ca;  // wave-uniform choices
        Crow = p.rb ? cb : ca;
        Call = ca + cb;
    }
}
template <int HP>
DEVI double total_sum(double Call) {
    if (kExp & 2u) return Call * 64.0;
    // HP < 64: lanes 0..HP-1 hold every column once (the other lanes repeat them): reduce those only
    return lanes_sum<(HP < 64 ? HP : 64)>(Call);
}
template <int HP, int R>
DEVI void write_colsums(ChainShared<HP, R>& sh, uint32_t pb, const ThreadPos& p, double part) {
    sh.psum[pb][p.rg][p.j] = part;
}
// u vector (c1 * column sums) for the thread's rows: UNI: each wave parks the vector of its row block
// in a wave-private LDS row and reads its R values back as broadcasts (~120 cycles; 2R v_readlane
// would cost ~20 cycles per value); !UNI (HP < 64): the same with the column sums themselves (lanes 0 .. HP-1 hold them all).
template <int HP, int R>
DEVI void publish_u(ChainShared<HP, R>& sh, const ThreadPos& p, double urow, double ucol) {
    using Cfg = ChainCfg<HP, R>;
    if (kExp & 4u) return;
    if constexpr (Cfg::UNI) sh.u[p.wave][p.lane] = urow;
    else { if (p.lane < (uint32_t)HP) sh.u[p.wave][p.j] = ucol; }   // (every wave its own copy: lanes 0 .. HP-1 hold every column once)
}
template <int HP, int R>
DEVI void fetch_u(const ChainShared<HP, R>& sh, const ThreadPos& p, double urow, double (&ui)[R]) {
    using Cfg = ChainCfg<HP, R>;
    if (kExp & 4u) {
#pragma unroll
        for (int k = 0; k < R; ++k) ui[k] = urow;
        return;
    }
    // no fence/wait between the write and these reads: the LDS executes one wave's instructions in
    // order, and writer and readers are the same wave (the row is wave-private / the wave is alone)
    const double* row = Cfg::UNI ? &sh.u[p.wave][p.i0 & 63u] : &sh.u[p.wave][p.i0];
#pragma unroll
    for (int k = 0; k < R; ++k) ui[k] = row[k];
}

template <int K>
DEVI double row_bcast_f64(double v) {
    // (old = the source itself: every lane has a valid source under row_newbcast, so no separate `old` register is set up)
    return __longlong_as_double(__builtin_amdgcn_update_dpp(__double_as_longlong(v), __double_as_longlong(v), 0x150 + K, 0xF, 0xF, true));
}
// R = 32 rows per wave (HP = 128): the u_i come out of two DPP-row registers — lane (l & 15) of urep[s] holds c1 * C of row
// i0 + 16 s + (l & 15) — by one v_mov_b64_dpp row_newbcast each (round 2 pulled them out of a lane-indexed register with
// two v_readlane_b32, ~17 cycles apiece, per state).  `k` is a constant once the caller's loop is unrolled.
template <int HP, int R>
DEVI void u_rows_setup(const ChainShared<HP, R>& sh, uint32_t pb, const ThreadPos& p, double c1, double (&urep)[2]) {
    using Cfg = ChainCfg<HP, R>;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t col = p.i0 + 16u * (uint32_t)s + (p.lane & 15u);
        double c = sh.psum[pb][0][col];
#pragma unroll
        for (int g = 1; g < Cfg::NRG; ++g) c += sh.psum[pb][g][col];
        urep[s] = c1 * c;
    }
}
DEVI double u_of_row(const double (&urep)[2], int k) {
    const double src = urep[(k >> 4) & 1];
    switch (k & 15) {
        case 0: return row_bcast_f64<0>(src);   case 1: return row_bcast_f64<1>(src);   case 2: return row_bcast_f64<2>(src);   case 3: return row_bcast_f64<3>(src);
        case 4: return row_bcast_f64<4>(src);   case 5: return row_bcast_f64<5>(src);   case 6: return row_bcast_f64<6>(src);   case 7: return row_bcast_f64<7>(src);
        case 8: return row_bcast_f64<8>(src);   case 9: return row_bcast_f64<9>(src);   case 10: return row_bcast_f64<10>(src); case 11: return row_bcast_f64<11>(src);
        case 12: return row_bcast_f64<12>(src); case 13: return row_bcast_f64<13>(src); case 14: return row_bcast_f64<14>(src); default: return row_bcast_f64<15>(src);
    }
}

// posterior partials of column c: acc[a] = sum over my rows with local allele a of pr = P' * beta'
// (P' = forward column BEFORE its emission multiply: the emission of a bin is applied once, to the
// finished bin, by k_bins — DESIGN.md §5)
template <int HP, int R>
DEVI void posterior(ChainShared<HP, R>& sh, gdouble* part_out, uint32_t part_slots, const RecInfo& ri, uint32_t c, const ThreadPos& p,
                    const double (&prod)[R]) {
    using Cfg = ChainCfg<HP, R>;
    if (kExp & 8u) return;
    const uint32_t nl = ri.nl;
    double acc[PG_AMAX + 1];
#pragma unroll
    for (int a = 0; a <= PG_AMAX; ++a) acc[a] = 0.0;
    // Accumulate with 0/1 multipliers instead of predicated adds: acc[a] = fma(pr, [a_i == a], acc[a])
    // rounds exactly like acc[a] += pr, needs no select chain through the accumulators, and for the
    // wave-uniform row alleles (UNI) the multiplier is a scalar operand: one VALU op per (row, allele).
    if (ri.fast) {
        // (re-assert wave-uniformity: the value may have travelled through loop-carried copies)
        const uint32_t rbits = Cfg::UNI ? (uint32_t)__builtin_amdgcn_readfirstlane(ri.fe.rowbits) : ri.fe.rowbits;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double pr = prod[k];
            const bool bit = (rbits >> k) & 1u;
            acc[1] = fma(pr, bit ? 1.0 : 0.0, acc[1]);
            acc[0] = fma(pr, bit ? 0.0 : 1.0, acc[0]);
        }
        if constexpr (Cfg::NW == 1) {
            if (part_slots == 0u) {
                // DevContig::cls4 (single-wave configurations, every object biallelic, no phantom paths): the wave IS the
                // column, so the four (row allele, column allele) class sums are formed here — four 64-lane sums — and
                // leave as 32 bytes, part[4 c + 2 (row allele) + (column allele)] (what k_sweep_lean2 writes; k_bins_lean2
                // turns them into bins, one thread per column), instead of 1 KB of per-thread partials for k_bins to reduce
                const bool aj1 = ri.em.ajr == 1u;
                const double s00 = wave_sum(aj1 ? 0.0 : acc[0]), s01 = wave_sum(aj1 ? acc[0] : 0.0);
                const double s10 = wave_sum(aj1 ? 0.0 : acc[1]), s11 = wave_sum(aj1 ? acc[1] : 0.0);
                if (p.tid == 0) {
                    gdouble2* o = (gdouble2*)part_out + (size_t)c * 2u;
                    o[0] = v2f64{s00, s01};
                    o[1] = v2f64{s10, s11};
                }
                return;
            }
        }
    } else {
        const unsigned char* al = sh.rec[c & 7u] + PG_REC_ALLELES;
        uint32_t aw[Cfg::UNI ? R / 4 : 1];
        if constexpr (Cfg::UNI) {
#pragma unroll
            for (int q = 0; q < R / 4; ++q)
                aw[q] = __builtin_amdgcn_readfirstlane(((const uint32_t*)(al + p.i0))[q]);
        }
        // (the local alleles of a column are 0 .. nl - 1, nl the same for every lane: a column with at most two — four out of
        // five in a chain that takes this path only because it has fewer paths than lanes — adds for two alleles, not five;
        // the others' sums stay exactly 0 either way)
        auto rows = [&](auto na_c) __attribute__((always_inline)) {
            constexpr int NA = decltype(na_c)::value;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                uint32_t ai;
                if constexpr (Cfg::UNI) ai = (aw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
                else ai = al[p.i0 + k];
                const double pr = prod[k];
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = fma(pr, ai == (uint32_t)a ? 1.0 : 0.0, acc[a]);
                if constexpr (R > 16) { if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
            }
        };
        if (nl <= 2u) rows(std::integral_constant<int, 2>{});
        else rows(std::integral_constant<int, PG_AMAX>{});
    }
    // Partials go out as 16-byte stores of allele PAIRS (slot pair q = alleles 2q, 2q+1): a
    // wave-wide 16-byte store costs the texture-address unit a third of two 8-byte ones, and
    // phase-2 compute waves have no loads in flight (partner columns arrive through the LDS ring),
    // so these stores never make them wait.  Layout: part[((c * part_slots/2 + q) * pT + pe) * 2 + (a & 1)], entry pe = tid of
    // pT = T entries — at HP = 32: lane j of the wave's lower half, wave * 32 + j of pT = T / 2, after the two halves were added.
    if constexpr (HP == 32) {
#pragma unroll
        for (int q = 0; q < (PG_AMAX + 1) / 2; ++q)
            if ((uint32_t)(2 * q) < nl) { acc[2 * q] = fold32(acc[2 * q]); acc[2 * q + 1] = fold32(acc[2 * q + 1]); }
    }
    if (p.pe == PG_NO_ENTRY) return;
    gdouble2* dst = (gdouble2*)part_out + (size_t)c * (part_slots >> 1) * p.pT + p.pe;
#pragma unroll
    for (int q = 0; q < (PG_AMAX + 1) / 2; ++q)
        if ((uint32_t)(2 * q) < nl) dst[(size_t)q * p.pT] = v2f64{acc[2 * q], acc[2 * q + 1]};
}

// ------------------------------------------------------------------------------------------
//  forward half-chain   (reference src/hmm.cpp:76-90, 175-273)
// ------------------------------------------------------------------------------------------
// PHASE 1: columns [0, mid), stored.  PHASE 2 (fused): columns [mid, C), posterior partials inline.
// PHASE 3 (chunked): columns [mid + chunk*K, +K), stored into the chunk scratch (posteriors by k_post).
template <int HP, int R, int PHASE, bool TST = false>
DEVI void forward_body(const DevContig& dc, ChainShared<HP, R>& sh, uint32_t C, unsigned char* ring, uint32_t chunk) {
    using Cfg = ChainCfg<HP, R, sweep_has_loader<HP, PHASE>()>;
    constexpr bool RING = Cfg::LOADER && PHASE == 2;  // partner columns via the LDS ring
    constexpr bool STORE = PHASE != 2;
    const uint32_t mid = C / 2;
    const uint32_t K = dc.chunk_cols;
    uint32_t lo = PHASE == 1 ? 0u : mid, hi = PHASE == 1 ? mid : C;
    if constexpr (PHASE == 3) {
        const unsigned long long l = (unsigned long long)mid + (unsigned long long)chunk * K;
        if (l >= C) return;
        lo = (uint32_t)l;
        hi = C - lo > K ? lo + K : C;
    }
    if (lo >= hi) return;
    const uint32_t first = lo == 0 ? 1u : lo;  // first column produced by a recursion step
    ThreadPos p;
    p.tid = threadIdx.x; p.lane = p.tid & 63u;
    p.wave = __builtin_amdgcn_readfirstlane(p.tid >> 6);
    gcu64* colrec = (gcu64*)dc.colrec;
    gdouble* part_out = (gdouble*)dc.part;
    const uint32_t part_slots = dc.cls4 ? 0u : dc.part_slots;   // (0: the four class sums of a column instead of per-thread partials, see posterior)
    const bool tri = PHASE == 2 && HP == 64 && __builtin_amdgcn_readfirstlane((int)dc.tri) != 0;  // stored columns are upper triangles
    // ... and phase 1 of such a chain that is not a lean chain (64 paths with multiallelic objects, round 6) STORES them: see tri_store_setup
    constexpr bool tri_st = TST;   // (a compile-time variant: as a run-time branch inside the unrolled steps it cost phase 1 ninety spilled registers)
    static_assert(!TST || (PHASE == 1 && HP == 64), "triangle stores: phase 1 at 64 paths");

    auto rec_load = [&](uint32_t c) -> unsigned long long {
        if (p.lane < (uint32_t)Cfg::WORDS && c < C) return colrec[(size_t)c * Cfg::WORDS + p.lane];
        return 0ull;
    };
    auto rec_stage = [&](uint32_t c, unsigned long long w) {
        if (p.lane < (uint32_t)Cfg::WORDS) ((unsigned long long*)sh.rec[c & 7u])[p.lane] = w;
    };
    if (Cfg::LOADER && p.wave >= (uint32_t)Cfg::NW) {
        // ------------------------------- loader waves --------------------------------
        // They only ISSUE LDS-DMA (HBM -> LDS, no VGPRs) and count completions.  Loader 0 moves the
        // column records (record t+6 at iteration t; a record is first read one step before its
        // column) and, in phase 2, its share of the partner column t+3; loader 1 the other share.
        // Column c is read during step c; its ring slot c&3 held column c-4.  Transfers complete
        // in order, so "at most KEEP = k * (transfers per iteration) outstanding" means that
        // everything issued k iterations ago has landed: k = 2 for the column stream (column t+1
        // and record t+4 are in LDS before B_t publishes them), k = 4 for the record-only stream
        // of phase 1.  Every transfer thus has >= 2 full steps (> the ~1.1 us DMA latency) to land.
        // In the tail (iterations that issue less) the loaders drain instead of counting.
        const uint32_t lw = p.wave - (uint32_t)Cfg::NW;
        LAS unsigned char* lring = (LAS unsigned char*)ring;
        LAS unsigned char* lrec = (LAS unsigned char*)&sh.rec[0][0];
        const gdouble* cols = (const gdouble*)dc.fwd;
        constexpr int QL = RING ? (HP * HP * 8) / 1024 / (Cfg::NLOAD > 0 ? Cfg::NLOAD : 1) : 0;  // column transfers per loader per iteration
        constexpr int KSL = kRingDist - 1;  // iterations of slack a column transfer gets
        constexpr int KEEP0 = RING ? KSL * (1 + QL) : 4, KEEP1 = KSL * QL;
        static_assert(KEEP0 < 64 && KEEP1 < 64, "vmcnt is 6 bits");
        if constexpr (RING && HP == 64) {
            if (tri) {
                // triangle ring: 9 full transfers per loader and column, 7 columns ahead.  Loader 0 also moves the
                // records, whose 8-slot ring fixes its slack at 2 iterations (record t+4 landed before B_t, as in the
                // full-column schedule); loader 1 keeps 6 iterations of its half in flight (column t+1 landed before
                // B_t).  A slot is rewritten by the issue of iteration c+1, after B_c closed the step that read it.
                if (lw == 0)
                    for (int q = -1; q < 6; ++q) dma_record<Cfg::RB>(colrec, (int64_t)first + q, C, lrec, p.lane);
                for (int q = 0; q < kTriDist; ++q) dma_column_tri(cols, dc.col_stride, (int64_t)lo + q, C, lring, lw, p.lane);
                wait_vmem_all();
                lds_barrier();  // P0
                lds_barrier();  // Bx
                for (uint32_t t = first; t < hi; ++t) {
                    if (lw == 0) dma_record<Cfg::RB>(colrec, (int64_t)t + 6, C, lrec, p.lane);
                    dma_column_tri(cols, dc.col_stride, (int64_t)t + kTriDist, C, lring, lw, p.lane);
                    if ((int64_t)t + kTriDist >= (int64_t)C) wait_vmem_all();  // tail
                    else if (lw == 0) wait_vmem_keep<kTriKeep0>();
                    else wait_vmem_keep<kTriKeep1>();
                    lds_barrier();  // B_t
                }
                lds_barrier();  // F
                return;
            }
        }
        if constexpr (RING && Cfg::SHORT) {
            // short columns (DevContig::live < HP): live / 4 transfers per column instead of HP / 4, the others' lanes never fetched
            const uint32_t live = (uint32_t)__builtin_amdgcn_readfirstlane((int)dc.live);
            if (live < (uint32_t)HP) {
                auto run = [&](auto nt_c) {
                    constexpr int NT = decltype(nt_c)::value;
                    constexpr int KP0 = KSL * (1 + NT);
                    static_assert(KP0 < 64 && Cfg::NLOAD == 1, "vmcnt is 6 bits");
                    for (int q = -1; q < 6; ++q) dma_record<Cfg::RB>(colrec, (int64_t)first + q, C, lrec, p.lane);
                    for (int q = 0; q < kRingDist; ++q) dma_column_live<HP, 1, NT>(cols, (int64_t)lo + q, C, lring, p.lane, 0, live);
                    wait_vmem_all();
                    lds_barrier();  // P0
                    lds_barrier();  // Bx
                    for (uint32_t t = first; t < hi; ++t) {
                        dma_record<Cfg::RB>(colrec, (int64_t)t + 6, C, lrec, p.lane);
                        dma_column_live<HP, 1, NT>(cols, (int64_t)t + kRingDist, C, lring, p.lane, 0, live);
                        if ((int64_t)t + 6 >= (int64_t)C) wait_vmem_all();  // tail
                        else wait_vmem_keep<KP0>();
                        lds_barrier();  // B_t
                    }
                    lds_barrier();  // F
                };
                constexpr int FULL = HP / 4;   // = HP * HP * 8 / 1024 at HP = 32
                static_assert(HP * HP * 8 / 1024 == FULL, "short columns: two row pairs per transfer");
                switch (live / 4u) {
                    case FULL - 3: run(std::integral_constant<int, FULL - 3>{}); break;
                    case FULL - 2: run(std::integral_constant<int, FULL - 2>{}); break;
                    default: run(std::integral_constant<int, FULL - 1>{}); break;   // (live / 4 == FULL - 1; fewer paths run at HP = 16)
                }
                return;
            }
        }
        if (lw == 0)  // records first-1 (column 0, or the column resumed from) .. first+5
            for (int q = -1; q < 6; ++q) dma_record<Cfg::RB>(colrec, (int64_t)first + q, C, lrec, p.lane);
        if (RING)
            for (int q = 0; q < kRingDist; ++q) dma_column<HP, Cfg::NLOAD>(cols, (int64_t)lo + q, C, lring, p.lane, lw);
        wait_vmem_all();
        lds_barrier();  // P0: first records staged
        lds_barrier();  // Bx: column lo initialised / resumed
        const bool nodma = (kExp & 128u) != 0;
        for (uint32_t t = first; t < hi; ++t) {
            if (lw == 0) dma_record<Cfg::RB>(colrec, (int64_t)t + 6, C, lrec, p.lane);
            if (RING && !nodma) dma_column<HP, Cfg::NLOAD>(cols, (int64_t)t + kRingDist, C, lring, p.lane, lw);
            if ((int64_t)t + 6 >= (int64_t)C) wait_vmem_all();  // tail
            else if (lw == 0) wait_vmem_keep<KEEP0>();
            else wait_vmem_keep<KEEP1>();
            lds_barrier();  // B_t
        }
        if (PHASE == 2) lds_barrier();  // F
        return;
    }

    // --------------------------------- compute waves -------------------------------------
    p.j = p.tid % HP; p.rg = p.tid / HP; p.i0 = p.rg * R; p.rb = (p.i0 / 64u) * 64u;
    const uint32_t H = dc.H;
    const bool full = H == (uint32_t)HP;
    const double unif = 1.0 / ((double)H * (double)H);
    gdouble* fwd = (gdouble*)dc.fwd;
    gdouble* fscale = (gdouble*)dc.fscale;
    gu8* fallback = (gu8*)dc.fwd_fallback;
    const size_t colsz = (tri || tri_st) ? (size_t)dc.col_stride : (size_t)HP * HP;
    const uint32_t dbg = dc.debug;
    // short columns (ChainCfg::SHORT): this thread stores / resumes from its rows k < kl only
    const uint32_t live = Cfg::SHORT ? (uint32_t)__builtin_amdgcn_readfirstlane((int)dc.live) : (uint32_t)HP;
    const int kl = Cfg::SHORT ? (p.j < live ? (int)live - (int)p.i0 : 0) : R;
    p.pe = p.tid; p.pT = (uint32_t)Cfg::T;
    if constexpr (HP == 32) { p.pT = (uint32_t)Cfg::T / 2u; p.pe = (p.lane < 32u && p.j < live) ? p.wave * 32u + p.j : PG_NO_ENTRY; }
    if constexpr (RING && Cfg::SHORT) {
        if (live < (uint32_t)HP) {   // the part of the ring that no transfer writes: zero, once (read as beta' = 0 by every step)
            for (uint32_t u = p.tid; u < (uint32_t)kRingSlots * HP * HP / 2u; u += (uint32_t)Cfg::T) {
                const uint32_t us = u % (uint32_t)(HP * HP / 2), q = us / (uint32_t)HP, j = us % (uint32_t)HP;
                if (j >= live || 2u * q >= live) ((v2f64*)ring)[u] = v2f64{0.0, 0.0};
            }
        }
    }

    // where this phase stores column c (c in [lo,hi)) and where the column to resume from lives
    gdouble* wr = fwd;
    gcdouble* resume = (gcdouble*)(fwd + (size_t)(lo > 0 ? lo - 1 : 0) * colsz);
    if constexpr (PHASE == 3) {
        gdouble* scr = (gdouble*)dc.scratch;
        wr = scr + (size_t)(PG_SCR_BUF(chunk) * 2u) * K * colsz - (size_t)lo * colsz;
        if (chunk > 0) resume = (gcdouble*)(scr + ((size_t)(PG_SCR_BUF(chunk - 1u) * 2u) * K + (K - 1u)) * colsz);
    }
    // Triangle stores (DevContig::tri, phase 1 at HP = 64): only the upper triangle is stored, the diagonal halved, the element
    // below the diagonal inside a straddling 16-byte unit as 0 — the layout k_sweep_lean_tri writes and the triangle ring /
    // load_col_tri of phase 2 read (a column is symmetric whatever the allele count of its object).  Fixed per thread.
    TriStore tst{};
    if constexpr (TST) tst = tri_store_setup(p.i0, p.j);
    auto store_col = [&](uint32_t c, const double (&x)[R]) {
        if (kExp & 1u) return;
        if constexpr (TST) {
            gdouble2* col = (gdouble2*)(wr + (size_t)c * colsz);
            static_for<0, R / 2>([&](auto qc) __attribute__((always_inline)) { constexpr int q = decltype(qc)::value; tri_store_pair(tst, col, q, x[2 * q], x[2 * q + 1]); });
            return;
        }
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + (size_t)(p.i0 >> 1) * HP + p.j;
#pragma unroll
        for (int k = 0; k < R; k += 2)
            if (!Cfg::SHORT || k < kl) dst[(size_t)(k >> 1) * HP] = v2f64{x[k], x[k + 1]};
    };
    // store of one row pair, issued from inside the recursion loop: eight back-to-back 16-byte
    // stores per wave queue behind each other in the texture-address unit (store-issue bound);
    // spread between the arithmetic of the following rows they cost their issue slots only
    auto store_pair = [&](uint32_t c, int k, double a, double b) __attribute__((always_inline)) {
        if (kExp & 1u) return;
        if (Cfg::SHORT && !(k < kl)) return;
        if constexpr (TST) { tri_store_pair(tst, (gdouble2*)(wr + (size_t)c * colsz), k >> 1, a, b); return; }
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + (size_t)(p.i0 >> 1) * HP + p.j;
        dst[(size_t)(k >> 1) * HP] = v2f64{a, b};
    };
    // DevContig::widef (phase 2 of 64-path chains of fused jobs): is the column of record `rec` a wide one with an aux slot?
    const bool widef = PHASE == 2 && HP == 64 && __builtin_amdgcn_readfirstlane((int)dc.widef) != 0;
    auto widef_col = [&](const unsigned char* rec, const RecInfo& ri) -> bool {
        if constexpr (PHASE == 2 && HP == 64)
            return __builtin_amdgcn_readfirstlane((int)(widef && ri.em.wide != nullptr && *(const uint32_t*)(rec + PG_REC_AUX) != PG_WIDE_NONE)) != 0;
        else return false;
    };
    auto aux_of = [&](const unsigned char* rec) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t*)(rec + PG_REC_AUX)); };
    uint32_t prev_ax = PG_WIDE_NONE;   // aux slot of the column the last step formed, if that was a wide one (flag_uniform)
    auto store_aux = [&](uint32_t ax, const double (&v)[R]) {   // the column in the stored columns' row-pair layout
        gdouble2* dst = (gdouble2*)(GAS unsigned char*)(dc.aux + (size_t)ax * 16u) + (size_t)(p.i0 >> 1) * HP + p.j;
#pragma unroll
        for (int k = 0; k < R; k += 2) dst[(size_t)(k >> 1) * HP] = v2f64{v[k], v[k + 1]};
    };
    auto load_col = [&](uint32_t c, double (&v)[R]) {
        if (c >= C) return;
        gcdouble2* src = (gcdouble2*)(c + 1 == lo ? resume : (gcdouble*)(fwd + (size_t)c * colsz)) + (size_t)(p.i0 >> 1) * HP + p.j;
#pragma unroll
        for (int k = 0; k < R; k += 2) {
            v2f64 t = v2f64{0.0, 0.0};
            if (!Cfg::SHORT || k < kl) t = src[(size_t)(k >> 1) * HP];
            v[k] = t.x; v[k + 1] = t.y;
        }
    };
    // a column stored as its upper triangle (DevContig::tri: diagonal halved, nothing below it): element (i, j)
    // of the full column = the stored (min, max), the diagonal doubled.  Once per launch (the resume column).
    auto load_col_tri = [&](gcdouble* col, double (&v)[R]) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint32_t i = p.i0 + (uint32_t)k, a = i <= p.j ? i : p.j, b = i <= p.j ? p.j : i;
            const double val = col[(size_t)tri_unit_of(a >> 1, b) * 2 + (a & 1u)];
            v[k] = a == b ? 2.0 * val : val;
        }
    };

    uint32_t tloff[R / 2], tvalid = 0;  // triangle ring: this thread's units inside a slot
    if constexpr (RING && HP == 64) { if (tri) tri_read_setup<R>(p.i0, p.j, tloff, tvalid); }
    auto ring_get = [&](int64_t c, double (&v)[R]) __attribute__((always_inline)) {
        if constexpr (HP == 64) { if (tri) { ring_read_tri<R>(ring, c, tloff, tvalid, v); return; } }
        ring_read<HP, R>(ring, c, p.i0, p.j, v);
    };
    double x[R], ui[R > 16 ? 1 : R];
    double vA[(PHASE == 2 && !RING) ? R : 1];  // register-prefetched beta' column (phase 2 without the LDS ring)
    // rows KP .. R-1 of the thread's own column live in LDS between steps (ChainCfg::PARK): slot q = rows KP + 2q, + 1
    constexpr int KP = (PHASE == 2 && !RING) ? R - Cfg::PARK : R;
    v2f64* const park = (v2f64*)ring + p.tid;
    unsigned long long tq = 0;  // inline loader (no loader wave): next record in flight
    if (!Cfg::LOADER && p.wave == 0) {
        rec_stage(first - 1, rec_load(first - 1));
        rec_stage(first, rec_load(first));
        rec_stage(first + 1, rec_load(first + 1));
        tq = rec_load(first + 2);
    }
    if constexpr (PHASE == 2 && !RING) load_col(mid, vA);
    lds_barrier();  // P0
    // cur = record of the column the next step produces, prev = record of the column before it
    RecInfo cur = decode_record<Cfg::UNI>(sh.rec[first & 7u], p.j, p.i0, full, dc.wide);
    RecInfo prev = decode_record<Cfg::UNI>(sh.rec[(first - 1) & 7u], p.j, p.i0, full, dc.wide);
    const bool prof = kChainProf && (dbg & 8u) != 0;

    if (lo == 0) {
        // column 0: v_0 = e_0 (reference src/hmm.cpp:236-238, previous_cell = 1).  Stored: P'_0 = 2^BIAS_F
        // (the column before its emission multiply, at the forward bias), fscale = 1.
        const double P0 = ldexp(1.0, PG_BIAS_F);
        double part = 0.0, pz[R];
#pragma unroll
        for (int k = 0; k < R; ++k) { pz[k] = P0; x[k] = emission_at(sh.rec[0], p.i0 + k, prev.em) * P0; part += x[k]; }
        if (STORE) store_col(0, pz);
        if (p.tid == 0) fscale[0] = 1.0;
        write_colsums<HP, R>(sh, 0, p, part);
        if constexpr (PHASE == 2) {  // lo == 0 in phase 2 <=> mid == 0 <=> C == 1
            if constexpr (RING) {
                double bt[R];
                ring_get(0, bt);
                if (widef_col(sh.rec[0], prev)) { prev_ax = aux_of(sh.rec[0]); store_aux(prev_ax, pz); }   // (DevContig::widef: see the step)
                else {
#pragma unroll
                    for (int k = 0; k < R; ++k) bt[k] *= P0;
                    posterior<HP, R>(sh, part_out, part_slots, prev, 0, p, bt);
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) vA[k] *= P0;
                posterior<HP, R>(sh, part_out, part_slots, prev, 0, p, vA);
                load_col(1, vA);
            }
        }
    } else {
        // resume behind the column the other phase (or the previous chunk) stored last: P'_{lo-1}, the
        // column before its emission multiply — or the uniform column itself if it was flagged
        if (tri) load_col_tri(resume, x);
        else load_col(lo - 1, x);
        const bool was_uniform = fallback[lo - 1] != 0;
        double part = 0.0;
        if (!was_uniform) {
            const unsigned char* recp = sh.rec[(lo - 1) & 7u];
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] *= emission_at(recp, p.i0 + k, prev.em);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) part += x[k];
        write_colsums<HP, R>(sh, (lo - 1) & 1u, p, part);
    }
    if constexpr (KP < R) {
#pragma unroll
        for (int k = KP; k < R; k += 2) park[(size_t)((k - KP) >> 1) * Cfg::T] = v2f64{x[k], x[k + 1]};
    }
    lds_barrier();  // Bx

    // Column cprev summed to zero: the reference replaces it by the uniform column (hmm.cpp:253-267).
    // alpha_hat*fsum = 1/H^2 is then an absolute value: it carries neither the emission exponent
    // X_c nor the column scale; k_bins treats flagged columns accordingly.  The register copy x of
    // the column (all zeros) is left alone — the caller folds the uniform column into u_j — so the
    // rare path changes no register array (no copies on the hot path).
    auto flag_uniform = [&](uint32_t cprev) {
        if (STORE && cprev >= lo) {  // (cprev = lo-1 was flagged and stored by the launch that produced it)
            double xu[R];
#pragma unroll
            for (int k = 0; k < R; ++k) xu[k] = (p.j < H && p.i0 + k < H) ? unif : 0.0;
            store_col(cprev, xu);
        }
        if constexpr (PHASE == 2 && HP == 64) {
            // (DevContig::widef: a wide column's slot holds the column itself — the uniform column then, as a chunk's scratch would)
            if (prev_ax != PG_WIDE_NONE && cprev >= lo) {
                double xu[R];
#pragma unroll
                for (int k = 0; k < R; ++k) xu[k] = (p.j < H && p.i0 + k < H) ? unif : 0.0;
                store_aux(prev_ax, xu);
            }
        }
        if (p.tid == 0) fallback[cprev] = 1;
    };

    // One recursion step: column t from column t-1 (reference src/hmm.cpp:175-273).
    //   P_t(i,j) = c0 x(i,j) + c1 (C_i + C_j) + c2 S,   v_t = e_t . P_t,   x = v_{t-1} (registers), C/S its sums.
    // alpha_hat_{t-1} = x / S.  Instead of dividing, P_t is scaled by 2^-es, es = exponent(S) - BIAS_F:
    // P'_t = (true P_t) * m * 2^BIAS_F with m = S * 2^-exponent(S) in [0.5,1); m goes to the side array.
    // P'_t — NOT v_t — is what gets stored: its entries are bounded below by q^2 times its sum, so the
    // stored column has a bounded dynamic range whatever the emissions are, and the emission of a genotype
    // bin is applied once, to the finished bin, as (mantissa, exponent) (k_bins / k_post).
    // The power of two rides on c0, u_j and the fma that adds u_i, so nothing on the path from the barrier
    // to the first multiply-add waits for the total S:
    //   barrier -> column sums (4 LDS reads) -> c1*C parked in LDS (round trip, gives the u_i)
    //           || S = DPP reduction of the column sums -> u_j = c1 C_j + c2 S, es, scaled constants
    //           -> R x { fma, fma, select, mul, add }  -> partial column sums to LDS -> barrier
    auto step = [&](uint32_t t, double (&vb)[(PHASE == 2 && !RING) ? R : 1]) {
        const unsigned char* rec = sh.rec[t & 7u];
        double Cj, Crow, Call;
        read_colsums<HP, R>(sh, (t - 1) & 1u, p, Cj, Crow, Call);
        double urow = cur.c1 * Crow, ucol = cur.c1 * Cj;
        // (R = 32, HP = 128: the store-only phases take the u_i from DPP-row registers; phase 2 — instruction-bound with its
        // posterior, and measured 5 % slower with them on 128 resident chains — keeps the lane-indexed register)
        constexpr bool UDPP = R > 16 && PHASE != 2;
        if constexpr (R <= 16) { publish_u<HP, R>(sh, p, urow, ucol); fetch_u<HP, R>(sh, p, urow, ui); }
        double urep[2] = {0.0, 0.0};
        if constexpr (UDPP) u_rows_setup<HP, R>(sh, (t - 1) & 1u, p, cur.c1, urep);
        __builtin_amdgcn_sched_barrier(0);  // the LDS round trip must be in flight BEFORE the reduction starts
        double S = total_sum<HP>(Call);
        double uj = fma(cur.c2, S, ucol);
        double c0 = cur.c0;
        if (__builtin_expect(!(S > 0.0), 0)) {
            // uniform column: x = 1/H^2, C = 1/H, S = 1.  The old column and its sums are all zero
            // (so are the u_i already fetched): everything goes into u_j, c0 * x drops out.
            flag_uniform(t - 1);
            const double Cu = (double)H * unif;
            S = 1.0;
            uj = fma(cur.c0, unif, fma(cur.c2, 1.0, 2.0 * cur.c1 * Cu));
            urow = 0.0;  // (the lane-indexed form of the u_i: zero like the fetched / DPP-row ones)
            c0 = 0.0;
        }
        // P'_t = (c0 x + c1 (C_i + C_j) + c2 S) * 2^-es with es = exponent(S) - BIAS_F: the stored column, BEFORE
        // its emission multiply, sums to (about) m * 2^BIAS_F; the power of two is folded into c0, u_j and
        // (as an fma multiplier) the u_i, so it costs no instruction per state.  (es is bounded below so
        // that 2^-es stays finite when S is far down the range — chains without mixing)
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        const double m = ldexp(S, -es - PG_BIAS_F);
        const bool fast = cur.fast;
        FastE fe = cur.fe;
        if (Cfg::UNI) fe.rowbits = __builtin_amdgcn_readfirstlane(fe.rowbits);
        const double sc = ldexp(1.0, -es), c0s = ldexp(c0, -es), ujs = ldexp(uj, -es);
        // LDS reads that nothing on the chain waits for are issued HERE, behind the sum exchange and
        // the u round trip (LDS returns in order: issued earlier they would delay both) and ahead
        // of the arithmetic that hides them: the partner column beta'_t out of the ring (32 KB per
        // workgroup) and the NEXT column's record, which is consumed one step later.
        __builtin_amdgcn_sched_barrier(0);
        double bt[RING ? R : 1];
        if constexpr (RING) ring_get(t, bt);
        // DevContig::widef, a WIDE column (uniform): no partials — the column P'_t itself goes to the variant's aux slot, k_bins_wide
        // forms the bins from it and the stored partner.  The states below multiply bt by P': with bt = 1 they leave P' there, exactly.
        const bool auxcol = RING && widef_col(rec, cur);
        const uint32_t ax_t = auxcol ? aux_of(rec) : PG_WIDE_NONE;
        if constexpr (RING) {
            if (auxcol) {
#pragma unroll
                for (int k = 0; k < R; ++k) bt[k] = 1.0;
            }
        }
        const RecInfo nxt = decode_record<Cfg::UNI>(sh.rec[(t + 1) & 7u], p.j, p.i0, full, dc.wide);
        __builtin_amdgcn_sched_barrier(0);
        double part = 0.0;
        // one state: P' = fma(c0s, x, fma(u_i, sc, ujs)); the recursion continues with x = P' * e; the
        // posterior (phase 2) takes P' * beta'
        v2f64 xq = v2f64{0.0, 0.0};  // the parked row pair in hand
        auto state = [&](int k, double uik, double e, double& pprev) __attribute__((always_inline)) {
            double xk;
            if (k < KP) xk = x[k < KP ? k : 0];
            else {
                if (!(k & 1)) xq = park[(size_t)((k - KP) >> 1) * Cfg::T];
                xk = (k & 1) ? xq.y : xq.x;
            }
            const double pk = fma(c0s, xk, fma(uik, sc, ujs));
            const double xn = pk * e;
            part += xn;
            if (k < KP) x[k < KP ? k : 0] = xn;
            else if (k & 1) { xq.y = xn; park[(size_t)((k - KP) >> 1) * Cfg::T] = xq; }
            else xq.x = xn;
            if constexpr (PHASE == 2) {
                if constexpr (RING) bt[k] *= pk;
                else vb[k] *= pk;
            }
            if constexpr (STORE) {
                if (k & 1) { store_pair(t, k, pprev, pk); __builtin_amdgcn_sched_barrier(0); }
                else pprev = pk;
            } else if constexpr (R > 16) { if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
        };
        // two straight-line loops (a per-state branch on `fast` would split the unrolled body into
        // tiny basic blocks and serialise it)
        if (fast) {
            double pprev = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                double uik;
                if constexpr (R <= 16) uik = ui[k];
                else if constexpr (UDPP) uik = u_of_row(urep, k);
                else uik = readlane_f64(urow, __builtin_amdgcn_readfirstlane((int)((p.i0 + k) & 63u)));
                state(k, uik, ((fe.rowbits >> k) & 1u) ? fe.eB : fe.eA, pprev);
            }
        } else {
            // (R > 16: the wide lookups get their own loop — a per-state choice drags their live ranges
            // through the 32-row loop and spills; R <= 16 measured faster with the choice per state)
            auto general_loop = [&](auto kind_c) {
                constexpr int KIND = decltype(kind_c)::value;  // 0 narrow, 1 wide, 2 per state
                double pprev = 0.0;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    double uik;
                    if constexpr (R <= 16) uik = ui[k];
                    else if constexpr (UDPP) uik = u_of_row(urep, k);
                    else uik = readlane_f64(urow, __builtin_amdgcn_readfirstlane((int)((p.i0 + k) & 63u)));
                    const double e = KIND == 0 ? emission_narrow(rec, p.i0 + k, cur.em)
                                   : KIND == 1 ? emission_wide(rec, p.i0 + k, cur.em) : emission_at(rec, p.i0 + k, cur.em);
                    state(k, uik, e, pprev);
                }
            };
            if constexpr (R <= 16) general_loop(std::integral_constant<int, 2>{});
            else if (cur.em.wide) general_loop(std::integral_constant<int, 1>{});
            else general_loop(std::integral_constant<int, 0>{});
        }
        write_colsums<HP, R>(sh, t & 1u, p, part);
        if (Cfg::NW == 1 || p.tid == 0) fscale[t] = m;  // (single-wave configurations: all lanes, same address, no exec juggling)
        if constexpr (PHASE == 2) {
            // posterior of the column just formed (optimistic: if the column turns out to sum to
            // zero, the next step flags it and k_bins re-forms its bins from the uniform column)
            if constexpr (RING) {
                if (auxcol) store_aux(ax_t, bt);
                else posterior<HP, R>(sh, part_out, part_slots, cur, t, p, bt);
            } else {
                // register prefetch of the next beta' column: a whole step ahead of its use and
                // AFTER the last read of the old one, so no vmcnt wait lands inside the recursion
                posterior<HP, R>(sh, part_out, part_slots, cur, t, p, vb);
                load_col(t + 1, vb);
            }
        }
        if (!Cfg::LOADER && p.wave == 0) {
            rec_stage(t + 2, tq);  // loaded one column ago
            tq = rec_load(t + 3);
        }
        prev = cur;
        cur = nxt;
        if constexpr (PHASE == 2 && HP == 64) prev_ax = ax_t;
        lds_barrier();  // B_t
    };
    const unsigned long long t_begin = prof ? __builtin_amdgcn_s_memtime() : 0;

    for (uint32_t t = first; t < hi; ++t) step(t, vA);
    if (prof && p.tid == 0) {
        unsigned long long* o = dc.prof + (PHASE == 1 ? 0 : 8);
        o[0] = __builtin_amdgcn_s_memtime() - t_begin; o[1] = 0; o[2] = hi - first;
    }
    {   // the last column of this phase may itself have summed to zero
        double Cj, Crow, Call;
        read_colsums<HP, R>(sh, (hi - 1) & 1u, p, Cj, Crow, Call);
        if (!(total_sum<HP>(Call) > 0.0)) flag_uniform(hi - 1);
    }
    if constexpr (PHASE == 2) lds_barrier();  // F (keeps the loader's barrier count)
}

// ------------------------------------------------------------------------------------------
//  backward half-chain   (reference src/hmm.cpp:92-110, 275-405)
//  VBUF  = number of forward-column register buffers (prefetch distance in columns)
//  KEEPW = keep w = beta_hat*e in registers across the column-sum exchange (else recompute it)
// ------------------------------------------------------------------------------------------
template <int HP, int R, int VBUF, bool KEEPW, int PHASE, bool TST = false>
DEVI void backward_body(const DevContig& dc, ChainShared<HP, R>& sh, uint32_t C, unsigned char* ring, uint32_t chunk) {
    using Cfg = ChainCfg<HP, R, sweep_has_loader<HP, PHASE>()>;
    constexpr bool RING = Cfg::LOADER && PHASE == 2;  // partner columns via the LDS ring
    constexpr bool STORE = PHASE != 2;
    const int64_t mid = C / 2;
    const int64_t K = dc.chunk_cols;
    // phase 1 computes columns C-1 .. mid (stores beta'); phase 2 computes mid-1 .. 0 (posteriors);
    // phase 3 (chunked) computes mid-1-chunk*K .. K columns down and stores them into the scratch
    int64_t top = PHASE == 1 ? (int64_t)C - 1 : mid - 1;  // first column of this phase
    int64_t bot = PHASE == 1 ? mid : 0;                    // last column of this phase
    if constexpr (PHASE == 3) {
        top = mid - 1 - (int64_t)chunk * K;
        if (top < 0) return;
        bot = top - K + 1 > 0 ? top - K + 1 : 0;
    }
    if (top < bot) return;
    ThreadPos p;
    p.tid = threadIdx.x; p.lane = p.tid & 63u;
    p.wave = __builtin_amdgcn_readfirstlane(p.tid >> 6);
    gcu64* colrec = (gcu64*)dc.colrec;
    gdouble* part_out = (gdouble*)dc.part;
    const uint32_t part_slots = dc.cls4 ? 0u : dc.part_slots;   // (0: the four class sums of a column instead of per-thread partials, see posterior)
    // recursion steps run over t = t0 .. bot where t0 = top-1 in phase 1 (column top is the
    // all-ones column) and t0 = top in phase 2 (resumed behind column mid)
    const int64_t t0 = PHASE == 1 ? top - 1 : top;
    const bool tri = PHASE == 2 && HP == 64 && __builtin_amdgcn_readfirstlane((int)dc.tri) != 0;  // stored columns are upper triangles
    constexpr bool tri_st = TST;   // (see forward_body)

    auto rec_load = [&](int64_t c) -> unsigned long long {
        if (p.lane < (uint32_t)Cfg::WORDS && c >= 0 && c < (int64_t)C) return colrec[(size_t)c * Cfg::WORDS + p.lane];
        return 0ull;
    };
    auto rec_stage = [&](int64_t c, unsigned long long w) {
        if (p.lane < (uint32_t)Cfg::WORDS && c >= 0) ((unsigned long long*)sh.rec[(uint32_t)c & 7u])[p.lane] = w;
    };
    if (Cfg::LOADER && p.wave >= (uint32_t)Cfg::NW) {
        // ------------------------------- loader waves --------------------------------
        // (see forward_body)  One iteration per recursion step t, i.e. per barrier interval
        // I_t = (B_{t+1}, B_t).  The compute waves read column t at the top of step t (inside I_t)
        // and record c during I_c .. I_{c-2} (step t reads record t+1 for the emission/transition
        // of the column behind and record t for the posterior).  Iteration t issues record t-5
        // (its slot held record t+3, dead since I_{t+1}) and column t-3 (slot of column t+1, last
        // read in I_{t+1}); KEEP guarantees column t-1 and record t-1 in LDS before B_t.
        const uint32_t lw = p.wave - (uint32_t)Cfg::NW;
        LAS unsigned char* lring = (LAS unsigned char*)ring;
        LAS unsigned char* lrec = (LAS unsigned char*)&sh.rec[0][0];
        const gdouble* cols = (const gdouble*)dc.fwd;
        constexpr int QL = RING ? (HP * HP * 8) / 1024 / (Cfg::NLOAD > 0 ? Cfg::NLOAD : 1) : 0;
        constexpr int KSL = kRingDist - 1;  // iterations of slack a column transfer gets
        constexpr int KEEP0 = RING ? KSL * (1 + QL) : 4, KEEP1 = KSL * QL;
        static_assert(KEEP0 < 64 && KEEP1 < 64, "vmcnt is 6 bits");
        if constexpr (RING && HP == 64) {
            if (tri) {  // triangle ring: see forward_body
                if (lw == 0)
                    for (int q = -1; q < 5; ++q) dma_record<Cfg::RB>(colrec, t0 - q, (int64_t)C, lrec, p.lane);  // t0+1 .. t0-4
                for (int q = 0; q < kTriDist; ++q) dma_column_tri(cols, dc.col_stride, t0 - q, (int64_t)C, lring, lw, p.lane);
                wait_vmem_all();
                lds_barrier();  // P0
                for (int64_t t = t0; t >= bot; --t) {
                    if (lw == 0) dma_record<Cfg::RB>(colrec, t - 5, (int64_t)C, lrec, p.lane);
                    dma_column_tri(cols, dc.col_stride, t - kTriDist, (int64_t)C, lring, lw, p.lane);
                    if (t - kTriDist < 0) wait_vmem_all();  // tail
                    else if (lw == 0) wait_vmem_keep<kTriKeep0>();
                    else wait_vmem_keep<kTriKeep1>();
                    lds_barrier();  // B_t
                }
                lds_barrier();  // F
                return;
            }
        }
        if constexpr (RING && Cfg::SHORT) {
            const uint32_t live = (uint32_t)__builtin_amdgcn_readfirstlane((int)dc.live);   // (see forward_body)
            if (live < (uint32_t)HP) {
                auto run = [&](auto nt_c) {
                    constexpr int NT = decltype(nt_c)::value;
                    constexpr int KP0 = KSL * (1 + NT);
                    static_assert(KP0 < 64 && Cfg::NLOAD == 1, "vmcnt is 6 bits");
                    for (int q = -1; q < 5; ++q) dma_record<Cfg::RB>(colrec, t0 - q, (int64_t)C, lrec, p.lane);
                    for (int q = 0; q < kRingDist; ++q) dma_column_live<HP, 1, NT>(cols, t0 - q, (int64_t)C, lring, p.lane, 0, live);
                    wait_vmem_all();
                    lds_barrier();  // P0
                    for (int64_t t = t0; t >= bot; --t) {
                        dma_record<Cfg::RB>(colrec, t - 5, (int64_t)C, lrec, p.lane);
                        dma_column_live<HP, 1, NT>(cols, t - kRingDist, (int64_t)C, lring, p.lane, 0, live);
                        if (t - 5 < 0) wait_vmem_all();  // tail
                        else wait_vmem_keep<KP0>();
                        lds_barrier();  // B_t
                    }
                    lds_barrier();  // F
                };
                constexpr int FULL = HP / 4;
                switch (live / 4u) {
                    case FULL - 3: run(std::integral_constant<int, FULL - 3>{}); break;
                    case FULL - 2: run(std::integral_constant<int, FULL - 2>{}); break;
                    default: run(std::integral_constant<int, FULL - 1>{}); break;
                }
                return;
            }
        }
        if (lw == 0)
            for (int q = -1; q < 5; ++q) dma_record<Cfg::RB>(colrec, t0 - q, (int64_t)C, lrec, p.lane);  // t0+1 .. t0-4
        if (RING)
            for (int q = 0; q < kRingDist; ++q) dma_column<HP, Cfg::NLOAD>(cols, t0 - q, (int64_t)C, lring, p.lane, lw);
        wait_vmem_all();
        lds_barrier();  // P0
        const bool nodma = (kExp & 128u) != 0;
        for (int64_t t = t0; t >= bot; --t) {
            if (lw == 0) dma_record<Cfg::RB>(colrec, t - 5, (int64_t)C, lrec, p.lane);
            if (RING && !nodma) dma_column<HP, Cfg::NLOAD>(cols, t - kRingDist, (int64_t)C, lring, p.lane, lw);
            if (t - 5 < 0) wait_vmem_all();  // tail
            else if (lw == 0) wait_vmem_keep<KEEP0>();
            else wait_vmem_keep<KEEP1>();
            lds_barrier();  // B_t
        }
        if (PHASE == 2) lds_barrier();  // F
        return;
    }

    // --------------------------------- compute waves -------------------------------------
    p.j = p.tid % HP; p.rg = p.tid / HP; p.i0 = p.rg * R; p.rb = (p.i0 / 64u) * 64u;
    const uint32_t H = dc.H;
    const bool full = H == (uint32_t)HP;
    const double unif = 1.0 / ((double)H * (double)H);
    gdouble* cols = (gdouble*)dc.fwd;
    gdouble* bscale = (gdouble*)dc.bscale;
    gdouble* bsum = (gdouble*)dc.bsum;
    const size_t colsz = (tri || tri_st) ? (size_t)dc.col_stride : (size_t)HP * HP;
    // short columns (ChainCfg::SHORT, see forward_body): this thread stores / resumes from its rows k < kl only
    const uint32_t live = Cfg::SHORT ? (uint32_t)__builtin_amdgcn_readfirstlane((int)dc.live) : (uint32_t)HP;
    const int kl = Cfg::SHORT ? (p.j < live ? (int)live - (int)p.i0 : 0) : R;
    p.pe = p.tid; p.pT = (uint32_t)Cfg::T;
    if constexpr (HP == 32) { p.pT = (uint32_t)Cfg::T / 2u; p.pe = (p.lane < 32u && p.j < live) ? p.wave * 32u + p.j : PG_NO_ENTRY; }
    if constexpr (RING && Cfg::SHORT) {
        if (live < (uint32_t)HP) {
            for (uint32_t u = p.tid; u < (uint32_t)kRingSlots * HP * HP / 2u; u += (uint32_t)Cfg::T) {
                const uint32_t us = u % (uint32_t)(HP * HP / 2), q = us / (uint32_t)HP, j = us % (uint32_t)HP;
                if (j >= live || 2u * q >= live) ((v2f64*)ring)[u] = v2f64{0.0, 0.0};
            }
        }
    }

    // where this phase stores column c (c in [bot,top]) and where the column to resume from lives
    gdouble* wr = cols;
    gcdouble* resume = (gcdouble*)(cols + (size_t)(top + 1 < (int64_t)C ? top + 1 : top) * colsz);
    if constexpr (PHASE == 3) {
        gdouble* scr = (gdouble*)dc.scratch;
        wr = scr + (size_t)(PG_SCR_BUF(chunk) * 2u + 1u) * (size_t)K * colsz - (size_t)bot * colsz;
        if (chunk > 0) resume = (gcdouble*)(scr + (size_t)(PG_SCR_BUF(chunk - 1u) * 2u + 1u) * (size_t)K * colsz);  // its slot 0
    }
    auto load_col = [&](int64_t c, double (&v)[R]) {
        if (c < 0) return;
        gcdouble2* src = (gcdouble2*)(c == top + 1 ? resume : (gcdouble*)(cols + (size_t)c * colsz)) + (size_t)(p.i0 >> 1) * HP + p.j;
#pragma unroll
        for (int k = 0; k < R; k += 2) {
            v2f64 t = v2f64{0.0, 0.0};
            if (!Cfg::SHORT || k < kl) t = src[(size_t)(k >> 1) * HP];
            v[k] = t.x; v[k + 1] = t.y;
        }
    };
    // DevContig::widef: see forward_body
    const bool widef = PHASE == 2 && HP == 64 && __builtin_amdgcn_readfirstlane((int)dc.widef) != 0;
    auto widef_col = [&](const unsigned char* rec, const RecInfo& ri) -> bool {
        if constexpr (PHASE == 2 && HP == 64)
            return __builtin_amdgcn_readfirstlane((int)(widef && ri.em.wide != nullptr && *(const uint32_t*)(rec + PG_REC_AUX) != PG_WIDE_NONE)) != 0;
        else return false;
    };
    auto store_aux = [&](const unsigned char* rec, const double (&v)[R]) {
        const uint32_t ax = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t*)(rec + PG_REC_AUX));
        gdouble2* dst = (gdouble2*)(GAS unsigned char*)(dc.aux + (size_t)ax * 16u) + (size_t)(p.i0 >> 1) * HP + p.j;
#pragma unroll
        for (int k = 0; k < R; k += 2) dst[(size_t)(k >> 1) * HP] = v2f64{v[k], v[k + 1]};
    };
    auto load_col_tri = [&](gcdouble* col, double (&v)[R]) {  // see forward_body
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const uint32_t i = p.i0 + (uint32_t)k, a = i <= p.j ? i : p.j, b = i <= p.j ? p.j : i;
            const double val = col[(size_t)tri_unit_of(a >> 1, b) * 2 + (a & 1u)];
            v[k] = a == b ? 2.0 * val : val;
        }
    };
    TriStore tst{};   // triangle stores: see forward_body
    if constexpr (TST) tst = tri_store_setup(p.i0, p.j);
    auto store_col = [&](int64_t c, const double (&y)[R]) {
        if constexpr (TST) {
            gdouble2* col = (gdouble2*)(wr + (size_t)c * colsz);
            static_for<0, R / 2>([&](auto qc) __attribute__((always_inline)) { constexpr int q = decltype(qc)::value; tri_store_pair(tst, col, q, y[2 * q], y[2 * q + 1]); });
            return;
        }
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + (size_t)(p.i0 >> 1) * HP + p.j;
#pragma unroll
        for (int k = 0; k < R; k += 2)
            if (!Cfg::SHORT || k < kl) dst[(size_t)(k >> 1) * HP] = v2f64{y[k], y[k + 1]};
    };
    auto store_pair = [&](int64_t c, int k, double a, double b) __attribute__((always_inline)) {  // see forward_body
        if (kExp & 1u) return;
        if (Cfg::SHORT && !(k < kl)) return;
        if constexpr (TST) { tri_store_pair(tst, (gdouble2*)(wr + (size_t)c * colsz), k >> 1, a, b); return; }
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + (size_t)(p.i0 >> 1) * HP + p.j;
        dst[(size_t)(k >> 1) * HP] = v2f64{a, b};
    };

    uint32_t tloff[R / 2], tvalid = 0;  // triangle ring: this thread's units inside a slot
    if constexpr (RING && HP == 64) { if (tri) tri_read_setup<R>(p.i0, p.j, tloff, tvalid); }
    auto ring_get = [&](int64_t c, double (&v)[R]) __attribute__((always_inline)) {
        if constexpr (HP == 64) { if (tri) { ring_read_tri<R>(ring, c, tloff, tvalid, v); return; } }
        ring_read<HP, R>(ring, c, p.i0, p.j, v);
    };
    constexpr int NV = (PHASE == 2 && !RING) ? R : 1;
    double y[R], vA[NV], vB[(PHASE == 2 && !RING && VBUF == 2) ? R : 1];
    // rows KP .. R-1 of the thread's own column live in LDS between steps (ChainCfg::PARK, see forward_body)
    constexpr int KP = (PHASE == 2 && !RING) ? R - Cfg::PARK : R;
    v2f64* const park = (v2f64*)ring + p.tid;
    auto y_get = [&](int k, v2f64& q) __attribute__((always_inline)) -> double {
        if (k < KP) return y[k < KP ? k : 0];
        if (!(k & 1)) q = park[(size_t)((k - KP) >> 1) * Cfg::T];
        return (k & 1) ? q.y : q.x;
    };
    auto y_put = [&](int k, v2f64& q, double val) __attribute__((always_inline)) {
        if (k < KP) y[k < KP ? k : 0] = val;
        else if (k & 1) { q.y = val; park[(size_t)((k - KP) >> 1) * Cfg::T] = q; }
        else q.x = val;
    };
    double Sy = 0.0;
    unsigned long long tq = 0;  // inline loader (no loader wave): next record in flight
    if (!Cfg::LOADER && p.wave == 0) {
        rec_stage(t0 + 1, rec_load(t0 + 1));
        rec_stage(t0, rec_load(t0));
        rec_stage(t0 - 1, rec_load(t0 - 1));
        tq = rec_load(t0 - 2);
    }
    if constexpr (PHASE == 1) {
        // column C-1: beta~ = 1 (reference src/hmm.cpp:356-358), stored at the backward bias: sum = H^2 2^BIAS_B
        const double B0 = ldexp(1.0, PG_BIAS_B);
#pragma unroll
        for (int k = 0; k < R; ++k) y[k] = (p.j < H && p.i0 + k < H) ? B0 : 0.0;
        Sy = (double)H * (double)H * B0;
        store_col(top, y);
        if (p.tid == 0) { bscale[top] = 1.0; bsum[top] = Sy; }
    } else {
        // resume behind the column stored last (by phase 1, or by the previous chunk)
        if (tri) load_col_tri(resume, y);
        else load_col(top + 1, y);
        Sy = bsum[top + 1];
        if constexpr (PHASE == 2 && !RING) {
            load_col(top, vA);
            if constexpr (VBUF == 2) load_col(top - 1, vB);
        }
    }
    if constexpr (KP < R) {
#pragma unroll
        for (int k = KP; k < R; k += 2) park[(size_t)((k - KP) >> 1) * Cfg::T] = v2f64{y[k], y[k + 1]};
    }
    lds_barrier();  // P0
    RecInfo cur = decode_record<Cfg::UNI>(sh.rec[(uint32_t)(t0 + 1) & 7u], p.j, p.i0, full, dc.wide);
    const bool prof = kChainProf && (dc.debug & 8u) != 0;
    const unsigned long long t_begin = prof ? __builtin_amdgcn_s_memtime() : 0;

    // One recursion step: beta'_t from beta'_{t+1} (reference src/hmm.cpp:275-405).
    //   w = beta_hat_{t+1} . e_{t+1};  beta~_t(i,j) = c0 w(i,j) + c1 (W_i + W_j) + c2 Sw,  W/Sw sums of w.
    // The scale 2^-es (es = exponent of the previous column's sum Sy, known analytically through
    // kappa) is folded into the transition constants before the barrier; behind it the path is
    //   column sums of w -> c1'W parked in LDS (u_i)  ||  Sw = DPP reduction -> u_j  ->  R x { add, fma }
    auto step = [&](int64_t t, double (&v)[NV]) {
        // beta_hat_{t+1} = y / Sy, uniform if the sum is zero (hmm.cpp:374-380)
        if (__builtin_expect(!(Sy > 0.0) || !(Sy < INFINITY), 0)) {
            v2f64 uq = v2f64{0.0, 0.0};
#pragma unroll
            for (int k = 0; k < R; ++k) y_put(k, uq, (p.j < H && p.i0 + k < H) ? unif : 0.0);
            Sy = 1.0;
        }
        // partner column v'_t out of the ring (landed before B_{t+1}), read ahead of its use; then
        // record t (posterior of this column, emission of the next step; the record of column
        // t+1 was decoded one step ago)
        double vt[RING ? R : 1];
        if constexpr (RING) ring_get(t, vt);
        const RecInfo nxt = decode_record<Cfg::UNI>(sh.rec[(uint32_t)t & 7u], p.j, p.i0, full, dc.wide);
        const unsigned char* rec1 = sh.rec[(uint32_t)(t + 1) & 7u];
        // beta~_t(true) = A (y/Sy . e) A^T; scaled by 2^-es: beta' = beta~ * m, m = Sy*2^-es
        int es = exponent_of(Sy) - PG_BIAS_B;  // stored backward columns carry 2^BIAS_B (see forward_body)
        es = es < -900 ? -900 : es;
        const double m = ldexp(Sy, -es - PG_BIAS_B);
        if (p.tid == 0) bscale[t] = m;
        const double k0 = ldexp(cur.c0, -es), k1 = ldexp(cur.c1, -es), k2 = ldexp(cur.c2, -es);
        const double kap = ldexp(cur.kappa, -es);
        const bool fast = cur.fast;
        FastE fe = cur.fe;
        if (Cfg::UNI) fe.rowbits = __builtin_amdgcn_readfirstlane(fe.rowbits);
        // (values, not fields: "bit ? fe.eB : fe.eA" is a conditional LVALUE — a select between two addresses inside the struct —
        // and where the struct then stayed in memory the 32 selects of a loop became 32 loads from SCRATCH, indexed by the bit)
        const double feA = fe.eA, feB = fe.eB;
        double w[KEEPW ? R : 1];
        double part = 0.0;
        v2f64 yq = v2f64{0.0, 0.0};  // the parked row pair in hand
        if (fast) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double yk = y_get(k, yq);
                const double wk = (kExp & 16u) ? yk : yk * (((fe.rowbits >> k) & 1u) ? +feB : +feA);
                if constexpr (KEEPW) w[k] = wk;
                part += (kExp & 16u) ? (k == 0 ? wk : 0.0) : wk;
            }
        } else if (cur.em.wide) {  // rare: branch at loop level, so the wide lookups stay out of the other loops' live ranges
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double wk = y_get(k, yq) * emission_wide(rec1, p.i0 + k, cur.em);
                if constexpr (KEEPW) w[k] = wk;
                part += wk;
                if constexpr (R > 16) { if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
            }
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double wk = y_get(k, yq) * emission_narrow(rec1, p.i0 + k, cur.em);
                if constexpr (KEEPW) w[k] = wk;
                part += wk;
                if constexpr (R > 16) { if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
            }
        }
        const uint32_t pb = (uint32_t)t & 1u;
        write_colsums<HP, R>(sh, pb, p, part);
        if (!Cfg::LOADER && p.wave == 0) {
            rec_stage(t - 2, tq);  // loaded one column ago
            tq = rec_load(t - 3);
        }
        lds_barrier();  // B_t
        double Cj, Crow, Call;
        read_colsums<HP, R>(sh, pb, p, Cj, Crow, Call);
        const double urow = k1 * Crow, ucol = k1 * Cj;
        double ui[R > 16 ? 1 : R];
        if constexpr (R <= 16) { publish_u<HP, R>(sh, p, urow, ucol); fetch_u<HP, R>(sh, p, urow, ui); }
        constexpr bool UDPP = R > 16 && PHASE != 2;   // (see forward_body)
        double urep[2] = {0.0, 0.0};
        if constexpr (UDPP) u_rows_setup<HP, R>(sh, pb, p, k1, urep);
        __builtin_amdgcn_sched_barrier(0);  // the LDS round trip must be in flight BEFORE the reduction starts
        const double Sw = total_sum<HP>(Call);
        const double uj = fma(k2, Sw, ucol);
        auto beta_loop = [&](auto kind_c) {  // 0 general narrow, 1 fast (or w kept), 2 general wide
            constexpr int KIND = decltype(kind_c)::value;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                double wk;
                if constexpr (KEEPW) wk = w[k];
                else {
                    const double yk = y_get(k, yq);
                    if constexpr (KIND == 1) wk = yk * (((fe.rowbits >> k) & 1u) ? +feB : +feA);
                    else if constexpr (KIND == 2) wk = yk * emission_wide(rec1, p.i0 + k, cur.em);
                    else wk = yk * emission_narrow(rec1, p.i0 + k, cur.em);
                }
                double uik;
                if constexpr (R <= 16) uik = ui[k];
                else if constexpr (UDPP) uik = u_of_row(urep, k);
                else uik = readlane_f64(urow, __builtin_amdgcn_readfirstlane((int)((p.i0 + k) & 63u)));
                const double yn = (kExp & 16u) ? (k == 0 ? wk + uik + uj : wk) : fma(k0, wk, uik + uj);  // beta'_t
                y_put(k, yq, yn);
                if constexpr (PHASE == 2 && !RING) v[k] *= yn;  // P'_t * beta'_t (posterior below)
                if constexpr (STORE) { if (k & 1) { store_pair(t, k, y[k - 1], y[k]); __builtin_amdgcn_sched_barrier(0); } }
                else if constexpr (R > 16) { if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0); }
            }
        };
        if (KEEPW || fast) beta_loop(std::integral_constant<int, 1>{});
        else if (cur.em.wide) beta_loop(std::integral_constant<int, 2>{});
        else beta_loop(std::integral_constant<int, 0>{});
        Sy = kap * Sw;  // = sum(beta'_t) over real states
        if constexpr (STORE) {
            if (p.tid == 0) bsum[t] = Sy;
        } else {
            if constexpr (RING) {
#pragma unroll
                for (int k = 0; k < R; ++k) vt[k] *= y[k];
                posterior<HP, R>(sh, part_out, part_slots, nxt, (uint32_t)t, p, vt);
            } else {  // (the products were formed in beta_loop)
                posterior<HP, R>(sh, part_out, part_slots, nxt, (uint32_t)t, p, v);
                load_col(t - VBUF, v);
            }
        }
        cur = nxt;
    };

    if constexpr (R <= 16) {
        // -------------------------------------------------------------------------------------
        // Fused formulation (all loader-wave configurations).  The emission multiply of the NEXT
        // step (w = beta_hat . e) is done in the same loop that forms beta'_t, so a step is
        //   barrier -> column sums of w -> u round trip || total -> R x { add, fma, select, mul, add }
        //           -> partial column sums of the new w -> barrier
        // exactly like the forward step: one dependent segment per column instead of two.
        // -------------------------------------------------------------------------------------
        static_assert(PHASE != 2 || RING, "loader configurations read partner columns from the LDS ring");
        double w[R];
        auto rowbits_of = [&](const RecInfo& ri) -> uint32_t {
            return Cfg::UNI ? (uint32_t)__builtin_amdgcn_readfirstlane(ri.fe.rowbits) : ri.fe.rowbits;
        };
        {   // prologue: w = beta_hat_{t0+1} . e_{t0+1}
            if (!(Sy > 0.0)) {  // (phase 2 resuming behind an all-zero column, hmm.cpp:374-380)
#pragma unroll
                for (int k = 0; k < R; ++k) y[k] = (p.j < H && p.i0 + k < H) ? unif : 0.0;
                Sy = 1.0;
            }
            const unsigned char* rec1 = sh.rec[(uint32_t)(t0 + 1) & 7u];
            const uint32_t rb1 = rowbits_of(cur);
            double part = 0.0;
            if (cur.fast) {
#pragma unroll
                for (int k = 0; k < R; ++k) { w[k] = y[k] * (((rb1 >> k) & 1u) ? cur.fe.eB : cur.fe.eA); part += w[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) { w[k] = y[k] * emission_at(rec1, p.i0 + k, cur.em); part += w[k]; }
            }
            write_colsums<HP, R>(sh, (uint32_t)t0 & 1u, p, part);
        }
        for (int64_t t = t0; t >= bot; --t) {
            // scale of this column: 2^-es, es = exponent of the previous column's sum (known
            // analytically through kappa, so all of this sits in front of the barrier)
            int es = exponent_of(Sy) - PG_BIAS_B;  // stored backward columns carry 2^BIAS_B
            es = es < -900 ? -900 : es;
            const double m = ldexp(Sy, -es - PG_BIAS_B);
            if (Cfg::NW == 1 || p.tid == 0) bscale[t] = m;
            const double k0 = ldexp(cur.c0, -es), k1 = ldexp(cur.c1, -es), k2 = ldexp(cur.c2, -es);
            const double kap = ldexp(cur.kappa, -es);
            double vt[RING ? R : 1];
            if constexpr (RING) ring_get(t, vt);  // landed before B_{t+1}
            const RecInfo nxt = decode_record<Cfg::UNI>(sh.rec[(uint32_t)t & 7u], p.j, p.i0, full, dc.wide);
            const unsigned char* rec0 = sh.rec[(uint32_t)t & 7u];
            const uint32_t rb0 = rowbits_of(nxt);
            if (!Cfg::LOADER && p.wave == 0) {   // (no loader wave: see the 32-row step)
                rec_stage(t - 2, tq);  // loaded one column ago
                tq = rec_load(t - 3);
            }
            lds_barrier();  // B_t
            double Cj, Crow, Call;
            read_colsums<HP, R>(sh, (uint32_t)t & 1u, p, Cj, Crow, Call);
            const double urow = k1 * Crow, ucol = k1 * Cj;
            publish_u<HP, R>(sh, p, urow, ucol);
            double ui[R];
            fetch_u<HP, R>(sh, p, urow, ui);
            __builtin_amdgcn_sched_barrier(0);  // the LDS round trip must be in flight BEFORE the reduction starts
            const double Sw = total_sum<HP>(Call);
            const double uj = fma(k2, Sw, ucol);
            const double Snew = kap * Sw;  // = sum(beta'_t) over real states
            double part = 0.0;
            if (__builtin_expect(!(Snew > 0.0), 0)) {
                // beta~_t is all zero: its own posteriors are 0, the next step starts from the
                // uniform column (hmm.cpp:374-380)
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    y[k] = 0.0;
                    const double bu = (p.j < H && p.i0 + k < H) ? unif : 0.0;
                    w[k] = bu * (nxt.fast ? (((rb0 >> k) & 1u) ? nxt.fe.eB : nxt.fe.eA) : emission_at(rec0, p.i0 + k, nxt.em));
                    part += w[k];
                }
                if constexpr (STORE) store_col(t, y);
            } else if (nxt.fast) {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    y[k] = (kExp & 16u) ? w[k] : fma(k0, w[k], ui[k] + uj);  // beta'_t
                    if constexpr (STORE) { if (k & 1) { store_pair(t, k, y[k - 1], y[k]); __builtin_amdgcn_sched_barrier(0); } }
                    w[k] = (kExp & 16u) ? y[k] : y[k] * (((rb0 >> k) & 1u) ? nxt.fe.eB : nxt.fe.eA);
                    part += (kExp & 16u) ? (k == 0 ? w[0] + ui[k] + uj : 0.0) : w[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    y[k] = fma(k0, w[k], ui[k] + uj);
                    if constexpr (STORE) { if (k & 1) { store_pair(t, k, y[k - 1], y[k]); __builtin_amdgcn_sched_barrier(0); } }
                    w[k] = y[k] * emission_at(rec0, p.i0 + k, nxt.em);
                    part += w[k];
                }
            }
            write_colsums<HP, R>(sh, (uint32_t)(t - 1) & 1u, p, part);
            if constexpr (STORE) { if (Cfg::NW == 1 || p.tid == 0) bsum[t] = Snew; }
            else if (widef_col(rec0, nxt)) store_aux(rec0, y);   // (DevContig::widef, a wide column: beta'_t to its aux slot — see forward_body)
            else {
#pragma unroll
                for (int k = 0; k < R; ++k) vt[k] *= y[k];  // P'_t * beta'_t
                posterior<HP, R>(sh, part_out, part_slots, nxt, (uint32_t)t, p, vt);
            }
            Sy = Snew > 0.0 ? Snew : 1.0;
            cur = nxt;
        }
    } else
    for (int64_t t = t0; t >= bot; t -= 2) {
        if constexpr (PHASE == 2 && VBUF == 2 && !RING) {
            step(t, vA);
            if (t - 1 >= bot) step(t - 1, vB);
        } else {
            step(t, vA);
            if (t - 1 >= bot) step(t - 1, vA);
        }
    }
    if (prof && p.tid == 0) {
        unsigned long long* o = dc.prof + (PHASE == 1 ? 16 : 24);
        o[0] = __builtin_amdgcn_s_memtime() - t_begin; o[1] = 0; o[2] = (unsigned long long)(t0 - bot + 1);
    }
    if constexpr (PHASE == 2) lds_barrier();  // F (keeps the loader's barrier count)
}

// grid = (n_contigs, 2): blockIdx.y = 0 forward half-chain, 1 backward half-chain
template <int HP, int R, int VBUF, bool KEEPW, int PHASE>
// HP = 32 (17 .. 32 paths: user-chosen panel sizes — the default of haplotype sampling, 15 + 1 = 16 paths, is k_sweep_small16[x]'s): 8 rows per lane = two compute waves (+ the loader in
// phase 2 only: sweep_has_loader) per half-chain, held to 128 registers (a few spills) so that a SIMD takes four waves.  With 16 rows per lane (one compute
// wave of 209 - 240 registers) a CU ran four half-chains at a time and two thirds of a 1024-chain cohort's time was
// waiting: 72 -> 59 ms per step on `cohort_h17` (PG_HP32_ROWS=16 PG_HP32_WAVES=1 builds the old configuration).
// Phase 2 (posterior inline) is better off with three waves per SIMD and 168 registers: 22.5 -> 20.5 ms there.
#ifndef PG_HP32_WAVES
#define PG_HP32_WAVES 4
#endif
#ifndef PG_HP32_WAVES_P2
#define PG_HP32_WAVES_P2 3
#endif
__global__ __launch_bounds__((ChainCfg<HP, R, sweep_has_loader<HP, PHASE>()>::TT), (HP == 32 ? (PHASE == 2 ? PG_HP32_WAVES_P2 : PG_HP32_WAVES) : 1)) void k_sweep(const DevContig* __restrict__ contigs, uint32_t chunk) {
    __shared__ ChainShared<HP, R> sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_ring[];  // phase 2: kRingSlots column slots
    const DevContig& dc = contigs[blockIdx.x];
    if (dc.HP != (uint32_t)HP) return;
    if (dc.split) return;   // the split path: k_sweep_small16[x] both phases; a chain left with one column needs no sweep (k_bins_s)
    if (PHASE != 2 && (dc.lean || dc.small || dc.smallx || dc.leanx)) return;  // store-only phases of all-biallelic H = 64 / H = 16 chains: k_sweep_lean / k_sweep_small16[x]
    if (PHASE == 2 && dc.leanx2) return;   // 64-path triangle chains with multiallelic objects: k_sweep_leanx2
    // (written by k_compact: a vector load as far as the compiler knows — make the trip count, and
    // with it every column index, ring slot and address derived from it, wave-uniform again)
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    if (PHASE == 2 && dc.tri == 2u && C >= 2) return;  // triangle chains: k_sweep_lean2
    if (PHASE == 2 && (dc.small == 2u || dc.smallx == 2u) && C >= 2) return;  // 16-path chains of fused jobs: k_sweep_small16<2> / k_sweep_small16x<2>
    if constexpr (PHASE == 2 && HP == 64 && ChainCfg<HP, R>::LOADER) {
        // triangle ring (DevContig::tri): the unit of zeros that stands for everything below the diagonal; first read
        // behind the P0 barrier of the bodies
        if (threadIdx.x == 0) *(v2f64*)(dyn_ring + (uint32_t)kTriSlots * kTriSlotB) = v2f64{0.0, 0.0};
    }
    if (PHASE == 1 && HP == 64 && dc.tri) return;   // phase 1 of 64-path triangle chains with multiallelic objects: k_sweep_tri1
    if (blockIdx.y == 0) forward_body<HP, R, PHASE>(dc, sh, C, dyn_ring, chunk);
    else backward_body<HP, R, VBUF, KEEPW, PHASE>(dc, sh, C, dyn_ring, chunk);
}
// phase 1 of the 64-path chains of fused jobs that store their columns as upper triangles and are not lean chains (objects with
// 3 .. PG_AMAX alleles, round 6): the general kernel with triangle stores — k_sweep_lean_tri's layout, read by phase 2 through the
// general kernel's triangle ring (DevContig::tri == 1)
__global__ __launch_bounds__((ChainCfg<64, 16, sweep_has_loader<64, 1>()>::TT), 1) void k_sweep_tri1(const DevContig* __restrict__ contigs) {
    __shared__ ChainShared<64, 16> sh;
    const DevContig& dc = contigs[blockIdx.x];
    if (dc.HP != 64u || dc.split || dc.lean || dc.leanx || !dc.tri) return;   // (leanx == 2: k_sweep_leanx_tri)
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    if (blockIdx.y == 0) forward_body<64, 16, 1, true>(dc, sh, C, nullptr, 0);
    else backward_body<64, 16, 1, true, 1, true>(dc, sh, C, nullptr, 0);
}

// ------------------------------------------------------------------------------------------
//  k_sweep_lean : the store-only half-chains (phases 1 and 3) of chains whose every column is
//  biallelic with H = HP = 64 (BASELINE.json configs[2], [3]) — the lone-chain regime, where the time of
//  a chain is (columns) x (instructions per column) x (4 cycles: one wave per SIMD issues one
//  instruction of ANY type every fourth cycle).  Same arithmetic and stored data as forward_body /
//  backward_body; what is gone is instruction count:
//    * no loader waves, no LDS record ring, no per-lane record decode: a column's constants
//      {c0, c1, c2, kappa, E'00, E'01, E'11, bits1} are a 64-byte compact record (k_records) fetched with
//      ONE scalar load a column ahead; everything wave-uniform stays in SGPRs and enters the VALU as
//      a scalar operand.  Four waves on four SIMDs: the per-column barrier costs ~10 cycles instead of ~115.
//    * the total S = sum over the 64 lanes: two fp64 MFMAs against a ones matrix (+ 3 adds) instead of
//      six DPP steps (~26 instructions).  D = A x 1 sums the four 16-lane groups; the second product
//      sums the sixteen rows (C/D layout of v_mfma_f64_16x16x4: row = (lane >> 4) + 4 reg, col = lane & 15).
//      This is a reduction, not a contraction of the model: the recursion itself stays rank-structured.
//    * emissions by row pairs: one SDWA add + one 16-byte LDS read per pair out of a table built when the block of
//      records is parked (lean_expand), fetched a step ahead; stores with immediate offsets off two per-thread bases.
// ------------------------------------------------------------------------------------------
#define PG_LEAN_BLOCK 64  // column records per LDS block (one 16-byte piece per thread: 64 x 64 B = 256 x 16 B)
template <int R>
struct LeanShared {
    static constexpr int NW = 64 / R;  // waves = row groups
    double psum[2][NW][64];
    double ptot[2][NW] __attribute__((aligned(32)));   // the waves' totals of their partial sums (kLeanPreTotal: lean_put_sums)
    double u[NW][64] __attribute__((aligned(16)));
    double rec[2][PG_LEAN_BLOCK][8] __attribute__((aligned(16)));  // two blocks of compact records
    double scal[2][64];   // per-column scalars on their way out (ColScalars)
    // emissions by ROW PAIRS (lean_expand): tab[b][r][c][a] = {e(row 2p, column allele a), e(row 2p+1, a)} for the row
    // pair's allele combination c = bit(2p) + 2 bit(2p+1); comb[b][r][w][p] = 32 c of pair p of wave w's rows
    v2f64 tab[2][PG_LEAN_BLOCK][4][2];
    unsigned char comb[2][PG_LEAN_BLOCK][NW][R / 2] __attribute__((aligned(8)));
};
struct FRec {  // compact column record (64 B)
    double c0, c1, c2, kappa, E00, E01, E11;
    unsigned long long bits1;
};
typedef double v4f64 __attribute__((ext_vector_type(4)));
// Records reach the waves in BLOCKS of 64 columns: every thread fetches one 16-byte piece of the block
// after next into a register (one global load per thread per 64 columns) and parks it in LDS a few
// columns before the block is needed; a column's record is then four broadcast LDS reads, issued one
// step ahead of use.  `rel` = distance of the record from the first one of this launch (ascending
// columns forward, descending backward).
// (Measured alternatives: scalar loads of the record — every LDS wait of the step then also waits out the
// scalar load's memory latency, lgkmcnt being shared: 2300 cycles per column; records and the u_i kept
// in registers and pulled out with v_readlane — ~17 cycles per v_readlane_b32: 2230 cycles per column.)
struct LeanRecs {
    const GAS char* base;   // frec
    int64_t origin, C;      // column of rel 0, number of columns
    int dir;                // +1 forward, -1 backward
    uint32_t tid;
    DEVI v2f64 fetch(uint32_t block) const {  // this thread's piece of block `block` (threads 0..255)
        int64_t c = origin + (int64_t)dir * ((int64_t)block * PG_LEAN_BLOCK + ((tid & 255u) >> 2));
        c = c < 0 ? 0 : (c >= C ? C - 1 : c);
        return *(const GAS v2f64*)(base + (size_t)c * 64u + (tid & 3u) * 16u);
    }
    template <class SH>
    DEVI void park(SH& sh, uint32_t block, v2f64 piece) const {
        if (tid < 256u) ((v2f64*)&sh.rec[block & 1u][0][0])[tid] = piece;
    }
};
template <class SH>
DEVI FRec read_frec(const SH& sh, uint32_t rel /*uniform*/) {
    const double* q = sh.rec[(rel / PG_LEAN_BLOCK) & 1u][rel % PG_LEAN_BLOCK];
    FRec r;
    r.c0 = q[0]; r.c1 = q[1]; r.c2 = q[2]; r.kappa = q[3]; r.E00 = q[4]; r.E01 = q[5]; r.E11 = q[6];
    r.bits1 = (unsigned long long)__double_as_longlong(q[7]);
    return r;
}

DEVI double wave_total_mfma(double v) {
    const v4f64 z = {0.0, 0.0, 0.0, 0.0};
    const v4f64 a = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, z, 0, 0, 0);   // rows: sums over the four 16-lane groups
    const double u = (a[0] + a[1]) + (a[2] + a[3]);                              // four of the sixteen rows per lane group
    const v4f64 b = __builtin_amdgcn_mfma_f64_16x16x4f64(u, 1.0, z, 0, 0, 0);   // + over the four lane groups: the total, in every lane
    return b[0];
}
// all-ones / all-zeros lane mask from bit k of a wave-uniform word, as a double-typed select
DEVI double sel_by_bit(uint32_t bits /*uniform*/, int k, double if0, double if1) {
    // 0 / ~0 from bit k on the scalar unit (one s_bfe_i32), then one bit-field insert per register half
    const uint32_t m = (uint32_t)(((int32_t)(bits << (31 - k))) >> 31);
    const uint32_t lo = ((uint32_t)__double2loint(if1) & m) | ((uint32_t)__double2loint(if0) & ~m);
    const uint32_t hi = ((uint32_t)__double2hiint(if1) & m) | ((uint32_t)__double2hiint(if0) & ~m);
    return __hiloint2double((int)hi, (int)lo);
}

// Per-column scalars (column scale mantissas, hand-over sums) of the lean kernel: one 8-byte store per
// column measured ~200 cycles on the chain, so the collecting wave parks the value of column t in entry t & 63 of an
// LDS row (one LDS write per column) and writes 64 columns with ONE coalesced store.
struct ColScalars {
    double* row;                      // 64 doubles of LDS, this wave's own
    unsigned long long valid = 0ull;  // uniform: entries holding a value not yet written
    DEVI void put(uint32_t /*lane*/, uint64_t t /*uniform*/, double v /*the same in every lane*/) {
        row[t & 63u] = v;             // (every lane the same address and value: one LDS instruction, no selects)
        valid |= 1ull << (t & 63u);
    }
    DEVI void flush(gdouble* arr, uint32_t lane, uint64_t t_any /*uniform: any column of the 64-block held*/) {
        if ((valid >> lane) & 1ull) arr[(t_any & ~(uint64_t)63u) + lane] = row[lane];
        valid = 0ull;
    }
};

// The u_i of a wave's R = 16 rows without an LDS round trip: every lane also adds up the column sum of index
// i0 + (lane & 15) (each 16-lane DPP row then holds the wave's sixteen row values).  Round 2 pulled value k into all
// lanes with one v_mov_b64_dpp row_newbcast:k per row and added it with a second instruction; DP-ALU DPP (gfx90a+:
// row_newbcast is the one DPP control the 64-bit ALU operations take) does both in ONE: t = u[row lane k] + c.
template <int R>
DEVI void lean_u_rows(double urep, double (&ui)[R]) {
    static_assert(R == 16, "one DPP row of 16 lanes = the wave's 16 rows");
    ui[0] = row_bcast_f64<0>(urep);   ui[1] = row_bcast_f64<1>(urep);   ui[2] = row_bcast_f64<2>(urep);   ui[3] = row_bcast_f64<3>(urep);
    ui[4] = row_bcast_f64<4>(urep);   ui[5] = row_bcast_f64<5>(urep);   ui[6] = row_bcast_f64<6>(urep);   ui[7] = row_bcast_f64<7>(urep);
    ui[8] = row_bcast_f64<8>(urep);   ui[9] = row_bcast_f64<9>(urep);   ui[10] = row_bcast_f64<10>(urep); ui[11] = row_bcast_f64<11>(urep);
    ui[12] = row_bcast_f64<12>(urep); ui[13] = row_bcast_f64<13>(urep); ui[14] = row_bcast_f64<14>(urep); ui[15] = row_bcast_f64<15>(urep);
}
// acc += u[lane (l & ~15) + K] * s.  DP-ALU DPP: of the 64-bit ALU operations only the VOP2 ones (v_fmac_f64) have a
// DPP form, with row_newbcast as the one control.  The value `u` must come out of dpp_source() (a VALU result needs two
// wait states before a DPP instruction may read it, and the compiler does not look inside an asm statement).
template <int K>
DEVI double fmac_row_bcast(double acc, double u, double s) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(s), "n"(K));
    return acc;
}
// keeps the instruction that forms `v` where the source puts it (the compiler otherwise sinks work whose result is only
// needed behind a branch into that branch's fall-through block — out of the MFMA shadow it was written to fill)
DEVI void pin_here(double& v) { asm volatile("" : "+v"(v)); }
DEVI void lean_fence() { __builtin_amdgcn_sched_barrier(0); }
DEVI double dpp_source(double v) {
    double r;
    asm("v_mov_b64 %0, %1\n\ts_nop 1" : "=v"(r) : "v"(v));
    return r;
}
// Row-allele selects of the lean step: bit K of the wave-uniform row bits as an all-lanes mask (ONE scalar instruction,
// s_bfe_i64 with width 1 sign-extends the bit over the pair), consumed by v_cndmask_b32 as its mask operand — three
// instructions per 64-bit select where bit test + mask + two selects took four.
template <int K>
DEVI unsigned long long row_mask64(unsigned long long bits /*uniform*/) {
    unsigned long long m;
    asm("s_bfe_i64 %0, %1, %2" : "=s"(m) : "s"(bits), "n"((1 << 16) | K) : "scc");
    return m;
}
DEVI double sel_by_mask(double if0, double if1, unsigned long long m /*uniform: 0 or ~0*/) {
    uint32_t lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(if0)), "v"(__double2loint(if1)), "s"(m));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(if0)), "v"(__double2hiint(if1)), "s"(m));
    return __hiloint2double((int)hi, (int)lo);
}
template <int K, int N, class F>
DEVI void static_for(F&& f) {
    if constexpr (K < N) { f(std::integral_constant<int, K>{}); static_for<K + 1, N>(f); }
}
// base + byte SEL of `bytes` in ONE instruction (SDWA source select): the LDS address of a state's emission
template <int SEL>
DEVI uint32_t add_byte(uint32_t bytes, uint32_t base) {
    uint32_t r;
    if constexpr (SEL == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(bytes), "v"(base));
    else if constexpr (SEL == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(bytes), "v"(base));
    else if constexpr (SEL == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(bytes), "v"(base));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(bytes), "v"(base));
    return r;
}
// The lean step takes the emissions of a column by row PAIRS: a pair of rows has one of four allele combinations, a lane
// (column) one of two alleles, so the eight possible {e(row 2p, j), e(row 2p+1, j)} of a column record are laid out as a
// 128-byte table when its block is parked, next to one byte per row pair (32 x combination: the offset into the table).
// A pair's two emissions are then ONE SDWA add (lane's table column + the pair's byte) and ONE 16-byte LDS read —
// where bit test, mask and two selects per row took six instructions per pair.
template <class SH>
DEVI void lean_expand(SH& sh, uint32_t block /*uniform*/, uint32_t tid) {
    static_assert(sizeof(sh.comb[0][0]) == 32, "four waves of eight row pairs");
    if (tid >= 256u) return;
    const uint32_t b = block & 1u;
    {
        const uint32_t r = tid >> 2, c = tid & 3u;   // two of the block's 512 table entries per thread: (r, c, column allele 0 / 1)
        const double* rc = sh.rec[b][r];
        const double E00 = rc[4], E01 = rc[5], E11 = rc[6];
        sh.tab[b][r][c][0] = v2f64{(c & 1u) ? E01 : E00, (c & 2u) ? E01 : E00};
        sh.tab[b][r][c][1] = v2f64{(c & 1u) ? E11 : E01, (c & 2u) ? E11 : E01};
    }
    {
        const uint32_t r = tid >> 2, w = tid & 3u;   // the eight pair bytes of wave w's rows of record r
        const unsigned long long bits = (unsigned long long)__double_as_longlong(sh.rec[b][r][7]);
        const uint32_t b16 = (uint32_t)(bits >> (16u * w)) & 0xFFFFu;
        auto spread = [](uint32_t v8) { return ((v8 & 3u) | (((v8 >> 2) & 3u) << 8) | (((v8 >> 4) & 3u) << 16) | (((v8 >> 6) & 3u) << 24)) << 5; };
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        *(u32x2*)&sh.comb[b][r][w][0] = u32x2{spread(b16 & 0xFFu), spread(b16 >> 8)};
    }
}
// what a step needs to fetch the emissions of column `rel`: the lane's table column and the wave's pair bytes
struct LeanPairs { uint32_t tbase, cd0, cd1; };
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <class SH>
DEVI u32x2 lean_pair_bytes(const SH& sh, uint32_t rel /*uniform*/, uint32_t wave /*uniform*/) {   // (one broadcast LDS read: issue it early)
    return *(const u32x2*)&sh.comb[(rel / PG_LEAN_BLOCK) & 1u][rel % PG_LEAN_BLOCK][wave][0];
}
template <class SH>
DEVI LeanPairs lean_pairs(const SH& sh, uint32_t rel /*uniform*/, u32x2 cd, uint32_t aj /*0 / 1: the lane's column allele*/) {
    return LeanPairs{(uint32_t)(uintptr_t)(LAS const unsigned char*)&sh.tab[(rel / PG_LEAN_BLOCK) & 1u][rel % PG_LEAN_BLOCK][0][0] + aj * 16u, cd.x, cd.y};
}
template <class SH>
DEVI LeanPairs lean_pairs(const SH& sh, uint32_t rel, uint32_t wave, uint32_t aj) { return lean_pairs(sh, rel, lean_pair_bytes(sh, rel, wave), aj); }
template <int P>
DEVI v2f64 lean_pair(const LeanPairs& lp) {
    return *(LAS const v2f64*)(uintptr_t)add_byte<(P & 3)>(P < 4 ? lp.cd0 : lp.cd1, lp.tbase);
}
template <int R>
DEVI void lean_pairs_all(const LeanPairs& lp, double (&e)[R]) {
    static_for<0, R / 2>([&](auto pc) __attribute__((always_inline)) { constexpr int q = decltype(pc)::value; const v2f64 t = lean_pair<q>(lp); e[2 * q] = t.x; e[2 * q + 1] = t.y; });
}
template <int R>
DEVI double lean_colsum(const LeanShared<R>& sh, uint32_t pb, uint32_t lane) {
    if constexpr (R == 16)
        return (sh.psum[pb][0][lane] + sh.psum[pb][1][lane]) + (sh.psum[pb][2][lane] + sh.psum[pb][3][lane]);
    else
        return ((sh.psum[pb][0][lane] + sh.psum[pb][1][lane]) + (sh.psum[pb][2][lane] + sh.psum[pb][3][lane])) +
               ((sh.psum[pb][4][lane] + sh.psum[pb][5][lane]) + (sh.psum[pb][6][lane] + sh.psum[pb][7][lane]));
}



// ------------------------------------------------------------------------------------------
//  Chunk hand-over of the PERSISTENT chunked phase 2 (PHASE 4 of k_sweep_lean, k_post_loop; DevContig::sync).  The sweep of a
//  half-chain stores chunk after chunk into the rotating scratch buffers without leaving the CU; k_post_loop's blocks (on the CUs
//  the chains leave idle) form the posteriors of a chunk as soon as both roles have published it, and hand the buffer back.
//  Stores -> agent-scope release fence by every wave (the XCD's L2 is written back) -> workgroup barrier -> the flag; the
//  reader's acquire invalidates its own XCD's L2.  Every wait is bounded (wall clock): a partner that never arrives — a kernel
//  that was not co-resident — raises PG_DEVERR_SYNC_TIMEOUT instead of hanging the device.
// ------------------------------------------------------------------------------------------
#define PG_SYNC_TIMEOUT_TICKS 1000000000ull   // 10 s of the 100 MHz constant clock
#ifndef PG_POLL_SLEEPS
#define PG_POLL_SLEEPS 2   // s_sleep 127 (~3.4 us) between two polls of a chunk flag
#endif
// (The poll is a RELAXED load every ~7 us — an acquire per poll would invalidate the XCD's L2 every time, and 192 blocks polling
//  every half microsecond cost the chains beside them 15 % (profiles/r06_persist.txt); the acquire follows the successful poll.)
DEVI bool chunk_spin(uint32_t* flag, uint32_t want, uint32_t* err) {   // one thread
    bool ok = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
    if (!ok) {
        const unsigned long long t0 = wall_clock64();
        for (uint32_t it = 0;; ++it) {
#pragma unroll
            for (int z = 0; z < PG_POLL_SLEEPS; ++z) __builtin_amdgcn_s_sleep(127);
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; }
            if ((it & 63u) == 63u) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & PG_DEVERR_SYNC_TIMEOUT) break;   // (the partner gave up)
                if (wall_clock64() - t0 > PG_SYNC_TIMEOUT_TICKS) { atomicOr(err, PG_DEVERR_SYNC_TIMEOUT); break; }
            }
        }
    }
    if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}
DEVI void chunk_publish(uint32_t* flag, uint32_t value) {   // the whole workgroup
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
DEVI void chunk_wait_buffer(uint32_t* flag, uint32_t want, uint32_t* err) {   // the whole workgroup (nothing is READ behind it: the buffer is only written)
    if (threadIdx.x == 0 && !(kPersistExp & 2u)) (void)chunk_spin(flag, want, err);
    __syncthreads();
}

// A wave parks its 64 partial column sums — and (kLeanPreTotal, round 6) their TOTAL over the wave, formed in front of the step's
// barrier by six DPP levels (the sum ends in lane 63, which stores it): behind the barrier the column's total S is then three
// additions of four broadcast LDS reads, next to the column sums — instead of two dependent fp64 MFMAs behind them
// (profiles/r06_lean_chain.txt: the exchange -> column sums -> MFMA -> MFMA -> exponent chain was 756 of a column's 1384 cycles).
// a column store of the lean sweeps (-DPG_NT_STORES: as a non-temporal store — the columns are streamed out and never read by this kernel)
DEVI void lean_store(gdouble2* p, v2f64 v) {
    if constexpr (kNtStores) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int R>
DEVI void lean_put_sums(LeanShared<R>& sh, uint32_t pb, uint32_t wave, uint32_t lane, double part) {
    sh.psum[pb][wave][lane] = part;
    if constexpr (kLeanPreTotal) {
        double v = part;
        v += dpp_f64<0x111, 0xF, true>(v);   // row_shr:1
        v += dpp_f64<0x112, 0xF, true>(v);   // row_shr:2
        v += dpp_f64<0x114, 0xF, true>(v);   // row_shr:4
        v += dpp_f64<0x118, 0xF, true>(v);   // row_shr:8
        v += dpp_f64<0x142, 0xA, false>(v);  // row_bcast:15 -> rows 1,3
        v += dpp_f64<0x143, 0xC, false>(v);  // row_bcast:31 -> rows 2,3
        if (lane == 63u) sh.ptot[pb][wave] = v;
    }
}
template <int R>
DEVI double lean_total(const LeanShared<R>& sh, uint32_t pb) {   // (kLeanPreTotal) the column's total: every lane reads the waves' totals
    static_assert(R == 16, "four waves");
    const v2f64 a = *(const v2f64*)&sh.ptot[pb][0], b = *(const v2f64*)&sh.ptot[pb][2];
    return (a.x + a.y) + (b.x + b.y);
}

template <int PHASE, int R, bool TRI>
DEVI void lean_forward(const DevContig& dc, LeanShared<R>& sh, uint32_t C, uint32_t chunk) {
    constexpr int HP = 64;
    constexpr uint32_t RMASK = (1u << R) - 1u;
    const uint32_t mid = C / 2, K = dc.chunk_cols;
    uint32_t lo = PHASE == 1 ? 0u : mid, hi = PHASE == 1 ? mid : C;
    if constexpr (PHASE == 3) {
        const unsigned long long l = (unsigned long long)mid + (unsigned long long)chunk * K;
        if (l >= C) return;
        lo = (uint32_t)l;
        hi = C - lo > K ? lo + K : C;
    }
    if (lo >= hi) return;
    const uint32_t first = lo == 0 ? 1u : lo;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t i0 = wave * R;
    const size_t colsz = TRI ? (size_t)dc.col_stride : (size_t)HP * HP;
    const double unif = 1.0 / 4096.0;
    LeanRecs recs{(const GAS char*)dc.frec, (int64_t)first - 1, (int64_t)C, +1, tid};
    recs.park(sh, 0, recs.fetch(0));
    v2f64 piece = recs.fetch(1);
    lds_barrier();
    lean_expand(sh, 0, tid);
    lds_barrier();
    gdouble* fwd = (gdouble*)dc.fwd;
    gdouble* fscale = (gdouble*)dc.fscale;
    gu8* fallback = (gu8*)dc.fwd_fallback;
    gdouble* wr = fwd;
    gcdouble* resume = (gcdouble*)(fwd + (size_t)(lo > 0 ? lo - 1 : 0) * colsz);
    if constexpr (PHASE == 3) {
        gdouble* scr = (gdouble*)dc.scratch;
        wr = scr + (size_t)(PG_SCR_BUF(chunk) * 2u) * K * colsz - (size_t)lo * colsz;
        if (chunk > 0) resume = (gcdouble*)(scr + ((size_t)(PG_SCR_BUF(chunk - 1u) * 2u) * K + (K - 1u)) * colsz);
    }
    // PHASE 4: the whole second half in ONE launch, chunk after chunk into the scratch buffers (`chunk` = k_post_loop's blocks per chain)
    if constexpr (PHASE == 4) wr = (gdouble*)dc.scratch - (size_t)lo * colsz;
    const size_t toff = (size_t)(i0 >> 1) * HP + lane;  // this thread's first row pair inside a column (in 16-byte units)
    auto emis = [&](const FRec& r, double& eA, double& eB) {  // e(i, j) = row bit ? eB : eA for this lane's column allele
        const bool aj = (r.bits1 >> lane) & 1ull;
        eA = aj ? r.E01 : r.E00;
        eB = aj ? r.E11 : r.E01;
    };
    // TRI (lean chains of fused jobs, DevContig::tri): only the upper triangle is stored, the diagonal halved, the
    // element below the diagonal inside a straddling 16-byte unit as 0.  The per-pair factors / skip bits are
    // fixed per thread (row pair i0 + 2q, column = lane).
    double tfa[TRI ? R / 2 : 1], tfb[TRI ? R / 2 : 1];
    uint32_t tskip = 0, tunit[TRI ? R / 2 : 1];
    if constexpr (TRI) {
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const uint32_t r0 = i0 + 2u * (uint32_t)q;
            tfa[q] = lane < r0 ? 0.0 : (lane == r0 ? 0.5 : 1.0);
            tfb[q] = lane <= r0 ? 0.0 : (lane == r0 + 1u ? 0.5 : 1.0);
            const bool off_q = lane < (r0 & ~7u);  // whole 128-byte lines only: the lanes between are written as zeros
            tskip |= (off_q ? 1u : 0u) << q;
            tunit[q] = off_q ? 0u : tri_unit_of(r0 >> 1, lane);  // compact: the stored half is one contiguous 18 KB run
        }
    }
    auto put_pair = [&](gdouble2* dst, int q, double a, double b) __attribute__((always_inline)) {
        if constexpr (TRI) {
            if (!((tskip >> q) & 1u)) lean_store((dst - toff) + tunit[q], v2f64{a * tfa[q], b * tfb[q]});
        } else {
            lean_store(dst + (size_t)q * HP, v2f64{a, b});
        }
    };
    auto store_col = [&](uint32_t c, const double (&v)[R]) {
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + toff;
#pragma unroll
        for (int k = 0; k < R; k += 2) put_pair(dst, k >> 1, v[k], v[k + 1]);
    };
    auto flag_uniform = [&](uint32_t cprev) {
        if (cprev >= lo) {
            double xu[R];
#pragma unroll
            for (int k = 0; k < R; ++k) xu[k] = unif;
            store_col(cprev, xu);
        }
        if (wave == 0) fallback[cprev] = 1;
    };

    ColScalars fsc{&sh.scal[0][0]};
    double x[R];
    {
        const FRec r0 = read_frec(sh, 0);
        double eA, eB;
        emis(r0, eA, eB);
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(r0.bits1 >> i0) & RMASK));
        double part = 0.0;
        if (lo == 0) {
            const double P0 = ldexp(1.0, PG_BIAS_F);
            double pz[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { pz[k] = P0; x[k] = sel_by_bit(rb, k, eA, eB) * P0; part += x[k]; }
            store_col(0, pz);
            if (wave == 0) fscale[0] = 1.0;
        } else {
            gcdouble2* src = (gcdouble2*)resume + toff;
#pragma unroll
            for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; x[k] = t.x; x[k + 1] = t.y; }
            // (the partial sum exactly as the step forms it — part = fma(e, P', part), row after row — so that a column resumed from
            //  memory and one carried in registers (PHASE 4) hand the same bits to the next step: the result does not depend on
            //  where the chunk boundaries fall)
            if (!fallback[lo - 1]) {
#pragma unroll
                for (int k = 0; k < R; ++k) { const double e = sel_by_bit(rb, k, eA, eB); part = fma(e, x[k], part); x[k] *= e; }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) part += x[k];
            }
        }
        lean_put_sums<R>(sh, (first - 1) & 1u, wave, lane, part);
    }
    // Emissions of a column are fetched by row pairs (lean_pairs) DURING the previous step: a pair's next emissions are
    // read right after its two states used the current ones, so the LDS reads ride under the following pairs' arithmetic.
    // (Measured alternatives: all eight reads in the shadow of the second MFMA into a second array: 696 instead of 604 ns
    // per column — they sit in front of the total's consumer; product-column multiplies deferred into that shadow: 656.)
    double ec[R];
    {
        const FRec r1 = read_frec(sh, 1);
        lean_pairs_all<R>(lean_pairs(sh, 1, wave, (uint32_t)((r1.bits1 >> lane) & 1ull)), ec);   // column `first`
    }
    // One column step.  `cur` = record of this step (constants of the gap t-1 -> t), `nxt` takes the record of the next
    // step (four broadcast LDS reads, a whole step ahead of use); the loop calls the step twice with the two record
    // variables and the two emission arrays in swapped roles, so nothing is ever moved.
    LeanTimeline tl;
    tl.init();
    auto step = [&](uint32_t t, const FRec& cur, FRec& nxt) __attribute__((always_inline)) {
        const uint32_t n = t - first;                 // step number: reads the record with rel = n + 2
        tl.template mark<0>(0.0);                     // (behind the barrier of the step before)
        nxt = read_frec(sh, n + 2u);
        const u32x2 cdn = lean_pair_bytes(sh, n + 2u, wave);
        if (((n + 4u) % PG_LEAN_BLOCK) == 0u) {       // (uniform) a few columns before the next block is needed
            const uint32_t blk = (n + 4u) / PG_LEAN_BLOCK;
            recs.park(sh, blk, piece);
            piece = recs.fetch(blk + 1u);
        } else if (((n + 3u) % PG_LEAN_BLOCK) == 0u) {
            lean_expand(sh, (n + 3u) / PG_LEAN_BLOCK, tid);   // the block parked a step ago (a barrier lies between)
        }
        const uint32_t pb = (t - 1) & 1u;
        // both column sums' partials are fetched up front (this lane's column; the column of index i0 + (lane & 15):
        // the DPP source of the wave's sixteen u_i); everything that does not need the total S — the second sum, the
        // next column's table address and pair bytes — is issued between the two MFMAs of the total and fills their latency
        double pc[64 / R], pr[64 / R];
#pragma unroll
        for (int q = 0; q < 64 / R; ++q) { pc[q] = sh.psum[pb][q][lane]; pr[q] = sh.psum[pb][q][i0 + (lane & 15u)]; }
        const double Spre = kLeanPreTotal ? lean_total<R>(sh, pb) : 0.0;   // (issued with the other LDS reads)
        const double Cj = (pc[0] + pc[1]) + (pc[2] + pc[3]);
        tl.template mark<1>(Cj);                      // the column sums are back from LDS
        const v4f64 zz = {0.0, 0.0, 0.0, 0.0};
        const v4f64 ma = (kLeanDppSum || kLeanPreTotal) ? zz : __builtin_amdgcn_mfma_f64_16x16x4f64(Cj, 1.0, zz, 0, 0, 0);
        lean_fence();
        const double ucol = cur.c1 * Cj;
        const double urep = dpp_source(cur.c1 * ((pr[0] + pr[1]) + (pr[2] + pr[3])));
        const LeanPairs lp = lean_pairs(sh, n + 2u, cdn, (uint32_t)((nxt.bits1 >> lane) & 1ull));   // column t+1
        const double msum = (ma[0] + ma[1]) + (ma[2] + ma[3]);
        tl.template mark<2>(msum);                    // first MFMA + three adds
        lean_fence();
        const v4f64 mb = (kLeanDppSum || kLeanPreTotal) ? zz : __builtin_amdgcn_mfma_f64_16x16x4f64(msum, 1.0, zz, 0, 0, 0);
        if constexpr (!kLeanPreTotal) asm volatile("" :: "v"(mb));   // (the whole result stays allocated: a temporary in one of its registers would wait out the MFMA)
        lean_fence();
        double S = (kLeanExp & 4) ? 64.0 * Cj : (kLeanPreTotal ? Spre : (kLeanDppSum ? wave_sum(Cj) : mb[0]));
        tl.template mark<3>(S);                       // second MFMA: the total
        double uj = fma(cur.c2, S, ucol);
        double c0 = cur.c0;
        if (__builtin_expect(!(S > 0.0), 0)) {
            // column t-1 summed to zero: the uniform column takes its place (hmm.cpp:253-267), see forward_body
            // (every x and with it every column sum is 0: the whole uniform step rides on u_j)
            flag_uniform(t - 1);
            const double Cu = 64.0 * unif;
            S = 1.0;
            uj = fma(cur.c0, unif, fma(cur.c2, 1.0, 2.0 * cur.c1 * Cu));
            c0 = 0.0;
        }
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        const double m = ldexp(S, -es - PG_BIAS_F);
        const double sc = ldexp(1.0, -es), c0s = ldexp(c0, -es), ujs = ldexp(uj, -es);
        tl.template mark<4>(c0s + ujs + sc);          // zero test, exponent, scaled constants
        gdouble2* dst = (gdouble2*)(wr + (size_t)t * colsz) + toff;
        double part = 0.0, pprev = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double pk = fmac_row_bcast<k>(fma(c0s, x[k], ujs), urep, sc);   // P'_t(i0 + k, lane) 2^-es = c0 x + u_j + u_i
            part = fma(ec[k], pk, part);
            x[k] = ec[k] * pk;
            pin_here(x[k]);
            if constexpr (k == 1) tl.template mark<5>(x[1]);    // the first row pair's states
            // (the fence keeps every pair's store where it is: eight 1 KB stores issued back to back stall the wave
            // on the memory pipeline's queue — 787 instead of ~650 ns per column)
            if constexpr (k & 1) {
                if (!(kLeanExp & 1)) put_pair(dst, k >> 1, pprev, pk);
                const v2f64 t2 = (kLeanExp & 2) ? v2f64{0.5, 0.5} : lean_pair<(k >> 1)>(lp);   // e_{t+1} of this row pair
                ec[k - 1] = t2.x; ec[k] = t2.y;
                __builtin_amdgcn_sched_barrier(0);
            } else pprev = pk;
        });
        tl.template mark<6>(part);                    // the other seven row pairs: states, stores, next emissions
        lean_put_sums<R>(sh, t & 1u, wave, lane, part);
        if (wave == 0) {  // (scalar branch)
            fsc.put(lane, t, m);
            if ((t & 63u) == 63u) fsc.flush(fscale, lane, t);
        }
        tl.template mark<7>(0.0);                     // partial sum parked, per-column scalar
        lds_barrier();
        tl.template mark<8>(0.0);                     // the barrier
        tl.template fold<8>();
    };
    FRec ra = read_frec(sh, 1), rb2;
    // every load of the prologue has landed before the loop is entered: a register still "waiting for a load" at the loop
    // head would make the compiler put a vmcnt wait — a cap on the stores in flight — into every step
    __builtin_amdgcn_s_waitcnt(0x0F70);
    lds_barrier();
    uint32_t t = first;
    if constexpr (PHASE == 4) {
        // chunk q = columns [mid + q K, mid + (q + 1) K) into scratch buffer q % PG_SCRATCH_BUFS; the state stays in registers
        // (K is even — the host sees to it — so the two record variables are back in their roles at every chunk boundary)
        const uint32_t post_groups = 2u * (uint32_t)K;   // k_post_loop's work items per chunk: one per column slot
        gdouble* scr = (gdouble*)dc.scratch;
        uint32_t q = 0, cend = hi - lo > K ? lo + K : hi;
        for (;;) {
            for (; t + 1 < cend; t += 2) {
                step(t, ra, rb2);
                step(t + 1, rb2, ra);
            }
            if (t < cend) { step(t, ra, rb2); ++t; }   // (the last chunk only)
            // what a chunk launch of PHASE 3 does when it ends: the scalars still parked, the zero test of the chunk's last column
            if (wave == 0 && fsc.valid) fsc.flush(fscale, lane, cend - 1);
            {
                const double Cj = lean_colsum<R>(sh, (cend - 1) & 1u, lane);
                if (!(wave_total_mfma(Cj) > 0.0)) flag_uniform(cend - 1);
            }
            chunk_publish(dc.sync + 0, q + 1u);
            if (cend >= hi) break;
            ++q;
            lo = cend;
            cend = hi - lo > K ? lo + K : hi;
            wr = scr + (size_t)(PG_SCR_BUF(q) * 2u) * K * colsz - (size_t)lo * colsz;
            if (q >= PG_SCRATCH_BUFS) chunk_wait_buffer(dc.sync + PG_SYNC_DONE + PG_SCR_BUF(q), (q / PG_SCRATCH_BUFS) * post_groups, dc.err);
        }
        return;
    }
    for (; t + 1 < hi; t += 2) {
        step(t, ra, rb2);
        step(t + 1, rb2, ra);
    }
    if (t < hi) step(t, ra, rb2);
    if (wave == 0 && fsc.valid) fsc.flush(fscale, lane, hi - 1);
    if (kLeanTimeline && tid == 0) tl.write(dc.prof + 32);
    {   // the last column of this phase may itself have summed to zero
        const uint32_t pb = (hi - 1) & 1u;
        const double Cj = lean_colsum<R>(sh, pb, lane);
        if (!(wave_total_mfma(Cj) > 0.0)) flag_uniform(hi - 1);
    }
}

template <int PHASE, int R, bool TRI>
DEVI void lean_backward(const DevContig& dc, LeanShared<R>& sh, uint32_t C, uint32_t chunk) {
    constexpr int HP = 64;
    constexpr uint32_t RMASK = (1u << R) - 1u;
    const int64_t mid = C / 2, K = dc.chunk_cols;
    int64_t top = PHASE == 1 ? (int64_t)C - 1 : mid - 1;
    int64_t bot = PHASE == 1 ? mid : 0;
    if constexpr (PHASE == 3) {
        top = mid - 1 - (int64_t)chunk * K;
        if (top < 0) return;
        bot = top - K + 1 > 0 ? top - K + 1 : 0;
    }
    if (top < bot) return;
    const int64_t t0 = PHASE == 1 ? top - 1 : top;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t i0 = wave * R;
    const size_t colsz = TRI ? (size_t)dc.col_stride : (size_t)HP * HP;
    const double unif = 1.0 / 4096.0;
    LeanRecs recs{(const GAS char*)dc.frec, t0 + 1, (int64_t)C, -1, tid};
    recs.park(sh, 0, recs.fetch(0));
    v2f64 piece = recs.fetch(1);
    lds_barrier();
    lean_expand(sh, 0, tid);
    lds_barrier();
    gdouble* cols = (gdouble*)dc.fwd;
    gdouble* bscale = (gdouble*)dc.bscale;
    gdouble* bsum = (gdouble*)dc.bsum;
    gdouble* wr = cols;
    gcdouble* resume = (gcdouble*)(cols + (size_t)(top + 1 < (int64_t)C ? top + 1 : top) * colsz);
    if constexpr (PHASE == 3) {
        gdouble* scr = (gdouble*)dc.scratch;
        wr = scr + (size_t)(PG_SCR_BUF(chunk) * 2u + 1u) * (size_t)K * colsz - (size_t)bot * colsz;
        if (chunk > 0) resume = (gcdouble*)(scr + (size_t)(PG_SCR_BUF(chunk - 1u) * 2u + 1u) * (size_t)K * colsz);
    }
    int64_t cbot = bot;   // PHASE 4 (see lean_forward): the lowest column of the chunk being stored
    if constexpr (PHASE == 4) {
        cbot = top - K + 1 > 0 ? top - K + 1 : 0;
        wr = (gdouble*)dc.scratch + (size_t)K * colsz - (size_t)cbot * colsz;
    }
    const size_t toff = (size_t)(i0 >> 1) * HP + lane;
    auto emis = [&](const FRec& r, double& eA, double& eB) {
        const bool aj = (r.bits1 >> lane) & 1ull;
        eA = aj ? r.E01 : r.E00;
        eB = aj ? r.E11 : r.E01;
    };
    double tfa[TRI ? R / 2 : 1], tfb[TRI ? R / 2 : 1];  // see lean_forward
    uint32_t tskip = 0, tunit[TRI ? R / 2 : 1];
    if constexpr (TRI) {
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const uint32_t r0 = i0 + 2u * (uint32_t)q;
            tfa[q] = lane < r0 ? 0.0 : (lane == r0 ? 0.5 : 1.0);
            tfb[q] = lane <= r0 ? 0.0 : (lane == r0 + 1u ? 0.5 : 1.0);
            const bool off_q = lane < (r0 & ~7u);  // whole 128-byte lines only: the lanes between are written as zeros
            tskip |= (off_q ? 1u : 0u) << q;
            tunit[q] = off_q ? 0u : tri_unit_of(r0 >> 1, lane);  // compact: the stored half is one contiguous 18 KB run
        }
    }
    auto put_pair = [&](gdouble2* dst, int q, double a, double b) __attribute__((always_inline)) {
        if constexpr (TRI) {
            if (!((tskip >> q) & 1u)) lean_store((dst - toff) + tunit[q], v2f64{a * tfa[q], b * tfb[q]});
        } else {
            lean_store(dst + (size_t)q * HP, v2f64{a, b});
        }
    };
    auto store_col = [&](int64_t c, const double (&v)[R]) {
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + toff;
#pragma unroll
        for (int k = 0; k < R; k += 2) put_pair(dst, k >> 1, v[k], v[k + 1]);
    };

    ColScalars bsc{&sh.scal[0][0]}, bsm{&sh.scal[1][0]};
    double w[R], Sy;
    FRec cur = read_frec(sh, 0);  // record t0+1: constants of the gap t0 -> t0+1, emission of column t0+1
    {
        double y[R];
        if constexpr (PHASE == 1) {
            // column C-1: beta~ = 1 (hmm.cpp:356-358), stored at the backward bias
            const double B0 = ldexp(1.0, PG_BIAS_B);
#pragma unroll
            for (int k = 0; k < R; ++k) y[k] = B0;
            Sy = 4096.0 * B0;
            store_col(top, y);
            if (wave == 0) { bscale[top] = 1.0; bsum[top] = Sy; }
        } else {
            gcdouble2* src = (gcdouble2*)resume + toff;
#pragma unroll
            for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; y[k] = t.x; y[k + 1] = t.y; }
            Sy = bsum[top + 1];
            if (!(Sy > 0.0)) {  // resuming behind an all-zero column: uniform (hmm.cpp:374-380)
#pragma unroll
                for (int k = 0; k < R; ++k) y[k] = unif;
                Sy = 1.0;
            }
        }
        double eA, eB;
        emis(cur, eA, eB);
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(cur.bits1 >> i0) & RMASK));
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) { const double e = sel_by_bit(rb, k, eA, eB); part = fma(e, y[k], part); w[k] = y[k] * e; }   // (as the step: see lean_forward)
        lean_put_sums<R>(sh, (uint32_t)t0 & 1u, wave, lane, part);
    }
    double ec[R];   // (see lean_forward)
    {
        const FRec r1 = read_frec(sh, 1);   // column t0: its emission goes into the first step's w
        lean_pairs_all<R>(lean_pairs(sh, 1, wave, (uint32_t)((r1.bits1 >> lane) & 1ull)), ec);
    }
    double one = 1.0;   // (in a register for the whole sweep: the DPP form of v_fmac_f64 takes no constant)
    asm volatile("" : "+v"(one));
    // One column step (see lean_forward): `cur` = record t+1 (constants of the gap t -> t+1), `nxt` takes record t
    // (constants of the next step); `ec` = emissions of column t (this step's w), `en` takes those of column t-1.
    LeanTimeline tl;
    tl.init();
    auto step = [&](int64_t t, const FRec& cur, FRec& nxt) __attribute__((always_inline)) {
        const uint32_t n = (uint32_t)(t0 - t);        // step number: column t is the record with rel = n + 1
        tl.template mark<0>(0.0);
        if (((n + 4u) % PG_LEAN_BLOCK) == 0u) {
            const uint32_t blk = (n + 4u) / PG_LEAN_BLOCK;
            recs.park(sh, blk, piece);
            piece = recs.fetch(blk + 1u);
        } else if (((n + 3u) % PG_LEAN_BLOCK) == 0u) {
            lean_expand(sh, (n + 3u) / PG_LEAN_BLOCK, tid);
        }
        int es = exponent_of(Sy) - PG_BIAS_B;
        es = es < -900 ? -900 : es;
        const double m = ldexp(Sy, -es - PG_BIAS_B);
        if (wave == 1) { asm volatile("" ::: "memory"); bsc.put(lane, (uint64_t)t, m); }   // (the per-column scalars are collected by two DIFFERENT waves: the waves meet
                                                        // at a barrier every column, one wave's extra instructions are everybody's wait)
        double k0 = ldexp(cur.c0, -es), k1 = ldexp(cur.c1, -es), k2 = ldexp(cur.c2, -es), kap = ldexp(cur.kappa, -es);
        pin_here(k0); pin_here(k1); pin_here(k2); pin_here(kap);   // (in front of the barrier, not behind it)
        tl.template mark<1>(k0 + kap);                // exponent, scaled constants, scale mantissa parked
        lds_barrier();
        tl.template mark<2>(0.0);                     // the barrier
        const uint32_t pb = (uint32_t)t & 1u;
        double pc[64 / R], pr[64 / R];  // (see lean_forward)
#pragma unroll
        for (int q = 0; q < 64 / R; ++q) { pc[q] = sh.psum[pb][q][lane]; pr[q] = sh.psum[pb][q][i0 + (lane & 15u)]; }
        const double Spre = kLeanPreTotal ? lean_total<R>(sh, pb) : 0.0;
        // the records of the next step are read HERE, behind the barrier and behind the column sums (in front of the barrier
        // the wave would wait out their LDS latency: the barrier's release waits for every outstanding LDS operation)
        lean_fence();
        nxt = read_frec(sh, n + 1u);
        const unsigned long long bits_n = (unsigned long long)__double_as_longlong(sh.rec[((n + 2u) / PG_LEAN_BLOCK) & 1u][(n + 2u) % PG_LEAN_BLOCK][7]);  // column t-1
        const u32x2 cdn = lean_pair_bytes(sh, n + 2u, wave);
        lean_fence();
        const double Cj = (pc[0] + pc[1]) + (pc[2] + pc[3]);
        tl.template mark<3>(Cj);                      // the column sums are back from LDS (the next records read behind them)
        const v4f64 zz = {0.0, 0.0, 0.0, 0.0};
        const v4f64 ma = (kLeanDppSum || kLeanPreTotal) ? zz : __builtin_amdgcn_mfma_f64_16x16x4f64(Cj, 1.0, zz, 0, 0, 0);
        lean_fence();
        const double ucol = k1 * Cj;
        const double urep = dpp_source(k1 * ((pr[0] + pr[1]) + (pr[2] + pr[3])));  // u_i of row i0 + (lane & 15)
        const LeanPairs lp = lean_pairs(sh, n + 2u, cdn, (uint32_t)((bits_n >> lane) & 1ull));   // column t-1
        const double msum = (ma[0] + ma[1]) + (ma[2] + ma[3]);
        tl.template mark<4>(msum);                    // first MFMA + three adds
        lean_fence();
        const v4f64 mb = (kLeanDppSum || kLeanPreTotal) ? zz : __builtin_amdgcn_mfma_f64_16x16x4f64(msum, 1.0, zz, 0, 0, 0);
        if constexpr (!kLeanPreTotal) asm volatile("" :: "v"(mb));
        lean_fence();
        const double Sw = (kLeanExp & 4) ? 64.0 * Cj : (kLeanPreTotal ? Spre : (kLeanDppSum ? wave_sum(Cj) : mb[0]));
        tl.template mark<5>(Sw);                      // second MFMA: the total
        const double uj = fma(k2, Sw, ucol);
        const double Snew = kap * Sw;  // = sum(beta'_t)
        Sy = Snew;                     // (1 behind an all-zero column, below)
        gdouble2* dst = (gdouble2*)(wr + (size_t)t * colsz) + toff;
        double part = 0.0, yprev = 0.0;
        const bool zero = !(Snew > 0.0);   // beta~_t is all zero (below)
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double yk = fmac_row_bcast<k>(fma(k0, w[k], uj), urep, one);  // beta'_t = k0 w + u_j + u_i
            part = fma(ec[k], yk, part);
            w[k] = ec[k] * yk;
            pin_here(w[k]);
            if constexpr (k == 1) tl.template mark<6>(w[1]);    // u_j, the first row pair's states
            if constexpr (k & 1) {
                if (!(kLeanExp & 1)) put_pair(dst, k >> 1, yprev, yk);
                const v2f64 t2 = (kLeanExp & 2) ? v2f64{0.5, 0.5} : lean_pair<(k >> 1)>(lp);   // e_{t-1} of this row pair
                ec[k - 1] = t2.x; ec[k] = t2.y;
                __builtin_amdgcn_sched_barrier(0);
            } else yprev = yk;
        });
        if (__builtin_expect(zero, 0)) {
            // beta~_t is all zero (a sum of non-negative terms: every y_k above IS 0, and so is what was stored): its own
            // posteriors are 0, the next step starts from the uniform column (hmm.cpp:374-380)
            double et[R];   // (ec holds the next column's emissions by now: this column's are fetched again)
            lean_pairs_all<R>(lean_pairs(sh, n + 1u, wave, (uint32_t)((nxt.bits1 >> lane) & 1ull)), et);
            part = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) { w[k] = unif * et[k]; part += w[k]; }
            Sy = 1.0;
        }
        tl.template mark<7>(part);                    // the other seven row pairs: states, stores, next emissions
        lean_put_sums<R>(sh, (uint32_t)(t - 1) & 1u, wave, lane, part);
        if (wave == 2) { asm volatile("" ::: "memory"); bsm.put(lane, (uint64_t)t, Snew); }   // (a branch, not predication: three of the four waves skip it)
        if (((uint64_t)t & 63u) == 0u) {
            if (wave == 1) bsc.flush(bscale, lane, (uint64_t)t);
            if (wave == 2) bsm.flush(bsum, lane, (uint64_t)t);
        }
        tl.template mark<8>(0.0);                     // partial sum parked, per-column scalars
        tl.template fold<8>();
    };
    FRec rb2;
    int64_t t = t0;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (see lean_forward: no load of the prologue is still in flight inside the loop)
    if constexpr (PHASE == 4) {
        const uint32_t post_groups = 2u * (uint32_t)K;   // k_post_loop's work items per chunk: one per column slot
        gdouble* scr = (gdouble*)dc.scratch;
        uint32_t q = 0;
        for (;;) {
            for (; t - 1 >= cbot; t -= 2) {
                step(t, cur, rb2);
                step(t - 1, rb2, cur);
            }
            if (t >= cbot) { step(t, cur, rb2); --t; }   // (the last chunk only: K is even)
            if (wave == 1 && bsc.valid) bsc.flush(bscale, lane, (uint64_t)cbot);
            if (wave == 2 && bsm.valid) bsm.flush(bsum, lane, (uint64_t)cbot);
            chunk_publish(dc.sync + 1, q + 1u);
            if (cbot == 0) break;
            ++q;
            const int64_t ntop = cbot - 1;
            cbot = ntop - K + 1 > 0 ? ntop - K + 1 : 0;
            wr = scr + (size_t)(PG_SCR_BUF(q) * 2u + 1u) * (size_t)K * colsz - (size_t)cbot * colsz;
            if (q >= PG_SCRATCH_BUFS) chunk_wait_buffer(dc.sync + PG_SYNC_DONE + PG_SCR_BUF(q), (q / PG_SCRATCH_BUFS) * post_groups, dc.err);
        }
        return;
    }
    for (; t - 1 >= bot; t -= 2) {
        step(t, cur, rb2);
        step(t - 1, rb2, cur);
    }
    if (t >= bot) step(t, cur, rb2);
    if (wave == 1 && bsc.valid) bsc.flush(bscale, lane, (uint64_t)bot);
    if (wave == 2 && bsm.valid) bsm.flush(bsum, lane, (uint64_t)bot);
    if (kLeanTimeline && tid == 0) tl.write(dc.prof + 48);
}

// ------------------------------------------------------------------------------------------
//  Lean PHASE 2 (fused jobs, triangle chains — DevContig::tri, C >= 2): the second half of a half-chain with the
//  posterior partials formed inline.  Same step as lean_forward / lean_backward; instead of storing the column the
//  thread multiplies it with the partner column (beta' for the forward role, P' for the backward role — stored by
//  phase 1 as a compact upper triangle) and adds the products up by the row allele.  No loader waves and no LDS
//  ring: every thread fetches its own (at most 8) 16-byte units of the partner column two columns ahead into
//  registers, so a workgroup is 4 waves with ~200 VGPRs and TWO of them share a CU (the general phase-2 kernel:
//  6 waves of 241 VGPRs, one workgroup per CU).  The partials of the four waves are added up through LDS (riding on
//  the step's barrier) before they leave: 1 KB per column instead of 4 (k_bins reads 64 entries per column).
// ------------------------------------------------------------------------------------------
template <int R>
struct LeanShared2 {
    LeanShared<R> a;
    v2f64 ppart[2][64 / R][64];  // posterior partials {row allele 0, row allele 1} per wave and lane, by step parity
};
template <int R>
struct LeanTri {  // this thread's units of a compact triangle column
    // unit of (row pair rp, lane) = (a number that depends on the row pair only) + lane: the row pairs of a wave are
    // uniform, so the per-pair part lives in scalar registers and ONE vector register (16 * lane) serves all pairs
    uint32_t sbase[R / 2];   // (uniform) byte offset of the pair's lane-0 unit (may be "negative": wraps, lanes below 8 g are not loaded)
    uint32_t lane16, valid;
    DEVI void setup(uint32_t i0 /*uniform*/, uint32_t lane) {
        valid = 0;
        lane16 = lane * 16u;
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const uint32_t rp = (i0 >> 1) + (uint32_t)q, g = rp >> 2;
            sbase[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)((256u * g - 16u * g * (g - 1u) + (rp & 3u) * (64u - 8u * g) - 8u * g) * 16u));
            valid |= (lane >= 8u * g ? 1u : 0u) << q;
        }
    }
    // units below the stored half keep the zeros the buffer was initialised with
    DEVI void load(gcdouble* col /*uniform*/, double (&v)[R]) const {
#pragma unroll
        for (int q = 0; q < R / 2; ++q)
            if ((valid >> q) & 1u) {
                const GAS char* cb = (const GAS char*)col + (int32_t)sbase[q];   // (uniform: scalar base of the load)
                const v2f64 t = *(const GAS v2f64*)(cb + lane16);
                v[2 * q] = t.x; v[2 * q + 1] = t.y;
            }
    }
};
// full element (i0 + k, lane) of a compact triangle column: the stored (min, max), the diagonal doubled (once per launch)
template <int R>
DEVI void lean_load_mirrored(gcdouble* col, uint32_t i0, uint32_t lane, double (&v)[R]) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const uint32_t i = i0 + (uint32_t)k, a = i <= lane ? i : lane, b = i <= lane ? lane : i;
        const double val = col[(size_t)tri_unit_of(a >> 1, b) * 2 + (a & 1u)];
        v[k] = a == b ? 2.0 * val : val;
    }
}
// Partials of column c -> its four bin sums.  The four waves' {acc0, acc1} (row allele 0 / 1, per column j = lane) of
// step parity pb are added per lane; wave w then sums ONE of the four (row allele, column allele) classes over the 64
// lanes and stores one double: part[c][w], w = 2 * (row allele) + (column allele) — 32 bytes per column instead of the
// 1 KB of per-lane partials that k_bins used to read back and reduce (same additions in the same order as before).
template <int R>
DEVI void lean2_flush_partials(const LeanShared2<R>& sh, uint32_t pb, gdouble* part, size_t c, uint32_t wave, uint32_t lane,
                               unsigned long long bits1 /* of column c */) {
    static_assert(64 / R == 4, "one class per wave");
    v2f64 t = sh.ppart[pb][0][lane];
#pragma unroll
    for (int w = 1; w < 64 / R; ++w) { const v2f64 o = sh.ppart[pb][w][lane]; t.x += o.x; t.y += o.y; }
    const bool aj = (bits1 >> lane) & 1ull;
    const double mine = (wave & 2u) ? t.y : t.x;  // (scalar select)
    const double s = wave_sum(aj == ((wave & 1u) != 0u) ? mine : 0.0);
    if (lane == 0) part[c * 4u + wave] = s;
}

template <int R>
DEVI void lean2_forward(const DevContig& dc, LeanShared2<R>& sh2, uint32_t C) {
    constexpr int HP = 64;
    constexpr uint32_t RMASK = (1u << R) - 1u;
    LeanShared<R>& sh = sh2.a;
    const uint32_t mid = C / 2, lo = mid, hi = C, first = lo;  // C >= 2: lo >= 1
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t i0 = wave * R;
    const size_t colsz = dc.col_stride;  // compact triangles
    const double unif = 1.0 / 4096.0;
    LeanRecs recs{(const GAS char*)dc.frec, (int64_t)first - 1, (int64_t)C, +1, tid};
    recs.park(sh, 0, recs.fetch(0));
    v2f64 piece = recs.fetch(1);
    lds_barrier();
    lean_expand(sh, 0, tid);
    gcdouble* cols = (gcdouble*)dc.fwd;
    gdouble* fscale = (gdouble*)dc.fscale;
    gu8* fallback = (gu8*)dc.fwd_fallback;
    gdouble* part = (gdouble*)dc.part;
    auto emis = [&](const FRec& r, double& eA, double& eB) {
        const bool aj = (r.bits1 >> lane) & 1ull;
        eA = aj ? r.E01 : r.E00;
        eB = aj ? r.E11 : r.E01;
    };
    LeanTri<R> tri;
    tri.setup(i0, lane);
    // partner columns beta'_t, two ahead of their use
    double b0[R], b1[R], b2[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { b0[k] = 0.0; b1[k] = 0.0; b2[k] = 0.0; }
    tri.load(cols + (size_t)lo * colsz, b0);
    if (lo + 1 < C) tri.load(cols + (size_t)(lo + 1) * colsz, b1);

    ColScalars fsc{&sh.scal[0][0]};
    double x[R];
    {
        const FRec r0 = read_frec(sh, 0);
        double eA, eB;
        emis(r0, eA, eB);
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(r0.bits1 >> i0) & RMASK));
        lean_load_mirrored<R>(cols + (size_t)(lo - 1) * colsz, i0, lane, x);  // P'_{lo-1}, stored by phase 1 of this role
        if (!fallback[lo - 1]) {
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] *= sel_by_bit(rb, k, eA, eB);
        }
        double part0 = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) part0 += x[k];
        sh.psum[(first - 1) & 1u][wave][lane] = part0;
    }
    unsigned long long pbits = 0;  // column alleles of the column whose partials are pending
    lds_barrier();
    // column alleles and row-pair bytes of the column a step produces: read a step ahead and carried
    unsigned long long cbits = read_frec(sh, 1).bits1;
    u32x2 ccd = lean_pair_bytes(sh, 1, wave);
    // One column step.  `ba` = the partner column beta'_t, `bc` takes beta'_{t+2} (two columns ahead of its use; the three
    // buffers rotate through the loop's three calls, so no column is ever copied).  Emissions and class multipliers
    // of the sixteen states come pair by pair out of LDS tables (lean_expand), fetched one pair ahead.
    auto step = [&](uint32_t t, const double (&ba)[R], double (&bc)[R]) __attribute__((always_inline)) {
        const uint32_t n = t - first;
        const double* rq = sh.rec[((n + 1u) / PG_LEAN_BLOCK) & 1u][(n + 1u) % PG_LEAN_BLOCK];   // record t: constants of the gap t-1 -> t
        const double rc0 = rq[0], rc1 = rq[1], rc2 = rq[2];
        const unsigned long long nbits = (unsigned long long)__double_as_longlong(sh.rec[((n + 2u) / PG_LEAN_BLOCK) & 1u][(n + 2u) % PG_LEAN_BLOCK][7]);
        const u32x2 ncd = lean_pair_bytes(sh, n + 2u, wave);
        if (((n + 4u) % PG_LEAN_BLOCK) == 0u) {
            const uint32_t blk = (n + 4u) / PG_LEAN_BLOCK;
            recs.park(sh, blk, piece);
            piece = recs.fetch(blk + 1u);
        } else if (((n + 3u) % PG_LEAN_BLOCK) == 0u) {
            lean_expand(sh, (n + 3u) / PG_LEAN_BLOCK, tid);   // the block parked a step ago (a barrier lies between)
        }
        if (t + 2 < C) tri.load(cols + (size_t)(t + 2) * colsz, bc);  // two columns ahead (bc was last read a step ago)
        if (t > first) lean2_flush_partials<R>(sh2, (t - 1) & 1u, part, (size_t)(t - 1), wave, lane, pbits);
        const uint32_t pb = (t - 1) & 1u;
        const double Cj = lean_colsum<R>(sh, pb, lane);
        const double ucol = rc1 * Cj;
        const double urep = dpp_source(rc1 * lean_colsum<R>(sh, pb, i0 + (lane & 15u)));   // u_i of row i0 + (lane & 15): the DPP source
        const LeanPairs lp = lean_pairs(sh, n + 1u, ccd, (uint32_t)((cbits >> lane) & 1ull));
        const v4f64 zz = {0.0, 0.0, 0.0, 0.0};
        const v4f64 ma = __builtin_amdgcn_mfma_f64_16x16x4f64(Cj, 1.0, zz, 0, 0, 0);
        double S = __builtin_amdgcn_mfma_f64_16x16x4f64((ma[0] + ma[1]) + (ma[2] + ma[3]), 1.0, zz, 0, 0, 0)[0];
        double uj = fma(rc2, S, ucol);
        double c0 = rc0;
        if (__builtin_expect(!(S > 0.0), 0)) {
            // column t-1 summed to zero: the uniform column takes its place (hmm.cpp:253-267); its own partials were
            // formed from the all-zero column — k_bins re-forms those bins from the flag
            if (wave == 0) fallback[t - 1] = 1;
            const double Cu = 64.0 * unif;
            S = 1.0;
            uj = fma(rc0, unif, fma(rc2, 1.0, 2.0 * rc1 * Cu));
            c0 = 0.0;
        }
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        const double m = ldexp(S, -es - PG_BIAS_F);
        const double sc = ldexp(1.0, -es), c0s = ldexp(c0, -es), ujs = ldexp(uj, -es);
        double part0 = 0.0, acc0 = 0.0, acc1 = 0.0;
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(cbits >> i0) & RMASK));   // row alleles: uniform
        v2f64 en = lean_pair<0>(lp);
        static_for<0, R / 2>([&](auto pc) __attribute__((always_inline)) {
            constexpr int q = decltype(pc)::value, k = 2 * q;
            const v2f64 e = en;
            if constexpr (q + 1 < R / 2) en = lean_pair<q + 1>(lp);
            const double pa = fmac_row_bcast<k>(fma(c0s, x[k], ujs), urep, sc);           // P'_t = c0 x + u_j + u_i (lean_forward)
            const double pb2 = fmac_row_bcast<k + 1>(fma(c0s, x[k + 1], ujs), urep, sc);
            x[k] = pa * e.x;
            part0 += x[k];
            x[k + 1] = pb2 * e.y;
            part0 += x[k + 1];
            const double pra = pa * ba[k], prb = pb2 * ba[k + 1];  // P'_t beta'_t (0 below the stored half)
            const bool bita = (rb >> k) & 1u, bitb = (rb >> (k + 1)) & 1u;
            acc1 = fma(pra, bita ? 1.0 : 0.0, acc1);  // exact 0/1 multipliers (scalar operands): rounds like a predicated add
            acc0 = fma(pra, bita ? 0.0 : 1.0, acc0);
            acc1 = fma(prb, bitb ? 1.0 : 0.0, acc1);
            acc0 = fma(prb, bitb ? 0.0 : 1.0, acc0);
        });
        sh.psum[t & 1u][wave][lane] = part0;
        sh2.ppart[t & 1u][wave][lane] = v2f64{acc0, acc1};
        if (wave == 0) {
            fsc.put(lane, t, m);
            if ((t & 63u) == 63u) fsc.flush(fscale, lane, t);
        }
        pbits = cbits; cbits = nbits; ccd = ncd;
        lds_barrier();
    };
    {
        uint32_t t = first;
        for (; t + 2 < hi; t += 3) {
            step(t, b0, b2);
            step(t + 1, b1, b0);
            step(t + 2, b2, b1);
        }
        if (t < hi) step(t, b0, b2);
        if (t + 1 < hi) step(t + 1, b1, b0);
    }
    lean2_flush_partials<R>(sh2, (hi - 1) & 1u, part, (size_t)(hi - 1), wave, lane, pbits);
    if (wave == 0 && fsc.valid) fsc.flush(fscale, lane, hi - 1);
    {   // the last column may itself have summed to zero
        const double Cj = lean_colsum<R>(sh, (hi - 1) & 1u, lane);
        if (!(wave_total_mfma(Cj) > 0.0) && wave == 0) fallback[hi - 1] = 1;
    }
}

template <int R>
DEVI void lean2_backward(const DevContig& dc, LeanShared2<R>& sh2, uint32_t C) {
    constexpr int HP = 64;
    constexpr uint32_t RMASK = (1u << R) - 1u;
    LeanShared<R>& sh = sh2.a;
    const int64_t mid = C / 2, top = mid - 1, bot = 0, t0 = top;  // C >= 2: top >= 0, column mid exists
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t i0 = wave * R;
    const size_t colsz = dc.col_stride;  // compact triangles
    const double unif = 1.0 / 4096.0;
    LeanRecs recs{(const GAS char*)dc.frec, t0 + 1, (int64_t)C, -1, tid};
    recs.park(sh, 0, recs.fetch(0));
    v2f64 piece = recs.fetch(1);
    lds_barrier();
    lean_expand(sh, 0, tid);
    gcdouble* cols = (gcdouble*)dc.fwd;
    gdouble* bscale = (gdouble*)dc.bscale;
    gcdouble* bsum = (gcdouble*)dc.bsum;
    gdouble* part = (gdouble*)dc.part;
    auto emis = [&](const FRec& r, double& eA, double& eB) {
        const bool aj = (r.bits1 >> lane) & 1ull;
        eA = aj ? r.E01 : r.E00;
        eB = aj ? r.E11 : r.E01;
    };
    LeanTri<R> tri;
    tri.setup(i0, lane);
    // partner columns P'_t (forward, stored by phase 1), two ahead of their use
    double b0[R], b1[R], b2[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { b0[k] = 0.0; b1[k] = 0.0; b2[k] = 0.0; }
    tri.load(cols + (size_t)t0 * colsz, b0);
    if (t0 - 1 >= 0) tri.load(cols + (size_t)(t0 - 1) * colsz, b1);

    ColScalars bsc{&sh.scal[0][0]};
    double w[R], Sy;
    FRec cur = read_frec(sh, 0);  // record t0+1: constants of the gap t0 -> t0+1, emission of column t0+1
    {
        double y[R];
        lean_load_mirrored<R>(cols + (size_t)(top + 1) * colsz, i0, lane, y);  // beta'_{mid}, stored by phase 1 of this role
        Sy = bsum[top + 1];
        if (!(Sy > 0.0)) {  // resuming behind an all-zero column: uniform (hmm.cpp:374-380)
#pragma unroll
            for (int k = 0; k < R; ++k) y[k] = unif;
            Sy = 1.0;
        }
        double eA, eB;
        emis(cur, eA, eB);
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(cur.bits1 >> i0) & RMASK));
        double part0 = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) { w[k] = y[k] * sel_by_bit(rb, k, eA, eB); part0 += w[k]; }
        sh.psum[(uint32_t)t0 & 1u][wave][lane] = part0;
    }
    double one = 1.0;   // (in a register: the DPP form of v_fmac_f64 takes no constant)
    asm volatile("" : "+v"(one));
    lds_barrier();   // (the tables of block 0 are complete)
    // carried from step to step (see lean2_forward): the constants of the gap t -> t+1, the column alleles and row-pair
    // bytes of column t, the column alleles of column t+1 (whose partials are pending)
    double kc0 = cur.c0, kc1 = cur.c1, kc2 = cur.c2, kck = cur.kappa;
    unsigned long long pbits = cur.bits1;
    unsigned long long cbits = read_frec(sh, 1).bits1;
    u32x2 ccd = lean_pair_bytes(sh, 1, wave);
    auto step = [&](int64_t t, const double (&ba)[R], double (&bc)[R]) __attribute__((always_inline)) {
        const uint32_t n = (uint32_t)(t0 - t);   // column t is the record with rel = n + 1
        if (((n + 4u) % PG_LEAN_BLOCK) == 0u) {
            const uint32_t blk = (n + 4u) / PG_LEAN_BLOCK;
            recs.park(sh, blk, piece);
            piece = recs.fetch(blk + 1u);
        } else if (((n + 3u) % PG_LEAN_BLOCK) == 0u) {
            lean_expand(sh, (n + 3u) / PG_LEAN_BLOCK, tid);
        }
        if (t - 2 >= 0) tri.load(cols + (size_t)(t - 2) * colsz, bc);
        int es = exponent_of(Sy) - PG_BIAS_B;
        es = es < -900 ? -900 : es;
        const double m = ldexp(Sy, -es - PG_BIAS_B);
        if (wave == 0) bsc.put(lane, (uint64_t)t, m);
        const double k0 = ldexp(kc0, -es), k1 = ldexp(kc1, -es), k2 = ldexp(kc2, -es), kap = ldexp(kck, -es);
        lds_barrier();
        // the next step's constants (record t: gap t-1 -> t), column alleles and pair bytes (column t-1)
        const double* rq = sh.rec[((n + 1u) / PG_LEAN_BLOCK) & 1u][(n + 1u) % PG_LEAN_BLOCK];
        const double nc0 = rq[0], nc1 = rq[1], nc2 = rq[2], nck = rq[3];
        const unsigned long long nbits = (unsigned long long)__double_as_longlong(sh.rec[((n + 2u) / PG_LEAN_BLOCK) & 1u][(n + 2u) % PG_LEAN_BLOCK][7]);
        const u32x2 ncd = lean_pair_bytes(sh, n + 2u, wave);
        if (t < t0) lean2_flush_partials<R>(sh2, (uint32_t)(t + 1) & 1u, part, (size_t)(t + 1), wave, lane, pbits);
        const double Cj = lean_colsum<R>(sh, (uint32_t)t & 1u, lane);
        const double ucol = k1 * Cj;
        const double urep = dpp_source(k1 * lean_colsum<R>(sh, (uint32_t)t & 1u, i0 + (lane & 15u)));
        const LeanPairs lp = lean_pairs(sh, n + 1u, ccd, (uint32_t)((cbits >> lane) & 1ull));   // emission of column t: this step's w
        const v4f64 zz = {0.0, 0.0, 0.0, 0.0};
        const v4f64 ma = __builtin_amdgcn_mfma_f64_16x16x4f64(Cj, 1.0, zz, 0, 0, 0);
        const double Sw = __builtin_amdgcn_mfma_f64_16x16x4f64((ma[0] + ma[1]) + (ma[2] + ma[3]), 1.0, zz, 0, 0, 0)[0];
        const double uj = fma(k2, Sw, ucol);
        const double Snew = kap * Sw;  // = sum(beta'_t)
        double part0 = 0.0, acc0 = 0.0, acc1 = 0.0;
        if (__builtin_expect(!(Snew > 0.0), 0)) {
            // beta~_t is all zero: its own posteriors are 0, the next step starts from the uniform column
            double et[R];
            lean_pairs_all<R>(lp, et);
#pragma unroll
            for (int k = 0; k < R; ++k) { w[k] = unif * et[k]; part0 += w[k]; }
        } else {
            const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(cbits >> i0) & RMASK));
            v2f64 en = lean_pair<0>(lp);
            static_for<0, R / 2>([&](auto pc) __attribute__((always_inline)) {
                constexpr int q = decltype(pc)::value, k = 2 * q;
                const v2f64 e = en;
                if constexpr (q + 1 < R / 2) en = lean_pair<q + 1>(lp);
                const double ya = fmac_row_bcast<k>(fma(k0, w[k], uj), urep, one);           // beta'_t = k0 w + u_j + u_i (lean_backward)
                const double yb = fmac_row_bcast<k + 1>(fma(k0, w[k + 1], uj), urep, one);
                w[k] = ya * e.x;
                part0 += w[k];
                w[k + 1] = yb * e.y;
                part0 += w[k + 1];
                const double pra = ba[k] * ya, prb = ba[k + 1] * yb;  // P'_t beta'_t
                const bool bita = (rb >> k) & 1u, bitb = (rb >> (k + 1)) & 1u;
                acc1 = fma(pra, bita ? 1.0 : 0.0, acc1);
                acc0 = fma(pra, bita ? 0.0 : 1.0, acc0);
                acc1 = fma(prb, bitb ? 1.0 : 0.0, acc1);
                acc0 = fma(prb, bitb ? 0.0 : 1.0, acc0);
            });
        }
        sh.psum[(uint32_t)(t - 1) & 1u][wave][lane] = part0;
        sh2.ppart[(uint32_t)t & 1u][wave][lane] = v2f64{acc0, acc1};
        if (wave == 0 && ((uint64_t)t & 63u) == 0u) bsc.flush(bscale, lane, (uint64_t)t);
        Sy = Snew > 0.0 ? Snew : 1.0;
        kc0 = nc0; kc1 = nc1; kc2 = nc2; kck = nck;
        pbits = cbits; cbits = nbits; ccd = ncd;
    };
    {
        int64_t t = t0;
        for (; t - 2 >= bot; t -= 3) {
            step(t, b0, b2);
            step(t - 1, b1, b0);
            step(t - 2, b2, b1);
        }
        if (t >= bot) step(t, b0, b2);
        if (t - 1 >= bot) step(t - 1, b1, b0);
    }
    lds_barrier();
    lean2_flush_partials<R>(sh2, (uint32_t)bot & 1u, part, (size_t)bot, wave, lane, pbits);
    if (wave == 0 && bsc.valid) bsc.flush(bscale, lane, (uint64_t)bot);
}

template <int R>
__global__ __launch_bounds__((64 * 64 / R)) __attribute__((amdgpu_waves_per_eu(2, 2)))  // two workgroups per CU: <= 256 registers per lane
void k_sweep_lean2(const DevContig* __restrict__ contigs) {
    __shared__ LeanShared2<R> sh;
    const DevContig& dc = contigs[blockIdx.x];
    if (!dc.lean || dc.tri != 2u) return;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C < 2) return;  // (a single column: the general kernel)
    if (blockIdx.y == 0) lean2_forward<R>(dc, sh, C);
    else lean2_backward<R>(dc, sh, C);
}

// (shared by the two kernels below)
template <int PHASE, int R, bool TRI>
DEVI void sweep_lean_body(const DevContig* __restrict__ contigs, uint32_t chunk, LeanShared<R>& sh);

// phase 1 of fused jobs (triangle stores): hundreds of chains, two workgroups per CU — at most 256 registers per lane
template <int PHASE, int R>
__global__ __launch_bounds__((64 * 64 / R)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_sweep_lean_tri(const DevContig* __restrict__ contigs, uint32_t chunk) {
    __shared__ LeanShared<R> sh;
    sweep_lean_body<PHASE, R, true>(contigs, chunk, sh);
}

template <int PHASE, int R, bool TRI = false>
__global__ __launch_bounds__((64 * 64 / R)) void k_sweep_lean(const DevContig* __restrict__ contigs, uint32_t chunk) {
    __shared__ LeanShared<R> sh;
    sweep_lean_body<PHASE, R, TRI>(contigs, chunk, sh);
}

template <int PHASE, int R, bool TRI>
DEVI void sweep_lean_body(const DevContig* __restrict__ contigs, uint32_t chunk, LeanShared<R>& sh) {
    const DevContig& dc = contigs[blockIdx.x];
    if (dc.lean != 1u) return;
    if ((dc.tri != 0u) != TRI) return;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    const unsigned long long t_begin = kChainProf ? __builtin_amdgcn_s_memtime() : 0ull;
    if (blockIdx.y == 0) lean_forward<PHASE, R, TRI>(dc, sh, C, chunk);
    else lean_backward<PHASE, R, TRI>(dc, sh, C, chunk);
    if (kChainProf && threadIdx.x == 0) {  // -DPG_CHAIN_PROF builds only: cycles of this role's launch (last chunk wins)
        unsigned long long* o = dc.prof + (blockIdx.y == 0 ? 0 : 16) + (PHASE == 1 ? 0 : 8);
        o[0] = __builtin_amdgcn_s_memtime() - t_begin;
    }
}


// ------------------------------------------------------------------------------------------
//  k_sweep_leanx : the store-only phases (1, 3) of chains at HP = 128 whose objects all have at most PG_AMAX alleles
//  (every column NARROW: its emission table is the 6 x 6 one inside the column record) — BASELINE configs[4], the
//  HPRC-style panel with multiallelic objects, H <= 128.  The lean step (k_sweep_lean: column sums through LDS, wave
//  totals by two fp64 MFMAs, u_i by v_fmac_f64_dpp row_newbcast, scale folded into the constants, one dependent segment
//  per column in BOTH directions — the general kernel's backward role at R = 32 runs two) with the emission of a state
//  taken from the column's table instead of a two-way select:
//    * the FULL column records travel through LDS in blocks of 16 (every thread fetches one 16-byte piece of the block
//      after next: one global load per thread per 16 columns); when a block is parked its allele bytes are rewritten in
//      place as table row offsets min(a, 5) * 48 (phantom paths, allele 255, hit the zero row / column);
//    * a state's emission is E[a_i][a_j] = one LDS read at (lane's table column) + (row offset, a scalar bit-field
//      extract of the wave-uniform row alleles): three issue slots, what the biallelic select costs — so there is ONE
//      path for biallelic and multiallelic columns alike;
//    * row alleles, the lane's column allele and the transition constants of a column are read one step ahead.
//  Layout as the general kernel's at HP = 128: 8 waves, wave -> (row group of 32 rows, 64-column block), lane =
//  column, 32 states per lane; two waves per SIMD.  Same stored columns, scales, fall-back rules and resume
//  conventions as the general kernel (k_post and the fused phase 2 read what this kernel writes).
// ------------------------------------------------------------------------------------------
template <int HP>
struct LxCfg {
    static constexpr int R = HP / 4;            // rows per wave: four row groups
    static constexpr int NCH = HP / 64;         // 64-column blocks
    static constexpr int NW = 4 * NCH;          // waves
    static constexpr int T = 64 * NW;
    static constexpr int NS = R / 16;           // DPP rows' worth of u_i per wave
    static constexpr int RB = (PG_REC_ALLELES + HP + 63) & ~63;
    static constexpr int BLK = 16;              // records per LDS block
    static constexpr int PIECES = BLK * RB / 16;
    static constexpr int PPT = (PIECES + T - 1) / T;   // 16-byte pieces per thread and block
    static_assert(R % 16 == 0 && (BLK * HP / 4) % T == 0, "bad lean-x geometry");
};
template <int HP>
struct LxShared {
    using Cfg = LxCfg<HP>;
    double psum[2][4][HP];
    double scal[2][64];   // per-column scalars on their way out (ColScalars)
    unsigned char rec[2][Cfg::BLK][Cfg::RB] __attribute__((aligned(16)));
};
template <int HP>
struct LxRecs {
    using Cfg = LxCfg<HP>;
    const GAS char* base;   // colrec
    int64_t origin, C;      // column of rel 0, number of columns
    int dir;                // +1 forward, -1 backward
    uint32_t tid;
    // (threads past the last piece — only when PIECES is not a multiple of T — fetch the last one again and park nothing:
    // no branch around the load, so nothing waits for it where it is issued)
    DEVI v2f64 fetch(uint32_t block, int p) const {
        uint32_t piece = tid + (uint32_t)p * Cfg::T;
        if constexpr (Cfg::PIECES % Cfg::T != 0) piece = piece < (uint32_t)Cfg::PIECES ? piece : (uint32_t)Cfg::PIECES - 1u;
        const uint32_t q = piece / (uint32_t)(Cfg::RB / 16), w = piece % (uint32_t)(Cfg::RB / 16);
        int64_t c = origin + (int64_t)dir * ((int64_t)block * Cfg::BLK + q);
        c = c < 0 ? 0 : (c >= C ? C - 1 : c);
        return *(const GAS v2f64*)(base + (size_t)c * Cfg::RB + w * 16u);
    }
    DEVI void park(LxShared<HP>& sh, uint32_t block, int p, v2f64 v) const {
        const uint32_t piece = tid + (uint32_t)p * Cfg::T;
        if (Cfg::PIECES % Cfg::T == 0 || piece < (uint32_t)Cfg::PIECES) ((v2f64*)&sh.rec[block & 1u][0][0])[piece] = v;
    }
};
// allele bytes of a freshly parked block -> table row offsets min(a, 5) * 48 (in place)
template <int HP>
DEVI void lx_transform(LxShared<HP>& sh, uint32_t block, uint32_t tid) {
    using Cfg = LxCfg<HP>;
    auto f = [](uint32_t v) {  // two bytes in the 16-bit halves
        uint32_t a = v & 0xFFFFu, b = v >> 16;
        a = a > (uint32_t)PG_AMAX ? (uint32_t)PG_AMAX : a;
        b = b > (uint32_t)PG_AMAX ? (uint32_t)PG_AMAX : b;
        return (a * (uint32_t)(PG_ESTRIDE * 8)) | ((b * (uint32_t)(PG_ESTRIDE * 8)) << 16);
    };
#pragma unroll
    for (uint32_t d = tid; d < (uint32_t)(Cfg::BLK * HP / 4); d += (uint32_t)Cfg::T) {
        uint32_t* ptr = (uint32_t*)(&sh.rec[block & 1u][d / (uint32_t)(HP / 4)][PG_REC_ALLELES]) + d % (uint32_t)(HP / 4);
        const uint32_t v = *ptr;
        *ptr = f(v & 0x00FF00FFu) | (f((v >> 8) & 0x00FF00FFu) << 8);
    }
}
template <int HP>
DEVI const unsigned char* lx_rec(const LxShared<HP>& sh, uint32_t rel /*uniform*/) {
    return sh.rec[(rel / (uint32_t)LxCfg<HP>::BLK) & 1u][rel % (uint32_t)LxCfg<HP>::BLK];
}
struct LxConsts { double c0, c1, c2, kappa; };
template <int HP>
DEVI LxConsts lx_consts(const LxShared<HP>& sh, uint32_t rel) {
    const double* q = (const double*)lx_rec(sh, rel);
    return LxConsts{q[0], q[1], q[2], q[3]};
}
// what a step needs of a column's alleles: the lane's table column (byte offset a_j * 8 inside a table row) and the
// row offsets of the wave's R rows (R bytes, the same in every lane: broadcast LDS reads)
template <int R>
struct LxAlleles { uint32_t col8; uint32_t rows[R / 4]; };
template <int HP>
DEVI LxAlleles<LxCfg<HP>::R> lx_alleles(const LxShared<HP>& sh, uint32_t rel, uint32_t j, uint32_t i0) {
    constexpr int R = LxCfg<HP>::R;
    const unsigned char* al = lx_rec(sh, rel) + PG_REC_ALLELES;
    LxAlleles<R> a;
    a.col8 = ((uint32_t)al[j] * 171u) >> 10;   // (row offset a * 48) / 6 = a * 8, a <= 5
#pragma unroll
    for (int q = 0; q < R / 4; ++q) a.rows[q] = ((const uint32_t*)(al + i0))[q];
    return a;
}
// LDS address of the lane's table column inside record `rel`
template <int HP>
DEVI uint32_t lx_ecol(const LxShared<HP>& sh, uint32_t rel, uint32_t col8) {
    return (uint32_t)(uintptr_t)(LAS const unsigned char*)(lx_rec(sh, rel) + PG_REC_E) + col8;
}
template <int HP, int K>
DEVI double lx_emission(uint32_t ecol, const LxAlleles<LxCfg<HP>::R>& a) {
    return *(LAS const double*)(uintptr_t)add_byte<(K & 3)>(a.rows[K >> 2], ecol);
}

// TRI (phase 1 at 64 paths, round 6): the column is stored as its upper triangle — k_sweep_lean_tri's layout (diagonal halved,
// zeros below it inside a stored line, nothing in the lines left of it) — for fused jobs whose 64-path chains carry
// multiallelic objects (DevContig::tri == 1): phase 2 reads it through the general kernel's triangle ring.
template <int R>
struct LxTri {
    double fa[R / 2], fb[R / 2];   // factor of the pair's two elements
    uint32_t off[R / 2];           // byte offset of the pair's unit inside the compact column
    uint32_t skip;                 // bit q: the unit's line is not stored
    DEVI void setup(uint32_t i0, uint32_t j) {
        skip = 0;
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const uint32_t r0 = i0 + 2u * (uint32_t)q;
            fa[q] = j < r0 ? 0.0 : (j == r0 ? 0.5 : 1.0);
            fb[q] = j <= r0 ? 0.0 : (j == r0 + 1u ? 0.5 : 1.0);
            const bool off_q = j < (r0 & ~7u);
            skip |= (off_q ? 1u : 0u) << q;
            off[q] = off_q ? 0u : tri_unit_of(r0 >> 1, j) * 16u;
        }
    }
    DEVI void put(GAS char* col, int q, double a, double b) const {
        if (!((skip >> q) & 1u)) *(gdouble2*)(col + off[q]) = v2f64{a * fa[q], b * fb[q]};
    }
};
template <int PHASE, int HP, bool TRI = false, bool WIDE = false>
DEVI void leanx_forward(const DevContig& dc, LxShared<HP>& sh, uint32_t C, uint32_t chunk) {
    static_assert(!TRI || (PHASE == 1 && HP == 64), "triangle stores: phase 1 at 64 paths");
    using Cfg = LxCfg<HP>;
    constexpr int R = Cfg::R, NS = Cfg::NS, BLK = Cfg::BLK, PPT = Cfg::PPT;
    const uint32_t mid = C / 2, K = dc.chunk_cols;
    uint32_t lo = PHASE == 1 ? 0u : mid, hi = PHASE == 1 ? mid : C;
    if constexpr (PHASE == 3) {
        const unsigned long long l = (unsigned long long)mid + (unsigned long long)chunk * K;
        if (l >= C) return;
        lo = (uint32_t)l;
        hi = C - lo > K ? lo + K : C;
    }
    if (lo >= hi) return;
    const uint32_t first = lo == 0 ? 1u : lo;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t rg = wave / (uint32_t)Cfg::NCH, i0 = rg * R, j = (wave % (uint32_t)Cfg::NCH) * 64u + lane;
    const uint32_t H = dc.H;
    const size_t colsz = TRI ? (size_t)dc.col_stride : (size_t)HP * HP;
    const double unif = 1.0 / ((double)H * (double)H);
    LxRecs<HP> recs{(const GAS char*)dc.colrec, (int64_t)first - 1, (int64_t)C, +1, tid};
    v2f64 piece[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) recs.park(sh, 0, p, recs.fetch(0, p));
#pragma unroll
    for (int p = 0; p < PPT; ++p) piece[p] = recs.fetch(1, p);
    lds_barrier();
    lx_transform<HP>(sh, 0, tid);
    lds_barrier();
    gdouble* fwd = (gdouble*)dc.fwd;
    gdouble* fscale = (gdouble*)dc.fscale;
    gu8* fallback = (gu8*)dc.fwd_fallback;
    gdouble* wr = fwd;
    gcdouble* resume = (gcdouble*)(fwd + (size_t)(lo > 0 ? lo - 1 : 0) * colsz);
    if constexpr (PHASE == 3) {
        gdouble* scr = (gdouble*)dc.scratch;
        wr = scr + (size_t)(PG_SCR_BUF(chunk) * 2u) * K * colsz - (size_t)lo * colsz;
        if (chunk > 0) resume = (gcdouble*)(scr + ((size_t)(PG_SCR_BUF(chunk - 1u) * 2u) * K + (K - 1u)) * colsz);
    }
    const size_t toff = (size_t)(i0 >> 1) * HP + j;  // this thread's first row pair inside a column (in 16-byte units)
    // the stores of a step: R/8 per-thread pointers (four row pairs each, reached by immediates), advanced by one column
    // per step — one 64-bit add each instead of an address computation per store
    GAS char* sp[R / 8];
#pragma unroll
    for (int g = 0; g < R / 8; ++g) sp[g] = (GAS char*)(wr + (size_t)first * colsz) + toff * 16u + (size_t)(4 * g + 2) * (size_t)(HP * 16);
    LxTri<TRI ? R : 2> ltri;
    if constexpr (TRI) ltri.setup(i0, j);
    auto store_col = [&](uint32_t c, const double (&v)[R]) {
        if constexpr (TRI) {
            GAS char* col = (GAS char*)(wr + (size_t)c * colsz);
#pragma unroll
            for (int k = 0; k < R; k += 2) ltri.put(col, k >> 1, v[k], v[k + 1]);
            return;
        }
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + toff;
#pragma unroll
        for (int k = 0; k < R; k += 2) dst[(size_t)(k >> 1) * HP] = v2f64{v[k], v[k + 1]};
    };
    auto flag_uniform = [&](uint32_t cprev) {
        if (cprev >= lo) {
            double xu[R];
#pragma unroll
            for (int k = 0; k < R; ++k) xu[k] = (j < H && i0 + (uint32_t)k < H) ? unif : 0.0;
            store_col(cprev, xu);
        }
        if (tid == 0) fallback[cprev] = 1;
    };
    auto gather = [&](auto lo_c, auto hi_c, uint32_t ecol, const LxAlleles<R>& al, double (&e)[R]) __attribute__((always_inline)) {
        static_for<decltype(lo_c)::value, decltype(hi_c)::value>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; e[k] = lx_emission<HP, k>(ecol, al); });
    };
    using I0 = std::integral_constant<int, 0>; using IH = std::integral_constant<int, R / 2>; using IR = std::integral_constant<int, R>;

    // DevContig::widef (phase 1 of triangle chains whose objects include wide ones, round 6): the emissions of a WIDE column do not
    // come from the record's 6 x 6 table (the LDS record's allele bytes are clamped table-row offsets) but from the column's side
    // table, indexed by the raw local alleles of the column-order record in memory — a rare column pays two dependent loads
    // (a kernel of its own, k_sweep_leanx_triw: with the branch in it the step of the chains WITHOUT wide columns ran 7 % slower)
    auto wide_fix = [&](uint32_t rel, int64_t c, double (&ev)[R]) __attribute__((always_inline)) {
        if constexpr (TRI && WIDE) {
            if (c < 0 || c >= (int64_t)C) return;
            const unsigned char* lrec = lx_rec<HP>(sh, rel);
            const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lrec[PG_REC_NLOCAL]);
            if (nl <= (uint32_t)PG_AMAX) return;
            const uint32_t woff = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t*)(lrec + PG_REC_WIDE_IDX));
            const GAS double* Ew = (const GAS double*)(const GAS unsigned char*)(dc.wide + (size_t)woff * 16u);
            const GAS unsigned char* al = (const GAS unsigned char*)dc.colrec + (size_t)c * Cfg::RB + PG_REC_ALLELES;
            uint32_t aj = al[j];
            aj = aj > nl ? nl : aj;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                uint32_t ai = al[i0 + (uint32_t)k];
                ai = ai > nl ? nl : ai;
                ev[k] = Ew[ai * (nl + 1u) + aj];
            }
        }
    };
    ColScalars fsc{&sh.scal[0][0]};
    double x[R], e[R];
    {
        const LxAlleles<R> a0 = lx_alleles<HP>(sh, 0, j, i0);
        gather(I0{}, IR{}, lx_ecol<HP>(sh, 0, a0.col8), a0, e);
        wide_fix(0, (int64_t)first - 1, e);
        double part = 0.0;
        if (lo == 0) {
            const double P0 = ldexp(1.0, PG_BIAS_F);
            double pz[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { pz[k] = P0; x[k] = e[k] * P0; part += x[k]; }
            store_col(0, pz);
            if (tid == 0) fscale[0] = 1.0;
        } else {
            gcdouble2* src = (gcdouble2*)resume + toff;
#pragma unroll
            for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; x[k] = t.x; x[k + 1] = t.y; }
            if (!fallback[lo - 1]) {
#pragma unroll
                for (int k = 0; k < R; ++k) x[k] *= e[k];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) part += x[k];
        }
        sh.psum[(first - 1) & 1u][rg][j] = part;
    }
    LxConsts cur = lx_consts<HP>(sh, 1);            // column `first`: constants of the gap first-1 -> first
    {
        const LxAlleles<R> a1 = lx_alleles<HP>(sh, 1, j, i0);   // and its emissions (every step fetches those of the next)
        gather(I0{}, IR{}, lx_ecol<HP>(sh, 1, a1.col8), a1, e);
        wide_fix(1, (int64_t)first, e);
    }
    double one = 1.0;
    asm volatile("" : "+v"(one));
    auto step = [&](uint32_t t) __attribute__((always_inline)) {
        const uint32_t n = t - first;                 // step number: column t is the record with rel = n + 1
        if (((n + 4u) % (uint32_t)BLK) == 0u) {       // (uniform) a few columns before the next block is needed
            const uint32_t blk = (n + 4u) / (uint32_t)BLK;
#pragma unroll
            for (int p = 0; p < PPT; ++p) { recs.park(sh, blk, p, piece[p]); piece[p] = recs.fetch(blk + 1u, p); }
        } else if (((n + 3u) % (uint32_t)BLK) == 0u) {
            lx_transform<HP>(sh, (n + 3u) / (uint32_t)BLK, tid);   // the block parked a step ago (a barrier lies between)
        }
        const uint32_t pb = (t - 1) & 1u;
        double pc[4], po[4], pr[NS][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pc[q] = sh.psum[pb][q][j];
            po[q] = Cfg::NCH == 2 ? sh.psum[pb][q][j ^ 64u] : 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) pr[s][q] = sh.psum[pb][q][i0 + 16u * (uint32_t)s + (lane & 15u)];
        }
        lean_fence();
        const double Cj = (pc[0] + pc[1]) + (pc[2] + pc[3]);
        const double Call = Cfg::NCH == 2 ? Cj + ((po[0] + po[1]) + (po[2] + po[3])) : Cj;
        const v4f64 zz = {0.0, 0.0, 0.0, 0.0};
        const v4f64 ma = __builtin_amdgcn_mfma_f64_16x16x4f64(Call, 1.0, zz, 0, 0, 0);
        // (in the shadows of the two MFMAs of the total: the u terms, the next column's constants and alleles)
        const double ucol = cur.c1 * Cj;
        double urep[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) urep[s] = dpp_source(cur.c1 * ((pr[s][0] + pr[s][1]) + (pr[s][2] + pr[s][3])));
        const LxConsts cnx = lx_consts<HP>(sh, n + 2u);
        const LxAlleles<R> anx = lx_alleles<HP>(sh, n + 2u, j, i0);
        lean_fence();
        const double msum = (ma[0] + ma[1]) + (ma[2] + ma[3]);
        const v4f64 mb = __builtin_amdgcn_mfma_f64_16x16x4f64(msum, 1.0, zz, 0, 0, 0);
        const uint32_t ecoln = lx_ecol<HP>(sh, n + 2u, anx.col8);   // the lane's table column of the NEXT column
        asm volatile("" :: "v"(mb));   // (the whole result stays allocated: a temporary in one of its registers would wait out the MFMA)
        lean_fence();
        double S = mb[0];
        double uj = fma(cur.c2, S, ucol);
        double c0 = cur.c0;
        if (__builtin_expect(!(S > 0.0), 0)) {
            // column t-1 summed to zero: the uniform column takes its place (hmm.cpp:253-267), see forward_body
            flag_uniform(t - 1);
            const double Cu = (double)H * unif;
            S = 1.0;
            uj = fma(cur.c0, unif, fma(cur.c2, 1.0, 2.0 * cur.c1 * Cu));
            c0 = 0.0;
        }
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        const double m = ldexp(S, -es - PG_BIAS_F);
        const double sc = ldexp(1.0, -es), c0s = ldexp(c0, -es), ujs = ldexp(uj, -es);
        double part = 0.0, pprev = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double pk = fmac_row_bcast<(k & 15)>(fma(c0s, x[k], ujs), urep[k >> 4], sc);   // P'_t(i0 + k, j) 2^-es
            part = fma(e[k], pk, part);
            x[k] = e[k] * pk;
            pin_here(x[k]);
            if (!(kLxExp & 2)) e[k] = lx_emission<HP, k>(ecoln, anx);   // e_{t+1}(i0 + k, j): the LDS reads of the next step ride under this step's arithmetic
            if constexpr (k & 1) {
                if constexpr (TRI) ltri.put((GAS char*)(wr + (size_t)t * colsz), k >> 1, pprev, pk);
                else if (!(kLxExp & 1)) *(gdouble2*)(sp[k >> 3] + (((k >> 1) & 3) - 2) * (HP * 16)) = v2f64{pprev, pk};
                __builtin_amdgcn_sched_barrier(0);
            }
            else pprev = pk;
        });
        wide_fix(n + 2u, (int64_t)t + 1, e);
        sh.psum[t & 1u][rg][j] = part;
        if (wave == 0) {  // (scalar branch)
            fsc.put(lane, t, m);
            if ((t & 63u) == 63u) fsc.flush(fscale, lane, t);
        }
        cur = cnx;
#pragma unroll
        for (int g = 0; g < R / 8; ++g) { sp[g] += colsz * 8u; asm("" : "+v"(sp[g])); }   // (opaque: kept as R/8 separate induction pointers)
        lds_barrier();
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (no load of the prologue in flight inside the loop: see lean_forward)
    lds_barrier();
    for (uint32_t t = first; t < hi; ++t) step(t);
    if (wave == 0 && fsc.valid) fsc.flush(fscale, lane, hi - 1);
    {   // the last column of this phase may itself have summed to zero
        const uint32_t pb = (hi - 1) & 1u;
        double Call = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) Call += sh.psum[pb][q][j] + (Cfg::NCH == 2 ? sh.psum[pb][q][j ^ 64u] : 0.0);
        if (!(wave_total_mfma(Call) > 0.0)) flag_uniform(hi - 1);
    }
}

template <int PHASE, int HP, bool TRI = false, bool WIDE = false>
DEVI void leanx_backward(const DevContig& dc, LxShared<HP>& sh, uint32_t C, uint32_t chunk) {
    static_assert(!TRI || (PHASE == 1 && HP == 64), "triangle stores: phase 1 at 64 paths");
    using Cfg = LxCfg<HP>;
    constexpr int R = Cfg::R, NS = Cfg::NS, BLK = Cfg::BLK, PPT = Cfg::PPT;
    const int64_t mid = C / 2, K = dc.chunk_cols;
    int64_t top = PHASE == 1 ? (int64_t)C - 1 : mid - 1;
    int64_t bot = PHASE == 1 ? mid : 0;
    if constexpr (PHASE == 3) {
        top = mid - 1 - (int64_t)chunk * K;
        if (top < 0) return;
        bot = top - K + 1 > 0 ? top - K + 1 : 0;
    }
    if (top < bot) return;
    const int64_t t0 = PHASE == 1 ? top - 1 : top;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t rg = wave / (uint32_t)Cfg::NCH, i0 = rg * R, j = (wave % (uint32_t)Cfg::NCH) * 64u + lane;
    const uint32_t H = dc.H;
    const size_t colsz = TRI ? (size_t)dc.col_stride : (size_t)HP * HP;
    const double unif = 1.0 / ((double)H * (double)H);
    LxRecs<HP> recs{(const GAS char*)dc.colrec, t0 + 1, (int64_t)C, -1, tid};   // rel r = column t0 + 1 - r
    v2f64 piece[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) recs.park(sh, 0, p, recs.fetch(0, p));
#pragma unroll
    for (int p = 0; p < PPT; ++p) piece[p] = recs.fetch(1, p);
    lds_barrier();
    lx_transform<HP>(sh, 0, tid);
    lds_barrier();
    gdouble* cols = (gdouble*)dc.fwd;
    gdouble* bscale = (gdouble*)dc.bscale;
    gdouble* bsum = (gdouble*)dc.bsum;
    gdouble* wr = cols;
    gcdouble* resume = (gcdouble*)(cols + (size_t)(top + 1 < (int64_t)C ? top + 1 : top) * colsz);
    if constexpr (PHASE == 3) {
        gdouble* scr = (gdouble*)dc.scratch;
        wr = scr + (size_t)(PG_SCR_BUF(chunk) * 2u + 1u) * (size_t)K * colsz - (size_t)bot * colsz;
        if (chunk > 0) resume = (gcdouble*)(scr + (size_t)(PG_SCR_BUF(chunk - 1u) * 2u + 1u) * (size_t)K * colsz);
    }
    const size_t toff = (size_t)(i0 >> 1) * HP + j;
    GAS char* sp[R / 8];   // (see leanx_forward; filled in below, once `wr` is known)
    LxTri<TRI ? R : 2> ltri;   // (see leanx_forward)
    if constexpr (TRI) ltri.setup(i0, j);
    auto store_col = [&](int64_t c, const double (&v)[R]) {
        if constexpr (TRI) {
            GAS char* col = (GAS char*)(wr + (size_t)c * colsz);
#pragma unroll
            for (int k = 0; k < R; k += 2) ltri.put(col, k >> 1, v[k], v[k + 1]);
            return;
        }
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + toff;
#pragma unroll
        for (int k = 0; k < R; k += 2) dst[(size_t)(k >> 1) * HP] = v2f64{v[k], v[k + 1]};
    };
    auto gather = [&](auto lo_c, auto hi_c, uint32_t ecol, const LxAlleles<R>& al, double (&e)[R]) __attribute__((always_inline)) {
        static_for<decltype(lo_c)::value, decltype(hi_c)::value>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; e[k] = lx_emission<HP, k>(ecol, al); });
    };
    using I0 = std::integral_constant<int, 0>; using IH = std::integral_constant<int, R / 2>; using IR = std::integral_constant<int, R>;

    // DevContig::widef (phase 1 of triangle chains whose objects include wide ones, round 6): the emissions of a WIDE column do not
    // come from the record's 6 x 6 table (the LDS record's allele bytes are clamped table-row offsets) but from the column's side
    // table, indexed by the raw local alleles of the column-order record in memory — a rare column pays two dependent loads
    // (a kernel of its own, k_sweep_leanx_triw: with the branch in it the step of the chains WITHOUT wide columns ran 7 % slower)
    auto wide_fix = [&](uint32_t rel, int64_t c, double (&ev)[R]) __attribute__((always_inline)) {
        if constexpr (TRI && WIDE) {
            if (c < 0 || c >= (int64_t)C) return;
            const unsigned char* lrec = lx_rec<HP>(sh, rel);
            const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lrec[PG_REC_NLOCAL]);
            if (nl <= (uint32_t)PG_AMAX) return;
            const uint32_t woff = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t*)(lrec + PG_REC_WIDE_IDX));
            const GAS double* Ew = (const GAS double*)(const GAS unsigned char*)(dc.wide + (size_t)woff * 16u);
            const GAS unsigned char* al = (const GAS unsigned char*)dc.colrec + (size_t)c * Cfg::RB + PG_REC_ALLELES;
            uint32_t aj = al[j];
            aj = aj > nl ? nl : aj;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                uint32_t ai = al[i0 + (uint32_t)k];
                ai = ai > nl ? nl : ai;
                ev[k] = Ew[ai * (nl + 1u) + aj];
            }
        }
    };
    ColScalars bsc{&sh.scal[0][0]}, bsm{&sh.scal[1][0]};
    double w[R], e[R], Sy;
    {
        double y[R];
        if constexpr (PHASE == 1) {
            // column C-1: beta~ = 1 (hmm.cpp:356-358), stored at the backward bias
            const double B0 = ldexp(1.0, PG_BIAS_B);
#pragma unroll
            for (int k = 0; k < R; ++k) y[k] = (j < H && i0 + (uint32_t)k < H) ? B0 : 0.0;
            Sy = (double)H * (double)H * B0;
            store_col(top, y);
            if (tid == 0) { bscale[top] = 1.0; bsum[top] = Sy; }
        } else {
            gcdouble2* src = (gcdouble2*)resume + toff;
#pragma unroll
            for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; y[k] = t.x; y[k + 1] = t.y; }
            Sy = bsum[top + 1];
            if (!(Sy > 0.0)) {  // resuming behind an all-zero column: uniform (hmm.cpp:374-380)
#pragma unroll
                for (int k = 0; k < R; ++k) y[k] = (j < H && i0 + (uint32_t)k < H) ? unif : 0.0;
                Sy = 1.0;
            }
        }
        const LxAlleles<R> a0 = lx_alleles<HP>(sh, 0, j, i0);   // column t0+1: its emission goes into the first w
        gather(I0{}, IR{}, lx_ecol<HP>(sh, 0, a0.col8), a0, e);
        wide_fix(0, t0 + 1, e);
        double part = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) { w[k] = y[k] * e[k]; part += w[k]; }
        sh.psum[(uint32_t)t0 & 1u][rg][j] = part;
    }
    LxConsts cur = lx_consts<HP>(sh, 0);             // constants of the gap t0 -> t0+1
    {
        const LxAlleles<R> a1 = lx_alleles<HP>(sh, 1, j, i0);   // emissions of column t0 (the first step's w); every step fetches the next
        gather(I0{}, IR{}, lx_ecol<HP>(sh, 1, a1.col8), a1, e);
        wide_fix(1, t0, e);
    }
    double one = 1.0;   // (in a register for the whole sweep: the DPP form of v_fmac_f64 takes no constant)
    asm volatile("" : "+v"(one));
    // One column step: beta'_t from w = e_{t+1} beta'_{t+1} (see lean_backward); `cur` = constants of the gap t -> t+1
    // (record t+1, rel n), `al` = alleles of column t (rel n + 1).
    auto step = [&](int64_t t) __attribute__((always_inline)) {
        const uint32_t n = (uint32_t)(t0 - t);
        if (((n + 4u) % (uint32_t)BLK) == 0u) {
            const uint32_t blk = (n + 4u) / (uint32_t)BLK;
#pragma unroll
            for (int p = 0; p < PPT; ++p) { recs.park(sh, blk, p, piece[p]); piece[p] = recs.fetch(blk + 1u, p); }
        } else if (((n + 3u) % (uint32_t)BLK) == 0u) {
            lx_transform<HP>(sh, (n + 3u) / (uint32_t)BLK, tid);
        }
        int es = exponent_of(Sy) - PG_BIAS_B;
        es = es < -900 ? -900 : es;
        const double m = ldexp(Sy, -es - PG_BIAS_B);
        if (wave == 1) { asm volatile("" ::: "memory"); bsc.put(lane, (uint64_t)t, m); }
        const double k0 = ldexp(cur.c0, -es), k1 = ldexp(cur.c1, -es), k2 = ldexp(cur.c2, -es), kap = ldexp(cur.kappa, -es);
        lds_barrier();
        const uint32_t pb = (uint32_t)t & 1u;
        double pc[4], po[4], pr[NS][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pc[q] = sh.psum[pb][q][j];
            po[q] = Cfg::NCH == 2 ? sh.psum[pb][q][j ^ 64u] : 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) pr[s][q] = sh.psum[pb][q][i0 + 16u * (uint32_t)s + (lane & 15u)];
        }
        lean_fence();
        const double Cj = (pc[0] + pc[1]) + (pc[2] + pc[3]);
        const double Call = Cfg::NCH == 2 ? Cj + ((po[0] + po[1]) + (po[2] + po[3])) : Cj;
        const v4f64 zz = {0.0, 0.0, 0.0, 0.0};
        const v4f64 ma = __builtin_amdgcn_mfma_f64_16x16x4f64(Call, 1.0, zz, 0, 0, 0);
        const double ucol = k1 * Cj;
        double urep[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) urep[s] = dpp_source(k1 * ((pr[s][0] + pr[s][1]) + (pr[s][2] + pr[s][3])));
        const LxConsts cnx = lx_consts<HP>(sh, n + 1u);  // next step: gap t-1 -> t = record t
        const LxAlleles<R> anx = lx_alleles<HP>(sh, n + 2u, j, i0);   // and the emission of column t-1
        lean_fence();
        const double msum = (ma[0] + ma[1]) + (ma[2] + ma[3]);
        const v4f64 mb = __builtin_amdgcn_mfma_f64_16x16x4f64(msum, 1.0, zz, 0, 0, 0);
        const uint32_t ecoln = lx_ecol<HP>(sh, n + 2u, anx.col8);
        asm volatile("" :: "v"(mb));
        lean_fence();
        const double Sw = mb[0];
        const double uj = fma(k2, Sw, ucol);
        const double Snew = kap * Sw;  // = sum(beta'_t)
        Sy = Snew;                     // (1 behind an all-zero column, below)
        double part = 0.0, yprev = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double yk = fmac_row_bcast<(k & 15)>(fma(k0, w[k], uj), urep[k >> 4], one);  // beta'_t = k0 w + u_j + u_i
            part = fma(e[k], yk, part);
            w[k] = e[k] * yk;
            pin_here(w[k]);
            if (!(kLxExp & 2)) e[k] = lx_emission<HP, k>(ecoln, anx);   // e_{t-1}(i0 + k, j) for the next step
            if constexpr (k & 1) {
                if constexpr (TRI) ltri.put((GAS char*)(wr + (size_t)t * colsz), k >> 1, yprev, yk);
                else if (!(kLxExp & 1)) *(gdouble2*)(sp[k >> 3] + (((k >> 1) & 3) - 2) * (HP * 16)) = v2f64{yprev, yk};
                __builtin_amdgcn_sched_barrier(0);
            }
            else yprev = yk;
        });
        if (__builtin_expect(!(Snew > 0.0), 0)) {
            // beta~_t is all zero (every y_k above IS 0, and so is what was stored): its own posteriors are 0, the next
            // step starts from the uniform column (hmm.cpp:374-380); phantom paths have zero emission
            // (e[] holds the next column's emissions by now: this column's are fetched again)
            const LxAlleles<R> at = lx_alleles<HP>(sh, n + 1u, j, i0);
            double et[R];
            gather(I0{}, IR{}, lx_ecol<HP>(sh, n + 1u, at.col8), at, et);
            wide_fix(n + 1u, t, et);
            part = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) { w[k] = unif * et[k]; part += w[k]; }
            Sy = 1.0;
        }
        wide_fix(n + 2u, t - 1, e);
        sh.psum[(uint32_t)(t - 1) & 1u][rg][j] = part;
        if (wave == 2) { asm volatile("" ::: "memory"); bsm.put(lane, (uint64_t)t, Snew); }
        if (((uint64_t)t & 63u) == 0u) {
            if (wave == 1) bsc.flush(bscale, lane, (uint64_t)t);
            if (wave == 2) bsm.flush(bsum, lane, (uint64_t)t);
        }
        cur = cnx;
#pragma unroll
        for (int g = 0; g < R / 8; ++g) { sp[g] -= colsz * 8u; asm("" : "+v"(sp[g])); }
    };
#pragma unroll
    for (int g = 0; g < R / 8; ++g) sp[g] = (GAS char*)(wr + (size_t)t0 * colsz) + toff * 16u + (size_t)(4 * g + 2) * (size_t)(HP * 16);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int64_t t = t0; t >= bot; --t) step(t);
    if (wave == 1 && bsc.valid) bsc.flush(bscale, lane, (uint64_t)bot);
    if (wave == 2 && bsm.valid) bsm.flush(bsum, lane, (uint64_t)bot);
}

// ------------------------------------------------------------------------------------------
//  k_sweep_leanx2 (round 6): PHASE 2 of the 64-path triangle chains of fused jobs WITH multiallelic objects (DevContig::leanx2) —
//  k_sweep_lean2's design (no loader waves, no LDS ring: every thread fetches its own units of the compact partner column two
//  columns ahead into registers, three buffers rotating through a three-step loop body; the four waves' posterior partials
//  added up through LDS on the step's barrier before they leave) on the lean-x step (emissions from the column's 6 x 6 table in
//  LDS, records in blocks of sixteen).  The products P' beta' are added by ROW allele with exact 0 / 1 multipliers that are
//  scalar operands (a row's allele is the same in every lane): two accumulators for a column with at most two local alleles
//  (four columns in five), five otherwise — a uniform branch around the state loop.  Out: 64 entries per column and slot pair
//  (DevContig::T = 64), what k_bins reads — 3 KB of a multiallelic column instead of the 12 KB of per-thread partials the general
//  kernel's triangle ring wrote (21 GB per step on cohort_h64m, read back by k_bins).
//  Until round 6 these chains ran phase 2 on the general kernel: one workgroup per CU (the ring takes the LDS), 0.45 of the HBM peak.
// ------------------------------------------------------------------------------------------
#define PG_LX2_PAIRS ((PG_AMAX + 1) / 2)
#ifndef PG_LX2_TWO
#define PG_LX2_TWO 1   // columns with at most two local alleles add into two accumulators (0: five for every column)
#endif
#ifndef PG_LX2_EW
#define PG_LX2_EW 4   // emissions of a column in flight (LDS reads issued this many states ahead of their use)
#endif
struct LxShared2 {
    LxShared<64> a;
    v2f64 pp[2][4][PG_LX2_PAIRS][64];   // posterior partials {row allele 2q, 2q + 1} per wave and lane, by column parity
    // the row-allele weights: row a (at the byte offset a * 48 — a row's table-row offset IS its address here) has 1.0 at a; row
    // PG_AMAX (a phantom path) is zeros.  A state's weights are one SDWA add + one to three broadcast LDS reads (k_sweep_small16x's
    // scheme) instead of a compare and a select per allele on the scalar unit — the A/B build -DPG_LX2_ONEHOT=1; the product takes the
    // scalar selects
    double onehot[PG_ESTRIDE][PG_ESTRIDE] __attribute__((aligned(16)));
};
#ifndef PG_LX2_ONEHOT
#define PG_LX2_ONEHOT 0   // (measured on cohort_h64m, tools/r06_runs/gpu44.sh: phase 2 17.0 ms with the scalar selects, 17.7 with the one-hot reads)
#endif
DEVI void lx2_init_onehot(LxShared2& sh, uint32_t tid) {
    if (tid < (uint32_t)(PG_ESTRIDE * PG_ESTRIDE)) sh.onehot[tid / (uint32_t)PG_ESTRIDE][tid % (uint32_t)PG_ESTRIDE] = (tid / (uint32_t)PG_ESTRIDE == tid % (uint32_t)PG_ESTRIDE && tid / (uint32_t)PG_ESTRIDE < (uint32_t)PG_AMAX) ? 1.0 : 0.0;
}
// the same through the one-hot rows: K = the state's row inside the wave, rows = the wave's sixteen row offsets, oh = LDS address of the table
template <int NA, int K>
DEVI void lx2_add_oh(double (&acc)[2 * PG_LX2_PAIRS], double pr, const uint32_t (&rows)[4], uint32_t oh) {
    const uint32_t wa = add_byte<(K & 3)>(rows[K >> 2], oh);
    const v2f64 w01 = *(LAS const v2f64*)(uintptr_t)wa;
    acc[0] = fma(pr, w01.x, acc[0]); acc[1] = fma(pr, w01.y, acc[1]);
    if constexpr (NA > 2) {
        const v2f64 w23 = *(LAS const v2f64*)(uintptr_t)(wa + 16u);
        const double w4 = *(LAS const double*)(uintptr_t)(wa + 32u);
        acc[2] = fma(pr, w23.x, acc[2]); acc[3] = fma(pr, w23.y, acc[3]);
        acc[4] = fma(pr, w4, acc[4]);
    }
}
// (behind the barrier that follows the step of column c) wave q: slot pair q of column c, the four waves' partials added, 64 entries out
DEVI void lx2_flush(const LxShared2& sh, uint32_t pb, gdouble* part, uint32_t part_slots, size_t c, uint32_t nl, uint32_t wave, uint32_t lane) {
    if (wave >= (uint32_t)PG_LX2_PAIRS || 2u * wave >= nl) return;
    v2f64 t = sh.pp[pb][0][wave][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) { const v2f64 o = sh.pp[pb][w][wave][lane]; t.x += o.x; t.y += o.y; }
    ((gdouble2*)part)[(c * (size_t)(part_slots >> 1) + wave) * 64u + lane] = t;
}
// what a step needs of its OWN column beyond the emissions: the sixteen row offsets of the wave's rows (a * 48, uniform), the
// lane's table column, the number of local alleles — read from the LDS record a step ahead and carried
struct Lx2Col { uint32_t rows[4]; uint32_t ecol; uint32_t nl; };
DEVI Lx2Col lx2_col(const LxShared<64>& sh, uint32_t rel, uint32_t lane, uint32_t i0) {
    const LxAlleles<16> a = lx_alleles<64>(sh, rel, lane, i0);
    Lx2Col c;
#pragma unroll
    for (int q = 0; q < 4; ++q) c.rows[q] = a.rows[q];
    c.ecol = lx_ecol<64>(sh, rel, a.col8);
    c.nl = lx_rec<64>(sh, rel)[PG_REC_NLOCAL];
    return c;
}
template <int K>
DEVI double lx2_emission(const Lx2Col& c) { return *(LAS const double*)(uintptr_t)add_byte<(K & 3)>(c.rows[K >> 2], c.ecol); }
// one state's product into the accumulators of its row's allele (ro = the row's table-row offset a * 48; 5 * 48 = a phantom path: none)
template <int NA>
DEVI void lx2_add(double (&acc)[2 * PG_LX2_PAIRS], double pr, uint32_t ro /*uniform*/) {
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a] = fma(pr, ro == (uint32_t)(a * PG_ESTRIDE * 8) ? 1.0 : 0.0, acc[a]);
}

DEVI void leanx2_forward(const DevContig& dc, LxShared2& sh2, uint32_t C, uint32_t oh /* LDS address of sh2.onehot */) {
    constexpr int R = 16, BLK = LxCfg<64>::BLK, PPT = LxCfg<64>::PPT;
    LxShared<64>& sh = sh2.a;
    const uint32_t mid = C / 2, lo = mid, hi = C, first = lo == 0 ? 1u : lo;   // (C == 1: lo = 0, the one column is the prologue's)
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t i0 = wave * R, H = dc.H;
    const size_t colsz = dc.col_stride;   // compact triangles
    const double unif = 1.0 / ((double)H * (double)H);
    LxRecs<64> recs{(const GAS char*)dc.colrec, (int64_t)first - 1, (int64_t)C, +1, tid};
    v2f64 piece[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) recs.park(sh, 0, p, recs.fetch(0, p));
#pragma unroll
    for (int p = 0; p < PPT; ++p) piece[p] = recs.fetch(1, p);
    lx2_init_onehot(sh2, tid);
    lds_barrier();
    lx_transform<64>(sh, 0, tid);
    lds_barrier();
    gcdouble* cols = (gcdouble*)dc.fwd;
    gdouble* fscale = (gdouble*)dc.fscale;
    gu8* fallback = (gu8*)dc.fwd_fallback;
    gdouble* part = (gdouble*)dc.part;
    const uint32_t part_slots = dc.part_slots;
    LeanTri<R> tri;
    tri.setup(i0, lane);
    double b0[R], b1[R], b2[R];   // partner columns beta'_t, two ahead of their use
#pragma unroll
    for (int k = 0; k < R; ++k) { b0[k] = 0.0; b1[k] = 0.0; b2[k] = 0.0; }
    tri.load(cols + (size_t)lo * colsz, b0);
    if (lo + 1 < C) tri.load(cols + (size_t)(lo + 1) * colsz, b1);

    ColScalars fsc{&sh.scal[0][0]};
    double x[R];
    uint32_t pnl = 0;   // local alleles of the column whose partials are pending
    if (lo == 0) {
        // C == 1: column 0 IS the chain: v_0 = e_0 2^BIAS_F, its partner the backward role's beta'_0 (phase 1); no step follows
        const Lx2Col c0 = lx2_col(sh, 0, lane, i0);
        const double P0 = ldexp(1.0, PG_BIAS_F);
        double acc[2 * PG_LX2_PAIRS];
#pragma unroll
        for (int a = 0; a < 2 * PG_LX2_PAIRS; ++a) acc[a] = 0.0;
        uint32_t rw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rw[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.rows[q]);
        double part0 = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            part0 += lx2_emission<k>(c0) * P0;
            lx2_add<PG_AMAX>(acc, P0 * b0[k], (rw[k >> 2] >> (8 * (k & 3))) & 0xFFu);
        });
        sh.psum[0][wave][lane] = part0;
#pragma unroll
        for (int q = 0; q < PG_LX2_PAIRS; ++q) sh2.pp[0][wave][q][lane] = v2f64{acc[2 * q], acc[2 * q + 1]};
        if (tid == 0) fscale[0] = 1.0;
        lds_barrier();
        lx2_flush(sh2, 0, part, part_slots, 0, (uint32_t)__builtin_amdgcn_readfirstlane((int)c0.nl), wave, lane);
        const double Cj = (sh.psum[0][0][lane] + sh.psum[0][1][lane]) + (sh.psum[0][2][lane] + sh.psum[0][3][lane]);
        if (!(wave_total_mfma(Cj) > 0.0) && tid == 0) fallback[0] = 1;
        return;
    }
    {
        const Lx2Col cp = lx2_col(sh, 0, lane, i0);   // column lo - 1: its emission makes x of the stored P'
        lean_load_mirrored<R>(cols + (size_t)(lo - 1) * colsz, i0, lane, x);   // P'_{lo-1}, stored by phase 1 of this role
        const bool was_uniform = fallback[lo - 1] != 0;
        double part0 = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if (!was_uniform) x[k] *= lx2_emission<k>(cp);
            part0 += x[k];
        });
        sh.psum[(first - 1) & 1u][wave][lane] = part0;
    }
    LxConsts cur = lx_consts<64>(sh, 1);      // column `first`: constants of the gap first-1 -> first
    Lx2Col col = lx2_col(sh, 1, lane, i0);    // ... its alleles, table column, allele count
    // One column step (lean2_forward's, with table emissions): `ba` = the partner column beta'_t, `bc` takes beta'_{t+2}
    auto step = [&](uint32_t t, const double (&ba)[R], double (&bc)[R]) __attribute__((always_inline)) {
        const uint32_t n = t - first;   // column t is the record with rel = n + 1
        if (((n + 4u) % (uint32_t)BLK) == 0u) {
            const uint32_t blk = (n + 4u) / (uint32_t)BLK;
#pragma unroll
            for (int p = 0; p < PPT; ++p) { recs.park(sh, blk, p, piece[p]); piece[p] = recs.fetch(blk + 1u, p); }
        } else if (((n + 3u) % (uint32_t)BLK) == 0u) {
            lx_transform<64>(sh, (n + 3u) / (uint32_t)BLK, tid);   // the block parked a step ago (a barrier lies between)
        }
        if (t + 2 < C) tri.load(cols + (size_t)(t + 2) * colsz, bc);   // two columns ahead (bc was last read a step ago)
        if (t > first) lx2_flush(sh2, (t - 1) & 1u, part, part_slots, (size_t)(t - 1), pnl, wave, lane);
        const uint32_t pb = (t - 1) & 1u;
        const double Cj = (sh.psum[pb][0][lane] + sh.psum[pb][1][lane]) + (sh.psum[pb][2][lane] + sh.psum[pb][3][lane]);
        const uint32_t ri = i0 + (lane & 15u);
        const double Cr = (sh.psum[pb][0][ri] + sh.psum[pb][1][ri]) + (sh.psum[pb][2][ri] + sh.psum[pb][3][ri]);
        const double ucol = cur.c1 * Cj;
        const double urep = dpp_source(cur.c1 * Cr);   // u_i of row i0 + (lane & 15): the DPP source
        const v4f64 zz = {0.0, 0.0, 0.0, 0.0};
        const v4f64 ma = __builtin_amdgcn_mfma_f64_16x16x4f64(Cj, 1.0, zz, 0, 0, 0);
        double S = __builtin_amdgcn_mfma_f64_16x16x4f64((ma[0] + ma[1]) + (ma[2] + ma[3]), 1.0, zz, 0, 0, 0)[0];
        double uj = fma(cur.c2, S, ucol);
        double c0 = cur.c0;
        if (__builtin_expect(!(S > 0.0), 0)) {
            // column t-1 summed to zero: the uniform column takes its place (hmm.cpp:253-267); its own partials were formed from
            // the all-zero column — k_bins re-forms those bins from the flag
            if (tid == 0) fallback[t - 1] = 1;
            const double Cu = (double)H * unif;
            S = 1.0;
            uj = fma(cur.c0, unif, fma(cur.c2, 1.0, 2.0 * cur.c1 * Cu));
            c0 = 0.0;
        }
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        const double m = ldexp(S, -es - PG_BIAS_F);
        const double sc = ldexp(1.0, -es), c0s = ldexp(c0, -es), ujs = ldexp(uj, -es);
        uint32_t rw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rw[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)col.rows[q]);
        const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)col.nl);
        double part0 = 0.0, acc[2 * PG_LX2_PAIRS];
#pragma unroll
        for (int a = 0; a < 2 * PG_LX2_PAIRS; ++a) acc[a] = 0.0;
        auto states = [&](auto na_c) __attribute__((always_inline)) {
            constexpr int NA = decltype(na_c)::value;
            double ew[PG_LX2_EW];   // emissions of the column, a few states ahead of their use
            static_for<0, PG_LX2_EW>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; ew[k] = lx2_emission<k>(col); });
            static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                const double e = ew[k % PG_LX2_EW];
                if constexpr (k + PG_LX2_EW < R) ew[k % PG_LX2_EW] = lx2_emission<k + PG_LX2_EW>(col);
                const double pk = fmac_row_bcast<k>(fma(c0s, x[k], ujs), urep, sc);   // P'_t = c0 x + u_j + u_i
                x[k] = pk * e;
                part0 += x[k];
                if constexpr (PG_LX2_ONEHOT) lx2_add_oh<NA, k>(acc, pk * ba[k], col.rows, oh);   // P'_t beta'_t (0 below the stored half)
                else lx2_add<NA>(acc, pk * ba[k], (rw[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                if constexpr ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            });
        };
        if (PG_LX2_TWO && nl <= 2u) states(std::integral_constant<int, 2>{});
        else states(std::integral_constant<int, PG_AMAX>{});
        sh.psum[t & 1u][wave][lane] = part0;
#pragma unroll
        for (int q = 0; q < PG_LX2_PAIRS; ++q)
            if ((uint32_t)(2 * q) < nl) sh2.pp[t & 1u][wave][q][lane] = v2f64{acc[2 * q], acc[2 * q + 1]};
        if (wave == 0) {
            fsc.put(lane, t, m);
            if ((t & 63u) == 63u) fsc.flush(fscale, lane, t);
        }
        pnl = nl;
        cur = lx_consts<64>(sh, n + 2u);          // the next column's constants, alleles, table column: read HERE, behind the state
        col = lx2_col(sh, n + 2u, lane, i0);      // loop (not a step ahead in front of it: fourteen registers less across the loop)
        lds_barrier();
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (see lean_forward)
    lds_barrier();
    {
        uint32_t t = first;
        for (; t + 2 < hi; t += 3) {
            step(t, b0, b2);
            step(t + 1, b1, b0);
            step(t + 2, b2, b1);
        }
        if (t < hi) step(t, b0, b2);
        if (t + 1 < hi) step(t + 1, b1, b0);
    }
    lx2_flush(sh2, (hi - 1) & 1u, part, part_slots, (size_t)(hi - 1), pnl, wave, lane);
    if (wave == 0 && fsc.valid) fsc.flush(fscale, lane, hi - 1);
    {   // the last column may itself have summed to zero
        const uint32_t pb = (hi - 1) & 1u;
        const double Cj = (sh.psum[pb][0][lane] + sh.psum[pb][1][lane]) + (sh.psum[pb][2][lane] + sh.psum[pb][3][lane]);
        if (!(wave_total_mfma(Cj) > 0.0) && tid == 0) fallback[hi - 1] = 1;
    }
}

DEVI void leanx2_backward(const DevContig& dc, LxShared2& sh2, uint32_t C, uint32_t oh) {
    constexpr int R = 16, BLK = LxCfg<64>::BLK, PPT = LxCfg<64>::PPT;
    LxShared<64>& sh = sh2.a;
    const int64_t mid = C / 2, top = mid - 1, bot = 0, t0 = top;
    if (top < 0) return;   // (C == 1: the forward role has the one column)
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t i0 = wave * R, H = dc.H;
    const size_t colsz = dc.col_stride;
    const double unif = 1.0 / ((double)H * (double)H);
    LxRecs<64> recs{(const GAS char*)dc.colrec, t0 + 1, (int64_t)C, -1, tid};   // rel r = column t0 + 1 - r
    v2f64 piece[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) recs.park(sh, 0, p, recs.fetch(0, p));
#pragma unroll
    for (int p = 0; p < PPT; ++p) piece[p] = recs.fetch(1, p);
    lx2_init_onehot(sh2, tid);
    lds_barrier();
    lx_transform<64>(sh, 0, tid);
    lds_barrier();
    gcdouble* cols = (gcdouble*)dc.fwd;
    gdouble* bscale = (gdouble*)dc.bscale;
    gcdouble* bsum = (gcdouble*)dc.bsum;
    gdouble* part = (gdouble*)dc.part;
    const uint32_t part_slots = dc.part_slots;
    LeanTri<R> tri;
    tri.setup(i0, lane);
    double b0[R], b1[R], b2[R];   // partner columns P'_t (forward, stored by phase 1), two ahead of their use
#pragma unroll
    for (int k = 0; k < R; ++k) { b0[k] = 0.0; b1[k] = 0.0; b2[k] = 0.0; }
    tri.load(cols + (size_t)t0 * colsz, b0);
    if (t0 - 1 >= 0) tri.load(cols + (size_t)(t0 - 1) * colsz, b1);

    ColScalars bsc{&sh.scal[0][0]};
    double w[R], Sy;
    {
        double y[R];
        lean_load_mirrored<R>(cols + (size_t)(top + 1) * colsz, i0, lane, y);   // beta'_{mid}, stored by phase 1 of this role
        Sy = bsum[top + 1];
        if (!(Sy > 0.0)) {   // resuming behind an all-zero column: uniform (hmm.cpp:374-380)
#pragma unroll
            for (int k = 0; k < R; ++k) y[k] = (lane < H && i0 + (uint32_t)k < H) ? unif : 0.0;
            Sy = 1.0;
        }
        const Lx2Col c1 = lx2_col(sh, 0, lane, i0);   // column t0 + 1: its emission goes into the first w
        double part0 = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            w[k] = y[k] * lx2_emission<k>(c1);
            part0 += w[k];
        });
        sh.psum[(uint32_t)t0 & 1u][wave][lane] = part0;
    }
    LxConsts cur = lx_consts<64>(sh, 0);      // constants of the gap t0 -> t0 + 1
    Lx2Col col = lx2_col(sh, 1, lane, i0);    // column t0: alleles, table column, allele count
    uint32_t pnl = 0;
    double one = 1.0;   // (in a register: the DPP form of v_fmac_f64 takes no constant)
    asm volatile("" : "+v"(one));
    auto step = [&](int64_t t, const double (&ba)[R], double (&bc)[R]) __attribute__((always_inline)) {
        const uint32_t n = (uint32_t)(t0 - t);   // column t is the record with rel = n + 1
        if (((n + 4u) % (uint32_t)BLK) == 0u) {
            const uint32_t blk = (n + 4u) / (uint32_t)BLK;
#pragma unroll
            for (int p = 0; p < PPT; ++p) { recs.park(sh, blk, p, piece[p]); piece[p] = recs.fetch(blk + 1u, p); }
        } else if (((n + 3u) % (uint32_t)BLK) == 0u) {
            lx_transform<64>(sh, (n + 3u) / (uint32_t)BLK, tid);
        }
        if (t - 2 >= 0) tri.load(cols + (size_t)(t - 2) * colsz, bc);
        int es = exponent_of(Sy) - PG_BIAS_B;
        es = es < -900 ? -900 : es;
        const double m = ldexp(Sy, -es - PG_BIAS_B);
        if (wave == 0) bsc.put(lane, (uint64_t)t, m);
        const double k0 = ldexp(cur.c0, -es), k1 = ldexp(cur.c1, -es), k2 = ldexp(cur.c2, -es), kap = ldexp(cur.kappa, -es);
        lds_barrier();
        if (t < t0) lx2_flush(sh2, (uint32_t)(t + 1) & 1u, part, part_slots, (size_t)(t + 1), pnl, wave, lane);
        const uint32_t pb = (uint32_t)t & 1u;
        const double Cj = (sh.psum[pb][0][lane] + sh.psum[pb][1][lane]) + (sh.psum[pb][2][lane] + sh.psum[pb][3][lane]);
        const uint32_t ri = i0 + (lane & 15u);
        const double Cr = (sh.psum[pb][0][ri] + sh.psum[pb][1][ri]) + (sh.psum[pb][2][ri] + sh.psum[pb][3][ri]);
        const double ucol = k1 * Cj;
        const double urep = dpp_source(k1 * Cr);
        const v4f64 zz = {0.0, 0.0, 0.0, 0.0};
        const v4f64 ma = __builtin_amdgcn_mfma_f64_16x16x4f64(Cj, 1.0, zz, 0, 0, 0);
        const double Sw = __builtin_amdgcn_mfma_f64_16x16x4f64((ma[0] + ma[1]) + (ma[2] + ma[3]), 1.0, zz, 0, 0, 0)[0];
        const double uj = fma(k2, Sw, ucol);
        const double Snew = kap * Sw;   // = sum(beta'_t)
        uint32_t rw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rw[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)col.rows[q]);
        const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)col.nl);
        double part0 = 0.0, acc[2 * PG_LX2_PAIRS];
#pragma unroll
        for (int a = 0; a < 2 * PG_LX2_PAIRS; ++a) acc[a] = 0.0;
        if (__builtin_expect(!(Snew > 0.0), 0)) {
            // beta~_t is all zero: its own posteriors are 0, the next step starts from the uniform column (phantom paths: emission 0)
            static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                w[k] = unif * lx2_emission<k>(col);
                part0 += w[k];
            });
        } else {
            auto states = [&](auto na_c) __attribute__((always_inline)) {
                constexpr int NA = decltype(na_c)::value;
                double ew[PG_LX2_EW];
                static_for<0, PG_LX2_EW>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; ew[k] = lx2_emission<k>(col); });
                static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    const double e = ew[k % PG_LX2_EW];
                    if constexpr (k + PG_LX2_EW < R) ew[k % PG_LX2_EW] = lx2_emission<k + PG_LX2_EW>(col);
                    const double yk = fmac_row_bcast<k>(fma(k0, w[k], uj), urep, one);   // beta'_t = k0 w + u_j + u_i
                    w[k] = yk * e;
                    part0 += w[k];
                    if constexpr (PG_LX2_ONEHOT) lx2_add_oh<NA, k>(acc, ba[k] * yk, col.rows, oh);   // P'_t beta'_t
                    else lx2_add<NA>(acc, ba[k] * yk, (rw[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                    if constexpr ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                });
            };
            if (PG_LX2_TWO && nl <= 2u) states(std::integral_constant<int, 2>{});
            else states(std::integral_constant<int, PG_AMAX>{});
        }
        sh.psum[(uint32_t)(t - 1) & 1u][wave][lane] = part0;
#pragma unroll
        for (int q = 0; q < PG_LX2_PAIRS; ++q)
            if ((uint32_t)(2 * q) < nl) sh2.pp[(uint32_t)t & 1u][wave][q][lane] = v2f64{acc[2 * q], acc[2 * q + 1]};
        if (wave == 0 && ((uint64_t)t & 63u) == 0u) bsc.flush(bscale, lane, (uint64_t)t);
        Sy = Snew > 0.0 ? Snew : 1.0;
        pnl = nl;
        cur = lx_consts<64>(sh, n + 1u);          // next step: the gap t-1 -> t = record t
        col = lx2_col(sh, n + 2u, lane, i0);      // ... and column t-1 (behind the state loop: see leanx2_forward)
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);
    {
        int64_t t = t0;
        for (; t - 2 >= bot; t -= 3) {
            step(t, b0, b2);
            step(t - 1, b1, b0);
            step(t - 2, b2, b1);
        }
        if (t >= bot) step(t, b0, b2);
        if (t - 1 >= bot) step(t - 1, b1, b0);
    }
    lds_barrier();
    lx2_flush(sh2, (uint32_t)bot & 1u, part, part_slots, (size_t)bot, pnl, wave, lane);
    if (wave == 0 && bsc.valid) bsc.flush(bscale, lane, (uint64_t)bot);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))   // two workgroups per CU: <= 256 registers per lane
void k_sweep_leanx2(const DevContig* __restrict__ contigs) {
    __shared__ LxShared2 sh;
    const DevContig& dc = contigs[blockIdx.x];
    if (!dc.leanx2 || dc.HP != 64u || dc.tri != 1u) return;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    // (LDS address of the one-hot table, taken relative to the record blocks': the direct cast of &sh.onehot trips this compiler's
    //  machine verifier — "V_CMP_NE_U32 0, $src_shared_base: operand has incorrect register class")
    const uint32_t oh = (uint32_t)(uintptr_t)(LAS const unsigned char*)((const unsigned char*)&sh.a.rec[0][0][0]) + (uint32_t)(offsetof(LxShared2, onehot) - offsetof(LxShared2, a) - offsetof(LxShared<64>, rec));
    if (blockIdx.y == 0) leanx2_forward(dc, sh, C, oh);
    else leanx2_backward(dc, sh, C, oh);
}

// phase 1 of the 64-path chains of fused jobs with multiallelic objects (DevContig::tri == 1, leanx == 2): the lean-x step with
// triangle stores (phase 2: the general kernel's triangle ring)
__global__ __launch_bounds__((LxCfg<64>::T)) void k_sweep_leanx_tri(const DevContig* __restrict__ contigs) {
    __shared__ LxShared<64> sh;
    const DevContig& dc = contigs[blockIdx.x];
    if (dc.leanx != 2u || dc.HP != 64u || !dc.tri || dc.widef) return;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    if (blockIdx.y == 0) leanx_forward<1, 64, true>(dc, sh, C, 0);
    else leanx_backward<1, 64, true>(dc, sh, C, 0);
}
// ... of such chains whose objects include WIDE ones (DevContig::widef): a wide column's emissions from its side table (wide_fix)
__global__ __launch_bounds__((LxCfg<64>::T)) void k_sweep_leanx_triw(const DevContig* __restrict__ contigs) {
    __shared__ LxShared<64> sh;
    const DevContig& dc = contigs[blockIdx.x];
    if (dc.leanx != 2u || dc.HP != 64u || !dc.tri || !dc.widef) return;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    if (blockIdx.y == 0) leanx_forward<1, 64, true, true>(dc, sh, C, 0);
    else leanx_backward<1, 64, true, true>(dc, sh, C, 0);
}
template <int PHASE, int HP>
__global__ __launch_bounds__((LxCfg<HP>::T)) void k_sweep_leanx(const DevContig* __restrict__ contigs, uint32_t chunk) {
    __shared__ LxShared<HP> sh;
    const DevContig& dc = contigs[blockIdx.x];
    if (dc.leanx != 1u || dc.HP != (uint32_t)HP) return;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    const unsigned long long t_begin = kChainProf ? __builtin_amdgcn_s_memtime() : 0ull;
    if (blockIdx.y == 0) leanx_forward<PHASE, HP>(dc, sh, C, chunk);
    else leanx_backward<PHASE, HP>(dc, sh, C, chunk);
    if (kChainProf && threadIdx.x == 0) {
        unsigned long long* o = dc.prof + (blockIdx.y == 0 ? 0 : 16) + (PHASE == 1 ? 0 : 8);
        o[0] = __builtin_amdgcn_s_memtime() - t_begin;
    }
}

// ------------------------------------------------------------------------------------------
//  k_sweep_small16 : all-biallelic chains with H = HP = 16 — the store-only phases (1, 3) and, for fused jobs (DevContig::small == 2),
//  phase 2 with the posterior's four class sums per column formed inside the row — BASELINE configs[1], and the
//  shape of many-sample runs over small (sampled) panels.  A 16 x 16 column is 256 states: FOUR half-chains share one
//  wave, each in its own 16-lane DPP row.  lane -> (half-chain r = lane >> 4, column j = lane & 15); the sixteen rows
//  of the column live in the lane's registers.  Everything a column step exchanges stays inside the DPP row:
//    * column sums C_j are in-lane sums (all rows of a column are in one lane);
//    * the row terms u_i = c1 C_i are the SAME numbers (the column is symmetric): lane i of the row holds u_i, and
//      v_fmac_f64_dpp row_newbcast:i adds it in the lanes of the row — no LDS, no barrier, no MFMA;
//    * the total S is four rotate-and-add steps inside the row (every lane of a row ends with the same bits).
//  Nothing is wave-uniform (four different chains): column records are read per lane (three in flight: the loop runs
//  three steps per iteration with the record variables in rotated roles), row-allele selects are per-lane bit-field
//  inserts, the trip count is the longest of the four half-chains and finished rows run on masked out.
//  Same stored columns, scales, fall-back rules and resume conventions as lean_forward / lean_backward (k_post and the
//  general phase-2 kernel read what this kernel writes).  One wave per workgroup; grid = (ceil(chains / 4), 2 roles).
// ------------------------------------------------------------------------------------------
DEVI double row16_sum(double v) {   // sum over the 16 lanes of the DPP row, in every lane of the row (bitwise the same)
    v += dpp_f64<0x128, 0xF, true>(v);   // row_ror:8
    v += dpp_f64<0x124, 0xF, true>(v);   // row_ror:4
    v += dpp_f64<0x122, 0xF, true>(v);   // row_ror:2
    v += dpp_f64<0x121, 0xF, true>(v);   // row_ror:1
    return v;
}
DEVI FRec load_frec(gcdouble* frec, int64_t c, int64_t C) {   // record of column c (clamped: records past the end are never used)
    c = c < 0 ? 0 : (c >= C ? C - 1 : c);
    gcdouble2* q = (gcdouble2*)(frec + (size_t)c * 8);
    const v2f64 a = q[0], b = q[1], e = q[2], f = q[3];
    FRec r;
    r.c0 = a.x; r.c1 = a.y; r.c2 = b.x; r.kappa = b.y; r.E00 = e.x; r.E01 = e.y; r.E11 = f.x;
    r.bits1 = (unsigned long long)__double_as_longlong(f.y);
    return r;
}
// bit K of the lane's row bits as 0 / ~0 (v_bfe_i32), the select through it — e(i, j) of row k: the bit picks between the two values of the
// lane's column, one bit-field insert per register half — and (phase 2) the bit as the doubles (1 - bit, bit): exact 0 / 1 multipliers
// for the posterior sums by row allele
template <int K>
DEVI uint32_t row_bit_mask(uint32_t rbits) { uint32_t m; asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(rbits), "n"(K)); return m; }
DEVI double sel_by_mask(uint32_t m, double if0, double if1) {
    uint32_t lo, hi;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(m), "v"(__double2loint(if1)), "v"(__double2loint(if0)));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(m), "v"(__double2hiint(if1)), "v"(__double2hiint(if0)));
    return __hiloint2double((int)hi, (int)lo);
}
struct SmallCtx {      // per lane: the half-chain of its DPP row
    bool live;         // the row has a half-chain with columns to do in this launch
    gcdouble* frec; gdouble* wr; gcdouble* resume; gdouble* sc_a; gdouble* sc_b; gu8* fallback;
    gcdouble* partner; gdouble* part;   // phase 2: the columns the other role stored, the class sums of the posterior (DevContig::cls4 layout)
    int64_t C, lo, hi;   // forward: columns [lo, hi) ascending; backward: [bot = lo, top = hi] descending
};

template <int PHASE>
DEVI void small16_forward(const DevContig* contigs, const uint32_t* ids, uint32_t n_ids, uint32_t chunk, double* dump) {
    constexpr int HP = 16, R = 16;
    // records in flight: five in the store-only phases — a wave's loads and stores share one in-order counter, so waiting for
    // a record caps the stores the wave may have in flight at the number issued since (pg_small16x.h; profiles/r05_small16x_ablation.txt);
    // five: with six the kernel needs more than 256 registers, one wave per SIMD instead of two
    constexpr int D = PHASE == 2 ? 3 : 5;
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u;
    const uint32_t slot = blockIdx.x * 4u + (lane >> 4);
    const size_t colsz = (size_t)HP * HP;
    const double unif = 1.0 / 256.0;
    SmallCtx cx{};
    uint32_t first = 1;
    if (slot < n_ids) {
        const DevContig& dc = contigs[ids[slot]];
        const uint32_t C = *dc.n_cols, mid = C / 2, K = dc.chunk_cols;
        uint32_t lo = PHASE == 1 ? 0u : mid, hi = PHASE == 1 ? mid : C;
        bool ok = C > 0;
        if constexpr (PHASE == 2) ok = dc.small == 2u && C >= 2u;   // (a chain left with a single column: the general kernel, like k_sweep_lean2's)
        if constexpr (PHASE == 3) {
            const unsigned long long l = (unsigned long long)mid + (unsigned long long)chunk * K;
            ok = ok && l < C;
            lo = ok ? (uint32_t)l : 0u;
            hi = ok ? (C - lo > K ? lo + K : C) : 0u;
        }
        ok = ok && lo < hi;
        if (ok) {
            cx.live = true; cx.C = C; cx.lo = lo; cx.hi = hi;
            cx.frec = (gcdouble*)dc.frec; cx.sc_a = (gdouble*)dc.fscale; cx.fallback = (gu8*)dc.fwd_fallback;
            gdouble* fwd = (gdouble*)dc.fwd;
            cx.wr = fwd; cx.partner = (gcdouble*)fwd; cx.part = (gdouble*)dc.part;
            cx.resume = (gcdouble*)(fwd + (size_t)(lo > 0 ? lo - 1 : 0) * colsz);
            if constexpr (PHASE == 3) {
                gdouble* scr = (gdouble*)dc.scratch;
                cx.wr = scr + (size_t)(PG_SCR_BUF(chunk) * 2u) * K * colsz - (size_t)lo * colsz;
                if (chunk > 0) cx.resume = (gcdouble*)(scr + ((size_t)(PG_SCR_BUF(chunk - 1u) * 2u) * K + (K - 1u)) * colsz);
            }
            first = lo == 0 ? 1u : lo;
        }
    }
    const int n_steps = __builtin_amdgcn_readfirstlane(wave_max_i32(cx.live ? (int)(cx.hi - (int64_t)first) : 0));   // (uniform) the longest of the four
    if (__builtin_amdgcn_readfirstlane(wave_max_i32(cx.live ? 1 : 0)) == 0) return;
    auto emis = [&](const FRec& r, double& eA, double& eB) {
        const bool aj = (r.bits1 >> j) & 1ull;
        eA = aj ? r.E01 : r.E00;
        eB = aj ? r.E11 : r.E01;
    };
    auto store_col = [&](int64_t c, const double (&v)[R]) {
        gdouble2* dst = (gdouble2*)(cx.wr + (size_t)c * colsz) + j;
#pragma unroll
        for (int k = 0; k < R; k += 2) dst[(size_t)(k >> 1) * HP] = v2f64{v[k], v[k + 1]};
    };
    // phase 2: the partner column beta'_t of this lane's column, fetched three steps ahead (clamped: a column past the end is never used)
    auto load_partner = [&](int64_t c, double (&v)[R]) {
        if (!cx.live) return;
        c = c < cx.lo ? cx.lo : (c >= cx.hi ? cx.hi - 1 : c);
        gcdouble2* src = (gcdouble2*)(cx.partner + (size_t)c * colsz) + j;
#pragma unroll
        for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; v[k] = t.x; v[k + 1] = t.y; }
    };
    auto flag_uniform = [&](int64_t cprev) {   // (lanes of rows whose column cprev summed to zero)
        if (PHASE != 2 && cprev >= cx.lo) {
            double xu[R];
#pragma unroll
            for (int k = 0; k < R; ++k) xu[k] = unif;
            store_col(cprev, xu);
        }
        if (j == 0) cx.fallback[cprev] = 1;
    };
    // the column before the first step: x = e (.) P'
    double x[R], buf = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) x[k] = 0.0;
    if (cx.live) {
        const FRec r0 = load_frec(cx.frec, (int64_t)first - 1, cx.C);
        double eA, eB;
        emis(r0, eA, eB);
        const uint32_t rb = (uint32_t)(r0.bits1 & 0xFFFFull);
        if (cx.lo == 0) {
            const double P0 = ldexp(1.0, PG_BIAS_F);
            double pz[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { pz[k] = P0; x[k] = (((rb >> k) & 1u) ? eB : eA) * P0; }
            store_col(0, pz);
            if (j == 0) cx.sc_a[0] = 1.0;
        } else {
            gcdouble2* src = (gcdouble2*)cx.resume + j;
#pragma unroll
            for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; x[k] = t.x; x[k + 1] = t.y; }
            if (!cx.fallback[cx.lo - 1]) {
#pragma unroll
                for (int k = 0; k < R; ++k) x[k] *= ((rb >> k) & 1u) ? eB : eA;
            }
        }
    }
    double ee[R], pp[R];   // x = ee pp (formed at the start of the next step, see lean_forward)
#pragma unroll
    for (int k = 0; k < R; ++k) { ee[k] = 1.0; pp[k] = x[k]; }
    // One column step of the (up to) four half-chains; `cur` = record of column first + n, `far` takes the record three
    // steps on.  Rows whose half-chain is done (or absent) compute on whatever their registers hold and store nothing.
    auto step = [&](int n, FRec& slot_rec, double (&vp)[PHASE == 2 ? R : 1]) __attribute__((always_inline)) {
        const int64_t t = (int64_t)first + n;
        const bool act = cx.live && t < cx.hi;
        const FRec cur = slot_rec;                                   // (its fields move on; the variable takes the far record)
        if (cx.live) slot_rec = load_frec(cx.frec, t + D, cx.C);
        double Cj = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; x[k] = ee[k] * pp[k]; Cj += x[k]; });
        double S = row16_sum(Cj);
        // a row that did its last column a step ago: that column may itself have summed to zero (rows of longer
        // half-chains keep the wave going; a finished row computes on, its registers no longer mean anything)
        if (cx.live && t == cx.hi && !(S > 0.0)) flag_uniform(t - 1);
        const double ucol = dpp_source(cur.c1 * Cj);   // u_i of row i = this lane's column (the column is symmetric)
        double uj = fma(cur.c2, S, ucol);
        double c0 = cur.c0;
        if (act && !(S > 0.0)) {
            // column t-1 summed to zero: the uniform column takes its place (hmm.cpp:253-267), see lean_forward
            flag_uniform(t - 1);
            const double Cu = 16.0 * unif;
            S = 1.0;
            uj = fma(cur.c0, unif, fma(cur.c2, 1.0, 2.0 * cur.c1 * Cu));
            c0 = 0.0;
        }
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        const double m = ldexp(S, -es - PG_BIAS_F);
        const double sc = ldexp(1.0, -es), c0s = ldexp(c0, -es), ujs = ldexp(uj, -es);
        double eA, eB;
        emis(cur, eA, eB);
        const uint32_t rb = (uint32_t)(cur.bits1 & 0xFFFFull);
        // (rows that are done keep computing: their stores go to a scrap column instead of under a mask)
        gdouble2* dst = act ? (gdouble2*)(cx.wr + (size_t)t * colsz) + j : (gdouble2*)dump + lane;
        double pprev = 0.0, acc0 = 0.0, acc1 = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double pk = fmac_row_bcast<k>(fma(c0s, x[k], ujs), ucol, sc);   // P'_t(k, j) 2^-es = c0 x + u_j + u_k
            const uint32_t mk = row_bit_mask<k>(rb);
            ee[k] = sel_by_mask(mk, eA, eB);
            pp[k] = pk;
            if constexpr (PHASE == 2) {
                // posterior: P'_t beta'_t added by row allele (exact 0 / 1 multipliers, as in the general kernel)
                const double pr = vp[k] * pk;
                acc1 = fma(pr, __hiloint2double((int)(mk & 0x3FF00000u), 0), acc1);
                acc0 = fma(pr, __hiloint2double((int)(~mk & 0x3FF00000u), 0), acc0);
            } else {
                if constexpr (k & 1) dst[(size_t)(k >> 1) * HP] = v2f64{pprev, pk};
                else pprev = pk;
            }
        });
        if constexpr (PHASE == 2) {
            // ... and by column allele over the sixteen lanes of the row: the four class sums of the column (DevContig::cls4 layout)
            const bool aj = (cur.bits1 >> j) & 1ull;
            const double s00 = row16_sum(aj ? 0.0 : acc0), s01 = row16_sum(aj ? acc0 : 0.0);
            const double s10 = row16_sum(aj ? 0.0 : acc1), s11 = row16_sum(aj ? acc1 : 0.0);
            if (act && j == 0) {
                gdouble2* o = (gdouble2*)cx.part + (size_t)t * 2u;
                o[0] = v2f64{s00, s01};
                o[1] = v2f64{s10, s11};
            }
            load_partner(t + 3, vp);
        }
        if (act) {   // the column's scale mantissa: sixteen columns collected in the row's lanes, one store per sixteen
            if (j == ((uint32_t)t & 15u)) buf = m;
            if (((uint32_t)t & 15u) == 15u || t + 1 == cx.hi) { if (j <= ((uint32_t)t & 15u) && (int64_t)((t & ~15ll) + j) >= (int64_t)first) cx.sc_a[(t & ~15ll) + j] = buf; }
        }
    };
    FRec rr[D];
    static_for<0, D>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; rr[i] = cx.live ? load_frec(cx.frec, (int64_t)first + i, cx.C) : FRec{}; });
    double vv[3][PHASE == 2 ? R : 1];
    if constexpr (PHASE == 2) { load_partner((int64_t)first, vv[0]); load_partner((int64_t)first + 1, vv[1]); load_partner((int64_t)first + 2, vv[2]); }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (no load of the prologue in flight inside the loop: see lean_forward)
    int n = 0;
    for (; n + D - 1 < n_steps; n += D)
        static_for<0, D>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; step(n + i, rr[i], vv[i % 3]); });
    static_for<0, D - 1>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; if (n < n_steps) { step(n, rr[i], vv[i % 3]); ++n; } });
    {   // the last column of the rows that ran to the wave's last step may itself have summed to zero
        double Cj = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) Cj += ee[k] * pp[k];
        const double Sl = row16_sum(Cj);
        if (cx.live && (int64_t)first + n_steps == cx.hi && !(Sl > 0.0)) flag_uniform(cx.hi - 1);
    }
}

template <int PHASE>
DEVI void small16_backward(const DevContig* contigs, const uint32_t* ids, uint32_t n_ids, uint32_t chunk, double* dump) {
    constexpr int HP = 16, R = 16;
    constexpr int D = PHASE == 2 ? 3 : 5;   // (see small16_forward)
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u;
    const uint32_t slot = blockIdx.x * 4u + (lane >> 4);
    const size_t colsz = (size_t)HP * HP;
    const double unif = 1.0 / 256.0;
    SmallCtx cx{};   // lo = bot, hi = top
    int64_t t0 = -1;
    double Sy = 1.0;
    double ee[R], pp[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { ee[k] = 1.0; pp[k] = 0.0; }
    if (slot < n_ids) {
        const DevContig& dc = contigs[ids[slot]];
        const int64_t C = *dc.n_cols, mid = C / 2, K = dc.chunk_cols;
        int64_t top = PHASE == 1 ? C - 1 : mid - 1, bot = PHASE == 1 ? mid : 0;
        bool ok = C > 0;
        if constexpr (PHASE == 2) ok = dc.small == 2u && C >= 2;
        if constexpr (PHASE == 3) {
            top = mid - 1 - (int64_t)chunk * K;
            ok = ok && top >= 0;
            bot = top - K + 1 > 0 ? top - K + 1 : 0;
        }
        ok = ok && top >= bot;
        if (ok) {
            cx.live = true; cx.C = C; cx.lo = bot; cx.hi = top;
            cx.frec = (gcdouble*)dc.frec; cx.sc_a = (gdouble*)dc.bscale; cx.sc_b = (gdouble*)dc.bsum;
            gdouble* cols = (gdouble*)dc.fwd;
            cx.wr = cols; cx.partner = (gcdouble*)cols; cx.part = (gdouble*)dc.part;
            cx.resume = (gcdouble*)(cols + (size_t)(top + 1 < C ? top + 1 : top) * colsz);
            if constexpr (PHASE == 3) {
                gdouble* scr = (gdouble*)dc.scratch;
                cx.wr = scr + (size_t)(PG_SCR_BUF(chunk) * 2u + 1u) * (size_t)K * colsz - (size_t)bot * colsz;
                if (chunk > 0) cx.resume = (gcdouble*)(scr + (size_t)(PG_SCR_BUF(chunk - 1u) * 2u + 1u) * (size_t)K * colsz);
            }
            t0 = PHASE == 1 ? top - 1 : top;
        }
    }
    const int n_steps = __builtin_amdgcn_readfirstlane(wave_max_i32(cx.live ? (int)(t0 - cx.lo + 1) : 0));
    if (__builtin_amdgcn_readfirstlane(wave_max_i32(cx.live ? 1 : 0)) == 0) return;
    auto emis = [&](const FRec& r, double& eA, double& eB) {
        const bool aj = (r.bits1 >> j) & 1ull;
        eA = aj ? r.E01 : r.E00;
        eB = aj ? r.E11 : r.E01;
    };
    if (cx.live) {
        double y[R];
        if constexpr (PHASE == 1) {
            // column C-1: beta~ = 1 (hmm.cpp:356-358), stored at the backward bias
            const double B0 = ldexp(1.0, PG_BIAS_B);
#pragma unroll
            for (int k = 0; k < R; ++k) y[k] = B0;
            Sy = 256.0 * B0;
            gdouble2* dst = (gdouble2*)(cx.wr + (size_t)cx.hi * colsz) + j;
#pragma unroll
            for (int k = 0; k < R; k += 2) dst[(size_t)(k >> 1) * HP] = v2f64{y[k], y[k + 1]};
            if (j == 0) { cx.sc_a[cx.hi] = 1.0; cx.sc_b[cx.hi] = Sy; }
        } else {
            gcdouble2* src = (gcdouble2*)cx.resume + j;
#pragma unroll
            for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; y[k] = t.x; y[k + 1] = t.y; }
            Sy = cx.sc_b[cx.hi + 1];
            if (!(Sy > 0.0)) {  // resuming behind an all-zero column: uniform (hmm.cpp:374-380)
#pragma unroll
                for (int k = 0; k < R; ++k) y[k] = unif;
                Sy = 1.0;
            }
        }
        const FRec r0 = load_frec(cx.frec, t0 + 1, cx.C);   // record t0+1: emission of column t0+1
        double eA, eB;
        emis(r0, eA, eB);
        const uint32_t rb = (uint32_t)(r0.bits1 & 0xFFFFull);
#pragma unroll
        for (int k = 0; k < R; ++k) { ee[k] = ((rb >> k) & 1u) ? eB : eA; pp[k] = y[k]; }
    }
    double one = 1.0, bufA = 0.0, bufB = 0.0;
    asm volatile("" : "+v"(one));
    // step n: column t = t0 - n.  `cur_rec` = record t+1 (constants of the gap t -> t+1; its variable then takes the
    // record three steps on), `nxt` = record t (emission of column t).
    // phase 2: the partner column P'_t of this lane's column, fetched three steps ahead (clamped: see small16_forward)
    auto load_partner = [&](int64_t c, double (&v)[R]) {
        if (!cx.live) return;
        c = c < cx.lo ? cx.lo : (c > cx.hi ? cx.hi : c);
        gcdouble2* src = (gcdouble2*)(cx.partner + (size_t)c * colsz) + j;
#pragma unroll
        for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; v[k] = t.x; v[k + 1] = t.y; }
    };
    auto step = [&](int n, FRec& cur_rec, const FRec& nxt, double (&vp)[PHASE == 2 ? R : 1]) __attribute__((always_inline)) {
        const int64_t t = t0 - n;
        const bool act = cx.live && t >= cx.lo;
        const FRec cur = cur_rec;
        if (cx.live) cur_rec = load_frec(cx.frec, t + 1 - D, cx.C);
        int es = exponent_of(Sy) - PG_BIAS_B;
        es = es < -900 ? -900 : es;
        const double m = ldexp(Sy, -es - PG_BIAS_B);
        const double k0 = ldexp(cur.c0, -es), k1 = ldexp(cur.c1, -es), k2 = ldexp(cur.c2, -es), kap = ldexp(cur.kappa, -es);
        double w[R], Cj = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; w[k] = ee[k] * pp[k]; Cj += w[k]; });
        const double Sw = row16_sum(Cj);
        const double ucol = dpp_source(k1 * Cj);
        const double uj = fma(k2, Sw, ucol);
        const double Snew = kap * Sw;  // = sum(beta'_t)
        double eA, eB;
        emis(nxt, eA, eB);
        const uint32_t rb = (uint32_t)(nxt.bits1 & 0xFFFFull);
        gdouble2* dst = act ? (gdouble2*)(cx.wr + (size_t)t * colsz) + j : (gdouble2*)dump + lane;
        double yprev = 0.0, acc0 = 0.0, acc1 = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double yk = fmac_row_bcast<k>(fma(k0, w[k], uj), ucol, one);  // beta'_t = k0 w + u_j + u_k
            const uint32_t mk = row_bit_mask<k>(rb);
            ee[k] = sel_by_mask(mk, eA, eB);
            pp[k] = yk;
            if constexpr (PHASE == 2) {   // (see small16_forward)
                const double pr = vp[k] * yk;
                acc1 = fma(pr, __hiloint2double((int)(mk & 0x3FF00000u), 0), acc1);
                acc0 = fma(pr, __hiloint2double((int)(~mk & 0x3FF00000u), 0), acc0);
            } else {
                if constexpr (k & 1) dst[(size_t)(k >> 1) * HP] = v2f64{yprev, yk};
                else yprev = yk;
            }
        });
        if constexpr (PHASE == 2) {
            const bool aj = (nxt.bits1 >> j) & 1ull;
            const double s00 = row16_sum(aj ? 0.0 : acc0), s01 = row16_sum(aj ? acc0 : 0.0);
            const double s10 = row16_sum(aj ? 0.0 : acc1), s11 = row16_sum(aj ? acc1 : 0.0);
            if (act && j == 0) {
                gdouble2* o = (gdouble2*)cx.part + (size_t)t * 2u;
                o[0] = v2f64{s00, s01};
                o[1] = v2f64{s10, s11};
            }
            load_partner(t - 3, vp);
        }
        Sy = Snew;
        if (act && !(Snew > 0.0)) {
            // beta~_t is all zero (what was stored IS zero): the next step starts from the uniform column (hmm.cpp:374-380)
#pragma unroll
            for (int k = 0; k < R; ++k) pp[k] = unif;
            Sy = 1.0;
        }
        if (act) {   // scale mantissa and sum of the column: sixteen columns collected in the row's lanes (descending)
            const uint32_t q = (uint32_t)t & 15u;
            if (j == q) { bufA = m; bufB = Snew; }
            if (q == 0u || t == cx.lo) {
                const int64_t c = (t & ~15ll) + j;
                if (j >= q && c <= t0) { cx.sc_a[c] = bufA; cx.sc_b[c] = bufB; }
            }
        }
    };
    FRec rr[D];   // rr[i] = record t0 + 1 - i: step n takes its constants from rr[n % D] (record t + 1), its emission from the next one (record t)
    static_for<0, D>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; rr[i] = cx.live ? load_frec(cx.frec, t0 + 1 - i, cx.C) : FRec{}; });
    double vv[3][PHASE == 2 ? R : 1];
    if constexpr (PHASE == 2) { load_partner(t0, vv[0]); load_partner(t0 - 1, vv[1]); load_partner(t0 - 2, vv[2]); }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int n = 0;
    for (; n + D - 1 < n_steps; n += D)
        static_for<0, D>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; step(n + i, rr[i], rr[(i + 1) % D], vv[i % 3]); });
    static_for<0, D - 1>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; if (n < n_steps) { step(n, rr[i], rr[(i + 1) % D], vv[i % 3]); ++n; } });
}

template <int PHASE>
__global__ __launch_bounds__(64) void k_sweep_small16(const DevContig* __restrict__ contigs, const uint32_t* __restrict__ ids, uint32_t n_ids, uint32_t chunk,
                                                      double* dump) {
    if (blockIdx.y == 0) small16_forward<PHASE>(contigs, ids, n_ids, chunk, dump);
    else small16_backward<PHASE>(contigs, ids, n_ids, chunk, dump);
}
#include "pg_small16x.h"   // k_sweep_small16x: the same step with table emissions — 16-path chains with multiallelic objects, wide columns per column

// ------------------------------------------------------------------------------------------
//  k_sweep_generic : the same half-chains for any HP = 64 .. 1024 (power of two), store-only phases
//  (1 and 3; the posteriors come from k_post).  Used for HP >= 256 — more states per column than a
//  workgroup's registers hold (reference README.md:260 allows up to 65534 paths; its own integration
//  fixture runs 215 in one subset, tests/CommandsTest.cpp:31) — and on request (PG_SWEEP_KERNEL=generic)
//  for HP = 64 / 128 as an independently written cross-check of the register-resident kernels.
//
//  1024 threads; thread -> column j = tid % HP and the row pairs ip = g + NG n (g = tid / HP, NG = 1024 / HP):
//  the same elements every column, so the column after its emission multiply (x forward, w backward)
//  lives in an element-private global scratch (xbuf: each thread re-reads only what it wrote itself).
//  Per column: partial column sums -> LDS -> barrier -> column sums -> LDS -> barrier -> total (every
//  wave, fixed order: deterministic) -> one pass over the thread's elements.  Same stored columns, side
//  arrays, fall-back rules and resume conventions as forward_body / backward_body.
// ------------------------------------------------------------------------------------------
#define PG_GEN_THREADS 1024
#define PG_GEN_RB (PG_REC_ALLELES + PG_MAX_PATHS + 64)
struct GenShared {
    unsigned char rec[2][PG_GEN_RB] __attribute__((aligned(16)));
    double psum[PG_GEN_THREADS];
    double colsum[PG_MAX_PATHS];
};
struct GenPos {
    uint32_t tid, lane, j, g, HP, NG, N, H;
};
DEVI void gen_stage_record(GenShared& sh, const DevContig& dc, int64_t c, uint32_t C, uint32_t tid) {
    if (c < 0 || c >= (int64_t)C) return;
    const uint32_t words = dc.RB / 8u;
    if (tid < words) ((unsigned long long*)sh.rec[(uint32_t)c & 1u])[tid] = ((const unsigned long long*)(dc.colrec + (size_t)c * dc.RB))[tid];
}
struct GenRec {
    double c0, c1, c2, kappa;
    const unsigned char* al;
    const double* E;      // narrow table (LDS)
    const double* wide;   // wide table (global) or nullptr
    uint32_t wn;
};
DEVI GenRec gen_decode(const GenShared& sh, const DevContig& dc, uint32_t c) {
    const unsigned char* rec = sh.rec[c & 1u];
    GenRec r;
    r.c0 = *(const double*)(rec + PG_REC_C0); r.c1 = *(const double*)(rec + PG_REC_C1);
    r.c2 = *(const double*)(rec + PG_REC_C2); r.kappa = *(const double*)(rec + PG_REC_KAPPA);
    r.al = rec + PG_REC_ALLELES;
    r.E = (const double*)(rec + PG_REC_E);
    r.wn = rec[PG_REC_NLOCAL];
    r.wide = nullptr;
    if (r.wn > PG_AMAX) r.wide = (const double*)(dc.wide + (size_t)(*(const uint32_t*)(rec + PG_REC_WIDE_IDX)) * 16u);
    return r;
}
DEVI double gen_emission(const GenRec& r, uint32_t i, uint32_t j) {
    const uint32_t air = r.al[i], ajr = r.al[j];
    if (r.wide) {
        const uint32_t ai = air > r.wn ? r.wn : air, aj = ajr > r.wn ? r.wn : ajr;
        return r.wide[ai * (r.wn + 1u) + aj];
    }
    const uint32_t ai = air > PG_AMAX ? PG_AMAX : air, aj = ajr > PG_AMAX ? PG_AMAX : ajr;
    return r.E[ai * PG_ESTRIDE + aj];
}
// column sums from the per-thread partials (call between two barriers); returns this thread's C_j
DEVI double gen_colsums(GenShared& sh, const GenPos& p) {
    double cj = 0.0;
    for (uint32_t g = 0; g < p.NG; ++g) cj += sh.psum[g * p.HP + p.j];
    if (p.g == 0) sh.colsum[p.j] = cj;
    return cj;
}
DEVI double gen_total(const GenShared& sh, const GenPos& p) {  // after the barrier behind gen_colsums
    double t = 0.0;
    for (uint32_t k = p.lane; k < p.HP; k += 64) t += sh.colsum[k];
    return wave_sum(t);
}

template <int PHASE>
DEVI void gen_forward(const DevContig& dc, GenShared& sh, uint32_t C, uint32_t chunk, const GenPos& p) {
    const uint32_t mid = C / 2, K = dc.chunk_cols, HP = p.HP, H = p.H;
    uint32_t lo = PHASE == 1 ? 0u : mid, hi = PHASE == 1 ? mid : C;
    if constexpr (PHASE == 3) {
        const unsigned long long l = (unsigned long long)mid + (unsigned long long)chunk * K;
        if (l >= C) return;
        lo = (uint32_t)l;
        hi = C - lo > K ? lo + K : C;
    }
    if (lo >= hi) return;
    const uint32_t first = lo == 0 ? 1u : lo;
    const size_t colsz = (size_t)HP * HP;
    const double unif = 1.0 / ((double)H * (double)H);
    double* fwd = dc.fwd;
    double* wr = fwd;
    const double* resume = fwd + (size_t)(lo > 0 ? lo - 1 : 0) * colsz;
    if constexpr (PHASE == 3) {
        wr = dc.scratch + (size_t)(PG_SCR_BUF(chunk) * 2u) * K * colsz - (size_t)lo * colsz;
        if (chunk > 0) resume = dc.scratch + ((size_t)(PG_SCR_BUF(chunk - 1u) * 2u) * K + (K - 1u)) * colsz;
    }
    v2f64* xb = (v2f64*)dc.xbuf;  // forward role: first half of xbuf
    auto real = [&](uint32_t i) { return (i < H && p.j < H); };

    gen_stage_record(sh, dc, (int64_t)first - 1, C, p.tid);
    gen_stage_record(sh, dc, (int64_t)first, C, p.tid);
    __syncthreads();
    {
        const GenRec r0 = gen_decode(sh, dc, first - 1);
        double part = 0.0;
        if (lo == 0) {
            const double P0 = ldexp(1.0, PG_BIAS_F);
            v2f64* dst = (v2f64*)wr;
            for (uint32_t n = 0; n < p.N; ++n) {
                const uint32_t ip = p.g + p.NG * n;
                const size_t e = (size_t)ip * HP + p.j;
                dst[e] = v2f64{P0, P0};
                const v2f64 xv = {gen_emission(r0, 2 * ip, p.j) * P0, gen_emission(r0, 2 * ip + 1, p.j) * P0};
                xb[e] = xv;
                part += xv.x + xv.y;
            }
            if (p.tid == 0) dc.fscale[0] = 1.0;
        } else {
            const bool was_uniform = dc.fwd_fallback[lo - 1] != 0;
            const v2f64* src = (const v2f64*)resume;
            for (uint32_t n = 0; n < p.N; ++n) {
                const uint32_t ip = p.g + p.NG * n;
                const size_t e = (size_t)ip * HP + p.j;
                v2f64 xv = src[e];
                if (!was_uniform) { xv.x *= gen_emission(r0, 2 * ip, p.j); xv.y *= gen_emission(r0, 2 * ip + 1, p.j); }
                xb[e] = xv;
                part += xv.x + xv.y;
            }
        }
        sh.psum[p.tid] = part;
    }
    auto flag_uniform = [&](uint32_t cprev) {
        if (cprev >= lo) {
            v2f64* dst = (v2f64*)(wr + (size_t)cprev * colsz);
            for (uint32_t n = 0; n < p.N; ++n) {
                const uint32_t ip = p.g + p.NG * n;
                dst[(size_t)ip * HP + p.j] = v2f64{real(2 * ip) ? unif : 0.0, real(2 * ip + 1) ? unif : 0.0};
            }
        }
        if (p.tid == 0) dc.fwd_fallback[cprev] = 1;
    };
    for (uint32_t t = first; t < hi; ++t) {
        __syncthreads();  // partial sums of column t-1, record t staged
        double Cj = gen_colsums(sh, p);
        gen_stage_record(sh, dc, (int64_t)t + 1, C, p.tid);
        __syncthreads();
        double S = gen_total(sh, p);
        const GenRec r = gen_decode(sh, dc, t);
        const bool fbk = !(S > 0.0);
        const double Cu = (double)H * unif;
        if (fbk) { flag_uniform(t - 1); S = 1.0; Cj = p.j < H ? Cu : 0.0; }
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        const double m = ldexp(S, -es - PG_BIAS_F);
        const double sc = ldexp(1.0, -es), c0s = ldexp(r.c0, -es), ujs = ldexp(fma(r.c2, S, r.c1 * Cj), -es);
        v2f64* dst = (v2f64*)(wr + (size_t)t * colsz);
        double part = 0.0;
        for (uint32_t n = 0; n < p.N; ++n) {
            const uint32_t ip = p.g + p.NG * n, i0 = 2 * ip, i1 = 2 * ip + 1;
            const size_t e = (size_t)ip * HP + p.j;
            v2f64 xv;
            double u0, u1;
            if (fbk) {
                xv = v2f64{real(i0) ? unif : 0.0, real(i1) ? unif : 0.0};
                u0 = i0 < H ? r.c1 * Cu : 0.0; u1 = i1 < H ? r.c1 * Cu : 0.0;
            } else {
                xv = xb[e];
                u0 = r.c1 * sh.colsum[i0]; u1 = r.c1 * sh.colsum[i1];
            }
            const double p0 = fma(c0s, xv.x, fma(u0, sc, ujs)), p1 = fma(c0s, xv.y, fma(u1, sc, ujs));
            dst[e] = v2f64{p0, p1};
            xv.x = p0 * gen_emission(r, i0, p.j);
            xv.y = p1 * gen_emission(r, i1, p.j);
            xb[e] = xv;
            part += xv.x + xv.y;
        }
        if (p.tid == 0) dc.fscale[t] = m;
        __syncthreads();  // every thread is done with colsum / psum of the previous column
        sh.psum[p.tid] = part;
    }
    __syncthreads();
    gen_colsums(sh, p);
    __syncthreads();
    if (!(gen_total(sh, p) > 0.0)) flag_uniform(hi - 1);  // the last column of this phase may itself have summed to zero
}

template <int PHASE>
DEVI void gen_backward(const DevContig& dc, GenShared& sh, uint32_t C, uint32_t chunk, const GenPos& p) {
    const int64_t mid = C / 2, K = dc.chunk_cols;
    const uint32_t HP = p.HP, H = p.H;
    int64_t top = PHASE == 1 ? (int64_t)C - 1 : mid - 1;
    int64_t bot = PHASE == 1 ? mid : 0;
    if constexpr (PHASE == 3) {
        top = mid - 1 - (int64_t)chunk * K;
        if (top < 0) return;
        bot = top - K + 1 > 0 ? top - K + 1 : 0;
    }
    if (top < bot) return;
    const int64_t t0 = PHASE == 1 ? top - 1 : top;
    const size_t colsz = (size_t)HP * HP;
    const double unif = 1.0 / ((double)H * (double)H);
    double* cols = dc.fwd;
    double* wr = cols;
    const double* resume = cols + (size_t)(top + 1 < (int64_t)C ? top + 1 : top) * colsz;
    if constexpr (PHASE == 3) {
        wr = dc.scratch + (size_t)(PG_SCR_BUF(chunk) * 2u + 1u) * (size_t)K * colsz - (size_t)bot * colsz;
        if (chunk > 0) resume = dc.scratch + (size_t)(PG_SCR_BUF(chunk - 1u) * 2u + 1u) * (size_t)K * colsz;
    }
    v2f64* wb = (v2f64*)(dc.xbuf + colsz);  // backward role: second half of xbuf
    auto real = [&](uint32_t i) { return (i < H && p.j < H); };

    gen_stage_record(sh, dc, t0 + 1, C, p.tid);
    gen_stage_record(sh, dc, t0, C, p.tid);
    __syncthreads();
    double Sy;
    {
        // w = beta'_{t0+1} . e_{t0+1}: the all-ones last column (phase 1) or the column stored last
        const GenRec r1 = gen_decode(sh, dc, (uint32_t)(t0 + 1));
        double part = 0.0;
        if constexpr (PHASE == 1) {
            const double B0 = ldexp(1.0, PG_BIAS_B);
            v2f64* dst = (v2f64*)(wr + (size_t)top * colsz);
            for (uint32_t n = 0; n < p.N; ++n) {
                const uint32_t ip = p.g + p.NG * n;
                const size_t e = (size_t)ip * HP + p.j;
                const v2f64 yv = {real(2 * ip) ? B0 : 0.0, real(2 * ip + 1) ? B0 : 0.0};
                dst[e] = yv;
                const v2f64 wv = {yv.x * gen_emission(r1, 2 * ip, p.j), yv.y * gen_emission(r1, 2 * ip + 1, p.j)};
                wb[e] = wv;
                part += wv.x + wv.y;
            }
            Sy = (double)H * (double)H * B0;
            if (p.tid == 0) { dc.bscale[top] = 1.0; dc.bsum[top] = Sy; }
        } else {
            Sy = dc.bsum[top + 1];
            const bool zero = !(Sy > 0.0);  // resuming behind an all-zero column: uniform (hmm.cpp:374-380)
            if (zero) Sy = 1.0;
            const v2f64* src = (const v2f64*)resume;
            for (uint32_t n = 0; n < p.N; ++n) {
                const uint32_t ip = p.g + p.NG * n;
                const size_t e = (size_t)ip * HP + p.j;
                v2f64 yv = src[e];
                if (zero) yv = v2f64{real(2 * ip) ? unif : 0.0, real(2 * ip + 1) ? unif : 0.0};
                const v2f64 wv = {yv.x * gen_emission(r1, 2 * ip, p.j), yv.y * gen_emission(r1, 2 * ip + 1, p.j)};
                wb[e] = wv;
                part += wv.x + wv.y;
            }
        }
        sh.psum[p.tid] = part;
    }
    for (int64_t t = t0; t >= bot; --t) {
        const GenRec r1 = gen_decode(sh, dc, (uint32_t)(t + 1));  // constants of the gap t -> t+1
        int es = exponent_of(Sy) - PG_BIAS_B;
        es = es < -900 ? -900 : es;
        const double m = ldexp(Sy, -es - PG_BIAS_B);
        const double k0 = ldexp(r1.c0, -es), k1 = ldexp(r1.c1, -es), k2 = ldexp(r1.c2, -es), kap = ldexp(r1.kappa, -es);
        __syncthreads();  // partial sums of w, record t staged, constants of record t+1 read by everyone
        const double Cj = gen_colsums(sh, p);
        gen_stage_record(sh, dc, t - 1, C, p.tid);  // into the buffer of record t+1
        __syncthreads();
        const double Sw = gen_total(sh, p);
        const GenRec r0 = gen_decode(sh, dc, (uint32_t)t);
        const double uj = fma(k2, Sw, k1 * Cj);
        const double Snew = kap * Sw;
        const bool zero = !(Snew > 0.0);
        v2f64* dst = (v2f64*)(wr + (size_t)t * colsz);
        double part = 0.0;
        for (uint32_t n = 0; n < p.N; ++n) {
            const uint32_t ip = p.g + p.NG * n, i0 = 2 * ip, i1 = 2 * ip + 1;
            const size_t e = (size_t)ip * HP + p.j;
            v2f64 yv, wv;
            if (zero) {
                // beta~_t is all zero: its own posteriors are 0, the next step starts from the uniform column
                yv = v2f64{0.0, 0.0};
                wv = v2f64{real(i0) ? unif : 0.0, real(i1) ? unif : 0.0};
            } else {
                const v2f64 wo = wb[e];
                yv.x = fma(k0, wo.x, k1 * sh.colsum[i0] + uj);
                yv.y = fma(k0, wo.y, k1 * sh.colsum[i1] + uj);
                wv = yv;
            }
            dst[e] = yv;
            wv.x *= gen_emission(r0, i0, p.j);
            wv.y *= gen_emission(r0, i1, p.j);
            wb[e] = wv;
            part += wv.x + wv.y;
        }
        if (p.tid == 0) { dc.bscale[t] = m; dc.bsum[t] = Snew; }
        Sy = zero ? 1.0 : Snew;
        __syncthreads();
        sh.psum[p.tid] = part;
    }
}

template <int PHASE>
__global__ __launch_bounds__(PG_GEN_THREADS) void k_sweep_generic(const DevContig* __restrict__ contigs, uint32_t chunk, uint32_t min_hp) {
    __shared__ GenShared sh;
    const DevContig& dc = contigs[blockIdx.x];
    if (dc.HP < min_hp || dc.HP < 64u) return;
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    GenPos p;
    p.tid = threadIdx.x; p.lane = p.tid & 63u; p.HP = dc.HP; p.H = dc.H;
    p.j = p.tid % p.HP; p.g = p.tid / p.HP; p.NG = PG_GEN_THREADS / p.HP; p.N = p.HP / (2u * p.NG);
    if (blockIdx.y == 0) gen_forward<PHASE>(dc, sh, C, chunk, p);
    else gen_backward<PHASE>(dc, sh, C, chunk, p);
}

// ------------------------------------------------------------------------------------------
//  k_bins : posterior partials -> genotype bins (one wave per column)
//  L_v({a,b}) = sum over states (i,j) with alleles {a,b} of alpha_hat * beta~ * fsum
//  (reference src/hmm.cpp:364-368); exponent = X_c + X_{c+1}.
// ------------------------------------------------------------------------------------------

// A finished posterior bin: sum * (pm * 2^pe) * 2^xexp  ->  mantissa in [0.5,1) (or 0) and exponent.
// `sum` is the fp64 sum of P' * beta' over the states of the bin divided by the column scales, (pm, pe)
// the emission product of the bin's allele pair (full range: pe is an int), xexp the exponents the
// sweeps carried outside the columns (emission exponent of the next column, column biases).
DEVI void store_bin(double* lik, int32_t* lik_exp, uint64_t idx, double sum, double pm, int pe, int xexp) {
    const double val = sum * pm;
    double mm = val;
    int ee = 0;
    if (val != 0.0 && val == val && !isinf(val)) {
        mm = frexp(val, &ee);
        ee += pe + xexp;
    }
    lik[idx] = mm;
    lik_exp[idx] = ee;
}

// chains of up to 32 paths (at most 64 partial entries per column and slot pair): one THREAD per column (k_bins_thin) —
// a wave per column spent ~460 vector instructions on each (eight 64-lane sums, the index arithmetic and a division, all
// wave-wide for one column), 7 ms for the 8.2 M columns of `cohort_h17`
DEVI bool bins_x(const DevContig& dc, uint32_t C) { return dc.smallx == 2u && C >= 2u && !dc.split; }   // k_bins_x / k_bins_wide (chains on k_sweep_small16x<2>)
DEVI bool bins_thin(const DevContig& dc) { return dc.T <= 64u && dc.HP <= 32u && !dc.cls4 && dc.chunk_cols == 0u && !bins_x(dc, *dc.n_cols) && !dc.split; }

DEVI void bins_unit(const DevContig& dc, uint32_t unit, double (&s_bins)[4][PG_AMAX * (PG_AMAX + 1) / 2]) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c = unit * 4 + wave;
    const uint32_t C = *dc.n_cols;
    if (c >= C) return;
    if (dc.split) return;                                // the split path: k_bins_s / k_bins_wide_s (pg_split.h)
    if ((dc.tri == 2u && C >= 2u) || dc.cls4) return;  // chains whose class sums arrive finished (k_sweep_lean2, DevContig::cls4): k_bins_lean2
    if (bins_x(dc, C)) return;                           // chains on k_sweep_small16x<2>: k_bins_x, k_bins_wide
    if (bins_thin(dc)) return;                           // few partial entries per column: k_bins_thin
    if (dc.leanx2) return;                               // chains on k_sweep_leanx2 (64 entries per column and slot pair): k_bins_q
    // (chains with compact records only have no column-order copy of the records: the variant's own record)
    const bool direct = compact_records_only(dc, C);
    const unsigned char* rec = direct ? dc.vrec + (size_t)dc.col_variant[c] * dc.RB : dc.colrec + (size_t)c * dc.RB;
    const uint32_t v = *(const uint32_t*)(rec + PG_REC_VARIANT);
    const uint32_t nl = rec[PG_REC_NLOCAL];
    if (dc.widef && nl > (uint32_t)PG_AMAX) return;      // a wide column of a DevContig::widef chain: k_bins_wide (its column is in the aux slot)
    const uint32_t T = dc.T, HP = dc.HP;
    const unsigned char* al = rec + PG_REC_ALLELES;
    if (lane < PG_AMAX * (PG_AMAX + 1) / 2) s_bins[wave][lane] = 0.0;
    wave_sync();
    const bool fb = dc.fwd_fallback[c] != 0;
    if (fb && c >= C / 2) {
        // The forward column of c fell back to uniform (alpha_hat*fsum = 1/H^2 for every real
        // state, reference src/hmm.cpp:259-266) AFTER the forward half-chain had already formed
        // this column's partials from the all-zero column: re-form the bins from the stored
        // backward column beta'_c (slot c, row-pair layout).  Rare path.
        const uint32_t H = dc.H;
        const double unif = 1.0 / ((double)H * (double)H);
        const double* col = dc.fwd + (size_t)c * dc.col_stride;
        for (uint32_t a = 0; a < nl; ++a)
            for (uint32_t b = 0; b < nl; ++b) {
                double s = 0.0;
                for (uint32_t st = lane; st < H * H; st += 64) {
                    const uint32_t i = st / H, jj = st % H;
                    if (al[i] == a && al[jj] == b) {
                        if (dc.tri) {  // the column was stored as its upper triangle, the diagonal halved
                            const uint32_t lo3 = i < jj ? i : jj, hi3 = i < jj ? jj : i;
                            const double val = col[(size_t)tri_unit_of(lo3 >> 1, hi3) * 2 + (lo3 & 1u)];
                            s += lo3 == hi3 ? 2.0 * val : val;
                        } else {
                            s += col[((size_t)(i >> 1) * HP + jj) * 2 + (i & 1u)];
                        }
                    }
                }
                const double tot = wave_sum(s) * unif;
                if (lane == 0) {
                    const uint32_t lo2 = a < b ? a : b, hi2 = a < b ? b : a;
                    s_bins[wave][tri_local(lo2, hi2)] += tot;
                }
            }
    } else {
        // partials of thread t for the row-allele pair q: part[((c * part_slots/2 + q) * T + t)] = {a = 2q, a = 2q+1};
        // the column allele of thread t is al[t % HP].  All of a lane's 16-byte loads of a pair are
        // issued together, then split by column allele with selects (no dynamic register indexing).
        // (chains on k_sweep_lean2 never get here: their class sums arrive finished and k_bins_lean2 turns them into bins)
        const uint32_t Tn = T;
        const v2f64* base = (const v2f64*)dc.part + (size_t)c * (dc.part_slots >> 1) * T;
        const uint32_t nq = (nl + 1u) >> 1;
        for (uint32_t q = 0; q < nq; ++q) {
            double acc0[PG_AMAX], acc1[PG_AMAX];
#pragma unroll
            for (int bb = 0; bb < PG_AMAX; ++bb) { acc0[bb] = 0.0; acc1[bb] = 0.0; }
            for (uint32_t t = lane; t < Tn; t += 64) {
                const uint32_t b = al[t % HP];
                v2f64 pv = v2f64{0.0, 0.0};
                if (b != PG_PHANTOM) pv = base[(size_t)q * Tn + t];   // (phantom paths' entries: zero, or — DevContig::live — never written)
#pragma unroll
                for (int bb = 0; bb < PG_AMAX; ++bb) {
                    acc0[bb] += b == (uint32_t)bb ? pv.x : 0.0;
                    acc1[bb] += b == (uint32_t)bb ? pv.y : 0.0;
                }
            }
            const uint32_t ra0 = 2u * q, ra1 = 2u * q + 1u;
#pragma unroll
            for (int bb = 0; bb < PG_AMAX; ++bb) {
                if ((uint32_t)bb < nl) {
                    const double t0 = wave_sum(acc0[bb]), t1 = wave_sum(acc1[bb]);
                    if (lane == 0) {
                        const uint32_t cb = (uint32_t)bb;
                        s_bins[wave][tri_local(ra0 < cb ? ra0 : cb, ra0 < cb ? cb : ra0)] += t0;
                        if (ra1 < nl) s_bins[wave][tri_local(ra1 < cb ? ra1 : cb, ra1 < cb ? cb : ra1)] += t1;
                    }
                }
            }
        }
    }
    wave_sync();
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0;
    const uint16_t* ls = (const uint16_t*)(rec + PG_REC_LOCAL_SLOT);
    // L_v({a,b}) = e(a,b) * sum over the bin of P'_c beta'_c / (m_f m_b) * 2^(X_{c+1} - BIAS_F - BIAS_B):
    // stored columns are (true value) * m * 2^bias (see forward_body).  A flagged forward column is the
    // uniform column itself: absolute value, no emission, no scale, no bias.
    const double scale = 1.0 / ((fb ? 1.0 : dc.fscale[c]) * dc.bscale[c]);
    int xexp = -((fb ? 0 : PG_BIAS_F) + PG_BIAS_B);
    if (c + 1 < C)
        xexp += *(const int32_t*)((direct ? dc.vrec + (size_t)dc.col_variant[c + 1] * dc.RB : dc.colrec + (size_t)(c + 1) * dc.RB) + PG_REC_EXP);
    // triangle storage: the partials are sums over the upper triangle with the diagonal halved = half of the bin
    if (dc.tri && !(fb && c >= C / 2)) xexp += 1;
    const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
    const unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
    if (lane < nl * nl) {
        const uint32_t la = lane / nl, lb = lane % nl;
        if (la <= lb) {
            const uint32_t sa = ls[la], sb = ls[lb];
            const uint64_t idx = dc.geno_off[v] + (uint64_t)sa * A - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
            const uint32_t pi = tri_n(la, lb, pn);
            const double pm = fb ? 0.5 : ((const double*)vp)[pi];
            const int pe = fb ? 1 : ((const int*)(vp + (size_t)NP * 8u))[pi];
            store_bin(dc.lik, dc.lik_exp, idx, s_bins[wave][tri_local(la, lb)] * scale, pm, pe, xexp);
        }
    }
}
// ------------------------------------------------------------------------------------------
//  k_bins_q (round 6) : the bins of chains on k_sweep_leanx2 — 64 partial entries per column and slot pair, already added up over
//  the sweep's four waves.  FOUR columns per wave, a DPP row of 16 lanes each (a lane takes four of the 64 entries; the sums over
//  the row are four DPP steps): a wave per column (k_bins) spent its time on the chain of dependent loads in front of and behind
//  1 - 3 KB of partials and on 64-lane sums — 3.95 ms for the 4.1 M columns of cohort_h64m.  Same factors, exponents, fall-back
//  rule (bins re-formed from the stored backward column) and triangle doubling as bins_unit.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bins_q(const DevContig* __restrict__ contigs) {
    constexpr int NB = PG_AMAX * (PG_AMAX + 1) / 2;
    __shared__ double s_bins[16][NB + 1];
    const DevContig& dc = contigs[blockIdx.y];
    if (!dc.leanx2 || dc.split) return;
    const uint32_t C = *dc.n_cols;
    if (blockIdx.x * 16u >= C) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, g = lane >> 4, l = lane & 15u;
    const uint32_t cc = blockIdx.x * 16u + wave * 4u + g;
    const bool valid = cc < C;
    const uint32_t c = valid ? cc : C - 1u;   // (rows past the last column run along on its data — DPP steps and LDS syncs are the wave's — and store nothing)
    double* sb = s_bins[wave * 4u + g];
    const unsigned char* rec = dc.colrec + (size_t)c * dc.RB;   // (triangle chains that are not lean chains keep column-order records)
    const uint32_t v = *(const uint32_t*)(rec + PG_REC_VARIANT);
    const uint32_t nl = rec[PG_REC_NLOCAL];
    const unsigned char* al = rec + PG_REC_ALLELES;
    uint32_t bl[4];   // column alleles of this lane's four entries (entry t belongs to column allele al[t])
#pragma unroll
    for (int i = 0; i < 4; ++i) bl[i] = al[l + 16u * (uint32_t)i];
    const bool fb = dc.fwd_fallback[c] != 0;
    const bool reform = fb && c >= C / 2;
    if (l < (uint32_t)NB) sb[l] = 0.0;
    wave_sync_lds();
    if (__any(reform)) {
        // (bins_unit) the forward column fell back to uniform after this column's partials had been formed from the all-zero
        // column: the bins are those of the uniform column times the stored backward column (upper triangle, diagonal halved)
        const uint32_t H = dc.H;
        const double unif = 1.0 / ((double)H * (double)H);
        const double* col = dc.fwd + (size_t)c * dc.col_stride;
        const uint32_t nlw = (uint32_t)__builtin_amdgcn_readfirstlane(wave_max_i32(reform ? (int)nl : 0));
        for (uint32_t a = 0; a < nlw; ++a)
            for (uint32_t b = 0; b < nlw; ++b) {
                double sacc = 0.0;
                if (reform && a < nl && b < nl)
                    for (uint32_t st = l; st < H * H; st += 16u) {
                        const uint32_t i = st / H, jj = st % H;
                        if (al[i] == a && al[jj] == b) {
                            const uint32_t lo3 = i < jj ? i : jj, hi3 = i < jj ? jj : i;
                            const double val = col[(size_t)tri_unit_of(lo3 >> 1, hi3) * 2 + (lo3 & 1u)];
                            sacc += lo3 == hi3 ? 2.0 * val : val;
                        }
                    }
                const double tot = row16_sum(sacc) * unif;
                if (l == 0 && reform && a < nl && b < nl) sb[tri_local(a < b ? a : b, a < b ? b : a)] += tot;
            }
    }
    {
        const uint32_t nq = reform ? 0u : (nl + 1u) >> 1;
        const uint32_t nqw = (uint32_t)__builtin_amdgcn_readfirstlane(wave_max_i32((int)nq));
        const uint32_t nlw = (uint32_t)__builtin_amdgcn_readfirstlane(wave_max_i32(reform ? 0 : (int)nl));
        const v2f64* base = (const v2f64*)dc.part + (size_t)c * (dc.part_slots >> 1) * 64u + l;
        for (uint32_t q = 0; q < nqw; ++q) {
            const bool mine = q < nq;
            v2f64 pv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { pv[i] = v2f64{0.0, 0.0}; if (mine) pv[i] = base[(size_t)q * 64u + 16u * (uint32_t)i]; }
            const uint32_t ra0 = 2u * q, ra1 = 2u * q + 1u;
#pragma unroll
            for (int bb = 0; bb < PG_AMAX; ++bb) {
                if ((uint32_t)bb < nlw) {   // (uniform)
                    double a0s = 0.0, a1s = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a0s += bl[i] == (uint32_t)bb ? pv[i].x : 0.0; a1s += bl[i] == (uint32_t)bb ? pv[i].y : 0.0; }
                    const double t0 = row16_sum(a0s), t1 = row16_sum(a1s);
                    if (l == 0 && mine && (uint32_t)bb < nl) {
                        const uint32_t cb = (uint32_t)bb;
                        sb[tri_local(ra0 < cb ? ra0 : cb, ra0 < cb ? cb : ra0)] += t0;
                        if (ra1 < nl) sb[tri_local(ra1 < cb ? ra1 : cb, ra1 < cb ? cb : ra1)] += t1;
                    }
                }
            }
        }
    }
    wave_sync_lds();
    if (!valid) return;
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0;
    const uint16_t* ls = (const uint16_t*)(rec + PG_REC_LOCAL_SLOT);
    const double scale = 1.0 / ((fb ? 1.0 : dc.fscale[c]) * dc.bscale[c]);
    int xexp = -((fb ? 0 : PG_BIAS_F) + PG_BIAS_B);
    if (c + 1 < C) xexp += *(const int32_t*)(dc.colrec + (size_t)(c + 1) * dc.RB + PG_REC_EXP);
    if (!reform) xexp += 1;   // triangle storage: the partials are sums over the stored half = half of the bin
    const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
    const unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
    for (uint32_t pidx = l; pidx < nl * nl; pidx += 16u) {
        const uint32_t la = pidx / nl, lb = pidx % nl;
        if (la <= lb) {
            const uint32_t sa = ls[la], sbb = ls[lb];
            const uint64_t idx = dc.geno_off[v] + (uint64_t)sa * A - (uint64_t)sa * (sa - 1) / 2 + (sbb - sa);
            const uint32_t pi = tri_n(la, lb, pn);
            const double pm = fb ? 0.5 : ((const double*)vp)[pi];
            const int pe = fb ? 1 : ((const int*)(vp + (size_t)NP * 8u))[pi];
            store_bin(dc.lik, dc.lik_exp, idx, sb[tri_local(la, lb)] * scale, pm, pe, xexp);
        }
    }
}

__global__ __launch_bounds__(256) void k_bins(const DevContig* __restrict__ contigs) {
    __shared__ double s_bins[4][PG_AMAX * (PG_AMAX + 1) / 2];
    const DevContig& dc = contigs[blockIdx.y];
    bins_unit(dc, blockIdx.x, s_bins);
}

// ------------------------------------------------------------------------------------------
//  k_bins_thin : the same bins for chains with at most 64 partial entries per column and slot pair (bins_thin) — one
//  THREAD per column.  The thread walks the real paths' entries of its column (entry t belongs to column allele
//  al[t % HP]) and adds each pair of row-allele sums to its bins, kept in a private LDS row (the bin index is data).
//  Same factors, exponents and fall-back rule as bins_unit; the sums are taken in entry order.
// ------------------------------------------------------------------------------------------
#define PG_NBINS (PG_AMAX * (PG_AMAX + 1) / 2)
__global__ __launch_bounds__(256) void k_bins_thin(const DevContig* __restrict__ contigs) {
    __shared__ double s_acc[256][PG_NBINS + 2];   // (17 doubles: an odd row stride spreads the threads' rows over the banks)
    const DevContig& dc = contigs[blockIdx.y];
    if (!bins_thin(dc)) return;
    const uint32_t C = *dc.n_cols;
    if (blockIdx.x * 256u >= C) return;
    // (threads beyond the last column run along on its data — the wave takes its loop bounds together — and store nothing)
    const bool active = blockIdx.x * 256u + threadIdx.x < C;
    const uint32_t c = active ? blockIdx.x * 256u + threadIdx.x : C - 1u;
    const bool direct = compact_records_only(dc, C);
    const unsigned char* rec = direct ? dc.vrec + (size_t)dc.col_variant[c] * dc.RB : dc.colrec + (size_t)c * dc.RB;
    const uint32_t v = *(const uint32_t*)(rec + PG_REC_VARIANT);
    const uint32_t nl = rec[PG_REC_NLOCAL];
    const uint32_t T = dc.T, HP = dc.HP, H = dc.H;
    if (active && nl > (uint32_t)PG_AMAX) atomicOr(dc.err, PG_DEVERR_WIDE_FUSED);   // (fused jobs take wide columns on k_sweep_small16x only; this is that job's single-column chain)
    const unsigned char* al = rec + PG_REC_ALLELES;
    double* acc = s_acc[threadIdx.x];
#pragma unroll
    for (int i = 0; i < PG_NBINS; ++i) acc[i] = 0.0;
    auto add = [&](uint32_t a, uint32_t b, double val) { acc[tri_local(a < b ? a : b, a < b ? b : a)] += val; };
    const bool fb = dc.fwd_fallback[c] != 0;
    const bool reform = fb && c >= C / 2;
    if (reform) {
        // (see bins_unit) re-formed from the stored backward column, which carries data in rows and lanes below H
        const double unif = 1.0 / ((double)H * (double)H);
        const double* col = dc.fwd + (size_t)c * dc.col_stride;
        for (uint32_t i = 0; i < H; ++i) {
            const uint32_t a = al[i];
            for (uint32_t jj = 0; jj < H; ++jj) {
                const uint32_t b = al[jj];
                if (a < nl && b < nl) add(a, b, col[((size_t)(i >> 1) * HP + jj) * 2 + (i & 1u)] * unif);
            }
        }
    }
    {
        // The column alleles of the (up to 32) paths: two 16-byte loads, then bytes out of registers.  Entries are fetched eight
        // at a time whatever the alleles say — indices clamped to what was written (entries at and above DevContig::live never
        // were), loop bounds wave-uniform — so that the loads of a run are in flight together; a thread alone with its
        // column's forty-odd entries, one load at a time, spent nine tenths of its cycles waiting.
        const uint32_t nq = reform ? 0u : (nl + 1u) >> 1;
        const uint32_t nq_w = (uint32_t)__builtin_amdgcn_readfirstlane(wave_max_i32((int)nq));
        const v2f64* base = (const v2f64*)dc.part + (size_t)c * (dc.part_slots >> 1) * T;
        const uint4 aw0 = ((const uint4*)al)[0], aw1 = HP > 16u ? ((const uint4*)al)[1] : uint4{0, 0, 0, 0};
        const uint32_t alw[8] = {aw0.x, aw0.y, aw0.z, aw0.w, aw1.x, aw1.y, aw1.z, aw1.w};
        const uint32_t jn = HP == 32u ? dc.live : HP;
        for (uint32_t q = 0; q < nq_w; ++q) {
            const bool mine = q < nq;
            const uint32_t ra0 = 2u * q, ra1 = 2u * q + 1u;
            for (uint32_t w = 0; w * HP < T; ++w) {
                const v2f64* bw = base + (size_t)q * T + (size_t)w * HP;
                static_for<0, 4>([&](auto jb) __attribute__((always_inline)) {
                    constexpr int j0 = decltype(jb)::value * 8;
                    // (threads whose column has no slot pair q fetch nothing: re-reading their first pair instead, as a
                    // way to keep the wave together, doubled the kernel's HBM reads — 23 GB for 12 — at 4.9 TB/s)
                    if ((uint32_t)j0 < jn && mine) {
                        v2f64 pv[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) pv[k] = bw[(uint32_t)(j0 + k) < jn ? (uint32_t)(j0 + k) : jn - 1u];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const uint32_t bcol = (alw[(j0 + k) >> 2] >> (8 * ((j0 + k) & 3))) & 0xFFu;
                            if ((uint32_t)(j0 + k) < jn && bcol < nl) {   // (not a phantom path)
                                add(ra0, bcol, pv[k].x);
                                if (ra1 < nl) add(ra1, bcol, pv[k].y);
                            }
                        }
                    }
                });
            }
        }
    }
    if (!active) return;
    const double scale = 1.0 / ((fb ? 1.0 : dc.fscale[c]) * dc.bscale[c]);
    int xexp = -((fb ? 0 : PG_BIAS_F) + PG_BIAS_B);
    if (c + 1 < C)
        xexp += *(const int32_t*)((direct ? dc.vrec + (size_t)dc.col_variant[c + 1] * dc.RB : dc.colrec + (size_t)(c + 1) * dc.RB) + PG_REC_EXP);
    if (dc.tri && !reform) xexp += 1;
    const uint16_t* ls = (const uint16_t*)(rec + PG_REC_LOCAL_SLOT);
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0;
    const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
    const unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
    for (uint32_t la = 0; la < nl; ++la)
        for (uint32_t lb = la; lb < nl; ++lb) {
            const uint32_t sa = ls[la], sb = ls[lb];
            const uint64_t idx = dc.geno_off[v] + (uint64_t)sa * A - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
            const uint32_t pi = tri_n(la, lb, pn);
            const double pm = fb ? 0.5 : ((const double*)vp)[pi];
            const int pe = fb ? 1 : ((const int*)(vp + (size_t)NP * 8u))[pi];
            store_bin(dc.lik, dc.lik_exp, idx, acc[tri_local(la, lb)] * scale, pm, pe, xexp);
        }
}

// ------------------------------------------------------------------------------------------
//  k_bins_lean2 : the bins of chains on k_sweep_lean2 (two local alleles at most; the four class sums of a column
//  arrive finished, part[c][2 * (row allele) + (column allele)]) — one THREAD per column.  The rare column whose bins
//  have to be re-formed from the stored backward column (forward fall-back, see k_bins) is walked by its thread alone.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bins_lean2(const DevContig* __restrict__ contigs) {
    const DevContig& dc = contigs[blockIdx.y];
    const uint32_t C = *dc.n_cols;
    const bool cls = dc.cls4 != 0u;   // class sums from the general kernel's single-wave configurations: full columns, no halving
    if (!((dc.tri == 2u && C >= 2u) || cls) || dc.split) return;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= C) return;
    const bool fb = dc.fwd_fallback[c] != 0;
    const bool reform = fb && c >= C / 2;
    const unsigned char* rec = dc.vrec + (size_t)dc.col_variant[c] * dc.RB;
    const uint32_t v = *(const uint32_t*)(rec + PG_REC_VARIANT);
    const uint32_t nl = rec[PG_REC_NLOCAL];
    const uint16_t* ls = (const uint16_t*)(rec + PG_REC_LOCAL_SLOT);
    double b00 = 0.0, b01 = 0.0, b11 = 0.0;
    if (!reform) {
        const double* p4 = dc.part + (size_t)c * 4u;
        b00 = 0.0 + p4[0];
        if (nl > 1u) { b01 = (0.0 + p4[1]) + p4[2]; b11 = 0.0 + p4[3]; }
    } else {
        // the forward column fell back to uniform after this column's partials had been formed from the all-zero
        // column: alpha_hat * fsum = 1 / H^2 for every state, times the stored backward column (upper triangle, the
        // diagonal halved — or, for class-sum chains of the general kernel, the full column in row-pair layout)
        const uint32_t H = dc.H, HP = dc.HP;
        const unsigned char* al = rec + PG_REC_ALLELES;
        const double* col = dc.fwd + (size_t)c * dc.col_stride;
        if (cls) {
            for (uint32_t i = 0; i < H; ++i)
                for (uint32_t j = 0; j < H; ++j) {
                    const double val = col[((size_t)(i >> 1) * HP + j) * 2 + (i & 1u)];
                    const uint32_t a = al[i], b = al[j];
                    if (a == b) { if (a == 0u) b00 += val; else b11 += val; }
                    else b01 += val;
                }
        } else {
            for (uint32_t i = 0; i < H; ++i)
                for (uint32_t j = i; j < H; ++j) {
                    const double val = col[(size_t)tri_unit_of(i >> 1, j) * 2 + (i & 1u)];
                    const double both = 2.0 * val;  // (i, j) and (j, i); the halved diagonal once, doubled
                    const uint32_t a = al[i], b = al[j];
                    if (a == b) { if (a == 0u) b00 += both; else b11 += both; }
                    else b01 += both;
                }
        }
        const double unif = 1.0 / ((double)H * (double)H);
        b00 *= unif; b01 *= unif; b11 *= unif;
    }
    const double scale = 1.0 / ((fb ? 1.0 : dc.fscale[c]) * dc.bscale[c]);
    int xexp = -((fb ? 0 : PG_BIAS_F) + PG_BIAS_B) + ((reform || cls) ? 0 : 1);  // (+1: the partials of triangle chains are sums over the stored half, see k_bins)
    if (c + 1 < C) xexp += *(const int32_t*)(dc.vrec + (size_t)dc.col_variant[c + 1] * dc.RB + PG_REC_EXP);
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0;
    const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
    const unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
    for (uint32_t la = 0; la < nl; ++la)
        for (uint32_t lb = la; lb < nl; ++lb) {
            const uint32_t sa = ls[la], sb = ls[lb];
            const uint64_t idx = dc.geno_off[v] + (uint64_t)sa * A - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
            const uint32_t pi = tri_n(la, lb, pn);
            const double pm = fb ? 0.5 : ((const double*)vp)[pi];
            const int pe = fb ? 1 : ((const int*)(vp + (size_t)NP * 8u))[pi];
            const double bin = la == lb ? (la == 0 ? b00 : b11) : b01;
            store_bin(dc.lik, dc.lik_exp, idx, bin * scale, pm, pe, xexp);
        }
}

// ------------------------------------------------------------------------------------------
//  k_post : chunked mode.  Posteriors of the columns of one finished chunk straight from the two
//  stored columns (one wave per column):  L_v({a,b}) = sum_(i,j) alpha'_c(i,j) * beta'_c(i,j) / (m_f m_b)
//  (reference src/hmm.cpp:364-368).  For c >= mid alpha' is in the chunk scratch and beta' in the
//  main slots, for c < mid the other way round.  Runs on the CUs the few chains leave idle: it is
//  launched with PG_POST_PLACEMENT_LDS bytes of (unused) dynamic LDS, more than fits next to a sweep
//  workgroup, so its 16-wave blocks never share a CU — and issue slots — with the latency-critical
//  recursion (side by side the chunk sweeps ran 35 % slower).
// ------------------------------------------------------------------------------------------
#define PG_POST_WAVES 16
#define PG_POST_PLACEMENT_LDS (152 * 1024)
// the bins of column c from its two stored columns A = alpha', B = beta' (full columns, row-pair layout) by one wave;
// s_bins = the wave's row of narrow bins
#define PG_WIDE_LDS_N 16   // wide columns with at most this many local alleles gather their raw bins in LDS (136 doubles per wave)
#define PG_WIDE_LDS_BINS (PG_WIDE_LDS_N * (PG_WIDE_LDS_N + 1) / 2)
DEVI void post_ab(const DevContig& dc, uint32_t C, uint32_t c, const double* A, const double* B, uint32_t lane,
                  double (&s_bins_row)[PG_AMAX * (PG_AMAX + 1) / 2], double* s_wide = nullptr, uint32_t tri_ab = 0u);
// one column (index idx inside the chunk: forward role first) by one wave
// The same column for LEAN chains (64 paths, at most two local alleles, the variant's record read in place — every chain of the
// whole-genome job).  post_ab spends a column as eight rounds of {issue eight 1 KB loads, wait, add up}: 34–46 us per column
// and wave, 1.5 TB/s on 192 CUs, and the chunk sweeps beside it wait for its scratch buffers (round 6: k_post bounds phase 2
// of genome24_h64, not the chains).  Here the loads are double-buffered — the next four row pairs are in flight while the four
// before them are added up —, the row alleles come from the column's compact record (no load behind the variant id) and
// everything the bins need at the end (variant id, record, slots, offsets, pair factors) is fetched before the first column
// load is waited for.  Same products, same order of additions, same bins as post_ab.
DEVI void post_lean64(const DevContig& dc, uint32_t C, uint32_t c, const double* A, const double* B, uint32_t lane,
                      double (&s_bins_row)[PG_AMAX * (PG_AMAX + 1) / 2]) {
    constexpr uint32_t HP = 64;
    // (the lane number behind an opaque move: a lane's 64 triangle weights and 32 load offsets are the same for every column, and
    //  the compiler otherwise keeps them across k_post's column loop — hundreds of registers, all spilled)
    uint32_t ln = lane;
    asm volatile("" : "+v"(ln));
    const GAS char* A2 = (const GAS char*)A;   // (uniform: the wave's column)
    const GAS char* B2 = (const GAS char*)B;
    v2f64 av[2][4], bv[2][4];
    // upper triangle only (see post_ab): a lane below the diagonal of row pair ip re-reads its own diagonal unit — in cache, weight 0 —
    // instead of being masked off (a masked load is a branch of its own to this compiler, with a full wait behind it)
    const uint32_t ipmax = ln >> 1;
    auto fetch = [&](auto btc) __attribute__((always_inline)) {
        constexpr int bt = decltype(btc)::value, buf = bt & 1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t ip = (uint32_t)(4 * bt + u), ipc = ip < ipmax ? ip : ipmax;
            const uint32_t off = (ipc * HP + ln) * 16u;
            if constexpr (kNtPost) { av[buf][u] = __builtin_nontemporal_load((gcdouble2*)(A2 + off)); bv[buf][u] = __builtin_nontemporal_load((gcdouble2*)(B2 + off)); }
            else { av[buf][u] = *(gcdouble2*)(A2 + off); bv[buf][u] = *(gcdouble2*)(B2 + off); }
        }
    };
    fetch(std::integral_constant<int, 0>{});
    const unsigned long long bits = (unsigned long long)__double_as_longlong(dc.frec[(size_t)c * 8u + 7u]);   // bit p: path p carries local allele 1
    const uint32_t cv = dc.col_variant[c];
    const uint32_t cvn = dc.col_variant[c + 1 < C ? c + 1 : c];
    const bool fb = dc.fwd_fallback[c] != 0;
    const double fscl = dc.fscale[c], bscl = dc.bscale[c];
    fetch(std::integral_constant<int, 1>{});
    if (lane < PG_AMAX * (PG_AMAX + 1) / 2) s_bins_row[lane] = 0.0;
    const unsigned char* rec = dc.vrec + (size_t)cv * dc.RB;
    const uint32_t nl = rec[PG_REC_NLOCAL];
    const uint16_t* ls = (const uint16_t*)(rec + PG_REC_LOCAL_SLOT);
    const uint32_t la = nl ? lane / nl : 0u, lb = nl ? lane % nl : 0u;
    const bool binlane = lane < nl * nl && la <= lb;   // the lane that finishes bin (la, lb)
    const uint32_t sa = binlane ? ls[la] : 0u, sb = binlane ? ls[lb] : 0u;
    const uint32_t Av = dc.allele_off[cv + 1] - dc.allele_off[cv];
    const uint64_t g0 = dc.geno_off[cv];
    const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
    const unsigned char* vp = dc.vpair + (size_t)cv * (NP * 12u);
    const uint32_t pi = binlane ? tri_n(la, lb, pn) : 0u;
    const double pm0 = ((const double*)vp)[pi];
    const int pe0 = ((const int*)(vp + (size_t)NP * 8u))[pi];
    const int nexp = *(const int32_t*)(dc.vrec + (size_t)cvn * dc.RB + PG_REC_EXP);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits), bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bits >> 32));
    double acc0 = 0.0, acc1 = 0.0;
    static_for<0, 8>([&](auto btc) __attribute__((always_inline)) {
        constexpr int bt = decltype(btc)::value, buf = bt & 1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t ip = (uint32_t)(4 * bt + u);
            const double w0 = ln > 2u * ip ? 2.0 : (ln == 2u * ip ? 1.0 : 0.0);
            const double w1 = ln > 2u * ip + 1u ? 2.0 : (ln == 2u * ip + 1u ? 1.0 : 0.0);
            const double p0 = av[buf][u].x * bv[buf][u].x * w0, p1 = av[buf][u].y * bv[buf][u].y * w1;
            const uint32_t word = ip < 16u ? blo : bhi;
            const bool r0 = (word >> ((2u * ip) & 31u)) & 1u, r1 = (word >> ((2u * ip + 1u) & 31u)) & 1u;   // (uniform) row alleles
            acc0 = fma(p0, r0 ? 0.0 : 1.0, acc0); acc1 = fma(p0, r0 ? 1.0 : 0.0, acc1);
            acc0 = fma(p1, r1 ? 0.0 : 1.0, acc0); acc1 = fma(p1, r1 ? 1.0 : 0.0, acc1);
        }
        asm volatile("" : "+v"(acc0), "+v"(acc1));   // (the sums are formed HERE: left alone, the additions of all eight rounds end up behind the last one, their products spilled)
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise hoists all 64 loads to the top: 600 spilled registers)
        if constexpr (bt + 2 < 8) fetch(std::integral_constant<int, bt + 2>{});   // (into the buffer just added up)
        __builtin_amdgcn_sched_barrier(0);
    });
    wave_sync_lds();   // (the zeros of s_bins_row)
    const uint32_t b = (uint32_t)((bits >> lane) & 1ull);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if ((uint32_t)a < nl) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                if ((uint32_t)bb < nl) {
                    const double tot = wave_sum(b == (uint32_t)bb ? (a ? acc1 : acc0) : 0.0);
                    if (lane == 0) s_bins_row[tri_local(a < bb ? a : bb, a < bb ? bb : a)] += tot;
                }
            }
        }
    }
    wave_sync_lds();
    if (binlane) {
        const double scale = 1.0 / ((fb ? 1.0 : fscl) * bscl);
        int xexp = -((fb ? 0 : PG_BIAS_F) + PG_BIAS_B);
        if (c + 1 < C) xexp += nexp;
        const uint64_t gi = g0 + (uint64_t)sa * Av - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
        store_bin(dc.lik, dc.lik_exp, gi, s_bins_row[tri_local(la, lb)] * scale, fb ? 0.5 : pm0, fb ? 1 : pe0, xexp);
    }
}

DEVI void post_column(const DevContig& dc, uint32_t chunk, uint32_t idx, uint32_t wave, uint32_t lane,
                      double (&s_bins)[PG_POST_WAVES][PG_AMAX * (PG_AMAX + 1) / 2]) {
    const uint32_t C = *dc.n_cols;
    const uint32_t K = dc.chunk_cols, HP = dc.HP;
    if (C == 0 || idx >= 2u * K) return;
    const uint32_t mid = C / 2;
    const size_t colsz = (size_t)HP * HP;
    const double* scr = dc.scratch + (size_t)(PG_SCR_BUF(chunk) * 2u) * K * colsz;
    uint32_t c;
    const double *A, *B;  // alpha', beta'
    if (idx < K) {        // forward role: columns mid + chunk*K ...
        const unsigned long long cc = (unsigned long long)mid + (unsigned long long)chunk * K + idx;
        if (cc >= C) return;
        c = (uint32_t)cc;
        A = scr + (size_t)idx * colsz;
        B = dc.fwd + (size_t)c * colsz;
    } else {              // backward role: columns mid-1-chunk*K downwards; slot = column - bot
        const long long top = (long long)mid - 1 - (long long)chunk * K;
        if (top < 0) return;
        const long long bot = top - (long long)K + 1 > 0 ? top - (long long)K + 1 : 0;
        const long long cc = bot + (long long)(idx - K);
        if (cc > top) return;
        c = (uint32_t)cc;
        B = scr + (size_t)K * colsz + (size_t)(idx - K) * colsz;
        A = dc.fwd + (size_t)c * colsz;
    }
    if (dc.lean == 1u && dc.tri == 0u && !kPostGeneric) post_lean64(dc, C, c, A, B, lane, s_bins[wave]);
    else post_ab(dc, C, c, A, B, lane, s_bins[wave]);
}
// s_wide (optional): PG_WIDE_LDS_BINS doubles of LDS of this wave's own — the raw bins of a wide column with at most
// PG_WIDE_LDS_N local alleles are added up there instead of in `lik` (a dependent global read-modify-write per (row allele,
// column allele) pair by one lane: ~45 of them in series cost a 16-path wide column more than everything else together)
// tri_ab (64 paths): bit 0 / bit 1 = A / B is a column stored as its upper triangle (DevContig::tri: units of tri_unit_of, the
// diagonal halved) — the partner of a wide column of a triangle chain (k_bins_wide); this function reads the upper triangle only anyway
DEVI void post_ab(const DevContig& dc, uint32_t C, uint32_t c, const double* A, const double* B, uint32_t lane,
                  double (&s_bins_row)[PG_AMAX * (PG_AMAX + 1) / 2], double* s_wide, uint32_t tri_ab) {
    const uint32_t HP = dc.HP;
    const bool direct = compact_records_only(dc, C);  // (no column-order copy of the records: the variant's own)
    // (split chains, pg_split.h: only their WIDE columns come here — variant and allele count from the index's bins record, the
    //  sixteen raw local alleles from the head of the column's sample record, the wide entry from the index's record)
    const bool sp = dc.split != 0u;
    const unsigned char* rec = sp ? nullptr : (direct ? dc.vrec + (size_t)dc.col_variant[c] * dc.RB : dc.colrec + (size_t)c * dc.RB);
    const IxBin* ixb = sp ? (const IxBin*)(dc.ix_bin + (size_t)c * PG_IXBIN_BYTES) : nullptr;
    const uint32_t v = sp ? ixb->v : *(const uint32_t*)(rec + PG_REC_VARIANT);
    const uint32_t nl = sp ? ixb->nl : rec[PG_REC_NLOCAL];
    const unsigned char* al = sp ? (const unsigned char*)dc.frec + (size_t)c * PG_SREC2_BYTES : rec + PG_REC_ALLELES;
    if (lane < PG_AMAX * (PG_AMAX + 1) / 2) s_bins_row[lane] = 0.0;
    wave_sync_lds();   // (LDS only: the header loads above stay in flight under the first column loads below)

    // element e of a column = row pair e / HP, column e % HP (16 bytes: rows 2*(e/HP), +1); lanes take
    // consecutive elements, so every load is a coalesced 1 KB per wave.  Per lane the column is fixed
    // when HP <= 64 (two columns for HP = 128: handled as two passes), so a lane accumulates by ROW
    // allele only and the split by column allele happens once at the end.
    const v2f64* A2 = (const v2f64*)A;
    const v2f64* B2 = (const v2f64*)B;
    const uint32_t npass = HP > 64 ? HP / 64 : 1;
    const uint32_t a0v = dc.allele_off[v], Av = dc.allele_off[v + 1] - a0v;
    // stored columns are (true value) * m * 2^bias (see k_bins); a flagged forward column is the uniform
    // column itself: absolute value, no emission, no scale, no bias
    const bool fb = dc.fwd_fallback[c] != 0;
    const double scale = 1.0 / ((fb ? 1.0 : dc.fscale[c]) * dc.bscale[c]);
    int xexp = -((fb ? 0 : PG_BIAS_F) + PG_BIAS_B);
    if (c + 1 < C)
        xexp += sp ? *(const int32_t*)((const unsigned char*)dc.frec + (size_t)(c + 1) * PG_SREC2_BYTES + 120u)
                   : *(const int32_t*)((direct ? dc.vrec + (size_t)dc.col_variant[c + 1] * dc.RB : dc.colrec + (size_t)(c + 1) * dc.RB) + PG_REC_EXP);
    // Wide columns (more than PG_AMAX alleles on the selected paths) take one sweep over the two
    // columns per block of PG_AMAX row alleles and add their bins straight into lik (zeroed at the
    // start of the run; this wave is the only writer of the variant's bins).
    const bool widec = nl > PG_AMAX;
    const uint16_t* wslots = nullptr;
    const unsigned char* went = nullptr;
    const uint32_t WS = nl + 1u;  // row stride of a wide entry's tables
    if (widec) {
        went = dc.wide + (size_t)(sp ? *(const uint32_t*)(dc.ix_rec + (size_t)c * PG_IXREC_BYTES + 4u) : *(const uint32_t*)(rec + PG_REC_WIDE_IDX)) * 16u;
        wslots = (const uint16_t*)(went + PG_WIDE_OFF_SLOT(WS));
    }
    const bool in_lds = widec && s_wide != nullptr && nl <= (uint32_t)PG_WIDE_LDS_N;
    if (in_lds) {
        for (uint32_t q = lane; q < nl * (nl + 1u) / 2u; q += 64u) s_wide[q] = 0.0;
        wave_sync_lds();
    }
    for (uint32_t abase = 0; abase < nl; abase += PG_AMAX)
    for (uint32_t ps = 0; ps < npass; ++ps) {
        double acc[PG_AMAX];
#pragma unroll
        for (int a = 0; a < PG_AMAX; ++a) acc[a] = 0.0;
        uint32_t j, ip0, ipstep;
        if (HP >= 64) { j = ps * 64 + lane; ip0 = 0; ipstep = 1; }
        else { j = lane % HP; ip0 = lane / HP; ipstep = 64 / HP; }
        // four row pairs (8 x 16-byte loads per lane) in flight at a time: this kernel has to stream
        // 128 KB per column at HBM rate to keep up with the chunk sweeps
        constexpr int UN = 4;
        for (uint32_t ipb = ip0; ipb < HP / 2; ipb += ipstep * UN) {
            v2f64 av[UN], bv[UN];
#pragma unroll
            // Columns are symmetric (to rounding), and a genotype bin takes (i,j) and (j,i) alike: only the
            // upper triangle j >= i is read — a lane skips the row pairs below its column — and off-diagonal
            // states count twice.  Halves what this kernel pulls out of HBM next to the running chains.
            for (int u = 0; u < UN; ++u) {
                const uint32_t ip = ipb + u * ipstep;
                av[u] = v2f64{0.0, 0.0}; bv[u] = v2f64{0.0, 0.0};
                if (ip < HP / 2 && j >= 2u * ip) {
                    const size_t e = (size_t)ip * HP + j;
                    if (tri_ab == 0u) { av[u] = A2[e]; bv[u] = B2[e]; }
                    else {
                        const size_t et = tri_unit_of(ip, j);
                        av[u] = A2[(tri_ab & 1u) ? et : e]; bv[u] = B2[(tri_ab & 2u) ? et : e];
                        const double d0 = j == 2u * ip ? 2.0 : 1.0, d1 = j == 2u * ip + 1u ? 2.0 : 1.0;   // (the stored diagonal is halved)
                        if (tri_ab & 1u) { av[u].x *= d0; av[u].y *= d1; }
                        if (tri_ab & 2u) { bv[u].x *= d0; bv[u].y *= d1; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t ip = ipb + u * ipstep;
                if (ip < HP / 2) {
                    // P' * beta': both factors are bounded below relative to their column sums (>= q^2) and
                    // carry their biases, so the products are normal fp64 numbers.  Weights: 2 above the
                    // diagonal, 1 on it, 0 below (those lanes loaded nothing).
                    const double w0 = j > 2u * ip ? 2.0 : (j == 2u * ip ? 1.0 : 0.0);
                    const double w1 = j > 2u * ip + 1u ? 2.0 : (j == 2u * ip + 1u ? 1.0 : 0.0);
                    const double p0 = av[u].x * bv[u].x * w0, p1 = av[u].y * bv[u].y * w1;
                    const uint32_t a0 = (uint32_t)al[2 * ip] - abase, a1 = (uint32_t)al[2 * ip + 1] - abase;
#pragma unroll
                    for (int a = 0; a < PG_AMAX; ++a) {
                        acc[a] = fma(p0, a0 == (uint32_t)a ? 1.0 : 0.0, acc[a]);
                        acc[a] = fma(p1, a1 == (uint32_t)a ? 1.0 : 0.0, acc[a]);
                    }
                }
            }
        }
        const uint32_t b = al[j];
        if (!widec) {
#pragma unroll
            for (int a = 0; a < PG_AMAX; ++a) {
                if ((uint32_t)a < nl) {
#pragma unroll
                    for (int bb = 0; bb < PG_AMAX; ++bb) {
                        if ((uint32_t)bb < nl) {
                            const double tot = wave_sum(b == (uint32_t)bb ? acc[a] : 0.0);
                            if (lane == 0) {
                                const uint32_t ra = (uint32_t)a, cb = (uint32_t)bb;
                                s_bins_row[tri_local(ra < cb ? ra : cb, ra < cb ? cb : ra)] += tot;
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < PG_AMAX; ++a) {
                const uint32_t ra = abase + (uint32_t)a;
                if (ra < nl) {
                    for (uint32_t cb = 0; cb < nl; ++cb) {
                        const double tot = wave_sum(b == cb ? acc[a] : 0.0);
                        if (lane == 0 && tot != 0.0) {
                            const uint32_t lo2 = ra < cb ? ra : cb, hi2 = ra < cb ? cb : ra;
                            if (in_lds) s_wide[tri_n(lo2, hi2, nl)] += tot;   // (la <= lb: the order decode_pair enumerates)
                            else {
                                const uint32_t sa = wslots[lo2], sb = wslots[hi2];
                                const uint64_t gi = dc.geno_off[v] + (uint64_t)sa * Av - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
                                dc.lik[gi] += tot;  // raw sum; finished below
                            }
                        }
                    }
                }
            }
        }
    }
    wave_sync_lds();
    if (!widec) {
        const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
        const unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
        if (lane < nl * nl) {
            const uint16_t* ls = (const uint16_t*)(rec + PG_REC_LOCAL_SLOT);
            const uint32_t la = lane / nl, lb = lane % nl;
            if (la <= lb) {
                const uint32_t sa = ls[la], sb = ls[lb];
                const uint64_t gi = dc.geno_off[v] + (uint64_t)sa * Av - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
                const uint32_t pi = tri_n(la, lb, pn);
                const double pm = fb ? 0.5 : ((const double*)vp)[pi];
                const int pe = fb ? 1 : ((const int*)(vp + (size_t)NP * 8u))[pi];
                store_bin(dc.lik, dc.lik_exp, gi, s_bins_row[tri_local(la, lb)] * scale, pm, pe, xexp);
            }
        }
    } else {
        // wide column: the raw sums of its bins sit in lik (lane 0 added them up above); finish every
        // local pair la <= lb, one per lane
        if (in_lds) wave_sync_lds();
        else { __threadfence(); wave_sync(); }
        const double* Pm = (const double*)(went + PG_WIDE_OFF_PM(WS));
        const int* Pe = (const int*)(went + PG_WIDE_OFF_PE(WS));
        const uint32_t npairs = nl * (nl + 1u) / 2u;
        for (uint32_t q = lane; q < npairs; q += 64) {
            uint32_t la, lb;
            decode_pair(q, nl, la, lb);
            const uint32_t sa = wslots[la], sb = wslots[lb];  // ascending with the local index
            const uint64_t gi = dc.geno_off[v] + (uint64_t)sa * Av - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
            const double pm = fb ? 0.5 : Pm[la * WS + lb];
            const int pe = fb ? 1 : Pe[la * WS + lb];
            store_bin(dc.lik, dc.lik_exp, gi, (in_lds ? s_wide[q] : dc.lik[gi]) * scale, pm, pe, xexp);
        }
        if (in_lds) wave_sync_lds();   // (the wave's row is reused by its next wide column)
    }
}

__global__ __launch_bounds__(64 * PG_POST_WAVES) void k_post(const DevContig* __restrict__ contigs, uint32_t chunk, uint32_t n_contigs) {
    __shared__ double s_bins[PG_POST_WAVES][PG_AMAX * (PG_AMAX + 1) / 2];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Grid = (blocks, 1): as many blocks as the chains leave CUs idle (the launcher), shared out over the chains that HAVE
    // columns in this chunk — the chains of a genome end one after the other, and with a fixed share per chain (8 of the
    // 208 idle CUs each, until round 6) the long chains' k_post ran at the sweep's own rate to the end (7 instead of 8 blocks
    // per chain: phase 2 136 instead of 128 ms) while the blocks of chains that had ended left at once.  Lane c looks at chain
    // c (at most 64 chains: more of them run fused, or this kernel with the y dimension as before).
    uint32_t ci = blockIdx.y, sub = blockIdx.x, share = gridDim.x;
    if (n_contigs) {
        bool active = false;
        if (lane < n_contigs) {
            const DevContig& d = contigs[lane];
            const uint32_t C = *d.n_cols, K = d.chunk_cols, mid = C / 2;
            // (forward role: columns mid + chunk K ...; backward role: mid - 1 - chunk K downwards — post_column)
            active = C > 0 && ((unsigned long long)mid + (unsigned long long)chunk * K < C || (unsigned long long)chunk * K < mid);
        }
        const unsigned long long am = __ballot(active);
        const uint32_t na = (uint32_t)__popcll(am);
        if (na == 0u) return;
        const uint32_t rank = blockIdx.x % na;
        sub = blockIdx.x / na;
        share = gridDim.x / na + (rank < gridDim.x % na ? 1u : 0u);
        unsigned long long m = am;   // the rank-th active chain (the first ones — the long chromosomes of a genome — get the odd blocks)
        for (uint32_t r = 0; r < rank; ++r) m &= m - 1ull;
        ci = (uint32_t)__builtin_ctzll(m);
    }
    const DevContig& dc = contigs[ci];
    for (uint32_t idx = sub * PG_POST_WAVES + wave; idx < 2u * dc.chunk_cols; idx += share * PG_POST_WAVES) {
        post_column(dc, chunk, (uint32_t)__builtin_amdgcn_readfirstlane((int)idx), wave, lane, s_bins);
        wave_sync_lds();   // (the wave's s_bins row is reused by its next column; its global stores need no wait)
    }
}

// ------------------------------------------------------------------------------------------
//  k_post_loop : k_post for the PERSISTENT chunked phase 2 (see chunk_spin above) — ONE launch; a block walks the chunks of its
//  chain in order: waits until both roles of k_sweep_lean<4> have published the chunk, forms its share of the posteriors, hands
//  the scratch buffer back.  Same grid shape and placement LDS as k_post (the blocks sit on the CUs the chains leave idle, for
//  the whole phase).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * PG_POST_WAVES) void k_post_loop(const DevContig* __restrict__ contigs) {
    __shared__ double s_bins[PG_POST_WAVES][PG_AMAX * (PG_AMAX + 1) / 2];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n = gridDim.y;   // chains of the job (<= 64: pg_shim.cpp)
    // Work items = columns (2 K slots per chunk, forward role first), handed out in order through the chain's
    // sync[PG_SYNC_NEXT], one per WAVE and turn: a wave takes what it can do — a block that never gets a CU takes nothing, and
    // nobody waits for it.  (A column of chunk q + PG_SCRATCH_BUFS is only handed out once every column of chunk q has been
    // TAKEN, by a running wave; the sweep publishes that chunk when they are all done.)  No block-wide barrier: the waves run on
    // their own.  A wave starts on the chain of its block; when that chain has handed out its last column the wave moves to
    // one of the chains that still have columns to come, drawn with the weight of what is left — the chains of a genome end
    // one after the other, the longest is the job's wall time, and the waves of the ended ones are its reserve.
    // (Measured dead end, profiles/r06_persist.txt: every wave looking at every chain before every column — three relaxed
    //  agent-scope loads per chain — 645 ms instead of 129: a line that thousands of waves poll serves ~45 M accesses a second,
    //  and the sweep's own flag traffic queues behind them.  Hence: no look before the take, polls only while a taken column is
    //  not yet published, the flags the sweep writes on a line of their own.)
    const uint32_t twoK = 2u * contigs[0].chunk_cols;
    uint32_t home = blockIdx.y;
    if (blockIdx.x == 0 && wave == 0) {
        // The chain's WATCHER: the one wave that reads the flags the sweep writes; it republishes "chunks with both roles
        // stored" on a line of its own (sync[PG_SYNC_READY]), and that is what every other wave polls.  (Thousands of waves
        // polling the sweep's own line made the sweep's publishing store queue behind them: 138 ms instead of 129.)
        const DevContig& dc = contigs[home];
        const uint32_t C = (dc.lean == 1u && dc.tri == 0u) ? *dc.n_cols : 0u;
        if (!C) return;
        const uint32_t K = dc.chunk_cols, mid = C / 2;
        const uint32_t nf = (C - mid + K - 1u) / K, nb = (mid + K - 1u) / K, nq = nf > nb ? nf : nb;
        if (lane != 0) return;
        uint32_t told = 0;
        const unsigned long long t0 = wall_clock64();
        unsigned long long t_last = t0;
        while (told < nq) {
            const uint32_t fpub = __hip_atomic_load(dc.sync + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t bpub = __hip_atomic_load(dc.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t fe = fpub >= nf ? nq : fpub, be = bpub >= nb ? nq : bpub;
            const uint32_t ready = fe < be ? fe : be;
            if (ready > told) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(dc.sync + PG_SYNC_READY, ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                told = ready;
                t_last = wall_clock64();
            } else {
                if (wall_clock64() - t_last > PG_SYNC_TIMEOUT_TICKS) { atomicOr(dc.err, PG_DEVERR_SYNC_TIMEOUT); return; }
                if (__hip_atomic_load(dc.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & PG_DEVERR_SYNC_TIMEOUT) return;
                __builtin_amdgcn_s_sleep(64);
            }
        }
        return;
    }
    for (;;) {
        const DevContig& dc = contigs[home];
        const uint32_t C = (dc.lean == 1u && dc.tri == 0u) ? *dc.n_cols : 0u;   // (what k_sweep_lean<4, 16, false> takes)
        if (C) {
            const uint32_t K = dc.chunk_cols, mid = C / 2;
            const uint32_t nf = (C - mid + K - 1u) / K, nb = (mid + K - 1u) / K;   // chunks of the forward / backward role
            const uint32_t nq = nf > nb ? nf : nb;
            auto take = [&]() -> uint32_t {
                uint32_t it = 0;
                if (lane == 0) it = __hip_atomic_fetch_add(dc.sync + PG_SYNC_NEXT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return it;   // (lane 0's value; made uniform where it is used)
            };
            uint32_t q_ready = 0;   // chunks of this chain the wave knows to be published
            uint32_t item = take();
            for (;;) {
                item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
                const uint32_t q = item / twoK, idx = item % twoK;
                if (q >= nq) break;
                const uint32_t nxt = take();   // (the next column's number is on its way while this one is worked on)
                if (q >= q_ready) {
                    uint32_t ok = 1u;
                    if (lane == 0) ok = chunk_spin(dc.sync + PG_SYNC_READY, q + 1u, dc.err) ? 1u : 0u;   // (the watcher's line)
                    if (!__builtin_amdgcn_readfirstlane((int)ok)) return;   // (PG_DEVERR_SYNC_TIMEOUT is up: pg_job_run fails the job)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    q_ready = q + 1u;
                }
                if (!(kPersistExp & 1u)) { post_column(dc, q, idx, wave, lane, s_bins); wave_sync_lds(); }
                // (what the wave read of the buffer has arrived — it was used —: the slot may be overwritten)
                if (lane == 0) __hip_atomic_fetch_add(dc.sync + PG_SYNC_DONE + PG_SCR_BUF(q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                item = nxt;
            }
        }
        // the chain has handed out its last column: on to another one (lane c looks at chain c)
        uint32_t left = 0;
        if (lane < n) {
            const DevContig& d = contigs[lane];
            const uint32_t Cc = (d.lean == 1u && d.tri == 0u) ? *d.n_cols : 0u;
            if (Cc) {
                const uint32_t K = d.chunk_cols, mid = Cc / 2;
                const uint32_t nf = (Cc - mid + K - 1u) / K, nb = (mid + K - 1u) / K;
                const uint32_t total = (nf > nb ? nf : nb) * twoK;
                const uint32_t next = __hip_atomic_load(d.sync + PG_SYNC_NEXT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                left = next < total ? (total - next + 1023u) >> 10 : 0u;   // (in units of 1024 columns: the sum below stays in 32 bits)
            }
        }
        uint32_t incl = left;   // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, o); if (lane >= (uint32_t)o) incl += v; }
        const uint32_t total_left = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total_left == 0u) return;
        // a draw in [0, total_left) that differs from wave to wave
        uint32_t r = (blockIdx.y * gridDim.x + blockIdx.x) * PG_POST_WAVES + wave;
        r = (r * 2654435761u) ^ (home * 40503u);
        r = (uint32_t)(((unsigned long long)r * total_left) >> 32);
        const unsigned long long hit = __ballot(left > 0u && r < incl);
        home = (uint32_t)__builtin_ctzll(hit);   // the first chain whose running sum passes the draw
    }
}

// ------------------------------------------------------------------------------------------
//  k_bins_x / k_bins_wide : the bins of chains whose phase 2 ran on k_sweep_small16x (DevContig::smallx == 2).
//  k_bins_x — one THREAD per column: a column with at most two local alleles has its four class sums in part[c][4]
//  (DevContig::cls4's layout); one with three to five has its bins, finished inside the sweep, in its aux slot (fifteen
//  doubles in tri_local order); the rare column whose bins are re-formed
//  from the stored backward column (forward fall-back, see k_bins) is walked by its thread alone.  Same factors,
//  exponents and fall-back rule as bins_unit.  WIDE columns are left to k_bins_wide — one WAVE per wide column: the
//  column this role's phase 2 put into the aux slot times the stored partner column, as k_post does it (post_ab).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bins_x(const DevContig* __restrict__ contigs) {
    // (the bins of a column live in registers, every index static: with a private LDS row per thread — k_bins_thin's way —
    //  four blocks fit a CU, and this kernel is a chain of dependent loads: it needs the waves)
    const DevContig& dc = contigs[blockIdx.y];
    const uint32_t C = *dc.n_cols;
    if (!bins_x(dc, C)) return;
    if (blockIdx.x * 256u >= C) return;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= C) return;
    const uint32_t cv = dc.col_variant[c], cvn = c + 1 < C ? dc.col_variant[c + 1] : cv;
    const unsigned char* rec = dc.vrec + (size_t)cv * dc.RB;
    const uint4 hd = *(const uint4*)(rec + PG_REC_VARIANT);   // variant, exponent, nlocal | flags << 8, wide entry
    if ((hd.z >> 8) & PG_REC_FLAG_WIDE) return;               // k_bins_wide
    const uint32_t v = hd.x;
    const uint32_t nl = hd.z & 0xFFu;
    const uint4 lsw = *(const uint4*)(rec + PG_REC_LOCAL_SLOT);   // local allele -> allele slot (five u16), then the aux slot
    const int32_t xnext = c + 1 < C ? *(const int32_t*)(dc.vrec + (size_t)cvn * dc.RB + PG_REC_EXP) : 0;
    const bool fb = dc.fwd_fallback[c] != 0;
    const bool reform = fb && c >= C / 2;
    double acc[PG_NBINS];
#pragma unroll
    for (int i = 0; i < PG_NBINS; ++i) acc[i] = 0.0;
    if (reform) {
        // (see bins_unit) alpha_hat * fsum = 1 / H^2 for every state, times the stored backward column.  Rare: the bin of a
        // state is picked by fifteen selects
        const uint32_t H = dc.H, HP = dc.HP;
        const unsigned char* al = rec + PG_REC_ALLELES;
        const double unif = 1.0 / ((double)H * (double)H);
        const double* col = dc.fwd + (size_t)c * dc.col_stride;
        for (uint32_t i = 0; i < H; ++i) {
            const uint32_t a = al[i];
            for (uint32_t jj = 0; jj < H; ++jj) {
                const uint32_t b = al[jj];
                if (a < nl && b < nl) {
                    const uint32_t idx = tri_local(a < b ? a : b, a < b ? b : a);
                    const double val = col[((size_t)(i >> 1) * HP + jj) * 2 + (i & 1u)] * unif;
#pragma unroll
                    for (int q = 0; q < PG_NBINS; ++q) acc[q] += idx == (uint32_t)q ? val : 0.0;
                }
            }
        }
    } else if (nl <= 2u) {
        const v2f64* p4 = (const v2f64*)(dc.part + (size_t)c * 4u);
        const v2f64 p01 = p4[0], p23 = p4[1];
        acc[0] = 0.0 + p01.x;                                                        // tri_local(0, 0)
        if (nl > 1u) { acc[1] = (0.0 + p01.y) + p23.x; acc[PG_AMAX] = 0.0 + p23.y; }   // tri_local(0, 1), tri_local(1, 1)
    } else {
        // the column's bins arrive finished (tri_local order), 120 bytes in its aux slot
        const v2f64* e = (const v2f64*)(dc.aux + (size_t)lsw.w * 16u);
        v2f64 pv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pv[k] = e[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc[2 * k] = pv[k].x; if (2 * k + 1 < PG_NBINS) acc[2 * k + 1] = pv[k].y; }
    }
    const double scale = 1.0 / ((fb ? 1.0 : dc.fscale[c]) * dc.bscale[c]);
    const int xexp = -((fb ? 0 : PG_BIAS_F) + PG_BIAS_B) + xnext;
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0;
    const uint64_t g0 = dc.geno_off[v];
    const uint32_t pn = dc.pair_n, NP = (pn * (pn + 1) / 2 + 1u) & ~1u;
    const unsigned char* vp = dc.vpair + (size_t)v * (NP * 12u);
    const uint32_t lsv[PG_AMAX] = {lsw.x & 0xFFFFu, lsw.x >> 16, lsw.y & 0xFFFFu, lsw.y >> 16, lsw.z & 0xFFFFu};
    static_for<0, PG_AMAX>([&](auto lac) __attribute__((always_inline)) {
        constexpr int la = decltype(lac)::value;
        static_for<la, PG_AMAX>([&](auto lbc) __attribute__((always_inline)) {
            constexpr int lb = decltype(lbc)::value;
            if ((uint32_t)lb < nl) {
                const uint32_t sa = lsv[la], sb = lsv[lb];
                const uint64_t idx = g0 + (uint64_t)sa * A - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
                const uint32_t pi = tri_n((uint32_t)la, (uint32_t)lb, pn);
                const double pm = fb ? 0.5 : ((const double*)vp)[pi];
                const int pe = fb ? 1 : ((const int*)(vp + (size_t)NP * 8u))[pi];
                store_bin(dc.lik, dc.lik_exp, idx, acc[la * PG_AMAX - la * (la - 1) / 2 + (lb - la)] * scale, pm, pe, xexp);
            }
        });
    });
}

__global__ __launch_bounds__(256) void k_bins_wide(const DevContig* __restrict__ contigs) {
    __shared__ double s_bins[4][PG_AMAX * (PG_AMAX + 1) / 2];
    __shared__ double s_wide[4][PG_WIDE_LDS_BINS];
    const DevContig& dc = contigs[blockIdx.y];
    const uint32_t C = *dc.n_cols;
    if (!(bins_x(dc, C) || dc.widef) || !dc.wcols) return;   // (no object of this chain's index has more than PG_AMAX alleles)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // one wave per entry of the chain's list of wide columns (k_records made it)
    const uint32_t id = blockIdx.x * 4u + wave;
    if (id >= *dc.n_wcols) return;
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)dc.wcols[id]);
    const unsigned char* rec = dc.vrec + (size_t)dc.col_variant[c] * dc.RB;
    const uint32_t ax = (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const uint32_t*)(rec + PG_REC_AUX));
    const size_t colsz = (size_t)dc.HP * dc.HP;
    const double* mine = (const double*)(dc.aux + (size_t)ax * 16u);   // what this column's phase-2 role stored
    const double* stored = dc.fwd + (size_t)c * (dc.tri ? (size_t)dc.col_stride : colsz);   // its partner, from phase 1 (DevContig::widef chains with triangle storage: a triangle)
    const uint32_t st = dc.tri ? 1u : 0u;
    // c >= mid: the forward role ran phase 2 (alpha' = its column, beta' = the stored one); below: the backward role
    if (c >= C / 2) post_ab(dc, C, c, mine, stored, lane, s_bins[wave], s_wide[wave], st << 1);
    else post_ab(dc, C, c, stored, mine, lane, s_bins[wave], s_wide[wave], st);
}

#include "pg_split.h"   // the split path: index-level kernels, sample-level emissions and bins of the 16-path chains of fused jobs

// ------------------------------------------------------------------------------------------
//  host-callable launchers (defined here so that the shim needs no kernel templates)
// ------------------------------------------------------------------------------------------
// hp_mask: bit0 HP=16, bit1 HP=32, bit2 HP=64, bit3 HP=128; phase 1 = store halves, 2 = posterior halves
// The opt-in to more than 64 KiB of dynamic LDS is a per-device attribute of a kernel: it is set once
// per (kernel, device) — jobs on several devices can share a process (HMM::set_device()).  A racing
// second thread at worst sets it twice.
#define PG_MAX_DEVICES 64
static bool lds_attr_pending(bool (&done)[PG_MAX_DEVICES]) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PG_MAX_DEVICES) return true;  // unknown: just set it again
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}
template <int HP, int R, int VBUF, bool KEEPW, int PHASE>
static void launch_one(const DevContig* d_contigs, uint32_t n_contigs, uint32_t chunk, hipStream_t s) {
    using Cfg = ChainCfg<HP, R, sweep_has_loader<HP, PHASE>()>;
    size_t dyn = (Cfg::LOADER && PHASE == 2) ? (size_t)kRingSlots * HP * HP * 8 : 0;  // partner-column ring
    if (!Cfg::LOADER && PHASE == 2) dyn = (size_t)(Cfg::PARK / 2) * 16 * Cfg::T;       // parked rows (ChainCfg::PARK)
    if (HP == 64 && dyn > 0) {  // the triangle ring of lean chains (compact slots + the zero unit)
        if (dyn < kTriRingB) dyn = kTriRingB;
    }
    auto kern = k_sweep<HP, R, VBUF, KEEPW, PHASE>;
    static bool attr_done[PG_MAX_DEVICES];
    if (dyn > 0 && lds_attr_pending(attr_done))  // more than the default 64 KiB of LDS per workgroup
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipLaunchKernelGGL(kern, dim3(n_contigs, 2), dim3(Cfg::TT), dyn, s, d_contigs, chunk);
}
template <int PHASE>
static void launch_sweep(const DevContig* d_contigs, uint32_t n_contigs, uint32_t hp_mask, uint32_t chunk, hipStream_t s) {
    if (hp_mask & 1u) launch_one<16, 4, 1, true, PHASE>(d_contigs, n_contigs, chunk, s);
#ifndef PG_HP32_ROWS
#define PG_HP32_ROWS 8
#endif
    if (hp_mask & 2u) launch_one<32, PG_HP32_ROWS, 1, true, PHASE>(d_contigs, n_contigs, chunk, s);
    if (hp_mask & 4u) launch_one<64, 16, 1, true, PHASE>(d_contigs, n_contigs, chunk, s);
    if constexpr (PHASE == 1) {
        if (hp_mask & 2048u)   // bit 11: 64-path triangle chains that are not lean chains, phase 1 on the general kernel (PG_KERNELS=noleanx)
            hipLaunchKernelGGL(k_sweep_tri1, dim3(n_contigs, 2), dim3(ChainCfg<64, 16, sweep_has_loader<64, 1>()>::TT), 0, s, d_contigs);
        if (hp_mask & 4096u)   // bit 12: ... on the lean-x step (DevContig::leanx == 2)
            hipLaunchKernelGGL(k_sweep_leanx_tri, dim3(n_contigs, 2), dim3(LxCfg<64>::T), 0, s, d_contigs);
        if (hp_mask & 16384u)  // bit 14: ... of chains with wide columns (DevContig::widef)
            hipLaunchKernelGGL(k_sweep_leanx_triw, dim3(n_contigs, 2), dim3(LxCfg<64>::T), 0, s, d_contigs);
    }
    if (hp_mask & 8u) launch_one<128, 32, 1, false, PHASE>(d_contigs, n_contigs, chunk, s);
    if constexpr (PHASE == 2) {
        if (hp_mask & 256u) hipLaunchKernelGGL((k_sweep_lean2<16>), dim3(n_contigs, 2), dim3(256), 0, s, d_contigs);  // bit 8: chains with tri == 2
        if (hp_mask & 8192u) hipLaunchKernelGGL(k_sweep_leanx2, dim3(n_contigs, 2), dim3(256), 0, s, d_contigs);      // bit 13: DevContig::leanx2
    }
    if constexpr (PHASE != 2) {
        if (hp_mask & 64u) {  // bit 6: the job has lean chains (all-biallelic, H = HP = 64)
            if (PHASE == 1 && (hp_mask & 128u))  // bit 7: fused job whose lean chains store triangles (DevContig::tri)
                hipLaunchKernelGGL((k_sweep_lean_tri<PHASE, 16>), dim3(n_contigs, 2), dim3(256), 0, s, d_contigs, chunk);
            else hipLaunchKernelGGL((k_sweep_lean<PHASE, 16, false>), dim3(n_contigs, 2), dim3(256), 0, s, d_contigs, chunk);
        }
        if (hp_mask & 512u)   // bit 9: the job has lean-x chains at HP = 128 (narrow columns only)
            hipLaunchKernelGGL((k_sweep_leanx<PHASE, 128>), dim3(n_contigs, 2), dim3(LxCfg<128>::T), 0, s, d_contigs, chunk);
        if (hp_mask & 1024u)  // bit 10: ... at HP = 64 (chains with multiallelic objects; all-biallelic H = 64 chains are bit 6)
            hipLaunchKernelGGL((k_sweep_leanx<PHASE, 64>), dim3(n_contigs, 2), dim3(LxCfg<64>::T), 0, s, d_contigs, chunk);
        // bit 4: contigs with HP >= 256; bit 5: (forced) the generic kernel for every HP >= 64
        if (hp_mask & 48u)
            hipLaunchKernelGGL(k_sweep_generic<PHASE>, dim3(n_contigs, 2), dim3(PG_GEN_THREADS), 0, s, d_contigs, chunk,
                               (hp_mask & 32u) ? 64u : 256u);
    }
}
extern "C" {

// max_w = the longest walk of k_prep over the chains (a chain's list of objects, or all its variants), max_m4 = the longest list of k_prep_m4
void pgk_launch_prep(const DevContig* d_contigs, uint32_t n_contigs, uint32_t max_v, uint32_t max_w, uint32_t max_m4, DevTable tab, hipStream_t s) {
    dim3 grid((max_w + 4 * PG_VREP - 1) / (4 * PG_VREP), n_contigs);
    if (max_w) hipLaunchKernelGGL(k_prep, grid, dim3(256), 0, s, d_contigs, tab);
    dim3 grid16((max_v + 16 * PG_VREP - 1) / (16 * PG_VREP), n_contigs);
    hipLaunchKernelGGL(k_prep_bi, grid16, dim3(256), 0, s, d_contigs, tab);  // (chains of biallelic objects; each kernel skips the other's)
    dim3 gridm((max_m4 + 16 * PG_VREP - 1) / (16 * PG_VREP), n_contigs);
    if (max_m4) hipLaunchKernelGGL(k_prep_m4, gridm, dim3(256), 0, s, d_contigs, tab);
}
void pgk_launch_compact(const DevContig* d_contigs, uint32_t n_contigs, hipStream_t s) {
    hipLaunchKernelGGL(k_compact, dim3(n_contigs), dim3(1024), 0, s, d_contigs);
}
// The index-level work of a job (pg_split.h), once per uploaded index: d_reps = one chain descriptor per index contig.
// ColumnIndexer flags -> column list -> (split chains) the per-column index records.
void pgk_launch_index(const DevContig* d_reps, uint32_t n_index, uint32_t max_v, uint32_t max_big, int any_split, hipStream_t s) {
    if (max_v == 0 || n_index == 0) return;
    // one thread per variant / column; the (rare) objects with more than 32 alleles from their list, one wave each
    hipLaunchKernelGGL(k_index_scan_t, dim3((max_v + 255u) / 256u, n_index), dim3(256), 0, s, d_reps);
    if (max_big) hipLaunchKernelGGL(k_index_scan, dim3((max_big + 3u) / 4u, n_index), dim3(256), 0, s, d_reps);
    hipLaunchKernelGGL(k_compact, dim3(n_index), dim3(1024), 0, s, d_reps);
    if (any_split) {
        hipLaunchKernelGGL(k_index_cols_t, dim3((max_v + 255u) / 256u, n_index), dim3(256), 0, s, d_reps);
        if (max_big) hipLaunchKernelGGL(k_index_cols, dim3((max_big + 3u) / 4u, n_index), dim3(256), 0, s, d_reps);
    }
}
// The sample-level emission kernels of split chains: max_b / max_m4 / max_w = the longest walks of k_prep_s_bi (a chain's list of
// biallelic objects, or all its variants), k_prep_s_m4 and k_prep_s_w over the chains
void pgk_launch_prep_split(const DevContig* d_contigs, uint32_t n_contigs, uint32_t max_b, uint32_t max_m4, uint32_t max_w, DevTable tab, hipStream_t s) {
    if (max_b) hipLaunchKernelGGL(k_prep_s_bi, dim3((max_b + 255u) / 256u, n_contigs), dim3(256), 0, s, d_contigs, tab);
    if (max_m4) hipLaunchKernelGGL(k_prep_s_m4, dim3((max_m4 + 16 * PG_VREP - 1) / (16 * PG_VREP), n_contigs), dim3(256), 0, s, d_contigs, tab);
    if (max_w) hipLaunchKernelGGL(k_prep_s_w, dim3((max_w + 4 * PG_VREP - 1) / (4 * PG_VREP), n_contigs), dim3(256), 0, s, d_contigs, tab);
}
void pgk_launch_records(const DevContig* d_contigs, uint32_t n_contigs, uint32_t max_v, hipStream_t s) {
    dim3 grid((max_v + 255) / 256, n_contigs);   // 256 columns per block either way (a thread or a quarter of a wave's 64 each)
    hipLaunchKernelGGL(k_records, grid, dim3(256), 0, s, d_contigs);
}
// which: bit 0 = the job has chains whose bins k_bins forms, bit 1 = chains on k_sweep_lean2 (k_bins_lean2), bit 2 = chains of k_bins_thin
void pgk_launch_bins(const DevContig* d_contigs, uint32_t n_contigs, uint32_t max_v, uint32_t which, uint32_t max_wide, hipStream_t s) {
    // (a chain on k_sweep_lean2 that ends up with a single column is k_bins' too: one block per chain covers that)
    dim3 grid((which & 1u) ? (max_v + 3) / 4 : 1u, n_contigs);
    hipLaunchKernelGGL(k_bins, grid, dim3(256), 0, s, d_contigs);
    dim3 grid256((max_v + 255) / 256, n_contigs);
    if (which & 2u) hipLaunchKernelGGL(k_bins_lean2, grid256, dim3(256), 0, s, d_contigs);  // (each kernel skips the other's columns)
    if (which & 4u) hipLaunchKernelGGL(k_bins_thin, grid256, dim3(256), 0, s, d_contigs);
    if (which & 8u) hipLaunchKernelGGL(k_bins_x, grid256, dim3(256), 0, s, d_contigs);      // bit 3: chains on k_sweep_small16x<2>
    if ((which & 16u) && max_wide)   // bit 4: ... with objects of more than PG_AMAX alleles: one wave per listed wide column
        hipLaunchKernelGGL(k_bins_wide, dim3((max_wide + 3u) / 4u, n_contigs), dim3(256), 0, s, d_contigs);
    if (which & 32u) hipLaunchKernelGGL(k_bins_s, grid256, dim3(256), 0, s, d_contigs);     // bit 5: split chains (pg_split.h)
    if ((which & 64u) && max_wide)   // bit 6: ... with wide columns
        hipLaunchKernelGGL(k_bins_wide_s, dim3((max_wide + 3u) / 4u, n_contigs), dim3(256), 0, s, d_contigs);
    if (which & 128u) hipLaunchKernelGGL(k_bins_q, dim3((max_v + 15u) / 16u, n_contigs), dim3(256), 0, s, d_contigs);   // bit 7: chains on k_sweep_leanx2
}
void pgk_launch_sweep(const DevContig* d_contigs, uint32_t n_contigs, uint32_t hp_mask, int phase, hipStream_t s) {
    if (phase == 1) launch_sweep<1>(d_contigs, n_contigs, hp_mask, 0, s);
    else launch_sweep<2>(d_contigs, n_contigs, hp_mask, 0, s);
}
// chunked mode: chunk `chunk` of the second half of every half-chain (store-only sweep), then its posteriors
void pgk_launch_sweep_chunk(const DevContig* d_contigs, uint32_t n_contigs, uint32_t hp_mask, uint32_t chunk, hipStream_t s) {
    launch_sweep<3>(d_contigs, n_contigs, hp_mask, chunk, s);
}
// the store-only phases of the H = 16 chains (DevContig::small): four half-chains per wave, phase 1 or chunk `chunk` of phase 3
void pgk_launch_sweep_small(const DevContig* d_contigs, const uint32_t* d_ids, uint32_t n_ids, int phase, uint32_t chunk, double* d_dump, hipStream_t s) {
    if (n_ids == 0) return;
    const dim3 grid((n_ids + 3u) / 4u, 2);
    if (phase == 1) hipLaunchKernelGGL(k_sweep_small16<1>, grid, dim3(64), 0, s, d_contigs, d_ids, n_ids, chunk, d_dump);
    else if (phase == 2) hipLaunchKernelGGL(k_sweep_small16<2>, grid, dim3(64), 0, s, d_contigs, d_ids, n_ids, chunk, d_dump);
    else hipLaunchKernelGGL(k_sweep_small16<3>, grid, dim3(64), 0, s, d_contigs, d_ids, n_ids, chunk, d_dump);
}
// ... and of the H = 16 chains with multiallelic objects (DevContig::smallx): k_sweep_small16x
void pgk_launch_sweep_smallx(const DevContig* d_contigs, const uint32_t* d_ids, uint32_t n_ids, int phase, uint32_t chunk, double* d_dump, hipStream_t s) {
    if (n_ids == 0) return;
    const dim3 grid((n_ids + 3u) / 4u, 2);
    if (phase == 1) hipLaunchKernelGGL(k_sweep_small16x<1>, grid, dim3(64), 0, s, d_contigs, d_ids, n_ids, chunk, d_dump);
    else if (phase == 2) hipLaunchKernelGGL(k_sweep_small16x<2>, grid, dim3(64), 0, s, d_contigs, d_ids, n_ids, chunk, d_dump);
    else hipLaunchKernelGGL(k_sweep_small16x<3>, grid, dim3(64), 0, s, d_contigs, d_ids, n_ids, chunk, d_dump);
}
// Blocks per chain of k_post / k_post_loop: they fill the CUs the chains leave idle and no more.  A block in excess would
// sit in the queue and take the CU of a chain workgroup the moment a chunk sweep ends — the next chunk's
// workgroup (which cannot share a CU with it, by LDS size) then waits for it: measured on the 24-contig
// genome, 8 blocks per chain (= (256 - 48) / 24) 144 ms for phase 2, 6 blocks 164 ms, 10 blocks 173 ms,
// uncapped 172 ms.  *cus_out (optional): the device's CU count.
uint32_t pgk_post_blocks(uint32_t n_contigs, uint32_t chunk_cols, uint32_t* cus_out) {
    static int cus[PG_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PG_MAX_DEVICES) dev = 0;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    if (cus_out) *cus_out = (uint32_t)cus[dev];
    const int idle = cus[dev] - 2 * (int)n_contigs;
    const uint32_t cap = idle > (int)n_contigs ? (uint32_t)(idle / (int)n_contigs) : 1u;
    uint32_t bx = (2u * chunk_cols + PG_POST_WAVES - 1u) / PG_POST_WAVES;
    if (bx > cap) bx = cap;
    if (const char* e = getenv("PG_POST_BLOCKS")) { const long v = strtol(e, nullptr, 0); if (v >= 1 && (uint32_t)v < bx) bx = (uint32_t)v; }   // (experiments: fewer)
    return bx ? bx : 1u;
}
void pgk_launch_post(const DevContig* d_contigs, uint32_t n_contigs, uint32_t chunk_cols, uint32_t chunk, hipStream_t s) {
    static bool attr_done[PG_MAX_DEVICES];
    if (lds_attr_pending(attr_done))
        (void)hipFuncSetAttribute((const void*)k_post, hipFuncAttributeMaxDynamicSharedMemorySize, PG_POST_PLACEMENT_LDS);
    // (blocks = the CUs the chains leave idle, shared out inside the kernel over the chains that have columns in this chunk)
    const uint32_t per_chain = pgk_post_blocks(n_contigs, chunk_cols, nullptr);
    if (n_contigs <= 64u && getenv("PG_POST_FIXED") == nullptr) {
        uint32_t cus = 0;
        (void)pgk_post_blocks(n_contigs, chunk_cols, &cus);
        uint32_t total = cus > 2u * n_contigs + n_contigs ? cus - 2u * n_contigs : n_contigs;
        if (const char* e = getenv("PG_POST_BLOCKS")) { const long v = strtol(e, nullptr, 0); if (v >= 1) total = std::min<uint32_t>(total, (uint32_t)v * n_contigs); }
        const uint32_t most = ((2u * chunk_cols + PG_POST_WAVES - 1u) / PG_POST_WAVES) * n_contigs;
        if (total > most) total = most;
        hipLaunchKernelGGL(k_post, dim3(total, 1), dim3(64 * PG_POST_WAVES), PG_POST_PLACEMENT_LDS, s, d_contigs, chunk, n_contigs);
    } else {
        hipLaunchKernelGGL(k_post, dim3(per_chain, n_contigs), dim3(64 * PG_POST_WAVES), PG_POST_PLACEMENT_LDS, s, d_contigs, chunk, 0u);
    }
}
// Do kernels on two streams really run side by side?  HIP maps streams onto a handful of hardware queues; two streams that share
// one run their kernels one after the other, and the persistent phase 2 (whose kernels wait for each other) would stall until its
// timeouts.  Role 0 (first stream) raises w[0] and waits — at most 20 ms — for w[1], which role 1 (second stream) raises: w[2] = 1
// if it arrived.  Asked once per (stream, stream2) pair (pg_shim.cpp: streams_concurrent).
__global__ void k_stream_handshake(uint32_t* w, uint32_t role) {
    if (threadIdx.x != 0) return;
    if (role == 1u) { __hip_atomic_store(w + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); return; }
    __hip_atomic_store(w + 0, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    uint32_t ok = 0;
    for (;;) {
        if (__hip_atomic_load(w + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) { ok = 1; break; }
        if (wall_clock64() - t0 > 2000000ull) break;   // 20 ms of the 100 MHz clock
        __builtin_amdgcn_s_sleep(20);
    }
    w[2] = ok;
}
void pgk_launch_stream_handshake(uint32_t* d_words, hipStream_t s, hipStream_t s2) {
    hipLaunchKernelGGL(k_stream_handshake, dim3(1), dim3(64), 0, s, d_words, 0u);
    hipLaunchKernelGGL(k_stream_handshake, dim3(1), dim3(64), 0, s2, d_words, 1u);
}
// The persistent chunked phase 2 of all-lean jobs (DevContig::sync): the second half of every half-chain in ONE launch of
// k_sweep_lean<4> on `s`, every chunk's posteriors in ONE launch of k_post_loop on `s2`.  The caller has checked that both grids
// fit the device with room to spare (2 n_contigs + n_contigs * post_blocks workgroups, one per CU): k_post_loop waits for the sweep's
// workgroups, which must all be running; the sweep waits for columns that running k_post_loop waves have taken.
void pgk_launch_phase2_persistent(const DevContig* d_contigs, uint32_t n_contigs, uint32_t post_blocks, hipStream_t s, hipStream_t s2) {
    static bool attr_done[PG_MAX_DEVICES];
    if (lds_attr_pending(attr_done))
        (void)hipFuncSetAttribute((const void*)k_post_loop, hipFuncAttributeMaxDynamicSharedMemorySize, PG_POST_PLACEMENT_LDS);
    hipLaunchKernelGGL((k_sweep_lean<4, 16, false>), dim3(n_contigs, 2), dim3(256), 0, s, d_contigs, post_blocks);
    hipLaunchKernelGGL(k_post_loop, dim3(post_blocks, n_contigs), dim3(64 * PG_POST_WAVES), PG_POST_PLACEMENT_LDS, s2, d_contigs);
}
void pgk_launch_emission_single(const DevContig* d_contig, DevTable tab, uint32_t v, double* out_m, int* out_e,
                                hipStream_t s) {
    hipLaunchKernelGGL(k_emission_single, dim3(1), dim3(64), 0, s, d_contig, tab, v, out_m, out_e);
}
void pgk_launch_transition_single(double d, uint32_t H, int uniform, double* out3, hipStream_t s) {
    hipLaunchKernelGGL(k_transition_single, dim3(1), dim3(64), 0, s, d, H, uniform, out3);
}
uint32_t pgk_threads_for_hp(uint32_t hp) {
    switch (hp) { case 16: return 64; case 32: return 32 * 32 / PG_HP32_ROWS; case 64: return 256; case 128: return 512; default: return hp >= 256 ? 1024 : 0; }
}

}  // extern "C"
