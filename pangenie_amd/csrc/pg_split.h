// pg_split.h — the split path (round 6): index-level kernels, run once per uploaded index, and the sample-level emission
// and bins kernels of the 16-path chains of fused jobs (DevContig::split).  Layouts: pg_device.h.  Included by
// pg_kernels.hip (its helpers: cn_on_the_fly, mix2 / mix3, split, store_bin, post_ab, transition_consts, ...).
//
//   k_index_scan   ColumnIndexer of one index contig: kept[v], allele_present (one wave per variant; any H, any A)
//                  reference src/columnindexer.cpp:12-31
//   k_compact      (pg_kernels.hip) kept -> col_variant, n_cols, col_of
//   k_index_cols   per column of a split chain's index: ix_pd, ix_rec (Li-Stephens constants, local allele of every path),
//                  ix_bin, the list of wide columns (one wave per column)
//                  reference src/transitionprobabilitycomputer.cpp:8-19, src/columnindexer.cpp:59-69
//   k_prep_s_bi    emission products of biallelic objects with <= 32 k-mers: ONE LANE per column, the product over the
//                  k-mers a sequential loop in the lane (no cross-lane step, no LDS), two 16-byte table loads per k-mer
//   k_prep_s_m4 / k_prep_s_w   (pg_kernels.hip: prep_m4_unit<true> / prep_unit<true>) 3 .. PG_AMAX alleles / everything else
//                  reference src/emissionprobabilitycomputer.cpp:9-53
//   k_bins_s       class sums / aux bins -> genotype bins, one thread per column; k_bins_wide_s: wide columns, one wave each
//                  reference src/hmm.cpp:364-368
#pragma once

// ------------------------------------------------------------------------------------------
//  k_index_scan : one wave per variant
// ------------------------------------------------------------------------------------------
// (objects with more than PG_IX_THREAD_AMAX alleles only — the list ix_big of the index contig; every other object is
//  k_index_scan_t's, one THREAD per variant: a wave per variant spent 46 ms on the 32.8 M objects of 4096 sixteen-path panels)
#define PG_IX_THREAD_AMAX 32u
__global__ __launch_bounds__(256) void k_index_scan(const DevContig* __restrict__ reps) {
    __shared__ uint32_t s_pres[4][8];
    __shared__ uint16_t s_aid[4][64];
    __shared__ uint8_t s_afl[4][64];
    const DevContig& dc = reps[blockIdx.y];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t bi = blockIdx.x * 4u + wave;
    if (bi >= dc.n_ix_big) return;
    const uint32_t v = dc.ix_big[bi];
    if (v >= dc.V) return;
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0, H = dc.H;
    if (A > PG_MAX_ALLELES_PER_VARIANT || A == 0) {
        if (lane == 0) { atomicOr(dc.ix_err, PG_DEVERR_TOO_MANY_ALLELES); dc.kept[v] = 0; }
        return;
    }
    uint32_t* pres = s_pres[wave];
    if (lane < 8) pres[lane] = 0;
    const bool staged = A <= 64u;
    if (staged && lane < A) { s_aid[wave][lane] = dc.allele_id[a0 + lane]; s_afl[wave][lane] = dc.allele_flags[a0 + lane]; }
    wave_sync_lds();
    bool nonref = false, bad = false;
    for (uint32_t p = lane; p < H; p += 64) {
        const uint16_t a = dc.path_allele[(size_t)v * H + p];
        int s = -1;
        if (staged) { for (uint32_t q = 0; q < A; ++q) if (s_aid[wave][q] == a) s = (int)q; }
        else s = slot_of(dc, a0, A, a);
        if (s < 0) bad = true;
        else {
            atomicOr(&pres[(uint32_t)s >> 5], 1u << ((uint32_t)s & 31u));
            const uint8_t fl = staged ? s_afl[wave][s] : dc.allele_flags[a0 + s];
            if (a != 0 && !(fl & 1)) nonref = true;
        }
    }
    wave_sync_lds();
    const bool kept = __any(nonref) != 0;
    if (__any(bad) != 0) {
        if (lane == 0) { atomicOr(dc.ix_err, PG_DEVERR_ALLELE_NOT_FOUND); dc.kept[v] = 0; }
        return;
    }
    for (uint32_t q = lane; q < A; q += 64) dc.allele_present[a0 + q] = slot_present(pres, q) ? 1 : 0;
    if (lane == 0) dc.kept[v] = kept ? 1 : 0;
    if (kept) {
        uint32_t n_local = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) n_local += __popc(pres[q]);
        if (n_local > PG_AMAX && (!dc.wide_idx || dc.wide_idx[v] == PG_WIDE_NONE || n_local > PG_WIDE_MAX))
            if (lane == 0) atomicOr(dc.ix_err, PG_DEVERR_TOO_MANY_LOCAL);
    }
}

// ------------------------------------------------------------------------------------------
//  k_index_cols : one wave per column of a split chain's index contig (H = HP = 16)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_index_cols(const DevContig* __restrict__ reps) {
    __shared__ uint32_t s_pres[4][8];
    __shared__ uint16_t s_aid[4][64];
    const DevContig& dc = reps[blockIdx.y];
    if (!dc.split) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t bi = blockIdx.x * 4u + wave;   // (the index contig's objects with more than PG_IX_THREAD_AMAX alleles: see k_index_scan)
    if (bi >= dc.n_ix_big) return;
    const uint32_t vb = dc.ix_big[bi];
    const uint32_t c = dc.col_of[vb];
    if (c == PG_COL_NONE) return;
    const uint32_t v = dc.col_variant[c];
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0, H = dc.H;
    uint32_t* pres = s_pres[wave];
    if (lane < 8) pres[lane] = 0;
    const bool staged = A <= 64u;
    if (staged && lane < A) s_aid[wave][lane] = dc.allele_id[a0 + lane];
    wave_sync_lds();
    for (uint32_t q = lane; q < A; q += 64)
        if (dc.allele_present[a0 + q]) atomicOr(&pres[q >> 5], 1u << (q & 31u));
    wave_sync_lds();
    uint32_t n_local = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) n_local += __popc(pres[q]);
    const bool wide = n_local > PG_AMAX;
    // local (dense) allele of the sixteen paths: lane p
    uint32_t loc = PG_PHANTOM;
    if (lane < H) {
        const uint16_t a = dc.path_allele[(size_t)v * H + lane];
        int s = -1;
        if (staged) { for (uint32_t q = 0; q < A; ++q) if (s_aid[wave][q] == a) s = (int)q; }
        else s = slot_of(dc, a0, A, a);
        loc = s < 0 ? (uint32_t)PG_PHANTOM : local_index(pres, (uint32_t)s);
    }
    const uint32_t bits1 = (uint32_t)(__ballot(loc == 1u) & 0xFFFFull);
    uint32_t cnt[PG_AMAX];
#pragma unroll
    for (int l = 0; l < PG_AMAX; ++l) cnt[l] = (uint32_t)__popcll(__ballot(loc == (uint32_t)l));
    // the sixteen bytes lane p < 16 contributes to: row offsets (narrow) — gathered with shuffles by lanes 0 .. 3
    const uint32_t ro = (lane < 16u && !wide && loc < (uint32_t)PG_AMAX) ? (loc + 1u) * (uint32_t)(PG_ESTRIDE * 8) : 0u;
    uint32_t row_word = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) row_word |= ((uint32_t)__shfl((int)ro, (int)((lane & 3u) * 4u + (uint32_t)b)) & 0xFFu) << (8 * b);
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, kappa = 0.0;
    if (c > 0) {
        const double d = (double)(dc.pos[v] - dc.pos[dc.col_variant[c - 1]]) * dc.dist_scale;
        transition_consts(d, dc.H, dc.uniform, c0, c1, c2, kappa);
    }
    const uint32_t widx = wide && dc.wide_idx ? dc.wide_idx[v] : PG_WIDE_NONE;
    const uint32_t aux = dc.aux_idx ? dc.aux_idx[v] : PG_WIDE_NONE;
    const uint32_t nlf = (n_local & 0xFFu) | (wide ? (uint32_t)PG_REC_FLAG_WIDE << 8 : 0u);
    unsigned char* rec = const_cast<unsigned char*>(dc.ix_rec) + (size_t)c * PG_IXREC_BYTES;
    if (dc.split == 1u) {
        if (lane == 0) {
            double* r = (double*)rec;
            r[0] = c0; r[1] = c1; r[2] = c2; r[3] = kappa;
            ((unsigned long long*)rec)[4] = (unsigned long long)bits1;
            r[5] = 0.0; r[6] = 0.0; r[7] = 0.0;
        }
    } else {
        if (lane == 0) {
            uint32_t* h = (uint32_t*)rec;
            h[0] = nlf; h[1] = widx; h[2] = aux; h[3] = 0u;
            double* r = (double*)(rec + 16);
            r[0] = c0; r[1] = c1; r[2] = c2; r[3] = kappa;
        }
        if (lane < 4u) ((uint32_t*)(rec + 48))[lane] = row_word;
    }
    // prep descriptor
    if (lane == 0) {
        const uint32_t k0 = dc.kmer_off[v], K = dc.kmer_off[v + 1] - k0;
        uint32_t kf = K > 0xFFu ? 0xFFu : K, on0 = 0, on1 = 0;
        if (A == 2u) {
            const uint32_t off0 = dc.allele_koff[a0], off1 = dc.allele_koff[a0 + 1];
            on0 = off0 < 32u ? dc.allele_kmask[a0] << off0 : 0u;
            on1 = off1 < 32u ? dc.allele_kmask[a0 + 1] << off1 : 0u;
            if (dc.allele_flags[a0] & 1) kf |= PG_IXPD_U0;
            if (dc.allele_flags[a0 + 1] & 1) kf |= PG_IXPD_U1;
            if (pres[0] & 1u) kf |= PG_IXPD_HAS0;
            if (pres[0] & 2u) kf |= PG_IXPD_HAS1;
        }
        uint32_t* pd = (uint32_t*)(const_cast<unsigned char*>(dc.ix_pd) + (size_t)c * PG_IXPD_BYTES);
        pd[0] = v; pd[1] = k0; pd[2] = kf; pd[3] = on0; pd[4] = on1; pd[5] = 0; pd[6] = 0; pd[7] = 0;
        IxBin b;
        b.g0 = (uint32_t)dc.geno_off[v]; b.v = v; b.aux = aux; b.A = (uint16_t)A; b.nl = (uint8_t)(n_local > 255u ? 255u : n_local);
        b.flags = wide ? (uint8_t)PG_SREC_FLAG_WIDE : 0; b.pad = 0;
        uint32_t l = 0;
        for (uint32_t sl = 0; sl < A && l < (uint32_t)PG_AMAX; ++sl)
            if (slot_present(pres, sl)) b.ls[l++] = (uint16_t)sl;
        for (; l < (uint32_t)PG_AMAX; ++l) b.ls[l] = 0;
#pragma unroll
        for (int q = 0; q < PG_AMAX; ++q) b.cnt[q] = (uint8_t)cnt[q];
        *(IxBin*)(const_cast<unsigned char*>(dc.ix_bin) + (size_t)c * PG_IXBIN_BYTES) = b;
        if (wide && dc.wcols) dc.wcols[atomicAdd(dc.n_wcols, 1u)] = c;
    }
}

// ------------------------------------------------------------------------------------------
//  k_index_scan_t / k_index_cols_t : the same, one THREAD per variant / column, for objects with at most PG_IX_THREAD_AMAX
//  alleles (the presence bitmap is one register) — all but the largest bubbles
// ------------------------------------------------------------------------------------------
DEVI int ix_slot_of(const uint16_t* aid, uint32_t A, uint16_t a) {   // the LAST matching slot, as slot_of
    int s = -1;
    for (uint32_t q = 0; q < A; ++q) if (aid[q] == a) s = (int)q;
    return s;
}
__global__ __launch_bounds__(256) void k_index_scan_t(const DevContig* __restrict__ reps) {
    const DevContig& dc = reps[blockIdx.y];
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= dc.V) return;
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0, H = dc.H;
    if (A > PG_IX_THREAD_AMAX || A == 0) return;   // k_index_scan (A == 0 cannot be: check_batch)
    const uint16_t* aid = dc.allele_id + a0;
    const uint8_t* afl = dc.allele_flags + a0;
    const uint16_t* pa = dc.path_allele + (size_t)v * H;
    uint32_t pres = 0;
    bool nonref = false, bad = false;
    if (A == 2u) {
        const uint16_t id0 = aid[0], id1 = aid[1];
        const bool u0 = afl[0] & 1, u1 = afl[1] & 1;
        for (uint32_t p = 0; p < H; ++p) {
            const uint16_t a = pa[p];
            const bool s1 = a == id1, s0 = !s1 && a == id0;
            bad = bad || !(s1 || s0);
            pres |= s1 ? 2u : (s0 ? 1u : 0u);
            nonref = nonref || ((s1 || s0) && a != 0 && !(s1 ? u1 : u0));
        }
    } else {
        for (uint32_t p = 0; p < H; ++p) {
            const uint16_t a = pa[p];
            const int sl = ix_slot_of(aid, A, a);
            if (sl < 0) bad = true;
            else { pres |= 1u << sl; if (a != 0 && !(afl[sl] & 1)) nonref = true; }
        }
    }
    if (bad) { atomicOr(dc.ix_err, PG_DEVERR_ALLELE_NOT_FOUND); dc.kept[v] = 0; return; }
    for (uint32_t q = 0; q < A; ++q) dc.allele_present[a0 + q] = (pres >> q) & 1u;
    dc.kept[v] = nonref ? 1 : 0;
    if (nonref && (uint32_t)__popc(pres) > (uint32_t)PG_AMAX && (!dc.wide_idx || dc.wide_idx[v] == PG_WIDE_NONE)) atomicOr(dc.ix_err, PG_DEVERR_TOO_MANY_LOCAL);
}

__global__ __launch_bounds__(256) void k_index_cols_t(const DevContig* __restrict__ reps) {
    const DevContig& dc = reps[blockIdx.y];
    if (!dc.split) return;
    const uint32_t C = *dc.n_cols;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= C) return;
    const uint32_t v = dc.col_variant[c];
    const uint32_t a0 = dc.allele_off[v], A = dc.allele_off[v + 1] - a0;   // (H = HP = 16: split chains)
    if (A > PG_IX_THREAD_AMAX) return;   // k_index_cols
    const uint16_t* aid = dc.allele_id + a0;
    uint32_t pres = 0;
    for (uint32_t q = 0; q < A; ++q) if (dc.allele_present[a0 + q]) pres |= 1u << q;
    const uint32_t n_local = (uint32_t)__popc(pres);
    const bool wide = n_local > (uint32_t)PG_AMAX;
    const uint4 pw0 = *(const uint4*)(dc.path_allele + (size_t)v * 16u), pw1 = *(const uint4*)(dc.path_allele + (size_t)v * 16u + 8u);
    const uint32_t pwords[8] = {pw0.x, pw0.y, pw0.z, pw0.w, pw1.x, pw1.y, pw1.z, pw1.w};
    uint32_t bits1 = 0, row[4] = {0, 0, 0, 0}, cnt[PG_AMAX] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int pth = 0; pth < 16; ++pth) {
        const uint16_t a = (uint16_t)(pwords[pth >> 1] >> (16 * (pth & 1)));
        uint32_t loc = PG_PHANTOM;
        if (A == 2u) loc = a == aid[1] ? ((pres & 1u) ? 1u : 0u) : 0u;   // (slot 1 is local 1 if slot 0 is present too; the scan has vouched for every allele)
        else {
            const int sl = ix_slot_of(aid, A, a);
            if (sl >= 0) loc = (uint32_t)__popc(pres & ((1u << sl) - 1u));
        }
        bits1 |= (loc == 1u ? 1u : 0u) << pth;
#pragma unroll
        for (int l = 0; l < PG_AMAX; ++l) cnt[l] += loc == (uint32_t)l ? 1u : 0u;
        const uint32_t ro = (!wide && loc < (uint32_t)PG_AMAX) ? (loc + 1u) * (uint32_t)(PG_ESTRIDE * 8) : 0u;
        row[pth >> 2] |= ro << (8 * (pth & 3));
    }
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, kappa = 0.0;
    if (c > 0) {
        const double d = (double)(dc.pos[v] - dc.pos[dc.col_variant[c - 1]]) * dc.dist_scale;
        transition_consts(d, dc.H, dc.uniform, c0, c1, c2, kappa);
    }
    const uint32_t widx = wide && dc.wide_idx ? dc.wide_idx[v] : PG_WIDE_NONE;
    const uint32_t aux = dc.aux_idx ? dc.aux_idx[v] : PG_WIDE_NONE;
    const uint32_t nlf = (n_local & 0xFFu) | (wide ? (uint32_t)PG_REC_FLAG_WIDE << 8 : 0u);
    v2f64* rec = (v2f64*)(const_cast<unsigned char*>(dc.ix_rec) + (size_t)c * PG_IXREC_BYTES);
    if (dc.split == 1u) {
        rec[0] = v2f64{c0, c1}; rec[1] = v2f64{c2, kappa};
        rec[2] = v2f64{__longlong_as_double((long long)(unsigned long long)bits1), 0.0}; rec[3] = v2f64{0.0, 0.0};
    } else {
        const uint4 h = uint4{nlf, widx, aux, 0u}, r4 = uint4{row[0], row[1], row[2], row[3]};
        rec[0] = *(const v2f64*)&h; rec[1] = v2f64{c0, c1}; rec[2] = v2f64{c2, kappa}; rec[3] = *(const v2f64*)&r4;
    }
    const uint32_t k0 = dc.kmer_off[v], K = dc.kmer_off[v + 1] - k0;
    uint32_t kf = K > 0xFFu ? 0xFFu : K, on0 = 0, on1 = 0;
    if (A == 2u) {
        const uint32_t off0 = dc.allele_koff[a0], off1 = dc.allele_koff[a0 + 1];
        on0 = off0 < 32u ? dc.allele_kmask[a0] << off0 : 0u;
        on1 = off1 < 32u ? dc.allele_kmask[a0 + 1] << off1 : 0u;
        if (dc.allele_flags[a0] & 1) kf |= PG_IXPD_U0;
        if (dc.allele_flags[a0 + 1] & 1) kf |= PG_IXPD_U1;
        if (pres & 1u) kf |= PG_IXPD_HAS0;
        if (pres & 2u) kf |= PG_IXPD_HAS1;
    }
    uint4* pd = (uint4*)(const_cast<unsigned char*>(dc.ix_pd) + (size_t)c * PG_IXPD_BYTES);
    pd[0] = uint4{v, k0, kf, on0}; pd[1] = uint4{on1, 0u, 0u, 0u};
    IxBin b;
    b.g0 = (uint32_t)dc.geno_off[v]; b.v = v; b.aux = aux; b.A = (uint16_t)A; b.nl = (uint8_t)n_local;
    b.flags = wide ? (uint8_t)PG_SREC_FLAG_WIDE : 0; b.pad = 0;
    uint32_t l = 0;
    for (uint32_t sl = 0; sl < A && l < (uint32_t)PG_AMAX; ++sl)
        if ((pres >> sl) & 1u) b.ls[l++] = (uint16_t)sl;
    for (; l < (uint32_t)PG_AMAX; ++l) b.ls[l] = 0;
#pragma unroll
    for (int q = 0; q < PG_AMAX; ++q) b.cnt[q] = (uint8_t)cnt[q];
    *(IxBin*)(const_cast<unsigned char*>(dc.ix_bin) + (size_t)c * PG_IXBIN_BYTES) = b;
    if (wide && dc.wcols) dc.wcols[atomicAdd(dc.n_wcols, 1u)] = c;
}

// ------------------------------------------------------------------------------------------
//  k_prep_s_bi : one lane per column
// ------------------------------------------------------------------------------------------
struct Cn3 { double m[3]; int e[3]; };
DEVI Cn3 cn_lookup_packed(const DevTable& t, uint32_t cov, uint32_t count) {
    Cn3 r;
    if (cov >= t.cov_min && cov < t.cov_max && count < t.count_max) {
        const unsigned char* p = t.packed + ((size_t)(cov - t.cov_min) * t.count_max + count) * 32u;
        const v2f64 a = *(const v2f64*)p, b = *(const v2f64*)(p + 16);
        r.m[0] = a.x; r.m[1] = a.y; r.m[2] = b.x;
        const unsigned long long ew = (unsigned long long)__double_as_longlong(b.y);
        r.e[0] = (int)(short)(ew & 0xFFFFull); r.e[1] = (int)(short)((ew >> 16) & 0xFFFFull); r.e[2] = (int)(short)((ew >> 32) & 0xFFFFull);
    } else {
        cn_on_the_fly(t.reg, cov, count, r.m, r.e);
    }
    return r;
}

// what the three pair products of a biallelic object become in its column's record (the rules of prep_bi_unit):
// local pairs, the largest exponent X over the present ones, the scaled entries, flags
struct BiOut { double E00, E01, E11, m00, m01, m11; int e00, e01, e11, X; uint32_t flags; };
DEVI BiOut bi_finish(double (&pm)[3], int (&pe)[3], bool has0, bool has1) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double mm; int ee;
        split(pm[q], mm, ee);
        pm[q] = mm; pe[q] += ee;
        if (pe[q] < PG_LD_MIN_EXP) { pm[q] = 0.0; pe[q] = 0; }  // underflows to 0 in the reference too
    }
    const bool all_zeros = !(pm[0] > 0.0 || pm[1] > 0.0 || pm[2] > 0.0);  // over ALL pairs of the object (emissionprobabilitycomputer.cpp:24)
    BiOut o;
    const int q00 = has0 ? 0 : 2;
    o.m00 = pm[q00]; o.m01 = pm[1]; o.m11 = pm[2];
    o.e00 = pe[q00]; o.e01 = pe[1]; o.e11 = pe[2];
    const bool two = has0 && has1;
    int X = -(1 << 30);
    if (o.m00 > 0.0) X = o.e00;
    if (two && o.m01 > 0.0 && o.e01 > X) X = o.e01;
    if (two && o.m11 > 0.0 && o.e11 > X) X = o.e11;
    if (X == -(1 << 30) || all_zeros) X = 0;
    auto scaled = [&](double m, int e) { return all_zeros ? 1.0 : (m > 0.0 ? ldexp(m, e - X) : m); };  // (0 or NaN stays)
    o.E00 = scaled(o.m00, o.e00); o.E01 = two ? scaled(o.m01, o.e01) : 0.0; o.E11 = two ? scaled(o.m11, o.e11) : 0.0;
    // an entry that leaves fp64's normal range below X cannot be recovered from the scaled value: the column is flagged
    bool precise = false;
    if (!all_zeros) {
        precise = (o.m00 > 0.0 && o.e00 - X < -1021) || (two && o.m01 > 0.0 && o.e01 - X < -1021) || (two && o.m11 > 0.0 && o.e11 - X < -1021);
    } else { o.m00 = 0.5; o.e00 = 1; o.m01 = 0.5; o.e01 = 1; o.m11 = 0.5; o.e11 = 1; }
    if (!two) { o.m01 = 0.0; o.e01 = 0; o.m11 = 0.0; o.e11 = 0; }
    o.X = X;
    o.flags = (all_zeros ? PG_SREC_FLAG_ALLZERO : 0u) | (precise ? PG_SREC_FLAG_PRECISE : 0u);
    return o;
}
// the (mantissa, exponent) tables of a flagged column: m[16] then e[16], tri_local order
DEVI void cprec_store_bi(unsigned char* cprec, uint32_t c, const BiOut& o) {
    double* m = (double*)(cprec + (size_t)c * PG_CPREC_BYTES);
    int* e = (int*)(cprec + (size_t)c * PG_CPREC_BYTES + 128u);
    m[0] = o.m00; m[1] = o.m01; m[PG_AMAX] = o.m11;
    e[0] = o.e00; e[1] = o.e01; e[PG_AMAX] = o.e11;
}

__global__ __launch_bounds__(256) void k_prep_s_bi(const DevContig* __restrict__ contigs, DevTable tab) {
    const DevContig& dc = contigs[blockIdx.y];
    if (!dc.split) return;
    const uint32_t n = dc.prep_b ? dc.n_prep_b : dc.V;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t vi = dc.prep_b ? dc.prep_b[i] : i;
    const uint32_t c = dc.col_of[vi];
    if (c == PG_COL_NONE) return;
    const uint4 pd = *(const uint4*)(dc.ix_pd + (size_t)c * PG_IXPD_BYTES);   // variant, first k-mer, K | flags, bits of slot 0
    const uint32_t on1 = *(const uint32_t*)(dc.ix_pd + (size_t)c * PG_IXPD_BYTES + 16u);
    const uint32_t v = pd.x, k0 = pd.y, K = pd.z & 0xFFu, on0 = pd.w;
    const bool u0 = (pd.z & PG_IXPD_U0) != 0u, u1 = (pd.z & PG_IXPD_U1) != 0u;
    const uint32_t cov = dc.cov[v];
    double pm[3] = {1.0, 1.0, 1.0};
    int pe[3] = {0, 0, 0};
    const uint16_t* kc = dc.kmer_count + k0;
    if (!(u0 || u1)) {
        for (uint32_t k = 0; k < K; ++k) {
            const Cn3 t = cn_lookup_packed(tab, cov, kc[k]);
            const bool b0 = (on0 >> k) & 1u, b1 = (on1 >> k) & 1u;
            // copy number of the k-mer under the genotypes 0/0, 0/1, 1/1: 2 b0, b0 + b1, 2 b1
            pm[0] *= b0 ? t.m[2] : t.m[0]; pe[0] += b0 ? t.e[2] : t.e[0];
            pm[2] *= b1 ? t.m[2] : t.m[0]; pe[2] += b1 ? t.e[2] : t.e[0];
            const bool one = b0 != b1, both = b0 && b1;
            pm[1] *= both ? t.m[2] : (one ? t.m[1] : t.m[0]); pe[1] += both ? t.e[2] : (one ? t.e[1] : t.e[0]);
        }
    } else {
        for (uint32_t k = 0; k < K; ++k) {
            Cn3 t = cn_lookup_packed(tab, cov, kc[k]);
            const uint32_t b0 = (on0 >> k) & 1u, b1 = (on1 >> k) & 1u;
            auto factor = [&](uint32_t cn, bool ua, bool ub, double& fm, int& fe) {   // (emissionprobabilitycomputer.cpp:36-53, as prep_bi_unit)
                const double mc = cn == 0 ? t.m[0] : (cn == 1 ? t.m[1] : t.m[2]);
                const int ec = cn == 0 ? t.e[0] : (cn == 1 ? t.e[1] : t.e[2]);
                if (ua && ub) mix3(t.m, t.e, 1.0 / 3.0, fm, fe);
                else if (ua || ub) {
                    const double m2 = cn == 0 ? t.m[1] : t.m[2];   // cn + 1, capped at 2 (the reference asserts cn < 2 here)
                    const int e2 = cn == 0 ? t.e[1] : t.e[2];
                    mix2(mc, ec, m2, e2, 0.5, fm, fe);
                } else { fm = mc; fe = ec; }
            };
            double fm; int fe;
            factor(2u * b0, u0, u0, fm, fe); pm[0] *= fm; pe[0] += fe;
            factor(b0 + b1, u0, u1, fm, fe); pm[1] *= fm; pe[1] += fe;
            factor(2u * b1, u1, u1, fm, fe); pm[2] *= fm; pe[2] += fe;
        }
    }
    const BiOut o = bi_finish(pm, pe, (pd.z & PG_IXPD_HAS0) != 0u, (pd.z & PG_IXPD_HAS1) != 0u);
    if (o.flags & PG_SREC_FLAG_PRECISE) cprec_store_bi(dc.cprec, c, o);
    if (dc.split == 1u) {
        const v2f64* ir = (const v2f64*)(dc.ix_rec + (size_t)c * PG_IXREC_BYTES);
        const v2f64 a = ir[0], b = ir[1];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(ir[2].x);
        const unsigned long long packed = (bits & 0xFFFFull) | ((unsigned long long)o.flags << 16) | ((unsigned long long)(uint32_t)o.X << 32);
        v2f64* dst = (v2f64*)((unsigned char*)dc.frec + (size_t)c * PG_SREC1_BYTES);
        dst[0] = a; dst[1] = b;
        dst[2] = v2f64{o.E00, o.E01};
        dst[3] = v2f64{o.E11, __longlong_as_double((long long)packed)};
    } else {
        // entries 0, 1 (piece 0), 5 (piece 2), X / flags (piece 7): every other entry of a biallelic column's record is zero since
        // the index was uploaded (pg_shim.cpp zeroes the sample records then; which columns are biallelic hangs on the index alone)
        v2f64* dst = (v2f64*)((unsigned char*)dc.frec + (size_t)c * PG_SREC2_BYTES);
        const unsigned long long xf = (unsigned long long)(uint32_t)o.X | ((unsigned long long)o.flags << 32);
        dst[0] = v2f64{o.E00, o.E01};
        dst[2] = v2f64{0.0, o.E11};
        dst[7] = v2f64{0.0, __longlong_as_double((long long)xf)};
    }
}

// ------------------------------------------------------------------------------------------
//  bins of split chains
// ------------------------------------------------------------------------------------------
// (mantissa, exponent) of the unscaled emission product of a local pair from its scaled entry: E' = m 2^(e - X) exactly
DEVI void pair_from_entry(double E, int X, double& pm, int& pe) {
    pm = E; pe = 0;
    if (E != 0.0 && E == E && !isinf(E)) { int ee; pm = frexp(E, &ee); pe = ee + X; }
}
// local allele of path p (< 16) of column c of a split chain; PG_PHANTOM if none
DEVI uint32_t split_path_allele(const DevContig& dc, uint32_t c, uint32_t p) {
    if (dc.split == 1u) return (uint32_t)((*(const unsigned long long*)(dc.ix_rec + (size_t)c * PG_IXREC_BYTES + 32u) >> p) & 1ull);
    const uint32_t ro = dc.ix_rec[(size_t)c * PG_IXREC_BYTES + 48u + p];
    return ro ? ro / (uint32_t)(PG_ESTRIDE * 8) - 1u : (uint32_t)PG_PHANTOM;
}
DEVI int split_exponent_of(const DevContig& dc, uint32_t c) {
    if (dc.split == 1u) return (int)(uint32_t)(*(const unsigned long long*)((const unsigned char*)dc.frec + (size_t)c * PG_SREC1_BYTES + 56u) >> 32);
    return *(const int*)((const unsigned char*)dc.frec + (size_t)c * PG_SREC2_BYTES + 120u);
}
DEVI uint32_t split_flags_of(const DevContig& dc, uint32_t c) {
    if (dc.split == 1u) return (uint32_t)(*(const unsigned long long*)((const unsigned char*)dc.frec + (size_t)c * PG_SREC1_BYTES + 56u) >> 16) & 0xFFFFu;
    return *(const uint32_t*)((const unsigned char*)dc.frec + (size_t)c * PG_SREC2_BYTES + 124u);
}

__global__ __launch_bounds__(256) void k_bins_s(const DevContig* __restrict__ contigs) {
    const DevContig& dc = contigs[blockIdx.y];
    if (!dc.split) return;
    const uint32_t C = *dc.n_cols;
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= C) return;
    const uint4 b0 = *(const uint4*)(dc.ix_bin + (size_t)c * PG_IXBIN_BYTES), b1 = *(const uint4*)(dc.ix_bin + (size_t)c * PG_IXBIN_BYTES + 16u);
    const uint32_t g0 = b0.x, aux = b0.z, A = b0.w & 0xFFFFu, nl = (b0.w >> 16) & 0xFFu, bflags = b0.w >> 24;
    if (bflags & PG_SREC_FLAG_WIDE) return;   // k_bins_wide_s
    const uint32_t lsv[PG_AMAX] = {b1.x & 0xFFFFu, b1.x >> 16, b1.y & 0xFFFFu, b1.y >> 16, b1.z & 0xFFFFu};
    const uint32_t cntv[PG_AMAX] = {(b1.z >> 16) & 0xFFu, b1.z >> 24, b1.w & 0xFFu, (b1.w >> 8) & 0xFFu, (b1.w >> 16) & 0xFFu};
    const uint32_t flags = split_flags_of(dc, c);
    const int X = split_exponent_of(dc, c);
    // the scaled entries of the column's local pairs
    double E[PG_NBINS];
#pragma unroll
    for (int i = 0; i < PG_NBINS; ++i) E[i] = 0.0;
    if (dc.split == 1u) {
        const v2f64* r = (const v2f64*)((const unsigned char*)dc.frec + (size_t)c * PG_SREC1_BYTES);
        const v2f64 e01 = r[2];
        E[0] = e01.x; E[1] = e01.y; E[PG_AMAX] = r[3].x;
    } else {
        const v2f64* r = (const v2f64*)((const unsigned char*)dc.frec + (size_t)c * PG_SREC2_BYTES);
        if (nl <= 2u) { const v2f64 p0 = r[0], p2 = r[2]; E[0] = p0.x; E[1] = p0.y; E[PG_AMAX] = p2.y; }
        else {
            v2f64 pv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pv[k] = r[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) { E[2 * k] = pv[k].x; if (2 * k + 1 < PG_NBINS) E[2 * k + 1] = pv[k].y; }
        }
    }
    double acc[PG_NBINS];
#pragma unroll
    for (int i = 0; i < PG_NBINS; ++i) acc[i] = 0.0;
    bool fb;
    double scale;
    int xexp;
    if (C == 1u) {
        // A chain left with ONE column: L(g) = sum over the states of g of e(a_i, a_j) = (ordered path pairs of g) E(g) — or, if
        // that column sums to zero, the uniform column's 1 / H^2 per state (hmm.cpp:228-267, 356-368); no sweep ran
        bool anynz = false;
        static_for<0, PG_AMAX>([&](auto lac) __attribute__((always_inline)) {
            constexpr int la = decltype(lac)::value;
            static_for<la, PG_AMAX>([&](auto lbc) __attribute__((always_inline)) {
                constexpr int lb = decltype(lbc)::value;
                constexpr int bi = la * PG_AMAX - la * (la - 1) / 2 + (lb - la);
                if ((uint32_t)lb < nl) {
                    acc[bi] = (double)(cntv[la] * cntv[lb] * (la == lb ? 1u : 2u));
                    if (E[bi] > 0.0 || E[bi] != E[bi]) anynz = true;
                }
            });
        });
        if (flags & PG_SREC_FLAG_PRECISE) anynz = true;
        fb = !anynz;
        scale = fb ? 1.0 / ((double)dc.H * (double)dc.H) : 1.0;
        xexp = 0;
    } else {
        fb = dc.fwd_fallback[c] != 0;
        const bool reform = fb && c >= C / 2;
        if (reform) {
            // (see bins_unit) alpha_hat * fsum = 1 / H^2 for every state, times the stored backward column
            const uint32_t H = dc.H, HP = dc.HP;
            const double unif = 1.0 / ((double)H * (double)H);
            const double* col = dc.fwd + (size_t)c * dc.col_stride;
            for (uint32_t i = 0; i < H; ++i) {
                const uint32_t a = split_path_allele(dc, c, i);
                for (uint32_t jj = 0; jj < H; ++jj) {
                    const uint32_t b = split_path_allele(dc, c, jj);
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
            acc[0] = 0.0 + p01.x;
            if (nl > 1u) { acc[1] = (0.0 + p01.y) + p23.x; acc[PG_AMAX] = 0.0 + p23.y; }
        } else {
            const v2f64* e = (const v2f64*)(dc.aux + (size_t)aux * 16u);
            v2f64 pv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pv[k] = e[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc[2 * k] = pv[k].x; if (2 * k + 1 < PG_NBINS) acc[2 * k + 1] = pv[k].y; }
        }
        scale = 1.0 / ((fb ? 1.0 : dc.fscale[c]) * dc.bscale[c]);
        xexp = -((fb ? 0 : PG_BIAS_F) + PG_BIAS_B) + (c + 1 < C ? split_exponent_of(dc, c + 1) : 0);
    }
    const double* Pm = (const double*)(dc.cprec + (size_t)c * PG_CPREC_BYTES);
    const int* Pe = (const int*)(dc.cprec + (size_t)c * PG_CPREC_BYTES + 128u);
    const bool precise = (flags & PG_SREC_FLAG_PRECISE) != 0u;
    static_for<0, PG_AMAX>([&](auto lac) __attribute__((always_inline)) {
        constexpr int la = decltype(lac)::value;
        static_for<la, PG_AMAX>([&](auto lbc) __attribute__((always_inline)) {
            constexpr int lb = decltype(lbc)::value;
            constexpr int bi = la * PG_AMAX - la * (la - 1) / 2 + (lb - la);
            if ((uint32_t)lb < nl) {
                const uint32_t sa = lsv[la], sb = lsv[lb];
                const uint64_t idx = (uint64_t)g0 + (uint64_t)sa * A - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
                double pm; int pe;
                if (fb) { pm = 0.5; pe = 1; }
                else if (precise) { pm = Pm[bi]; pe = Pe[bi]; }
                else pair_from_entry(E[bi], X, pm, pe);
                store_bin(dc.lik, dc.lik_exp, idx, acc[bi] * scale, pm, pe, xexp);
            }
        });
    });
}

// WIDE columns of split chains: one wave per entry of the index's list of wide columns — the column this role's phase 2 put
// into the aux slot times the stored partner column (post_ab, as k_bins_wide / k_post do it).
__global__ __launch_bounds__(256) void k_bins_wide_s(const DevContig* __restrict__ contigs) {
    __shared__ double s_bins[4][PG_AMAX * (PG_AMAX + 1) / 2];
    __shared__ double s_wide[4][PG_WIDE_LDS_BINS];
    const DevContig& dc = contigs[blockIdx.y];
    if (dc.split != 2u || !dc.wcols) return;
    const uint32_t C = *dc.n_cols;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t id = blockIdx.x * 4u + wave;
    if (id >= *dc.n_wcols) return;
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)dc.wcols[id]);
    const IxBin* ixb = (const IxBin*)(dc.ix_bin + (size_t)c * PG_IXBIN_BYTES);
    if (C == 1u) {
        // the chain's only column (see k_bins_s): L(g) = (ordered path pairs of g) E(g), or 1 / H^2 per state if the column sums to zero
        const uint32_t nl = ixb->nl, WS = nl + 1u, v = ixb->v, H = dc.H;
        const unsigned char* al = (const unsigned char*)dc.frec + (size_t)c * PG_SREC2_BYTES;
        const unsigned char* went = dc.wide + (size_t)(*(const uint32_t*)(dc.ix_rec + (size_t)c * PG_IXREC_BYTES + 4u)) * 16u;
        const double* Pm = (const double*)(went + PG_WIDE_OFF_PM(WS));
        const int* Pe = (const int*)(went + PG_WIDE_OFF_PE(WS));
        const uint16_t* wslots = (const uint16_t*)(went + PG_WIDE_OFF_SLOT(WS));
        const uint32_t a0v = dc.allele_off[v], Av = dc.allele_off[v + 1] - a0v;
        const uint32_t npairs = nl * (nl + 1u) / 2u;
        bool nz = false;
        for (uint32_t q = lane; q < npairs; q += 64) {
            uint32_t la, lb;
            decode_pair(q, nl, la, lb);
            const double m = Pm[la * WS + lb];
            nz = nz || m > 0.0 || m != m;
        }
        const bool fb = __any(nz) == 0;
        for (uint32_t q = lane; q < npairs; q += 64) {
            uint32_t la, lb;
            decode_pair(q, nl, la, lb);
            uint32_t na = 0, nb = 0;
            for (uint32_t pth = 0; pth < H; ++pth) { na += al[pth] == la ? 1u : 0u; nb += al[pth] == lb ? 1u : 0u; }
            const double cntp = (double)(na * nb * (la == lb ? 1u : 2u));
            const uint32_t sa = wslots[la], sb = wslots[lb];
            const uint64_t gi = dc.geno_off[v] + (uint64_t)sa * Av - (uint64_t)sa * (sa - 1) / 2 + (sb - sa);
            store_bin(dc.lik, dc.lik_exp, gi, fb ? cntp / ((double)H * (double)H) : cntp, fb ? 0.5 : Pm[la * WS + lb], fb ? 1 : Pe[la * WS + lb], 0);
        }
        return;
    }
    const size_t colsz = (size_t)dc.HP * dc.HP;
    const double* mine = (const double*)(dc.aux + (size_t)ixb->aux * 16u);   // what this column's phase-2 role stored
    const double* stored = dc.fwd + (size_t)c * colsz;                       // its partner, from phase 1
    if (c >= C / 2) post_ab(dc, C, c, mine, stored, lane, s_bins[wave], s_wide[wave]);
    else post_ab(dc, C, c, stored, mine, lane, s_bins[wave], s_wide[wave]);
}
