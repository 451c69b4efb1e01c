// pg_small16x.h — k_sweep_small16x: the four-half-chains-per-wave step (k_sweep_small16) for 16-path chains whose objects are
// NOT all biallelic: the default production shape of every panel with more than 100 haplotypes — 15 sampled paths + the
// reference path (reference src/commands.cpp:799-803, src/haplotypesampler.cpp:43), bubbles keeping every allele those
// paths carry (src/multiallelicuniquekmers.cpp:195-232).  Included by pg_kernels.hip behind k_sweep_small16 (its helpers).
//
// One DPP row of 16 lanes per half-chain, lane = column, the sixteen rows of the column in the lane's registers: the
// recursion is k_sweep_small16's (in-lane column sums, u_i by v_fmac_f64_dpp row_newbcast, total by four rotate-and-add
// steps).  What differs is everything that depends on the alleles — ONE path for every column, eligibility is per COLUMN,
// not per chain or job ("what's missing" 1-2):
//   * records: 192 bytes per column (k_records, DevContig::smallx): the fifteen entries E(a, b), a <= b, of the column's
//     (symmetric) emission table, a 16-byte header (local alleles, flags, wide entry, aux slot), the transition constants,
//     the table-row offset (a + 1) * 48 of every path's allele.  Lane j fetches entry j (8 bytes) and piece 8 + (j & 3)
//     three columns ahead (two global loads per lane and column where the biallelic kernel issues four) and parks them in
//     the half-chain's LDS slot one step before the column is due — the entry at [a][b] and [b][a] of the slot's 6 x 6 table;
//     constants, header and row offsets come back as broadcast LDS reads a step ahead of use.  (Round 5's first version
//     carried the five table rows and the raw alleles: 320 bytes, an eighth of the sweep's traffic.)
//   * emission of a state = E[a_k][a_j] = one v_add_u32_sdwa (the row's byte + the lane's table column) + one ds_read_b64
//     (two issue slots; the biallelic select takes three), fetched during the step before it is used.  Row offset 0 = a row
//     of zeros in front of the table: phantom alleles and the columns below.
//   * phase 2: P' beta' is added by ROW allele into five accumulators with exact 0 / 1 weights that are 16-byte LDS reads
//     from a constant one-hot table addressed by the same row byte (ten issue slots per state, no compare, no select).
//     Columns with at most two local alleles leave the sweep as the four class sums of DevContig::cls4's layout; columns
//     with three to five as their (up to) fifteen genotype bins, formed inside the row through an LDS transpose, in the
//     column's 128-byte aux slot (k_bins_x finishes both kinds).
//   * WIDE columns (more than PG_AMAX alleles on the sixteen paths) cost that COLUMN, not the job: their emissions are
//     gathered from the side table (a wave-uniform branch, 16 loads per lane of the rows concerned), and in phase 2 the
//     column itself goes to its aux slot — k_bins_wide forms its bins from the two stored columns the way k_post does.
// Same stored columns, scales, fall-back rules and resume conventions as every other sweep kernel.
#pragma once

#define PG_XREC_BYTES 192u          // per column: 15 table entries (tri_local order; a WIDE column: its sixteen raw alleles) + 8 bytes, then
#define PG_XREC_HDR 128u            //   header {nlocal | flags << 8, wide entry / 16, aux slot / 16, 0},
#define PG_XREC_CONSTS 144u         //   {c0, c1}, {c2, kappa},
#define PG_XREC_ROWOFF 176u         //   the table-row offsets of the sixteen paths' alleles
#define PG_XREC_FLAG_WIDE 0x200u    // header dword 0: nlocal | flags << 8 (PG_REC_FLAG_* << 8)
// LDS slot of one record: a row of zeros, the 6 x 6 table (rows at (a + 1) * 48), then the four 16-byte pieces
#define PG_XSLOT_TABLE 48u
#define PG_XSLOT_HDR 288u
#define PG_XSLOT_CONSTS 304u
#define PG_XSLOT_ROWOFF 336u
#define PG_XSLOT_BYTES 352u

typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));   // native vector type (address-space qualifiable)
struct SmallXShared {
    double onehot[PG_ESTRIDE][PG_ESTRIDE];                 // row 0: zeros; row a + 1: 1.0 at a (a < PG_AMAX) — the row-allele weights
    double red[4][PG_ESTRIDE][16];                         // phase 2: a multiallelic column's accumulators [half-chain][row allele][lane]; row PG_AMAX: zeros
    unsigned char rec[4][2][PG_XSLOT_BYTES] __attribute__((aligned(16)));   // [half-chain of the wave][column parity]
};
struct XPieces { double p0; v2f64 p1; };
// table entry p (tri_local order: rows of 5, 4, 3, 2, 1 entries) = E(a, b), a <= b
DEVI void x_pair_of(uint32_t p, uint32_t& a, uint32_t& b) {
    a = (p >= 5u ? 1u : 0u) + (p >= 9u ? 1u : 0u) + (p >= 12u ? 1u : 0u) + (p >= 14u ? 1u : 0u);
    b = p - (a * (uint32_t)PG_AMAX - a * (a - 1u) / 2u) + a;
}
struct XConsts { double c0, c1, c2, kappa; };
struct XCol {            // what a step needs of a column's record, read from LDS a step ahead
    uint32_t ro[4];      // row offsets of the sixteen paths' alleles (bytes)
    uint32_t ecol;       // LDS address of the lane's table column: slot + 8 * min(a_j, 5)
    uint32_t nlf;        // nlocal | flags << 8
    uint32_t widx, aux;  // wide entry offset / 16, aux slot offset / 16
    uint32_t roj;        // the lane's own row offset: (its allele + 1) * 48, 0 = phantom / wide column
    uint32_t slot;       // LDS address of the record's slot
};

DEVI XPieces load_xrec(gcdouble* xrec, int64_t c, int64_t C, uint32_t j) {   // (clamped: records past the end are never used)
    c = c < 0 ? 0 : (c >= C ? C - 1 : c);
    const GAS char* b = (const GAS char*)xrec + (size_t)c * PG_XREC_BYTES;
    XPieces r;
    r.p0 = *(const GAS double*)(b + 8u * j);
    r.p1 = *(const GAS v2f64*)(b + PG_XREC_HDR + 16u * (j & 3u));
    return r;
}
// tab_ab / tab_ba: where this lane's table entry goes inside a slot — [a][b] and [b][a] (lane 15: none)
DEVI void park_xrec(uint32_t slot, uint32_t j, uint32_t tab_ab, uint32_t tab_ba, const XPieces& r) {
    if (j < 15u) {
        *(LAS double*)(uintptr_t)(slot + tab_ab) = r.p0;
        *(LAS double*)(uintptr_t)(slot + tab_ba) = r.p0;   // (a == b: the same place again)
    }
    if (j < 4u) *(LAS v2f64*)(uintptr_t)(slot + PG_XSLOT_HDR + 16u * j) = r.p1;
}
DEVI XConsts read_xconsts(uint32_t slot) {
    const v2f64 a = *(LAS const v2f64*)(uintptr_t)(slot + PG_XSLOT_CONSTS), b = *(LAS const v2f64*)(uintptr_t)(slot + PG_XSLOT_CONSTS + 16u);
    return XConsts{a.x, a.y, b.x, b.y};
}
DEVI XCol read_xcol(uint32_t slot, uint32_t j) {
    XCol c;
    const v4u32 ro = *(LAS const v4u32*)(uintptr_t)(slot + PG_XSLOT_ROWOFF);
    const v4u32 hd = *(LAS const v4u32*)(uintptr_t)(slot + PG_XSLOT_HDR);
    const uint32_t roj = *(LAS const unsigned char*)(uintptr_t)(slot + PG_XSLOT_ROWOFF + j);
    c.ro[0] = ro.x; c.ro[1] = ro.y; c.ro[2] = ro.z; c.ro[3] = ro.w;
    c.nlf = hd.x; c.widx = hd.y; c.aux = hd.z;
    c.roj = roj;
    // the lane's table column: 8 * its allele = roj / 6 - 8 ((roj * 171) >> 10 is roj / 6 for the six values roj takes); phantom: column 5
    c.ecol = slot + (roj ? ((roj * 171u) >> 10) - 8u : 8u * (uint32_t)PG_AMAX);
    c.slot = slot;
    return c;
}
template <int K>
DEVI double x_emission(const XCol& c) { return *(LAS const double*)(uintptr_t)add_byte<(K & 3)>(c.ro[K >> 2], c.ecol); }

// the emissions of a WIDE column's rows from the side table: E[a_k][a_j], S = nlocal + 1 doubles per row (pg_device.h);
// the sixteen raw local alleles of such a column travel where the (unused) table entries 0 and 1 do: slot bytes 48 .. 63
DEVI void x_wide_emissions(const XCol& c, const unsigned char* wide, bool mine, uint32_t j, double (&ee)[16]) {
    const v4u32 rw = *(LAS const v4u32*)(uintptr_t)(c.slot + PG_XSLOT_TABLE);
    const uint32_t rawj = *(LAS const unsigned char*)(uintptr_t)(c.slot + PG_XSLOT_TABLE + j);
    const uint32_t raw[4] = {rw.x, rw.y, rw.z, rw.w};
    if (mine) {
        const uint32_t S = (c.nlf & 0xFFu) + 1u;
        gcdouble* Ew = (gcdouble*)(wide + (size_t)c.widx * 16u);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t ak = (raw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            ee[k] = Ew[(size_t)ak * S + rawj];
        }
    }
}

struct SmallXCtx {     // per lane: the half-chain of its DPP row
    bool live;
    gcdouble* xrec; gdouble* wr; gcdouble* resume; gdouble* sc_a; gdouble* sc_b; gu8* fallback;
    gcdouble* partner; gdouble* part; GAS char* aux; const unsigned char* wide;
    int64_t C, lo, hi;
};

DEVI void smallx_init_shared(SmallXShared& sh, uint32_t lane) {
    if (lane < (uint32_t)PG_ETAB) {
        const uint32_t r = lane / (uint32_t)PG_ESTRIDE, a = lane % (uint32_t)PG_ESTRIDE;
        sh.onehot[r][a] = (r > 0u && a == r - 1u && a < (uint32_t)PG_AMAX) ? 1.0 : 0.0;
    }
    // every record slot starts as zeros: the row of zeros in front of each table stays (nothing parks there), and the rows
    // of the wave that carry no half-chain read zero offsets, not whatever the LDS held (8 slots x 352 bytes = 176 pieces);
    // column 5 of every table row stays zero too (the entries parked cover [a][b], a, b < 5)
#pragma unroll
    for (uint32_t q = 0; q < 3u; ++q) if (lane + 64u * q < 8u * PG_XSLOT_BYTES / 16u) *(v2f64*)(&sh.rec[0][0][0] + (lane + 64u * q) * 16u) = v2f64{0.0, 0.0};
    sh.red[lane >> 4][PG_AMAX][lane & 15u] = 0.0;
    wave_sync_lds();
}

// phase 2, one column of one half-chain: the posterior sums leave the sweep (see the header)
//   acc[a] = sum over the rows with allele a of P'(k, j) beta'(k, j), this lane's column j
// Columns with at most two local alleles: four class sums (row allele x column allele) by masked sums over the row's lanes.
// Columns with three to five: the sixteen lanes' accumulators are transposed through LDS and lane p of the row adds up
// genotype bin p = {a, b} (tri_local order: the accumulators a of the lanes whose column carries b, and b of those that
// carry a) — 120 bytes leave for the column's aux slot where 768 bytes of accumulators did (a wave-uniform branch: in a
// cohort the four rows of a wave are samples of one contig, the same columns are multiallelic in all of them).
DEVI void smallx_posterior_out(const SmallXCtx& cx, SmallXShared& sh, const XCol& col, bool act, int64_t t, uint32_t j, uint32_t row,
                               const double (&acc)[PG_AMAX], const double (&colv)[16]) {
    const uint32_t nl = col.nlf & 0xFFu;
    const bool wide = (col.nlf & PG_XREC_FLAG_WIDE) != 0u;
    const bool b1 = col.roj == 2u * (uint32_t)(PG_ESTRIDE * 8);   // the lane's column carries local allele 1
    const double s00 = row16_sum(b1 ? 0.0 : acc[0]), s01 = row16_sum(b1 ? acc[0] : 0.0);
    const double s10 = row16_sum(b1 ? 0.0 : acc[1]), s11 = row16_sum(b1 ? acc[1] : 0.0);
    if (act && !wide && nl <= 2u && j == 0) {
        gdouble2* o = (gdouble2*)cx.part + (size_t)t * 2u;
        o[0] = v2f64{s00, s01};
        o[1] = v2f64{s10, s11};
    }
    const bool multi = act && !wide && nl > 2u;
    if (__any(multi)) {
        const uint32_t red = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.red[row][0][0];
#pragma unroll
        for (int a = 0; a < PG_AMAX; ++a) *(LAS double*)(uintptr_t)(red + (uint32_t)a * 128u + 8u * j) = acc[a];
        // this lane's bin p = j: {pa, pb}, pa <= pb (tri_local: rows of 5, 4, 3, 2, 1 entries)
        uint32_t pa, pb;
        x_pair_of(j, pa, pb);
        const uint32_t offa = red + pa * 128u, offb = red + pb * 128u, offz = red + (uint32_t)PG_AMAX * 128u;
        const uint32_t roa = (pa + 1u) * (uint32_t)(PG_ESTRIDE * 8), rob = (pb + 1u) * (uint32_t)(PG_ESTRIDE * 8);
        double sum = 0.0;
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            const uint32_t al = (col.ro[l >> 2] >> (8 * (l & 3))) & 0xFFu;   // row offset of the column allele of lane l
            const uint32_t src = al == rob ? offa : (al == roa ? offb : offz);
            sum += *(LAS const double*)(uintptr_t)(src + 8u * (uint32_t)l);
        }
        if (multi && j < (uint32_t)(PG_AMAX * (PG_AMAX + 1) / 2)) *((gdouble*)(cx.aux + (size_t)col.aux * 16u) + j) = sum;
    }
    if (act && wide) {   // the column itself (row-pair layout, like a stored column): k_bins_wide multiplies it with its partner
        gdouble2* o = (gdouble2*)(cx.aux + (size_t)col.aux * 16u) + j;
#pragma unroll
        for (int k = 0; k < 16; k += 2) o[(size_t)(k >> 1) * 16] = v2f64{colv[k], colv[k + 1]};
    }
}

template <int PHASE>
DEVI void small16x_forward(const DevContig* contigs, const uint32_t* ids, uint32_t n_ids, uint32_t chunk, double* dump, SmallXShared& sh) {
    constexpr int HP = 16, R = 16;
    // Records in flight (steps between a record's loads and its parking).  The store-only phases keep SIX: a wave's loads and
    // stores share one in-order counter, so waiting for a record caps the stores the wave may have in flight at the number
    // issued since — at three steps (24 operations) that cap cost the store-bound phase 1 a third of its time
    // (profiles/r05_small16x_ablation.txt: 14.6 ms, 9.9 without the wait).  Phase 2 waits for its partner columns anyway.
    constexpr int D = PHASE == 2 ? 3 : 6;
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u, row = lane >> 4;
    const uint32_t slot_id = blockIdx.x * 4u + row;
    const size_t colsz = (size_t)HP * HP;
    const double unif = 1.0 / 256.0;
    SmallXCtx cx{};
    uint32_t first = 1;
    if (slot_id < n_ids) {
        const DevContig& dc = contigs[ids[slot_id]];
        const uint32_t C = *dc.n_cols, mid = C / 2, K = dc.chunk_cols;
        uint32_t lo = PHASE == 1 ? 0u : mid, hi = PHASE == 1 ? mid : C;
        bool ok = C > 0;
        if constexpr (PHASE == 2) ok = dc.smallx == 2u && C >= 2u;   // (a chain left with a single column: the general kernel)
        if constexpr (PHASE == 3) {
            const unsigned long long l = (unsigned long long)mid + (unsigned long long)chunk * K;
            ok = ok && l < C;
            lo = ok ? (uint32_t)l : 0u;
            hi = ok ? (C - lo > K ? lo + K : C) : 0u;
        }
        ok = ok && lo < hi;
        if (ok) {
            cx.live = true; cx.C = C; cx.lo = lo; cx.hi = hi;
            cx.xrec = (gcdouble*)dc.frec; cx.sc_a = (gdouble*)dc.fscale; cx.fallback = (gu8*)dc.fwd_fallback;
            gdouble* fwd = (gdouble*)dc.fwd;
            cx.wr = fwd; cx.partner = (gcdouble*)fwd; cx.part = (gdouble*)dc.part; cx.aux = (GAS char*)dc.aux; cx.wide = dc.wide;
            cx.resume = (gcdouble*)(fwd + (size_t)(lo > 0 ? lo - 1 : 0) * colsz);
            if constexpr (PHASE == 3) {
                gdouble* scr = (gdouble*)dc.scratch;
                cx.wr = scr + (size_t)((chunk & 1u) * 2u) * K * colsz - (size_t)lo * colsz;
                if (chunk > 0) cx.resume = (gcdouble*)(scr + ((size_t)(((chunk - 1u) & 1u) * 2u) * K + (K - 1u)) * colsz);
            }
            first = lo == 0 ? 1u : lo;
        }
    }
    const int n_steps = __builtin_amdgcn_readfirstlane(wave_max_i32(cx.live ? (int)(cx.hi - (int64_t)first) : 0));   // (uniform) the longest of the four
    if (__builtin_amdgcn_readfirstlane(wave_max_i32(cx.live ? 1 : 0)) == 0) return;
    smallx_init_shared(sh, lane);
    const uint32_t slot0 = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.rec[row][0][0];
    const uint32_t onehot = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.onehot[0][0];
    auto slot_of = [&](int64_t c) { return slot0 + ((uint32_t)c & 1u) * PG_XSLOT_BYTES; };
    uint32_t tab_ab, tab_ba;
    {
        uint32_t pa, pb;
        x_pair_of(j < 15u ? j : 14u, pa, pb);
        tab_ab = PG_XSLOT_TABLE + pa * (uint32_t)(PG_ESTRIDE * 8) + pb * 8u;
        tab_ba = PG_XSLOT_TABLE + pb * (uint32_t)(PG_ESTRIDE * 8) + pa * 8u;
    }
    auto emissions = [&](const XCol& col, double (&ee)[R]) __attribute__((always_inline)) {
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; ee[k] = x_emission<k>(col); });
        const bool wide = cx.live && (col.nlf & PG_XREC_FLAG_WIDE) != 0u;
        if (__any(wide)) {   // (uniform) some row's column is wide: its sixteen emissions from the side table
            x_wide_emissions(col, cx.wide, wide, j, ee);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        }
    };
    auto store_col = [&](gdouble* base, const double (&v)[R]) {
        gdouble2* dst = (gdouble2*)base + j;
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
    // (lanes of rows whose column cprev summed to zero; pcol = that column's record)
    auto flag_uniform = [&](int64_t cprev, const XCol& pcol) {
        double xu[R];
#pragma unroll
        for (int k = 0; k < R; ++k) xu[k] = unif;
        if (PHASE != 2 && cprev >= cx.lo) store_col(cx.wr + (size_t)cprev * colsz, xu);
        // phase 2: the bins of column cprev are re-formed from the stored backward column (k_bins_x); a WIDE column's come
        // from the two stored columns, so the uniform column takes the place of the one this role put into the aux slot
        if (PHASE == 2 && cprev >= cx.lo && (pcol.nlf & PG_XREC_FLAG_WIDE)) store_col((gdouble*)(cx.aux + (size_t)pcol.aux * 16u), xu);
        if (j == 0) cx.fallback[cprev] = 1;
    };
    // the column before the first step: x = e (.) P'
    double x[R], buf = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) x[k] = 0.0;
    XCol prevcol{};
    if (cx.live) park_xrec(slot_of((int64_t)first - 1), j, tab_ab, tab_ba, load_xrec(cx.xrec, (int64_t)first - 1, cx.C, j));
    if (cx.live) park_xrec(slot_of((int64_t)first), j, tab_ab, tab_ba, load_xrec(cx.xrec, (int64_t)first, cx.C, j));
    {
        const XCol c0 = read_xcol(slot_of((int64_t)first - 1), j);
        double e0[R];
        emissions(c0, e0);
        prevcol = c0;
        if (cx.live) {
            if (cx.lo == 0) {
                const double P0 = ldexp(1.0, PG_BIAS_F);
                double pz[R];
#pragma unroll
                for (int k = 0; k < R; ++k) { pz[k] = P0; x[k] = e0[k] * P0; }
                store_col(cx.wr, pz);
                if (j == 0) cx.sc_a[0] = 1.0;
            } else {
                gcdouble2* src = (gcdouble2*)cx.resume + j;
#pragma unroll
                for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; x[k] = t.x; x[k + 1] = t.y; }
                if (!cx.fallback[cx.lo - 1]) {
#pragma unroll
                    for (int k = 0; k < R; ++k) x[k] *= e0[k];
                }
            }
        }
    }
    XConsts cur = read_xconsts(slot_of((int64_t)first));   // column `first`: constants of the gap first-1 -> first
    XCol col = read_xcol(slot_of((int64_t)first), j);      // ... its alleles, header, table
    double ee[R], pp[R];   // x = ee pp (formed at the start of the next step, see lean_forward)
#pragma unroll
    for (int k = 0; k < R; ++k) { ee[k] = 1.0; pp[k] = x[k]; }
    // One column step of the (up to) four half-chains.  `slot_rec` holds the pieces of column t + 1 (parked now, read back at
    // the end of the step) and then takes those of column t + 1 + D.  Rows whose half-chain is done (or absent) compute on
    // whatever their registers hold and store nothing.
    auto step = [&](int n, XPieces& slot_rec, double (&vp)[PHASE == 2 ? R : 1]) __attribute__((always_inline)) {
        const int64_t t = (int64_t)first + n;
        const bool act = cx.live && t < cx.hi;
        if (!(kXExp & 2u)) park_xrec(slot_of(t + 1), j, tab_ab, tab_ba, slot_rec);
        if (cx.live && !(kXExp & 4u)) slot_rec = load_xrec(cx.xrec, t + 1 + D, cx.C, j);
        double Cj = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; x[k] = ee[k] * pp[k]; Cj += x[k]; });
        double S = row16_sum(Cj);
        // a row that did its last column a step ago: that column may itself have summed to zero
        if (cx.live && t == cx.hi && !(S > 0.0)) flag_uniform(t - 1, prevcol);
        const double ucol = dpp_source(cur.c1 * Cj);   // u_i of row i = this lane's column (the column is symmetric)
        double uj = fma(cur.c2, S, ucol);
        double c0 = cur.c0;
        if (act && !(S > 0.0)) {
            // column t-1 summed to zero: the uniform column takes its place (hmm.cpp:253-267), see lean_forward
            flag_uniform(t - 1, prevcol);
            const double Cu = 16.0 * unif;
            S = 1.0;
            uj = fma(cur.c0, unif, fma(cur.c2, 1.0, 2.0 * cur.c1 * Cu));
            c0 = 0.0;
        }
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        const double m = ldexp(S, -es - PG_BIAS_F);
        const double sc = ldexp(1.0, -es), c0s = ldexp(c0, -es), ujs = ldexp(uj, -es);
        gdouble2* dst = act ? (gdouble2*)(cx.wr + (size_t)t * colsz) + j : (gdouble2*)dump + lane;
        double pprev = 0.0, acc[PG_AMAX];
#pragma unroll
        for (int a = 0; a < PG_AMAX; ++a) acc[a] = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double pk = fmac_row_bcast<k>(fma(c0s, x[k], ujs), ucol, sc);   // P'_t(k, j) 2^-es = c0 x + u_j + u_k
            ee[k] = (kXExp & 1u) ? 1.0 : x_emission<k>(col);
            pp[k] = pk;
            if constexpr (PHASE == 2) {
                // posterior: P'_t beta'_t added by row allele — the weights are the one-hot row of the row's allele (exact 0 / 1)
                const double pr = vp[k] * pk;
                if constexpr (!(kXExp & 16u)) {
                const uint32_t wa = add_byte<(k & 3)>(col.ro[k >> 2], onehot);
                const v2f64 w01 = *(LAS const v2f64*)(uintptr_t)wa, w23 = *(LAS const v2f64*)(uintptr_t)(wa + 16u);
                const double w4 = *(LAS const double*)(uintptr_t)(wa + 32u);
                acc[0] = fma(pr, w01.x, acc[0]); acc[1] = fma(pr, w01.y, acc[1]);
                acc[2] = fma(pr, w23.x, acc[2]); acc[3] = fma(pr, w23.y, acc[3]);
                acc[4] = fma(pr, w4, acc[4]);
                } else acc[0] += pr;
                if constexpr ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (the weight reads of at most four states in flight: hoisted together they took 160 registers)
            } else {
                if constexpr (k & 1) { if (!(kXExp & 8u)) dst[(size_t)(k >> 1) * HP] = v2f64{pprev, pk}; else asm volatile("" :: "v"(pprev), "v"(pk)); }
                else pprev = pk;
            }
        });
        {
            const bool wide = act && (col.nlf & PG_XREC_FLAG_WIDE) != 0u;
            if (__any(wide)) {
                x_wide_emissions(col, cx.wide, wide, j, ee);
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
        }
        if constexpr (PHASE == 2) {
            if (!(kXExp & 32u)) smallx_posterior_out(cx, sh, col, act, t, j, row, acc, pp);
            else if (act && j == 0 && acc[0] + acc[1] + acc[2] + acc[3] + acc[4] == 1.2345) cx.part[t] = 0.0;
            if (!(kXExp & 64u)) load_partner(t + 3, vp);
        }
        if (act) {   // the column's scale mantissa: sixteen columns collected in the row's lanes, one store per sixteen
            if (j == ((uint32_t)t & 15u)) buf = m;
            if (((uint32_t)t & 15u) == 15u || t + 1 == cx.hi) { if (j <= ((uint32_t)t & 15u) && (int64_t)((t & ~15ll) + j) >= (int64_t)first) cx.sc_a[(t & ~15ll) + j] = buf; }
        }
        prevcol = col;
        if (!(kXExp & 2u)) {
            cur = read_xconsts(slot_of(t + 1));
            col = read_xcol(slot_of(t + 1), j);
        }
    };
    XPieces rr[D];
    static_for<0, D>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; rr[i] = cx.live ? load_xrec(cx.xrec, (int64_t)first + 1 + i, cx.C, j) : XPieces{}; });
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
        if (cx.live && (int64_t)first + n_steps == cx.hi && !(Sl > 0.0)) flag_uniform(cx.hi - 1, prevcol);
    }
}

template <int PHASE>
DEVI void small16x_backward(const DevContig* contigs, const uint32_t* ids, uint32_t n_ids, uint32_t chunk, double* dump, SmallXShared& sh) {
    constexpr int HP = 16, R = 16;
    // Records in flight (steps between a record's loads and its parking).  The store-only phases keep SIX: a wave's loads and
    // stores share one in-order counter, so waiting for a record caps the stores the wave may have in flight at the number
    // issued since — at three steps (24 operations) that cap cost the store-bound phase 1 a third of its time
    // (profiles/r05_small16x_ablation.txt: 14.6 ms, 9.9 without the wait).  Phase 2 waits for its partner columns anyway.
    constexpr int D = PHASE == 2 ? 3 : 6;
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u, row = lane >> 4;
    const uint32_t slot_id = blockIdx.x * 4u + row;
    const size_t colsz = (size_t)HP * HP;
    const double unif = 1.0 / 256.0;
    SmallXCtx cx{};   // lo = bot, hi = top
    int64_t t0 = -1;
    double Sy = 1.0;
    double ee[R], pp[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { ee[k] = 1.0; pp[k] = 0.0; }
    if (slot_id < n_ids) {
        const DevContig& dc = contigs[ids[slot_id]];
        const int64_t C = *dc.n_cols, mid = C / 2, K = dc.chunk_cols;
        int64_t top = PHASE == 1 ? C - 1 : mid - 1, bot = PHASE == 1 ? mid : 0;
        bool ok = C > 0;
        if constexpr (PHASE == 2) ok = dc.smallx == 2u && C >= 2;
        if constexpr (PHASE == 3) {
            top = mid - 1 - (int64_t)chunk * K;
            ok = ok && top >= 0;
            bot = top - K + 1 > 0 ? top - K + 1 : 0;
        }
        ok = ok && top >= bot;
        if (ok) {
            cx.live = true; cx.C = C; cx.lo = bot; cx.hi = top;
            cx.xrec = (gcdouble*)dc.frec; cx.sc_a = (gdouble*)dc.bscale; cx.sc_b = (gdouble*)dc.bsum;
            gdouble* cols = (gdouble*)dc.fwd;
            cx.wr = cols; cx.partner = (gcdouble*)cols; cx.part = (gdouble*)dc.part; cx.aux = (GAS char*)dc.aux; cx.wide = dc.wide;
            cx.resume = (gcdouble*)(cols + (size_t)(top + 1 < C ? top + 1 : top) * colsz);
            if constexpr (PHASE == 3) {
                gdouble* scr = (gdouble*)dc.scratch;
                cx.wr = scr + (size_t)((chunk & 1u) * 2u + 1u) * (size_t)K * colsz - (size_t)bot * colsz;
                if (chunk > 0) cx.resume = (gcdouble*)(scr + (size_t)(((chunk - 1u) & 1u) * 2u + 1u) * (size_t)K * colsz);
            }
            t0 = PHASE == 1 ? top - 1 : top;
        }
    }
    const int n_steps = __builtin_amdgcn_readfirstlane(wave_max_i32(cx.live ? (int)(t0 - cx.lo + 1) : 0));
    if (__builtin_amdgcn_readfirstlane(wave_max_i32(cx.live ? 1 : 0)) == 0) return;
    smallx_init_shared(sh, lane);
    const uint32_t slot0 = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.rec[row][0][0];
    const uint32_t onehot = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.onehot[0][0];
    auto slot_of = [&](int64_t c) { return slot0 + ((uint32_t)c & 1u) * PG_XSLOT_BYTES; };
    uint32_t tab_ab, tab_ba;
    {
        uint32_t pa, pb;
        x_pair_of(j < 15u ? j : 14u, pa, pb);
        tab_ab = PG_XSLOT_TABLE + pa * (uint32_t)(PG_ESTRIDE * 8) + pb * 8u;
        tab_ba = PG_XSLOT_TABLE + pb * (uint32_t)(PG_ESTRIDE * 8) + pa * 8u;
    }
    auto emissions = [&](const XCol& col, bool on, double (&e)[R]) __attribute__((always_inline)) {
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; e[k] = x_emission<k>(col); });
        const bool wide = on && (col.nlf & PG_XREC_FLAG_WIDE) != 0u;
        if (__any(wide)) {
            x_wide_emissions(col, cx.wide, wide, j, e);
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    };
    // records: column t0 + 1 (its emission enters the first step, its constants are those of the gap t0 -> t0 + 1) and
    // column t0 (alleles / table of the first step's own column)
    if (cx.live) park_xrec(slot_of(t0 + 1), j, tab_ab, tab_ba, load_xrec(cx.xrec, t0 + 1, cx.C, j));
    if (cx.live) park_xrec(slot_of(t0), j, tab_ab, tab_ba, load_xrec(cx.xrec, t0, cx.C, j));
    {
        const XCol c1 = read_xcol(slot_of(t0 + 1), j);
        double e1[R];
        emissions(c1, cx.live, e1);
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
#pragma unroll
            for (int k = 0; k < R; ++k) { ee[k] = e1[k]; pp[k] = y[k]; }
        }
    }
    XConsts cur = read_xconsts(slot_of(t0 + 1));   // constants of the gap t0 -> t0 + 1
    XCol col = read_xcol(slot_of(t0), j);          // column t0: its alleles, header, table
    double one = 1.0, bufA = 0.0, bufB = 0.0;
    asm volatile("" : "+v"(one));
    // phase 2: the partner column P'_t of this lane's column, fetched three steps ahead (clamped: see small16x_forward)
    auto load_partner = [&](int64_t c, double (&v)[R]) {
        if (!cx.live) return;
        c = c < cx.lo ? cx.lo : (c > cx.hi ? cx.hi : c);
        gcdouble2* src = (gcdouble2*)(cx.partner + (size_t)c * colsz) + j;
#pragma unroll
        for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; v[k] = t.x; v[k + 1] = t.y; }
    };
    // step n: column t = t0 - n.  `cur` = constants of record t + 1, `col` = record t; `slot_rec` holds the pieces of record
    // t - 1 (parked now into the slot record t + 1 leaves) and then takes those of record t - 1 - D.
    auto step = [&](int n, XPieces& slot_rec, double (&vp)[PHASE == 2 ? R : 1]) __attribute__((always_inline)) {
        const int64_t t = t0 - n;
        const bool act = cx.live && t >= cx.lo;
        if (!(kXExp & 2u)) park_xrec(slot_of(t - 1), j, tab_ab, tab_ba, slot_rec);
        if (cx.live && !(kXExp & 4u)) slot_rec = load_xrec(cx.xrec, t - 1 - D, cx.C, j);
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
        gdouble2* dst = act ? (gdouble2*)(cx.wr + (size_t)t * colsz) + j : (gdouble2*)dump + lane;
        double yprev = 0.0, acc[PG_AMAX];
#pragma unroll
        for (int a = 0; a < PG_AMAX; ++a) acc[a] = 0.0;
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            const double yk = fmac_row_bcast<k>(fma(k0, w[k], uj), ucol, one);  // beta'_t = k0 w + u_j + u_k
            ee[k] = (kXExp & 1u) ? 1.0 : x_emission<k>(col);
            pp[k] = yk;
            if constexpr (PHASE == 2) {   // (see small16x_forward)
                const double pr = vp[k] * yk;
                if constexpr (!(kXExp & 16u)) {
                const uint32_t wa = add_byte<(k & 3)>(col.ro[k >> 2], onehot);
                const v2f64 w01 = *(LAS const v2f64*)(uintptr_t)wa, w23 = *(LAS const v2f64*)(uintptr_t)(wa + 16u);
                const double w4 = *(LAS const double*)(uintptr_t)(wa + 32u);
                acc[0] = fma(pr, w01.x, acc[0]); acc[1] = fma(pr, w01.y, acc[1]);
                acc[2] = fma(pr, w23.x, acc[2]); acc[3] = fma(pr, w23.y, acc[3]);
                acc[4] = fma(pr, w4, acc[4]);
                } else acc[0] += pr;
                if constexpr ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (the weight reads of at most four states in flight: hoisted together they took 160 registers)
            } else {
                if constexpr (k & 1) { if (!(kXExp & 8u)) dst[(size_t)(k >> 1) * HP] = v2f64{yprev, yk}; else asm volatile("" :: "v"(yprev), "v"(yk)); }
                else yprev = yk;
            }
        });
        {
            const bool wide = act && (col.nlf & PG_XREC_FLAG_WIDE) != 0u;
            if (__any(wide)) {
                x_wide_emissions(col, cx.wide, wide, j, ee);
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
        }
        if constexpr (PHASE == 2) {
            if (!(kXExp & 32u)) smallx_posterior_out(cx, sh, col, act, t, j, row, acc, pp);
            else if (act && j == 0 && acc[0] + acc[1] + acc[2] + acc[3] + acc[4] == 1.2345) cx.part[t] = 0.0;
            if (!(kXExp & 64u)) load_partner(t - 3, vp);
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
        if (!(kXExp & 2u)) {
            cur = read_xconsts(slot_of(t));        // constants of the gap t - 1 -> t (record t: still in its slot)
            col = read_xcol(slot_of(t - 1), j);    // record t - 1, parked at the top of this step
        }
    };
    XPieces rr[D];
    static_for<0, D>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; rr[i] = cx.live ? load_xrec(cx.xrec, t0 - 1 - i, cx.C, j) : XPieces{}; });
    double vv[3][PHASE == 2 ? R : 1];
    if constexpr (PHASE == 2) { load_partner(t0, vv[0]); load_partner(t0 - 1, vv[1]); load_partner(t0 - 2, vv[2]); }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int n = 0;
    for (; n + D - 1 < n_steps; n += D)
        static_for<0, D>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; step(n + i, rr[i], vv[i % 3]); });
    static_for<0, D - 1>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; if (n < n_steps) { step(n, rr[i], vv[i % 3]); ++n; } });
}

template <int PHASE>
__global__ __launch_bounds__(64) void k_sweep_small16x(const DevContig* __restrict__ contigs, const uint32_t* __restrict__ ids, uint32_t n_ids, uint32_t chunk,
                                                       double* dump) {
    __shared__ SmallXShared sh;
    if (blockIdx.y == 0) small16x_forward<PHASE>(contigs, ids, n_ids, chunk, dump, sh);
    else small16x_backward<PHASE>(contigs, ids, n_ids, chunk, dump, sh);
}
