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
//     the table-row offset (a + 1) * 48 of every path's allele.  Records reach the wave in BLOCKS of eight (phase 2: six)
//     columns: the sixteen lanes of a row fetch their half-chain's 1.5 KB with six 16-byte loads once per block, a whole
//     block ahead, and park them in an LDS staging ring; every step expands the next column's fifteen entries into the
//     half-chain's 6 x 6 table slot (entry j at [a][b] and [b][a]) and reads its constants, header and row offsets straight
//     from the staging ring, a step ahead of use.  (A record fetched per step — two loads per lane and column, three to six
//     columns ahead — cost the store-bound phase 1 a third of its time, whatever the distance: 14.6 ms against 10.0 with
//     the loads compiled out, profiles/r05_small16x_ablation*.txt — a trickle of small reads into a memory system that is
//     streaming writes.  Round 5's first version also carried five table rows and the raw alleles: 320 bytes.)
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
// LDS table slot of one column: a row of zeros, then the 6 x 6 table (rows at (a + 1) * 48; a wide column: its raw alleles at 48)
#define PG_XSLOT_TABLE 48u
#define PG_XSLOT_BYTES 336u
#define PG_XBLOCK_MAX 8             // columns per record block (store-only phases: 8, phase 2: 6 — twice its partner-column rotation)

typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));   // native vector type (address-space qualifiable)
struct SmallXShared {
    double onehot[PG_ESTRIDE][PG_ESTRIDE];                 // row 0: zeros; row a + 1: 1.0 at a (a < PG_AMAX) — the row-allele weights
    double red[4][PG_ESTRIDE][16];                         // phase 2: a multiallelic column's accumulators [half-chain][row allele][lane]; row PG_AMAX: zeros
    unsigned char rec[4][2][PG_XSLOT_BYTES] __attribute__((aligned(16)));   // [half-chain of the wave][column parity]: table slots
    unsigned char stg[4][2][PG_XBLOCK_MAX * PG_XREC_BYTES] __attribute__((aligned(16)));   // [half-chain][block parity]: record blocks as they lie in memory
};
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
    uint32_t slot;       // LDS address of the column's table slot
};

// A block of BL records = BL * 12 pieces of 16 bytes, contiguous in memory (columns c0 .. c0 + BL - 1 of the half-chain's record
// array, clamped to the array: a block may reach past either end — what is fetched there is never used): lane j of the row
// takes the pieces j, j + 16, ... — NQ loads per lane and block.
template <int BL>
struct XBlock {
    static constexpr int NP = BL * 12, NQ = (NP + 15) / 16;
    v2f64 v[NQ];
};
// Split chains (DevContig::split == 2, pg_split.h) keep a column's record as TWO streams — the sample's 128 bytes (the fifteen
// entries, X, flags: written every run) and the index's 64 (header, constants, row offsets: written once, shared by every sample
// over the index contig): piece g of a block is then piece g % 8 of sample record g / 8 for g < 8 BL, and piece (g - 8 BL) % 4 of
// index record (g - 8 BL) / 4 behind — parked where the same bytes of the one-stream record would lie, so that nothing
// downstream of the staging ring knows.
template <int BL>
DEVI void x_piece(bool two, uint32_t g, uint32_t& q, uint32_t& mem_off, uint32_t& lds_off, bool& from_index) {
    if (!two) { q = g / 12u; const uint32_t w = g - q * 12u; mem_off = 16u * w; lds_off = q * PG_XREC_BYTES + 16u * w; from_index = false; return; }
    if (g < (uint32_t)BL * 8u) { q = g >> 3; const uint32_t w = g & 7u; mem_off = 16u * w; lds_off = q * PG_XREC_BYTES + 16u * w; from_index = false; }
    else { const uint32_t gg = g - (uint32_t)BL * 8u; q = gg >> 2; const uint32_t w = gg & 3u; mem_off = 16u * w; lds_off = q * PG_XREC_BYTES + PG_XREC_HDR + 16u * w; from_index = true; }
}
template <int BL>
DEVI void x_block_load(XBlock<BL>& b, gcdouble* xrec, gcdouble* irec, int64_t c0, int64_t C, uint32_t j) {
    const bool two = irec != nullptr;
#pragma unroll
    for (int q = 0; q < XBlock<BL>::NQ; ++q) {
        const uint32_t g = j + 16u * (uint32_t)q;
        if (XBlock<BL>::NP % 16 == 0 || g < (uint32_t)XBlock<BL>::NP) {
            uint32_t cq, mo, lo;
            bool fi;
            x_piece<BL>(two, g, cq, mo, lo, fi);
            int64_t c = c0 + (int64_t)cq;
            c = c < 0 ? 0 : (c >= C ? C - 1 : c);
            const GAS char* base = fi ? (const GAS char*)irec + (size_t)c * PG_IXREC_BYTES
                                      : (const GAS char*)xrec + (size_t)c * (two ? PG_SREC2_BYTES : PG_XREC_BYTES);
            b.v[q] = *(const GAS v2f64*)(base + mo);
        }
    }
}
template <int BL>
DEVI void x_block_park(const XBlock<BL>& b, uint32_t stg, uint32_t j, bool two) {   // stg: LDS address of the block's staging area
#pragma unroll
    for (int q = 0; q < XBlock<BL>::NQ; ++q) {
        const uint32_t g = j + 16u * (uint32_t)q;
        if (XBlock<BL>::NP % 16 == 0 || g < (uint32_t)XBlock<BL>::NP) {
            uint32_t cq, mo, lo;
            bool fi;
            x_piece<BL>(two, g, cq, mo, lo, fi);
            *(LAS v2f64*)(uintptr_t)(stg + lo) = b.v[q];
        }
    }
}
// the fifteen table entries of the staged record at `src` -> the 6 x 6 table of `slot`: entry j at [a][b] and [b][a] (lane 15:
// none).  A wide column's entries 0 and 1 are its sixteen raw alleles: they land at slot + 48 .. 63, where x_wide_emissions reads them.
DEVI void x_expand(uint32_t src, uint32_t slot, uint32_t j, uint32_t tab_ab, uint32_t tab_ba) {
    if (j < 15u) {
        const double e = *(LAS const double*)(uintptr_t)(src + 8u * j);
        *(LAS double*)(uintptr_t)(slot + tab_ab) = e;
        *(LAS double*)(uintptr_t)(slot + tab_ba) = e;   // (a == b: the same place again)
    }
}
DEVI XConsts read_xconsts(uint32_t src) {   // src: LDS address of the staged record
    const v2f64 a = *(LAS const v2f64*)(uintptr_t)(src + PG_XREC_CONSTS), b = *(LAS const v2f64*)(uintptr_t)(src + PG_XREC_CONSTS + 16u);
    return XConsts{a.x, a.y, b.x, b.y};
}
DEVI XCol read_xcol(uint32_t src, uint32_t slot, uint32_t j) {   // src: the staged record, slot: the table slot its entries went to
    XCol c;
    const v4u32 ro = *(LAS const v4u32*)(uintptr_t)(src + PG_XREC_ROWOFF);
    const v4u32 hd = *(LAS const v4u32*)(uintptr_t)(src + PG_XREC_HDR);
    const uint32_t roj = *(LAS const unsigned char*)(uintptr_t)(src + PG_XREC_ROWOFF + j);
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
    gcdouble* xrec; gcdouble* irec; gdouble* wr; gcdouble* resume; gdouble* sc_a; gdouble* sc_b; gu8* fallback;
    gcdouble* partner; gdouble* part; GAS char* aux; const unsigned char* wide;
    int64_t C, lo, hi;
};

DEVI void smallx_init_shared(SmallXShared& sh, uint32_t lane) {
    if (lane < (uint32_t)PG_ETAB) {
        const uint32_t r = lane / (uint32_t)PG_ESTRIDE, a = lane % (uint32_t)PG_ESTRIDE;
        sh.onehot[r][a] = (r > 0u && a == r - 1u && a < (uint32_t)PG_AMAX) ? 1.0 : 0.0;
    }
    // every table slot and the staging ring start as zeros: the row of zeros in front of each table and column 5 of its rows
    // stay (the entries expanded cover [a][b], a, b < 5), and the rows of the wave that carry no half-chain read zero offsets,
    // not whatever the LDS held
    for (uint32_t q = lane; q < (uint32_t)(sizeof(sh.rec) + sizeof(sh.stg)) / 16u; q += 64u) *(v2f64*)(&sh.rec[0][0][0] + q * 16u) = v2f64{0.0, 0.0};
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

// The record pipeline of one role.  Records are numbered in the order the role meets them: rel 0 is the column whose emission
// enters the first step (forward: first - 1, backward: t0 + 1), rel r the column r steps further (DIR = +1: first - 1 + r,
// DIR = -1: t0 + 1 - r).  Block b = rel BL b .. BL b + BL - 1, a contiguous run of the half-chain's record array whichever the
// direction; the staging ring keeps a block as it lies in memory, so rel r sits at position r % BL (forward) or BL - 1 - r % BL
// (backward) of block r / BL.  Step n (uniform over the wave: every row counts from its own first column) needs rel n + 2 at
// its end; the loop is unrolled by BL, so the phase of a step inside its block is static:
//   (n + 2) % BL == BL - 1 : the block after this one — fetched at phase 0, BL - 1 steps ago — is parked;
//   (n + 2) % BL == 0      : a new block is in use (parked a step ago); the loads of the next one are issued.
template <int BL, int DIR>
struct XPipe {
    uint32_t stg0;      // LDS address of the row's staging ring
    uint32_t slot0;     // ... of its two table slots
    uint32_t tab_ab, tab_ba;
    uint32_t j;
    gcdouble* xrec;
    gcdouble* irec;     // split chains: the index's stream of the records (see x_piece); null: one stream
    int64_t origin, C;  // column of rel 0
    bool live;
    XBlock<BL> blk;
    DEVI uint32_t staged(uint32_t rel) const {   // LDS address of the staged record rel (rel uniform)
        const uint32_t b = rel / (uint32_t)BL, i = rel % (uint32_t)BL;
        return stg0 + (b & 1u) * (uint32_t)(PG_XBLOCK_MAX * PG_XREC_BYTES) + (DIR > 0 ? i : (uint32_t)BL - 1u - i) * PG_XREC_BYTES;
    }
    DEVI uint32_t slot_of(uint32_t rel) const { return slot0 + (rel & 1u) * PG_XSLOT_BYTES; }
    DEVI int64_t block_first_column(uint32_t b) const {   // lowest column of block b
        return DIR > 0 ? origin + (int64_t)b * BL : origin - (int64_t)b * BL - (BL - 1);
    }
    DEVI void fetch(uint32_t b) { if (live && !(kXExp & 4u)) x_block_load<BL>(blk, xrec, irec, block_first_column(b), C, j); }
    DEVI void park(uint32_t b) const { x_block_park<BL>(blk, stg0 + (b & 1u) * (uint32_t)(PG_XBLOCK_MAX * PG_XREC_BYTES), j, irec != nullptr); }
    DEVI void expand(uint32_t rel) const { x_expand(staged(rel), slot_of(rel), j, tab_ab, tab_ba); }
    DEVI XConsts consts(uint32_t rel) const { return read_xconsts(staged(rel)); }
    DEVI XCol column(uint32_t rel) const { return read_xcol(staged(rel), slot_of(rel), j); }
    // what step n does for the pipeline, at its end (I = n % BL, static)
    template <int I>
    DEVI void advance(uint32_t n) {
        constexpr int P = (I + 2) % BL;
        if constexpr (P == BL - 1) park((n + 2u) / (uint32_t)BL + 1u);
        if constexpr (P == 0) fetch((n + 2u) / (uint32_t)BL + 1u);
    }
    DEVI void init(SmallXShared& sh, uint32_t row, uint32_t lane_j, gcdouble* recs, gcdouble* irecs, int64_t origin_col, int64_t n_cols, bool on) {
        j = lane_j; xrec = recs; irec = irecs; origin = origin_col; C = n_cols; live = on;
        stg0 = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.stg[row][0][0];
        slot0 = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.rec[row][0][0];
        uint32_t pa, pb;
        x_pair_of(j < 15u ? j : 14u, pa, pb);
        tab_ab = PG_XSLOT_TABLE + pa * (uint32_t)(PG_ESTRIDE * 8) + pb * 8u;
        tab_ba = PG_XSLOT_TABLE + pb * (uint32_t)(PG_ESTRIDE * 8) + pa * 8u;
#pragma unroll
        for (int q = 0; q < XBlock<BL>::NQ; ++q) blk.v[q] = v2f64{0.0, 0.0};
        // block 0 now, block 1 on its way (parked by step BL - 3)
        if (live) x_block_load<BL>(blk, xrec, irec, block_first_column(0), C, j);
        park(0);
        if (live) x_block_load<BL>(blk, xrec, irec, block_first_column(1), C, j);
    }
};

template <int PHASE>
DEVI void small16x_forward(const DevContig* contigs, const uint32_t* ids, uint32_t n_ids, uint32_t chunk, double* dump, SmallXShared& sh) {
    constexpr int HP = 16, R = 16;
    constexpr int BL = PHASE == 2 ? 6 : 8;   // columns per record block (phase 2: a multiple of its partner-column rotation of three)
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
            cx.xrec = (gcdouble*)dc.frec; cx.irec = dc.split == 2u ? (gcdouble*)dc.ix_rec : nullptr; cx.sc_a = (gdouble*)dc.fscale; cx.fallback = (gu8*)dc.fwd_fallback;
            gdouble* fwd = (gdouble*)dc.fwd;
            cx.wr = fwd; cx.partner = (gcdouble*)fwd; cx.part = (gdouble*)dc.part; cx.aux = (GAS char*)dc.aux; cx.wide = dc.wide;
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
    smallx_init_shared(sh, lane);
    const uint32_t onehot = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.onehot[0][0];
    XPipe<BL, +1> pipe;
    pipe.init(sh, row, j, cx.xrec, cx.irec, (int64_t)first - 1, cx.C, cx.live);   // rel r = column first - 1 + r
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
    pipe.expand(0);
    pipe.expand(1);
    {
        const XCol c0 = pipe.column(0);
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
    XConsts cur = pipe.consts(1);   // column `first`: constants of the gap first-1 -> first
    XCol col = pipe.column(1);      // ... its alleles, header, table
    double ee[R], pp[R];   // x = ee pp (formed at the start of the next step, see lean_forward)
#pragma unroll
    for (int k = 0; k < R; ++k) { ee[k] = 1.0; pp[k] = x[k]; }
    // One column step of the (up to) four half-chains: column t = first + n = rel n + 1; at its end the next column's record
    // (rel n + 2) is expanded and read.  Rows whose half-chain is done (or absent) compute on whatever their registers hold
    // and store nothing.
    auto step = [&](int n, auto ic, double (&vp)[PHASE == 2 ? R : 1]) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;   // n % BL
        const int64_t t = (int64_t)first + n;
        const bool act = cx.live && t < cx.hi;
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
            pipe.template advance<I>((uint32_t)n);
            pipe.expand((uint32_t)n + 2u);
            cur = pipe.consts((uint32_t)n + 2u);
            col = pipe.column((uint32_t)n + 2u);
        }
    };
    double vv[3][PHASE == 2 ? R : 1];
    if constexpr (PHASE == 2) { load_partner((int64_t)first, vv[0]); load_partner((int64_t)first + 1, vv[1]); load_partner((int64_t)first + 2, vv[2]); }
    int n = 0;
    for (; n + BL - 1 < n_steps; n += BL)
        static_for<0, BL>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; step(n + i, ic, vv[i % 3]); });
    static_for<0, BL - 1>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; if (n < n_steps) { step(n, ic, vv[i % 3]); ++n; } });
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
    constexpr int BL = PHASE == 2 ? 6 : 8;   // (see small16x_forward)
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
            cx.xrec = (gcdouble*)dc.frec; cx.irec = dc.split == 2u ? (gcdouble*)dc.ix_rec : nullptr; cx.sc_a = (gdouble*)dc.bscale; cx.sc_b = (gdouble*)dc.bsum;
            gdouble* cols = (gdouble*)dc.fwd;
            cx.wr = cols; cx.partner = (gcdouble*)cols; cx.part = (gdouble*)dc.part; cx.aux = (GAS char*)dc.aux; cx.wide = dc.wide;
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
    smallx_init_shared(sh, lane);
    const uint32_t onehot = (uint32_t)(uintptr_t)(LAS unsigned char*)&sh.onehot[0][0];
    XPipe<BL, -1> pipe;
    pipe.init(sh, row, j, cx.xrec, cx.irec, t0 + 1, cx.C, cx.live);   // rel r = column t0 + 1 - r
    auto emissions = [&](const XCol& col, bool on, double (&e)[R]) __attribute__((always_inline)) {
        static_for<0, R>([&](auto kc) __attribute__((always_inline)) { constexpr int k = decltype(kc)::value; e[k] = x_emission<k>(col); });
        const bool wide = on && (col.nlf & PG_XREC_FLAG_WIDE) != 0u;
        if (__any(wide)) {
            x_wide_emissions(col, cx.wide, wide, j, e);
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    };
    // records: rel 0 = column t0 + 1 (its emission enters the first step, its constants are those of the gap t0 -> t0 + 1),
    // rel 1 = column t0 (alleles / table of the first step's own column)
    pipe.expand(0);
    pipe.expand(1);
    {
        const XCol c1 = pipe.column(0);
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
    XConsts cur = pipe.consts(0);   // constants of the gap t0 -> t0 + 1 (record t0 + 1)
    XCol col = pipe.column(1);      // column t0: its alleles, header, table
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
    // step n: column t = t0 - n = rel n + 1.  `cur` = constants of record t + 1 (rel n), `col` = record t; at the end of the step
    // the constants of record t (rel n + 1: the gap t - 1 -> t) and record t - 1 (rel n + 2) are read.
    auto step = [&](int n, auto ic, double (&vp)[PHASE == 2 ? R : 1]) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;   // n % BL
        const int64_t t = t0 - n;
        const bool act = cx.live && t >= cx.lo;
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
                if constexpr ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
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
            pipe.template advance<I>((uint32_t)n);
            pipe.expand((uint32_t)n + 2u);
            cur = pipe.consts((uint32_t)n + 1u);   // record t: the gap t - 1 -> t
            col = pipe.column((uint32_t)n + 2u);   // record t - 1
        }
    };
    double vv[3][PHASE == 2 ? R : 1];
    if constexpr (PHASE == 2) { load_partner(t0, vv[0]); load_partner(t0 - 1, vv[1]); load_partner(t0 - 2, vv[2]); }
    int n = 0;
    for (; n + BL - 1 < n_steps; n += BL)
        static_for<0, BL>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; step(n + i, ic, vv[i % 3]); });
    static_for<0, BL - 1>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; if (n < n_steps) { step(n, ic, vv[i % 3]); ++n; } });
}

template <int PHASE>
__global__ __launch_bounds__(64) void k_sweep_small16x(const DevContig* __restrict__ contigs, const uint32_t* __restrict__ ids, uint32_t n_ids, uint32_t chunk,
                                                       double* dump) {
    __shared__ SmallXShared sh;
    if (blockIdx.y == 0) small16x_forward<PHASE>(contigs, ids, n_ids, chunk, dump, sh);
    else small16x_backward<PHASE>(contigs, ids, n_ids, chunk, dump, sh);
}
