// pg_device.h — device-side data layout shared by the HIP kernels and the C-ABI shim.
//
// HBM layout per contig (one chain = one (contig, path-subset)):
//   inputs        : the SoA of include/pangenie_hmm.h, uploaded verbatim
//   vrec[V][RB]   : per-variant record written by k_prep (emission table, local allele
//                   index of every selected path, emission exponent)
//   colrec[C][RB] : the same records gathered in column order by k_records, with the
//                   Li-Stephens constants of the gap (c-1 -> c) filled in.  This is the
//                   ONE stream the chain kernels read besides the forward columns.
//   fwd[C][HP*HP] : column slots (fp64): slot c holds the forward column of c < C/2 and the
//                   backward column of c >= C/2 — written once in sweep phase 1, read once in
//                   sweep phase 2: the 16*H^2 algorithmic bytes per column.
//   part[C][part_slots/2][T][2] : (fused mode) per-thread posterior partials by row-allele pair,
//                   reduced by k_bins
//   scratch / handoff-free chunk buffers : (chunked mode) see DevContig::scratch
//   vpair[V][..]  : (mantissa, exponent) of the emission product of every local allele pair of a
//                   variant — applied ONCE to a finished posterior bin (k_bins / k_post), so a bin keeps
//                   full relative precision however small its emission is
//   wide[..]      : emission tables of columns with more than PG_AMAX alleles on the selected paths
//   xbuf[2][HP*HP]: generic sweep kernel only (HP >= 256): the column after the emission multiply
//   lik / lik_exp : outputs, one (mantissa, exponent) pair per genotype bin
//   vit_*         : (run_phasing) Viterbi backtrace and haplotypes, see DevContig
#pragma once
#include <stdint.h>

#define PG_AMAX 5                 // max distinct alleles on the selected paths of a NARROW column (table inside the record)
#define PG_ESTRIDE (PG_AMAX + 1)  // row stride of the expanded emission table; row/col PG_AMAX = 0 (phantom paths)
#define PG_ETAB (PG_ESTRIDE * PG_ESTRIDE)
#define PG_MAX_ALLELES_PER_VARIANT 256  // all alleles of one UniqueKmers object (k_prep keeps their presence in a 256-bit LDS bitmap)
#define PG_MAX_PATHS 1024               // selected paths per chain (HP = 16..128: register-resident kernels; 256..1024: generic kernel)
#define PG_PHANTOM 255

// byte offsets inside a record
#define PG_REC_C0 0
#define PG_REC_C1 8
#define PG_REC_C2 16
#define PG_REC_KAPPA 24
#define PG_REC_VARIANT 32
#define PG_REC_EXP 36
#define PG_REC_NLOCAL 40
#define PG_REC_FLAGS 41
#define PG_REC_LOCAL_SLOT 48  // u16[8] (PG_AMAX used)
#define PG_REC_BITS1 64       // u64[2]: bit p = selected path p carries local allele 1 (biallelic fast path)
#define PG_REC_E 80           // double[PG_ETAB]
#define PG_REC_ALLELES (80 + 8 * PG_ETAB)  // u8[HP]
#define PG_REC_FLAG_ALLZERO 1
#define PG_REC_FLAG_WIDE 2    // more than PG_AMAX alleles on the selected paths: table in DevContig::wide
#define PG_REC_WIDE_IDX 44    // u32: byte offset / 16 of the variant's wide entry inside DevContig::wide
#define PG_REC_AUX 60         // u32: byte offset / 16 of the variant's slot inside DevContig::aux (PG_WIDE_NONE: none); the last two of the eight local-slot entries
// INVARIANT of vrec: the records of a job are zeroed ONCE, when the job is built (pg_shim.cpp), and k_prep_bi stores only
// the 16-byte pieces of a biallelic object's record that are not zero by construction (header, row bits, four table entries,
// path alleles).  That is sound because the kernel that prepares an object is a function of the index shape alone (alleles and
// k-mers per variant, which pg_job_upload holds fixed) and no kernel writes anything else into a record: a new writer into
// vrec has to keep both, or store whole records.  The split path (below) has no vrec at all.

// Wide entries (columns with PG_AMAX < n_local <= PG_WIDE_MAX distinct alleles on the selected paths;
// chunked sweep mode — or, round 5, a fused job's 16-path chain on k_sweep_small16x): the emission table no longer fits the column record, so it lives in a
// side buffer.  With S = n_local + 1 (row/column n_local is zero: phantom paths) an entry holds
//   double  E [S][S]   emission products scaled by 2^-X (symmetric), read by the recursion
//   double  Pm[S][S]   mantissa in [0.5,1) (or 0) of the unscaled product of the local pair
//   int32   Pe[S][S]   its exponent                      (Pm, Pe: applied to finished bins)
//   u16     slot[n_local]  local allele -> allele slot of the variant
// The host reserves room for S = min(A, H) + 1 per variant with more than PG_AMAX alleles.
#define PG_WIDE_MAX 254
#define PG_WIDE_NONE 0xFFFFFFFFu
static inline uint64_t pg_wide_entry_bytes(uint32_t n_local) {
    const uint64_t s = (uint64_t)n_local + 1;
    return (s * s * 20 + (uint64_t)n_local * 2 + 15) & ~(uint64_t)15;
}
#define PG_WIDE_OFF_PM(S) ((size_t)(S) * (S) * 8)
#define PG_WIDE_OFF_PE(S) ((size_t)(S) * (S) * 16)
#define PG_WIDE_OFF_SLOT(S) ((size_t)(S) * (S) * 20)

// Chunked mode: the store-only chunk sweeps of phase 2 write into PG_SCRATCH_BUFS rotating scratch buffers (two roles each),
// k_post reads a finished one from a second stream.  Three (round 5; two before): with two, sweep i + 2 had to wait for the
// posteriors of chunk i, and while all chains of a genome are still running k_post takes longer than a sweep (3.0 against
// 2.7 ms on genome24_h64) — 6 ms of idle gaps between the 50 chunk sweeps (profiles/r05_genome24_timeline.txt); with three the
// early deficit is worked off once the short chains have ended.
#ifndef PG_SCRATCH_BUFS
#define PG_SCRATCH_BUFS 3u
#endif
#define PG_SCR_BUF(c) ((c) % PG_SCRATCH_BUFS)
#define PG_SYNC_NEXT 16u
#define PG_SYNC_DONE 17u
#define PG_SYNC_READY 32u   // chunks with both roles stored, republished by k_post_loop's watcher wave (what its other waves poll)
#define PG_SYNC_WORDS 48u   // DevContig::sync: three 64-byte lines per chain

// Column biases: stored forward columns (before the emission multiply) sum to about
// 2^PG_BIAS_F, stored backward columns to about 2^PG_BIAS_B times their emission-weighted mass.
#define PG_BIAS_F 400
#define PG_BIAS_B 400

// (mantissa, exponent) pair tables of narrow variants: vpair[v] = double m[NP] then int32 e[NP],
// NP = pair_n (pair_n + 1) / 2 padded to even, index la * pair_n - la (la - 1) / 2 + (lb - la)
static inline uint32_t pg_pair_count(uint32_t pair_n) { return ((pair_n * (pair_n + 1) / 2) + 1u) & ~1u; }
static inline uint32_t pg_pair_bytes(uint32_t pair_n) { return pg_pair_count(pair_n) * 12u; }

static inline uint32_t pg_rec_bytes(uint32_t hp) { return (PG_REC_ALLELES + hp + 63u) & ~63u; }

// error bits raised by kernels in DevContig::err
#define PG_DEVERR_ALLELE_NOT_FOUND 1u
#define PG_DEVERR_TOO_MANY_ALLELES 2u
#define PG_DEVERR_TOO_MANY_LOCAL 4u
#define PG_DEVERR_SYNC_TIMEOUT 16u // a persistent phase-2 kernel (k_sweep_lean<4> / k_post_loop) gave up waiting for its partner's chunk flag
#define PG_DEVERR_WIDE_FUSED 8u   // a wide column reached a fused bins kernel that has no path for it (a single-column chain of k_sweep_small16x's job)

struct DevTable {
    uint32_t cov_min, cov_max, count_max;  // dense box [cov_min,cov_max) x [0,count_max)
    uint32_t pad;
    double reg;           // regularization constant (on-the-fly path)
    const double* mant;   // [cov-cov_min][count][3]
    const int32_t* expo;  // [cov-cov_min][count][3]
    // the same entries as ONE 32-byte piece each, {m0, m1, m2, int16 e0, e1, e2, 0}: two 16-byte loads per k-mer where the
    // split arrays take six (the sample-level kernels of the split path, pg_split.h; every exponent of the table lies in
    // [PG_LD_MIN_EXP, 1], so int16 holds it)
    const unsigned char* packed;   // [cov-cov_min][count][32]
};

// ------------------------------------------------------------------------------------------------------------------
//  The split path (round 6): what depends on the INDEX alone is formed once, when the index is uploaded, and shared by
//  every chain (sample) over that index contig; a run of the job forms only what depends on the sample's read counts.
//  reference: the index is sample-independent, update_readcount / set_coverage are the per-sample part
//  (src/commands.cpp:118-138); the column list hangs on path alleles and undefined flags (src/columnindexer.cpp:8-33), the
//  transition probabilities on positions (src/transitionprobabilitycomputer.cpp:8-19), the emissions on the counts
//  (src/emissionprobabilitycomputer.cpp:9-53).
//
//  For EVERY chain (all kernels): kept[V], allele_present[sumA], col_variant[C], n_cols, col_of[V] are per index contig
//  (k_index_scan + k_compact at upload; no k_compact in a run).
//  For the 16-path chains of fused jobs (DevContig::split; k_sweep_small16 / k_sweep_small16x both phases), per COLUMN:
//    ix_pd [C][32]  what the sample-level emission kernel needs of a column: {variant, first k-mer, K | flags << 8,
//                   k-mer bits of allele slot 0, of slot 1 (biallelic objects), 0, 0, 0}
//    ix_rec[C][64]  what the sweeps need besides the emissions.  split == 1 (every object biallelic): {c0, c1, c2, kappa,
//                   row bits, 0, 0, 0} — copied into the 64-byte record of k_sweep_small16 by the emission kernel;
//                   split == 2: bytes 128 .. 191 of k_sweep_small16x's record (header, constants, row offsets), read
//                   by the sweep as a second stream
//    ix_bin[C][32]  what the bins kernels need: {first bin of the variant, variant, aux slot, A (u16), nlocal, flags,
//                   allele slot of local allele 0 .. 4 (u16), paths carrying local allele 0 .. 4 (u8), 0}
//  and per chain, in COLUMN order, written by the emission kernels k_prep_s_* every run:
//    split == 1: frec[C][64]  {c0, c1, c2, kappa, E'00, E'01, E'11, row bits | flags << 16 | X << 32}
//    split == 2: frec[C][128] the fifteen table entries (tri_local order; a wide column: its sixteen raw local alleles in
//                the first sixteen bytes) then {int32 X, uint32 flags}
//    cprec[C][192] only for columns flagged PG_SREC_FLAG_PRECISE: the unscaled products as (mantissa[16], exponent[16]).
//  The bins kernels recover a pair's (mantissa, exponent) from the scaled entry and X — E' = m 2^(e - X) exactly unless
//  that is subnormal: a column with such an entry (a present pair 2^-1021 below the column's largest) is flagged and its
//  products go to cprec as well.  No vrec / colrec / vpair for such chains.
#define PG_IXPD_BYTES 32u
#define PG_IXREC_BYTES 64u
#define PG_IXBIN_BYTES 32u
#define PG_SREC1_BYTES 64u
#define PG_SREC2_BYTES 128u
#define PG_CPREC_BYTES 192u
#define PG_COL_NONE 0xFFFFFFFFu
#define PG_IXPD_U0 0x100u        // kf: K | these
#define PG_IXPD_U1 0x200u
#define PG_IXPD_HAS0 0x400u
#define PG_IXPD_HAS1 0x800u
#define PG_SREC_FLAG_ALLZERO 1u  // (= PG_REC_FLAG_ALLZERO)
#define PG_SREC_FLAG_WIDE 2u     // (= PG_REC_FLAG_WIDE)
#define PG_SREC_FLAG_PRECISE 4u
struct IxBin {
    uint32_t g0, v, aux;
    uint16_t A;
    uint8_t nl, flags;
    uint16_t ls[5];
    uint8_t cnt[5];
    uint8_t pad;
};

struct DevContig {
    uint32_t V, H, HP, RB;
    uint32_t T;            // entries of posterior partials per column and slot pair: threads per chain workgroup = HP*HP/R (half of them at HP = 32)
    uint32_t pad0;
    double dist_scale;     // 0.000004 * recombrate * effective_N
    int32_t uniform;
    uint32_t debug;        // bit 3: in-kernel cycle counters (-DPG_CHAIN_PROF builds only; no effect in the product build)
    uint32_t part_slots;   // allele slots per column in `part` = min(PG_AMAX, max alleles of a variant)
    uint32_t pair_n;       // local alleles the vpair tables are laid out for = min(PG_AMAX, max alleles of a variant)
    // inputs
    const uint64_t* pos;
    const uint16_t* cov;
    const uint32_t* kmer_off;
    const uint16_t* kmer_count;
    const uint32_t* allele_off;
    const uint16_t* allele_id;
    const uint8_t*  allele_flags;
    const uint16_t* allele_koff;
    const uint32_t* allele_kmask;
    const uint16_t* path_allele;
    const uint64_t* geno_off;
    // intermediates
    uint8_t*  vrec;
    uint8_t*  kept;
    uint8_t*  allele_present;
    uint32_t* n_cols;
    uint32_t* col_variant;
    uint8_t*  colrec;
    double*   fwd;
    double*   part;
    double*   fscale;        // [V] mantissa m of the scale applied to forward column c (see chain kernels)
    double*   bscale;        // [V] same for the backward column
    double*   bsum;          // [V] sum of the stored backward column (hand-over between the phases)
    uint8_t*  fwd_fallback;  // [V] column c fell back to the uniform forward column (fsum := 1, no emission scale)
    // chunked mode (few chains, see pg_shim.cpp): phase 2 is run in chunks of chunk_cols columns per
    // half-chain that only STORE their columns into this scratch ([PG_SCRATCH_BUFS buffers][2 roles][chunk_cols][HP*HP],
    // forward role first); k_post forms the posteriors of a finished chunk on the idle CUs.
    double*   scratch;
    // persistent chunked phase 2 (all chains lean, few of them — pg_shim.cpp `persist`): ONE launch of k_sweep_lean<4> walks
    // every chunk of a half-chain and ONE launch of k_post_loop every chunk's posteriors; the two hand chunks over through
    // these words (zeroed at the start of every run; agent-scope release / acquire):
    //   sync[0] / sync[1]      = chunks the forward / backward role has finished storing (written by the sweep; a 64-byte line of their own),
    //   sync[PG_SYNC_NEXT]     = the next column slot to hand out (k_post_loop's work queue: chunk after chunk, 2 chunk_cols slots each),
    //   sync[PG_SYNC_DONE + b] = column slots of scratch buffer b whose posteriors are done, over all its chunks so far (k_post_loop)
    uint32_t* sync;            // [PG_SYNC_WORDS]
    uint32_t  chunk_cols;
    uint32_t  col_stride;      // doubles between consecutive columns of `fwd`: HP*HP, or 2304 (the 18 KB of a compact triangle) when tri
    uint8_t*  wide;            // wide entries (see above)
    const uint32_t* wide_idx;  // [V]: byte offset / 16 of the entry of variant v, PG_WIDE_NONE if it has <= PG_AMAX alleles
    uint8_t*  vpair;           // [V][pg_pair_bytes(pair_n)]
    // lean sweep (k_sweep_lean: HP = H = 64, every object biallelic): compact column records
    // {c0, c1, c2, kappa, E'00, E'01, E'11, bits1} (64 B, read with scalar loads) and the flag
    double*   frec;            // [V][8]
    // 1: every object of the chain is biallelic with at most 32 k-mers and H <= 64: k_prep_bi prepares four variants
    // per wave (a DPP row of 16 lanes each) instead of k_prep's one (PG_KERNELS=prepwave keeps k_prep: cross-check); 2: at least
    // half of the objects are such: k_prep_bi takes those, k_prep the others (each kernel skips the other's objects)
    uint32_t  prep_fast;
    // prep_fast == 2: the objects with 3 .. PG_AMAX alleles and <= 64 k-mers (k_prep_m4: four per wave) and every other object
    // that is not k_prep_bi's (k_prep: one per wave), as lists of variant ids per index contig — null: k_prep walks all variants
    uint32_t  n_prep_m4, n_prep_w;
    uint32_t  n_prep_b;        // ... and k_prep_bi's own objects (two alleles, <= 32 k-mers)
    const uint32_t* prep_m4;
    const uint32_t* prep_w;
    const uint32_t* prep_b;
    uint32_t  lean;            // 1: the store-only phases of this chain run on k_sweep_lean
    // 1 or 2 (lean chains of FUSED jobs; 2: phase 2 on k_sweep_lean2, 1: on the general kernel's triangle ring): phase 1 stores only the upper triangle of its (symmetric) columns, COMPACT at the
    // start of the column's slot (1152 16-byte units: row pair q, lanes 8 (q >> 2) .. 63, see tri_unit_of); elements
    // below the diagonal inside those units are written as 0, the diagonal is stored HALVED — so that phase 2 can
    // take the posterior sums over the stored half alone and k_bins doubles them: half the HBM bytes written by
    // phase 1 and read by phase 2
    uint32_t  tri;
    // 1 (tri == 1 and the chain is not a lean chain: 64 paths, multiallelic objects, fused job): phase 2 on k_sweep_leanx2 — the
    // posterior partials leave added up over the four waves, T = 64 entries per column and slot pair (PG_KERNELS=noleanx2: the
    // general kernel's triangle ring, T = 256)
    uint32_t  leanx2;
    // 1 (round 6, 64-path chains of FUSED jobs on the general kernel whose objects include wide ones — more than PG_AMAX alleles on the
    // selected paths): a wide column costs that column, not the job — its phase-2 role stores its own column P' / beta' in the
    // variant's aux slot (8 HP^2 bytes, PG_REC_AUX) instead of forming posterior partials, and k_bins_wide forms the bins from
    // that column and the stored partner (post_ab: what k_post does for every column of a chunked job).  PG_KERNELS=nowidef:
    // such a job runs chunked, as before.
    uint32_t  widef;
    // 1: every object of the chain is biallelic and H = HP = 16: the store-only phases run on k_sweep_small16 (four
    // half-chains per wave); the chain keeps its compact records (frec) next to the full ones
    // 2 (fused jobs, with cls4): phase 2 runs there too — partner columns prefetched into registers three steps ahead, the four
    // class sums of a column formed inside the 16-lane row
    uint32_t  small;
    // 1: HP = 128 or 64, every object has at most PG_AMAX alleles (all columns narrow) and the chain is not `lean`: the
    // store-only phases run on k_sweep_leanx (the lean step with table emissions)
    uint32_t  leanx;
    // 1 (fused jobs): HP = 16 or 32, H = HP, every object biallelic: phase 2 (general kernel, one compute wave per
    // half-chain) writes the four class sums of a column, part[4 c + 2 (row allele) + (column allele)], and
    // k_bins_lean2 turns them into bins — instead of per-thread partials reduced by k_bins
    uint32_t  cls4;
    // 1 / 2: HP = H = 16 and NOT every object biallelic: the store-only phases (1) / both phases (2: fused jobs) run on
    // k_sweep_small16x (pg_small16x.h): 192-byte column records in `frec`, class sums of columns with at most two local
    // alleles in `part` ([C][4]), the fifteen bins of columns with three to five and the phase-2 column of WIDE columns in
    // the variant's `aux` slot (k_bins_x, k_bins_wide)
    uint32_t  smallx;
    // 1 / 2 (small == 2 / smallx == 2 chains of fused jobs without run_phasing): the split path, see above
    uint32_t  split;
    const uint32_t* col_of;    // [V] per index contig: column of a kept variant, PG_COL_NONE otherwise
    const unsigned char* ix_pd;
    const unsigned char* ix_rec;
    const unsigned char* ix_bin;
    unsigned char* cprec;      // [V][192] per chain (touched for flagged columns only)
    uint32_t* ix_err;          // per index contig: error bits of k_index_scan (PG_DEVERR_*)
    const uint32_t* ix_big;    // per index contig: its variants with more than 32 alleles (the wave-per-object index kernels' list)
    uint32_t n_ix_big;
    uint32_t pad2;
    // the WIDE columns of the chain (smallx == 2, index with objects of more than PG_AMAX alleles): k_records appends every wide
    // column it meets, k_bins_wide walks the list — one wave per entry instead of a scan of all columns for the rare one
    uint32_t* wcols;           // [wide candidates of the index contig]
    uint32_t* n_wcols;         // (zeroed at the start of every run)
    unsigned char* aux;        // per chain: slots of the variants with more than two alleles (128 B) / more than PG_AMAX (8 HP^2 B)
    const uint32_t* aux_idx;   // [V] per index contig: byte offset / 16 of the variant's slot, PG_WIDE_NONE if it has two alleles
    // rows and lanes of a stored column that carry data: H rounded up to a multiple of 4 (fused jobs at HP = 32, where
    // 17 paths — a user-chosen panel size of 16 + the reference path; the DEFAULT, 15 + 1 = 16 paths, runs on k_sweep_small16[x] — would otherwise move 32 x 32 states per column for 17 x 17 real ones), else HP.
    // Phase 1 stores only rows and lanes below `live` (whole 64-byte sectors), the loader of phase 2 fetches only those,
    // and the rest of the LDS ring is zeroed once.  Nothing else reads the columns of a fused job.
    uint32_t  live;
    double*   xbuf;            // [2][HP*HP] generic kernel scratch (forward role first)
    uint32_t* err;
    unsigned long long* prof;  // [64] in-kernel cycle counters (-DPG_CHAIN_PROF / -DPG_LEAN_TIMELINE builds), profiling only
    // Viterbi phasing (run_phasing, pg_viterbi.hip): transition probabilities {p^2, pq, q^2} of every column,
    // the backtrace (index of the best previous state of every state, H*H per column), the best state of
    // the last column, and the haplotype alleles per variant
    double*   vit_tq;          // [V][8]: {t0, t1, t2} as exact (hi, lo) pairs of the long double values, formed on the host
    uint16_t* vit_back;        // [V][H][HP]: state index (p1 * H + p2) of the best previous state of state (p1, p2)
    uint32_t* vit_best;        // [1]
    uint16_t* hap1;            // [V] allele of the first / second haplotype at kept variants, 0 elsewhere
    uint16_t* hap2;
    // outputs
    double*   lik;             // [n_lik] mantissa in [0.5,1) or 0
    int32_t*  lik_exp;         // [n_lik] exponent: L = lik * 2^lik_exp
};
