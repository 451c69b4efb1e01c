/*
 * pg_oracle.c — TEST INFRASTRUCTURE ONLY (see pg_oracle.h).
 *
 * Plain-C restatement of the reference hot path in x87 long double.  Every
 * function cites the reference lines whose behaviour (including operation
 * order) it follows.  Written against the flat pg_contig_batch, so it consumes
 * exactly the bytes the HIP path consumes.
 */
#include "pg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/*  CopyNumber / ProbabilityTable                                      */
/* ------------------------------------------------------------------ */

/* reference src/copynumber.cpp:22-28 (ctor) + :30-41 (get_probability_of) */
void pgo_copynumber_regularized(long double cn0, long double cn1, long double cn2,
                                long double reg, long double out3[3]) {
    long double sum = cn0 + cn1 + cn2 + 3.0L * reg;
    out3[0] = (cn0 + reg) / sum;
    out3[1] = (cn1 + reg) / sum;
    out3[2] = 1.0L - out3[0] - out3[1];
}

struct pgo_table {
    uint16_t cov_min, cov_max, count_max;
    long double reg;
    long double* p; /* [count_max][cov_max-cov_min][3] */
};

/* reference src/probabilitytable.cpp:7-19 */
static double get_error_param(double kmer_coverage) {
    double cn0;
    if (kmer_coverage < 10.0) cn0 = 0.99;
    else if (kmer_coverage < 20) cn0 = 0.95;
    else if (kmer_coverage < 40) cn0 = 0.9;
    else cn0 = 0.8;
    return cn0;
}

/* reference src/probabilitytable.cpp:75-81.  `log(i)` there is the double
 * overload (integer argument), `log(mean)`/`exp` the long double ones. */
static long double poisson(long double mean, unsigned int value) {
    long double sum = 0.0L;
    int v = (int)value;
    for (size_t i = 1; i <= value; ++i) sum += log((double)i);
    long double log_val = -mean + v * logl(mean) - sum;
    return expl(log_val);
}

/* reference src/probabilitytable.cpp:83-85 */
static long double geometric(long double p, unsigned int value) {
    return powl(1.0L - p, (long double)value) * p;
}

/* reference src/probabilitytable.cpp:55-65 */
static void compute_probability(const pgo_table* t, uint16_t cov, uint16_t count,
                                long double out3[3]) {
    long double p_cn0 = geometric(get_error_param(cov), count);
    long double p_cn1 = poisson(cov / 2.0, count);
    long double p_cn2 = poisson(cov, count);
    if (t->reg > 0) {
        pgo_copynumber_regularized(p_cn0, p_cn1, p_cn2, t->reg, out3);
    } else {
        out3[0] = p_cn0; out3[1] = p_cn1; out3[2] = p_cn2;
    }
}

/* reference src/probabilitytable.cpp:28-45 */
pgo_table* pgo_table_create(uint16_t cov_min, uint16_t cov_max, uint16_t count_max,
                            long double regularization) {
    pgo_table* t = (pgo_table*)calloc(1, sizeof(pgo_table));
    if (!t) return NULL;
    t->cov_min = cov_min; t->cov_max = cov_max; t->count_max = count_max;
    t->reg = regularization;
    size_t ncov = cov_max > cov_min ? (size_t)(cov_max - cov_min) : 0;
    size_t n = (size_t)count_max * ncov * 3;
    t->p = (long double*)malloc((n ? n : 1) * sizeof(long double));
    if (!t->p) { free(t); return NULL; }
    for (uint32_t i = 0; i < count_max; ++i)
        for (uint32_t j = 0; j < ncov; ++j)
            compute_probability(t, (uint16_t)(j + cov_min), (uint16_t)i,
                                t->p + ((size_t)i * ncov + j) * 3);
    return t;
}

/* reference src/probabilitytable.cpp:21-26 */
pgo_table* pgo_table_create_default(void) { return pgo_table_create(0, 0, 0, 0.0L); }

void pgo_table_destroy(pgo_table* t) {
    if (!t) return;
    free(t->p);
    free(t);
}

static int in_table(const pgo_table* t, uint16_t cov, uint16_t count) {
    return (cov >= t->cov_min) && (cov < t->cov_max) && (count < t->count_max);
}

/* reference src/probabilitytable.cpp:67-73 */
int pgo_table_modify(pgo_table* t, uint16_t cov, uint16_t count,
                     long double p0, long double p1, long double p2) {
    if (!in_table(t, cov, count)) return PG_ERR_INVALID;
    size_t ncov = (size_t)(t->cov_max - t->cov_min);
    long double* e = t->p + ((size_t)count * ncov + (cov - t->cov_min)) * 3;
    e[0] = p0; e[1] = p1; e[2] = p2;
    return PG_OK;
}

/* reference src/probabilitytable.cpp:47-53 */
void pgo_table_get(const pgo_table* t, uint16_t cov, uint16_t count, long double out3[3]) {
    if (in_table(t, cov, count)) {
        size_t ncov = (size_t)(t->cov_max - t->cov_min);
        const long double* e = t->p + ((size_t)count * ncov + (cov - t->cov_min)) * 3;
        out3[0] = e[0]; out3[1] = e[1]; out3[2] = e[2];
    } else {
        compute_probability(t, cov, count, out3);
    }
}

/* ------------------------------------------------------------------ */
/*  TransitionProbabilityComputer                                      */
/* ------------------------------------------------------------------ */

/* reference src/transitionprobabilitycomputer.cpp:8-19 and :33-39 */
void pgo_transition_probs(uint64_t from_pos, uint64_t to_pos, double recombrate,
                          uint32_t nr_paths, int uniform, long double effective_N,
                          long double out3[3]) {
    if (uniform) { out3[0] = out3[1] = out3[2] = 1.0L; return; }
    long double distance = (to_pos - from_pos) * 0.000004L * ((long double)recombrate) * effective_N;
    long double recomb_prob = (1.0L - expl(-distance / (long double)nr_paths)) * (1.0L / (long double)nr_paths);
    long double no_recomb_prob = expl(-distance / (long double)nr_paths) + recomb_prob;
    out3[0] = no_recomb_prob * no_recomb_prob;
    out3[1] = no_recomb_prob * recomb_prob;
    out3[2] = recomb_prob * recomb_prob;
}

/* ------------------------------------------------------------------ */
/*  UniqueKmers accessors on the flat batch                            */
/* ------------------------------------------------------------------ */

/* reference src/kmerpath.cpp:33-48 (KmerPath::get_position) */
static inline unsigned kmer_on_slot(const pg_contig_batch* b, uint32_t slot, uint32_t k) {
    uint32_t off = b->allele_kmer_off[slot];
    if (k < off || k >= off + 32u) return 0;
    return (b->allele_kmer_mask[slot] >> (k - off)) & 1u;
}

static inline int slot_of_allele(const pg_contig_batch* b, uint32_t v, uint16_t allele) {
    for (uint32_t s = b->allele_off[v]; s < b->allele_off[v + 1]; ++s)
        if (b->allele_id[s] == allele) return (int)(s - b->allele_off[v]);
    return -1;
}

void pgo_geno_offsets(const pg_contig_batch* b, uint64_t* geno_off) {
    geno_off[0] = 0;
    for (uint32_t v = 0; v < b->n_variants; ++v) {
        uint64_t A = b->allele_off[v + 1] - b->allele_off[v];
        geno_off[v + 1] = geno_off[v] + A * (A + 1) / 2;
    }
}

static inline uint64_t tri_index(uint32_t A, uint32_t a, uint32_t c) { /* a <= c */
    return (uint64_t)a * A - (uint64_t)a * (a - 1) / 2 + (c - a);
}

/* ------------------------------------------------------------------ */
/*  EmissionProbabilityComputer                                        */
/* ------------------------------------------------------------------ */

/* reference src/emissionprobabilitycomputer.cpp:36-53 */
static long double compute_emission_probability(const pg_contig_batch* b, const pgo_table* t,
                                                uint32_t v, uint32_t s1, uint32_t s2,
                                                int a1_undefined, int a2_undefined) {
    long double result = 1.0L;
    uint32_t k0 = b->kmer_off[v], k1 = b->kmer_off[v + 1];
    uint16_t cov = b->coverage[v];
    uint32_t base = b->allele_off[v];
    for (uint32_t i = 0; i < k1 - k0; ++i) {
        unsigned expected = kmer_on_slot(b, base + s1, i) + kmer_on_slot(b, base + s2, i);
        long double p[3];
        pgo_table_get(t, cov, b->kmer_count[k0 + i], p);
        if (a1_undefined && a2_undefined) {
            result *= (1.0L / 3.0L) * (p[0] + p[1] + p[2]);
        } else if (a1_undefined || a2_undefined) {
            /* reference asserts expected < 2 here */
            result *= 0.5L * (p[expected] + p[expected + 1 > 2 ? 2 : expected + 1]);
        } else {
            result *= p[expected];
        }
    }
    return result;
}

/* reference src/emissionprobabilitycomputer.cpp:9-34: table over ALL alleles of the
 * object; all_zeros => every emission is 1.0 */
int pgo_emission_table(const pg_contig_batch* b, const pgo_table* t, uint32_t v,
                       long double* out, int32_t* all_zeros_out) {
    uint32_t base = b->allele_off[v];
    uint32_t A = b->allele_off[v + 1] - base;
    int all_zeros = 1;
    for (uint32_t s1 = 0; s1 < A; ++s1) {
        for (uint32_t s2 = 0; s2 < A; ++s2) {
            int u1 = b->allele_flags[base + s1] & 1;
            int u2 = b->allele_flags[base + s2] & 1;
            long double e = compute_emission_probability(b, t, v, s1, s2, u1, u2);
            out[(size_t)s1 * A + s2] = e;
            if (e > 0) all_zeros = 0;
        }
    }
    if (all_zeros)
        for (size_t i = 0; i < (size_t)A * A; ++i) out[i] = 1.0L;
    if (all_zeros_out) *all_zeros_out = all_zeros;
    return PG_OK;
}

/* ------------------------------------------------------------------ */
/*  HMM                                                                */
/* ------------------------------------------------------------------ */

typedef struct {
    long double* column; /* H*H, index p1*H + p2 (reference src/columnindexer.cpp:71-78) */
    long double forward_normalization_sum;
} hmm_column;

typedef struct {
    const pg_contig_batch* b;
    const pgo_table* t;
    const pg_hmm_params* p;
    uint32_t H, C;
    uint32_t* col_variant;   /* ColumnIndexer::variant_positions */
    uint16_t* slot_of_path;  /* [V*H] allele slot of path at variant (for kept columns) */
    hmm_column** forward;    /* [C] sparse table */
    hmm_column* prev_backward;
    long double* emis;       /* scratch A x A of the current emission computer */
    long double* emis2;
    uint32_t maxA;
    long double* lik;        /* output bins */
    uint64_t* geno_off;
} hmm_ctx;

static void free_column(hmm_column* c) {
    if (!c) return;
    free(c->column);
    free(c);
}

/* reference src/hmm.cpp:175-273 */
static int compute_forward_column(hmm_ctx* x, uint32_t ci) {
    if (x->forward[ci]) return PG_OK; /* :180-181 */
    const pg_contig_batch* b = x->b;
    uint32_t H = x->H;
    uint32_t v = x->col_variant[ci];
    long double tp[3] = {1.0L, 1.0L, 1.0L};
    hmm_column* prev = NULL;
    if (ci > 0) {
        prev = x->forward[ci - 1];
        uint32_t pv = x->col_variant[ci - 1];
        pgo_transition_probs(b->variant_pos[pv], b->variant_pos[v], x->p->recombrate, H,
                             x->p->uniform, x->p->effective_N, tp); /* :192-196 */
    }
    hmm_column* cur = (hmm_column*)malloc(sizeof(hmm_column));
    if (!cur) return PG_ERR_NOMEM;
    cur->column = (long double*)malloc(sizeof(long double) * (size_t)H * H);
    if (!cur->column) { free(cur); return PG_ERR_NOMEM; }

    uint32_t A = b->allele_off[v + 1] - b->allele_off[v];
    pgo_emission_table(b, x->t, v, x->emis, NULL); /* :203 */

    long double* helper_i = (long double*)calloc(H, sizeof(long double));
    long double* helper_j = (long double*)calloc(H, sizeof(long double));
    long double helper_ij = 0.0L;
    if (ci > 0) { /* :209-220 */
        size_t i = 0;
        for (uint32_t p1 = 0; p1 < H; ++p1)
            for (uint32_t p2 = 0; p2 < H; ++p2) {
                long double pf = prev->column[i];
                helper_i[p1] += pf;
                helper_j[p2] += pf;
                helper_ij += pf;
                i += 1;
            }
    }
    long double normalization_sum = 0.0L;
    size_t i = 0;
    const uint16_t* slots = x->slot_of_path + (size_t)v * H;
    for (uint32_t p1 = 0; p1 < H; ++p1) { /* :228-251 */
        for (uint32_t p2 = 0; p2 < H; ++p2) {
            long double previous_cell;
            if (ci > 0) {
                long double pc = prev->column[i];
                previous_cell = tp[0] * pc +
                                tp[1] * (helper_i[p1] + helper_j[p2] - 2 * pc) +
                                tp[2] * (helper_ij - helper_i[p1] - helper_j[p2] + pc);
            } else {
                previous_cell = 1.0L;
            }
            long double emission_prob = x->emis[(size_t)slots[p1] * A + slots[p2]];
            long double current_cell = previous_cell * emission_prob;
            cur->column[i] = current_cell;
            normalization_sum += current_cell;
            i += 1;
        }
    }
    size_t n = (size_t)H * H;
    if (normalization_sum > 0.0L) { /* :253-267 */
        for (size_t s = 0; s < n; ++s) cur->column[s] = cur->column[s] / normalization_sum;
        cur->forward_normalization_sum = normalization_sum;
    } else {
        long double uniform = 1.0L / (long double)n;
        for (size_t s = 0; s < n; ++s) cur->column[s] = uniform;
        cur->forward_normalization_sum = 1.0L;
    }
    x->forward[ci] = cur;
    free(helper_i);
    free(helper_j);
    return PG_OK;
}

/* reference src/hmm.cpp:275-405 */
static int compute_backward_column(hmm_ctx* x, uint32_t ci) {
    const pg_contig_batch* b = x->b;
    uint32_t H = x->H, C = x->C;
    uint32_t v = x->col_variant[ci];
    long double tp[3] = {1.0L, 1.0L, 1.0L};
    uint32_t An = 0;
    const uint16_t* nslots = NULL;
    if (ci < C - 1) {
        uint32_t nv = x->col_variant[ci + 1];
        pgo_transition_probs(b->variant_pos[v], b->variant_pos[nv], x->p->recombrate, H,
                             x->p->uniform, x->p->effective_N, tp); /* :289-294 */
        An = b->allele_off[nv + 1] - b->allele_off[nv];
        pgo_emission_table(b, x->t, nv, x->emis2, NULL); /* :295 */
        nslots = x->slot_of_path + (size_t)nv * H;
        if (!x->forward[ci]) { /* :298-305 */
            size_t k = (size_t)sqrt((double)C);
            size_t next = ((size_t)ci / k) * k;
            if (next > (size_t)C - 1) next = C - 1;
            for (size_t j = next + 1; j <= ci; ++j) {
                int rc = compute_forward_column(x, (uint32_t)j);
                if (rc) return rc;
            }
        }
    }
    hmm_column* fwd = x->forward[ci];
    if (!fwd) return PG_ERR_INVALID;

    long double* helper_i = (long double*)calloc(H, sizeof(long double));
    long double* helper_j = (long double*)calloc(H, sizeof(long double));
    long double helper_ij = 0.0L;
    if (ci < C - 1) { /* :314-327 */
        size_t i = 0;
        for (uint32_t p1 = 0; p1 < H; ++p1)
            for (uint32_t p2 = 0; p2 < H; ++p2) {
                long double w = x->prev_backward->column[i] *
                                x->emis2[(size_t)nslots[p1] * An + nslots[p2]];
                helper_i[p1] += w;
                helper_j[p2] += w;
                helper_ij += w;
                i += 1;
            }
    }
    hmm_column* cur = (hmm_column*)malloc(sizeof(hmm_column));
    cur->column = (long double*)malloc(sizeof(long double) * (size_t)H * H);
    long double normalization_sum = 0.0L;
    uint32_t A = b->allele_off[v + 1] - b->allele_off[v];
    const uint16_t* slots = x->slot_of_path + (size_t)v * H;
    long double* bins = x->lik + x->geno_off[v];
    size_t i = 0;
    for (uint32_t p1 = 0; p1 < H; ++p1) { /* :341-371 */
        for (uint32_t p2 = 0; p2 < H; ++p2) {
            long double current_cell;
            if (ci < C - 1) {
                long double helper_cell = x->prev_backward->column[i] *
                                          x->emis2[(size_t)nslots[p1] * An + nslots[p2]];
                current_cell = tp[0] * helper_cell +
                               tp[1] * (helper_i[p1] + helper_j[p2] - 2 * helper_cell) +
                               tp[2] * (helper_ij - helper_i[p1] - helper_j[p2] + helper_cell);
            } else {
                current_cell = 1.0L;
            }
            cur->column[i] = current_cell;
            normalization_sum += current_cell;
            long double forward_backward_prob = fwd->column[i] * current_cell;
            uint32_t s1 = slots[p1], s2 = slots[p2];
            uint32_t lo = s1 < s2 ? s1 : s2, hi = s1 < s2 ? s2 : s1;
            /* GenotypingResult::add_to_likelihood, src/genotypingresult.cpp:25-28 */
            bins[tri_index(A, lo, hi)] += forward_backward_prob * fwd->forward_normalization_sum;
            i += 1;
        }
    }
    size_t n = (size_t)H * H;
    if (normalization_sum > 0.0L) { /* :374-380 */
        for (size_t s = 0; s < n; ++s) cur->column[s] = cur->column[s] / normalization_sum;
    } else {
        long double uniform = 1.0L / (long double)n;
        for (size_t s = 0; s < n; ++s) cur->column[s] = uniform;
    }
    free_column(x->prev_backward);
    x->prev_backward = cur;
    free_column(x->forward[ci]); /* :397-400 */
    x->forward[ci] = NULL;
    free(helper_i);
    free(helper_j);
    return PG_OK;
}

int pgo_genotype_contig(const pg_contig_batch* b, const pgo_table* t,
                        const pg_hmm_params* p, pgo_result* out) {
    uint32_t V = b->n_variants, H = b->n_paths;
    if (V > 0 && H == 0) return PG_ERR_NO_PATHS; /* src/columnindexer.cpp:18-22 */
    if (p->run_phasing) return PG_ERR_UNSUPPORTED;
    hmm_ctx x;
    memset(&x, 0, sizeof(x));
    x.b = b; x.t = t; x.p = p; x.H = H;
    x.geno_off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)V + 1));
    pgo_geno_offsets(b, x.geno_off);
    memset(out->lik, 0, sizeof(long double) * x.geno_off[V]);
    x.lik = out->lik;
    uint32_t sumA = V ? b->allele_off[V] : 0;
    memset(out->allele_present, 0, sumA);
    memset(out->kept, 0, V);
    memset(out->n_kmers, 0, sizeof(uint16_t) * V);
    memset(out->coverage, 0, sizeof(uint16_t) * V);
    x.col_variant = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)V + 1));
    x.slot_of_path = (uint16_t*)malloc(sizeof(uint16_t) * ((size_t)V * H + 1));
    int rc = PG_OK;

    /* ColumnIndexer, reference src/columnindexer.cpp:8-33 */
    uint32_t C = 0, maxA = 1;
    for (uint32_t v = 0; v < V; ++v) {
        uint32_t A = b->allele_off[v + 1] - b->allele_off[v];
        if (A > maxA) maxA = A;
        int all_absent = 1;
        for (uint32_t pth = 0; pth < H; ++pth) {
            uint16_t a = b->path_allele[(size_t)v * H + pth];
            int s = slot_of_allele(b, v, a);
            if (s < 0) { rc = PG_ERR_INVALID; goto done; }
            x.slot_of_path[(size_t)v * H + pth] = (uint16_t)s;
            out->allele_present[b->allele_off[v] + s] = 1;
            if (a != 0 && !(b->allele_flags[b->allele_off[v] + s] & 1)) all_absent = 0;
        }
        if (!all_absent) { x.col_variant[C++] = v; out->kept[v] = 1; }
    }
    x.C = C; x.maxA = maxA;
    out->n_columns = C;
    x.emis = (long double*)malloc(sizeof(long double) * (size_t)maxA * maxA);
    x.emis2 = (long double*)malloc(sizeof(long double) * (size_t)maxA * maxA);

    if (p->run_genotyping) {
        /* compute_forward_prob, reference src/hmm.cpp:76-90 */
        x.forward = (hmm_column**)calloc(C ? C : 1, sizeof(hmm_column*));
        size_t k = (size_t)sqrt((double)C);
        for (uint32_t ci = 0; ci < C; ++ci) {
            rc = compute_forward_column(&x, ci);
            if (rc) goto done;
            if ((k > 1) && (ci > 0) && (((ci - 1) % k != 0))) {
                free_column(x.forward[ci - 1]);
                x.forward[ci - 1] = NULL;
            }
        }
        /* compute_backward_prob, reference src/hmm.cpp:92-110 */
        if (C > 0) {
            for (int64_t ci = (int64_t)C - 1; ci >= 0; --ci) {
                rc = compute_backward_column(&x, (uint32_t)ci);
                if (rc) goto done;
            }
            for (uint32_t v = 0; v < V; ++v) { /* :106-109 */
                out->n_kmers[v] = (uint16_t)(b->kmer_off[v + 1] - b->kmer_off[v]);
                out->coverage[v] = b->coverage[v];
            }
        }
    }
done:
    if (x.forward) {
        for (uint32_t ci = 0; ci < C; ++ci) free_column(x.forward[ci]);
        free(x.forward);
    }
    free_column(x.prev_backward);
    free(x.emis); free(x.emis2);
    free(x.col_variant); free(x.slot_of_path); free(x.geno_off);
    return rc;
}

/* ------------------------------------------------------------------ */
/*  Viterbi (phasing)                                                  */
/* ------------------------------------------------------------------ */

/* HMM::compute_viterbi_path / compute_viterbi_column, reference src/hmm.cpp:112-173, :408-511.
 * Max-product over ordered path pairs; a column is divided by its sum (uniform if the sum is 0, :484-491); ties are
 * resolved by `>=` while scanning the previous states in index order, so the LAST maximum wins (:468, :139).
 *   form 0: the reference's loop itself — every state scans all H^2 previous states (O(H^4) per column)
 *   form 1: the same maximum from the row / column / global maxima of the previous column (t0 >= t1 >= t2, so
 *           max_j prev[j] t(j -> i) = max(t0 prev[i], t1 rowmax, t1 colmax, t2 gmax)); among equal values the largest
 *           index, and the last index of all when every product is 0 — what the scan ends on.  O(H^2) per column.
 * The reference keeps sqrt(C) columns and recomputes the rest while backtracking (:118-128, :152-158); recomputed
 * columns are the same values, so all backtrace columns are simply kept here.
 * hap1/hap2[v]: alleles of the two haplotypes at kept variants (GenotypingResult::add_first/second_haplotype_allele),
 * 0 elsewhere.  n_kmers / coverage: set at index c (the COLUMN index, sic: reference :164-165) for c < C. */
int pgo_viterbi_contig(const pg_contig_batch* b, const pgo_table* t, const pg_hmm_params* p, int form,
                       uint16_t* hap1, uint16_t* hap2, uint8_t* kept, uint32_t* n_columns,
                       uint16_t* n_kmers, uint16_t* coverage) {
    const uint32_t V = b->n_variants, H = b->n_paths;
    if (V > 0 && H == 0) return PG_ERR_NO_PATHS;
    memset(hap1, 0, sizeof(uint16_t) * V);
    memset(hap2, 0, sizeof(uint16_t) * V);
    memset(kept, 0, V);
    memset(n_kmers, 0, sizeof(uint16_t) * V);
    memset(coverage, 0, sizeof(uint16_t) * V);
    uint32_t* col_variant = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)V + 1));
    uint16_t* slot_of_path = (uint16_t*)malloc(sizeof(uint16_t) * ((size_t)V * H + 1));
    uint32_t C = 0, maxA = 1;
    int rc = PG_OK;
    long double *emis = NULL, *prev = NULL, *cur = NULL, *rowmax = NULL, *colmax = NULL;
    uint32_t *back = NULL, *rowidx = NULL, *colidx = NULL;
    for (uint32_t v = 0; v < V; ++v) { /* ColumnIndexer, src/columnindexer.cpp:8-33 */
        uint32_t A = b->allele_off[v + 1] - b->allele_off[v];
        if (A > maxA) maxA = A;
        int all_absent = 1;
        for (uint32_t pth = 0; pth < H; ++pth) {
            uint16_t a = b->path_allele[(size_t)v * H + pth];
            int s = slot_of_allele(b, v, a);
            if (s < 0) { rc = PG_ERR_INVALID; goto done; }
            slot_of_path[(size_t)v * H + pth] = (uint16_t)s;
            if (a != 0 && !(b->allele_flags[b->allele_off[v] + s] & 1)) all_absent = 0;
        }
        if (!all_absent) { col_variant[C++] = v; kept[v] = 1; }
    }
    *n_columns = C;
    if (C == 0) goto done; /* :114 */
    {
        const size_t n = (size_t)H * H;
        emis = (long double*)malloc(sizeof(long double) * (size_t)maxA * maxA);
        prev = (long double*)malloc(sizeof(long double) * n);
        cur = (long double*)malloc(sizeof(long double) * n);
        rowmax = (long double*)malloc(sizeof(long double) * H);
        colmax = (long double*)malloc(sizeof(long double) * H);
        rowidx = (uint32_t*)malloc(sizeof(uint32_t) * H);
        colidx = (uint32_t*)malloc(sizeof(uint32_t) * H);
        back = (uint32_t*)malloc(sizeof(uint32_t) * n * C);
        if (!emis || !prev || !cur || !rowmax || !colmax || !rowidx || !colidx || !back) { rc = PG_ERR_NOMEM; goto done; }
        for (uint32_t c = 0; c < C; ++c) { /* compute_viterbi_column, :408-511 */
            const uint32_t v = col_variant[c];
            const uint32_t A = b->allele_off[v + 1] - b->allele_off[v];
            long double tp[3] = {1.0L, 1.0L, 1.0L};
            if (c > 0)
                pgo_transition_probs(b->variant_pos[col_variant[c - 1]], b->variant_pos[v], p->recombrate, H, p->uniform,
                                     p->effective_N, tp); /* :420-428 */
            pgo_emission_table(b, t, v, emis, NULL); /* :434 */
            long double gmax = 0.0L;
            uint32_t gidx = 0;
            if (c > 0 && form == 1) {
                for (uint32_t a = 0; a < H; ++a) { rowmax[a] = colmax[a] = 0.0L; rowidx[a] = colidx[a] = 0; }
                for (size_t j = 0; j < n; ++j) {
                    const uint32_t q1 = (uint32_t)(j / H), q2 = (uint32_t)(j % H);
                    if (prev[j] >= rowmax[q1]) { rowmax[q1] = prev[j]; rowidx[q1] = (uint32_t)j; }
                    if (prev[j] >= colmax[q2]) { colmax[q2] = prev[j]; colidx[q2] = (uint32_t)j; }
                    if (prev[j] >= gmax) { gmax = prev[j]; gidx = (uint32_t)j; }
                }
            }
            const uint16_t* slots = slot_of_path + (size_t)v * H;
            long double normalization_sum = 0.0L;
            size_t i = 0;
            for (uint32_t p1 = 0; p1 < H; ++p1) {
                for (uint32_t p2 = 0; p2 < H; ++p2) {
                    long double previous_cell = 1.0L;
                    if (c > 0) {
                        long double max_value = 0.0L;
                        size_t max_index = 0;
                        if (form == 0) { /* :446-473 */
                            size_t j = 0;
                            for (uint32_t q1 = 0; q1 < H; ++q1)
                                for (uint32_t q2 = 0; q2 < H; ++q2) {
                                    long double prev_prob = prev[j];
                                    prev_prob *= tp[(q1 != p1) + (q2 != p2)]; /* compute_transition_prob, transitionprobabilitycomputer.cpp:21-31 */
                                    if (prev_prob >= max_value) { max_value = prev_prob; max_index = j; }
                                    j += 1;
                                }
                        } else {
                            const long double cv[4] = {prev[i] * tp[0], rowmax[p1] * tp[1], colmax[p2] * tp[1], gmax * tp[2]};
                            const size_t ci[4] = {i, rowidx[p1], colidx[p2], gidx};
                            for (int q = 0; q < 4; ++q)
                                if (cv[q] > max_value || (cv[q] == max_value && ci[q] >= max_index)) { max_value = cv[q]; max_index = ci[q]; }
                            if (max_value == 0.0L) max_index = n - 1; /* every product is 0: the scan ends on the last state */
                        }
                        previous_cell = max_value;
                        back[(size_t)c * n + i] = (uint32_t)max_index;
                    }
                    long double emission_prob = emis[(size_t)slots[p1] * A + slots[p2]];
                    long double current_cell = previous_cell * emission_prob;
                    cur[i] = current_cell;
                    normalization_sum += current_cell;
                    i += 1;
                }
            }
            if (normalization_sum > 0.0L) { /* :484-491 */
                for (size_t s = 0; s < n; ++s) cur[s] = cur[s] / normalization_sum;
            } else {
                long double uniform = 1.0L / (long double)n;
                for (size_t s = 0; s < n; ++s) cur[s] = uniform;
            }
            long double* tmp = prev; prev = cur; cur = tmp;
        }
        /* best state of the last column (:131-141), backtracking (:144-172) */
        size_t best_index = 0;
        long double best_value = 0.0L;
        for (size_t s = 0; s < n; ++s)
            if (prev[s] >= best_value) { best_value = prev[s]; best_index = s; }
        for (uint32_t c = C; c-- > 0;) {
            const uint32_t v = col_variant[c];
            hap1[v] = b->path_allele[(size_t)v * H + best_index / H];
            hap2[v] = b->path_allele[(size_t)v * H + best_index % H];
            n_kmers[c] = (uint16_t)(b->kmer_off[c + 1] - b->kmer_off[c]); /* sic: by column index, :164-165 */
            coverage[c] = b->coverage[c];
            if (c > 0) best_index = back[(size_t)c * n + best_index];
        }
    }
done:
    free(emis); free(prev); free(cur); free(rowmax); free(colmax); free(rowidx); free(colidx); free(back);
    free(col_variant); free(slot_of_path);
    return rc;
}
