// pg_experiments.h — compile-time knobs of the TIMING EXPERIMENTS and in-kernel profilers (tools/exp_pipe.py,
// tools/exp_timeline.py, tools/bench_leanx.py).  The product library is built with none of them defined: every mask below
// is 0 / false and the code it guards folds away.  Builds with a *_EXP mask set produce WRONG results by design (a part of
// the step is compiled out so that its share of the time can be read): only their timing is of interest, and their results
// are recorded under profiles/ (r03_lean_ablation.txt, r04_lean_chain.txt).  Included by pg_kernels.hip only.
#pragma once

// general kernel (k_sweep): 1 no column stores, 2 total := 64 C_j, 4 no u round trip, 8 no posterior, 16 no emission
// multiply, 32 / 64 / 128: see the sites.  PG_CHAIN_PROF: cycle counters of a role's launch in DevContig::prof.
#ifndef PG_EXP
#define PG_EXP 0
#endif
static constexpr unsigned kExp = PG_EXP;
#ifdef PG_CHAIN_PROF
static constexpr bool kChainProf = true;
#else
static constexpr bool kChainProf = false;
#endif

// lean step (k_sweep_lean): 1 no column stores, 2 no emission fetches, 4 no MFMA total; PG_LEAN_DPPSUM: the wave total by
// six DPP steps (wave_sum) instead of the two fp64 MFMAs
#ifndef PG_LEAN_EXP
#define PG_LEAN_EXP 0
#endif
static constexpr unsigned kLeanExp = PG_LEAN_EXP;
#ifdef PG_LEAN_DPPSUM   // experiment: the wave total by six DPP steps (wave_sum) instead of the two fp64 MFMAs
static constexpr bool kLeanDppSum = true;
#else
static constexpr bool kLeanDppSum = false;
#endif

// -DPG_POST_GENERIC: k_post / k_post_loop form the columns of lean chains with the general post_ab instead of post_lean64 (results
// identical; the A/B build of round 6's k_post measurement)
#ifdef PG_POST_GENERIC
static constexpr bool kPostGeneric = true;
#else
static constexpr bool kPostGeneric = false;
#endif

// persistent phase 2 (k_sweep_lean<4> + k_post_loop), timing experiments — results WRONG: 1 k_post_loop hands the buffers back
// without forming posteriors, 2 the sweep does not wait for its scratch buffer
#ifndef PG_PERSIST_EXP
#define PG_PERSIST_EXP 0
#endif
static constexpr unsigned kPersistExp = PG_PERSIST_EXP;

// -DPG_LEAN_PRETOTAL=1 (round 6's structural attempt on the lean step, measured and NOT adopted — profiles/r06_lean_chain.txt): every
// wave adds up its own partial sums IN FRONT of the barrier (six DPP levels) and the column's total S is three additions of four
// broadcast LDS reads behind it, instead of two dependent fp64 MFMAs behind the exchange.  Correct (the lean suites pass), slower:
// phase 1 of genome24_h64 125.8 ms against 121.5 — the MFMAs' latency was already covered by the work pinned into their shadows,
// the eighteen DPP instructions in front of the barrier are not covered by anything.
#ifndef PG_LEAN_PRETOTAL
#define PG_LEAN_PRETOTAL 0
#endif
static constexpr bool kLeanPreTotal = PG_LEAN_PRETOTAL != 0;

// -DPG_NT_STORES: the lean sweeps' column stores as non-temporal stores; -DPG_NT_POST: k_post's column loads (post_lean64) as
// non-temporal loads (round 6 A/B builds)
#ifdef PG_NT_STORES
static constexpr bool kNtStores = true;
#else
static constexpr bool kNtStores = false;
#endif
#ifdef PG_NT_POST
static constexpr bool kNtPost = true;
#else
static constexpr bool kNtPost = false;
#endif

// lean-x step (k_sweep_leanx): 1 no column stores, 2 no emission fetches
#ifndef PG_LX_EXP
#define PG_LX_EXP 0
#endif
static constexpr unsigned kLxExp = PG_LX_EXP;   // timing experiments: 1 no column stores, 2 no emission fetches — results WRONG

// pipelined lean step (k_sweep_leanp — tools/lean_pipe/pg_lean_pipe.h, outside the product since round 5): 1 no column stores, 2 no class totals
#ifndef PG_LEANP_EXP
#define PG_LEANP_EXP 0
#endif
static constexpr unsigned kLeanpExp = PG_LEANP_EXP;

// -DPG_LEAN_TIMELINE builds only (tools/exp_pipe.py, profiles/r04_lean_chain.txt): s_memtime stamps at the segment
// boundaries of one column step of wave 0, each issued behind a use of the value that ends the segment; the stamps are
// only read behind the step's barrier (reading one earlier would drain the LDS queue with it).  Sums per segment over
// the launch go to DevContig::prof[32 + segment] (forward role) / [48 + segment] (backward role), [.. + 15] = steps.
#ifdef PG_LEAN_TIMELINE
static constexpr bool kLeanTimeline = true;
#else
static constexpr bool kLeanTimeline = false;
#endif
struct LeanTimeline {
    unsigned long long t[10], acc[10];
    DEVI void init() { if constexpr (kLeanTimeline) { for (int i = 0; i < 10; ++i) { t[i] = 0; acc[i] = 0; } } }
    template <int I>
    DEVI void mark(double dep) {
        if constexpr (kLeanTimeline) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" :: "v"(dep));
            t[I] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    template <int N>
    DEVI void fold() {   // behind the barrier: t[0] .. t[N] are this step's stamps
        if constexpr (kLeanTimeline) {
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] += t[i + 1] - t[i];
            acc[9] += 1;
        }
    }
    DEVI void write(unsigned long long* o) const {
        if constexpr (kLeanTimeline) { for (int i = 0; i < 9; ++i) o[i] = acc[i]; o[15] = acc[9]; }
    }
};

// k_sweep_small16x (pg_small16x.h): 1 no emission reads (e := 1), 2 no parking / no next-record reads (the first record's
// constants and alleles for every column), 4 no record loads, 8 no column stores (phases 1, 3), 16 no row-allele accumulation
// (phase 2), 32 no class sums / bins / aux stores (phase 2), 64 no partner-column loads (phase 2).  (The masks 128 / 256 / 512 named in
// profiles/r05_small16x_ablation*.txt belonged to the per-step record loads those measurements replaced by blocks; they are gone.)
#ifndef PG_X_EXP
#define PG_X_EXP 0
#endif
static constexpr unsigned kXExp = PG_X_EXP;

// -DPG_VIT_TIMELINE builds only (tools/exp_viterbi_timeline.py, profiles/r06_viterbi.txt): s_memtime stamps at the segment
// boundaries of one column step of k_viterbi's wave 0, each issued behind a use of the value that ends the segment; sums per
// segment over the launch go to DevContig::prof[0 .. 11], [15] = steps.  -DPG_VIT_EXP=mask: timing experiments (results WRONG):
// 1 rows: no lo pass, 2 rows: no index pass, 4 column as a whole: no lo pass, 8 no back-pointer stores, 16 no emission fetches.
#ifdef PG_VIT_TIMELINE
static constexpr bool kVitTimeline = true;
#else
static constexpr bool kVitTimeline = false;
#endif
#ifndef PG_VIT_EXP
#define PG_VIT_EXP 0
#endif
static constexpr unsigned kVitExp = PG_VIT_EXP;
struct VitTimeline {
    unsigned long long t[13], acc[13];
    DEVI void init() { if constexpr (kVitTimeline) { for (int i = 0; i < 13; ++i) { t[i] = 0; acc[i] = 0; } } }
    template <int I>
    DEVI void mark(double dep) {
        if constexpr (kVitTimeline) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" :: "v"(dep));
            t[I] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    template <int I>
    DEVI void mark_u(unsigned dep) {
        if constexpr (kVitTimeline) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" :: "v"(dep));
            t[I] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    template <int N>
    DEVI void fold() {
        if constexpr (kVitTimeline) {
#pragma unroll
            for (int i = 0; i < N; ++i) acc[i] += t[i + 1] - t[i];
            acc[12] += 1;
        }
    }
    DEVI void write(unsigned long long* o) const {
        if constexpr (kVitTimeline) { for (int i = 0; i < 12; ++i) o[i] = acc[i]; o[15] = acc[12]; }
    }
};
