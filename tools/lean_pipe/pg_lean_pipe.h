// pg_lean_pipe.h — the PIPELINED lean step (k_sweep_leanp): store-only phases (1, 3) of lone all-biallelic H = HP = 64
// chains (BASELINE configs[2], [3]).  Included by pg_kernels.hip behind the lean step's helpers.
//
// k_sweep_lean walks, every column, the dependent chain  partial sums -> LDS -> barrier -> LDS -> total -> scale -> first
// state  with its four waves stalled through it (~176 instructions issue in ~740 cycles, the column takes
// 1360-1470).  Here the exchange leaves that chain.  With x_t = e_t . P'_t,
//     P'_t(i,j) = sc (c0 x_{t-1}(i,j) + c1 C_i + c1 C_j + c2 S)            (C, S: column sums / total of x_{t-1}),
// the column sums of x_t follow from quantities of column t-1 alone:
//     C^t_j = sc (c0 Y_j + (c1 C_j + c2 S) N_t(j) + c1 (T_t[0][b_j] Q_0 + T_t[1][b_j] Q_1)),
//     Y_j   = sum_i e_t(i,j) x_{t-1}(i,j)     the emission-weighted column sum, accumulated DURING step t-1 (the emissions
//                                             of column t are in registers a step ahead anyway) and exchanged through LDS,
//     Q_b   = sum_{i : allele_t(i) = b} C_i   two masked wave totals of the previous column sums (S = Q_0 + Q_1),
//     N_t(j) = n_0 T_t[0][b_j] + n_1 T_t[1][b_j]   per record (two values, laid out when the record block is parked).
// So step t's state block needs only C^{t-1}, S_{t-1} (known since the middle of step t-1), and the chain
//     LDS read of the Y partials -> closed form -> two DPP wave totals -> scale
// of step t runs BESIDE the state block of step t, sliced between its row pairs.  The step is then bound by instruction
// issue, not by the latency of its exchange.  tools/lean_pipe_model.py restates the identities on a CPU.
// Same stored columns, scales, fall-back rules and resume conventions as k_sweep_lean (k_post reads what it writes).
// Reference: src/hmm.cpp:175-273 (forward column), :275-405 (backward column).
#pragma once

struct LeanSharedP {
    double psum[2][4][64];   // per wave and lane: the Y partials of a step (by step parity); plain partials when priming
    double u[4][64] __attribute__((aligned(16)));      // wave-private: the new column sums, read back by row index
    double rec[2][PG_LEAN_BLOCK][8] __attribute__((aligned(16)));
    double scal[2][64];
    v2f64 tab[2][PG_LEAN_BLOCK][4][2] __attribute__((aligned(128)));
    unsigned char comb[2][PG_LEAN_BLOCK][4][8] __attribute__((aligned(8)));
    double nn[2][PG_LEAN_BLOCK][2] __attribute__((aligned(16)));   // N(a) = n0 T[0][a] + n1 T[1][a] of a record
};

// lean_expand + the N table of the block
DEVI void leanp_expand(LeanSharedP& sh, uint32_t block /*uniform*/, uint32_t tid) {
    lean_expand(sh, block, tid);
    if (tid < 128u) {
        const uint32_t b = block & 1u, r = tid >> 1, a = tid & 1u;
        const double* rc = sh.rec[b][r];
        const unsigned long long bits = (unsigned long long)__double_as_longlong(rc[7]);
        const double n1 = (double)__popcll(bits), n0 = 64.0 - n1;
        const double T0 = a ? rc[5] : rc[4], T1 = a ? rc[6] : rc[5];   // T[0][a], T[1][a]
        sh.nn[b][r][a] = fma(n0, T0, n1 * T1);
    }
}

// what a step needs of one column: the lane's table column (emission pairs, {T[0][b_j], T[1][b_j]}), the address of
// N(b_j), the lane's allele as a 0.0 / 1.0 multiplier (the mask of the class totals), the wave's row-pair bytes
struct PipeCol {
    uint32_t tbase, nbase, cd0, cd1;
    double ajf;
};
// (the two blocks of every per-record array are contiguous: record `rel` lies at entry rel & 127)
struct PipeRaw { unsigned long long bits; u32x2 cd; };   // what a column's descriptor is made of: two LDS reads, issued a step early
DEVI PipeRaw leanp_raw(const LeanSharedP& sh, uint32_t rel /*uniform*/, uint32_t wave /*uniform*/) {
    const uint32_t r = rel & (2u * PG_LEAN_BLOCK - 1u);
    PipeRaw w;
    w.bits = (unsigned long long)__double_as_longlong((&sh.rec[0][0][0])[r * 8u + 7u]);
    w.cd = *(const u32x2*)((&sh.comb[0][0][0][0]) + r * 32u + wave * 8u);
    return w;
}
DEVI PipeCol leanp_col_of(const LeanSharedP& sh, const PipeRaw& w, uint32_t rel /*uniform*/, uint32_t lane) {
    const uint32_t r = rel & (2u * PG_LEAN_BLOCK - 1u);
    const uint32_t aj = (uint32_t)((w.bits >> lane) & 1ull);
    PipeCol pc;
    pc.tbase = (uint32_t)(uintptr_t)(LAS const unsigned char*)&sh.tab[0][0][0][0] + r * 128u + aj * 16u;
    pc.nbase = (uint32_t)(uintptr_t)(LAS const unsigned char*)&sh.nn[0][0][0] + r * 16u + aj * 8u;
    pc.cd0 = w.cd.x; pc.cd1 = w.cd.y;
    pc.ajf = (double)aj;
    return pc;
}
DEVI PipeCol leanp_col(const LeanSharedP& sh, uint32_t rel /*uniform*/, uint32_t wave /*uniform*/, uint32_t lane) {
    return leanp_col_of(sh, leanp_raw(sh, rel, wave), rel, lane);
}
template <int P>
DEVI v2f64 leanp_pair(const PipeCol& pc) {   // {e(row 2P, lane), e(row 2P + 1, lane)} of the column
    return *(LAS const v2f64*)(uintptr_t)add_byte<(P & 3)>(P < 4 ? pc.cd0 : pc.cd1, pc.tbase);
}
DEVI void leanp_pairs_all(const PipeCol& pc, double (&e)[16]) {
    static_for<0, 8>([&](auto qc) __attribute__((always_inline)) { constexpr int q = decltype(qc)::value; const v2f64 t = leanp_pair<q>(pc); e[2 * q] = t.x; e[2 * q + 1] = t.y; });
}
DEVI v2f64 leanp_T(const PipeCol& pc) { return *(LAS const v2f64*)(uintptr_t)(pc.tbase + 64u); }   // row-pair combination 2: {T[0][b_j], T[1][b_j]}
DEVI double leanp_N(const PipeCol& pc) { return *(LAS const double*)(uintptr_t)pc.nbase; }

template <int CTRL, int ROW_MASK>
DEVI double dpp_keep_f64(double v) {   // (rows outside the row mask: whatever the destination held — never read)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// one level of the two wave totals (row_shr 1, 2, 4, 8 inside the DPP rows, then row_bcast 15 / 31 across them: the
// totals end up in lane 63) — six levels, sliced between the row pairs of the state block
template <int L>
DEVI void leanp_level(double& a, double& b) {
    if constexpr (L == 0) { a += dpp_f64<0x111, 0xF, true>(a); b += dpp_f64<0x111, 0xF, true>(b); }
    else if constexpr (L == 1) { a += dpp_f64<0x112, 0xF, true>(a); b += dpp_f64<0x112, 0xF, true>(b); }
    else if constexpr (L == 2) { a += dpp_f64<0x114, 0xF, true>(a); b += dpp_f64<0x114, 0xF, true>(b); }
    else if constexpr (L == 3) { a += dpp_f64<0x118, 0xF, true>(a); b += dpp_f64<0x118, 0xF, true>(b); }
    // (rows outside the row mask keep `old` = the source: only row 3 — lane 63 — is read at the end, and it is inside both masks)
    else if constexpr (L == 4) { a += dpp_keep_f64<0x142, 0xA>(a); b += dpp_keep_f64<0x142, 0xA>(b); }
    else { a += dpp_keep_f64<0x143, 0xC>(a); b += dpp_keep_f64<0x143, 0xC>(b); }
}
DEVI void leanp_totals(double& a, double& b) {   // (not sliced: priming)
    leanp_level<0>(a, b); leanp_level<1>(a, b); leanp_level<2>(a, b); leanp_level<3>(a, b); leanp_level<4>(a, b); leanp_level<5>(a, b);
    a = readlane_f64(a, 63); b = readlane_f64(b, 63);
}

// Priming (start of a launch; behind a backward fall-back): the carried quantities by summation.  `v` = the product
// column in registers, `e1` = the emissions the first step multiplies with, `ajf1` = that column's lane allele.  The
// plain partials go through psum[rb ^ 1]; the Y partials of the first step are left in psum[rb] (the buffer it reads).
struct PipeCarry { double Cj, Crep, Q0, Q1; };
DEVI PipeCarry leanp_prime(LeanSharedP& sh, const double (&v)[16], const double (&e1)[16], double ajf1, uint32_t rb, uint32_t wave, uint32_t lane) {
    double part = 0.0, yp = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { part += v[k]; yp = fma(e1[k], v[k], yp); }
    sh.psum[rb ^ 1u][wave][lane] = part;
    lds_barrier();
    const uint32_t ri = wave * 16u + (lane & 15u);
    PipeCarry c;
    c.Cj = (sh.psum[rb ^ 1u][0][lane] + sh.psum[rb ^ 1u][1][lane]) + (sh.psum[rb ^ 1u][2][lane] + sh.psum[rb ^ 1u][3][lane]);
    c.Crep = (sh.psum[rb ^ 1u][0][ri] + sh.psum[rb ^ 1u][1][ri]) + (sh.psum[rb ^ 1u][2][ri] + sh.psum[rb ^ 1u][3][ri]);
    double m1 = c.Cj * ajf1, m0 = c.Cj - m1;
    leanp_totals(m0, m1);
    c.Q0 = m0; c.Q1 = m1;
    sh.psum[rb][wave][lane] = yp;
    lds_barrier();
    return c;
}

// One state in ONE asm statement: P' = acc + u[row lane K] * sc (DP-ALU DPP fmac, see fmac_row_bcast), the Y term of
// the state before (independent: it sits in the slot behind the DPP operation), x = e * P'.  hipcc pads every inline asm
// whose result the next VALU instruction reads with an s_nop (it cannot see that the asm holds no SDWA destination
// select): three instructions in one statement leave nothing to pad — 33 s_nop per step otherwise.
template <int K>
DEVI void leanp_state(double& acc /*in: c0 x + u_j, out: P'*/, double u, double sc, double& yacc, double ya, double yb, double& xn, double e) {
    asm("v_fmac_f64_dpp %0, %3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_e32 %1, %5, %6\n\t"
        "v_mul_f64 %2, %7, %0"
        : "+v"(acc), "+v"(yacc), "=v"(xn) : "v"(u), "v"(sc), "v"(ya), "v"(yb), "v"(e), "n"(K));
}
template <int K>
DEVI void leanp_state0(double& acc, double u, double sc, double& xn, double e) {   // (the first state: no Y term yet)
    asm("v_fmac_f64_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f64 %1, %4, %0"
        : "+v"(acc), "=v"(xn) : "v"(u), "v"(sc), "v"(e), "n"(K));
}

// the per-record arrays hold two blocks: block b is parked PARK_AHEAD steps before its first record is read
// (a step reads up to record n + 4: the raw descriptor of the column three steps on) and expanded a step later
#define PG_LEANP_PARK 6u

struct PipeFwdK { double c0s, c1s, ujs, sc, urep; };   // constants of a forward step

template <int PHASE>
DEVI void leanp_forward(const DevContig& dc, LeanSharedP& sh, uint32_t C, uint32_t chunk) {
    constexpr int HP = 64, R = 16;
    constexpr uint32_t RMASK = 0xFFFFu;
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
    const size_t colsz = (size_t)HP * HP;
    const double unif = 1.0 / 4096.0;
    LeanRecs recs{(const GAS char*)dc.frec, (int64_t)first - 1, (int64_t)C, +1, tid};   // rel k = column first - 1 + k
    recs.park(sh, 0, recs.fetch(0));
    v2f64 piece = recs.fetch(1);
    lds_barrier();
    leanp_expand(sh, 0, tid);
    lds_barrier();
    gdouble* fwd = (gdouble*)dc.fwd;
    gdouble* fscale = (gdouble*)dc.fscale;
    gu8* fallback = (gu8*)dc.fwd_fallback;
    gdouble* wr = fwd;
    gcdouble* resume = (gcdouble*)(fwd + (size_t)(lo > 0 ? lo - 1 : 0) * colsz);
    if constexpr (PHASE == 3) {
        gdouble* scr = (gdouble*)dc.scratch;
        wr = scr + (size_t)((chunk & 1u) * 2u) * K * colsz - (size_t)lo * colsz;
        if (chunk > 0) resume = (gcdouble*)(scr + ((size_t)(((chunk - 1u) & 1u) * 2u) * K + (K - 1u)) * colsz);
    }
    const size_t toff = (size_t)(i0 >> 1) * HP + lane;  // this thread's first row pair inside a column (in 16-byte units)
    auto store_col = [&](uint32_t c, const double (&v)[R]) {
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + toff;
#pragma unroll
        for (int k = 0; k < R; k += 2) dst[(size_t)(k >> 1) * HP] = v2f64{v[k], v[k + 1]};
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
    {   // column first - 1 (rel 0): x = e . P' (see lean_forward)
        const FRec r0 = read_frec(sh, 0);
        const bool aj = (r0.bits1 >> lane) & 1ull;
        const double eA = aj ? r0.E01 : r0.E00, eB = aj ? r0.E11 : r0.E01;
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(r0.bits1 >> i0) & RMASK));
        if (lo == 0) {
            const double P0 = ldexp(1.0, PG_BIAS_F);
            double pz[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { pz[k] = P0; x[k] = sel_by_bit(rb, k, eA, eB) * P0; }
            store_col(0, pz);
            if (wave == 0) fscale[0] = 1.0;
        } else {
            gcdouble2* src = (gcdouble2*)resume + toff;
#pragma unroll
            for (int k = 0; k < R; k += 2) { const v2f64 t = src[(size_t)(k >> 1) * HP]; x[k] = t.x; x[k + 1] = t.y; }
            if (!fallback[lo - 1]) {
#pragma unroll
                for (int k = 0; k < R; ++k) x[k] *= sel_by_bit(rb, k, eA, eB);
            }
        }
    }
    // The emissions of two columns are in registers: `eC` multiplies this step's states, `eN` (the next column's) weights
    // the Y partials; a pair of eC is overwritten with the column after next's right after its two states used it.
    double ea[R], eb[R];
    PipeCol pa = leanp_col(sh, 1, wave, lane), pb_ = leanp_col(sh, 2, wave, lane);   // columns first, first + 1
    PipeRaw raw = leanp_raw(sh, 3, wave);                                             // column first + 2: what the first step makes its p2 of
    leanp_pairs_all(pa, ea);
    leanp_pairs_all(pb_, eb);
    PipeCarry cy = leanp_prime(sh, x, ea, pa.ajf, (first - 1) & 1u, wave, lane);
    // Constants of step tn from the totals of column tn - 1 (S = 0: that column summed to zero — the uniform column takes
    // its place, hmm.cpp:253-267: every x, every column sum and both class totals are 0, the whole uniform step rides on
    // u_j; the caller stores the uniform column and raises the flag once the column's own stores are out).
    auto constants = [&](uint32_t tn, double c0, double c1, double c2, const PipeCarry& c, bool& fb, double& m) __attribute__((always_inline)) {
        double S = c.Q0 + c.Q1;
        double uj = fma(c2, S, c1 * c.Cj);
        fb = !(S > 0.0);
        if (__builtin_expect(fb, 0)) {
            S = 1.0;
            uj = fma(c0, unif, fma(c2, 1.0, 2.0 * c1 * (64.0 * unif)));
            c0 = 0.0;
        }
        int es = exponent_of(S) - PG_BIAS_F;
        es = es < -900 ? -900 : es;
        m = ldexp(S, -es - PG_BIAS_F);
        PipeFwdK k;
        k.sc = ldexp(1.0, -es); k.c0s = ldexp(c0, -es); k.c1s = ldexp(c1, -es); k.ujs = ldexp(uj, -es);
        k.urep = dpp_source(c1 * c.Crep);
        return k;
    };
    auto put_scale = [&](uint32_t tn, double m) __attribute__((always_inline)) {
        if (wave == 0) {  // (scalar branch)
            fsc.put(lane, tn, m);
            if ((tn & 63u) == 63u) fsc.flush(fscale, lane, tn);
        }
    };
    PipeFwdK kc;
    {
        const FRec r1 = read_frec(sh, 1);
        bool fb; double m;
        kc = constants(first, r1.c0, r1.c1, r1.c2, cy, fb, m);
        if (fb) flag_uniform(first - 1);
        if (first < hi) put_scale(first, m);
    }
    LeanTimeline tl;
    tl.init();
    // One column step: eC / pC = emissions / descriptor of column t, eN / pN = of column t + 1; pC ends up as column
    // t + 2's.  The chain of the column sums of THIS column is sliced between the sixteen states (static `slot`).
    auto step = [&](uint32_t t, double (&eC)[R], double (&eN)[R], PipeCol& pC, const PipeCol& pN) __attribute__((always_inline)) {
        const uint32_t n = t - first;                 // column t = rel n + 1
        const uint32_t pbuf = (t - 1) & 1u;
        const PipeFwdK k = kc;
        tl.template mark<0>(0.0);                     // (behind the barrier of the step before)
        double yq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) yq[q] = sh.psum[pbuf][q][lane];
        const v2f64 Tj = leanp_T(pC);
        const double Nj = leanp_N(pC);
        const double* rn = (&sh.rec[0][0][0]) + ((n + 2u) & (2u * PG_LEAN_BLOCK - 1u)) * 8u;   // record t + 1: the next step's constants
        const v2f64 n01 = *(const v2f64*)rn;
        const double n2 = rn[2];
        const PipeCol p2 = leanp_col_of(sh, raw, n + 3u, lane);   // column t + 2 (its two LDS reads were issued a step ago)
        if (((n + PG_LEANP_PARK) % PG_LEAN_BLOCK) == 0u) {        // (uniform) a few columns before the next block is needed
            const uint32_t blk = (n + PG_LEANP_PARK) / PG_LEAN_BLOCK;
            recs.park(sh, blk, piece);
            piece = recs.fetch(blk + 1u);
        } else if (((n + PG_LEANP_PARK - 1u) % PG_LEAN_BLOCK) == 0u) {
            leanp_expand(sh, (n + PG_LEANP_PARK - 1u) / PG_LEAN_BLOCK, tid);   // the block parked a step ago (a barrier lies between)
        }
        gdouble2* dst = (gdouble2*)(wr + (size_t)t * colsz) + toff;
        double yp = 0.0, yp2 = 0.0, pprev = 0.0, sprev0 = 0.0, sprev1 = 0.0;
        double Y = 0.0, m0 = 0.0, m1 = 0.0, mn = 0.0;
        PipeCarry cn{0.0, 0.0, 0.0, 0.0};
        PipeFwdK kn = k;
        bool fb = false;
        double nS = 0.0, nuj = 0.0, nc0 = 0.0;
        int nes = 0;
        lean_fence();
        tl.template mark<1>(0.0);                     // top: LDS reads issued, descriptor of column t + 2
        static_for<0, R>([&](auto kcst) __attribute__((always_inline)) {
            constexpr int s = decltype(kcst)::value;
            double pk = fma(k.c0s, x[s], k.ujs);       // P'_t(i0 + s, lane) 2^-es = c0 x + u_j + u_i
            if constexpr (s == 0) leanp_state0<s>(pk, k.urep, k.sc, x[s], eC[s]);
            else if constexpr (s & 1) leanp_state<s>(pk, k.urep, k.sc, yp, eN[s - 1], x[s - 1], x[s], eC[s]);
            else leanp_state<s>(pk, k.urep, k.sc, yp2, eN[s - 1], x[s - 1], x[s], eC[s]);
            if constexpr (s & 1) {
                constexpr int q = s >> 1;
                // (a pair's store goes out one pair late: its registers are then not the ones the next states are formed in —
                // a store's data registers must not be rewritten within two wait states)
                if constexpr (q > 0) { if (!(kLeanpExp & 1)) dst[(size_t)(q - 1) * HP] = v2f64{sprev0, sprev1}; }
                sprev0 = pprev; sprev1 = pk;
                const v2f64 t2 = leanp_pair<q>(p2);   // e_{t+2} of this row pair
                eC[s - 1] = t2.x; eC[s] = t2.y;
            } else pprev = pk;
            lean_fence();
            // ---- slot s of the column-sum chain of THIS column (independent of the state block) ----
            if constexpr (s == 1) tl.template mark<2>(x[1]);    // states 0, 1
            // (the Y partials are ~200 cycles away behind the barrier — the four waves' reads queue in one LDS pipe: first use at slot 4)
            if constexpr (s == 4) {
                Y = (yq[0] + yq[1]) + (yq[2] + yq[3]);
            } else if constexpr (s == 5) {
                const double W = fma(Tj.y, cy.Q1, Tj.x * cy.Q0);
                cn.Cj = fma(k.c0s, Y, fma(k.c1s, W, k.ujs * Nj));
                sh.u[wave][lane] = cn.Cj;
                m1 = cn.Cj * pN.ajf; m0 = cn.Cj - m1;
                tl.template mark<3>(m0);              // states 2 .. 5 + the Y partials back from LDS, closed form
            } else if constexpr (s >= 6 && s <= 11) {
                if (!(kLeanpExp & 2)) leanp_level<s - 6>(m0, m1);
                if constexpr (s == 9) cn.Crep = sh.u[wave][i0 + (lane & 15u)];
            } else if constexpr (s == 12) {
                // the next step's constants in three slices (a dependent chain: a state's worth of issue between its links)
                tl.template mark<4>(m0 + m1);         // states 6 .. 12 + six DPP levels of the two totals
                cn.Q0 = readlane_f64(m0, 63); cn.Q1 = readlane_f64(m1, 63);
                nS = cn.Q0 + cn.Q1;
                nuj = fma(n2, nS, n01.y * cn.Cj);
                nc0 = n01.x;
                fb = !(nS > 0.0);
                if (__builtin_expect(fb, 0)) {   // column t summed to zero: see `constants`
                    nS = 1.0;
                    nuj = fma(n01.x, unif, fma(n2, 1.0, 2.0 * n01.y * (64.0 * unif)));
                    nc0 = 0.0;
                }
                nes = exponent_of(nS) - PG_BIAS_F;
                nes = nes < -900 ? -900 : nes;
            } else if constexpr (s == 13) {
                mn = ldexp(nS, -nes - PG_BIAS_F);
                kn.sc = ldexp(1.0, -nes); kn.c0s = ldexp(nc0, -nes);
                pin_here(mn); pin_here(kn.sc); pin_here(kn.c0s);
            } else if constexpr (s == 14) {
                kn.c1s = ldexp(n01.y, -nes); kn.ujs = ldexp(nuj, -nes);
                kn.urep = dpp_source(n01.y * cn.Crep);
                pin_here(kn.c1s); pin_here(kn.ujs); pin_here(kn.urep);
                tl.template mark<5>(kn.ujs);          // states 13, 14 + readlanes, zero test, the next step's constants
            }
            if constexpr (s == 8) raw = leanp_raw(sh, n + 4u, wave);   // column t + 3: next step's p2 (two LDS reads, long before the barrier)
            if constexpr (s >= 4) lean_fence();
        });
        yp = fma(eN[R - 1], x[R - 1], yp);
        tl.template mark<6>(yp);                      // states 12 .. 15, the next descriptor's reads
        if (!(kLeanpExp & 1)) dst[(size_t)(R / 2 - 1) * HP] = v2f64{sprev0, sprev1};
        sh.psum[t & 1u][wave][lane] = yp + yp2;
        if (__builtin_expect(fb, 0)) flag_uniform(t);   // (behind the column's own stores)
        if (t + 1u < hi) put_scale(t + 1u, mn);
        cy = cn; kc = kn;
        pC = p2;
        tl.template mark<7>(0.0);                     // Y partial parked, per-column scalar
        lds_barrier();
        tl.template mark<8>(0.0);                     // the barrier
        tl.template fold<8>();
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);   // (see lean_forward: no load of the prologue is still in flight inside the loop)
    uint32_t t = first;
    for (; t + 1 < hi; t += 2) {
        step(t, ea, eb, pa, pb_);
        step(t + 1, eb, ea, pb_, pa);
    }
    if (t < hi) step(t, ea, eb, pa, pb_);
    if (wave == 0 && fsc.valid) fsc.flush(fscale, lane, hi - 1);
    if (kLeanTimeline && tid == 0) tl.write(dc.prof + 32);
}

struct PipeBwdK { double k0, k1, uj, urep, Snew; };   // constants of a backward step; Snew = sum of the column it stores

template <int PHASE>
DEVI void leanp_backward(const DevContig& dc, LeanSharedP& sh, uint32_t C, uint32_t chunk) {
    constexpr int HP = 64, R = 16;
    constexpr uint32_t RMASK = 0xFFFFu;
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
    const size_t colsz = (size_t)HP * HP;
    const double unif = 1.0 / 4096.0;
    LeanRecs recs{(const GAS char*)dc.frec, t0 + 1, (int64_t)C, -1, tid};   // rel k = column t0 + 1 - k
    recs.park(sh, 0, recs.fetch(0));
    v2f64 piece = recs.fetch(1);
    lds_barrier();
    leanp_expand(sh, 0, tid);
    lds_barrier();
    gdouble* cols = (gdouble*)dc.fwd;
    gdouble* bscale = (gdouble*)dc.bscale;
    gdouble* bsum = (gdouble*)dc.bsum;
    gdouble* wr = cols;
    gcdouble* resume = (gcdouble*)(cols + (size_t)(top + 1 < (int64_t)C ? top + 1 : top) * colsz);
    if constexpr (PHASE == 3) {
        gdouble* scr = (gdouble*)dc.scratch;
        wr = scr + (size_t)((chunk & 1u) * 2u + 1u) * (size_t)K * colsz - (size_t)bot * colsz;
        if (chunk > 0) resume = (gcdouble*)(scr + (size_t)(((chunk - 1u) & 1u) * 2u + 1u) * (size_t)K * colsz);
    }
    const size_t toff = (size_t)(i0 >> 1) * HP + lane;
    auto store_col = [&](int64_t c, const double (&v)[R]) {
        gdouble2* dst = (gdouble2*)(wr + (size_t)c * colsz) + toff;
#pragma unroll
        for (int k = 0; k < R; k += 2) dst[(size_t)(k >> 1) * HP] = v2f64{v[k], v[k + 1]};
    };

    ColScalars bsc{&sh.scal[0][0]}, bsm{&sh.scal[1][0]};
    double w[R], Sy;
    {
        const FRec cur = read_frec(sh, 0);  // record t0 + 1: emission of column t0 + 1
        double y[R];
        if constexpr (PHASE == 1) {
            const double B0 = ldexp(1.0, PG_BIAS_B);   // column C-1: beta~ = 1 (hmm.cpp:356-358), stored at the backward bias
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
        const bool aj = (cur.bits1 >> lane) & 1ull;
        const double eA = aj ? cur.E01 : cur.E00, eB = aj ? cur.E11 : cur.E01;
        const uint32_t rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(cur.bits1 >> i0) & RMASK));
#pragma unroll
        for (int k = 0; k < R; ++k) w[k] = y[k] * sel_by_bit(rb, k, eA, eB);
    }
    double ea[R], eb[R];
    PipeCol pa = leanp_col(sh, 1, wave, lane), pb_ = leanp_col(sh, 2, wave, lane);   // columns t0, t0 - 1
    PipeRaw raw = leanp_raw(sh, 3, wave);                                             // column t0 - 2
    leanp_pairs_all(pa, ea);
    leanp_pairs_all(pb_, eb);
    PipeCarry cy = leanp_prime(sh, w, ea, pa.ajf, (uint32_t)t0 & 1u, wave, lane);
    double one = 1.0;   // (in a register for the whole sweep: the DPP form of v_fmac_f64 takes no constant)
    asm volatile("" : "+v"(one));
    // Constants of step tn (record tn + 1: the gap tn -> tn + 1) from the totals of w = beta'_{tn+1} . e_{tn+1}: the scale
    // from sy = sum(beta'_{tn+1}), the sum of the column the step stores (Snew == 0: the zero rule, in the step itself)
    auto constants = [&](double c0, double c1, double c2, double kappa, const PipeCarry& c, double sy, double& m) __attribute__((always_inline)) {
        const double Sw = c.Q0 + c.Q1;
        int es = exponent_of(sy) - PG_BIAS_B;
        es = es < -900 ? -900 : es;
        m = ldexp(sy, -es - PG_BIAS_B);
        PipeBwdK k;
        k.k0 = ldexp(c0, -es); k.k1 = ldexp(c1, -es);
        const double k2 = ldexp(c2, -es), kap = ldexp(kappa, -es);
        k.uj = fma(k2, Sw, k.k1 * c.Cj);
        k.urep = dpp_source(k.k1 * c.Crep);
        k.Snew = kap * Sw;   // = sum(beta'_tn)
        return k;
    };
    auto put_scalars = [&](int64_t tn, double m, double Snew) __attribute__((always_inline)) {
        if (wave == 1) { asm volatile("" ::: "memory"); bsc.put(lane, (uint64_t)tn, m); }
        if (wave == 2) { asm volatile("" ::: "memory"); bsm.put(lane, (uint64_t)tn, Snew); }
        if (((uint64_t)tn & 63u) == 0u) {
            if (wave == 1) bsc.flush(bscale, lane, (uint64_t)tn);
            if (wave == 2) bsm.flush(bsum, lane, (uint64_t)tn);
        }
    };
    PipeBwdK kc{0.0, 0.0, 0.0, 0.0, 0.0};
    if (t0 >= bot) {
        const FRec cur = read_frec(sh, 0);
        double m;
        kc = constants(cur.c0, cur.c1, cur.c2, cur.kappa, cy, Sy, m);
        put_scalars(t0, m, kc.Snew);
    }
    LeanTimeline tl;
    tl.init();
    // One column step: eC / pC = emissions / descriptor of column t, eN / pN = of column t - 1; pC ends up as column t - 2's.
    auto step = [&](int64_t t, double (&eC)[R], double (&eN)[R], PipeCol& pC, const PipeCol& pN) __attribute__((always_inline)) {
        const uint32_t n = (uint32_t)(t0 - t);        // column t = rel n + 1
        const uint32_t pbuf = (uint32_t)t & 1u;
        const PipeBwdK k = kc;
        tl.template mark<0>(0.0);
        const bool zero_t = !(k.Snew > 0.0);          // beta~_t is all zero (below)
        double yq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) yq[q] = sh.psum[pbuf][q][lane];
        const v2f64 Tj = leanp_T(pC);
        const double Nj = leanp_N(pC);
        const double* rn = (&sh.rec[0][0][0]) + ((n + 1u) & (2u * PG_LEAN_BLOCK - 1u)) * 8u;   // record t: the next step's constants
        const v2f64 n01 = *(const v2f64*)rn, n23 = *(const v2f64*)(rn + 2);
        const PipeCol p2 = leanp_col_of(sh, raw, n + 3u, lane);   // column t - 2
        if (((n + PG_LEANP_PARK) % PG_LEAN_BLOCK) == 0u) {
            const uint32_t blk = (n + PG_LEANP_PARK) / PG_LEAN_BLOCK;
            recs.park(sh, blk, piece);
            piece = recs.fetch(blk + 1u);
        } else if (((n + PG_LEANP_PARK - 1u) % PG_LEAN_BLOCK) == 0u) {
            leanp_expand(sh, (n + PG_LEANP_PARK - 1u) / PG_LEAN_BLOCK, tid);
        }
        gdouble2* dst = (gdouble2*)(wr + (size_t)t * colsz) + toff;
        double yp = 0.0, yp2 = 0.0, yprev = 0.0, sprev0 = 0.0, sprev1 = 0.0;
        double Y = 0.0, m0 = 0.0, m1 = 0.0, mn = 0.0;
        PipeCarry cn{0.0, 0.0, 0.0, 0.0};
        PipeBwdK kn = k;
        double nk2 = 0.0, nkap = 0.0;
        int nes = 0;
        lean_fence();
        tl.template mark<1>(0.0);
        static_for<0, R>([&](auto kcst) __attribute__((always_inline)) {
            constexpr int s = decltype(kcst)::value;
            double yk = fma(k.k0, w[s], k.uj);           // beta'_t = k0 w + u_j + u_i
            if constexpr (s == 0) leanp_state0<s>(yk, k.urep, one, w[s], eC[s]);
            else if constexpr (s & 1) leanp_state<s>(yk, k.urep, one, yp, eN[s - 1], w[s - 1], w[s], eC[s]);
            else leanp_state<s>(yk, k.urep, one, yp2, eN[s - 1], w[s - 1], w[s], eC[s]);
            if constexpr (s & 1) {
                constexpr int q = s >> 1;
                if constexpr (q > 0) { if (!(kLeanpExp & 1)) dst[(size_t)(q - 1) * HP] = v2f64{sprev0, sprev1}; }
                sprev0 = yprev; sprev1 = yk;
                const v2f64 t2 = leanp_pair<q>(p2);   // e_{t-2} of this row pair
                eC[s - 1] = t2.x; eC[s] = t2.y;
            } else yprev = yk;
            lean_fence();
            if constexpr (s == 1) tl.template mark<2>(w[1]);
            if constexpr (s == 1) {
                // the part of the next step's constants that hangs on this step's Snew alone (known since the step before)
                nes = exponent_of(k.Snew) - PG_BIAS_B;
                nes = nes < -900 ? -900 : nes;
                mn = ldexp(k.Snew, -nes - PG_BIAS_B);
                kn.k0 = ldexp(n01.x, -nes); kn.k1 = ldexp(n01.y, -nes);
                pin_here(mn); pin_here(kn.k0); pin_here(kn.k1);
            } else if constexpr (s == 2) {
                nk2 = ldexp(n23.x, -nes); nkap = ldexp(n23.y, -nes);
                pin_here(nk2); pin_here(nkap);
            } else if constexpr (s == 4) {
                Y = (yq[0] + yq[1]) + (yq[2] + yq[3]);
            } else if constexpr (s == 5) {
                const double W = fma(Tj.y, cy.Q1, Tj.x * cy.Q0);
                cn.Cj = fma(k.k0, Y, fma(k.k1, W, k.uj * Nj));
                sh.u[wave][lane] = cn.Cj;
                m1 = cn.Cj * pN.ajf; m0 = cn.Cj - m1;
                tl.template mark<3>(m0);
            } else if constexpr (s >= 6 && s <= 11) {
                if (!(kLeanpExp & 2)) leanp_level<s - 6>(m0, m1);
                if constexpr (s == 9) cn.Crep = sh.u[wave][i0 + (lane & 15u)];
            } else if constexpr (s == 12) {
                tl.template mark<4>(m0 + m1);
                cn.Q0 = readlane_f64(m0, 63); cn.Q1 = readlane_f64(m1, 63);
            } else if constexpr (s == 13) {
                const double Sw = cn.Q0 + cn.Q1;
                kn.uj = fma(nk2, Sw, kn.k1 * cn.Cj);
                kn.Snew = nkap * Sw;   // = sum(beta'_{t-1})
                kn.urep = dpp_source(kn.k1 * cn.Crep);
                pin_here(kn.uj); pin_here(kn.urep); pin_here(kn.Snew);
                tl.template mark<5>(kn.uj);
            }
            if constexpr (s == 8) raw = leanp_raw(sh, n + 4u, wave);   // column t - 3: next step's p2
            if constexpr (s >= 1) lean_fence();
        });
        yp = fma(eN[R - 1], w[R - 1], yp);
        tl.template mark<6>(yp);
        if (!(kLeanpExp & 1)) dst[(size_t)(R / 2 - 1) * HP] = v2f64{sprev0, sprev1};
        if (__builtin_expect(zero_t, 0)) {
            // beta~_t is all zero (a sum of non-negative terms: every y_k above IS 0, and so is what was stored): its own
            // posteriors are 0, the next step starts from the uniform column (hmm.cpp:374-380) — w = unif . e_t, primed again
            double et[R];   // (eC holds the column after next's emissions by now: this column's are fetched again)
            leanp_pairs_all(leanp_col(sh, n + 1u, wave, lane), et);
#pragma unroll
            for (int q = 0; q < R; ++q) w[q] = unif * et[q];
            lds_barrier();   // (every wave is done with the Y partials of this step: the buffer takes the plain partials)
            cn = leanp_prime(sh, w, eN, pN.ajf, (uint32_t)(t - 1) & 1u, wave, lane);
            kn = constants(n01.x, n01.y, n23.x, n23.y, cn, 1.0, mn);
        } else {
            sh.psum[(uint32_t)(t - 1) & 1u][wave][lane] = yp + yp2;
        }
        if (t - 1 >= bot) put_scalars(t - 1, mn, kn.Snew);
        cy = cn; kc = kn;
        pC = p2;
        tl.template mark<7>(0.0);
        lds_barrier();
        tl.template mark<8>(0.0);
        tl.template fold<8>();
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int64_t t = t0;
    for (; t - 1 >= bot; t -= 2) {
        step(t, ea, eb, pa, pb_);
        step(t - 1, eb, ea, pb_, pa);
    }
    if (t >= bot) step(t, ea, eb, pa, pb_);
    if (wave == 1 && bsc.valid) bsc.flush(bscale, lane, (uint64_t)bot);
    if (wave == 2 && bsm.valid) bsm.flush(bsum, lane, (uint64_t)bot);
    if (kLeanTimeline && tid == 0) tl.write(dc.prof + 48);
}

template <int PHASE>
__global__ __launch_bounds__(256) void k_sweep_leanp(const DevContig* __restrict__ contigs, uint32_t chunk) {
    __shared__ LeanSharedP sh;
    const DevContig& dc = contigs[blockIdx.x];
    if (dc.lean != 2u) return;   // (DevContig::lean: 2 = the pipelined step)
    const uint32_t C = (uint32_t)__builtin_amdgcn_readfirstlane((int)*dc.n_cols);
    if (C == 0) return;
    const unsigned long long t_begin = kChainProf ? __builtin_amdgcn_s_memtime() : 0ull;
    if (blockIdx.y == 0) leanp_forward<PHASE>(dc, sh, C, chunk);
    else leanp_backward<PHASE>(dc, sh, C, chunk);
    if (kChainProf && threadIdx.x == 0) {  // -DPG_CHAIN_PROF builds only: cycles of this role's launch (last chunk wins)
        unsigned long long* o = dc.prof + (blockIdx.y == 0 ? 0 : 16) + (PHASE == 1 ? 0 : 8);
        o[0] = __builtin_amdgcn_s_memtime() - t_begin;
    }
}
