// pg_devmath.h — device arithmetic shared by the kernel files (included by .hip sources only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

// ------------------------------------------------------------------------------------------
//  1 - exp(-x) as the reference forms it (src/transitionprobabilitycomputer.cpp:16).
//  expm1 removes the 1-exp(-x) cancellation that plain fp64 would suffer.
//  The reference forms 1 - expl(-x) in 80-bit arithmetic, i.e. with e^-x rounded to a multiple of
//  2^-64 (for e^-x in [1/2,1)): its result is the exact difference rounded to a multiple of 2^-64.
//  That rounding is its dominant error once x is tiny (relative 5.4e-20/x: 1e-10 at the default
//  parameters, 1e-7..1e-6 at recombrate 1e-3), so it is reproduced here — the bar is parity with the
//  reference, not with the exact value — including q == 0 once x < 2^-65.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double one_minus_exp_neg_like_reference(double x) {
    double t = -expm1(-x);
    if (x < 0.6931471805599453) {
        const double s = t * 0x1p64;
        if (s < 0x1p52) t = rint(s) * 0x1p-64;
    }
    return t;
}

// {no switch, one switch, two switches} = {p^2, p q, q^2} (src/transitionprobabilitycomputer.cpp:14-18, :33-39)
__device__ __forceinline__ void transition_probs_f64(double d, uint32_t H, int uniform, double& t0, double& t1, double& t2) {
    if (uniform) { t0 = t1 = t2 = 1.0; return; }
    const double x = d / (double)H;
    const double q = one_minus_exp_neg_like_reference(x) / (double)H;
    const double p = exp(-x) + q;
    t0 = p * p; t1 = p * q; t2 = q * q;
}
