// fit_kernel.hpp -- the per-voxel fit kernel (motor:113-162 + motor:443-472) and what it is made of: the lambda searches (scipy's bounded Brent
// restated, with the tie guard), the L-curve corner, the metrics epilogue, the plan-level seed and factor-table kernels, and the launcher
// template.  Shared by the translation units of libmet2_hip.so: met2_hip.hip holds the C ABI and every other kernel; with -DMET2_SPLIT_TU
// (the shipped build) the fit kernels are instantiated in met2_fit_*.hip, one family of methods per file, so that they compile side by side
// (one translation unit took 100 s); without it (development builds: -DMET2_ONLY, -DMET2_CYCSTATS ...) met2_hip.hip instantiates them all.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/met2_hip.h"
#include "abi_common.hpp"
#include "nnls_wave.hpp"
#include "objectives.hpp"

using namespace met2;

struct SortBufs {
    int *key;          // [nvox]  fa index, or -1 if not fitted
    int *perm;         // [nvox]  fitted voxels ordered by fa
    int *hist;         // [nfa]
    int *cursor;       // [nfa]
    int *bucket_start; // [nfa+1]
    int *chunk_start;  // [nfa+1]
    int *queue;        // [1]
    int *xq;           // [8]  one queue cursor per XCD (fit kernel)
    int *err;          // [3]  [0] bit0: FA index out of range; [1], [2]: tail and head of the spill-over queue
    int *ovf;          // [nvox] the spill-over queue: voxels whose passive set outgrew the wave's LDS region
};

// ------------------------------------------------------------------------------------------
// fit kernel
// ------------------------------------------------------------------------------------------
#ifndef MET2_BAYES_TABLE
#define MET2_BAYES_TABLE 10               // shared Brent abscissae with plan-level factors (0 disables the tables)
#endif
// Internal kernel variant of MET2_GCV (not part of the ABI): the trace from the 17 x 17 form in the plan's low-rank basis (objectives.hpp,
// gcv_basis_kernel); chosen by fit_impl when every flip angle's dictionary is of numerical rank <= 16.
#define MET2_GCV_LR 6
constexpr int LC_SAVE_DOUBLES = 18;      // per lane, two bins per lane: log norms 2, kept iterates 8, iterate 2, and 12 ints (kept sets 8, position + pivot order 4)

struct FitArgs {
    int n, m, nfa, kmax, waves, chunk;
    int wave_doubles;   // LDS doubles owned by each wave (>= kmax(kmax+1)/2; n*n for GCV)
    int method, nlam, maxfun;
    double x2_factor, t2sparc_lambda, xtol;
    double cut_m, cut_ie;
    double lam_lo, lam_hi;                // the interval of the method's lambda search (met2_options: x2_lo .. bayes_hi).  Read by the BIG = true
                                          // instance of the voxel routine only: the fit kernels' own instance carries the reference's literals (with the
                                          // interval in registers fit_kernel<X2, 1> went from 112 to 140 bytes of scratch per lane -- out of the L2s:
                                          // 1.4x -> 2.8x of the algorithmic HBM bytes), and a plan with other intervals sends EVERY voxel through the
                                          // spill-over kernel (all_queued)
    int all_queued;                       // the spill-over kernel takes every fitted voxel (the sorted list sb.perm), not the queue
    double log_detL;
    const double *Dfa;    // [nfa][m][n]
    const double *Bfa;    // [nfa][n][n]
    const double *Dtfa;   // [nfa][n][m]
    const double *Aq;     // [nfa][n][16]: Q^T D of the flip angle, transposed (MET2_GCV_LR), or NULL
    const double *kband;  // [5][64]
    const double *lband;  // [5][64]
    const double *Kd;     // [n][n] dense L^T L
    const double *lam_grid;
    const double *t2s;    // [n]
    const double *data;   // echo e of voxel v at data[v * vs + e * es]
    int64_t vs, es;
    SortBufs sb;
    double *fsol, *sig, *reg, *lam, *maps;
    int32_t *status;
    int64_t nvox;
    const char *seed;                     // [nfa] SeedRec: first-Brent-point seeds of the method (seed_kernel), or NULL
    const double *btab;                   // BayesReg: [nfa][nbtab][btab_stride] factors of B + lambda_j K and, behind each, log det (bayes_table_kernel), or NULL
    int nbtab, btab_stride;
    double *chol;                         // BayesReg at two bins per lane: [grid * waves][chol_stride] scratch for the factor (chol_lean), or NULL
    int chol_stride;
    double *big;                          // [grid * waves][big_stride]: every wave's spill-over slot for factor columns >= kmax (nnls_big.hpp), or NULL
    int big_stride;
    double *lc_save;                      // L-curve at two bins per lane: [lc_cap][LC_SAVE_DOUBLES][64] sweep states of queued voxels (fit_voxel), or NULL
    int spill_w2;                         // dev: the spill-over kernel's waves per workgroup (0: its own choice)
    int *lc_at;                           // [lc_cap]: the grid point the sweep goes on from (nlam: the solve at the corner)
    int lc_cap;
    double blam[MET2_BAYES_TABLE];        // the shared Brent abscissae lambda_j
};

// The searches below run redundantly on all lanes, on wave-uniform doubles that the VALU computes into vector registers: eleven of them live across
// every objective evaluation (a whole NNLS solve), 22 VGPRs of the solver's budget.  uni() moves such a value into a scalar register pair
// (v_readfirstlane x 2); what the scalar file cannot hold the compiler keeps in the lanes of one VGPR (v_writelane), 64 scalars per register.
// Per kernel (bit of MET2_UNI_MASK: X2 1 | 2, GCV 4 | 8, BayesReg 16 | 32 at one | two bins per lane): measured, the scalar copies cost the
// two-bins-per-lane X2 and GCV kernels more in lane reads and writes than their (L2-resident) spills did.
#ifndef MET2_UNI_MASK
#define MET2_UNI_MASK 37
#endif
constexpr bool uni_brent(int method, int nb)
{
    const int base = method >= 10 ? method - 10 : method;
    const int bit = base == MET2_X2 ? 0 : ((base == MET2_GCV || base == MET2_GCV_LR) ? 2 : (base == MET2_BAYESREG ? 4 : -1));
    return bit >= 0 && ((MET2_UNI_MASK >> (bit + (nb == 2 ? 1 : 0))) & 1);
}
template <bool PIN>
__device__ __forceinline__ double uni(double v)
{
    if constexpr (PIN) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    } else return v;
}
#define MET2_UNI_PIN() do { a = uni<PIN>(a); b = uni<PIN>(b); fulc = uni<PIN>(fulc); nfc = uni<PIN>(nfc); xf = uni<PIN>(xf); rat = uni<PIN>(rat); e = uni<PIN>(e); \
                            x = uni<PIN>(x); fx = uni<PIN>(fx); ffulc = uni<PIN>(ffulc); fnfc = uni<PIN>(fnfc); } while (0)

// SciPy's bounded Brent (scipy.optimize.fminbound, called at algorithms.py:219,280 and
// bayesian_interpolation.py:101), restated; executed redundantly by all lanes on uniform values.
// on_best() is called whenever the abscissa just evaluated becomes Brent's best point xf (the value fminbound returns): callers
// keep the solver state of that evaluation and skip the solve scipy's callers repeat at the returned lambda.
template <bool PIN, class F, class G>
__device__ __forceinline__ double fminbound_dev(F &&fn, G &&on_best, double x1, double x2, double xatol, int maxfun, int &flag)
{
    const double sqrt_eps = sqrt(2.2e-16);
    const double golden_mean = 0.5 * (3.0 - sqrt(5.0));
    double a = x1, b = x2;
    double fulc = a + golden_mean * (b - a);
    double nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf;
    double fx = fn(x);
    on_best();
    int num = 1;
    flag = 0;
    double fu = INFINITY;
    double ffulc = fx, fnfc = fx;
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
    double tol2 = 2.0 * tol1;
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        bool golden = true;
        if (fabs(e) > tol1) {
            golden = false;
            double r = (xf - nfc) * (fx - ffulc);
            double q = (xf - fulc) * (fx - fnfc);
            double p = (xf - fulc) * q - (xf - nfc) * r;
            q = 2.0 * (q - r);
            if (q > 0.0) p = -p;
            q = fabs(q);
            r = e;
            e = rat;
            if ((fabs(p) < fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
                rat = (p + 0.0) / q;
                x = xf + rat;
                if (((x - a) < tol2) || ((b - x) < tol2)) {
                    double d = xm - xf;
                    double si = (double)((d > 0.0) - (d < 0.0) + (d == 0.0));
                    rat = tol1 * si;
                }
            } else golden = true;
        }
        if (golden) {
            e = (xf >= xm) ? a - xf : b - xf;
            rat = golden_mean * e;
        }
        double si = (double)((rat > 0.0) - (rat < 0.0) + (rat == 0.0));
        double ar = fabs(rat);
        x = xf + si * (ar > tol1 ? ar : tol1);
        MET2_UNI_PIN();
        fu = fn(x);
        num++;
        if (fu <= fx) {
            if (x >= xf) a = xf; else b = xf;
            fulc = nfc; ffulc = fnfc;
            nfc = xf; fnfc = fx;
            xf = x; fx = fu;
            on_best();
        } else {
            if (x < xf) a = x; else b = x;
            if ((fu <= fnfc) || (nfc == xf)) {
                fulc = nfc; ffulc = fnfc;
                nfc = x; fnfc = fu;
            } else if ((fu <= ffulc) || (fulc == xf) || (fulc == nfc)) {
                fulc = x; ffulc = fu;
            }
        }
        xm = 0.5 * (a + b);
        tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
        tol2 = 2.0 * tol1;
        if (num >= maxfun) { flag = 1; break; }
    }
    if (isnan(xf) || isnan(fx) || isnan(fu)) flag = 2;
    return xf;
}

// X2: Brent's near-ties are decided on refined objective values (fminbound_tie_dev).  Measured on one MI355X (profiles/r04_tie_guard.txt):
// the 13 voxels of tests/golden/golden_x2_failset.npz (HIP and the CPU checker disagree) -- HIP equals the REFERENCE in 9 instead of 4, as the
// checker does; the 65 536-voxel reference fixture -- 2 voxels beyond 1e-5 instead of 3 (the checker: 2); 3.4 % of the voxels take a
// refined evaluation; configs[1] 149.2 -> 150.1 ms (+0.4 % for the code being there, the rest for the refinements).
// 0: scipy's search verbatim (fminbound_dev); 2: also flag the voxels that refined and why (status bits 64, 256..4096: debugging).
#ifndef MET2_TIE_GUARD
#define MET2_TIE_GUARD 1
#endif
#ifndef MET2_TIE_ABS
#define MET2_TIE_ABS 1e-9      // objective values closer than this are a tie (the Gram-form noise is ~1e-10 of SSE / SSE_0)
#endif
#ifndef MET2_TIE_REL
#define MET2_TIE_REL 1e-4      // a margin of the parabola's acceptance tests below this share of its terms is a tie (1e-3: same results, 3.8 % of the voxels)
#endif
// The same search with a guard on its comparisons (X2, MET2_TIE_GUARD).  fn(x, refined): the objective at x; refined = true asks for the value
// after one step of iterative refinement of the solve (refine_csne: the Gram-form solve carries ~1e-10 of noise into the objective,
// the QR-form solve of the reference ~1e-13).  Whenever a decision of the search -- the three acceptance tests of the parabolic step,
// `fu <= fx`, `fu <= fnfc`, `fu <= ffulc` -- is closer than that noise can decide, the values involved are evaluated again, refined
// (each retained point at most once), and the decision is taken on those.  The re-evaluations are not counted in `num`: the sequence
// of abscissae is scipy's.  on_best() follows xf as before (also when xf's value has just been refined: the state in hand is xf's).
template <bool PIN, class F, class G>
__device__ __forceinline__ double fminbound_tie_dev(F &&fn, G &&on_best, double x1, double x2, double xatol, int maxfun, int &flag, int &nref)
{
    const double sqrt_eps = sqrt(2.2e-16);
    const double golden_mean = 0.5 * (3.0 - sqrt(5.0));
    const double TAU = MET2_TIE_ABS, KAP = MET2_TIE_REL;            // |fu - f| below TAU; a margin of the parabola's tests below KAP of its terms
    double a = x1, b = x2;
    double fulc = a + golden_mean * (b - a);
    double nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf;
    double fx = fn(x, false);
    on_best();
    int num = 1;
    flag = 0;
    double fu = INFINITY;
    double ffulc = fx, fnfc = fx;
    bool rx = false, rn = false, rf = false;                         // fx, fnfc, ffulc are refined values
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
    double tol2 = 2.0 * tol1;
    // refine the three retained values (distinct abscissae only; xf last, so that the solver's state -- and on_best's copy -- is xf's)
    auto refine3 = [&]() {
        if (!rf) { ffulc = (fulc == nfc && rn) ? fnfc : ((fulc == xf && rx) ? fx : fn(fulc, true)); rf = true; ++nref; }
        if (!rn) { fnfc = (nfc == fulc) ? ffulc : ((nfc == xf && rx) ? fx : fn(nfc, true)); rn = true; ++nref; }
        if (!rx) { fx = (xf == nfc) ? fnfc : ((xf == fulc) ? ffulc : fn(xf, true)); rx = true; ++nref; if (xf != nfc && xf != fulc) on_best(); }
    };
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        bool golden = true;
        if (fabs(e) > tol1) {
            golden = false;
            double r, q, p;
            for (int pass = 0; pass < 2; ++pass) {
                r = (xf - nfc) * (fx - ffulc);
                q = (xf - fulc) * (fx - fnfc);
                p = (xf - fulc) * q - (xf - nfc) * r;
                q = 2.0 * (q - r);
                if (q > 0.0) p = -p;
                q = fabs(q);
                if (pass == 1 || (rx && rn && rf)) break;
                // how close are the three tests?  p and q are differences of products of the f-differences: judge every margin against
                // the size of the terms it is the difference of
                const double t1 = fabs(0.5 * q * e), t2 = q * (a - xf), t3 = q * (b - xf);
                const bool c1 = fabs(fabs(p) - t1) < KAP * (fabs(p) + t1), c2 = fabs(p - t2) < KAP * (fabs(p) + fabs(t2)), c3 = fabs(t3 - p) < KAP * (fabs(p) + fabs(t3));
                const bool c4 = (fulc != xf && fabs(fx - ffulc) < TAU) || (nfc != xf && fabs(fx - fnfc) < TAU);      // (a retained point that IS xf: no parabola, not a tie)
                const bool close = c1 || c2 || c3 || c4;
                if (__builtin_expect(!close, 1)) break;
                nref |= (c1 ? 1 << 8 : 0) | (c2 ? 1 << 9 : 0) | (c3 ? 1 << 10 : 0) | (c4 ? 1 << 11 : 0);
                refine3();
            }
            r = e;
            e = rat;
            if ((fabs(p) < fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
                rat = (p + 0.0) / q;
                x = xf + rat;
                if (((x - a) < tol2) || ((b - x) < tol2)) {
                    double d = xm - xf;
                    double si = (double)((d > 0.0) - (d < 0.0) + (d == 0.0));
                    rat = tol1 * si;
                }
            } else golden = true;
        }
        if (golden) {
            e = (xf >= xm) ? a - xf : b - xf;
            rat = golden_mean * e;
        }
        double si = (double)((rat > 0.0) - (rat < 0.0) + (rat == 0.0));
        double ar = fabs(rat);
        x = xf + si * (ar > tol1 ? ar : tol1);
        MET2_UNI_PIN();
        fu = fn(x, false);
        bool ru = false;
        num++;
        if (__builtin_expect(fabs(fu - fx) < TAU || fabs(fu - fnfc) < TAU || fabs(fu - ffulc) < TAU, 0)) {
            nref |= 1 << 12;
            refine3();                                                // (leaves the solver at xf)
            fu = fn(x, true); ru = true; ++nref;
        }
        if (fu <= fx) {
            if (x >= xf) a = xf; else b = xf;
            fulc = nfc; ffulc = fnfc; rf = rn;
            nfc = xf; fnfc = fx; rn = rx;
            xf = x; fx = fu; rx = ru;
            on_best();
        } else {
            if (x < xf) a = x; else b = x;
            if ((fu <= fnfc) || (nfc == xf)) {
                fulc = nfc; ffulc = fnfc; rf = rn;
                nfc = x; fnfc = fu; rn = ru;
            } else if ((fu <= ffulc) || (fulc == xf) || (fulc == nfc)) {
                fulc = x; ffulc = fu; rf = ru;
            }
        }
        xm = 0.5 * (a + b);
        tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
        tol2 = 2.0 * tol1;
        if (num >= maxfun) { flag = 1; break; }
    }
    if (isnan(xf) || isnan(fx) || isnan(fu)) flag = 2;
    return xf;
}

// L-curve corner (algorithms.py:150-206): lane i < nl holds point i
__device__ __forceinline__ double scale_curve_dev(double a, int nl, int lane)
{
    double vmin = wave_min(lane < nl ? a : INFINITY);
    double vmax = wave_max(lane < nl ? a : -INFINITY);
    const double l = -10.0, u = 10.0;
    double s = (u - l) / (vmax - vmin), off = (u * vmin - l * vmax) / (u - l);
    return s * (a - off);
}
__device__ __forceinline__ int select_corner_dev(double le, double ln, int nl, int lane)
{
    double xs = scale_curve_dev(le, nl, lane), ys = scale_curve_dev(ln, nl, lane);
    const double cte = 7.0 * M_PI / 8.0;
    const double c0 = bcast(xs, nl - 1), c1 = bcast(ys, nl - 1);
    double best = INFINITY; int bestk = 1 << 20;
    const double a0 = xs, a1 = ys;
    const double ac = sqrt((a0 - c0) * (a0 - c0) + (a1 - c1) * (a1 - c1));
    for (int k = 0; k < nl - 2; ++k) {
        double b0 = bcast(xs, k), b1 = bcast(ys, k);
        double ab = sqrt((a0 - b0) * (a0 - b0) + (a1 - b1) * (a1 - b1));
        double bc = sqrt((b0 - c0) * (b0 - c0) + (b1 - c1) * (b1 - c1));
        double cosa = (ab * ab + ac * ac - bc * bc) / (2.0 * ab * ac);
        double t = (1.0 < cosa) ? 1.0 : cosa;
        cosa = (t > -1.0) ? t : -1.0;
        double ang = acos(cosa);
        double area = 0.5 * ((b0 - a0) * (a1 - c1) - (a0 - c0) * (b1 - a1));
        bool ok = (lane > k) && (lane < nl - 1) && (area > 0.0) && (ang < cte) && (ang < best);
        if (ok) { best = ang; bestk = k; }
    }
    double amin = wave_min(best);
    if (!(amin < INFINITY)) return nl - 1;
    bool tie = (best == amin);
    double kmin = wave_min(tie ? (double)bestk : 1e9);
    return first_lane(ballot(tie && (double)bestk == kmin));
}

// Per-lane constants of the metrics windows (motor:221-224, 444)
template <int NB>
struct MetricLanes {
    double logt2[NB];
    bool isM[NB], isIE[NB], isCSF[NB];
};

template <int NB>
__device__ __forceinline__ void metric_lanes(MetricLanes<NB> &ml, const double *t2s, int n, double cut_m, double cut_ie, int lane)
{
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = lane + 64 * b;
        const double t2 = (j < n) ? t2s[j] : 1.0;
        ml.logt2[b] = (j < n) ? log(t2) : 0.0;
        ml.isM[b] = (j < n) && (t2 <= cut_m);
        ml.isIE[b] = (j < n) && (t2 > cut_m) && (t2 <= cut_ie);
        ml.isCSF[b] = (j < n) && (t2 >= cut_ie);
    }
}

// motor:448-468 for one voxel held bin-indexed in xs (already un-normalised); lane 0 writes the six maps
template <int NB>
__device__ __forceinline__ void write_metrics(const MetricLanes<NB> &ml, const double (&xs)[NB], bool mk, double *maps, int64_t nvox,
                                              int64_t v, int lane)
{
    const double epsilon = 1.0e-16;
    double tot = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) tot += xs[b];
    const double vt = wave_sum(tot) + epsilon;
    double fm = 0.0, fie = 0.0, fcsf = 0.0, lm = 0.0, lie = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const double xn = xs[b] / vt;
        fm += ml.isM[b] ? xn : 0.0;
        fie += ml.isIE[b] ? xn : 0.0;
        fcsf += ml.isCSF[b] ? xn : 0.0;
        lm += ml.isM[b] ? xn * ml.logt2[b] : 0.0;
        lie += ml.isIE[b] ? xn * ml.logt2[b] : 0.0;
    }
    fcsf = wave_sum(fcsf);
    wave_sum2(fm, fie);
    wave_sum2(lm, lie);
    if (lane == 0) {
        maps[0 * nvox + v] = mk ? fm : 0.0;
        maps[1 * nvox + v] = mk ? fie : 0.0;
        maps[2 * nvox + v] = mk ? fcsf : 0.0;
        maps[3 * nvox + v] = mk ? exp(lm / (fm + epsilon)) : 0.0;
        maps[4 * nvox + v] = mk ? exp(lie / (fie + epsilon)) : 0.0;
        maps[5 * nvox + v] = mk ? vt : 0.0;
    }
}

template <int NB>
__device__ __forceinline__ void load_band(Band<NB> &bd, const double *kband, const double *lband, int lane)
{
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int d = 0; d < 5; ++d) bd.lb[b][d] = lband[d * 128 + lane + 64 * b];
}

// METHOD: met2_method, or 10 + method for the objective-grid diagnostic.  NB: T2 bins per lane.
// D, D^T, B of the voxel's flip angle and K are read through L1/L2 (an LDS-staged variant was measured slower in rounds 1
// and 2 -- it costs three resident waves per CU -- and has been removed).
// Waves per workgroup the kernel is compiled for: 16 (128 VGPRs) where the method fits that budget without
// spilling (NNLS, T2SPARC, X2, L-curve), 12 (168 VGPRs) for the two with a second large phase (GCV, BayesReg).
#ifndef MET2_GCV_WAVES
#define MET2_GCV_WAVES 12      // measured, GCV/L2 at 32x60, 131 072 voxels: 8 waves per CU (198 VGPRs, no spills) 585 k voxels/s; 12 (168 VGPRs) 709 k
                               // with 1.4 KB of HBM traffic per voxel; 16 (128 VGPRs) 729 k but 135 KB per voxel of scratch spills
#endif
#ifndef MET2_BAYES_WAVES
#define MET2_BAYES_WAVES 12
#endif
#ifndef MET2_ONE_REFAC
#define MET2_ONE_REFAC 3      // the GCV and BayesReg kernels at two bins per lane, while k <= 64: 2 = the warm re-factorisation takes the one-slot
                              // row-by-row form, 3 = the back substitutions run on one slot too (0: two slots, blocked MFMA re-factorisation).
                              // Their registers do not hold the one-slot legs of the whole iteration (nnls_wave.hpp: MET2_ONE_SLOT), but these
                              // two routines they do.  GCV of configs[4] on 131 072 voxels: 257.1 (0) -> 244.8 (2) -> 222.7 ms (3; with the
                              // removals and appends on one slot as well: 224.6 ms at 103 spilled VGPRs); BayesReg at 48 x 120 on 32 768:
                              // 41.5 -> 39.9 -> 38.2 ms (the fourth level: 71 ms).  profiles/r03_other_ab.txt
#endif
__host__ __device__ constexpr int method_max_waves(int method, int nb = 1)
{
    const int base = method >= 10 ? method - 10 : method;
    if (base == MET2_GCV || base == MET2_GCV_LR) return (nb == 2) ? 8 : MET2_GCV_WAVES;      // two bins per lane: the LDS holds 7 waves anyway -> 256 VGPRs, no spills;
                                                                      // one bin per lane: the latency-bound recurrences want waves, the spills of a 128-VGPR build go to HBM
    if (base == MET2_BAYESREG) return (nb == 2) ? 8 : MET2_BAYES_WAVES;  // two bins per lane: 241 VGPRs; the factor is built a block row at a time (chol_lean)
    // two bins per lane: the factor's LDS footprint (kmax = 72 at nT2 = 120) holds 7 waves per CU anyway, so those kernels are
    // compiled for 8 (256 VGPRs) instead of spilling at 128 (X2 at 48 x 120: 103 spilled VGPRs)
    if (base <= MET2_LCURVE) return (nb == 2) ? 8 : 16;
    return 12;
}

// Seeds for the first Brent point.  The first abscissa of scipy's bounded Brent is a + 0.382 (b - a) for every voxel, and at
// that (large) lambda the passive set is broad and nearly the same for all voxels of a flip angle, while the lambda = 0
// solution a voxel would otherwise start from has ~8 bins: growing it to ~44 bin by bin (one forward substitution, one
// triangular solve and one dual per bin) was ~10 % of the X2 kernel.  seed_kernel solves one canonical signal (a two-peak
// spectrum pushed through the flip angle's dictionary) at that lambda per flip angle when the plan's dictionary or penalty
// changes; every voxel's first evaluation starts from that passive set and iterate: one refactorisation and a few exchanges.
// Any x >= 0 is a feasible start for Lawson-Hanson and the regularised problem is strictly convex, so the solution is the
// cold-start one up to rounding, and because the seed depends on the plan only, a voxel's result stays independent of its
// neighbours and of the order of the voxel list.  T2SPARC's single solve at its fixed lambda is seeded the same way.  The lambda = 0 solves keep the cold path: x(0) need not be unique and NNLS,
// the L-curve and BayesReg's degrees of freedom use x(0) itself.
struct SeedArgs {
    int n, m, nfa;
    const double *Dfa, *Bfa, *Dtfa, *kband, *lband, *Kd;
    double lam[4];          // slot 0: X2, slot 1: BayesReg, slot 2: T2SPARC's fixed lambda, slot 3: GCV (the first Brent abscissa of each method's interval)
    char *out;              // [4][nfa] SeedRec
};
// one record per (slot, flip angle): a single kernel-argument pointer reaches all of it (the fit kernels are short of SGPRs)
struct SeedRec {
    double x[128];          // bin-indexed iterate
    int pos[128];           // bin -> position (-1 outside the set)
    int ord[128];           // position -> bin
    int k, pad[3];
};

template <int NB>
__global__ __launch_bounds__(64) void seed_kernel(SeedArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id(), n = A.n, m = A.m, fa = (int)blockIdx.x, slot = (int)blockIdx.y;
    WaveShared S;
    S.R = smem; S.n = n; S.m = m; S.kmax = n; S.rcap = col_base(n); S.K = A.Kd; S.kband = A.kband;
    S.B = A.Bfa + (size_t)fa * n * n; S.D = A.Dfa + (size_t)fa * m * n; S.Dt = A.Dtfa + (size_t)fa * m * n; S.DtG = S.Dt;
    S.bstride = n; S.dstride = n; S.dtstride = m; S.buffer_rows = true; S.reorder = true; S.have_bdiag = false; S.bdiag[0] = S.bdiag[1] = 0.0;
    Band<NB> bd;
    load_band<NB>(bd, A.kband, A.lband, lane);
    // canonical spectrum on the (log-spaced) T2 axis: 15 % at 13 % of the axis, 85 % at 39 % (20 ms and 80 ms on 10..2000 ms)
    double b = 0.0;
    for (int j = 0; j < n; ++j) {
        const double u = (double)j / (double)(n - 1), d1 = (u - 0.13) / 0.05, d2 = (u - 0.39) / 0.05;
        const double xc = 0.15 * exp(-0.5 * d1 * d1) + 0.85 * exp(-0.5 * d2 * d2);
        if (lane < m) b = fma(S.Dt[(size_t)j * m + lane], xc, b);
    }
    b = b / bcast(b, 0);
    NnlsState<NB> st; st.itmax_hit = 0;
    MET2_CYC_INIT(st);
    nnls_reset<NB>(st);
    project<NB>(S, b, lane, st.h);
    nnls_solve<NB>(S, bd, st, A.lam[slot], true, lane);
    SeedRec *rec = (SeedRec *)A.out + ((size_t)slot * A.nfa + fa);
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        rec->x[lane + 64 * bb] = st.x[bb];
        rec->pos[lane + 64 * bb] = st.pos[bb];
        rec->ord[lane + 64 * bb] = st.ord[bb];
    }
    if (lane == 0) rec->k = (st.itmax_hit == 0) ? st.k : 0;
}

// Plan-level Cholesky factors for BayesReg's shared Brent abscissae (see BayesTable in objectives.hpp): one wave per (flip angle,
// abscissa) factorises B + lambda_j K in LDS with the routine the fit kernel uses (chol_full, beta = 1) and stores the packed
// triangle and log det U0 = sum log U0_ii.  A non-positive pivot stores NaN as log det: the fit kernel then sees NaN where the
// reference would raise LinAlgError (bayesian_interpolation.py:115) -- same as its own factorisation failing.
struct BayesTabArgs {
    int n, m, nfa, nj, stride;
    const double *Bfa, *Kd, *kband, *lband;
    double lam[MET2_BAYES_TABLE > 0 ? MET2_BAYES_TABLE : 1];
    double *out;
};
template <int NB>
__global__ __launch_bounds__(64) void bayes_table_kernel(BayesTabArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id(), n = A.n, fa = (int)blockIdx.x, j = (int)blockIdx.y;
    WaveShared S;
    S.R = smem; S.n = n; S.m = A.m; S.kmax = n; S.rcap = col_base(n); S.K = A.Kd; S.kband = A.kband;
    S.B = A.Bfa + (size_t)fa * n * n; S.D = nullptr; S.Dt = nullptr; S.DtG = nullptr;
    S.bstride = n; S.dstride = n; S.dtstride = A.m; S.buffer_rows = true; S.reorder = true; S.have_bdiag = false; S.bdiag[0] = S.bdiag[1] = 0.0;
    Band<NB> bd;
    load_band<NB>(bd, A.kband, A.lband, lane);
    double det_u;
    const bool ok = chol_full<NB>(S, bd, 1.0, A.lam[j], lane, det_u);
    __builtin_amdgcn_wave_barrier();
    double *rec = A.out + ((size_t)fa * A.nj + j) * A.stride;
    const int tri = col_base(n);
    for (int i = lane; i < tri; i += 64) rec[i] = smem[i];
    double ls = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) { const int c = lane + 64 * b; if (c < n) ls += log(smem[col_base(c) + c]); }
    ls = wave_sum(ls);
    if (lane == 0) rec[A.stride - 1] = ok ? ls : NAN;
}

#ifndef MET2_SEED
#define MET2_SEED 1            // 0: every voxel grows its first passive set bin by bin from the lambda = 0 solution
#endif
template <int NB>
__device__ __forceinline__ void seed_load(NnlsState<NB> &st, const char *seed, int k, int fa, int lane)
{
    // raw buffer loads: the record's offset travels in a scalar register, the lane part is a 32-bit offset (plain indexing made
    // the compiler keep two 64-bit per-lane record addresses per chunk, and spill them)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)seed, 0, 0x40000000, 0x00020000);
    const unsigned rec = (unsigned)fa * (unsigned)sizeof(SeedRec);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const unsigned e = (unsigned)(lane + 64 * b);
        const auto xv = __builtin_amdgcn_raw_buffer_load_b64(r, 8u * e, rec, 0);
        st.x[b] = __hiloint2double((int)xv[1], (int)xv[0]);
        st.pos[b] = (int)__builtin_amdgcn_raw_buffer_load_b32(r, 4u * e, rec + (unsigned)offsetof(SeedRec, pos), 0);
        st.ord[b] = (int)__builtin_amdgcn_raw_buffer_load_b32(r, 4u * e, rec + (unsigned)offsetof(SeedRec, ord), 0);
        st.P[b] = ballot(st.pos[b] >= 0);
    }
    st.k = k;
}

// The solver calls of the voxel routine.  BIG = false (the kernel's own voxel loop): the inlined solver, as in rounds 1-4.  BIG = true (the
// spill-over voxel routine): one NOT-inlined instance of the solver with the spill-over legs compiled in (nnls_big.hpp: big_solve_fn), the
// state handed over through the stack -- that routine then holds the lambda search and the objectives plus calls, not nine copies of the solver.
template <int NB, int ONE, bool BIG>
__device__ __forceinline__ void solve_warm(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, bool aug, int lane)
{
    if constexpr (BIG) { NnlsState<NB> sc = st; big_solve_fn<NB, ONE>(S, (big_lds_dp)S.R, bd, &sc, lam, (aug ? 1 : 0) | 2); st = sc; }
    else nnls_solve_warm<NB, ONE>(S, bd, st, lam, aug, lane);
}
template <int NB, int ONE, bool BIG>
__device__ __forceinline__ void solve_cold(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, bool aug, int lane)
{
    if constexpr (BIG) { NnlsState<NB> sc = st; big_solve_fn<NB, ONE>(S, (big_lds_dp)S.R, bd, &sc, lam, aug ? 1 : 0); st = sc; }
    else nnls_solve<NB, ONE>(S, bd, st, lam, aug, lane);
}
template <int NB, bool BIG>
__device__ __forceinline__ void refine(const WaveShared &S, NnlsState<NB> &st, double lam, double bvec, int lane)
{
    if constexpr (BIG) { NnlsState<NB> sc = st; big_refine_fn<NB>(S, (big_lds_dp)S.R, &sc, lam, bvec); st = sc; }
    else refine_csne<NB>(S, st, lam, bvec, lane);
}

// One voxel (motor:129-155 + motor:448-468): load, normalise, the method's lambda search, epilogue.  BIG = false: the instance in the kernel's own
// voxel loop; a passive set that wants to outgrow the LDS capacity makes it return true, the voxel's outputs carrying MET2_ST_KOVERFLOW until the
// spill-over kernel (or, without slots, the caller's second launch) has written them again.  BIG = true: the instance inside fit_voxel_spill, which goes on in the wave's
// global slot (nnls_big.hpp).
template <int METHOD, int NB, bool BIG>
__device__ __forceinline__ bool fit_voxel(const FitArgs &A, const WaveShared &S, const Band<NB> &bd, const MetricLanes<NB> &ml, int64_t v, int fa,
                                          int seed_k, bool have_seed, int lane, int wslot, int qslot)
{
    constexpr int ONE = (NB == 2) ? (((METHOD >= 10 ? METHOD - 10 : METHOD) <= MET2_LCURVE) ? 1 : MET2_ONE_REFAC) : 0;
    const int n = A.n, m = A.m;
    // ---- load, normalise by the first echo (motor:129-132), h = D^T b
    double b = (lane < m) ? A.data[v * A.vs + lane * A.es] : 0.0;
    const double km = bcast(b, 0);
    b = b / km;
    NnlsState<NB> st; st.itmax_hit = 0;
    MET2_CYC_INIT(st);
    MET2_CYC_BEGIN(c_vox);
    nnls_reset<NB>(st);
    project<NB>(S, b, lane, st.h);
    double regv = 0.0, lamv = 0.0; int stat = MET2_ST_FITTED;
    bool queued = false;                  // (the L-curve queues its voxel itself)

    if (METHOD == MET2_NNLS) {
        solve_cold<NB, ONE, BIG>(S, bd, st, 0.0, false, lane);
    } else if (METHOD == MET2_T2SPARC) {
        if (have_seed) { seed_load<NB>(st, A.seed, seed_k, fa, lane); solve_warm<NB, ONE, BIG>(S, bd, st, A.t2sparc_lambda, true, lane); }
        else solve_cold<NB, ONE, BIG>(S, bd, st, A.t2sparc_lambda, true, lane);
        regv = lamv = A.t2sparc_lambda;
    } else if (METHOD == MET2_X2) {
        // algorithms.py:211-233
        solve_cold<NB, ONE, BIG>(S, bd, st, 0.0, false, lane);
        constexpr bool PIN = uni_brent(METHOD, NB);
        const double SSE = uni<PIN>(sse_of<NB>(S, st, b, lane));
        const double target = uni<PIN>(A.x2_factor * SSE);
        int flag;
        double last_x = -1.0, last_sse = 0.0;
        if (have_seed) seed_load<NB>(st, A.seed, seed_k, fa, lane);     // start of the first Brent point
        // algorithms.py:220 solves once more at reg_opt.  The solution of the evaluation that made reg_opt Brent's best point is
        // that solution up to rounding (the solve is deterministic and the minimiser unique): its spectrum, passive set and
        // SSE are kept as they come by and restored at the end (85 % of the voxels of configs[1] would repeat the solve)
        double best_x[NB], best_sse = 0.0; int best_pos[NB], best_ord[NB];
#ifdef MET2_CYCSTATS
        int evi = 0; double canon = 0.0;
#endif
#if MET2_TIE_GUARD
        int nref = 0;
        double lam = fminbound_tie_dev<uni_brent(METHOD, NB)>([&](double x, bool refined) {
            if (NB == 2 && (st.itmax_hit & 2)) return 0.0;
            if (!(refined && x == last_x)) solve_warm<NB, ONE, BIG>(S, bd, st, x, true, lane);
            if (refined) refine<NB, BIG>(S, st, x, b, lane);
            const double SSEr = uni<PIN>(sse_of<NB>(S, st, b, lane));
            last_x = x; last_sse = SSEr;
            return fabs(SSEr - target) / SSE;
        }, [&]() {
            best_sse = last_sse;
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) { best_x[bb] = st.x[bb]; best_pos[bb] = st.pos[bb]; best_ord[bb] = st.ord[bb]; }
        }, BIG ? A.lam_lo : 0.0, BIG ? A.lam_hi : 10.0, A.xtol, A.maxfun, flag, nref);
        if (MET2_TIE_GUARD == 2 && nref) stat |= 64 | (nref & 0x1f00);
#else
        double lam = fminbound_dev<uni_brent(METHOD, NB)>([&](double x) {
#ifdef MET2_CYCSTATS
            unsigned long long snap[8];
            for (int q_ = 0; q_ < 8; ++q_) snap[q_] = st.cyc[q_];
            const int k0_ = st.k;
            const double gm_ = 0.5 * (3.0 - sqrt(5.0));
            canon = (evi == 0) ? gm_ * 10.0 : (evi == 1 ? canon + gm_ * (10.0 - canon) : (evi == 2 ? gm_ * 10.0 * (1.0 - gm_) : canon * (1.0 - gm_)));
            const bool is_canon = fabs(x - canon) <= 1e-12 * canon;
#endif
            if (NB == 2 && (st.itmax_hit & 2)) return 0.0;  // the passive set hit the pass's capacity: the voxel is solved again in the
                                                            // next pass, the rest of its Brent path here costs nothing (two bins per lane,
                                                            // where 5-10 % of the voxels do; at one bin per lane ~1 % do and the test cost
                                                            // the X2 kernel two more spilled registers)
            solve_warm<NB, ONE, BIG>(S, bd, st, x, true, lane);
            double SSEr = uni<PIN>(sse_of<NB>(S, st, b, lane));
#ifdef MET2_CYCSTATS
            if (lane == 0) {
                const int e_ = evi < 39 ? evi : 39;
                atomicAdd(&g_ev[e_][0], 1ull);
                for (int q_ = 1; q_ < 8; ++q_) atomicAdd(&g_ev[e_][q_], (st.cyc[q_] - snap[q_]) % 1000000000000ull);
                atomicAdd(&g_ev[e_][8], (unsigned long long)k0_); atomicAdd(&g_ev[e_][9], (unsigned long long)st.k);
                if (is_canon) atomicAdd(&g_ev[e_][10], 1ull);
            }
            ++evi;
#endif
            last_x = x; last_sse = SSEr;
            return fabs(SSEr - target) / SSE;
        }, [&]() {
            best_sse = last_sse;
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) { best_x[bb] = st.x[bb]; best_pos[bb] = st.pos[bb]; best_ord[bb] = st.ord[bb]; }
        }, BIG ? A.lam_lo : 0.0, BIG ? A.lam_hi : 10.0, A.xtol, A.maxfun, flag);
#endif
        if (flag == 1) stat |= MET2_ST_BRENT_MAXFUN;
        if (lam != last_x && !(NB == 2 && (st.itmax_hit & 2))) {
            int kk = 0;
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                st.x[bb] = best_x[bb]; st.pos[bb] = best_pos[bb]; st.ord[bb] = best_ord[bb];
                st.P[bb] = ballot(st.pos[bb] >= 0); kk += __popcll(st.P[bb]);
            }
            st.k = kk; last_sse = best_sse;
        }
        regv = last_sse / SSE;                            // k_est (motor:141-143)
        lamv = lam;
    } else if (METHOD == MET2_LCURVE) {
        // algorithms.py:88-113
        // The solve at the corner (algorithms.py:111) starts from the sweep's state nearest to it: besides the last grid point
        // the states of four evenly spaced ones are kept (iterate, and position | pivot bin packed in one word): from the last
        // point alone the passive set had to shrink by up to ~25 bins, one plane-rotation chain each.
        constexpr int NS = 4;
        double le = 0.0, ln = 0.0, keep_x[NS][NB];
        int keep_p[NS][NB];
        // A sweep whose set outgrows the LDS capacity (two bins per lane: ~5 % of the voxels, nearly all of them in the last grid points towards
        // lambda = 0) is not solved again from its first grid point by the spill-over kernel: the voxel is queued HERE, the sweep's state (the log
        // norms so far, the kept states, the iterate in hand) goes to the queue entry's record, and the spill-over instance goes on from the grid
        // point that overflowed (LC_SAVE_DOUBLES per lane; entries beyond lc_cap start over).
        int at = -1, i0 = 0;
        bool from_saved = false;
        if (BIG && NB == 2 && A.lc_save && qslot >= 0 && qslot < A.lc_cap) {
            const double *rec = A.lc_save + (size_t)qslot * (LC_SAVE_DOUBLES * 64) + lane;
            const int *reci = (const int *)(A.lc_save + (size_t)qslot * (LC_SAVE_DOUBLES * 64) + (2 + 5 * NB) * 64) + lane;
            i0 = A.lc_at[qslot];
            le = rec[0]; ln = rec[64];
            int kk = 0;
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
                for (int q = 0; q < NS; ++q) { keep_x[q][bb] = rec[(2 + q * NB + bb) * 64]; keep_p[q][bb] = reci[(q * NB + bb) * 64]; }
                st.x[bb] = rec[(2 + NS * NB + bb) * 64];
                st.pos[bb] = reci[(NS * NB + bb) * 64]; st.ord[bb] = reci[(NS * NB + NB + bb) * 64];
                st.P[bb] = ballot(st.pos[bb] >= 0); kk += __popcll(st.P[bb]);
            }
            st.k = kk;
            from_saved = i0 >= A.nlam;
        }
        for (int i = i0; i < A.nlam; ++i) {
            double lam = A.lam_grid[i];
            solve_warm<NB, ONE, BIG>(S, bd, st, lam, true, lane);
            if (!BIG && NB == 2 && (st.itmax_hit & 2)) { at = i; break; }      // capacity hit: the spill-over kernel goes on
            double sse = sse_of<NB>(S, st, b, lane);
            double sn = seminorm2<NB>(bd, st.x, n, lane);
            if (lane == i) { le = log(sse + 1e-200); ln = log(sn + 1e-200); }
#pragma unroll
            for (int q = 0; q < NS; ++q)
                if (i == (q + 1) * A.nlam / (NS + 1) - 1) {
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) { keep_x[q][bb] = st.x[bb]; keep_p[q][bb] = (st.pos[bb] + 1) | (st.ord[bb] << 9); }
                }
        }
        if (at < 0) {
        int corner = select_corner_dev(le, ln, A.nlam, lane);
        regv = lamv = A.lam_grid[corner];
        if (!from_saved) {
            int best = -1, dist = A.nlam - 1 - corner;               // the state in hand belongs to the last grid point
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int iq = (q + 1) * A.nlam / (NS + 1) - 1;
                const int dq = abs(iq - corner);
                if (iq >= 0 && dq < dist) { dist = dq; best = q; }
            }
#pragma unroll
            for (int q = 0; q < NS; ++q)
                if (best == q) {
                    int kk = 0;
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) {
                        st.x[bb] = keep_x[q][bb]; st.pos[bb] = (keep_p[q][bb] & 0x1ff) - 1; st.ord[bb] = keep_p[q][bb] >> 9;
                        st.P[bb] = ballot(st.pos[bb] >= 0); kk += __popcll(st.P[bb]);
                    }
                    st.k = kk;
                }
        }
        solve_warm<NB, ONE, BIG>(S, bd, st, regv, true, lane);
        if (!BIG && NB == 2 && (st.itmax_hit & 2)) at = A.nlam;
        }
        if (!BIG && NB == 2 && at >= 0 && A.lc_save) {
            int t = 0;
            if (lane == 0) { t = atomicAdd(A.sb.err + 1, 1); A.sb.ovf[t] = (int)v; }
            t = __builtin_amdgcn_readfirstlane(t);
            queued = true;
            if (t < A.lc_cap) {
                double *rec = A.lc_save + (size_t)t * (LC_SAVE_DOUBLES * 64) + lane;
                int *reci = (int *)(A.lc_save + (size_t)t * (LC_SAVE_DOUBLES * 64) + (2 + 5 * NB) * 64) + lane;
                if (lane == 0) A.lc_at[t] = at;
                rec[0] = le; rec[64] = ln;
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) {
#pragma unroll
                    for (int q = 0; q < NS; ++q) { rec[(2 + q * NB + bb) * 64] = keep_x[q][bb]; reci[(q * NB + bb) * 64] = keep_p[q][bb]; }
                    rec[(2 + NS * NB + bb) * 64] = st.x[bb];
                    reci[(NS * NB + bb) * 64] = st.pos[bb]; reci[(NS * NB + NB + bb) * 64] = st.ord[bb];
                }
            }
        }
    } else if (METHOD == MET2_BAYESREG) {
        // bayesian_interpolation.py:84-105
        solve_cold<NB, 0, BIG>(S, bd, st, 0.0, false, lane);
        int nnz = 0;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) nnz += __popcll(ballot((lane + 64 * bb < n) && (st.x[bb] > 0.0)));
        double dof = (double)(m - nnz); dof = dof < 1.0 ? 1.0 : dof;
        const double sigma = sqrt(sse_of<NB>(S, st, b, lane) / dof);
        BayesCtx bc; bc.beta = 1.0 / (sigma * sigma); bc.log_detL = A.log_detL; bc.failed = 0;
        int flag, ev = 0;
        if (have_seed) seed_load<NB>(st, A.seed, seed_k, fa, lane);
        double *cholG = (NB == 2 && A.chol) ? A.chol + (size_t)wslot * (size_t)A.chol_stride : nullptr;
        double lam = fminbound_dev<uni_brent(METHOD, NB)>([&](double x) {
            if (NB == 2 && (st.itmax_hit & 2)) return 0.0;  // capacity hit: solved again in the next pass
            solve_warm<NB, ONE, BIG>(S, bd, st, x, true, lane);
            BayesTable tab{nullptr, 0.0};
            if (A.btab && ev < A.nbtab) {                  // still on the abscissae every voxel shares?  (a rounding-level match: the table's
                const double tl = A.blam[ev];              //  lambda_j is formed on the host, and B + lambda K does not care about an ulp)
                if (fabs(x - tl) <= 1e-14 * tl) {
                    const double *rec = A.btab + ((size_t)fa * A.nbtab + ev) * A.btab_stride;
                    const double ld0 = rec[A.btab_stride - 1];
                    if (fabs(ld0) <= 1.79769313486231570815e308) { tab.U0 = rec; tab.logdet0 = ld0; }      // (nan: the table's factorisation failed --
                                                                                                            //  the voxel factorises itself and reports its own failure)
                }
            }
            ++ev;
            return bayes_objective<NB>(S, bd, st, bc, x, b, lane, tab, cholG);
        }, []() {}, BIG ? A.lam_lo : 1e-8, BIG ? A.lam_hi : 2.0, A.xtol, A.maxfun, flag);
        if (flag == 1) stat |= MET2_ST_BRENT_MAXFUN;
        if (bc.failed) stat |= MET2_ST_CHOLFAIL;
        if (!(NB == 2 && (st.itmax_hit & 2))) solve_warm<NB, ONE, BIG>(S, bd, st, lam, true, lane);
        regv = lamv = lam;
    } else if (METHOD == MET2_GCV || METHOD == MET2_GCV_LR) {
        // algorithms.py:276-283
        int flag, overflow = 0;
        GcvCache<NB> gc; gc.valid = 0; gc.next = 0;
        if (have_seed) seed_load<NB>(st, A.seed, seed_k, fa, lane);
        double lam = fminbound_dev<uni_brent(METHOD, NB)>([&](double x) {
            if (NB == 2 && (st.itmax_hit & 2)) return 0.0;  // capacity hit: solved again in the next pass
            solve_warm<NB, ONE, BIG>(S, bd, st, x, true, lane);
            return gcv_objective<NB, METHOD == MET2_GCV_LR>(S, bd, st, x, b, lane, overflow, gc);
        }, []() {}, BIG ? A.lam_lo : 1e-8, BIG ? A.lam_hi : 10.0, A.xtol, A.maxfun, flag);
        if (flag == 1) stat |= MET2_ST_BRENT_MAXFUN;
        if (overflow) stat |= MET2_ST_KOVERFLOW;
        if (!(NB == 2 && (st.itmax_hit & 2))) solve_warm<NB, ONE, BIG>(S, bd, st, lam, true, lane);
        regv = lamv = lam;
    }
    if (METHOD >= 10) {
        // objective values of method METHOD-10 on the plan's lambda grid -> fsol[v][0..nlam)
        constexpr int BASE = METHOD - 10;
        double SSE = 1.0; BayesCtx bc; bc.failed = 0; bc.log_detL = A.log_detL; bc.beta = 1.0;
        if (BASE == MET2_X2 || BASE == MET2_BAYESREG) {
            nnls_solve<NB>(S, bd, st, 0.0, false, lane);
            SSE = sse_of<NB>(S, st, b, lane);
            int nnz = 0;
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) nnz += __popcll(ballot((lane + 64 * bb < n) && (st.x[bb] > 0.0)));
            double dof = (double)(m - nnz); dof = dof < 1.0 ? 1.0 : dof;
            const double sigma = sqrt(SSE / dof);
            bc.beta = 1.0 / (sigma * sigma);
        }
        double keep = 0.0; int overflow = 0;
        GcvCache<NB> gc; gc.valid = 0; gc.next = 0;
        for (int i = 0; i < A.nlam; ++i) {
            const double x = A.lam_grid[i];
            nnls_solve<NB>(S, bd, st, x, true, lane);
            double val;
            if (BASE == MET2_X2) val = fabs(sse_of<NB>(S, st, b, lane) - A.x2_factor * SSE) / SSE;
            else if (BASE == MET2_GCV || BASE == MET2_GCV_LR) val = gcv_objective<NB, BASE == MET2_GCV_LR>(S, bd, st, x, b, lane, overflow, gc);
            else val = bayes_objective<NB>(S, bd, st, bc, x, b, lane);
            if (lane == i) keep = val;
        }
        if (lane < n) A.fsol[(size_t)v * n + lane] = keep;      // nlam <= 64 values, zero-padded to n
        if (NB == 2 && lane + 64 < n) A.fsol[(size_t)v * n + lane + 64] = 0.0;
        if (lane == 0) { A.reg[v] = 0.0; if (A.status) A.status[v] = stat | (overflow ? MET2_ST_KOVERFLOW : 0); }
        return false;
    }
    if (st.itmax_hit & 1) stat |= MET2_ST_ITMAX;
    if (st.itmax_hit & 2) stat |= MET2_ST_KOVERFLOW;
    MET2_CYC_END(0, c_vox);
    MET2_CYC_FLUSH(st);

    // ---- epilogue: un-normalise (motor:153-155) + metrics (motor:448-468)
    double xs[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        xs[bb] = st.x[bb] * km;
        if (lane + 64 * bb < n) A.fsol[(size_t)v * n + lane + 64 * bb] = xs[bb]; else xs[bb] = 0.0;
    }
    if (A.sig) {
        double sg = model_signal<NB>(S, st, lane) * km;
        if (lane < m) A.sig[(size_t)v * m + lane] = sg;
    }
    if (A.maps) write_metrics<NB>(ml, xs, true, A.maps, A.nvox, v, lane);
    if (lane == 0) {
        A.reg[v] = regv;
        if (A.lam) A.lam[v] = lamv;
        if (A.status) A.status[v] = stat;
    }
    return !BIG && !queued && (st.itmax_hit & 2) != 0;                     // the set outgrew the wave's LDS region (the outputs just written carry MET2_ST_KOVERFLOW): the caller queues the voxel for the spill-over kernel
}

// the wave's view of the plan: everything of WaveShared that does not depend on the flip angle
__device__ __forceinline__ void fit_shared(const FitArgs &A, WaveShared &S, double *sR, int wslot)
{
    S.R = sR; S.n = A.n; S.m = A.m; S.kmax = A.kmax; S.rcap = A.wave_doubles; S.K = A.Kd; S.kband = A.kband; S.Dt = nullptr; S.DtG = A.Dtfa; S.dtstride = A.m; S.buffer_rows = true; S.reorder = true; S.have_bdiag = false; S.bdiag[0] = S.bdiag[1] = 0.0;
    S.B = A.Bfa; S.D = A.Dfa; S.bstride = A.n; S.dstride = A.n;
    S.Rg = A.big ? A.big + (size_t)wslot * (size_t)A.big_stride : nullptr; S.gbase = col_base(A.kmax);
}
template <int METHOD>
__device__ __forceinline__ void fit_shared_fa(const FitArgs &A, WaveShared &S, int fa)
{
    const int n = A.n, m = A.m;
    S.B = A.Bfa + (size_t)fa * n * n; S.D = A.Dfa + (size_t)fa * m * n; S.Dt = A.Dtfa + (size_t)fa * m * n;
    S.DtG = ((METHOD >= 10 ? METHOD - 10 : METHOD) == MET2_GCV_LR) ? A.Aq + (size_t)fa * n * MET2_GCV_LR_RANK : S.Dt;
}

// One queued spill-over voxel, solved from its first echo with the spill-over routines compiled in (fit_voxel<BIG = true>): the evaluations that
// fit the LDS capacity run the fast legs as before, the others keep the factor's columns beyond it in the wave's global slot.  NOT inlined into
// the spill-over kernel's queue loop, and everything it needs is rebuilt inside from the arguments: with the voxel routine inlined in that
// loop, what the loop keeps across the solver calls (the plan's scalars, the penalty bands, the metric windows) did not survive a wave's first
// voxel -- every wave faulted on its second one (round 5: 3 020 queued voxels on 2 048 waves; 1 549 ran clean).  The arguments arrive in
// vector registers and the plan's arguments through a private copy: everything wave-uniform is made scalar again (readfirstlane).
template <int METHOD, int NB>
__device__ __attribute__((noinline)) void fit_voxel_spill(const FitArgs *Ap, big_lds_dp sRl, int64_t v, int fa, int wslot, int qslot)
{
    const int lane = big_lane();
    double *sR = (double *)sRl;
    FitArgs A;
    {
        static_assert(sizeof(FitArgs) % 4 == 0, "FitArgs is copied word by word");
        const int *src = (const int *)Ap;
        int *dst = (int *)&A;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(FitArgs) / 4); ++i) dst[i] = big_rfl(src[i]);
    }
    v = (int64_t)big_rfl((u64)v); fa = big_rfl(fa); wslot = big_rfl(wslot); qslot = big_rfl(qslot);
    WaveShared S;
    fit_shared(A, S, sR, wslot);
    fit_shared_fa<METHOD>(A, S, fa);
    Band<NB> bd;
    load_band<NB>(bd, A.kband, A.lband, lane);
    MetricLanes<NB> ml;
    metric_lanes<NB>(ml, A.t2s, A.n, A.cut_m, A.cut_ie, lane);
    const int seed_k = (MET2_SEED && A.seed) ? ((const SeedRec *)A.seed)[fa].k : 0;
    const bool have_seed = seed_k > 0 && seed_k <= A.kmax;
    (void)fit_voxel<METHOD, NB, true>(A, S, bd, ml, v, fa, seed_k, have_seed, lane, wslot, qslot);
}

// SECOND = false: the fit kernel proper.  Every wave pulls voxels from the FA-sorted queue; a voxel whose passive set outgrows the wave's LDS region
// (capacity A.kmax: as many resident waves as the registers allow) is put on the spill-over queue, nothing written.
// SECOND = true: the spill-over kernel, launched behind it with the same geometry: the same voxel routine with BIG = true -- the solver with the
// spill-over legs compiled in (nnls_big.hpp), one not-inlined instance -- on the queued voxels, at the first kernel's occupancy (8 waves per CU at
// nT2 = 120 where the full-size factor of rounds 1-4's clean-up pass allowed 2) and without the re-sort (five small kernels) that pass needed.
// Why a second kernel and not a call behind (or inside) the first one's voxel loop: a kernel that contains a call keeps registers for the calling
// convention throughout -- fit_kernel<X2, 1> went from 29 spill stores / 47 reloads to 52 / 160 and configs[1] from 150.4 to 157.5 ms with the
// call BEHIND the loop (inside it: 7 000 reloads); and with the spill-over legs inlined into the nine copies of the solver the X2 kernel holds,
// one translation unit compiles for six minutes.  The voxels wait for the first kernel's last wave either way.
template <int METHOD, int NB, bool SECOND>
__global__ __launch_bounds__(64 * method_max_waves(METHOD, NB)) void fit_kernel(FitArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int n = A.n, m = A.m, kmax = A.kmax;
    const int tri = A.wave_doubles;
    double *sR = smem + (size_t)wave * tri;             // every wave's region starts 16-byte aligned

    const int wslot = (int)(blockIdx.x * (unsigned)A.waves + (unsigned)wave);      // this wave's slot in the per-wave scratch arrays
    WaveShared S;
    fit_shared(A, S, sR, wslot);
    Band<NB> bd;
    load_band<NB>(bd, A.kband, A.lband, lane);
    MetricLanes<NB> ml;
    metric_lanes<NB>(ml, A.t2s, n, A.cut_m, A.cut_ie, lane);

    if constexpr (!SECOND) {
    const int nchunks = A.sb.chunk_start[A.nfa];
    for (int round = 0; round <= nchunks; ++round) {      // the queue hands out each chunk once
        // every WAVE pulls its own (small) chunk from the global queue -- no workgroup barrier anywhere, so a wave never
        // idles while its neighbours finish their voxels.
        // eight cursors, one per XCD, each over a contiguous eighth of the FA-sorted list: neighbouring voxels are
        // solved on the same XCD, so their 8-byte outputs (maps, reg, lambda) merge into whole lines in that XCD's L2
        // before they go to HBM; a wave whose own eighth is used up takes from the next ones
        int c = nchunks;
        if (lane == 0) {
            const int xcd = (int)(blockIdx.x & 7u);
            for (int t = 0; t < 8; ++t) {
                const int q = (xcd + t) & 7;
                const int qlo = (int)(((int64_t)nchunks * q) >> 3), qhi = (int)(((int64_t)nchunks * (q + 1)) >> 3);
                if (qlo >= qhi) continue;
                const int i = atomicAdd(A.sb.xq + q, 1);
                if (i < qhi - qlo) { c = qlo + i; break; }
            }
        }
        c = __builtin_amdgcn_readfirstlane(c);
        MET2_STAT(5, round);
        if (c >= nchunks) break;
        int lo = 0, hi = A.nfa;                       // largest fa with chunk_start[fa] <= c
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (A.sb.chunk_start[mid] <= c) lo = mid; else hi = mid; }
        const int fa = lo;
        const int first = A.sb.bucket_start[fa] + (c - A.sb.chunk_start[fa]) * A.chunk;
        const int cnt = min(A.chunk, A.sb.bucket_start[fa + 1] - first);
        fit_shared_fa<METHOD>(A, S, fa);
        const int seed_k = (MET2_SEED && A.seed) ? ((const SeedRec *)A.seed)[fa].k : 0;
        const bool have_seed = seed_k > 0 && seed_k <= kmax;
        for (int slot = 0; slot < cnt; ++slot) {
            MET2_STAT(4, slot);
            const int64_t v = A.sb.perm[first + slot];

            if (fit_voxel<METHOD, NB, false>(A, S, bd, ml, v, fa, seed_k, have_seed, lane, wslot, -1) && A.big) {
                // the set outgrew the wave's LDS region: the voxel goes to the spill-over queue
                if (lane == 0) A.sb.ovf[atomicAdd(A.sb.err + 1, 1)] = (int)v;
            }
        }
    }
    } else {
    // ---- the spill-over queue (written by the launch before this one): every wave takes entries until none is left
    if constexpr (METHOD < 10) {
    const int ntail = A.all_queued ? A.sb.bucket_start[A.nfa] : A.sb.err[1];      // (all_queued: every fitted voxel, in the sorted list's order)
    const int *queue = A.all_queued ? A.sb.perm : A.sb.ovf;
    FitArgs Ac = A;                                       // (a private copy for the not-inlined voxel routine)
    // The workgroup's LDS is carved for as few waves as give every queued voxel a wave of its own (this kernel's duration is the latency of its
    // slowest voxel when the queue is short, its throughput when it is long): w2 waves, each with the largest factor capacity its share holds --
    // at nT2 = 120: 8 waves at capacity 71 (the first kernel's), 7 / 75, 6 / 81, 5 / 89, 4 / 100, 3 / 116, 2 and 1 at 120 (no spill-over leg runs);
    // at nT2 = 60: 16 / 50 ... 11 and fewer at 60.  The other waves leave at once.
    {
        const int total = A.waves * A.wave_doubles, wgs = (int)gridDim.x;
        int w2 = A.waves;
        while (w2 > 1 && (int64_t)(w2 - 1) * wgs >= ntail) --w2;
        if (A.spill_w2 > 0 && A.spill_w2 <= A.waves) w2 = A.spill_w2;
        const int per = (total / w2) & ~1;
        int k2 = A.kmax;
        while (k2 < n && col_base(k2 + 1) <= per) ++k2;
        if (wave >= w2) return;
        Ac.kmax = k2; Ac.wave_doubles = per;
        sR = smem + (size_t)wave * per;
    }
    for (;;) {
        int i = ntail;
        if (lane == 0) i = atomicAdd(A.sb.err + 2, 1);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= ntail) break;
        const int64_t v = queue[i];
        fit_voxel_spill<METHOD, NB>(&Ac, (big_lds_dp)sR, v, A.sb.key[v], wslot, A.all_queued ? -1 : i);
    }
    }
    }
}

struct LaunchGeom { int grid, block, waves, kmax, lds, wave_doubles, nb; };

template <int METHOD, int NB, bool SECOND>
int launch_fit_nb(const FitArgs &A, const LaunchGeom &g, hipStream_t s)
{
    if (g.block > 64 * method_max_waves(METHOD, NB)) return fail(MET2_E_INVALID, "launch geometry exceeds the kernel's launch bounds");
    HIPCHK(hipFuncSetAttribute((const void *)fit_kernel<METHOD, NB, SECOND>, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds));
    hipLaunchKernelGGL((fit_kernel<METHOD, NB, SECOND>), dim3(g.grid), dim3(g.block), g.lds, s, A);
    HIPCHK(hipGetLastError());
    return MET2_OK;
}
