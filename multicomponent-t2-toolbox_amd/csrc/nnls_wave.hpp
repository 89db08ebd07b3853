// nnls_wave.hpp -- one Tikhonov-regularised NNLS per wavefront, Gram form.
//
//   min_x || [D; sqrt(lam) L] x - [b; 0] ||,  x >= 0
//
// solved by Lawson-Hanson's active-set iteration (the algorithm behind the reference's
// nnls(), intravoxel_algorithms/algorithms.py:55-82) restated on the normal equations
//   G = B + lam K,  B = D^T D,  K = L^T L,  h = D^T b :
//   dual  w = h - G x,  entering variable = argmax_Z w  (same rule as Lawson-Hanson),
//   passive sub-problem  G_PP z = h_P  through an incrementally updated Cholesky factor
//   R^T R = G_PP kept in LDS (append = one forward substitution, removal = Givens
//   re-triangularisation), ratio test and removal rule as in Lawson-Hanson.
//
// Mapping: lane j <-> T2 bin j (n <= 64).  "Position" p is the order in which bins
// joined the passive set; lane p also holds the position-indexed quantities
// (ord[p] = bin at position p, y[p] = (R^-T h_P)[p], 1/R[p][p]).
//
// LDS per workgroup (doubles): sB[n][np]  Gram matrix of the workgroup's flip angle
//                              sD[m][np]  dictionary rows (np odd -> conflict-free columns)
//              per wave:       R packed upper-triangular rows, kmax(kmax+1)/2
#pragma once
#include "wave_ops.hpp"

namespace met2 {

#ifdef MET2_LOOPSTATS
__device__ int g_loopstats[8];
#define MET2_STAT(slot, v) atomicMax(&g_loopstats[slot], (int)(v))
#else
#define MET2_STAT(slot, v)
#endif

struct WaveShared {
    const double *sB;   // [n][np]
    const double *sD;   // [m][np]
    double *R;          // this wave's packed factor
    int n, m, np, kmax;
    int rcap;           // doubles available at R
};

// K = L^T L and L itself as 5 diagonals per lane: kb[d] = K[j][j+d-2], lb[d] = L[j][j+d-2]
struct Band {
    double kb[5];
    double lb[5];
};

struct NnlsState {
    double h;      // (D^T b)_j
    double x;      // current iterate, bin-indexed
    double y;      // position-indexed  R^-T h_P
    double rinv;   // position-indexed  1 / R[p][p]
    int ord;       // position-indexed  bin at position p
    int pos;       // bin-indexed       position of bin j, -1 if in Z
    int k;         // |P|                (uniform)
    u64 P;         // passive-set mask   (uniform)
    int itmax_hit; // uniform flag
};

__device__ __forceinline__ int row_base(int i, int kmax) { return i * kmax - (i * (i - 1)) / 2 - i; } // entry (i,c) at row_base + c

__device__ __forceinline__ double band_pick(const double (&b)[5], int d) // d in [-2,2] else 0
{
    double v = 0.0;
    v = (d == -2) ? b[0] : v;
    v = (d == -1) ? b[1] : v;
    v = (d == 0) ? b[2] : v;
    v = (d == 1) ? b[3] : v;
    v = (d == 2) ? b[4] : v;
    return v;
}

// (K v)_j for a bin-indexed vector v (zero outside [0,n))
__device__ __forceinline__ double band_mul(const double (&b)[5], double v, int lane)
{
    double acc = b[2] * v;
    double t;
    t = gather(v, (lane + 62) & 63); acc += b[0] * t;   // v[j-2]  (b[0] is zero where j-2 < 0)
    t = gather(v, (lane + 63) & 63); acc += b[1] * t;   // v[j-1]
    t = gather(v, (lane + 1) & 63);  acc += b[3] * t;   // v[j+1]  (b[3] zero where j+1 >= n)
    t = gather(v, (lane + 2) & 63);  acc += b[4] * t;   // v[j+2]
    return acc;
}

// Lawson-Hanson plane rotation (g1)
__device__ __forceinline__ void givens(double a, double b, double &c, double &s, double &sig)
{
    if (fabs(a) > fabs(b)) {
        double xr = b / a, yr = sqrt(1.0 + xr * xr);
        c = copysign(1.0 / yr, a); s = c * xr; sig = fabs(a) * yr;
    } else if (b != 0.0) {
        double xr = a / b, yr = sqrt(1.0 + xr * xr);
        s = copysign(1.0 / yr, b); c = s * xr; sig = fabs(b) * yr;
    } else { sig = 0.0; c = 0.0; s = 1.0; }
}

// back substitution R z = y ; returns z position-indexed
__device__ __forceinline__ double back_subst(const WaveShared &S, const NnlsState &st, int lane)
{
    const int k = st.k;
    const int rbl = row_base(lane, S.kmax);
    double y = st.y;
    double rv = (k > 0 && lane < k - 1) ? S.R[rbl + (k - 1)] : 0.0;
    for (int c = k - 1; c >= 0; --c) {
        double rvn = (c > 0 && lane < c - 1) ? S.R[rbl + (c - 1)] : 0.0;   // prefetch next column
        double s = bcast(y * st.rinv, c);
        y = fma(-rv, s, y);                                                 // lanes >= c are final (rv = 0 there)
        rv = rvn;
    }
    return (lane < k) ? y * st.rinv : 0.0;
}

// Remove position p from the passive set: delete column p of R and re-triangularise.
__device__ __forceinline__ void remove_pos(const WaveShared &S, NnlsState &st, int p, int lane)
{
    const int k = st.k, kmax = S.kmax;
    const int tb = bcast_i(st.ord, p);
    if (p < k - 1) {
        // rows above p: shift the entries right of column p one place left
        const bool mv = (lane >= p) && (lane <= k - 2);
        for (int i = 0; i < p; ++i) {
            const int rb = row_base(i, kmax);
            double v = mv ? S.R[rb + lane + 1] : 0.0;
            if (mv) S.R[rb + lane] = v;
        }
        // rows p..k-1: chain of plane rotations, lane = old column index
        const int rbp = row_base(p, kmax);
        double carry = (lane > p && lane < k) ? S.R[rbp + lane] : 0.0;
        double ycar = bcast(st.y, p);
        for (int j = p + 1; j < k; ++j) {
            const int rbj = row_base(j, kmax);
            double rowj = (lane >= j && lane < k) ? S.R[rbj + lane] : 0.0;
            double a = bcast(carry, j), b = bcast(rowj, j);
            double c, s, sig;
            givens(a, b, c, s, sig);
            double yj = bcast(st.y, j);
            double ynew = c * ycar + s * yj;
            ycar = -s * ycar + c * yj;
            double nv = c * carry + s * rowj;
            carry = -s * carry + c * rowj;
            const int rbn = row_base(j - 1, kmax);
            if (lane > j && lane < k) S.R[rbn + lane - 1] = nv;
            if (lane == j) S.R[rbn + j - 1] = sig;
            if (lane == j - 1) { st.y = ynew; st.rinv = 1.0 / sig; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    int ordn = gather_i(st.ord, (lane + 1) & 63);
    st.ord = (lane >= p) ? ordn : st.ord;
    st.pos = (st.pos > p) ? st.pos - 1 : st.pos;
    if (lane == tb) { st.pos = -1; st.x = 0.0; }
    st.P &= ~(1ull << tb);
    st.k = k - 1;
}

// Try to move bin t from Z to P.  Returns false (state untouched) when the column is
// numerically dependent on the passive columns or its trial coefficient is not
// positive (Lawson-Hanson's two acceptance tests).
__device__ __forceinline__ bool try_append(const WaveShared &S, const Band &bd, NnlsState &st, double lam, int t, int lane,
                                           bool forced = false)
{
    const int k = st.k, kmax = S.kmax;
    double gb = (lane < S.n) ? S.sB[t * S.np + lane] : 0.0;
    gb = fma(lam, band_pick(bd.kb, t - lane), gb);               // G[lane][t]
    const double gtt = bcast(gb, t);
    double g = gather(gb, st.ord);                               // position-indexed G[ord_p][t]
    g = (lane < k) ? g : 0.0;
    // forward substitution R^T r = g
    {
        double rv = (k > 0 && lane > 0 && lane < k) ? S.R[row_base(0, kmax) + lane] : 0.0;
        for (int i = 0; i < k; ++i) {
            double rvn = (i + 1 < k && lane > i + 1 && lane < k) ? S.R[row_base(i + 1, kmax) + lane] : 0.0;
            double s = bcast(g * st.rinv, i);
            g = fma(-rv, s, g);                                  // lanes <= i are final (rv = 0 there)
            rv = rvn;
        }
    }
    const double r = (lane < k) ? g * st.rinv : 0.0;
    double rr = r * r, ry = r * st.y;       // both zero for lanes >= k
    wave_sum2(rr, ry);
    const double rho2 = gtt - rr;
    if (!(rho2 > 1e-14 * gtt)) return false;                     // dependent column (noise floor of gtt - r.r)
    const double rho = sqrt(rho2);
    const double ynew = (bcast(st.h, t) - ry) / rho;
    if (!forced && !(ynew / rho > 0.0)) return false;            // ztest
    if (lane < k) S.R[row_base(lane, kmax) + k] = r;
    if (lane == k) { S.R[row_base(k, kmax) + k] = rho; st.rinv = 1.0 / rho; st.y = ynew; st.ord = t; }
    __builtin_amdgcn_wave_barrier();
    if (lane == t) st.pos = k;
    st.P |= (1ull << t);
    st.k = k + 1;
    return true;
}

// dual vector w = h - (B + lam K) x, bin-indexed
__device__ __forceinline__ double dual(const WaveShared &S, const Band &bd, const NnlsState &st, double lam, int lane)
{
    double acc = 0.0;
    const int k = st.k;
    for (int p = 0; p < k; ++p) {
        int i = bcast_i(st.ord, p);
        double xi = bcast(st.x, i);
        double bv = (lane < S.n) ? S.sB[i * S.np + lane] : 0.0;
        acc = fma(bv, xi, acc);
    }
    double w = st.h - acc;
    if (lam != 0.0) w = fma(-lam, band_mul(bd.kb, st.x, lane), w);
    return w;
}

// Lawson-Hanson's secondary loop: from a feasible x and a factor consistent with (P, lambda), move to
// the solution of the passive sub-problem, dropping variables that hit zero on the way.
// Returns false when the iteration cap is reached.
__device__ __forceinline__ bool nnls_inner(const WaveShared &S, NnlsState &st, int &iter, int itmax, int lane)
{
    for (;;) {
        if (++iter > itmax) return false;
        double z = back_subst(S, st, lane);                 // position-indexed
        double xp = gather(st.x, st.ord);                   // x at position
        bool neg = (lane < st.k) && (z <= 0.0);
        u64 negm = ballot(neg);
        double zb = gather(z, st.pos < 0 ? 0 : st.pos);     // bin-indexed
        zb = (st.pos >= 0) ? zb : 0.0;
        if (!negm) { st.x = zb; return true; }
        double ratio = neg ? xp / (xp - z) : 2.0;
        ratio = (ratio == ratio) ? ratio : 2.0;             // 0/0: Lawson-Hanson's `alpha > t` is false for NaN
        double alpha = wave_min(ratio);
        if (!(alpha < 2.0)) { st.x = zb; return true; }     // "alpha still 2": accept z
        int jj = first_lane(ballot(neg && ratio == alpha));
        st.x = (st.pos >= 0) ? fma(alpha, zb - st.x, st.x) : 0.0;
        remove_pos(S, st, jj, lane);
        for (int sweep = 0; sweep < 64; ++sweep) {   // round-off stragglers (Lawson-Hanson: "any that are nonpositive ...")
            double xq = gather(st.x, st.ord);
            u64 bad = ballot((lane < st.k) && (xq <= 0.0));
            if (!bad) break;
            MET2_STAT(1, sweep + 1);
            remove_pos(S, st, first_lane(bad), lane);
        }
    }
}

// Passive-set iterations until the KKT conditions hold.  mrows = rows of the (augmented) system.
// warm: x is a feasible point whose support is the current passive set and R/y were just rebuilt for
// (P, lam) -- start with the secondary loop instead of from the empty set.
__device__ __forceinline__ void nnls_iterate(const WaveShared &S, const Band &bd, NnlsState &st, double lam, int mrows, int lane,
                                             bool warm = false)
{
    const int n = S.n, itmax = 3 * n;
    int iter = 0;
    if (warm && st.k > 0 && !nnls_inner(S, st, iter, itmax, lane)) { st.itmax_hit = 1; return; }
    for (int outer = 0; outer <= itmax + 1; ++outer) {     // every pass runs >= 1 counted inner pass
        if (st.k >= n || st.k >= mrows || st.k >= S.kmax) break;
        double w = dual(S, bd, st, lam, lane);
        // entering variable: largest positive dual among Z; rejected candidates are skipped
        u64 rejected = 0;
        bool accepted = false;
        for (int tries = 0; tries < 64; ++tries) {           // each failed try rejects one more bin
            bool cand = (lane < n) && !((st.P >> lane) & 1ull) && !((rejected >> lane) & 1ull);
            double val = cand ? w : -1.0;
            double wmax = wave_max(val);
            if (!(wmax > 0.0)) break;
            int t = first_lane(ballot(cand && val == wmax));
            if (try_append(S, bd, st, lam, t, lane)) { accepted = true; break; }
            rejected |= (1ull << t);
            MET2_STAT(0, tries + 1);
        }
        if (!accepted) break;
        if (!nnls_inner(S, st, iter, itmax, lane)) { st.itmax_hit = 1; break; }
        MET2_STAT(2, outer + 1);
        MET2_STAT(3, iter);
    }
}

__device__ __forceinline__ void nnls_reset(NnlsState &st)
{
    st.x = 0.0; st.y = 0.0; st.rinv = 0.0; st.ord = 0; st.pos = -1; st.k = 0; st.P = 0ull;
}

// cold-start solve; on return st.x is the solution
__device__ __forceinline__ void nnls_solve(const WaveShared &S, const Band &bd, NnlsState &st, double lam, bool aug, int lane)
{
    nnls_reset(st);
    nnls_iterate(S, bd, st, lam, aug ? S.m + S.n : S.m, lane);
}

// Warm start: keep the previous solution's passive set and x (a feasible point for any lambda),
// rebuild the factor for the new lambda in the same pivot order, then iterate.  The minimiser of the
// strictly convex problem does not depend on the starting point, so this returns the same x as the
// cold start up to rounding; it only skips the passes that would rebuild the same passive set.
__device__ __forceinline__ void nnls_solve_warm(const WaveShared &S, const Band &bd, NnlsState &st, double lam, bool aug, int lane)
{
    const int kold = st.k, ordold = st.ord;
    if (kold == 0) { nnls_solve(S, bd, st, lam, aug, lane); return; }
    st.k = 0; st.P = 0ull; st.pos = -1;
    for (int p = 0; p < kold; ++p) {
        const int t = bcast_i(ordold, p);
        if (!try_append(S, bd, st, lam, t, lane, true) && lane == t) st.x = 0.0;   // column became dependent: drop it
    }
    nnls_iterate(S, bd, st, lam, aug ? S.m + S.n : S.m, lane, true);
}

// D x  (lane e < m holds (D x)_e), using the passive set of st
__device__ __forceinline__ double model_signal(const WaveShared &S, const NnlsState &st, int lane)
{
    double acc = 0.0;
    for (int p = 0; p < st.k; ++p) {
        int i = bcast_i(st.ord, p);
        double xi = bcast(st.x, i);
        double dv = (lane < S.m) ? S.sD[lane * S.np + i] : 0.0;
        acc = fma(dv, xi, acc);
    }
    return acc;
}

// || D x - b ||^2
__device__ __forceinline__ double sse_of(const WaveShared &S, const NnlsState &st, double b, int lane)
{
    double r = model_signal(S, st, lane) - b;
    r = (lane < S.m) ? r : 0.0;
    return wave_sum(r * r);
}

} // namespace met2
