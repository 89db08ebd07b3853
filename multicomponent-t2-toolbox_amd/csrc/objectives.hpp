// objectives.hpp -- lambda-selection objectives that need more than the NNLS solve:
//   BayesReg  bayesian_interpolation.py:107-126  full n x n Cholesky of beta (B + lambda K), erf, log
//   GCV       algorithms.py:285-296              truncated pseudo-inverse of the support Gram matrix
// Both reuse the wave's LDS region once the NNLS solution is in st.x (the factor is rebuilt by the
// next warm start).
#pragma once
#include "nnls_wave.hpp"

namespace met2 {

__device__ __forceinline__ double op_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double wave_prod(double v)
{
    MET2_ROW_REDUCE(v, op_mul)
    return (bcast(v, 0) * bcast(v, 16)) * (bcast(v, 32) * bcast(v, 48));
}

// Upper Cholesky factor U (A = U^T U) of A = beta*B + (beta*lam)*K, rows packed into S.R
// (needs kmax == n).  A lane owns columns lane + 64 b.  Returns false when a pivot is not positive
// (scipy raises LinAlgError there).
template <int NB>
__device__ __forceinline__ bool chol_full(const WaveShared &S, const Band<NB> &bd, double beta, double lam, int lane, double &det_u)
{
    const int n = S.n, kmax = S.kmax;
    const double bl = beta * lam;
    double diag[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) diag[b] = 1.0;
    for (int j = 0; j < n; ++j) {
        double a[NB], colj[NB], s[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c = lane + 64 * b;
            a[b] = (c < n) ? beta * S.B[j * S.bstride + c] : 0.0;
            a[b] = fma(bl, band_pick(bd.kb[b], j - c), a[b]);          // A[j][c]
            colj[b] = (c < j) ? S.R[row_base(c, kmax) + j] : 0.0;      // U[c][j] of the rows already done
            s[b] = 0.0;
        }
        for (int k = 0; k < j; ++k) {
            const double ukj = bcastN<NB>(colj, k);
            const int rb = row_base(k, kmax);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int c = lane + 64 * b;
                double uki = (c >= j && c < n) ? S.R[rb + c] : 0.0;
                s[b] = fma(ukj, uki, s[b]);
            }
        }
        double v[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) v[b] = a[b] - s[b];
        const double d = bcastN<NB>(v, j);
        if (!(d > 0.0)) return false;
        const double ujj = sqrt(d);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c = lane + 64 * b;
            double u = (c == j) ? ujj : v[b] / ujj;
            if (c >= j && c < n) S.R[row_base(j, kmax) + c] = u;
            if (c == j) diag[b] = ujj;
        }
        __builtin_amdgcn_wave_barrier();
    }
    double dp = 1.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) dp *= (lane + 64 * b < n) ? diag[b] : 1.0;
    det_u = wave_prod(dp);
    return true;
}

// (U f)_i for the factor in S.R; the owner of row i gets row i . f
template <int NB>
__device__ __forceinline__ void upper_times(const WaveShared &S, const double (&f)[NB], int lane, double (&out)[NB])
{
    const int n = S.n, kmax = S.kmax;
    int rbl[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { rbl[b] = row_base(lane + 64 * b, kmax); out[b] = 0.0; }
    for (int j = 0; j < n; ++j) {
        const double fj = bcastN<NB>(f, j);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int i = lane + 64 * b;
            double u = (i <= j && i < n) ? S.R[rbl[b] + j] : 0.0;
            out[b] = fma(u, fj, out[b]);
        }
    }
}

struct BayesCtx {
    double beta, log_detL;
    int failed;     // Cholesky failure seen
};

// bayesian_interpolation.py:107-126, given the NNLS solution st.x at lambda = x
template <int NB>
__device__ __forceinline__ double bayes_objective(const WaveShared &S, const Band<NB> &bd, const NnlsState<NB> &st, BayesCtx &bc,
                                                  double x, double b, int lane)
{
    const int n = S.n, m = S.m;
    const double beta = bc.beta;
    const double ED = 0.5 * sse_of<NB>(S, st, b, lane);
    const double EW = 0.5 * seminorm2<NB>(bd, st.x, n, lane);
    double det_u;
    if (!chol_full<NB>(S, bd, beta, x, lane, det_u)) { bc.failed = 1; return NAN; }
    double uf[NB];
    upper_times<NB>(S, st.x, lane, uf);
    double term = 0.0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) term += (lane + 64 * bb < n) ? log(1.0 + erf((1.0 / sqrt(2.0)) * uf[bb])) : 0.0;
    const double series = wave_sum(term);
    const double PI = M_PI;
    double cost1 = beta * ED + beta * x * EW + log(det_u) - (n / 2.0) * log(PI / 2.0) - series;
    double cost2 = (m / 2.0) * log(2.0 * PI) - (m / 2.0) * log(beta) + (n / 2.0) * log(PI) - (n / 2.0) * log(2 * beta * x) - bc.log_detL;
    return cost1 + cost2;
}

// algorithms.py:285-296 given the NNLS solution st.x at lambda = x.
// trace(Dr G^+ Dr^T) with G = Dr^T Dr + x*(sum_S L_jj^2) * ones, G^+ = SVD-truncated pseudo-inverse
// (singular values <= eps*k*s_max dropped, np.linalg.lstsq(rcond=None)).  One-sided Jacobi on the
// columns of G held column-major in S.R (k*k <= rcap), a lane owns rows lane + 64 b; u = 1^T V is carried
// along so that trace = sum_retained (1 - c u_i^2 / lambda_i)  (G symmetric: G v_i = lambda_i v_i,
// Dr^T Dr = G - c 11^T).
template <int NB>
__device__ __forceinline__ double gcv_objective(const WaveShared &S, const Band<NB> &bd, const NnlsState<NB> &st, double x, double b,
                                                int lane, int &overflow)
{
    const int n = S.n, m = S.m;
    const double sse = sse_of<NB>(S, st, b, lane);
    const double rn2 = sse + x * seminorm2<NB>(bd, st.x, n, lane);   // squared residual norm of the augmented system
    bool inS[NB];
    u64 Sm[NB];
    int k = 0;
    double l2 = 0.0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        inS[bb] = (lane + 64 * bb < n) && (st.x[bb] > 0.0);
        Sm[bb] = ballot(inS[bb]);
        k += __popcll(Sm[bb]);
        const double ld = bd.lb[bb][2];
        l2 += inS[bb] ? ld * ld : 0.0;
    }
    if (k == 0) return NAN;
    if (k * k > S.rcap) { overflow = 1; return INFINITY; }
    const double c = x * wave_sum(l2);
    // support list through LDS: rank-th support bin -> sp of the owner of row `rank`
    int *list = (int *)S.R;
    int base = 0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        const int rank = base + __popcll(Sm[bb] & ((1ull << lane) - 1ull));
        if (inS[bb]) list[rank] = lane + 64 * bb;
        base += __popcll(Sm[bb]);
    }
    __builtin_amdgcn_wave_barrier();
    int sp[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) sp[bb] = (lane + 64 * bb < k) ? list[lane + 64 * bb] : 0;
    __builtin_amdgcn_wave_barrier();
    double *A = S.R;                                                       // column-major k x k
    for (int q = 0; q < k; ++q) {
        const int sq = bcastN_i<NB>(sp, q);
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const int r = lane + 64 * bb;
            if (r < k) A[q * k + r] = S.B[sq * S.bstride + sp[bb]] + c;
        }
    }
    __builtin_amdgcn_wave_barrier();
    double u[NB];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) u[bb] = (lane + 64 * bb < k) ? 1.0 : 0.0;      // owner of q holds u_q = 1^T v_q
    for (int sweep = 0; sweep < 40; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                double ap[NB], aq[NB], alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) {
                    const int r = lane + 64 * bb;
                    ap[bb] = (r < k) ? A[p * k + r] : 0.0;
                    aq[bb] = (r < k) ? A[q * k + r] : 0.0;
                    alpha = fma(ap[bb], ap[bb], alpha); beta = fma(aq[bb], aq[bb], beta); gamma = fma(ap[bb], aq[bb], gamma);
                }
                wave_sum2(alpha, beta);
                gamma = wave_sum(gamma);
                if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                const double up = bcastN<NB>(u, p), uq = bcastN<NB>(u, q);
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) {
                    const int r = lane + 64 * bb;
                    if (r < k) { A[p * k + r] = cs * ap[bb] - sn * aq[bb]; A[q * k + r] = sn * ap[bb] + cs * aq[bb]; }
                    if (r == p) u[bb] = cs * up - sn * uq;
                    if (r == q) u[bb] = sn * up + cs * uq;
                }
            }
        if (!rotated) break;
    }
    __builtin_amdgcn_wave_barrier();
    // singular values = column norms: the owner of q runs down column q
    double sv[NB], colsum[NB], svmax = 0.0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        const int q = lane + 64 * bb;
        double s2 = 0.0, cs = 0.0;
        for (int r = 0; r < k; ++r) {
            double v = (q < k) ? A[q * k + r] : 0.0;
            s2 = fma(v, v, s2); cs += v;
        }
        sv[bb] = sqrt(s2); colsum[bb] = cs;
        svmax = fmax(svmax, q < k ? sv[bb] : 0.0);
    }
    const double smax = wave_max(svmax);
    const double cut = 2.220446049250313e-16 * (double)k * smax;
    double trp = 0.0;
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        const bool keep = (lane + 64 * bb < k) && (sv[bb] > cut);
        // sign of the eigenvalue: 1^T A_q = lambda_q u_q
        const double lamq = (colsum[bb] * u[bb] >= 0.0) ? sv[bb] : -sv[bb];
        trp += keep ? (1.0 - c * u[bb] * u[bb] / lamq) : 0.0;
    }
    const double tr = wave_sum(trp);
    const double num = (1.0 / m) * rn2;
    const double den = (1.0 / m) * ((double)m - tr);
    return log(num / (den * den));
}

} // namespace met2
