// objectives.hpp -- lambda-selection objectives that need more than the NNLS solve:
//   BayesReg  bayesian_interpolation.py:107-126  full n x n Cholesky of beta (B + lambda K), erf, log
//   GCV       algorithms.py:285-296              truncated pseudo-inverse of the support Gram matrix
// Both reuse the wave's packed-triangle LDS region (kmax == n) once the NNLS solution is in st.x.
#pragma once
#include "nnls_wave.hpp"

namespace met2 {

__device__ __forceinline__ double op_mul(double a, double b) { return a * b; }
__device__ __forceinline__ double wave_prod(double v)
{
    MET2_ROW_REDUCE(v, op_mul)
    return (bcast(v, 0) * bcast(v, 16)) * (bcast(v, 32) * bcast(v, 48));
}

// Upper Cholesky factor U (A = U^T U) of A = beta*B + (beta*lam)*K, rows packed into S.R.
// Lane i owns column i.  Returns false when a pivot is not positive (scipy raises LinAlgError).
__device__ __forceinline__ bool chol_full(const WaveShared &S, const Band &bd, double beta, double lam, int lane, double &det_u)
{
    const int n = S.n, kmax = S.kmax;
    const double bl = beta * lam;
    double diag = 1.0;      // lane j keeps U[j][j]
    for (int j = 0; j < n; ++j) {
        // a = A[j][lane]  (K[lane][j] = kb[j - lane + 2] of lane's own row)
        double a = (lane < n) ? beta * S.sB[j * S.np + lane] : 0.0;
        a = fma(bl, band_pick(bd.kb, j - lane), a);
        // column j of the rows already computed: lane k < j holds U[k][j]
        double colj = (lane < j) ? S.R[row_base(lane, kmax) + j] : 0.0;
        double s = 0.0;
        for (int k = 0; k < j; ++k) {
            double ukj = bcast(colj, k);
            double uki = (lane >= j && lane < n) ? S.R[row_base(k, kmax) + lane] : 0.0;
            s = fma(ukj, uki, s);
        }
        double v = a - s;
        double d = bcast(v, j);
        if (!(d > 0.0)) return false;
        double ujj = sqrt(d);
        double u = (lane == j) ? ujj : v / ujj;
        if (lane >= j && lane < n) S.R[row_base(j, kmax) + lane] = u;
        if (lane == j) diag = ujj;
        __builtin_amdgcn_wave_barrier();
    }
    det_u = wave_prod(lane < n ? diag : 1.0);
    return true;
}

// (U f)_i for the factor in S.R; lane i < n gets row i . f
__device__ __forceinline__ double upper_times(const WaveShared &S, double f, int lane)
{
    const int n = S.n, kmax = S.kmax;
    const int rbl = row_base(lane, kmax);
    double acc = 0.0;
    for (int j = 0; j < n; ++j) {
        double fj = bcast(f, j);
        double u = (lane <= j && lane < n) ? S.R[rbl + j] : 0.0;
        acc = fma(u, fj, acc);
    }
    return acc;
}

struct BayesCtx {
    double beta, log_detL;
    int failed;     // Cholesky failure seen
};

// bayesian_interpolation.py:107-126, given the NNLS solution st.x at lambda = x
__device__ __forceinline__ double bayes_objective(const WaveShared &S, const Band &bd, const NnlsState &st, BayesCtx &bc,
                                                  double x, double b, int lane)
{
    const int n = S.n, m = S.m;
    const double beta = bc.beta;
    const double ED = 0.5 * sse_of(S, st, b, lane);
    double lf = band_mul(bd.lb, st.x, lane);
    lf = (lane < n) ? lf : 0.0;
    const double EW = 0.5 * wave_sum(lf * lf);
    double det_u;
    if (!chol_full(S, bd, beta, x, lane, det_u)) { bc.failed = 1; return NAN; }
    double uf = upper_times(S, st.x, lane);
    double term = (lane < n) ? log(1.0 + erf((1.0 / sqrt(2.0)) * uf)) : 0.0;
    const double series = wave_sum(term);
    const double PI = M_PI;
    double cost1 = beta * ED + beta * x * EW + log(det_u) - (n / 2.0) * log(PI / 2.0) - series;
    double cost2 = (m / 2.0) * log(2.0 * PI) - (m / 2.0) * log(beta) + (n / 2.0) * log(PI) - (n / 2.0) * log(2 * beta * x) - bc.log_detL;
    return cost1 + cost2;
}

// algorithms.py:285-296 given the NNLS solution st.x at lambda = x.
// trace(Dr G^+ Dr^T) with G = Dr^T Dr + x*(sum_S L_jj^2) * ones, G^+ = SVD-truncated pseudo-inverse
// (singular values <= eps*k*s_max dropped, np.linalg.lstsq(rcond=None)).  One-sided Jacobi on the
// columns of G held in S.R (k*k <= kmax(kmax+1)/2), lane = row; u = 1^T V is carried along so that
// trace = sum_retained (1 - c u_i^2 / s_i)  (G symmetric: G v_i = s_i v_i, Dr^T Dr = G - c 11^T).
__device__ __forceinline__ double gcv_objective(const WaveShared &S, const Band &bd, const NnlsState &st, double x, double b,
                                                int lane, int &overflow)
{
    const int n = S.n, m = S.m;
    const double sse = sse_of(S, st, b, lane);
    double lf = band_mul(bd.lb, st.x, lane);
    lf = (lane < n) ? lf : 0.0;
    const double rn2 = sse + x * wave_sum(lf * lf);           // squared residual norm of the augmented system
    const bool inS = (lane < n) && (st.x > 0.0);
    const u64 Sm = ballot(inS);
    const int k = __popcll(Sm);
    if (k == 0) return NAN;
    if (k * k > S.rcap) { overflow = 1; return INFINITY; }
    const double ld = bd.lb[2];
    const double c = x * wave_sum(inS ? ld * ld : 0.0);
    // lane p < k learns its bin s_p (p-th set bit of Sm)
    const int rank = __popcll(Sm & ((1ull << lane) - 1ull));
    int sp = __builtin_amdgcn_ds_permute((inS ? rank : 63) << 2, lane);   // push bin index to lane `rank`
    sp = (lane < k) ? sp : 0;
    double *A = S.R;                                                       // column-major k x k
    for (int q = 0; q < k; ++q) {
        int sq = bcast_i(sp, q);
        if (lane < k) A[q * k + lane] = S.sB[sq * S.np + sp] + c;
    }
    __builtin_amdgcn_wave_barrier();
    double u = (lane < k) ? 1.0 : 0.0;      // lane q holds u_q = 1^T v_q
    for (int sweep = 0; sweep < 40; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                double ap = (lane < k) ? A[p * k + lane] : 0.0;
                double aq = (lane < k) ? A[q * k + lane] : 0.0;
                double alpha = ap * ap, beta = aq * aq;
                wave_sum2(alpha, beta);
                double gamma = wave_sum(ap * aq);
                if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                if (lane < k) { A[p * k + lane] = cs * ap - sn * aq; A[q * k + lane] = sn * ap + cs * aq; }
                double up = bcast(u, p), uq = bcast(u, q);
                if (lane == p) u = cs * up - sn * uq;
                if (lane == q) u = sn * up + cs * uq;
            }
        if (!rotated) break;
    }
    __builtin_amdgcn_wave_barrier();
    // singular values = column norms: lane q computes ||A[:,q]|| by a serial loop over rows
    double s2 = 0.0;
    for (int r = 0; r < k; ++r) {
        double v = (lane < k) ? A[lane * k + r] : 0.0;
        s2 = fma(v, v, s2);
    }
    const double sv = sqrt(s2);
    const double smax = wave_max(lane < k ? sv : 0.0);
    const double cut = 2.220446049250313e-16 * (double)k * smax;
    const bool keep = (lane < k) && (sv > cut);
    // sign of the eigenvalue: (1^T A_q) = lambda_q u_q, so lambda_q = sign * sv
    double colsum = 0.0;
    for (int r = 0; r < k; ++r) colsum += (lane < k) ? A[lane * k + r] : 0.0;
    double lamq = (colsum * u >= 0.0) ? sv : -sv;
    double tr = wave_sum(keep ? (1.0 - c * u * u / lamq) : 0.0);
    const double num = (1.0 / m) * rn2;
    const double den = (1.0 / m) * ((double)m - tr);
    return log(num / (den * den));
}

} // namespace met2
