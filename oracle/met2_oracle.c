/*
 * met2_oracle.c -- CPU restatement (plain C, fp64) of the reference's per-voxel
 * regularised-NNLS T2-spectrum path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity oracle for the HIP path and the
 * `cpu_baseline` ("port") leg of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path never calls it.
 *
 * Parity pinning: every function below is checked against golden vectors produced by
 * running the reference itself in the build container (tests/golden/make_goldens.py,
 * tests/test_oracle_golden.py).
 *
 * Reference anchors (all paths relative to the reference repository):
 *   epg/epg.py:47-162                         -> o_epg_*, met2o_dictionary
 *   motor/motor_recon_met2_real_data.py:86-111, 263-269 -> met2o_penalty
 *   intravoxel_algorithms/algorithms.py:55-82 -> o_nnls   (arithmetic: Lawson & Hanson,
 *        "Solving Least Squares Problems", ch. 23 -- SciPy's nnls is this algorithm)
 *   intravoxel_algorithms/algorithms.py:262-269 -> o_tik
 *   intravoxel_algorithms/algorithms.py:211-233 -> o_x2
 *   intravoxel_algorithms/algorithms.py:88-113,150-206 -> o_lcurve, o_select_corner
 *   intravoxel_algorithms/algorithms.py:276-296 -> o_gcv
 *   intravoxel_algorithms/bayesian_interpolation.py:84-126 -> o_bayes
 *   scipy.optimize.fminbound (third party, bounded Brent) -> o_fminbound
 *   motor/motor_recon_met2_real_data.py:113-162 -> met2o_fit_batch
 *   motor/motor_recon_met2_real_data.py:443-472 -> met2o_metrics
 *   flip_angle_algorithms/fa_estimation.py:74-90 -> met2o_fa_bruteforce
 *   motor/motor_recon_met2_real_data.py:305-333 -> met2o_nesma
 *   motor/motor_recon_met2_real_data.py:337-343 -> met2o_gaussian_smooth (scipy.ndimage.gaussian_filter, SciPy 1.x:
 *       gaussian_filter1d per axis = correlate1d with the normalised kernel, mode 'reflect', truncate 4)
 *   flip_angle_algorithms/fa_estimation.py:35-70 -> met2o_fa_spline (scipy interp1d(kind='cubic') =
 *        not-a-knot cubic spline; scipy minimize_scalar(method='Bounded') = the same bounded Brent)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MET2O_API __attribute__((visibility("default")))

enum { M_NNLS = 0, M_T2SPARC = 1, M_X2 = 2, M_LCURVE = 3, M_GCV = 4, M_BAYES = 5 };

/* status bits (same meaning as include/met2_hip.h) */
enum { ST_FITTED = 1, ST_ITMAX = 2, ST_NONFINITE = 4, ST_CHOLFAIL = 8, ST_BRENT_MAXFUN = 16 };

/* ------------------------------------------------------------------ EPG (E1-E3) */
/* epg/epg.py:64-153.  State layout: F0 | (F_k, F_-k, Z_k) k=1..n.  One inter-echo
 * period is P*T*P with P = relax(tau/2) o shift and T the per-order RF mixing; the
 * echo amplitude is F0 after each period.  F0 itself is not mixed by T (epg.py:125). */
static void o_epg_one(int n, double tau, double R1, double R2, double alpha, double alpha_exc,
                      double *out, int ostride, double *work /* 6*(n+2) */)
{
    double *Fp = work, *Fm = work + (n + 2), *Z = work + 2 * (n + 2);
    double *Fp2 = work + 3 * (n + 2), *Fm2 = work + 4 * (n + 2), *Z2 = work + 5 * (n + 2);
    double th = tau / 2.0;
    double E2 = exp(-th * R2), E1 = exp(-th * R1);
    double ca2 = cos(alpha / 2.0), sa2 = sin(alpha / 2.0);
    double c2 = ca2 * ca2, s2 = sa2 * sa2, sa = sin(alpha), ca = cos(alpha);
    double F0 = sin(alpha_exc);
    memset(work, 0, sizeof(double) * 6 * (n + 2));
    Fm[1] = cos(alpha_exc);
    for (int e = 0; e < n; ++e) {
        for (int half = 0; half < 2; ++half) {
            /* shift + relax (P) */
            double nF0 = Fm[1];
            Fp2[1] = F0;
            for (int k = 2; k <= n; ++k) Fp2[k] = Fp[k - 1];
            for (int k = 1; k < n; ++k) Fm2[k] = Fm[k + 1];
            Fm2[n] = 0.0;
            F0 = nF0 * E2;
            for (int k = 1; k <= n; ++k) {
                Fp[k] = Fp2[k] * E2;
                Fm[k] = Fm2[k] * E2;
                Z[k] = Z[k] * E1;
            }
            if (half == 0) { /* RF mixing (T) */
                for (int k = 1; k <= n; ++k) {
                    double a = Fp[k], b = Fm[k], z = Z[k];
                    Fp2[k] = c2 * a + s2 * b + sa * z;
                    Fm2[k] = s2 * a + c2 * b - sa * z;
                    Z2[k] = -0.5 * sa * a + 0.5 * sa * b + ca * z;
                }
                for (int k = 1; k <= n; ++k) { Fp[k] = Fp2[k]; Fm[k] = Fm2[k]; Z[k] = Z2[k]; }
            }
        }
        out[(size_t)e * ostride] = F0;
    }
}

MET2O_API void met2o_epg_signal(int n, double tau, int nrates, const double *R1, const double *R2,
                                double alpha, double alpha_exc, double *H /* [n][nrates] */)
{
    double *work = (double *)malloc(sizeof(double) * 6 * (n + 2));
    for (int r = 0; r < nrates; ++r) o_epg_one(n, tau, R1[r], R2[r], alpha, alpha_exc, H + r, nrates, work);
    free(work);
}

/* Dictionary in the device/oracle layout  D[fa][te][t2]  (the Python wrapper converts
 * to/from the reference's [te][t2][fa], epg.py:155-162). */
MET2O_API void met2o_dictionary(int nte, int nt2, int nfa, const double *T2s, const double *T1s,
                                double tau, const double *alpha_deg, double TR, double *D)
{
    const double rad = M_PI / 180.0;
    double *work = (double *)malloc(sizeof(double) * 6 * (nte + 2));
    for (int f = 0; f < nfa; ++f)
        for (int j = 0; j < nt2; ++j) {
            double *col = D + ((size_t)f * nte) * nt2 + j;
            o_epg_one(nte, tau, 1.0 / T1s[j], 1.0 / T2s[j], alpha_deg[f] * rad, alpha_deg[f] / 2.0 * rad, col, nt2, work);
            double sc = 1.0 - exp(-TR / T1s[j]);
            for (int e = 0; e < nte; ++e) col[(size_t)e * nt2] *= sc;
        }
    free(work);
}

/* ------------------------------------------------------------------ penalties (P1) */
/* order 0/1/2 -> I / L1 / L2 (motor:86-111); order 3 -> InvT2 (motor:263-269) */
MET2O_API void met2o_penalty(int n, int order, const double *T2s, double *L)
{
    memset(L, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) {
        if (order == 0) L[i * n + i] = 1.0;
        else if (order == 1) { L[i * n + i] = 1.0; if (i > 0) L[i * n + i - 1] = -1.0; }
        else if (order == 2) {
            L[i * n + i] = 2.0;
            if (i > 0) L[i * n + i - 1] = -1.0;
            if (i < n - 1) L[i * n + i + 1] = -1.0;
        } else {
            double prev = (i == 0) ? T2s[0] - 1.0 : T2s[i - 1];
            double d = T2s[i] - prev;
            if (i == 0) d = T2s[1] - T2s[0];
            L[i * n + i] = 1.0 / d;
        }
    }
    if (order == 2) { L[0] = 1.0; L[(n - 1) * n + n - 1] = 1.0; }
}

/* ------------------------------------------------------------------ NNLS (N1) */
/* Householder construct on column u (stride 1), pivot lp, zeroing rows l1..m-1 */
static void h12_construct(int lp, int l1, int m, double *u, double *up)
{
    double cl = fabs(u[lp]);
    for (int j = l1; j < m; ++j) if (fabs(u[j]) > cl) cl = fabs(u[j]);
    if (cl <= 0.0) { *up = 0.0; return; }
    double clinv = 1.0 / cl;
    double sm = (u[lp] * clinv) * (u[lp] * clinv);
    for (int j = l1; j < m; ++j) sm += (u[j] * clinv) * (u[j] * clinv);
    cl *= sqrt(sm);
    if (u[lp] > 0.0) cl = -cl;
    *up = u[lp] - cl;
    u[lp] = cl;
}

static void h12_apply(int lp, int l1, int m, const double *u, double up, double *c)
{
    double b = up * u[lp];
    if (b >= 0.0) return;
    b = 1.0 / b;
    double sm = c[lp] * up;
    for (int i = l1; i < m; ++i) sm += c[i] * u[i];
    if (sm != 0.0) {
        sm *= b;
        c[lp] += sm * up;
        for (int i = l1; i < m; ++i) c[i] += sm * u[i];
    }
}

static void g1(double a, double b, double *c, double *s, double *sig)
{
    if (fabs(a) > fabs(b)) {
        double xr = b / a, yr = sqrt(1.0 + xr * xr);
        *c = copysign(1.0 / yr, a); *s = (*c) * xr; *sig = fabs(a) * yr;
    } else if (b != 0.0) {
        double xr = a / b, yr = sqrt(1.0 + xr * xr);
        *s = copysign(1.0 / yr, b); *c = (*s) * xr; *sig = fabs(b) * yr;
    } else { *sig = 0.0; *c = 0.0; *s = 1.0; }
}

typedef struct {
    int m, n;
    double *a;   /* column-major m x n working copy */
    double *b, *zz, *w;
    int *index;
} nnls_ws;

static nnls_ws *ws_new(int m, int n)
{
    nnls_ws *w = (nnls_ws *)malloc(sizeof(nnls_ws));
    w->m = m; w->n = n;
    w->a = (double *)malloc(sizeof(double) * m * n);
    w->b = (double *)malloc(sizeof(double) * m);
    w->zz = (double *)malloc(sizeof(double) * m);
    w->w = (double *)malloc(sizeof(double) * n);
    w->index = (int *)malloc(sizeof(int) * n);
    return w;
}
static void ws_free(nnls_ws *w) { free(w->a); free(w->b); free(w->zz); free(w->w); free(w->index); free(w); }

/* Lawson-Hanson NNLS on ws->a (col-major, destroyed) and ws->b (destroyed).
 * Returns mode (1 ok, 3 iteration cap); x[n], *rnorm out.  itmax = 3n as in the
 * reference (algorithms.py:69, maxiter=-1). */
static int o_nnls_core(nnls_ws *W, int m, int n, double *x, double *rnorm)
{
    double *a = W->a, *b = W->b, *zz = W->zz, *w = W->w;
    int *index = W->index;
    const double factor = 0.01;
    int mode = 1, iter = 0, itmax = 3 * n;
    for (int i = 0; i < n; ++i) { x[i] = 0.0; index[i] = i; }
    int iz2 = n - 1, iz1 = 0, nsetp = 0, npp1 = 0;
#define A_(r, c) a[(size_t)(c) * m + (r)]
    for (;;) {
        if (iz1 > iz2 || nsetp >= m) break;
        for (int iz = iz1; iz <= iz2; ++iz) {
            int j = index[iz];
            double sm = 0.0;
            for (int l = npp1; l < m; ++l) sm += A_(l, j) * b[l];
            w[j] = sm;
        }
        int izsel = -1, jsel = -1;
        double up = 0.0;
        for (;;) {
            double wmax = 0.0; int izmax = -1;
            for (int iz = iz1; iz <= iz2; ++iz) {
                int j = index[iz];
                if (w[j] > wmax) { wmax = w[j]; izmax = iz; }
            }
            if (wmax <= 0.0) break;
            int iz = izmax, j = index[iz];
            double asave = A_(npp1, j);
            h12_construct(npp1, npp1 + 1, m, &A_(0, j), &up);
            double unorm = 0.0;
            for (int l = 0; l < nsetp; ++l) unorm += A_(l, j) * A_(l, j);
            unorm = sqrt(unorm);
            volatile double t1 = unorm + fabs(A_(npp1, j)) * factor;
            if ((t1 - unorm) > 0.0) {
                memcpy(zz, b, sizeof(double) * m);
                h12_apply(npp1, npp1 + 1, m, &A_(0, j), up, zz);
                double ztest = zz[npp1] / A_(npp1, j);
                if (ztest > 0.0) { izsel = iz; jsel = j; break; }
            }
            A_(npp1, j) = asave;
            w[j] = 0.0;
        }
        if (izsel < 0) break;
        {
            int iz = izsel, j = jsel;
            memcpy(b, zz, sizeof(double) * m);
            index[iz] = index[iz1]; index[iz1] = j; iz1++; nsetp = npp1 + 1; npp1++;
            for (int jz = iz1; jz <= iz2; ++jz) {
                int jj = index[jz];
                h12_apply(nsetp - 1, npp1, m, &A_(0, j), up, &A_(0, jj));
            }
            for (int l = npp1; l < m; ++l) A_(l, j) = 0.0;
            w[j] = 0.0;
        }
        /* solve triangular system into zz */
        memcpy(zz, b, sizeof(double) * m);
        for (int l = 0; l < nsetp; ++l) {
            int ip = nsetp - 1 - l;
            if (l != 0) { int jj0 = index[ip + 1]; for (int ii = 0; ii <= ip; ++ii) zz[ii] -= A_(ii, jj0) * zz[ip + 1]; }
            int jj0 = index[ip];
            zz[ip] /= A_(ip, jj0);
        }
        int done_outer = 0;
        for (;;) {
            iter++;
            if (iter > itmax) { mode = 3; done_outer = 1; break; }
            double alpha = 2.0; int jj = -1;
            for (int ip = 0; ip < nsetp; ++ip) {
                int l = index[ip];
                if (zz[ip] <= 0.0) {
                    double t = -x[l] / (zz[ip] - x[l]);
                    if (alpha > t) { alpha = t; jj = ip; }
                }
            }
            if (alpha == 2.0) break;
            for (int ip = 0; ip < nsetp; ++ip) { int l = index[ip]; x[l] += alpha * (zz[ip] - x[l]); }
            int i = index[jj];
            for (;;) {
                x[i] = 0.0;
                if (jj != nsetp - 1) {
                    jj++;
                    for (int j = jj; j < nsetp; ++j) {
                        int ii = index[j];
                        index[j - 1] = ii;
                        double cc, ss, sig;
                        g1(A_(j - 1, ii), A_(j, ii), &cc, &ss, &sig);
                        A_(j - 1, ii) = sig; A_(j, ii) = 0.0;
                        for (int l = 0; l < n; ++l) if (l != ii) {
                            double temp = A_(j - 1, l);
                            A_(j - 1, l) = cc * temp + ss * A_(j, l);
                            A_(j, l) = -ss * temp + cc * A_(j, l);
                        }
                        double temp = b[j - 1];
                        b[j - 1] = cc * temp + ss * b[j];
                        b[j] = -ss * temp + cc * b[j];
                    }
                }
                npp1 = nsetp - 1; nsetp--; iz1--; index[iz1] = i;
                int again = 0;
                for (jj = 0; jj < nsetp; ++jj) { i = index[jj]; if (x[i] <= 0.0) { again = 1; break; } }
                if (!again) break;
            }
            memcpy(zz, b, sizeof(double) * m);
            for (int l = 0; l < nsetp; ++l) {
                int ip = nsetp - 1 - l;
                if (l != 0) { int jj0 = index[ip + 1]; for (int ii = 0; ii <= ip; ++ii) zz[ii] -= A_(ii, jj0) * zz[ip + 1]; }
                int jj0 = index[ip];
                zz[ip] /= A_(ip, jj0);
            }
        }
        if (done_outer) break;
        for (int ip = 0; ip < nsetp; ++ip) x[index[ip]] = zz[ip];
    }
    double sm = 0.0;
    if (npp1 < m) for (int i = npp1; i < m; ++i) sm += b[i] * b[i];
    *rnorm = sqrt(sm);
#undef A_
    return mode;
}

/* load [D; sqrt(lam) L] (row-major inputs) and [M; 0] into the workspace, solve */
static int o_nnls_aug(nnls_ws *W, const double *D, const double *M, const double *L, double lam,
                      int m, int n, int aug, double *x, double *rnorm)
{
    int mm = aug ? m + n : m;
    for (int j = 0; j < n; ++j) {
        double *col = W->a + (size_t)j * mm;
        for (int i = 0; i < m; ++i) col[i] = D[(size_t)i * n + j];
        if (aug) { double s = sqrt(lam); for (int i = 0; i < n; ++i) col[m + i] = s * L[(size_t)i * n + j]; }
    }
    memcpy(W->b, M, sizeof(double) * m);
    if (aug) memset(W->b + m, 0, sizeof(double) * n);
    return o_nnls_core(W, mm, n, x, rnorm);
}

static double o_sse(const double *D, const double *f, const double *M, int m, int n)
{
    double s = 0.0;
    for (int i = 0; i < m; ++i) {
        double r = 0.0;
        for (int j = 0; j < n; ++j) r += D[(size_t)i * n + j] * f[j];
        r -= M[i];
        s += r * r;
    }
    return s;
}

/* generic NNLS entry (A row-major m x n) */
MET2O_API int met2o_nnls(int m, int n, const double *A, const double *b, double *x, double *rnorm)
{
    nnls_ws *W = ws_new(m, n);
    int mode = o_nnls_aug(W, A, b, NULL, 0.0, m, n, 0, x, rnorm);
    ws_free(W);
    return mode;
}

/* ------------------------------------------------------------------ bounded Brent */
typedef double (*objfn)(double x, void *ctx);

typedef struct { double *xs, *fs; int cap, n; } brent_trace;

static double o_fminbound(objfn fn, void *ctx, double x1, double x2, double xatol, int maxfun,
                          int *nfev, int *flag_out, brent_trace *tr)
{
    const double sqrt_eps = sqrt(2.2e-16);
    const double golden_mean = 0.5 * (3.0 - sqrt(5.0));
    double a = x1, b = x2;
    double fulc = a + golden_mean * (b - a);
    double nfc = fulc, xf = fulc;
    double rat = 0.0, e = 0.0;
    double x = xf;
    double fx = fn(x, ctx);
    if (tr && tr->n < tr->cap) { tr->xs[tr->n] = x; tr->fs[tr->n] = fx; tr->n++; }
    int num = 1, flag = 0;
    double fu = INFINITY;
    double ffulc = fx, fnfc = fx;
    double xm = 0.5 * (a + b);
    double tol1 = sqrt_eps * fabs(xf) + xatol / 3.0;
    double tol2 = 2.0 * tol1;
    while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
        int golden = 1;
        if (fabs(e) > tol1) {
            golden = 0;
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
                    double si = (d > 0.0) - (d < 0.0) + (d == 0.0);
                    rat = tol1 * si;
                }
            } else golden = 1;
        }
        if (golden) {
            if (xf >= xm) e = a - xf; else e = b - xf;
            rat = golden_mean * e;
        }
        double si = (rat > 0.0) - (rat < 0.0) + (rat == 0.0);
        double ar = fabs(rat);
        x = xf + si * (ar > tol1 ? ar : tol1);   /* np.maximum propagates nan; rat is finite here */
        fu = fn(x, ctx);
        num++;
        if (tr && tr->n < tr->cap) { tr->xs[tr->n] = x; tr->fs[tr->n] = fu; tr->n++; }
        if (fu <= fx) {
            if (x >= xf) a = xf; else b = xf;
            fulc = nfc; ffulc = fnfc;
            nfc = xf; fnfc = fx;
            xf = x; fx = fu;
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
    if (nfev) *nfev = num;
    if (flag_out) *flag_out = flag;
    return xf;
}

/* analytic-objective entry for Brent-trace tests: f(x) = sum_k c[k]*|x - r[k]|^p[k] */
typedef struct { int nk; const double *c, *r, *p; } poly_ctx;
static double poly_obj(double x, void *v)
{
    poly_ctx *c = (poly_ctx *)v; double s = 0.0;
    for (int k = 0; k < c->nk; ++k) s += c->c[k] * pow(fabs(x - c->r[k]), c->p[k]);
    return s;
}
MET2O_API double met2o_fminbound_poly(int nk, const double *c, const double *r, const double *p, double x1, double x2,
                                      double xatol, int maxfun, int cap, double *xs, double *fs, int *nfev)
{
    poly_ctx pc = { nk, c, r, p };
    brent_trace tr = { xs, fs, cap, 0 };
    return o_fminbound(poly_obj, &pc, x1, x2, xatol, maxfun, nfev, NULL, &tr);
}

/* ------------------------------------------------------------------ per-voxel context */
typedef struct {
    int m, n;
    const double *D, *L, *M;
    nnls_ws *Wp, *Wa;        /* plain (m x n) and augmented ((m+n) x n) workspaces */
    double *f;               /* scratch solution n */
    int mode_or;             /* OR of nnls modes != 1 */
    /* x2 */
    double SSE, factor;
    /* bayes */
    double *B, *K, *U;       /* n x n each */
    double beta, detL;
    /* gcv scratch */
    double *G, *V, *Dr;      /* k x k, k x k, m x k  (allocated n-sized) */
    int *sup;
} vox_ctx;

static void solve_aug(vox_ctx *c, double lam, double *f, double *rnorm)
{
    int mode = o_nnls_aug(c->Wa, c->D, c->M, c->L, lam, c->m, c->n, 1, f, rnorm);
    if (mode != 1) c->mode_or |= ST_ITMAX;
}
static void solve_plain(vox_ctx *c, double *f, double *rnorm)
{
    int mode = o_nnls_aug(c->Wp, c->D, c->M, NULL, 0.0, c->m, c->n, 0, f, rnorm);
    if (mode != 1) c->mode_or |= ST_ITMAX;
}

/* ---- X1 (algorithms.py:211-233) */
static double obj_x2(double x, void *v)
{
    vox_ctx *c = (vox_ctx *)v; double rn;
    solve_aug(c, x, c->f, &rn);
    double SSEr = o_sse(c->D, c->f, c->M, c->m, c->n);
    return fabs(SSEr - c->factor * c->SSE) / c->SSE;
}
/* The intervals of the three lambda searches: the reference's literals (algorithms.py:219, :280, bayesian_interpolation.py:101) unless a test
 * sets others (met2o_set_intervals: the product takes them from met2_options since ABI 6).  Process-wide; set before a batch, not during one. */
static double g_iv[6] = {0.0, 10.0, 1e-8, 10.0, 1e-8, 2.0};
MET2O_API void met2o_set_intervals(const double *iv6)
{
    static const double dflt[6] = {0.0, 10.0, 1e-8, 10.0, 1e-8, 2.0};
    for (int i = 0; i < 6; ++i) g_iv[i] = iv6 ? iv6[i] : dflt[i];
}
static void o_x2(vox_ctx *c, double factor, double *f, double *lam, double *kest, int *st, brent_trace *tr)
{
    double rn; int flag;
    solve_plain(c, c->f, &rn);
    c->SSE = o_sse(c->D, c->f, c->M, c->m, c->n);
    c->factor = factor;
    *lam = o_fminbound(obj_x2, c, g_iv[0], g_iv[1], 1e-5, 300, NULL, &flag, tr);
    if (flag == 1) *st |= ST_BRENT_MAXFUN;
    solve_aug(c, *lam, f, &rn);
    *kest = o_sse(c->D, f, c->M, c->m, c->n) / c->SSE;
}

/* ---- LC (algorithms.py:88-113,150-206) */
static void o_scale_curve(double *a, int n)
{
    double vmin = a[0], vmax = a[0];
    for (int i = 1; i < n; ++i) { if (a[i] < vmin) vmin = a[i]; if (a[i] > vmax) vmax = a[i]; }
    const double l = -10.0, u = 10.0;
    double s = (u - l) / (vmax - vmin), off = (u * vmin - l * vmax) / (u - l);
    for (int i = 0; i < n; ++i) a[i] = s * (a[i] - off);
}
static int o_select_corner(const double *xin, const double *yin, int n, double *xs, double *ys)
{
    memcpy(xs, xin, sizeof(double) * n); memcpy(ys, yin, sizeof(double) * n);
    o_scale_curve(xs, n); o_scale_curve(ys, n);
    int corner = n - 1;
    const double cte = 7.0 * M_PI / 8.0;
    int have = 0; double angmin = 0.0;
    double c0 = xs[n - 1], c1 = ys[n - 1];
    for (int k = 0; k < n - 2; ++k) {
        double b0 = xs[k], b1 = ys[k];
        for (int j = k + 1; j < n - 1; ++j) {
            double a0 = xs[j], a1 = ys[j];
            double ab = sqrt((a0 - b0) * (a0 - b0) + (a1 - b1) * (a1 - b1));
            double ac = sqrt((a0 - c0) * (a0 - c0) + (a1 - c1) * (a1 - c1));
            double bc = sqrt((b0 - c0) * (b0 - c0) + (b1 - c1) * (b1 - c1));
            double cosa = (ab * ab + ac * ac - bc * bc) / (2.0 * ab * ac);
            double t = (1.0 < cosa) ? 1.0 : cosa;      /* python min(cosa, 1.0): nan stays nan */
            cosa = (t > -1.0) ? t : -1.0;              /* python max(-1.0, t): nan -> -1.0     */
            double ang = acos(cosa);
            double area = 0.5 * ((b0 - a0) * (a1 - c1) - (a0 - c0) * (b1 - a1));
            if (area > 0 && (ang < cte && (!have || ang < angmin))) { corner = j; angmin = ang; have = 1; }
        }
    }
    return corner;
}
MET2O_API int met2o_select_corner(int n, const double *x, const double *y, double *scaled /* 2n */)
{
    return o_select_corner(x, y, n, scaled, scaled + n);
}
static double o_lcurve(vox_ctx *c, const double *lam_grid, int nl, double *logerr, double *lognorm)
{
    int n = c->n;
    double *tmp = (double *)malloc(sizeof(double) * 4 * nl);
    double *le = logerr ? logerr : tmp, *ln = lognorm ? lognorm : tmp + nl;
    for (int i = 0; i < nl; ++i) {
        double rn;
        solve_aug(c, lam_grid[i], c->f, &rn);
        le[i] = log(o_sse(c->D, c->f, c->M, c->m, n) + 1e-200);
        double s = 0.0;
        for (int r = 0; r < n; ++r) { double t = 0.0; for (int j = 0; j < n; ++j) t += c->L[(size_t)r * n + j] * c->f[j]; s += t * t; }
        ln[i] = log(s + 1e-200);
    }
    int corner = o_select_corner(le, ln, nl, tmp + 2 * nl, tmp + 3 * nl);
    free(tmp);
    return lam_grid[corner];
}

/* ---- symmetric-input SVD by one-sided Jacobi (Hestenes): A (k x k) -> U diag(s) V^T.
 * Used for the minimum-norm least-squares solve of algorithms.py:293 (np.linalg.lstsq,
 * rcond=None: singular values below eps*k*s_max are dropped). */
static void o_jacobi_svd(int k, double *A /* in: matrix cols; out: U*s (col-major) */, double *V, double *s)
{
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) V[(size_t)j * k + i] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                double *ap = A + (size_t)p * k, *aq = A + (size_t)q * k;
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < k; ++i) { alpha += ap[i] * ap[i]; beta += aq[i] * aq[i]; gamma += ap[i] * aq[i]; }
                if (gamma == 0.0 || fabs(gamma) <= 1e-15 * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                double *vp = V + (size_t)p * k, *vq = V + (size_t)q * k;
                for (int i = 0; i < k; ++i) {
                    double x = ap[i], y = aq[i]; ap[i] = cs * x - sn * y; aq[i] = sn * x + cs * y;
                    x = vp[i]; y = vq[i]; vp[i] = cs * x - sn * y; vq[i] = sn * x + cs * y;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < k; ++j) { double t = 0; for (int i = 0; i < k; ++i) t += A[(size_t)j * k + i] * A[(size_t)j * k + i]; s[j] = sqrt(t); }
}

/* ---- GC (algorithms.py:276-296) */
static double obj_gcv(double x, void *v)
{
    vox_ctx *c = (vox_ctx *)v; int m = c->m, n = c->n; double SSEr;
    solve_aug(c, x, c->f, &SSEr);
    int k = 0; double ltl = 0.0;
    for (int j = 0; j < n; ++j) if (c->f[j] > 0.0) { c->sup[k++] = j; double d = c->L[(size_t)j * n + j]; ltl += d * d; }
    if (k == 0) return NAN;
    double *G = c->G, *V = c->V, *Dr = c->Dr;
    for (int i = 0; i < m; ++i) for (int a = 0; a < k; ++a) Dr[(size_t)i * k + a] = c->D[(size_t)i * n + c->sup[a]];
    /* DTD + x*LTL: LTL is a scalar (Lr is the vector of selected diagonal entries) broadcast to every element */
    for (int a = 0; a < k; ++a) for (int b = 0; b < k; ++b) {
        double t = 0.0; for (int i = 0; i < m; ++i) t += Dr[(size_t)i * k + a] * Dr[(size_t)i * k + b];
        G[(size_t)b * k + a] = t + x * ltl;
    }
    double *s = (double *)malloc(sizeof(double) * k);
    o_jacobi_svd(k, G, V, s);                      /* G now holds U*s by columns */
    double smax = 0.0; for (int j = 0; j < k; ++j) if (s[j] > smax) smax = s[j];
    double cut = DBL_EPSILON * (double)k * smax;
    /* trace(Dr G^+ Dr^T) = sum_j [s_j>cut] (Dr v_j).(Dr u_j)/s_j, u_j = G[:,j]/s_j */
    double tr = 0.0;
    for (int j = 0; j < k; ++j) {
        if (!(s[j] > cut)) continue;
        double dot = 0.0;
        for (int i = 0; i < m; ++i) {
            double dv = 0.0, du = 0.0;
            for (int a = 0; a < k; ++a) { dv += Dr[(size_t)i * k + a] * V[(size_t)j * k + a]; du += Dr[(size_t)i * k + a] * G[(size_t)j * k + a]; }
            dot += dv * du;
        }
        tr += dot / (s[j] * s[j]);
    }
    free(s);
    double num = (1.0 / m) * (SSEr * SSEr);
    double den = (1.0 / m) * ((double)m - tr);
    return log(num / (den * den));
}
static void o_gcv(vox_ctx *c, double *f, double *lam, int *st, brent_trace *tr)
{
    double rn; int flag;
    *lam = o_fminbound(obj_gcv, c, g_iv[2], g_iv[3], 1e-5, 300, NULL, &flag, tr);
    if (flag == 1) *st |= ST_BRENT_MAXFUN;
    solve_aug(c, *lam, f, &rn);
}

/* ---- BR (bayesian_interpolation.py:84-126) */
static double o_det(int n, const double *Ain)
{
    double *A = (double *)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    double det = 1.0;
    for (int c = 0; c < n; ++c) {
        int p = c; double mx = fabs(A[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(A[(size_t)r * n + c]) > mx) { mx = fabs(A[(size_t)r * n + c]); p = r; }
        if (mx == 0.0) { det = 0.0; break; }
        if (p != c) { for (int j = 0; j < n; ++j) { double t = A[(size_t)c * n + j]; A[(size_t)c * n + j] = A[(size_t)p * n + j]; A[(size_t)p * n + j] = t; } det = -det; }
        double piv = A[(size_t)c * n + c];
        det *= piv;
        for (int r = c + 1; r < n; ++r) {
            double l = A[(size_t)r * n + c] / piv;
            if (l != 0.0) for (int j = c + 1; j < n; ++j) A[(size_t)r * n + j] -= l * A[(size_t)c * n + j];
        }
    }
    free(A);
    return det;
}
static int o_chol_upper(int n, double *A /* row-major, in: full sym; out: U in upper */)
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)k * n + j] * A[(size_t)k * n + j];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double t = A[(size_t)j * n + i];
            for (int k = 0; k < j; ++k) t -= A[(size_t)k * n + j] * A[(size_t)k * n + i];
            A[(size_t)j * n + i] = t / d;
        }
    }
    return 0;
}
static double obj_bayes(double x, void *v)
{
    vox_ctx *c = (vox_ctx *)v; int m = c->m, n = c->n; double rn;
    solve_aug(c, x, c->f, &rn);
    double ED = 0.5 * o_sse(c->D, c->f, c->M, m, n);
    double EW = 0.0;
    for (int r = 0; r < n; ++r) { double t = 0.0; for (int j = 0; j < n; ++j) t += c->L[(size_t)r * n + j] * c->f[j]; EW += t * t; }
    EW *= 0.5;
    double beta = c->beta;
    for (int i = 0; i < n * n; ++i) c->U[i] = beta * c->B[i] + (beta * x) * c->K[i];
    if (o_chol_upper(n, c->U) != 0) { c->mode_or |= ST_CHOLFAIL; return NAN; }
    double det_U = 1.0;
    for (int i = 0; i < n; ++i) det_U *= c->U[(size_t)i * n + i];
    double series = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = i; j < n; ++j) t += c->U[(size_t)i * n + j] * c->f[j];
        series += log(1.0 + erf((1.0 / sqrt(2.0)) * t));
    }
    double cost1 = beta * ED + beta * x * EW + log(det_U) - (n / 2.0) * log(M_PI / 2.0) - series;
    double cost2 = (m / 2.0) * log(2.0 * M_PI) - (m / 2.0) * log(beta) + (n / 2.0) * log(M_PI) - (n / 2.0) * log(2 * beta * x) - log(c->detL);
    return cost1 + cost2;
}
static void bayes_prepare(vox_ctx *c)
{
    int m = c->m, n = c->n; double rn;
    solve_plain(c, c->f, &rn);
    int nnz = 0; for (int j = 0; j < n; ++j) if (c->f[j] > 0.0) nnz++;
    double dof = (double)(m - nnz); if (dof < 1.0) dof = 1.0;
    double sigma = sqrt(o_sse(c->D, c->f, c->M, m, n) / dof);
    c->beta = 1.0 / (sigma * sigma);
}
static void o_bayes(vox_ctx *c, double *f, double *lam, int *st, brent_trace *tr)
{
    double rn; int flag;
    bayes_prepare(c);
    *lam = o_fminbound(obj_bayes, c, g_iv[4], g_iv[5], 1e-5, 200, NULL, &flag, tr);
    if (flag == 1) *st |= ST_BRENT_MAXFUN;
    solve_aug(c, *lam, f, &rn);
}

static vox_ctx *ctx_new(int m, int n, const double *L)
{
    vox_ctx *c = (vox_ctx *)calloc(1, sizeof(vox_ctx));
    c->m = m; c->n = n; c->L = L;
    c->Wp = ws_new(m, n); c->Wa = ws_new(m + n, n);
    c->f = (double *)malloc(sizeof(double) * n);
    c->B = (double *)malloc(sizeof(double) * n * n);
    c->K = (double *)malloc(sizeof(double) * n * n);
    c->U = (double *)malloc(sizeof(double) * n * n);
    c->G = (double *)malloc(sizeof(double) * n * n);
    c->V = (double *)malloc(sizeof(double) * n * n);
    c->Dr = (double *)malloc(sizeof(double) * m * n);
    c->sup = (int *)malloc(sizeof(int) * n);
    if (L) {
        for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) {
            double t = 0.0; for (int i = 0; i < n; ++i) t += L[(size_t)i * n + a] * L[(size_t)i * n + b];
            c->K[(size_t)a * n + b] = t;
        }
        c->detL = o_det(n, L);
    }
    return c;
}
static void ctx_set_D(vox_ctx *c, const double *D)
{
    int m = c->m, n = c->n;
    if (c->D == D) return;
    c->D = D;
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) {
        double t = 0.0; for (int i = 0; i < m; ++i) t += D[(size_t)i * n + a] * D[(size_t)i * n + b];
        c->B[(size_t)a * n + b] = t;
    }
}
static void ctx_free(vox_ctx *c)
{
    ws_free(c->Wp); ws_free(c->Wa);
    free(c->f); free(c->B); free(c->K); free(c->U); free(c->G); free(c->V); free(c->Dr); free(c->sup); free(c);
}

/* ------------------------------------------------------------------ single-problem entry points */
/* method-level functions with the reference's argument meaning; D row-major [m][n] */
MET2O_API int met2o_solve(int method, int m, int n, const double *D, const double *M, const double *L,
                          double param /* lambda for tik, factor for X2 */, const double *lam_grid, int nl,
                          double *f, double *reg /* lambda */, double *extra /* k_est (X2) / rnorm (NNLS) */,
                          int trace_cap, double *trace_x, double *trace_f, int *trace_n)
{
    vox_ctx *c = ctx_new(m, n, L);
    ctx_set_D(c, D); c->M = M;
    int st = ST_FITTED; double rn = 0.0;
    brent_trace tr = { trace_x, trace_f, trace_cap, 0 };
    brent_trace *trp = trace_cap > 0 ? &tr : NULL;
    *reg = 0.0; if (extra) *extra = 0.0;
    switch (method) {
    case M_NNLS: solve_plain(c, f, &rn); if (extra) *extra = rn; break;
    case M_T2SPARC: solve_aug(c, param, f, &rn); *reg = param; if (extra) *extra = rn; break;
    case M_X2: { double k; o_x2(c, param, f, reg, &k, &st, trp); if (extra) *extra = k; } break;
    case M_LCURVE: *reg = o_lcurve(c, lam_grid, nl, trace_cap >= nl ? trace_x : NULL, trace_cap >= nl ? trace_f : NULL);
                   solve_aug(c, *reg, f, &rn); if (trace_n && trace_cap >= nl) tr.n = nl; break;
    case M_GCV: o_gcv(c, f, reg, &st, trp); break;
    case M_BAYES: o_bayes(c, f, reg, &st, trp); break;
    default: st = 0;
    }
    if (trace_n) *trace_n = tr.n;
    st |= c->mode_or;
    ctx_free(c);
    return st;
}

/* objective values on a caller-supplied lambda list (GCV / BayesReg) */
MET2O_API void met2o_objective(int method, int m, int n, const double *D, const double *M, const double *L,
                               int nlam, const double *lams, double *vals)
{
    vox_ctx *c = ctx_new(m, n, L);
    ctx_set_D(c, D); c->M = M;
    if (method == M_BAYES) bayes_prepare(c);
    if (method == M_X2) { double rn; solve_plain(c, c->f, &rn); c->SSE = o_sse(D, c->f, M, m, n); c->factor = 1.02; }
    for (int i = 0; i < nlam; ++i)
        vals[i] = method == M_GCV ? obj_gcv(lams[i], c) : method == M_BAYES ? obj_bayes(lams[i], c) : obj_x2(lams[i], c);
    ctx_free(c);
}

/* ------------------------------------------------------------------ V1: voxel batch */
/* motor:113-162 over a flat voxel list.  D is [nfa][nte][nt2]; data [nvox][nte];
 * fa_index as float64 like the reference (cast with (int)), mask float64. */
/* lam_out (may be NULL): the selected lambda per voxel (equals reg except for X2, where reg holds k_est) */
MET2O_API int met2o_fit_batch_lam(int method, int nte, int nt2, int nfa, const double *Dfa, const double *L,
                              const double *lam_grid, int nl, double x2_factor, double t2sparc_lambda,
                              int64_t nvox, const double *data, const double *fa_index, const double *mask,
                              double *fsol, double *sig, double *reg, double *lam_out, int32_t *status, int nthreads)
{
    int bad = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        vox_ctx *c = ctx_new(nte, nt2, L);
        double *Mn = (double *)malloc(sizeof(double) * nte);
        double *f = (double *)malloc(sizeof(double) * nt2);
#pragma omp for schedule(dynamic, 16)
        for (int64_t v = 0; v < nvox; ++v) {
            const double *M = data + (size_t)v * nte;
            double *fo = fsol + (size_t)v * nt2, *so = sig + (size_t)v * nte;
            memset(fo, 0, sizeof(double) * nt2); memset(so, 0, sizeof(double) * nte);
            reg[v] = 0.0; int st = 0;
            if (lam_out) lam_out[v] = 0.0;
            double sum = 0.0; int finite = 1;
            for (int e = 0; e < nte; ++e) { sum += M[e]; if (!isfinite(M[e])) finite = 0; }
            if (!finite) { st = ST_NONFINITE; if (status) status[v] = st; continue; }
            if (!(mask[v] > 0.0) || !(sum > 0.0) || !(M[0] > 0.0)) { if (status) status[v] = 0; continue; }
            int fi = (int)fa_index[v];
            if (fi < 0 || fi >= nfa) { bad = 1; if (status) status[v] = 0; continue; }
            const double *D = Dfa + (size_t)fi * nte * nt2;
            double km = M[0];
            for (int e = 0; e < nte; ++e) Mn[e] = M[e] / km;
            ctx_set_D(c, D); c->M = Mn; c->mode_or = 0;
            st = ST_FITTED; double rn, r = 0.0, k = 0.0;
            switch (method) {
            case M_NNLS: solve_plain(c, f, &rn); r = 0.0; break;
            case M_T2SPARC: solve_aug(c, t2sparc_lambda, f, &rn); r = t2sparc_lambda; break;
            case M_X2: o_x2(c, x2_factor, f, &r, &k, &st, NULL); if (lam_out) lam_out[v] = r; r = k; break;   /* motor:141-143 stores k_est */
            case M_LCURVE: r = o_lcurve(c, lam_grid, nl, NULL, NULL); solve_aug(c, r, f, &rn); break;
            case M_GCV: o_gcv(c, f, &r, &st, NULL); break;
            case M_BAYES: o_bayes(c, f, &r, &st, NULL); break;
            }
            st |= c->mode_or;
            reg[v] = r;
            if (lam_out && method != M_X2) lam_out[v] = r;
            for (int j = 0; j < nt2; ++j) fo[j] = f[j] * km;
            for (int e = 0; e < nte; ++e) { double t = 0.0; for (int j = 0; j < nt2; ++j) t += D[(size_t)e * nt2 + j] * f[j]; so[e] = t * km; }
            if (status) status[v] = st;
        }
        free(Mn); free(f); ctx_free(c);
    }
    return bad ? -1 : 0;
}

MET2O_API int met2o_fit_batch(int method, int nte, int nt2, int nfa, const double *Dfa, const double *L,
                              const double *lam_grid, int nl, double x2_factor, double t2sparc_lambda,
                              int64_t nvox, const double *data, const double *fa_index, const double *mask,
                              double *fsol, double *sig, double *reg, int32_t *status, int nthreads)
{
    return met2o_fit_batch_lam(method, nte, nt2, nfa, Dfa, L, lam_grid, nl, x2_factor, t2sparc_lambda, nvox, data, fa_index, mask,
                               fsol, sig, reg, NULL, status, nthreads);
}

/* ------------------------------------------------------------------ M1: metrics */
/* motor:443-472.  maps: [6][nvox] = MWF, IEWF, FWF, T2_M, T2_IE, TWC; voxels with mask<=0 stay 0 */
MET2O_API void met2o_metrics(int nt2, const double *T2s, double t2_myelin, double t2_ie, int64_t nvox,
                             const double *fsol, const double *mask, double *maps)
{
    const double epsilon = 1.0e-16;
    for (int64_t v = 0; v < nvox; ++v) {
        for (int q = 0; q < 6; ++q) maps[(size_t)q * nvox + v] = 0.0;
        if (!(mask[v] > 0.0)) continue;
        const double *x = fsol + (size_t)v * nt2;
        double vt = 0.0; for (int j = 0; j < nt2; ++j) vt += x[j];
        vt += epsilon;
        double fm = 0, fie = 0, fcsf = 0, lm = 0, lie = 0;
        for (int j = 0; j < nt2; ++j) {
            double xs = x[j] / vt, t = T2s[j];
            if (t <= t2_myelin) { fm += xs; lm += xs * log(t); }
            if (t > t2_myelin && t <= t2_ie) { fie += xs; lie += xs * log(t); }
            if (t >= t2_ie) fcsf += xs;
        }
        maps[0 * (size_t)nvox + v] = fm;
        maps[1 * (size_t)nvox + v] = fie;
        maps[2 * (size_t)nvox + v] = fcsf;
        maps[3 * (size_t)nvox + v] = exp(lm / (fm + epsilon));
        maps[4 * (size_t)nvox + v] = exp(lie / (fie + epsilon));
        maps[5 * (size_t)nvox + v] = vt;
    }
}

/* ------------------------------------------------------------------ F1: brute-force FA */
/* fa_estimation.py:74-90 per voxel, gating of fa_estimation.py:100.  Outputs: idx (as
 * float64 like FA_index), km = sum(f), sse, f[nt2], resid[nfa] (optional). */
MET2O_API void met2o_fa_bruteforce(int nte, int nt2, int nfa, const double *Dfa, int64_t nvox,
                                   const double *data, const double *mask, double *idx, double *km,
                                   double *sse, double *fout, double *resid_out, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        nnls_ws *W = ws_new(nte, nt2);
        double *f = (double *)malloc(sizeof(double) * nt2);
#pragma omp for schedule(dynamic, 4)
        for (int64_t v = 0; v < nvox; ++v) {
            const double *M = data + (size_t)v * nte;
            idx[v] = 0.0; km[v] = 0.0; if (sse) sse[v] = 0.0;
            if (fout) memset(fout + (size_t)v * nt2, 0, sizeof(double) * nt2);
            double sum = 0.0; for (int e = 0; e < nte; ++e) sum += M[e];
            if (!(mask[v] > 0.0) || !(sum > 0.0)) continue;
            int best = 0; double rbest = 0.0;
            for (int a = 0; a < nfa; ++a) {
                double rn;
                o_nnls_aug(W, Dfa + (size_t)a * nte * nt2, M, NULL, 0.0, nte, nt2, 0, f, &rn);
                if (resid_out) resid_out[(size_t)v * nfa + a] = rn;
                if (a == 0 || rn < rbest) { rbest = rn; best = a; }
            }
            double rn;
            const double *D = Dfa + (size_t)best * nte * nt2;
            o_nnls_aug(W, D, M, NULL, 0.0, nte, nt2, 0, f, &rn);
            double s = 0.0; for (int j = 0; j < nt2; ++j) s += f[j];
            idx[v] = (double)best; km[v] = s;
            if (sse) sse[v] = o_sse(D, f, M, nte, nt2);
            if (fout) memcpy(fout + (size_t)v * nt2, f, sizeof(double) * nt2);
        }
        free(f); ws_free(W);
    }
}

/* ------------------------------------------------------------------ spline FA (fa_estimation.py:35-70) */
/* Not-a-knot cubic spline through (x_i, y_i), i < n (n >= 4), evaluated from its knot slopes (Hermite form).
 * interp1d(kind='cubic') builds the same interpolant in B-spline form; the function is unique. */
static void spline_slopes(int n, const double *x, const double *y, double *s, double *work /* 4n */)
{
    double *dl = work, *dd = work + n, *du = work + 2 * n, *rhs = work + 3 * n;
    double h0 = x[1] - x[0], h1 = x[2] - x[1], d0 = (y[1] - y[0]) / h0, d1 = (y[2] - y[1]) / h1;
    dd[0] = h1; du[0] = h0 + h1; dl[0] = 0.0;
    rhs[0] = ((3.0 * h0 + 2.0 * h1) * h1 * d0 + h0 * h0 * d1) / (h0 + h1);
    for (int i = 1; i < n - 1; ++i) {
        double hm = x[i] - x[i - 1], hp = x[i + 1] - x[i];
        double dm = (y[i] - y[i - 1]) / hm, dp = (y[i + 1] - y[i]) / hp;
        dl[i] = hp; dd[i] = 2.0 * (hm + hp); du[i] = hm;
        rhs[i] = 3.0 * (hp * dm + hm * dp);
    }
    double ha = x[n - 2] - x[n - 3], hb = x[n - 1] - x[n - 2];
    double da = (y[n - 2] - y[n - 3]) / ha, db = (y[n - 1] - y[n - 2]) / hb;
    dl[n - 1] = ha + hb; dd[n - 1] = ha; du[n - 1] = 0.0;
    rhs[n - 1] = (hb * hb * da + (2.0 * ha + 3.0 * hb) * ha * db) / (ha + hb);
    /* the first and last rows couple three unknowns: eliminate to tridiagonal form with dense Gaussian steps */
    /* small n: solve the almost-tridiagonal system by dense elimination with partial pivoting */
    double *A = (double *)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n; ++i) {
        if (i == 0) { A[0] = dd[0]; A[1] = du[0]; }
        else if (i == n - 1) { A[(size_t)i * n + n - 2] = dl[i]; A[(size_t)i * n + n - 1] = dd[i]; }
        else { A[(size_t)i * n + i - 1] = dl[i]; A[(size_t)i * n + i] = dd[i]; A[(size_t)i * n + i + 1] = du[i]; }
        s[i] = rhs[i];
    }
    for (int c = 0; c < n; ++c) {
        int p = c; double mx = fabs(A[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(A[(size_t)r * n + c]) > mx) { mx = fabs(A[(size_t)r * n + c]); p = r; }
        if (p != c) { for (int j = 0; j < n; ++j) { double t = A[(size_t)c * n + j]; A[(size_t)c * n + j] = A[(size_t)p * n + j]; A[(size_t)p * n + j] = t; } double t = s[c]; s[c] = s[p]; s[p] = t; }
        for (int r = c + 1; r < n; ++r) {
            double l = A[(size_t)r * n + c] / A[(size_t)c * n + c];
            if (l != 0.0) { for (int j = c; j < n; ++j) A[(size_t)r * n + j] -= l * A[(size_t)c * n + j]; s[r] -= l * s[c]; }
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        double t = s[c];
        for (int j = c + 1; j < n; ++j) t -= A[(size_t)c * n + j] * s[j];
        s[c] = t / A[(size_t)c * n + c];
    }
    free(A);
}
typedef struct { int n; const double *x, *y, *s; } spline_ctx;
static double spline_eval(double xx, void *v)
{
    spline_ctx *c = (spline_ctx *)v;
    int i = 0;
    while (i < c->n - 2 && xx >= c->x[i + 1]) ++i;
    double h = c->x[i + 1] - c->x[i], t = (xx - c->x[i]) / h;
    double h00 = (1.0 + 2.0 * t) * (1.0 - t) * (1.0 - t), h10 = t * (1.0 - t) * (1.0 - t);
    double h01 = t * t * (3.0 - 2.0 * t), h11 = t * t * (t - 1.0);
    return h00 * c->y[i] + h10 * h * c->s[i] + h01 * c->y[i + 1] + h11 * h * c->s[i + 1];
}
/* slope weights W (n x n): slopes = W y (the system matrix depends on x only) */
MET2O_API void met2o_spline_weights(int n, const double *x, double *W)
{
    double *y = (double *)calloc(n, sizeof(double)), *s = (double *)malloc(sizeof(double) * n), *work = (double *)malloc(sizeof(double) * 4 * n);
    for (int j = 0; j < n; ++j) {
        memset(y, 0, sizeof(double) * n); y[j] = 1.0;
        spline_slopes(n, x, y, s, work);
        for (int i = 0; i < n; ++i) W[(size_t)i * n + j] = s[i];
    }
    free(y); free(s); free(work);
}
/* fa_estimation.py:35-70 per voxel: residual norms of the plain NNLS over the coarse FA grid (Dlr), cubic
 * interpolation, bounded minimisation over [alpha_lr[0], alpha_lr[nlr-1]], snap to the fine grid alpha_hr,
 * km = sum of the NNLS spectrum at the snapped FA (Dhr).  Outputs: idx (float64), km, xmin (optional). */
MET2O_API void met2o_fa_spline(int nte, int nt2, int nlr, const double *Dlr, const double *alpha_lr, int nhr, const double *Dhr,
                               const double *alpha_hr, int64_t nvox, const double *data, const double *mask, double *idx,
                               double *km, double *xmin, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        nnls_ws *W = ws_new(nte, nt2);
        double *f = (double *)malloc(sizeof(double) * nt2);
        double *res = (double *)malloc(sizeof(double) * nlr), *sl = (double *)malloc(sizeof(double) * nlr);
        double *work = (double *)malloc(sizeof(double) * 4 * nlr);
#pragma omp for schedule(dynamic, 4)
        for (int64_t v = 0; v < nvox; ++v) {
            const double *M = data + (size_t)v * nte;
            idx[v] = 0.0; km[v] = 0.0; if (xmin) xmin[v] = 0.0;
            double sum = 0.0; for (int e = 0; e < nte; ++e) sum += M[e];
            if (!(mask[v] > 0.0) || !(sum > 0.0)) continue;
            for (int a = 0; a < nlr; ++a) o_nnls_aug(W, Dlr + (size_t)a * nte * nt2, M, NULL, 0.0, nte, nt2, 0, f, &res[a]);
            spline_slopes(nlr, alpha_lr, res, sl, work);
            spline_ctx sc = { nlr, alpha_lr, res, sl };
            double xs = o_fminbound(spline_eval, &sc, 90.0, 180.0, 1e-5, 500, NULL, NULL, NULL);
            int best = 0; double dbest = fabs(alpha_hr[0] - xs);
            for (int a = 1; a < nhr; ++a) { double d = fabs(alpha_hr[a] - xs); if (d < dbest) { dbest = d; best = a; } }
            double rn;
            o_nnls_aug(W, Dhr + (size_t)best * nte * nt2, M, NULL, 0.0, nte, nt2, 0, f, &rn);
            double s = 0.0; for (int j = 0; j < nt2; ++j) s += f[j];
            idx[v] = (double)best; km[v] = s; if (xmin) xmin[v] = xs;
        }
        free(f); free(res); free(sl); free(work); ws_free(W);
    }
}

/* ------------------------------------------------------------------ NESMA filter (motor:305-333) */
/* numpy's pairwise summation of a contiguous vector (np.sum(..., axis=1) at motor:325): 8 running sums,
 * combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), remainder added in order; blocks of at most 128. */
static double np_pairwise_sum(const double *a, int n)
{
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

/* data [nx][ny][nz][nt] (already multiplied by the mask and clipped, motor:180-182, 279), mask as float64;
 * voxels with mask == 1 get the mean of the window voxels whose relative L1 distance is below 2.5 % */
MET2O_API void met2o_nesma(int nx, int ny, int nz, int nt, const double *data, const double *mask, double *out, int nthreads)
{
    const int hw = 6;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    memset(out, 0, sizeof(double) * (size_t)nx * ny * nz * nt);
#pragma omp parallel
    {
        double *diff = (double *)malloc(sizeof(double) * nt), *acc = (double *)malloc(sizeof(double) * nt);
#pragma omp for schedule(dynamic, 8) collapse(2)
        for (int x = 0; x < nx; ++x)
            for (int y = 0; y < ny; ++y)
                for (int z = 0; z < nz; ++z) {
                    const size_t v = ((size_t)x * ny + y) * nz + z;
                    if (mask[v] != 1.0) continue;
                    const double *c = data + v * nt;
                    const double sumc = np_pairwise_sum(c, nt);
                    const int x0 = x - hw < 0 ? 0 : x - hw, x1 = x + hw > nx ? nx : x + hw;
                    const int y0 = y - hw < 0 ? 0 : y - hw, y1 = y + hw > ny ? ny : y + hw;
                    const int z0 = z - hw < 0 ? 0 : z - hw, z1 = z + hw > nz ? nz : z + hw;
                    for (int e = 0; e < nt; ++e) acc[e] = 0.0;
                    int cnt = 0;
                    for (int i = x0; i < x1; ++i) for (int j = y0; j < y1; ++j) for (int k = z0; k < z1; ++k) {
                        const double *nb = data + (((size_t)i * ny + j) * nz + k) * nt;
                        for (int e = 0; e < nt; ++e) diff[e] = fabs(nb[e] - c[e]);
                        const double RE = 100.0 * np_pairwise_sum(diff, nt) / sumc;
                        if (RE < 2.5) { for (int e = 0; e < nt; ++e) acc[e] += nb[e]; cnt++; }
                    }
                    for (int e = 0; e < nt; ++e) out[v * nt + e] = acc[e] / (double)cnt;      /* mean of an empty set: nan */
                }
        free(diff); free(acc);
    }
}

/* ------------------------------------------------------------------ Gaussian pre-smoothing for the FA step (motor:337-343) */
/* scipy.ndimage 'reflect' extension (d c b a | a b c d | d c b a), any offset */
static int reflect_index(int i, int n)
{
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period; if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

/* data, out [nx][ny][nz][nt]; every echo volume filtered along x, then y, then z with the symmetric kernel w[0..2r]
 * (w[r] the centre), accumulated as scipy's correlate1d does for symmetric kernels: centre first, then the pairs from
 * the outermost inwards. */
MET2O_API void met2o_gaussian_smooth(int nx, int ny, int nz, int nt, int radius, const double *w, const double *data, double *out, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const size_t total = (size_t)nx * ny * nz * nt;
    double *tmp = (double *)malloc(sizeof(double) * total);
    const int dims[3] = {nx, ny, nz};
    const size_t strides[3] = {(size_t)ny * nz * nt, (size_t)nz * nt, (size_t)nt};
    const double *src = data;
    double *dst = out;
    for (int ax = 0; ax < 3; ++ax) {
        const int n = dims[ax];
        const size_t st = strides[ax];
        const size_t nlines = total / (size_t)n;
#pragma omp parallel for schedule(static)
        for (long long line = 0; line < (long long)nlines; ++line) {
            /* decompose the line index into the offset of its first element */
            size_t inner = (size_t)line % st, outer = (size_t)line / st;
            const double *s0 = src + outer * st * (size_t)n + inner;
            double *d0 = dst + outer * st * (size_t)n + inner;
            for (int i = 0; i < n; ++i) {
                double t = s0[(size_t)i * st] * w[radius];
                for (int j = -radius; j < 0; ++j)
                    t += (s0[(size_t)reflect_index(i + j, n) * st] + s0[(size_t)reflect_index(i - j, n) * st]) * w[j + radius];
                d0[(size_t)i * st] = t;
            }
        }
        /* ping-pong: x: data -> out, y: out -> tmp, z: tmp -> out */
        src = dst;
        dst = (dst == out) ? tmp : out;
    }
    free(tmp);
}

MET2O_API int met2o_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
