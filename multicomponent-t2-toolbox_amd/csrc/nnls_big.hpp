// nnls_big.hpp -- the passive-set iteration for sets that outgrow the wave's LDS region ("spill-over" mode).
//
// The fit kernels give every wave an LDS region for a factor of capacity S.kmax < n (as many resident waves as the registers allow:
// 16 per CU at one bin per lane, 8 at two).  Until round 4 a voxel whose set wanted to grow past that capacity was dropped, re-queued and
// solved again from scratch by a second launch at full capacity (2 waves per CU at nT2 = 120) -- a serial tail behind every fit.  Now the
// voxel goes on IN PLACE: columns c >= S.kmax of the factor live in a per-wave slot in global memory (S.Rg, L2-resident: written and read
// by this wave only), columns below stay in LDS, and the routines below address an entry through a generic pointer whose aperture decides
// (flat loads and stores).  Where a column lives is a function of its index alone, so the fast routines of nnls_wave.hpp are valid again
// as soon as a later solve starts with k <= S.kmax; only the evaluations that really hold a large set pay the global round trips
// (X2: the first two or three Brent abscissae of 5-10 % of the voxels at 48 x 120; the L-curve: the last grid points).
//
// The arithmetic is that of the fast routines (same sums in the same order); the code is the plain form of each -- no one-slot legs,
// no pairing of LDS reads -- because it runs for a few evaluations of a few voxels.  None of it is compiled into the fit kernels' own
// voxel loop (template argument BIG = false there: the round-4 code token for token, a set at the capacity flags the voxel); the voxel
// is then solved again, at once and by the same wave, by a NOT-inlined instance of the voxel routine with BIG = true (fit_kernel.hpp:
// fit_voxel_spill) -- one call site per kernel, so the hot loop's register allocation does not see this code.  Stores to the factor are followed by a
// workgroup-scope fence before other lanes read them (big_sync): the slot is private to the wave, the fence orders the wave's own
// stores and loads.
#pragma once

namespace met2 {

#ifdef MET2_BIGSTATS
// development: [0] solver calls of the spill-over voxels, [1] of them with spill-over legs, [2] cycles of the plain calls, [3] of the spill-over calls,
// [4] appends, [5] removals, [6] re-factorisations in the slot, [7] sum of k at the end of the spill-over calls, [8] sum of k at their start, [9] cycles of the re-factorisations
__device__ unsigned long long g_bigstats[16];
#define MET2_BIGSTAT(i, v) do { if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd(&g_bigstats[i], (unsigned long long)(v)); } while (0)
#else
#define MET2_BIGSTAT(i, v)
#endif
__device__ __forceinline__ double *big_col(const WaveShared &S, int c)
{
    const int cb = col_base(c);
    return (c < S.kmax) ? (S.R + cb) : (S.Rg + (cb - S.gbase));
}

__device__ __forceinline__ void big_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// back substitution R z = y ; z position-indexed (back_subst)
template <int NB>
__device__ __forceinline__ void back_subst_big(const WaveShared &S, const NnlsState<NB> &st, int lane, double (&z)[NB])
{
    const int k = st.k;
    double y[NB], ra[NB], rb[NB];
    {
        const double *cp = big_col(S, max(k - 1, 0));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int pl = lane + 64 * b;
            y[b] = st.y[b];
            ra[b] = (k > 0 && pl < k - 1) ? cp[pl] : 0.0;                  // column k-1
            rb[b] = 0.0;
        }
    }
    for (int c = k - 1; c >= 1; --c) {
        const double *cn = big_col(S, c - 1);
#pragma unroll
        for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; rb[b] = (pl < c - 1) ? cn[pl] : 0.0; }       // column c-1, in flight under this step
        const double t = (NB == 2 && (c >> 6)) ? y[NB - 1] * st.rinv[NB - 1] : y[0] * st.rinv[0];
        const double s = bcast(t, c & 63);
#pragma unroll
        for (int b = 0; b < NB; ++b) { y[b] = fma(-ra[b], s, y[b]); ra[b] = rb[b]; }                               // positions >= c are final
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) z[b] = (lane + 64 * b < k) ? y[b] * st.rinv[b] : 0.0;
}

// Remove position p from the passive set (remove_pos): delete column p of R and re-triangularise.
template <int NB>
__device__ __forceinline__ void remove_pos_big(const WaveShared &S, NnlsState<NB> &st, int p, int lane)
{
    const int k = st.k;
    const int tb = bcastN_i<NB>(st.ord, p);
    MET2_BIGSTAT(5, 1);
    if (p < k - 1) {
        double *cpl[NB], *cpm[NB];                                          // the lane's own (old) column, and the one before it
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int pl = lane + 64 * b;
            cpl[b] = big_col(S, min(pl, S.n - 1));
            cpm[b] = big_col(S, min(max(pl - 1, 0), S.n - 1));
        }
        // rows above p: column c+1 moves into column c, one column at a time (lane <-> row)
        for (int c = p; c <= k - 2; ++c) {
            const double *src = big_col(S, c + 1);
            double *dst = big_col(S, c);
            double v[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; v[b] = (pl < p) ? src[pl] : 0.0; }
#pragma unroll
            for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; if (pl < p) dst[pl] = v[b]; }
        }
        // rows p..k-1: chain of plane rotations, owned index = old column index
        double carry[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int pl = lane + 64 * b;
            carry[b] = (pl > p && pl < k) ? cpl[b][p] : 0.0;
        }
        double ycar = bcastN<NB>(st.y, p);
        for (int j = p + 1; j < k; ++j) {
            double rowj[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int pl = lane + 64 * b;
                rowj[b] = (pl >= j && pl < k) ? cpl[b][j] : 0.0;
            }
            double a = bcastN<NB>(carry, j), bb = bcastN<NB>(rowj, j);
            double c, s, sig, sinv;
            givens(a, bb, c, s, sig, sinv);
            double yj = bcastN<NB>(st.y, j);
            double ynew = c * ycar + s * yj;
            ycar = -s * ycar + c * yj;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int pl = lane + 64 * b;
                double nv = c * carry[b] + s * rowj[b];
                carry[b] = -s * carry[b] + c * rowj[b];
                if (pl > j && pl < k) cpm[b][j - 1] = nv;                    // new entry (j-1, pl-1)
                if (pl == j) cpm[b][j - 1] = sig;
                if (pl == j - 1) { st.y[b] = ynew; st.rinv[b] = sinv; }
            }
        }
        big_sync();
    }
    int ordn[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) ordn[b] = gatherN_i<NB>(st.ord, lane + 64 * b + 1);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        st.ord[b] = (pl >= p) ? ordn[b] : st.ord[b];                          // by position
        st.pos[b] = (st.pos[b] > p) ? st.pos[b] - 1 : st.pos[b];              // by bin
        if (pl == tb) { st.pos[b] = -1; st.x[b] = 0.0; }
    }
    clear_bit<NB>(st.P, tb);
    st.k = k - 1;
}

// Try to move bin t from Z to P (try_append).
template <int NB>
__device__ __forceinline__ bool try_append_big(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int t, int lane,
                                               bool forced = false)
{
    const int k = st.k;
    double gb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = lane + 64 * b;
        const unsigned je = (unsigned)min(j, S.n - 1);
        const double bv = ld_row_sel(S.buffer_rows, S.B, t * S.bstride, je);
        gb[b] = (j < S.n) ? bv : 0.0;
        if (lam != 0.0) gb[b] = fma(lam, (j < S.n) ? ld_row_sel(S.buffer_rows, S.K, t * S.n, je) : 0.0, gb[b]);
    }
    const double gtt = bcastN<NB>(gb, t);
    double g[NB], rv[NB], rw[NB];
    const double *cpl[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        cpl[b] = big_col(S, min(pl, S.n - 1));
        double gg = gatherN<NB>(gb, st.ord[b]);                          // position-indexed G[ord_p][t]
        g[b] = (pl < k) ? gg : 0.0;
        rv[b] = (k > 0 && pl > 0 && pl < k) ? cpl[b][0] : 0.0;           // row 0
        rw[b] = 0.0;
    }
    // forward substitution R^T r = g: a lane walks down its own column, the next row in flight under the current step
    for (int i = 0; i + 1 < k; ++i) {
#pragma unroll
        for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; rw[b] = (pl > i + 1 && pl < k) ? cpl[b][i + 1] : 0.0; }
        const double tt = (NB == 2 && (i >> 6)) ? g[NB - 1] * st.rinv[NB - 1] : g[0] * st.rinv[0];
        const double s = bcast(tt, i & 63);
#pragma unroll
        for (int b = 0; b < NB; ++b) { g[b] = fma(-rv[b], s, g[b]); rv[b] = rw[b]; }      // positions <= i are final
    }
    double r[NB], rr = 0.0, ry = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        r[b] = (lane + 64 * b < k) ? g[b] * st.rinv[b] : 0.0;
        rr = fma(r[b], r[b], rr);
        ry = fma(r[b], st.y[b], ry);
    }
    wave_sum2(rr, ry);
    const double rho2 = gtt - rr;
    if (!(rho2 > 1e-14 * gtt)) return false;                     // dependent column
    const double rhoinv = rsqrt_nr(rho2), rho = rho2 * rhoinv;
    const double ynew = (bcastN<NB>(st.h, t) - ry) * rhoinv;
    if (!forced && !(ynew > 0.0)) return false;
    MET2_BIGSTAT(4, 1);
    double *ck = big_col(S, k);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        if (pl < k) ck[pl] = r[b];
        if (pl == k) { ck[k] = rho; st.rinv[b] = rhoinv; st.y[b] = ynew; st.ord[b] = t; }
        if (pl == t) st.pos[b] = k;
    }
    big_sync();
    set_bit<NB>(st.P, t);
    st.k = k + 1;
    return true;
}

// Lawson-Hanson's secondary loop (nnls_inner)
template <int NB>
__device__ __forceinline__ bool nnls_inner_big(const WaveShared &S, NnlsState<NB> &st, int &iter, int itmax, int lane)
{
    for (;;) {
        if (++iter > itmax) return false;
        double z[NB], xp[NB], zb[NB], ratio[NB];
        bool neg[NB];
        back_subst_big<NB>(S, st, lane, z);
        bool any = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int pl = lane + 64 * b;
            xp[b] = gatherN<NB>(st.x, st.ord[b]);
            neg[b] = (pl < st.k) && (z[b] <= 0.0);
            any = any || (ballot(neg[b]) != 0ull);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ps = st.pos[b] < 0 ? 0 : st.pos[b];
            double t = gatherN<NB>(z, ps);
            zb[b] = (st.pos[b] >= 0) ? t : 0.0;
        }
        if (!any) {
#pragma unroll
            for (int b = 0; b < NB; ++b) st.x[b] = zb[b];
            return true;
        }
        double rmin = 2.0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double r = neg[b] ? xp[b] / (xp[b] - z[b]) : 2.0;
            ratio[b] = (r == r) ? r : 2.0;
            rmin = fmin(rmin, ratio[b]);
        }
        const double alpha = wave_min(rmin);
        if (!(alpha < 2.0)) {
#pragma unroll
            for (int b = 0; b < NB; ++b) st.x[b] = zb[b];
            return true;
        }
        u64 hit[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) hit[b] = ballot(neg[b] && ratio[b] == alpha);
        const int jj = first_bit<NB>(hit);
#pragma unroll
        for (int b = 0; b < NB; ++b) st.x[b] = (st.pos[b] >= 0) ? fma(alpha, zb[b] - st.x[b], st.x[b]) : 0.0;
        remove_pos_big<NB>(S, st, jj, lane);
        for (int sweep = 0; sweep < 64 * NB; ++sweep) {
            u64 bad[NB];
            bool anyb = false;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                double xq = gatherN<NB>(st.x, st.ord[b]);
                bad[b] = ballot((lane + 64 * b < st.k) && (xq <= 0.0));
                anyb = anyb || (bad[b] != 0ull);
            }
            if (!anyb) break;
            remove_pos_big<NB>(S, st, first_bit<NB>(bad), lane);
        }
    }
}

// The passive-set iteration from wherever the fast legs left it (iterate_leg); the capacity is the grid itself.
template <int NB>
__device__ __forceinline__ void iterate_big(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int mrows, int lane,
                                            bool &warm, int &iter, int &outer)
{
    const int n = S.n, itmax = 3 * n;
    st.itmax_hit |= 4;                                     // (this voxel used the spill-over slot: counted in the plan's statistics)
    if (warm && st.k > 0) {
        warm = false;
        if (!nnls_inner_big<NB>(S, st, iter, itmax, lane)) { st.itmax_hit |= 1; return; }
    }
    warm = false;
    for (; outer <= itmax + 1; ++outer) {
        if (st.k >= n || st.k >= mrows) break;
        double w[NB];
        dual<NB>(S, bd, st, lam, lane, w);
        u64 rejected[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) rejected[b] = 0ull;
        bool accepted = false;
        for (int tries = 0; tries < 64 * NB; ++tries) {
            bool cand[NB];
            double val[NB], vmax = -1.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                cand[b] = (lane + 64 * b < n) && !((st.P[b] >> lane) & 1ull) && !((rejected[b] >> lane) & 1ull);
                val[b] = cand[b] ? w[b] : -1.0;
                vmax = fmax(vmax, val[b]);
            }
            const double wmax = wave_max(vmax);
            if (!(wmax > 0.0)) break;
            u64 hit[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) hit[b] = ballot(cand[b] && val[b] == wmax);
            const int t = first_bit<NB>(hit);
            if (try_append_big<NB>(S, bd, st, lam, t, lane)) { accepted = true; break; }
            set_bit<NB>(rejected, t);
        }
        if (!accepted) break;
        if (!nnls_inner_big<NB>(S, st, iter, itmax, lane)) { st.itmax_hit |= 1; break; }
    }
}

// Rebuild R, 1/diag and y for the current passive set and pivot order at a new lambda, row by row (refactor_rowwise): one bin per lane.
template <int NB>
__device__ __forceinline__ bool refactor_rowwise_big(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int lane)
{
    const int k = st.k, n = S.n;
    double *cpl[NB];
    unsigned jc[NB];
    double g[NB], gb0[NB], gk0[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        jc[b] = (unsigned)min(pl, n - 1);
        cpl[b] = big_col(S, min(pl, n - 1));                             // (reads of it are predicated: a run past a short column's end would
                                                                         //  leave the wave's LDS region where the columns continue in the slot)
        const double hh = gatherN<NB>(st.h, st.ord[b]);
        g[b] = (pl < k) ? hh : 0.0;
    }
    auto fetch = [&](int p, double (&vb)[NB], double (&vk)[NB]) {
        const int t = bcastN_i<NB>(st.ord, min(p, k - 1));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            vb[b] = ld_row_sel(S.buffer_rows, S.B, t * S.bstride, jc[b]);
            vk[b] = ld_row_sel(S.buffer_rows, S.K, t * n, jc[b]);
        }
    };
    double gdp[NB];
    {
        double gdb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double bd0 = S.have_bdiag ? S.bdiag[b] : S.B[jc[b] * S.bstride + jc[b]];
            gdb[b] = (lam != 0.0) ? fma(lam, S.K[jc[b] * n + jc[b]], bd0) : bd0;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) gdp[b] = gatherN<NB>(gdb, st.ord[b]);
    }
    fetch(0, gb0, gk0);
    for (int i = 0; i < k; ++i) {
        double a[NB], a2[NB];
        {
            double t0[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) t0[b] = fma(lam, gk0[b], gb0[b]);
            fetch(i + 1, gb0, gk0);
#pragma unroll
            for (int b = 0; b < NB; ++b) { a[b] = gatherN<NB>(t0, st.ord[b]); a2[b] = 0.0; }
        }
        const double *ci = big_col(S, i);
        int j = 0;
#pragma clang loop unroll(disable)
        for (; j + 4 <= i; j += 4) {
            const double s0 = ci[j], s1 = ci[j + 1], s2 = ci[j + 2], s3 = ci[j + 3];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int pl = lane + 64 * b;
                double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
                if (pl >= i && pl < k) { q0 = cpl[b][j]; q1 = cpl[b][j + 1]; q2 = cpl[b][j + 2]; q3 = cpl[b][j + 3]; }
                a[b] = fma(-s0, q0, a[b]); a2[b] = fma(-s1, q1, a2[b]);
                a[b] = fma(-s2, q2, a[b]); a2[b] = fma(-s3, q3, a2[b]);
            }
        }
        for (; j < i; ++j) {
            const double s0 = ci[j];
#pragma unroll
            for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; const double q0 = (pl >= i && pl < k) ? cpl[b][j] : 0.0; a[b] = fma(-s0, q0, a[b]); }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) a[b] += a2[b];
        const double d = bcastN<NB>(a, i);
        const double rinv = rsqrt_nr(d);
        const double yi = bcastN<NB>(g, i) * rinv;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int pl = lane + 64 * b;
            const double r = a[b] * rinv;
            if (pl >= i && pl < k) cpl[b][i] = r;
            g[b] = (pl > i) ? fma(-r, yi, g[b]) : g[b];
        }
        big_sync();
    }
    bool bad = false;
    double dgl[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        dgl[b] = (pl < k) ? cpl[b][pl] : 1.0;
        bad = bad || (ballot((pl < k) && !(dgl[b] * dgl[b] > 1e-14 * gdp[b])) != 0ull);
    }
    if (bad) return false;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        const double ri = rcp_nr(dgl[b]);
        st.rinv[b] = (pl < k) ? ri : st.rinv[b];
        st.y[b] = (pl < k) ? g[b] * ri : st.y[b];
    }
    return true;
}

// The same, blocked, with the trailing updates on the matrix cores (refactor_blocked): two bins per lane.
template <int NB>
__device__ __forceinline__ bool refactor_blocked_big(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int lane)
{
    const int k = st.k, n = S.n;
    double *cpl[NB];
    unsigned jc[NB];
    double g[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        cpl[b] = big_col(S, min(pl, n - 1));                             // (reads predicated, as above)
        jc[b] = (unsigned)min(pl, n - 1);
        const double hh = gatherN<NB>(st.h, st.ord[b]);
        g[b] = (pl < k) ? hh : 0.0;
    }
    double gdp[NB];
    {
        double gdb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double bd0 = S.have_bdiag ? S.bdiag[b] : S.B[jc[b] * S.bstride + jc[b]];
            gdb[b] = (lam != 0.0) ? fma(lam, S.K[jc[b] * n + jc[b]], bd0) : bd0;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) gdp[b] = gatherN<NB>(gdb, st.ord[b]);
    }
    // ---- (0) A into the factor's storage
#pragma clang loop unroll(disable)
    for (int i = 0; i < k; i += 4) {
        double vb[4][NB], vk[4][NB];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = bcastN_i<NB>(st.ord, i + q);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                vb[q][b] = ld_row_sel(S.buffer_rows, S.B, t * S.bstride, jc[b]);
                vk[q][b] = (lam != 0.0) ? ld_row_sel(S.buffer_rows, S.K, t * n, jc[b]) : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double t0[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) t0[b] = fma(lam, vk[q][b], vb[q][b]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int pl = lane + 64 * b;
                const double av = gatherN<NB>(t0, st.ord[b]);
                if (i + q < k && pl >= i + q && pl < k) cpl[b][i + q] = av;
            }
        }
    }
    big_sync();
    auto finish = [&](int i, double (&a)[NB], double (&r)[NB]) {
        const double d = bcastN<NB>(a, i);
        const double rinv = rsqrt_nr(d);
        const double yi = bcastN<NB>(g, i) * rinv;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int pl = lane + 64 * b;
            r[b] = a[b] * rinv;
            if (pl >= i && pl < k) cpl[b][i] = r[b];
            g[b] = (pl > i) ? fma(-r[b], yi, g[b]) : g[b];
        }
    };
    const int li = lane & 15, lk = lane >> 4;
    const int nt = (k + 15) >> 4;
    for (int kb = 0; kb < nt; ++kb) {
        const int r0 = 16 * kb, r1 = min(k, r0 + 16);
        int i = r0;
        for (; i + 1 < r1; i += 2) {
            double a[NB], c[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int pl = lane + 64 * b;
                a[b] = (pl >= i && pl < k) ? cpl[b][i] : 0.0;
                c[b] = (pl > i && pl < k) ? cpl[b][i + 1] : 0.0;
            }
            const double *ci = big_col(S, i), *cj = big_col(S, i + 1);
            int j = r0;
#pragma clang loop unroll(disable)
            for (; j + 4 <= i; j += 4) {
                const double s0 = ci[j], s1 = ci[j + 1], s2 = ci[j + 2], s3 = ci[j + 3];
                const double u0 = cj[j], u1 = cj[j + 1], u2 = cj[j + 2], u3 = cj[j + 3];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int pl = lane + 64 * b;
                    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
                    if (pl >= i && pl < k) { q0 = cpl[b][j]; q1 = cpl[b][j + 1]; q2 = cpl[b][j + 2]; q3 = cpl[b][j + 3]; }
                    a[b] = fma(-s0, q0, a[b]); c[b] = fma(-u0, q0, c[b]);
                    a[b] = fma(-s1, q1, a[b]); c[b] = fma(-u1, q1, c[b]);
                    a[b] = fma(-s2, q2, a[b]); c[b] = fma(-u2, q2, c[b]);
                    a[b] = fma(-s3, q3, a[b]); c[b] = fma(-u3, q3, c[b]);
                }
            }
            if (j < i) {
                const double s0 = ci[j], s1 = ci[j + 1], u0 = cj[j], u1 = cj[j + 1];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int pl = lane + 64 * b;
                    double q0 = 0.0, q1 = 0.0;
                    if (pl >= i && pl < k) { q0 = cpl[b][j]; q1 = cpl[b][j + 1]; }
                    a[b] = fma(-s0, q0, a[b]); c[b] = fma(-u0, q0, c[b]);
                    a[b] = fma(-s1, q1, a[b]); c[b] = fma(-u1, q1, c[b]);
                }
            }
            double r[NB], r2[NB];
            finish(i, a, r);
            const double sr = bcastN<NB>(r, i + 1);
#pragma unroll
            for (int b = 0; b < NB; ++b) c[b] = fma(-sr, r[b], c[b]);
            finish(i + 1, c, r2);
            big_sync();
        }
        if (i < r1) {
            double a[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; a[b] = (pl >= i && pl < k) ? cpl[b][i] : 0.0; }
            const double *ci = big_col(S, i);
            int j = r0;
#pragma clang loop unroll(disable)
            for (; j + 2 <= i; j += 2) {
                const double s0 = ci[j], s1 = ci[j + 1];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int pl = lane + 64 * b;
                    double q0 = 0.0, q1 = 0.0;
                    if (pl >= i && pl < k) { q0 = cpl[b][j]; q1 = cpl[b][j + 1]; }
                    a[b] = fma(-s0, q0, a[b]);
                    a[b] = fma(-s1, q1, a[b]);
                }
            }
            double r[NB];
            finish(i, a, r);
            big_sync();
        }
        // ---- trailing tiles on the matrix cores
        for (int ti = kb + 1; ti < nt; ++ti) {
            const int ca = 16 * ti + li;
            const double *pa = big_col(S, min(ca, k - 1));
            double aop[4];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const double v = pa[r0 + 4 * sidx + lk];
                aop[sidx] = (ca < k) ? -v : 0.0;
            }
            for (int tj = ti; tj < nt; ++tj) {
                const int cc = 16 * tj + li;
                double *pb = big_col(S, min(cc, k - 1));
                double bop[4];
                met2_d4 acc;
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const double v = pb[r0 + 4 * sidx + lk];
                    bop[sidx] = (cc < k) ? v : 0.0;
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * ti + lk + 4 * v;
                    acc[v] = (cc < k && row <= cc) ? pb[row] : 0.0;
                }
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[sidx], bop[sidx], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * ti + lk + 4 * v;
                    if (cc < k && row <= cc) pb[row] = acc[v];
                }
            }
        }
        big_sync();
    }
    bool bad = false;
    double dgl[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        dgl[b] = (pl < k) ? cpl[b][pl] : 1.0;
        bad = bad || (ballot((pl < k) && !(dgl[b] * dgl[b] > 1e-14 * gdp[b])) != 0ull);
    }
    if (bad) return false;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        const double ri = rcp_nr(dgl[b]);
        st.rinv[b] = (pl < k) ? ri : st.rinv[b];
        st.y[b] = (pl < k) ? g[b] * ri : st.y[b];
    }
    return true;
}

template <int NB>
__device__ __forceinline__ bool refactor_big(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int lane)
{
    MET2_BIGSTAT(6, 1);
#ifdef MET2_BIGSTATS
    const unsigned long long t0 = __builtin_readcyclecounter();
    const bool ok = (NB >= 2) ? refactor_blocked_big<NB>(S, bd, st, lam, lane) : refactor_rowwise_big<NB>(S, bd, st, lam, lane);
    MET2_BIGSTAT(9, __builtin_readcyclecounter() - t0);
    return ok;
#else
    if (NB >= 2) return refactor_blocked_big<NB>(S, bd, st, lam, lane);
    return refactor_rowwise_big<NB>(S, bd, st, lam, lane);
#endif
}

// ---- helpers for the not-inlined voxel routine that runs this code (fit_kernel.hpp: fit_voxel_spill).  Arguments of a device function
// arrive in vector registers: what is wave-uniform is made scalar again on entry (readfirstlane), so that loops, broadcasts and LDS
// addresses stay scalar in there.
__device__ __forceinline__ int big_rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 big_rfl(u64 v) { return ((u64)(unsigned)big_rfl((int)(v >> 32)) << 32) | (u64)(unsigned)big_rfl((int)v); }
template <class T>
__device__ __forceinline__ T *big_rfl(T *p) { return (T *)big_rfl((u64)p); }
// The not-inlined routines take the wave's LDS region as an LDS-typed pointer argument (a generic pointer would turn every ds_ access of the fast
// legs into a flat one) and find their lane without the work-item id: a callee that needs no implicit argument (work-item / workgroup ids, the
// dynamic-LDS base) leaves the calling kernel free of the registers it would have to keep for it.
typedef __attribute__((address_space(3))) double *big_lds_dp;
__device__ __forceinline__ int big_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

} // namespace met2
