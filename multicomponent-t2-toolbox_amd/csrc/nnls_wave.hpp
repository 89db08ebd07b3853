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
// Mapping: a lane owns NB T2 bins, bin = lane + 64*b (NB = 1: nT2 <= 64, NB = 2: nT2 <= 128).
// "Position" p is the order in which bins joined the passive set; lane p&63, slot p>>6 also
// holds the position-indexed quantities (ord[p] = bin at position p, y[p] = (R^-T h_P)[p],
// 1/R[p][p]).
//
// Per wave in LDS: R, upper triangular, packed by COLUMNS (entry (r, c), r <= c, at c(c+1)/2 + r), kmax(kmax+1)/2
// doubles.  A lane's walk down its own column is then a run of consecutive addresses (ds_read2 with immediate
// offsets in the refactorisation and the substitutions) and lanes of neighbouring columns never share a bank
// (triangular numbers are distinct mod 32).
// B (n x n), K, D (m x n) and D^T of the voxel's flip angle are read from global memory through L1/L2 (S.B, S.K, S.D, S.Dt).
#pragma once
#include "wave_ops.hpp"

namespace met2 {

// -DMET2_DOUBLE=<phase> (development): an idempotent phase is executed twice -- 1 refactorisation, 2 dual, 3 model signal,
// 4 back substitution -- so that the difference in kernel time against the plain build is that phase's MARGINAL cost (the share of
// the wave cycles a phase takes says little when four waves per SIMD fill each other's waits)
#ifndef MET2_DOUBLE
#define MET2_DOUBLE 0
#endif
#ifndef MET2_REORDER
#define MET2_REORDER 2        // bins per lane from which warm starts re-order the passive set by descending x (reorder_by_x below); 1: always, 3: never
#endif

#ifdef MET2_LOOPSTATS
__device__ int g_loopstats[8];
#define MET2_STAT(slot, v) atomicMax(&g_loopstats[slot], (int)(v))
#else
#define MET2_STAT(slot, v)
#endif

#ifdef MET2_CYCSTATS
__device__ unsigned long long g_cyc[16];
// per Brent-evaluation index (X2 kernel): [0] evaluations, [1] refactor, [2] inner, [3] dual, [4] append, [5] sse cycles, [6] removals,
// [7] removal cycles, [8] sum of k at the start, [9] sum of k at the end, [10] evaluations on the canonical (voxel-independent) abscissa
__device__ unsigned long long g_ev[40][12];
#define MET2_CYC_BEGIN(var) const unsigned long long var = __builtin_readcyclecounter()
#define MET2_CYC_END(slot, var) st.cyc[slot] += __builtin_readcyclecounter() - var
#define MET2_CYC_ADD(slot, v) st.cyc[slot] += (unsigned long long)(v)
#define MET2_CYC_INIT(st) do { for (int q_ = 0; q_ < 16; ++q_) st.cyc[q_] = 0; } while (0)
#define MET2_CYC_FLUSH(st) do { if (lane_id() == 0) for (int q_ = 0; q_ < 16; ++q_) atomicAdd(&g_cyc[q_], st.cyc[q_]); } while (0)
#else
#define MET2_CYC_BEGIN(var)
#define MET2_CYC_END(slot, var)
#define MET2_CYC_ADD(slot, v)
#define MET2_CYC_INIT(st)
#define MET2_CYC_FLUSH(st)
#endif

struct WaveShared {
    const double *B;    // [n][bstride]
    const double *K;    // [n][n] dense L^T L in global memory (fit kernel only)
    const double *Dt;   // [n][dtstride] transposed D in global memory
    const double *DtG;  // the same array (the GCV Gram contraction gathers its operands from it)
    const double *kband; // [5][128] diagonals of K in global memory: kband[d*128 + j] = K[j][j+d-2] (loaded where the stencil is applied)
    int dtstride;
    const double *D;    // [m][dstride]
    double *R;          // this wave's LDS region
    int n, m, bstride, dstride, kmax;
    int rcap;           // doubles available at R
    bool have_bdiag;    // bdiag holds B[j][j] of the lane's bins (the FA walk sets it once per flip angle: its refactorisations are too
    double bdiag[2];    // short -- k ~ 8 -- to hide the latency of loading the diagonal inside refactor())
    bool reorder;       // warm starts re-order the pivots by descending x (reorder_by_x) -- off in the FA walk, where neighbouring flip angles
                        // keep the passive set and the ranking is pure overhead (FA walk of configs[4] on 131 072 voxels: 45.0 -> 42.1 ms)
    bool buffer_rows;   // row loads of the global matrices as raw buffer loads (fit kernels); false: plain global loads (the FA walk, whose
                        // 19 MB of dictionaries at 48 x 120 missed L2 1.7x more often through the buffer path: 57 -> 66 ms per 131 k voxels)
    double *Rg = nullptr; // this wave's spill-over slot in global memory: columns c >= kmax of the factor, entry (r, c) at col_base(c) - gbase + r
    int gbase = 0;        // col_base(kmax)   (nnls_big.hpp; used by the BIG = true instances of the routines below only)
};

// L as 5 diagonals per owned bin: lb[b][d] = L[j][j+d-2].  (K = L^T L is not held in registers: its rows come from the
// dense copy S.K, its diagonals for the 5-point stencil of the dual from S.kband -- 10 VGPRs less to carry through
// the whole voxel loop.)
template <int NB>
struct Band {
    double lb[NB][5];
};

template <int NB>
struct NnlsState {
    double h[NB];      // (D^T b)_j                     bin-indexed
    double x[NB];      // current iterate               bin-indexed
    int pos[NB];       // position of bin j, -1 if in Z bin-indexed
    double y[NB];      // R^-T h_P                      position-indexed
    double rinv[NB];   // 1 / R[p][p]                   position-indexed
    int ord[NB];       // bin at position p             position-indexed
    int k;             // |P|                (uniform)
    u64 P[NB];         // passive-set mask   (uniform)
    int itmax_hit;     // uniform flags: bit0 iteration cap reached, bit1 passive set hit the capacity kmax < n
#ifdef MET2_CYCSTATS
    unsigned long long cyc[16];
#endif
};

__device__ __forceinline__ int row_base(int i, int kmax) { return i * kmax - (i * (i - 1)) / 2 - i; } // row-packed factors of objectives.hpp: entry (i,c) at row_base + c
// the solver's factor: entry (r,c) at col_base(c) + r, packed by columns without padding.  Lane c owns column c, and the triangular
// numbers c (c + 1) / 2 are distinct mod 32 for 32 consecutive c: a wave-wide ds_read_b64 / ds_write_b64 of one row of all columns
// is bank-conflict-free (2 LDS-array cycles per half wave).  Columns padded to 16 bytes for ds_read_b128 (tried first: 162.5 ->
// 152.9 ms on configs[1]) lose exactly that: SQ_LDS_BANK_CONFLICT rose to 20 % of the LDS cycles of the X2 kernel and 44 % of
// BayesReg's, modelled 6-13 cycles per b128 instead of 4.  What the padding was for -- four consecutive rows in 8 cycles instead
// of the 16 of two ds_read2_b64 -- comes from four SINGLE ds_read_b64 (2 cycles each): the loads are volatile (and typed as LDS
// pointers, so they stay ds_ instructions) so that the compiler does not pair them into ds_read2_b64.
__host__ __device__ __forceinline__ constexpr int col_len(int c) { return c + 1; }                         // entries 0..c
__host__ __device__ __forceinline__ constexpr int col_base(int c) { return (c * (c + 1)) >> 1; }           // sum of col_len below c

// four / two consecutive doubles from LDS as single 8-byte reads
typedef double met2_d2 __attribute__((ext_vector_type(2)));
typedef double met2_d4 __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) double *met2_lds_cvp;
__device__ __forceinline__ void lds_quad(const double *p, double &a, double &b, double &c, double &d)
{
    met2_lds_cvp q = (met2_lds_cvp)p;                                  // p points into the wave's LDS region
    a = q[0]; b = q[1]; c = q[2]; d = q[3];
}
__device__ __forceinline__ void lds_pair(const double *p, double &a, double &b)
{
    met2_lds_cvp q = (met2_lds_cvp)p;
    a = q[0]; b = q[1];
}
// the same from a 16-byte aligned address (ds_read_b128): rows of GCV's M, whose stride keeps the lanes' 16-byte chunks apart
__device__ __forceinline__ void lds_quad128(const double *p, double &a, double &b, double &c, double &d)
{
    const met2_d2 *q = (const met2_d2 *)__builtin_assume_aligned(p, 16);
    const met2_d2 u = q[0], v = q[1];
    a = u.x; b = u.y; c = v.x; d = v.y;
}

// ---- NB-aware cross-lane helpers (idx / src index bins or positions 0..64*NB-1) ----
template <int NB>
__device__ __forceinline__ double bcastN(const double (&v)[NB], int idx)
{
    if (NB == 1) return bcast(v[0], idx);
    double t = (idx >> 6) ? v[NB - 1] : v[0];
    return bcast(t, idx & 63);
}
template <int NB>
__device__ __forceinline__ int bcastN_i(const int (&v)[NB], int idx)
{
    if (NB == 1) return bcast_i(v[0], idx);
    int t = (idx >> 6) ? v[NB - 1] : v[0];
    return bcast_i(t, idx & 63);
}
template <int NB>
__device__ __forceinline__ double gatherN(const double (&v)[NB], int src)
{
    double lo = gather(v[0], src & 63);
    if (NB == 1) return lo;
    double hi = gather(v[NB - 1], src & 63);
    return ((src >> 6) & 1) ? hi : lo;
}
template <int NB>
__device__ __forceinline__ int gatherN_i(const int (&v)[NB], int src)
{
    int lo = gather_i(v[0], src & 63);
    if (NB == 1) return lo;
    int hi = gather_i(v[NB - 1], src & 63);
    return ((src >> 6) & 1) ? hi : lo;
}
// index of the first set bit over NB masks (caller guarantees one is set)
template <int NB>
__device__ __forceinline__ int first_bit(const u64 (&m)[NB])
{
    if (NB == 1) return first_lane(m[0]);
    return m[0] ? first_lane(m[0]) : 64 + first_lane(m[NB - 1]);
}
template <int NB>
__device__ __forceinline__ void set_bit(u64 (&m)[NB], int idx)
{
    if (NB == 2 && (idx >> 6)) m[NB - 1] |= (1ull << (idx & 63)); else m[0] |= (1ull << (idx & 63));
}
template <int NB>
__device__ __forceinline__ void clear_bit(u64 (&m)[NB], int idx)
{
    if (NB == 2 && (idx >> 6)) m[NB - 1] &= ~(1ull << (idx & 63)); else m[0] &= ~(1ull << (idx & 63));
}


// NP = number of register slots that hold POSITION-indexed quantities (y, 1/diag, ord, the rows and columns of the factor): NB in
// general, 1 while the passive set fits 64 positions.  At two bins per lane every position-indexed vector operation is issued once per
// slot, and with k <= 64 the second slot is all inactive lanes -- half of the vector instructions of the substitutions, the
// plane-rotation chains and the row-by-row factorisation did nothing (k ~ 40 at 48 x 120, k <= 48 in the FA walk).  The routines
// below take NP as a second template argument; bin-indexed quantities (x, h, pos, rows of B and K) keep their NB slots.
template <int NB, int NP>
__device__ __forceinline__ double bcastPos(const double (&v)[NB], int idx)
{
    if (NP == 1) return bcast(v[0], idx);
    return bcastN<NB>(v, idx);
}
template <int NB, int NP>
__device__ __forceinline__ int bcastPos_i(const int (&v)[NB], int idx)
{
    if (NP == 1) return bcast_i(v[0], idx);
    return bcastN_i<NB>(v, idx);
}

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

// value of the bin-indexed vector v at bin (j - d) / (j + d), d in {1,2}; out-of-range bins return
// arbitrary data (their band coefficient is zero)
template <int NB>
__device__ __forceinline__ void shifted(const double (&v)[NB], int d, int lane, double (&below)[NB], double (&above)[NB])
{
    const int lm = (lane - d) & 63, lp = (lane + d) & 63;
    double lo_m = gather(v[0], lm), lo_p = gather(v[0], lp);
    if (NB == 1) { below[0] = lo_m; above[0] = lo_p; return; }
    double hi_m = gather(v[NB - 1], lm), hi_p = gather(v[NB - 1], lp);
    below[0] = lo_m;                                   // bins < 0 wrap: coefficient zero
    below[NB - 1] = (lane >= d) ? hi_m : lo_m;         // 64+lane-d: slot 1 if lane >= d else slot 0 lanes 64-d+lane
    above[0] = (lane + d < 64) ? lo_p : hi_p;
    above[NB - 1] = hi_p;                              // >= 128 wraps: coefficient zero
}

// (band matrix) * v for a bin-indexed vector v
template <int NB>
__device__ __forceinline__ void band_mul(const double (&bnd)[NB][5], const double (&v)[NB], int lane, double (&out)[NB])
{
    double m1[NB], p1[NB], m2[NB], p2[NB];
    shifted<NB>(v, 1, lane, m1, p1);
    shifted<NB>(v, 2, lane, m2, p2);
#pragma unroll
    for (int b = 0; b < NB; ++b)
        out[b] = bnd[b][2] * v[b] + bnd[b][0] * m2[b] + bnd[b][1] * m1[b] + bnd[b][3] * p1[b] + bnd[b][4] * p2[b];
}

// Row loads from the L2-resident matrices (B, K, D, D^T): element [row][col] of a row-major array whose row index is
// wave-uniform and whose column differs per lane.  As a raw buffer load the uniform part travels in scalar registers
// (descriptor = array base, soffset = row offset) and the lane part is one 32-bit VGPR byte offset: no VALU address
// arithmetic at all.  The plain `base[row * stride + col]` costs one 64-bit v_lshl_add_u64 per load -- 3 500 of the X2 kernel's
// 78 k VALU instructions per voxel.  Out-of-range offsets return 0 (the descriptor spans 1 GiB from the array base).
__device__ __forceinline__ double ld_row(const double *array_base, int row_off_elems, unsigned lane_off_bytes)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)array_base, 0, 0x40000000, 0x00020000);
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_off_bytes, (unsigned)row_off_elems * 8u, 0);
    return __hiloint2double((int)v[1], (int)v[0]);
}

__device__ __forceinline__ double ld_row_sel(bool buffer, const double *array_base, int row_off_elems, unsigned lane_elem)
{
    return buffer ? ld_row(array_base, row_off_elems, 8u * lane_elem) : array_base[row_off_elems + (int)lane_elem];
}

// 1/sqrt(d) to fp64 accuracy from v_rsq_f64 (relative error 5.2e-8, measured: scripts/probes/rsq_precision.hip) and ONE
// third-order correction: with e = 1/2 - (d/2) r^2 = -delta - delta^2/2 the product r (1 + e + 3/2 e^2) = (1 + delta^3) / sqrt(d),
// i.e. 1.4e-22 before rounding -- one instruction less than two Newton steps (d > 0, normal range)
__device__ __forceinline__ double rsqrt_nr(double d)
{
    const double r = __builtin_amdgcn_rsq(d);
    const double e = fma(-(0.5 * d) * r, r, 0.5);
    return fma(r, e * fma(1.5, e, 1.0), r);
}

// 1/d to fp64 accuracy from v_rcp_f64 (relative error 4.6e-8) and one third-order correction: r (1 + e + e^2), e = 1 - d r
__device__ __forceinline__ double rcp_nr(double d)
{
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// Lawson-Hanson plane rotation (g1): c = a / sig, s = b / sig, sig = sqrt(a^2 + b^2) >= 0 (g1's two branches
// reduce to this; the operands here are O(1), so the unscaled a^2 + b^2 neither overflows nor underflows).
// rinv = 1 / sig comes out of the same rsq + Newton chain that replaces the sqrt and the two divisions.
__device__ __forceinline__ void givens(double a, double b, double &c, double &s, double &sig, double &rinv)
{
    const double r2 = fma(a, a, b * b);
    if (r2 > 0.0) { rinv = rsqrt_nr(r2); c = a * rinv; s = b * rinv; sig = r2 * rinv; }
    else { sig = 0.0; c = 0.0; s = 1.0; rinv = INFINITY; }
}

// back substitution R z = y ; z position-indexed.
// The column loop is unrolled by two with two named prefetch registers: the LDS read issued in one step is
// consumed a full step later, so the compiler can wait with lgkmcnt(1) instead of draining the read it has
// just issued (a single rotating register forces lgkmcnt(0) in front of every FMA).
template <int NB, int NP = NB>
__device__ __forceinline__ void back_subst(const WaveShared &S, const NnlsState<NB> &st, int lane, double (&z)[NB])
{
    const int k = st.k;
    double y[NB], ra[NB], rb[NB];
    int cb = col_base(k - 1);                                           // column c of the loop below
#pragma unroll
    for (int b = 0; b < NP; ++b) {
        const int pl = lane + 64 * b;
        y[b] = st.y[b];
        ra[b] = (k > 0 && pl < k - 1) ? S.R[cb + pl] : 0.0;             // column k-1
        rb[b] = 0.0;
    }
    int c = k - 1;
    for (; c >= 1; c -= 2) {
        const int cb1 = cb - col_len(c - 1), cb2 = cb1 - col_len(c - 2);   // col_base(c-1), col_base(c-2)
#pragma unroll
        for (int b = 0; b < NP; ++b) { const int pl = lane + 64 * b; rb[b] = (pl < c - 1) ? S.R[cb1 + pl] : 0.0; }   // column c-1
        {
            double t = (NP == 2 && (c >> 6)) ? y[NB - 1] * st.rinv[NB - 1] : y[0] * st.rinv[0];
            double s = bcast(t, c & 63);
#pragma unroll
            for (int b = 0; b < NP; ++b) y[b] = fma(-ra[b], s, y[b]);        // positions >= c are final
        }
#pragma unroll
        for (int b = 0; b < NP; ++b) { const int pl = lane + 64 * b; ra[b] = (c >= 2 && pl < c - 2) ? S.R[cb2 + pl] : 0.0; }   // column c-2
        {
            const int c1 = c - 1;
            double t = (NP == 2 && (c1 >> 6)) ? y[NB - 1] * st.rinv[NB - 1] : y[0] * st.rinv[0];
            double s = bcast(t, c1 & 63);
#pragma unroll
            for (int b = 0; b < NP; ++b) y[b] = fma(-rb[b], s, y[b]);
        }
        cb = cb2;
    }
    // c == 0 needs no update (nothing lies above row 0's diagonal entry); c == -1: done
#pragma unroll
    for (int b = 0; b < NB; ++b) z[b] = (b < NP && lane + 64 * b < k) ? y[b] * st.rinv[b] : 0.0;
}

// Remove position p from the passive set: delete column p of R and re-triangularise.
template <int NB, int NP = NB>
__device__ __forceinline__ void remove_pos(const WaveShared &S, NnlsState<NB> &st, int p, int lane)
{
    const int k = st.k;
    MET2_CYC_BEGIN(c_rm);
    const int tb = bcastPos_i<NB, NP>(st.ord, p);
    if (p < k - 1) {
        int cbl[NB];                                                    // col_base of the lane's own (old) column
#pragma unroll
        for (int b = 0; b < NP; ++b) cbl[b] = col_base(lane + 64 * b);
        // rows above p: column c+1 moves into column c.  With the pivot order of reorder_by_x() the variable that leaves sits near the
        // END of the order (k - 1 - p is 1 or 2): then the few columns behind p are moved one by one, lane <-> row; otherwise row by
        // row, lane <-> column (all reads of a row, then its writes).
        if (k - 1 - p < p) {
            for (int c = p; c <= k - 2; ++c) {
                const int src = col_base(c + 1), dst = col_base(c);
                double v[NB];
#pragma unroll
                for (int b = 0; b < NP; ++b) { const int pl = lane + 64 * b; v[b] = (pl < p) ? S.R[src + pl] : 0.0; }
#pragma unroll
                for (int b = 0; b < NP; ++b) { const int pl = lane + 64 * b; if (pl < p) S.R[dst + pl] = v[b]; }
            }
            __builtin_amdgcn_wave_barrier();
        } else
        for (int i = 0; i < p; ++i) {
            double v[NB];
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                const int pl = lane + 64 * b;
                v[b] = (pl >= p && pl <= k - 2) ? S.R[cbl[b] + col_len(pl) + i] : 0.0;   // col_base(pl+1) = col_base(pl) + col_len(pl)
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                const int pl = lane + 64 * b;
                if (pl >= p && pl <= k - 2) S.R[cbl[b] + i] = v[b];
            }
        }
        // rows p..k-1: chain of plane rotations, owned index = old column index
        double carry[NB];
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            const int pl = lane + 64 * b;
            carry[b] = (pl > p && pl < k) ? S.R[cbl[b] + p] : 0.0;
        }
        double ycar = bcastPos<NB, NP>(st.y, p);
        for (int j = p + 1; j < k; ++j) {
            double rowj[NB];
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                const int pl = lane + 64 * b;
                rowj[b] = (pl >= j && pl < k) ? S.R[cbl[b] + j] : 0.0;
            }
            double a = bcastPos<NB, NP>(carry, j), bb = bcastPos<NB, NP>(rowj, j);
            double c, s, sig, sinv;
            givens(a, bb, c, s, sig, sinv);
            double yj = bcastPos<NB, NP>(st.y, j);
            double ynew = c * ycar + s * yj;
            ycar = -s * ycar + c * yj;
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                const int pl = lane + 64 * b;
                double nv = c * carry[b] + s * rowj[b];
                carry[b] = -s * carry[b] + c * rowj[b];
                // new entry (j-1, pl-1): col_base(pl-1) = col_base(pl) - col_len(pl-1)
                if (pl > j && pl < k) S.R[cbl[b] - col_len(pl - 1) + j - 1] = nv;
                if (pl == j) S.R[cbl[b] - col_len(pl - 1) + j - 1] = sig;
                if (pl == j - 1) { st.y[b] = ynew; st.rinv[b] = sinv; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    int ordn[NB];
#pragma unroll
    for (int b = 0; b < NP; ++b) ordn[b] = (NP == 1) ? gather_i(st.ord[0], (lane + 1) & 63) : gatherN_i<NB>(st.ord, lane + 64 * b + 1);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        if (b < NP) st.ord[b] = (pl >= p) ? ordn[b] : st.ord[b];                          // by position
        st.pos[b] = (st.pos[b] > p) ? st.pos[b] - 1 : st.pos[b];                          // by bin
        if (pl == tb) { st.pos[b] = -1; st.x[b] = 0.0; }
    }
    clear_bit<NB>(st.P, tb);
    st.k = k - 1;
    MET2_CYC_END(7, c_rm);
    MET2_CYC_ADD(6, 1);
}

// Try to move bin t from Z to P.  Returns false (state untouched) when the column is numerically
// dependent on the passive columns or -- unless forced -- its trial coefficient is not positive
// (Lawson-Hanson's two acceptance tests).
template <int NB, int NP = NB>
__device__ __forceinline__ bool try_append(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int t, int lane,
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
        if (lam != 0.0) gb[b] = fma(lam, (j < S.n) ? ld_row_sel(S.buffer_rows, S.K, t * S.n, je) : 0.0, gb[b]);       // G[j][t] (K is symmetric: row t)
    }
    const double gtt = bcastN<NB>(gb, t);
    double g[NB], rv[NB];
    int cbl[NB];
#pragma unroll
    for (int b = 0; b < NP; ++b) {
        const int pl = lane + 64 * b;
        cbl[b] = col_base(pl);
        double gg = gatherN<NB>(gb, st.ord[b]);                          // position-indexed G[ord_p][t]
        g[b] = (pl < k) ? gg : 0.0;
        rv[b] = (k > 0 && pl > 0 && pl < k) ? S.R[cbl[b]] : 0.0;         // row 0
    }
    // forward substitution R^T r = g, unrolled by two with two prefetch registers (see back_subst); a lane walks
    // down its own column, so the row index is an immediate offset from a fixed per-lane address
    {
        double rw[NB];
#pragma unroll
        for (int b = 0; b < NP; ++b) rw[b] = 0.0;
        int i = 0;
        for (; i + 1 < k; i += 2) {
#pragma unroll
            for (int b = 0; b < NP; ++b) { const int pl = lane + 64 * b; rw[b] = (pl > i + 1 && pl < k) ? S.R[cbl[b] + i + 1] : 0.0; }   // row i+1
            {
                double tt = (NP == 2 && (i >> 6)) ? g[NB - 1] * st.rinv[NB - 1] : g[0] * st.rinv[0];
                double s = bcast(tt, i & 63);
#pragma unroll
                for (int b = 0; b < NP; ++b) g[b] = fma(-rv[b], s, g[b]);      // positions <= i are final
            }
#pragma unroll
            for (int b = 0; b < NP; ++b) { const int pl = lane + 64 * b; rv[b] = (i + 2 < k && pl > i + 2 && pl < k) ? S.R[cbl[b] + i + 2] : 0.0; }   // row i+2
            {
                const int i1 = i + 1;
                double tt = (NP == 2 && (i1 >> 6)) ? g[NB - 1] * st.rinv[NB - 1] : g[0] * st.rinv[0];
                double s = bcast(tt, i1 & 63);
#pragma unroll
                for (int b = 0; b < NP; ++b) g[b] = fma(-rw[b], s, g[b]);
            }
        }
        // an odd last step i == k-1 updates nobody (no position lies beyond k-1)
    }
    double r[NB], rr = 0.0, ry = 0.0;
#pragma unroll
    for (int b = 0; b < NP; ++b) {
        r[b] = (lane + 64 * b < k) ? g[b] * st.rinv[b] : 0.0;
        rr = fma(r[b], r[b], rr);
        ry = fma(r[b], st.y[b], ry);
    }
    wave_sum2(rr, ry);
    const double rho2 = gtt - rr;
    if (!(rho2 > 1e-14 * gtt)) return false;                     // dependent column (noise floor of gtt - r.r)
    const double rhoinv = rsqrt_nr(rho2), rho = rho2 * rhoinv;
    const double ynew = (bcastN<NB>(st.h, t) - ry) * rhoinv;
    if (!forced && !(ynew > 0.0)) return false;                  // ztest: the trial coefficient ynew / rho has ynew's sign
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        if (b < NP) {                                           // by position
            if (pl < k) S.R[col_base(k) + pl] = r[b];
            if (pl == k) { S.R[col_base(k) + k] = rho; st.rinv[b] = rhoinv; st.y[b] = ynew; st.ord[b] = t; }
        }
        if (pl == t) st.pos[b] = k;                             // by bin
    }
    __builtin_amdgcn_wave_barrier();
    set_bit<NB>(st.P, t);
    st.k = k + 1;
    return true;
}

// dual vector w = h - (B + lam K) x, bin-indexed
template <int NB>
__device__ __forceinline__ void dual(const WaveShared &S, const Band<NB> &bd, const NnlsState<NB> &st, double lam, int lane, double (&w)[NB])
{
    double kbl[NB][5];                                       // this lane's 5 diagonals of K, in flight during the B loop
    if (lam != 0.0) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int d = 0; d < 5; ++d) kbl[b][d] = S.kband[d * 128 + lane + 64 * b];
    }
    // B x over the passive set, four rows of B in flight per step (the rows come from L2: issued back to back they
    // overlap their latency, one at a time each step would wait for its own load)
    double acc[NB], acc2[NB], xp[NB];
    unsigned jc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        acc[b] = 0.0; acc2[b] = 0.0;
        jc[b] = (unsigned)min(lane + 64 * b, S.n - 1);
    }
    const int k = st.k;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const double t = gatherN<NB>(st.x, st.ord[b]);                   // x by position; zero past the set, where ord still
        xp[b] = (lane + 64 * b < k) ? t : 0.0;                            // names a valid bin: the last group needs no clamping
    }
#pragma clang loop unroll(disable)
    for (int p = 0; p < k; p += 4) {
        double v[4][NB], xs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pp = p + q;                                        // <= 64 NB - 1 (p <= k - 1 <= 64 NB - 1 is a multiple of 4)
            const int trow = bcastN_i<NB>(st.ord, pp) * S.bstride;
#pragma unroll
            for (int b = 0; b < NB; ++b) v[q][b] = ld_row_sel(S.buffer_rows, S.B, trow, jc[b]);
            xs[q] = bcastN<NB>(xp, pp);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            acc[b] = fma(v[0][b], xs[0], acc[b]); acc2[b] = fma(v[1][b], xs[1], acc2[b]);
            acc[b] = fma(v[2][b], xs[2], acc[b]); acc2[b] = fma(v[3][b], xs[3], acc2[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] += acc2[b];
    double kx[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) kx[b] = 0.0;
    if (lam != 0.0) band_mul<NB>(kbl, st.x, lane, kx);
#pragma unroll
    for (int b = 0; b < NB; ++b) w[b] = (lane + 64 * b < S.n) ? fma(-lam, kx[b], st.h[b] - acc[b]) : 0.0;
}

} // namespace met2
#include "nnls_big.hpp"
namespace met2 {

// Lawson-Hanson's secondary loop: from a feasible x and a factor consistent with (P, lambda), move to
// the solution of the passive sub-problem, dropping variables that hit zero on the way.
// Returns false when the iteration cap is reached.
template <int NB, int NP = NB, bool BS1 = false>
__device__ __forceinline__ bool nnls_inner(const WaveShared &S, NnlsState<NB> &st, int &iter, int itmax, int lane)
{
    for (;;) {
        if (++iter > itmax) return false;
        double z[NB], xp[NB], zb[NB], ratio[NB];
        bool neg[NB];
        if (MET2_DOUBLE == 4) { back_subst<NB, NP>(S, st, lane, z); asm volatile("" :: "v"(z[0])); }
        if (BS1 && NP == 2 && st.k <= 64) back_subst<NB, 1>(S, st, lane, z);     // (ONE = 3: the substitution on one slot inside the two-slot iteration)
        else back_subst<NB, NP>(S, st, lane, z);             // position-indexed
        bool any = false;
#pragma unroll
        for (int b = NP; b < NB; ++b) { neg[b] = false; ratio[b] = 2.0; xp[b] = 0.0; }
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            const int pl = lane + 64 * b;
            xp[b] = gatherN<NB>(st.x, st.ord[b]);            // x at position
            neg[b] = (pl < st.k) && (z[b] <= 0.0);
            any = any || (ballot(neg[b]) != 0ull);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ps = st.pos[b] < 0 ? 0 : st.pos[b];
            double t = (NP == 1) ? gather(z[0], ps & 63) : gatherN<NB>(z, ps);   // bin-indexed
            zb[b] = (st.pos[b] >= 0) ? t : 0.0;
        }
        if (!any) {
#pragma unroll
            for (int b = 0; b < NB; ++b) st.x[b] = zb[b];
            return true;
        }
        double rmin = 2.0;
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            double r = neg[b] ? xp[b] / (xp[b] - z[b]) : 2.0;
            ratio[b] = (r == r) ? r : 2.0;                   // 0/0: Lawson-Hanson's `alpha > t` is false for NaN
            rmin = fmin(rmin, ratio[b]);
        }
        const double alpha = wave_min(rmin);
        if (!(alpha < 2.0)) {                                // "alpha still 2": accept z
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
        remove_pos<NB, NP>(S, st, jj, lane);
        for (int sweep = 0; sweep < 64 * NB; ++sweep) {   // round-off stragglers (Lawson-Hanson: "any that are nonpositive ...")
            u64 bad[NB];
            bool anyb = false;
#pragma unroll
            for (int b = NP; b < NB; ++b) bad[b] = 0ull;
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                double xq = gatherN<NB>(st.x, st.ord[b]);
                bad[b] = ballot((lane + 64 * b < st.k) && (xq <= 0.0));
                anyb = anyb || (bad[b] != 0ull);
            }
            if (!anyb) break;
            MET2_STAT(1, sweep + 1);
            remove_pos<NB, NP>(S, st, first_bit<NB>(bad), lane);
        }
    }
}

// The passive-set iteration on NB position slots (the general form).
// BIG (here and below): the instance compiled into the spill-over voxel routine (nnls_big.hpp) -- a set that wants to outgrow the LDS capacity
// S.kmax goes on with its columns beyond it in the wave's global slot; BIG = false flags the voxel instead (st.itmax_hit bit 1).
template <int NB, bool BS1 = false, bool BIG = false>
__device__ __forceinline__ void nnls_iterate_plain(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int mrows, int lane,
                                                   bool warm)
{
    const int n = S.n, itmax = 3 * n;
    int iter = 0;
    if constexpr (BIG) if (st.k > S.kmax) { int outer0 = 0; iterate_big<NB>(S, bd, st, lam, mrows, lane, warm, iter, outer0); return; }      // (a warm start beyond the LDS capacity)
    MET2_CYC_BEGIN(c_in0);
    if (warm && st.k > 0 && !nnls_inner<NB, NB, BS1>(S, st, iter, itmax, lane)) { st.itmax_hit |= 1; return; }
    MET2_CYC_END(2, c_in0);
    for (int outer = 0; outer <= itmax + 1; ++outer) {     // every pass runs >= 1 counted inner pass
        if (st.k >= n || st.k >= mrows) break;
        double w[NB];
        MET2_CYC_BEGIN(c_du);
        if (MET2_DOUBLE == 2) { dual<NB>(S, bd, st, lam, lane, w); asm volatile("" :: "v"(w[0])); }
        dual<NB>(S, bd, st, lam, lane, w);
        MET2_CYC_END(3, c_du);
        MET2_CYC_ADD(13, 1);
        if (st.k >= S.kmax) {
            // capacity of the wave's LDS region reached: if a variable still wants to enter, the voxel is flagged -- and solved again at once by
            // the spill-over voxel routine, whose instance of this code (BIG) goes on here with the columns beyond the capacity in global memory
            double vmax = -1.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) vmax = fmax(vmax, ((lane + 64 * b < n) && !((st.P[b] >> lane) & 1ull)) ? w[b] : -1.0);
            if (wave_max(vmax) > 0.0) {
                if (BIG && S.Rg) { bool wf = false; iterate_big<NB>(S, bd, st, lam, mrows, lane, wf, iter, outer); }
                else st.itmax_hit |= 2;
            }
            break;
        }
        // entering variable: largest positive dual among Z; rejected candidates are skipped
        u64 rejected[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) rejected[b] = 0ull;
        bool accepted = false;
        MET2_CYC_BEGIN(c_ap);
        for (int tries = 0; tries < 64 * NB; ++tries) {      // each failed try rejects one more bin
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
            if (try_append<NB>(S, bd, st, lam, t, lane)) { accepted = true; break; }
            set_bit<NB>(rejected, t);
            MET2_STAT(0, tries + 1);
        }
        MET2_CYC_END(4, c_ap);
        MET2_CYC_ADD(14, 1);
        if (!accepted) break;
        MET2_CYC_BEGIN(c_in);
        const bool inner_ok = nnls_inner<NB, NB, BS1>(S, st, iter, itmax, lane);
        MET2_CYC_END(2, c_in);
        MET2_CYC_ADD(15, 1);
        if (!inner_ok) { st.itmax_hit |= 1; break; }
        MET2_STAT(2, outer + 1);
        MET2_STAT(3, iter);
    }
}

// One leg of the passive-set iteration with NP position slots.  NP < NB (one slot: k <= 64) hands over to the general leg when the
// set is about to outgrow 64 positions: returns false then ("not finished"), with iter / outer / warm carried in the arguments.
template <int NB, int NP, bool BIG = false>
__device__ __forceinline__ bool iterate_leg(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int mrows, int lane,
                                            bool &warm, int &iter, int &outer)
{
    const int n = S.n, itmax = 3 * n;
    MET2_CYC_BEGIN(c_in0);
    if (warm && st.k > 0) {
        warm = false;
        if (!nnls_inner<NB, NP>(S, st, iter, itmax, lane)) { st.itmax_hit |= 1; return true; }
    }
    warm = false;
    MET2_CYC_END(2, c_in0);
    for (; outer <= itmax + 1; ++outer) {                  // every pass runs >= 1 counted inner pass
        if (st.k >= n || st.k >= mrows) break;
        if (NP < NB && st.k >= 63) return false;           // the next append would need position 64: the two-slot leg takes over
        double w[NB];
        MET2_CYC_BEGIN(c_du);
        if (MET2_DOUBLE == 2) { dual<NB>(S, bd, st, lam, lane, w); asm volatile("" :: "v"(w[0])); }
        dual<NB>(S, bd, st, lam, lane, w);
        MET2_CYC_END(3, c_du);
        MET2_CYC_ADD(13, 1);
        if (st.k >= S.kmax) {
            // capacity of the wave's LDS region reached: if a variable still wants to enter, the voxel is flagged (BIG: the spill-over leg takes over)
            double vmax = -1.0;
#pragma unroll
            for (int b = 0; b < NB; ++b) vmax = fmax(vmax, ((lane + 64 * b < n) && !((st.P[b] >> lane) & 1ull)) ? w[b] : -1.0);
            if (wave_max(vmax) > 0.0) {
                if (BIG && S.Rg) return false;
                st.itmax_hit |= 2;
            }
            break;
        }
        // entering variable: largest positive dual among Z; rejected candidates are skipped
        u64 rejected[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) rejected[b] = 0ull;
        bool accepted = false;
        MET2_CYC_BEGIN(c_ap);
        for (int tries = 0; tries < 64 * NB; ++tries) {      // each failed try rejects one more bin
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
            if (try_append<NB, NP>(S, bd, st, lam, t, lane)) { accepted = true; break; }
            set_bit<NB>(rejected, t);
            MET2_STAT(0, tries + 1);
        }
        MET2_CYC_END(4, c_ap);
        MET2_CYC_ADD(14, 1);
        if (!accepted) break;
        MET2_CYC_BEGIN(c_in);
        const bool inner_ok = nnls_inner<NB, NP>(S, st, iter, itmax, lane);
        MET2_CYC_END(2, c_in);
        MET2_CYC_ADD(15, 1);
        if (!inner_ok) { st.itmax_hit |= 1; break; }
        MET2_STAT(2, outer + 1);
        MET2_STAT(3, iter);
    }
    return true;
}

#ifndef MET2_ONE_SLOT
#define MET2_ONE_SLOT 1       // 1: at two bins per lane the position-indexed work runs on one register slot while k <= 64 (0: always two),
                              //    in the kernels that ask for it (template argument ONE = 1: those with registers to spare -- the GCV kernel at
                              //    255 VGPRs spilled and lost 31 % with both code paths in it; ONE = 2: the re-factorisation only; ONE = 3: the re-factorisation
                              //    and the back substitution, inside the two-slot iteration)
#endif
// Passive-set iterations until the KKT conditions hold.  mrows = rows of the (augmented) system.
// warm: x is a feasible point whose support is the current passive set and R/y were just rebuilt for
// (P, lam) -- start with the secondary loop instead of from the empty set.
template <int NB, int ONE = 0, bool BIG = false>
__device__ __forceinline__ void nnls_iterate(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int mrows, int lane,
                                             bool warm = false)
{
    if constexpr (NB == 2 && ONE == 1 && MET2_ONE_SLOT) {
        int iter = 0, outer = 0;
        if (st.k <= 63 && iterate_leg<NB, 1, BIG>(S, bd, st, lam, mrows, lane, warm, iter, outer)) return;
        if constexpr (BIG) {
            if (st.k <= S.kmax && iterate_leg<NB, NB, true>(S, bd, st, lam, mrows, lane, warm, iter, outer)) return;
            if (S.Rg) iterate_big<NB>(S, bd, st, lam, mrows, lane, warm, iter, outer);      // the set wants to outgrow (or a warm start is beyond) the LDS capacity
        } else
            (void)iterate_leg<NB, NB>(S, bd, st, lam, mrows, lane, warm, iter, outer);
    } else
        nnls_iterate_plain<NB, (NB == 2 && ONE == 3), BIG>(S, bd, st, lam, mrows, lane, warm);
}

template <int NB>
__device__ __forceinline__ void nnls_reset(NnlsState<NB> &st)
{
#pragma unroll
    for (int b = 0; b < NB; ++b) { st.x[b] = 0.0; st.y[b] = 0.0; st.rinv[b] = 0.0; st.ord[b] = 0; st.pos[b] = -1; st.P[b] = 0ull; }
    st.k = 0;
}

// Row-by-row form of the refactorisation: every row sums over ALL rows above it.  Used at one bin per lane (k <= 64), where it
// is the faster of the two (X2/L2 on configs[1]: 152.9 ms against 165.3 ms for the blocked form below: with at most four
// trailing tiles the blocked form saves little arithmetic and exposes the latency of its gather phase).
// Rebuild R, 1/diag and y = R^-T h_P for the CURRENT passive set and pivot order at a new lambda.
// Row-by-row (left-looking) Cholesky: row i of R is  (G[ord_i][ord_c] - sum_{j<i} R[j][i] R[j][c]) / R[i][i]  with
// lane <-> column c, so a row costs i independent LDS row reads + FMAs (issued four at a time) instead of the i
// dependent substitution steps, reductions, sqrt and divisions of appending column i; the factor R[j][i] is lane i
// of the row just read.  The B and K rows of the next pivot are fetched while the current row is reduced, and
// y comes out of the same sweep (one elimination step per finished row).
// Returns false -- R is then unusable, ord/pos/P/k/x are untouched -- when a pivot falls under the independence
// threshold of try_append; the caller re-appends column by column, which drops such columns.
template <int NB, int NP = NB>
__device__ __forceinline__ bool refactor_rowwise(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int lane)
{
    const int k = st.k, n = S.n;
    int cbl[NB], cbc[NB];
    unsigned jc[NB];
    double g[NB], gb0[NB], gk0[NB], gb1[NB], gk1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        jc[b] = (unsigned)min(pl, n - 1);                               // by bin: the column a lane loads of every row of B and K
        if (b < NP) {                                                   // by position
            cbl[b] = col_base(pl);
            cbc[b] = col_base(min(pl, k - 1));                          // an existing column for the unpredicated reads
            const double hh = gatherN<NB>(st.h, st.ord[b]);
            g[b] = (pl < k) ? hh : 0.0;
        }
    }
    // rows of B and K for pivots p and p + 1 (clamped to the last pivot)
    auto fetch = [&](int p, double (&vb)[NB], double (&vk)[NB]) {
        const int t = bcastPos_i<NB, NP>(st.ord, min(p, k - 1));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            vb[b] = ld_row_sel(S.buffer_rows, S.B, t * S.bstride, jc[b]);
            vk[b] = ld_row_sel(S.buffer_rows, S.K, t * n, jc[b]);
        }
    };
    // finish a row: a holds A[i][c] - sum_{j<i} R[j][i] R[j][c]; scale, store column entries, one elimination step for y
    // (the independence test of the pivots is done for all rows at once after the sweep: a non-positive or NaN pivot only
    //  poisons the rows below it, and the whole factor is discarded then)
    auto finish = [&](int i, double (&a)[NB], double (&r)[NB]) {
        const double d = bcastPos<NB, NP>(a, i);
        const double rinv = rsqrt_nr(d);
        const double yi = bcastPos<NB, NP>(g, i) * rinv;
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            const int pl = lane + 64 * b;
            r[b] = a[b] * rinv;                                         // lane i: d * rinv = R[i][i]
            if (pl >= i && pl < k) S.R[cbl[b] + i] = r[b];
            g[b] = (pl > i) ? fma(-r[b], yi, g[b]) : g[b];
        }
    };
    // diagonal of G = B + lam K at the passive bins, by position (for the test after the sweep)
    double gdp[NB];
    {
        double gdb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double bd0 = S.have_bdiag ? S.bdiag[b] : S.B[jc[b] * S.bstride + jc[b]];
            gdb[b] = (lam != 0.0) ? fma(lam, S.K[jc[b] * n + jc[b]], bd0) : bd0;
        }
#pragma unroll
        for (int b = 0; b < NP; ++b) gdp[b] = gatherN<NB>(gdb, st.ord[b]);
    }
    fetch(0, gb0, gk0);
    fetch(1, gb1, gk1);
    int cbi = 0;                                                        // col_base(i)
    bool bad = false;
    int i = 0;
    // two rows per step: rows i and i + 1 share the reads of the rows above them, and row i + 1 takes row i's
    // contribution from registers, so the pair costs one LDS round trip instead of two
    for (; i + 1 < k; i += 2) {
        double a[NB], c[NB];
        {
            double t0[NB], t1[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { t0[b] = fma(lam, gk0[b], gb0[b]); t1[b] = fma(lam, gk1[b], gb1[b]); }   // G[.][ord_i], G[.][ord_i+1]
            fetch(i + 2, gb0, gk0);
            fetch(i + 3, gb1, gk1);
#pragma unroll
            for (int b = 0; b < NP; ++b) { a[b] = gatherN<NB>(t0, st.ord[b]); c[b] = gatherN<NB>(t1, st.ord[b]); }
        }
        const double *ci = S.R + cbi, *cj = ci + col_len(i);            // columns i and i + 1
        int j = 0;
#pragma clang loop unroll(disable)
        for (; j + 4 <= i; j += 4) {                                    // four rows above per step: the three LDS pointers are bumped once per eight FMAs
            double s0, s1, s2, s3, u0, u1, u2, u3;                      // columns are NOT 16-byte aligned: lds_quad must stay four single 8-byte reads
            lds_quad(ci + j, s0, s1, s2, s3);
            lds_quad(cj + j, u0, u1, u2, u3);
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                double q0, q1, q2, q3;
                lds_quad(S.R + cbc[b] + j, q0, q1, q2, q3);
                a[b] = fma(-s0, q0, a[b]); c[b] = fma(-u0, q0, c[b]);      // the two rows are the two independent chains
                a[b] = fma(-s1, q1, a[b]); c[b] = fma(-u1, q1, c[b]);
                a[b] = fma(-s2, q2, a[b]); c[b] = fma(-u2, q2, c[b]);
                a[b] = fma(-s3, q3, a[b]); c[b] = fma(-u3, q3, c[b]);
            }
        }
        if (j + 2 <= i) {                                               // two rows above (one ds_read2 per column)
            double s0, s1, u0, u1;
            lds_pair(ci + j, s0, s1);
            lds_pair(cj + j, u0, u1);
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                double q0, q1;
                lds_pair(S.R + cbc[b] + j, q0, q1);
                a[b] = fma(-s0, q0, a[b]); c[b] = fma(-u0, q0, c[b]);
                a[b] = fma(-s1, q1, a[b]); c[b] = fma(-u1, q1, c[b]);
            }
            j += 2;
        }
        for (; j < i; ++j) {
            const double s0 = ci[j], u0 = cj[j];
#pragma unroll
            for (int b = 0; b < NP; ++b) { const double q0 = S.R[cbc[b] + j]; a[b] = fma(-s0, q0, a[b]); c[b] = fma(-u0, q0, c[b]); }
        }
        double r[NB], r1[NB];
        finish(i, a, r);
        const double sr = bcastPos<NB, NP>(r, i + 1);                         // R[i][i+1]
#pragma unroll
        for (int b = 0; b < NP; ++b) c[b] = fma(-sr, r[b], c[b]);
        finish(i + 1, c, r1);
        __builtin_amdgcn_wave_barrier();
        cbi += col_len(i) + col_len(i + 1);                             // col_base(i + 2) - col_base(i)
    }
    if (i < k) {                                                        // odd k: the last row on its own
        double a[NB], a2[NB];
        {
            double t0[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) t0[b] = fma(lam, gk0[b], gb0[b]);
#pragma unroll
            for (int b = 0; b < NP; ++b) { a[b] = gatherN<NB>(t0, st.ord[b]); a2[b] = 0.0; }
        }
        const double *ci = S.R + cbi;
        int j = 0;
#pragma clang loop unroll(disable)
        for (; j + 4 <= i; j += 4) {
            double s0, s1, s2, s3;
            lds_quad(ci + j, s0, s1, s2, s3);
#pragma unroll
            for (int b = 0; b < NP; ++b) {
                double q0, q1, q2, q3;
                lds_quad(S.R + cbc[b] + j, q0, q1, q2, q3);
                a[b] = fma(-s0, q0, a[b]); a2[b] = fma(-s1, q1, a2[b]);
                a[b] = fma(-s2, q2, a[b]); a2[b] = fma(-s3, q3, a2[b]);
            }
        }
        for (; j < i; ++j) {
            const double s0 = ci[j];
#pragma unroll
            for (int b = 0; b < NP; ++b) a[b] = fma(-s0, S.R[cbc[b] + j], a[b]);
        }
        double r[NB];
#pragma unroll
        for (int b = 0; b < NP; ++b) a[b] += a2[b];
        finish(i, a, r);
        __builtin_amdgcn_wave_barrier();
    }
    // lane p: R[p][p]; its square is the pivot the row was scaled with: all pivots against the independence threshold of
    // try_append at once (NaN from a negative pivot fails the comparison too)
    double dgl[NB];
#pragma unroll
    for (int b = 0; b < NP; ++b) {
        const int pl = lane + 64 * b;
        dgl[b] = (pl < k) ? S.R[cbl[b] + pl] : 1.0;
        bad = bad || (ballot((pl < k) && !(dgl[b] * dgl[b] > 1e-14 * gdp[b])) != 0ull);
    }
    if (bad) return false;
    // lane p: 1 / R[p][p] and y_p = g_p / R[p][p] (g_p is final once rows < p are eliminated)
#pragma unroll
    for (int b = 0; b < NP; ++b) {
        const int pl = lane + 64 * b;
        const double ri = rcp_nr(dgl[b]);
        st.rinv[b] = (pl < k) ? ri : st.rinv[b];
        st.y[b] = (pl < k) ? g[b] * ri : st.y[b];
    }
    return true;
}

// Rebuild R, 1/diag and y = R^-T h_P for the CURRENT passive set and pivot order at a new lambda: blocked right-looking
// Cholesky of A = G_PP (G = B + lam K, rows and columns in pivot order) in the packed LDS triangle.
//   (0) A's upper triangle is gathered into LDS, four pivots' rows of B and K in flight;
//   per block row of 16 pivots:
//   (a) its rows are finished two at a time, lane <-> column, with inner products over the rows of the block only (<= 15
//       terms: everything above the block has been subtracted by the trailing updates), y by one elimination step per row;
//   (b) every trailing 16 x 16 tile C(ti, tj), ti <= tj, takes C -= R(block, ti)^T R(block, tj) as four v_mfma_f64_16x16x4
//       (operands one f64 per lane from the packed columns, accumulator = the tile).
// Used at two bins per lane (k up to 128), where the row-by-row form's sums get long: the brute-force-FA GCV pipeline of
// configs[4] 220 -> 225 k voxels/s.
// Returns false -- R is then unusable, ord/pos/P/k/x are untouched -- when a pivot falls under the independence
// threshold of try_append; the caller re-appends column by column, which drops such columns.
template <int NB>
__device__ __forceinline__ bool refactor_blocked(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int lane)
{
    const int k = st.k, n = S.n;
    int cbl[NB], cbc[NB];
    unsigned jc[NB];
    double g[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        cbl[b] = col_base(pl);
        cbc[b] = col_base(min(pl, k - 1));                              // an existing column for the unpredicated reads
        jc[b] = (unsigned)min(pl, n - 1);
        const double hh = gatherN<NB>(st.h, st.ord[b]);
        g[b] = (pl < k) ? hh : 0.0;
    }
    // diagonal of G at the passive bins, by position (for the test after the sweep)
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
    // ---- (0) A into LDS.  Positions past the set still name valid bins (their rows are loaded and dropped).
#pragma clang loop unroll(disable)
    for (int i = 0; i < k; i += 4) {
        double vb[4][NB], vk[4][NB];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = bcastN_i<NB>(st.ord, i + q);                  // i + q <= 64 NB - 1 (i is a multiple of 4 below k <= 64 NB)
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
            for (int b = 0; b < NB; ++b) t0[b] = fma(lam, vk[q][b], vb[q][b]);     // G[ord_{i+q}][.] by bin
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int pl = lane + 64 * b;
                const double av = gatherN<NB>(t0, st.ord[b]);                       // by position
                if (i + q < k && pl >= i + q && pl < k) S.R[cbl[b] + i + q] = av;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // finish a row: a holds A[i][c] - sum_{j<i} R[j][i] R[j][c]; scale, store column entries, one elimination step for y
    // (the independence test of the pivots is done for all rows at once after the sweep: a non-positive or NaN pivot only
    //  poisons the rows below it, and the whole factor is discarded then)
    auto finish = [&](int i, double (&a)[NB], double (&r)[NB]) {
        const double d = bcastN<NB>(a, i);
        const double rinv = rsqrt_nr(d);
        const double yi = bcastN<NB>(g, i) * rinv;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int pl = lane + 64 * b;
            r[b] = a[b] * rinv;                                         // lane i: d * rinv = R[i][i]
            if (pl >= i && pl < k) S.R[cbl[b] + i] = r[b];
            g[b] = (pl > i) ? fma(-r[b], yi, g[b]) : g[b];
        }
    };
    const int li = lane & 15, lk = lane >> 4;
    const int nt = (k + 15) >> 4;
    for (int kb = 0; kb < nt; ++kb) {
        const int r0 = 16 * kb, r1 = min(k, r0 + 16);
        int cbi = col_base(r0);
        int i = r0;
        // ---- (a) rows i and i + 1 together: they share the reads of the block's rows above them, and row i + 1 takes
        //      row i's contribution from registers
        for (; i + 1 < r1; i += 2) {
            double a[NB], c[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int pl = lane + 64 * b;
                double a0, c0;
                lds_pair(S.R + cbc[b] + i, a0, c0);                     // rows i, i + 1 of the lane's column (i is even)
                a[b] = (pl >= i && pl < k) ? a0 : 0.0;
                c[b] = (pl > i && pl < k) ? c0 : 0.0;
            }
            const double *ci = S.R + cbi, *cj = ci + col_len(i);        // columns i and i + 1
            int j = r0;
#pragma clang loop unroll(disable)
            for (; j + 4 <= i; j += 4) {
                double s0, s1, s2, s3, u0, u1, u2, u3;                  // columns are NOT 16-byte aligned: lds_quad must stay four single 8-byte reads
                lds_quad(ci + j, s0, s1, s2, s3);
                lds_quad(cj + j, u0, u1, u2, u3);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1, q2, q3;
                    lds_quad(S.R + cbc[b] + j, q0, q1, q2, q3);
                    a[b] = fma(-s0, q0, a[b]); c[b] = fma(-u0, q0, c[b]);  // the two rows are the two independent chains
                    a[b] = fma(-s1, q1, a[b]); c[b] = fma(-u1, q1, c[b]);
                    a[b] = fma(-s2, q2, a[b]); c[b] = fma(-u2, q2, c[b]);
                    a[b] = fma(-s3, q3, a[b]); c[b] = fma(-u3, q3, c[b]);
                }
            }
            if (j < i) {                                                // i - r0 is even: two rows left
                double s0, s1, u0, u1;
                lds_pair(ci + j, s0, s1);
                lds_pair(cj + j, u0, u1);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1;
                    lds_pair(S.R + cbc[b] + j, q0, q1);
                    a[b] = fma(-s0, q0, a[b]); c[b] = fma(-u0, q0, c[b]);
                    a[b] = fma(-s1, q1, a[b]); c[b] = fma(-u1, q1, c[b]);
                }
            }
            double r[NB], r2[NB];
            finish(i, a, r);
            const double sr = bcastN<NB>(r, i + 1);                     // R[i][i+1]
#pragma unroll
            for (int b = 0; b < NB; ++b) c[b] = fma(-sr, r[b], c[b]);
            finish(i + 1, c, r2);
            __builtin_amdgcn_wave_barrier();
            cbi += col_len(i) + col_len(i + 1);
        }
        if (i < r1) {                                                   // odd k: the last row on its own
            double a[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; const double a0 = S.R[cbc[b] + i]; a[b] = (pl >= i && pl < k) ? a0 : 0.0; }
            const double *ci = S.R + cbi;
            int j = r0;
#pragma clang loop unroll(disable)
            for (; j + 2 <= i; j += 2) {
                double s0, s1;
                lds_pair(ci + j, s0, s1);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    double q0, q1;
                    lds_pair(S.R + cbc[b] + j, q0, q1);
                    a[b] = fma(-s0, q0, a[b]);
                    a[b] = fma(-s1, q1, a[b]);
                }
            }
            double r[NB];
            finish(i, a, r);
            __builtin_amdgcn_wave_barrier();
        }
        // ---- (b) trailing tiles on the matrix cores (only a full block has columns behind it)
        for (int ti = kb + 1; ti < nt; ++ti) {
            const int ca = 16 * ti + li;                                // this lane's column inside tile column ti
            const int cba = col_base(min(ca, k - 1));
            double aop[4];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const double v = S.R[cba + r0 + 4 * sidx + lk];
                aop[sidx] = (ca < k) ? -v : 0.0;
            }
            for (int tj = ti; tj < nt; ++tj) {
                const int cc = 16 * tj + li;
                const int cbb = col_base(min(cc, k - 1));
                double bop[4];
                met2_d4 acc;
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const double v = S.R[cbb + r0 + 4 * sidx + lk];
                    bop[sidx] = (cc < k) ? v : 0.0;
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * ti + lk + 4 * v;
                    acc[v] = (cc < k && row <= cc) ? S.R[cbb + row] : 0.0;
                }
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[sidx], bop[sidx], acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 16 * ti + lk + 4 * v;
                    if (cc < k && row <= cc) S.R[cbb + row] = acc[v];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // lane p: R[p][p]; its square is the pivot the row was scaled with: all pivots against the independence threshold of
    // try_append at once (NaN from a negative pivot fails the comparison too)
    bool bad = false;
    double dgl[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        dgl[b] = (pl < k) ? S.R[cbl[b] + pl] : 1.0;
        bad = bad || (ballot((pl < k) && !(dgl[b] * dgl[b] > 1e-14 * gdp[b])) != 0ull);
    }
    if (bad) return false;
    // lane p: 1 / R[p][p] and y_p = g_p / R[p][p] (g_p is final once rows < p are eliminated)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int pl = lane + 64 * b;
        const double ri = rcp_nr(dgl[b]);
        st.rinv[b] = (pl < k) ? ri : st.rinv[b];
        st.y[b] = (pl < k) ? g[b] * ri : st.y[b];
    }
    return true;
}

#ifndef MET2_ONE_SLOT_REFACTOR
#define MET2_ONE_SLOT_REFACTOR 1    // 1: with k <= 64 at two bins per lane the warm re-factorisation is the row-by-row form on one slot (0: the blocked MFMA form)
#endif
#ifndef MET2_REFACTOR_BLOCKED_FROM
#define MET2_REFACTOR_BLOCKED_FROM 2        // bins per lane from which the blocked form is used
#endif
template <int NB, int ONE = 0, bool BIG = false>
__device__ __forceinline__ bool refactor(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, int lane)
{
    if constexpr (BIG) if (st.k > S.kmax) return refactor_big<NB>(S, bd, st, lam, lane);   // beyond the LDS capacity: columns >= kmax in the wave's global slot
    if (NB == 2 && ONE && MET2_ONE_SLOT && MET2_ONE_SLOT_REFACTOR && st.k <= 64) return refactor_rowwise<NB, 1>(S, bd, st, lam, lane);   // k <= 64: one position slot, row by row
    if (NB >= MET2_REFACTOR_BLOCKED_FROM) return refactor_blocked<NB>(S, bd, st, lam, lane);
    if (MET2_DOUBLE == 1) { (void)refactor_rowwise<NB>(S, bd, st, lam, lane); __builtin_amdgcn_wave_barrier(); }
    return refactor_rowwise<NB>(S, bd, st, lam, lane);
}

// cold-start solve; on return st.x is the solution
template <int NB, int ONE = 0, bool BIG = false>
__device__ __forceinline__ void nnls_solve(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, bool aug, int lane)
{
    nnls_reset<NB>(st);
    nnls_iterate<NB, ONE, BIG>(S, bd, st, lam, aug ? S.m + S.n : S.m, lane);
}

// Pivot order for a warm start: the passive bins by DESCENDING x.  The re-factorisation builds the factor in whatever order it is
// given, so the order is free -- and it decides what a later exchange costs: removing position p re-triangularises the k - 1 - p
// columns behind it (a chain of plane rotations, ~40 vector instructions each).  The bins that leave when lambda moves are the ones
// whose coefficient is already small: measured on the reference's recipe (consecutive Brent abscissae, 150 voxels) the leaving bin is
// the smallest-x bin of the set in 42 % of the removals and among the three smallest in 83 %, so with the smallest coefficients LAST
// the chains have one or two links where the entering order (largest dual first) gave ~15.
// MEASURED (round 3): at one bin per lane (configs[1]) the removals' wave cycles fell 6x (5.4 k -> 0.9 k per Brent evaluation) but
// the kernel did not get faster (147.5 vs 147.7 ms) and its vector-instruction count barely moved (60.7 k -> 60.5 k per voxel): the
// chains were long in latency (an LDS round trip per link), not in instructions, and that kernel is bound by vector-instruction
// issue -- off there.  At two bins per lane (k ~ 40, every link two registers wide) it pays: X2/L2 at 48 x 120 32.1 -> 30.2 ms
// per 32 768 voxels (+7 %), the GCV kernel of configs[4] +1.4 %; L-curve at 32 x 60 +0.4 %.  On from MET2_REORDER bins per lane.
// Keys: the bit pattern of x (non-negative doubles order like unsigned integers) with its low 7 bits replaced by 127 - bin: all
// different, so the ranks form a permutation whatever the data (any pivot order is valid; only the cost depends on it).  The new
// position -> bin table goes through the wave's factor region, which the re-factorisation overwrites next.
template <int NB>
__device__ __forceinline__ void reorder_by_x(const WaveShared &S, NnlsState<NB> &st, int lane)
{
    const int k = st.k;
    typedef __attribute__((address_space(3))) unsigned long long *lds_u64p;
    unsigned long long *tabk = (unsigned long long *)S.R;          // [k + 4] keys by OLD position, zero padded (a zero key outranks nobody)
    int *tabi = (int *)(tabk + k + 4);                             // [64 NB] new position -> bin
    unsigned long long key[NB];
    bool in[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = lane + 64 * b;
        in[b] = (st.P[b] >> lane) & 1ull;
        key[b] = ((unsigned long long)__double_as_longlong(st.x[b]) & ~127ull) | (unsigned long long)(127 - j);
        if (in[b]) tabk[st.pos[b]] = key[b];
        if (j >= k && j < k + 4) tabk[j] = 0ull;
    }
    __builtin_amdgcn_wave_barrier();
    // rank = number of passive bins with a larger key: k uniform-address LDS reads (independent of each other: no broadcast chain)
    int rank[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) rank[b] = 0;
#pragma clang loop unroll(disable)
    for (int p = 0; p < k; p += 4) {
        const unsigned long long k0 = tabk[p], k1 = tabk[p + 1], k2 = tabk[p + 2], k3 = tabk[p + 3];
#pragma unroll
        for (int b = 0; b < NB; ++b) rank[b] += (int)(k0 > key[b]) + (int)(k1 > key[b]) + (int)(k2 > key[b]) + (int)(k3 > key[b]);
    }
    // bins outside the set take the positions behind it, in bin order: a full permutation of the 64 NB entries
    int zbase = k;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int below = __popcll(~st.P[b] & ((1ull << lane) - 1ull));
        tabi[in[b] ? rank[b] : zbase + below] = lane + 64 * b;
        st.pos[b] = in[b] ? rank[b] : -1;
        zbase += 64 - __popcll(st.P[b]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < NB; ++b) st.ord[b] = min(tabi[lane + 64 * b], S.n - 1);   // positions past the set must still name valid bins (row loads in fours)
    __builtin_amdgcn_wave_barrier();
}

// Warm start: keep the previous solution's passive set and x (a feasible point for any lambda),
// rebuild the factor for the new lambda in the same pivot order, then iterate.  The minimiser of the
// strictly convex problem does not depend on the starting point, so this returns the same x as the
// cold start up to rounding; it only skips the passes that would rebuild the same passive set.
template <int NB, int ONE = 0, bool BIG = false>
__device__ __forceinline__ void nnls_solve_warm(const WaveShared &S, const Band<NB> &bd, NnlsState<NB> &st, double lam, bool aug, int lane)
{
    const int kold = st.k;
    if (kold == 0) { nnls_solve<NB, ONE, BIG>(S, bd, st, lam, aug, lane); return; }
    MET2_CYC_BEGIN(c_ref);
    if (NB >= MET2_REORDER && S.reorder && kold >= 4 && S.rcap >= kold + 4 + 32 * NB + 2) reorder_by_x<NB>(S, st, lane);
    if (!refactor<NB, ONE, BIG>(S, bd, st, lam, lane)) {
        MET2_CYC_ADD(4, 1000000000000ull);               // fallbacks show up in the 1e12 digits of the append slot
        int ordold[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { ordold[b] = st.ord[b]; st.pos[b] = -1; st.P[b] = 0ull; }
        st.k = 0;
        for (int p = 0; p < kold; ++p) {
            const int t = bcastN_i<NB>(ordold, p);
            bool okp;
            if constexpr (BIG) okp = (st.k >= S.kmax) ? try_append_big<NB>(S, bd, st, lam, t, lane, true) : try_append<NB>(S, bd, st, lam, t, lane, true);
            else okp = try_append<NB>(S, bd, st, lam, t, lane, true);
            if (!okp) {     // column became dependent: drop it
#pragma unroll
                for (int b = 0; b < NB; ++b) if (lane + 64 * b == t) st.x[b] = 0.0;
            }
        }
    }
    MET2_CYC_END(1, c_ref);
    MET2_CYC_ADD(12, 1);
    nnls_iterate<NB, ONE, BIG>(S, bd, st, lam, aug ? S.m + S.n : S.m, lane, true);
}

// D x  (lane e < m holds (D x)_e), using the passive set of st
template <int NB>
__device__ __forceinline__ double model_signal(const WaveShared &S, const NnlsState<NB> &st, int lane)
{
    const int k = st.k;
    // rows of D^T: coalesced, four in flight per step
    double acc = 0.0, acc2 = 0.0, xp[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { const double t = gatherN<NB>(st.x, st.ord[b]); xp[b] = (lane + 64 * b < k) ? t : 0.0; }   // as in dual()
    const unsigned ec = (unsigned)min(lane, S.m - 1);
#pragma clang loop unroll(disable)
    for (int p = 0; p < k; p += 4) {
        double v[4], xs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pp = p + q;
            v[q] = ld_row_sel(S.buffer_rows, S.Dt, bcastN_i<NB>(st.ord, pp) * S.dtstride, ec);
            xs[q] = bcastN<NB>(xp, pp);
        }
        acc = fma(v[0], xs[0], acc); acc2 = fma(v[1], xs[1], acc2);
        acc = fma(v[2], xs[2], acc); acc2 = fma(v[3], xs[3], acc2);
    }
    return acc + acc2;
}

// || D x - b ||^2
template <int NB>
__device__ __forceinline__ double sse_of(const WaveShared &S, const NnlsState<NB> &st, double b, int lane)
{
#ifdef MET2_CYCSTATS
    const unsigned long long c0 = __builtin_readcyclecounter();
#endif
    if (MET2_DOUBLE == 3) { const double r0 = model_signal<NB>(S, st, lane); asm volatile("" :: "v"(r0)); }
    double r = model_signal<NB>(S, st, lane) - b;
    r = (lane < S.m) ? r : 0.0;
    const double out = wave_sum(r * r);
#ifdef MET2_CYCSTATS
    const_cast<NnlsState<NB> &>(st).cyc[5] += __builtin_readcyclecounter() - c0;
#endif
    return out;
}

template <int NB>
__device__ __forceinline__ void project(const WaveShared &S, double bvec, int lane, double (&h)[NB]);

// One step of iterative refinement of the passive sub-problem with the residual taken from D itself (corrected semi-normal equations):
//     g = D^T (b - D x) - lam K x,    R^T R delta = g_P,    x_P += delta.
// The Gram-form solve leaves x with a relative error of ~cond(G_PP) eps (1e-10 .. 1e-8: B = D^T D is itself a rounded matrix); the
// correction computed from D brings it to the level of a QR solve of the augmented system (~cond(D_P) eps), which is what the
// reference's Lawson-Hanson works at.  Called only where a Brent comparison sits inside the Gram-form noise (fminbound_tie_dev): a plain,
// unpipelined forward substitution is good enough.
template <int NB, bool BIG = false>
__device__ __forceinline__ void refine_csne(const WaveShared &S, NnlsState<NB> &st, double lam, double bvec, int lane)
{
    const int k = st.k;
    if (k == 0) return;
    bool big = false;
    if constexpr (BIG) big = k > S.kmax;                               // columns beyond the LDS capacity live in the wave's global slot (nnls_big.hpp)
    double r = bvec - model_signal<NB>(S, st, lane);
    r = (lane < S.m) ? r : 0.0;
    double g[NB];
    project<NB>(S, r, lane, g);                                        // D^T (b - D x), bin-indexed
    if (lam != 0.0) {
        double kbl[NB][5], kx[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int d = 0; d < 5; ++d) kbl[b][d] = S.kband[d * 128 + lane + 64 * b];
        band_mul<NB>(kbl, st.x, lane, kx);
#pragma unroll
        for (int b = 0; b < NB; ++b) g[b] = fma(-lam, kx[b], g[b]);
    }
    double u[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { const double t = gatherN<NB>(g, st.ord[b]); u[b] = (lane + 64 * b < k) ? t : 0.0; }      // by position
    if (BIG && big) {
        const double *cpl[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) cpl[b] = big_col(S, min(lane + 64 * b, S.n - 1));
        for (int i = 0; i + 1 < k; ++i) {
            double t[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) t[b] = u[b] * st.rinv[b];
            const double si = bcastN<NB>(t, i);
#pragma unroll
            for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; if (pl > i && pl < k) u[b] = fma(-cpl[b][i], si, u[b]); }
        }
    } else
    for (int i = 0; i + 1 < k; ++i) {                                  // R^T u = g_P: row i of R is entry i of every later column
        double t[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) t[b] = u[b] * st.rinv[b];
        const double si = bcastN<NB>(t, i);
#pragma unroll
        for (int b = 0; b < NB; ++b) { const int pl = lane + 64 * b; if (pl > i && pl < k) u[b] = fma(-S.R[col_base(pl) + i], si, u[b]); }
    }
    double ysave[NB], dl[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { ysave[b] = st.y[b]; st.y[b] = (lane + 64 * b < k) ? u[b] * st.rinv[b] : 0.0; }
    if (BIG && big) back_subst_big<NB>(S, st, lane, dl);
    else back_subst<NB>(S, st, lane, dl);                              // R delta = u
#pragma unroll
    for (int b = 0; b < NB; ++b) st.y[b] = ysave[b];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const double d = gatherN<NB>(dl, max(st.pos[b], 0));           // by bin
        if (st.pos[b] >= 0) st.x[b] += d;
    }
}

// ---- the NOT-inlined instances of the solver with the spill-over legs compiled in (BIG = true), called by the spill-over voxel routine
// (fit_kernel.hpp: fit_voxel_spill).  Arguments of a device function arrive in vector registers: what is wave-uniform is made scalar again
// on entry; the factor's LDS region comes as an LDS-typed pointer beside the struct.
__device__ __forceinline__ void big_uniform(WaveShared &S)
{
    S.B = big_rfl(S.B); S.K = big_rfl(S.K); S.Dt = big_rfl(S.Dt); S.DtG = big_rfl(S.DtG); S.kband = big_rfl(S.kband); S.D = big_rfl(S.D);
    S.Rg = big_rfl(S.Rg);
    S.dtstride = big_rfl(S.dtstride); S.n = big_rfl(S.n); S.m = big_rfl(S.m); S.bstride = big_rfl(S.bstride); S.dstride = big_rfl(S.dstride);
    S.kmax = big_rfl(S.kmax); S.rcap = big_rfl(S.rcap); S.gbase = big_rfl(S.gbase);
    S.have_bdiag = big_rfl((int)S.have_bdiag) != 0; S.reorder = big_rfl((int)S.reorder) != 0; S.buffer_rows = big_rfl((int)S.buffer_rows) != 0;
}
template <int NB>
__device__ __forceinline__ void big_uniform(NnlsState<NB> &st)
{
    st.k = big_rfl(st.k); st.itmax_hit = big_rfl(st.itmax_hit);
#pragma unroll
    for (int b = 0; b < NB; ++b) st.P[b] = big_rfl(st.P[b]);
}
// flags: bit 0 = the augmented system's row count (aug), bit 1 = warm start
template <int NB, int ONE>
__device__ __attribute__((noinline)) void big_solve_fn(WaveShared S, big_lds_dp Rl, Band<NB> bd, NnlsState<NB> *stp, double lam, int flags)
{
    const int lane = big_lane();
    big_uniform(S);
    S.R = (double *)Rl;
    NnlsState<NB> st = *stp;
    big_uniform<NB>(st);
    lam = __hiloint2double(big_rfl(__double2hiint(lam)), big_rfl(__double2loint(lam)));
    flags = big_rfl(flags);
#ifdef MET2_BIGSTATS
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int k0 = st.k;
    st.itmax_hit &= ~4;
#endif
    if (flags & 2) nnls_solve_warm<NB, ONE, true>(S, bd, st, lam, (flags & 1) != 0, lane);
    else nnls_solve<NB, ONE, true>(S, bd, st, lam, (flags & 1) != 0, lane);
#ifdef MET2_BIGSTATS
    {
        const unsigned long long dt = __builtin_readcyclecounter() - t0;
        const bool big = (st.itmax_hit & 4) || k0 > S.kmax;
        MET2_BIGSTAT(0, 1);
        if (big) { MET2_BIGSTAT(1, 1); MET2_BIGSTAT(3, dt); MET2_BIGSTAT(7, st.k); MET2_BIGSTAT(8, k0); } else MET2_BIGSTAT(2, dt);
    }
#endif
    *stp = st;
}
template <int NB>
__device__ __attribute__((noinline)) void big_refine_fn(WaveShared S, big_lds_dp Rl, NnlsState<NB> *stp, double lam, double bvec)
{
    const int lane = big_lane();
    big_uniform(S);
    S.R = (double *)Rl;
    NnlsState<NB> st = *stp;
    big_uniform<NB>(st);
    lam = __hiloint2double(big_rfl(__double2hiint(lam)), big_rfl(__double2loint(lam)));
    refine_csne<NB, true>(S, st, lam, bvec, lane);
    *stp = st;
}

// h = D^T b for the bins a lane owns (bvec: lane e holds echo e)
template <int NB>
__device__ __forceinline__ void project(const WaveShared &S, double bvec, int lane, double (&h)[NB])
{
#pragma unroll
    for (int b = 0; b < NB; ++b) h[b] = 0.0;
    unsigned jc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) jc[b] = (unsigned)min(lane + 64 * b, S.n - 1);
#pragma clang loop unroll(disable)
    for (int e = 0; e < S.m; e += 4) {                     // four rows of D in flight per step
        double dv[4][NB], be[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ee = min(e + q, S.m - 1);
            const double *Drow = S.D + ee * S.dstride;
#pragma unroll
            for (int b = 0; b < NB; ++b) dv[q][b] = Drow[jc[b]];
            be[q] = (e + q < S.m) ? bcast(bvec, ee) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < NB; ++b) h[b] = fma(dv[q][b], be[q], h[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) h[b] = (lane + 64 * b < S.n) ? h[b] : 0.0;
}

// h_v = D^T b_v for V voxels at once: every row of D is loaded once and used V times (eight rows in flight)
template <int NB, int V>
__device__ __forceinline__ void project_multi(const WaveShared &S, const double (&bvec)[V], int lane, double (&h)[V][NB])
{
    unsigned jc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) jc[b] = (unsigned)min(lane + 64 * b, S.n - 1);
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int b = 0; b < NB; ++b) h[v][b] = 0.0;
#pragma clang loop unroll(disable)
    for (int e = 0; e < S.m; e += 8) {
        double dv[8][NB];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double *Drow = S.D + min(e + q, S.m - 1) * S.dstride;
#pragma unroll
            for (int b = 0; b < NB; ++b) dv[q][b] = Drow[jc[b]];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double be = (e + q < S.m) ? bcast(bvec[v], min(e + q, S.m - 1)) : 0.0;
#pragma unroll
                for (int b = 0; b < NB; ++b) h[v][b] = fma(dv[q][b], be, h[v][b]);
            }
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int b = 0; b < NB; ++b) h[v][b] = (lane + 64 * b < S.n) ? h[v][b] : 0.0;
}

// ||L x||^2 for the bin-indexed x (L as 5 diagonals)
template <int NB>
__device__ __forceinline__ double seminorm2(const Band<NB> &bd, const double (&x)[NB], int n, int lane)
{
    double lf[NB], s = 0.0;
    band_mul<NB>(bd.lb, x, lane, lf);
#pragma unroll
    for (int b = 0; b < NB; ++b) s += (lane + 64 * b < n) ? lf[b] * lf[b] : 0.0;
    return wave_sum(s);
}

} // namespace met2
