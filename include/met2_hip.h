/*
 * met2_hip.h -- C ABI of libmet2_hip.so, the MI355X (gfx950) implementation of the
 * per-voxel regularised-NNLS T2-spectrum path of ejcanalesr/multicomponent-T2-toolbox.
 *
 * The reference has no FFI layer: its boundary is Python function calls inside one
 * process.  Each entry point below names the reference function (or
 * driver code) it replaces; INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.  Plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - all arithmetic is fp64 (fused multiply-adds; the factorisation's reciprocal square roots are v_rsq_f64 plus two
 *     Newton steps, ~1 ulp; the lambda search and its objective use IEEE division and square root);
 *   - "host" pointers are ordinary process memory (small parameter arrays);
 *     "device" pointers are HIP device memory on the plan's device (bulk voxel arrays);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous with respect to the host unless stated otherwise;
 *   - every function returns 0 on success, a negative MET2_E_* code otherwise;
 *     met2_last_error() gives the message of the calling thread's last failure;
 *   - a plan may be used from one host thread at a time and serves one stream at a time (its sort scratch, error word and
 *     timing events are per plan: a fit or met2_plan_finish on another stream while enqueued fits are pending returns
 *     MET2_E_STATE); different plans may be used concurrently, on one device or on several;
 *   - multi-GPU: the library keeps no global state besides the calling thread's error message and
 *     what met2_fit_host parks with a plan, so "one plan per device (met2_options.device), each driven by its own host thread or
 *     process, all at once" works from any host language.  Two drivers are built on it:
 *       met2_fit_host (ABI 5, below)  ONE process, one host thread per plan inside the call, host arrays in and out -- what the
 *                                     reference's single Python process binds; no communicator, the outputs meet in host memory;
 *       dist.py                       one process per GPU under torch.distributed; the path's single collective -- the gather of
 *                                     the outputs to the root -- belongs to the host's communicator (RCCL).
 */
#ifndef MET2_HIP_H
#define MET2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MET2_ABI_VERSION 6

/* reg_method of motor/motor_recon_met2_real_data.py:134-150 */
enum met2_method {
    MET2_NNLS = 0,      /* algorithms.py:55  nnls                      reg_param = 0      */
    MET2_T2SPARC = 1,   /* algorithms.py:262 nnls_tik, lambda = 1.8    reg_param = 1.8    */
    MET2_X2 = 2,        /* algorithms.py:211 nnls_x2, factor = 1.02    reg_param = k_est  */
    MET2_LCURVE = 3,    /* algorithms.py:88  nnls_lcurve_wrapper + nnls_tik, reg_param = lambda */
    MET2_GCV = 4,       /* algorithms.py:276 nnls_gcv                  reg_param = lambda */
    MET2_BAYESREG = 5   /* bayesian_interpolation.py:84 BayesReg_nnls  reg_param = lambda */
};

/* reg_matrix of motor:254-273 */
enum met2_penalty { MET2_PEN_I = 0, MET2_PEN_L1 = 1, MET2_PEN_L2 = 2, MET2_PEN_INVT2 = 3 };

/* per-voxel status bits written by met2_fit */
enum met2_status {
    MET2_ST_FITTED = 1,        /* passed the gates of motor:124,131 and was solved                 */
    MET2_ST_ITMAX = 2,         /* some inner NNLS hit the 3n iteration cap (reference ignores it)  */
    MET2_ST_NONFINITE = 4,     /* NaN/Inf in the voxel's echoes: outputs zero (reference: ValueError) */
    MET2_ST_CHOLFAIL = 8,      /* BayesReg: Cholesky of beta(B + lambda K) failed (reference: LinAlgError) */
    MET2_ST_BRENT_MAXFUN = 16, /* lambda search stopped on maxfun                                  */
    MET2_ST_KOVERFLOW = 32     /* passive set outgrew the wave's LDS region: set only transiently -- the fit kernel queues such a voxel and
                                  the spill-over kernel behind it solves it again, so no voxel returned by met2_fit carries it */
};

enum met2_error {
    MET2_OK = 0,
    MET2_E_INVALID = -1,       /* bad argument (shape, NULL, enum)        */
    MET2_E_UNSUPPORTED = -2,   /* shape/penalty outside the built kernels */
    MET2_E_HIP = -3,           /* a HIP runtime call failed               */
    MET2_E_NODEVICE = -4,      /* no gfx950 device visible                */
    MET2_E_STATE = -5          /* plan not fully configured for this call, or fits pending on another stream */
};

typedef struct met2_plan met2_plan;

/* Constants the reference buries in its driver; defaults = reference values. */
typedef struct met2_options {
    int32_t struct_size;      /* sizeof(met2_options), for ABI growth                      */
    int32_t device;           /* HIP device ordinal                                        */
    double x2_factor;         /* motor:141            1.02                                 */
    double t2sparc_lambda;    /* motor:138            1.8                                  */
    double brent_xtol;        /* algorithms.py:219    1e-5                                 */
    int32_t brent_maxfun;     /* 0 = reference value: 300 (X2, GCV), 200 (BayesReg)        */
    int32_t reserved0;
    double t2_myelin_cut;     /* motor:216 (myelin_T2 CLI flag)   40.0                     */
    double t2_ie_cut;         /* motor:217            200.0                                */
    /* ABI 6: the intervals of the lambda searches (scipy.optimize.fminbound's x1, x2).  A caller that passes a shorter struct (struct_size of
     * ABI <= 5) gets the reference values.  0 <= lo < hi, finite.  The plan-level seeds and BayesReg's factor tables follow them.  */
    double x2_lo, x2_hi;      /* algorithms.py:219              0, 10     (the reference's evaluation scripts search wider grids: :215 of
                                                                          evaluate_all_methods_two_lobes_SNR50_150.py tops at 100)          */
    double gcv_lo, gcv_hi;    /* algorithms.py:280              1e-8, 10  */
    double bayes_lo, bayes_hi;/* bayesian_interpolation.py:101  1e-8, 2   */
} met2_options;

void met2_default_options(met2_options *opt);

int met2_abi_version(void);
int met2_device_count(void);
const char *met2_last_error(void);

/* ---- plan: the shared nTE x nT2 x nFA problem (dictionary, Gram matrices, penalty) ---- */
int met2_plan_create(met2_plan **out, int32_t n_te, int32_t n_t2, int32_t n_fa, const met2_options *opt);
int met2_plan_destroy(met2_plan *plan);
/* change x2_factor / t2sparc_lambda / Brent settings / metric windows of an existing plan
 * (the per-call arguments `factor` of nnls_x2 and `reg_opt` of nnls_tik, algorithms.py:211,262) */
int met2_plan_set_options(met2_plan *plan, const met2_options *opt);
int met2_plan_get_options(met2_plan *plan, met2_options *opt);
/* the shape the plan was created with (any of the three may be NULL) */
int met2_plan_get_shape(met2_plan *plan, int32_t *n_te, int32_t *n_t2, int32_t *n_fa);

/* epg/epg.py:155 create_Dic_3D -- EPG dictionary built on the device, one wave per
 * (T2, flip angle).  T2s/T1s [n_t2], alpha_deg [n_fa] are host arrays.  Also forms the
 * per-flip-angle Gram matrices D^T D used by the solver. */
int met2_plan_build_dictionary_epg(met2_plan *plan, const double *T2s, const double *T1s, double tau,
                                   const double *alpha_deg, double TR, void *stream);

/* Dictionary handed in by the caller, reference layout Dic_3D[n_te][n_t2][n_fa] (host).
 * Replaces the `Dic_3D` argument of fitting_slice_T2 (motor:113). */
int met2_plan_set_dictionary(met2_plan *plan, const double *dic3d_host);

/* Copies the dictionary back in the reference layout [n_te][n_t2][n_fa] (host, blocking). */
int met2_plan_get_dictionary(met2_plan *plan, double *dic3d_host);

/* motor:86 create_Laplacian_matrix / motor:263 InvT2.  T2s (host, [n_t2]) is needed for
 * InvT2 only.  The dense form takes any `Laplac` [n_t2][n_t2] (host) whose L^T L has
 * bandwidth <= 2 (true for I, L1, L2, InvT2); wider ones return MET2_E_UNSUPPORTED.
 * The plan-level seeds of the lambda searches are (re)built here and wherever the dictionary changes (these entries block);
 * they are used only if D^T D + lambda L^T L is positive definite (checked for flip angle 0) -- with a penalty whose null
 * space meets the dictionary's the minimiser is not unique and every voxel starts cold, as the reference does. */
int met2_plan_set_penalty(met2_plan *plan, int32_t which, const double *T2s);
int met2_plan_set_penalty_dense(met2_plan *plan, const double *laplac_host);
int met2_plan_get_penalty(met2_plan *plan, double *laplac_host);

/* motor:248-251 lambda_reg (host, [n]); default = the reference's 50-point grid */
int met2_plan_set_lambda_grid(met2_plan *plan, const double *lambda_reg, int32_t n);

/* motor:215-224 T2 grid for the metrics windows (host, [n_t2]); set automatically by
 * met2_plan_build_dictionary_epg */
int met2_plan_set_t2_grid(met2_plan *plan, const double *T2s);

/* ---- hot path ------------------------------------------------------------------------
 * motor:113 fitting_slice_T2 over a flat voxel list, fused with the step-4 metrics of
 * motor:443-472.  All array arguments are DEVICE pointers:
 *   data     [nvox][n_te]  echoes (un-normalised, as the driver passes them)
 *   fa_index [nvox]        float64 index into the dictionary's FA axis (motor:127), NULL = 0
 *   mask     [nvox]        uint8, voxel is fitted iff mask != 0 (motor:124), NULL = all ones
 *   fsol     [nvox][n_t2]  out: x * km                       (motor:154)
 *   sig      [nvox][n_te]  out: Kernel @ x * km              (motor:155), may be NULL
 *   reg      [nvox]        out: reg_param                    (motor:153)
 *   lam      [nvox]        out: the selected lambda (equals reg except for X2, where the driver
 *                               stores k_est in reg_param, motor:141-143), may be NULL
 *   maps     [6][nvox]     out: MWF, IEWF, FWF, T2_M, T2_IE, TWC (motor:455-468), may be NULL
 *   status   [nvox]        out: met2_status bits, may be NULL
 * Gated-out voxels get zeros (motor:115-117) and, if mask != 0, the all-zero-spectrum
 * metrics of motor:448-468. */
int met2_fit(met2_plan *plan, int32_t method, int64_t nvox, const double *data, const double *fa_index,
             const uint8_t *mask, double *fsol, double *sig, double *reg, double *lam, double *maps,
             int32_t *status, void *stream);

/* The same with an explicit layout of `data`: echo e of voxel v is read at data[v * voxel_stride + e * echo_stride]
 * (strides in doubles).  The driver loads its volume with nibabel (motor:167-173), whose arrays are Fortran-ordered:
 * for such an [nx][ny][nz][nt] array voxel_stride = 1 and echo_stride = nx*ny*nz, and voxel v is the voxel at
 * x + nx*(y + ny*z) -- the outputs then come out in that (Fortran) voxel order, i.e. fsol is the Fortran-ordered
 * [nx][ny][nz] volume of spectra.  met2_fit is the special case voxel_stride = n_te, echo_stride = 1.  The classify pass
 * and the solver read every echo exactly once either way; no transposed copy is made. */
int met2_fit_strided(met2_plan *plan, int32_t method, int64_t nvox, const double *data, int64_t voxel_stride,
                     int64_t echo_stride, const double *fa_index, const uint8_t *mask, double *fsol, double *sig, double *reg,
                     double *lam, double *maps, int32_t *status, void *stream);

/* The same without waiting for the GPU: the launches are enqueued on `stream` and the call returns (the blocking entries above
 * wait only to report an FA index outside the dictionary, the reference's IndexError at motor:127-128).  A host pipeline --
 * H2D of the next chunk of a volume, this fit, D2H of the previous chunk's outputs, on separate streams (motor:167-182,
 * :427-503 are that loop in the reference, one image row at a time) -- enqueues chunk after chunk and calls met2_plan_finish
 * once: it waits for `stream` and returns MET2_E_INVALID if any fit enqueued since the last finish saw such an index.
 * One plan serves one stream at a time (its sort scratch is per plan): enqueueing on, or finishing, another stream while fits are pending
 * returns MET2_E_STATE. */
int met2_fit_enqueue_strided(met2_plan *plan, int32_t method, int64_t nvox, const double *data, int64_t voxel_stride,
                             int64_t echo_stride, const double *fa_index, const uint8_t *mask, double *fsol, double *sig,
                             double *reg, double *lam, double *maps, int32_t *status, void *stream);
int met2_plan_finish(met2_plan *plan, void *stream);

/* ---- host to host, one or several devices (ABI 5; run-wise dealing since ABI 6) ----------
 * motor:349-373 + motor:427-472 for a voxel list that lives in HOST memory, as the reference's driver holds it (motor:167-182), from ONE
 * process: what a binding of the reference calls instead of its joblib loop over image rows (motor:427-441) -- numpy arrays in, numpy
 * arrays out, no device memory, stream or communicator on the caller's side.
 *   plans [n_plans]   1..64 distinct plans of one shape, configured alike (dictionary, penalty, options), each on the device of its
 *                     met2_options.device; several plans may share a device.  With several plans the voxel list is dealt in RUNS of
 *                     4 096 voxels, run j -> plan j mod n_plans (interleaved: tissue classes cluster in space and differ 10x in
 *                     iteration count; ABI 6 -- ABI 5 dealt whole blocks), and a plan's DMA block is `chunk`
 *                     voxels of ITS runs, moved by pitched copies; rows with a pitch (echo_stride 1, voxel_stride > n_te) and general
 *                     strides keep whole blocks, block b -> plan b mod n_plans.  Every plan is driven by its own host thread inside
 *                     the call (one plan: the calling thread) through three streams of its device: H2D of its block c + 1 |
 *                     [FA estimation and] fit of block c | D2H of block c - 1.  There is no exchange between devices.  The coarse plans
 *                     of estimate_fa = 2 must be distinct, one per plan, and none of them in plans[].
 *   ALL array arguments are HOST pointers (data and fa_data may ALSO be device pointers: the volume as met2_tv_chambolle / met2_nesma /
 *   met2_smooth_separable left it -- then copied block by block device to device); arrays in pinned memory (hipHostMalloc,
 *   hipHostRegister) are copied from / to in place, pageable ones are staged through pinned block buffers by the plan's thread while its
 *   device works:
 *   data              echo e of voxel v at data[v * voxel_stride + e * echo_stride] (strides in doubles, > 0): [nvox][n_te] rows
 *                     (echo_stride 1) and the Fortran-ordered volume of nibabel (voxel_stride 1, echo_stride nvox) are copied as they
 *                     lie and read in place on the device; any other layout is gathered on the host
 *   fa_data           NULL, or the same voxel list (same layout and strides) as the FA estimation shall see it: the Gaussian-smoothed
 *                     volume of motor:337-343 (FA_smooth='yes', the CLI default); needs estimate_fa != 0
 *   mask_values       NULL, or [nvox] float64: the driver's preparation on the device -- every echo of voxel v is multiplied by
 *                     mask_values[v] and negative values are clipped to 0 (motor:180-182, :279) before anything else sees the block
 *                     (fa_data, when given, is taken as prepared already)
 *   fa_index, mask    [nvox] float64 / uint8 as for met2_fit, NULL = flip angle 0 / all ones
 *   estimate_fa       0: the flip angles are given (fa_index, or angle 0);  fa_index must be NULL otherwise;
 *                     1: brute-force search over the plans' FA axis on every block (met2_fa_bruteforce, fa_estimation.py:74-111);
 *                     2: the spline method (fa_estimation.py:35-70, the CLI default): plain-NNLS residuals on the coarse grid of the
 *                        plan attached with met2_plan_attach_fa_spline, a cubic spline through them, its bounded minimum snapped to the
 *                        plans' own FA axis (met2_fa_bruteforce on the coarse plan + met2_fa_spline_select, per block)
 *   fsol [nvox][n_t2], sig [nvox][n_te], reg, lam [nvox], maps [6][nvox], status [nvox]   as for met2_fit (sig, lam, maps, status may be NULL)
 *   fa_out [nvox]     out, may be NULL: the FA index every voxel was fitted with
 *   fa_gate [nvox]    out, may be NULL: 1.0 where the FA step's gate holds (fa_estimation.py:45: mask and a positive echo sum of what
 *                     the FA step sees), else 0.0 -- the driver reports a flip angle only there (motor:366-370)
 *   chunk             voxels per DMA block (several plans: rounded up to whole runs of 4 096); 0 = a quarter of a plan's share, in multiples of
 *                     4 096, at most 262 144 and (unless the share itself is smaller) at least 65 536
 *   plan_ms [n_plans] out, may be NULL: wall-clock ms every plan's thread spent in the call
 * Blocking.  Every voxel is solved on its own, so the outputs are bit for bit those of one met2_fit over the whole list, whatever
 * n_plans, chunk and the devices.  Returns the first failing plan's code (an FA index outside the dictionary: MET2_E_INVALID) after
 * ALL plans' streams have drained -- nothing writes to the caller's arrays after the return.  The block buffers (two slots of
 * chunk x ~8 (2 n_te + n_t2 + 10) bytes on the device, the same pinned when a pageable array takes part), three streams and six
 * events stay with each plan until met2_plan_destroy (and then wait for the next plan on that device: met2_host_trim). */
int met2_fit_host(met2_plan *const *plans, int32_t n_plans, int32_t method, int64_t nvox, const double *data, const double *fa_data,
                  int64_t voxel_stride, int64_t echo_stride, const double *mask_values, const double *fa_index, const uint8_t *mask,
                  int32_t estimate_fa, double *fsol, double *sig, double *reg, double *lam, double *maps, int32_t *status, double *fa_out,
                  double *fa_gate, int64_t chunk, double *plan_ms);
/* For estimate_fa = 2: `plan_lr` holds the coarse-grid dictionary (motor:237-238: 15 flip angles from 90 to 180 degrees), same n_te x n_t2
 * and device as `plan`; alpha_lr [n_lr = its flip angles] and alpha_hr [n_hr = the plan's flip angles] are the two grids in degrees (HOST
 * arrays, copied).  plan_lr must outlive the attachment; plan_lr = NULL detaches. */
int met2_plan_attach_fa_spline(met2_plan *plan, met2_plan *plan_lr, int32_t n_lr, const double *alpha_lr, int32_t n_hr, const double *alpha_hr);
/* met2_plan_destroy hands a plan's block buffers of met2_fit_host to the next plan created on the same device (at most two sets per device
 * wait; a driver that builds its plans per call then allocates and pins nothing per call); this frees the waiting ones. */
int met2_host_trim(void);

/* Test/diagnostic entry: `method` = 10 + MET2_X2 / MET2_GCV / MET2_BAYESREG passed to met2_fit
 * evaluates that method's lambda-selection objective (algorithms.py:226-233, :285-296,
 * bayesian_interpolation.py:107-126) on the plan's lambda grid (n <= n_t2 points) and stores the
 * values in fsol[v][0..n); sig, lam and maps are not written. */
#define MET2_OBJECTIVE_GRID 10

/* flip_angle_algorithms/fa_estimation.py:74-111 (brute force over the plan's FA axis).
 * DEVICE pointers: data [nvox][n_te] (un-normalised), mask [nvox] (NULL = ones);
 * out fa_index [nvox] float64 (0 where gated out), km [nvox] = sum(f) at the best FA
 * (may be NULL), resid [nvox][n_fa] NNLS residual norms (may be NULL). */
int met2_fa_bruteforce(met2_plan *plan, int64_t nvox, const double *data, const uint8_t *mask,
                       double *fa_index, double *km, double *resid, void *stream);

int met2_fa_bruteforce_strided(met2_plan *plan, int64_t nvox, const double *data, int64_t voxel_stride, int64_t echo_stride,
                               const uint8_t *mask, double *fa_index, double *km, double *resid, void *stream);

/* flip_angle_algorithms/fa_estimation.py:54-59, the selection step of the spline FA method (the CLI default,
 * run_real_data_script.py:34): given the plain-NNLS residual norms on a coarse FA grid (`resid` from
 * met2_fa_bruteforce on a plan built with the coarse grid, motor:237-238), interpolate them with a cubic
 * spline (scipy interp1d(kind='cubic')), minimise over [90, 180] with the bounded Brent of
 * scipy minimize_scalar(method='Bounded') and snap to the fine grid.  alpha_lr [n_lr] and alpha_hr [n_hr] are
 * host arrays; resid [nvox][n_lr], data [nvox][n_te], mask are DEVICE pointers (data/mask only gate voxels,
 * fa_estimation.py:45); out fa_index [nvox] float64 index into alpha_hr, xmin [nvox] the continuous minimiser
 * (may be NULL).  Blocking. */
int met2_fa_spline_select(int32_t device, int64_t nvox, int32_t n_lr, const double *alpha_lr, const double *resid,
                          int32_t n_hr, const double *alpha_hr, int32_t n_te, const double *data, const uint8_t *mask,
                          double *fa_index, double *xmin, void *stream);

int met2_fa_spline_select_strided(int32_t device, int64_t nvox, int32_t n_lr, const double *alpha_lr, const double *resid,
                                  int32_t n_hr, const double *alpha_hr, int32_t n_te, const double *data, int64_t voxel_stride,
                                  int64_t echo_stride, const uint8_t *mask, double *fa_index, double *xmin, void *stream);

/* motor:337-343, the Gaussian pre-smoothing of the FA step (FA_smooth='yes', the CLI default): the reference runs
 * scipy.ndimage.gaussian_filter(volume, 2.0) on every echo volume.  This is the separable filter behind it: one pass per
 * spatial axis with the symmetric kernel weights[0 .. 2 radius] (HOST array, weights[radius] the centre; for the
 * reference: radius = int(4 sigma + 0.5), weights = exp(-x^2 / (2 sigma^2)) normalised to sum 1), boundary mode 'reflect',
 * accumulated in scipy's order, so the result is bit-identical to gaussian_filter.  DEVICE pointers: data and out
 * [nx][ny][nz][n_te], work the same size (NULL: allocated and freed inside, which makes the call blocking); all distinct.
 * radius <= 32. */
int met2_smooth_separable(int32_t device, int32_t nx, int32_t ny, int32_t nz, int32_t n_te, int32_t radius, const double *weights,
                          const double *data, double *out, double *work, void *stream);

/* motor:305-333, the NESMA filter (denoise='NESMA').  DEVICE pointers: data [nx][ny][nz][n_te] (already
 * multiplied by the mask and clipped at 0, motor:180-182 and :279), mask [nx][ny][nz] uint8 -- voxels with
 * mask == 1 are filtered (motor:317), all others get zeros (NULL = filter every voxel); out [nx][ny][nz][n_te],
 * must not alias data.  Each filtered voxel becomes the mean of the voxels in its half-open window
 * [x-6, x+6) x [y-6, y+6) x [z-6, z+6) (clipped to the volume) whose relative L1 distance to it is < 2.5 %
 * (nan when none qualifies, e.g. an all-zero signal, like np.mean of an empty selection).  n_te <= 128.
 * Asynchronous on `stream`. */
int met2_nesma(int32_t device, int32_t nx, int32_t ny, int32_t nz, int32_t n_te, const double *data,
               const uint8_t *mask, double *out, void *stream);

/* motor:293-304, TV denoising (denoise='TV'; the reference's example pipeline runs it, example_script_run_MET2_preproc_and_recon.sh:54):
 *     for every echo volume:  sigma_est = mean(estimate_sigma(vol));  vol <- denoise_tv_chambolle(vol, weight = 2 sigma_est, eps = 2e-4,
 *                                                                                                 max_num_iter = 200)
 * (scikit-image's functions, restated from the published algorithms: Donoho-Johnstone's db2 median estimator and Chambolle's
 * projection algorithm in 3-D).  ALL n_te echo volumes go through every step in the same launches; the stopping rule is applied
 * on the device per echo, the host reads nothing per iteration.
 * DEVICE pointers: data and out, both [nx][ny][nz][n_te] (echo_major = 0, the C-ordered array of the driver) or both [n_te][nz][ny][nx]
 * (echo_major = 1: the memory order of the Fortran-ordered array nibabel hands the driver, motor:167-173); out may alias data.
 * HOST: weight [n_te] = the `weight` of denoise_tv_chambolle per echo, or NULL: weight = weight_factor x the echo's estimated sigma
 * (the reference: weight_factor = 2).  An echo whose weight is not a positive finite number is copied through (an all-zero volume
 * has sigma = 0; scikit-image would return nan for it).
 * eps, max_num_iter: the reference passes 2e-4 and 200.  poll_every: 0 = every iteration of max_num_iter is enqueued and the call
 * returns without waiting (launches for echoes that have converged return at once); k > 0 = the host looks at the device's flags
 * every k iterations -- one batch of k behind the stream, so that the stream never waits for the host -- and stops enqueueing when
 * every echo has converged (the call then blocks until about that point; up to 2k launches that return at once are enqueued past it).
 * DEVICE out, may be NULL: sigma [n_te] the estimated noise level per echo (nan if the echo holds a nan or inf -- the reference's
 * finite check would raise), iters [n_te] int32 Chambolle iterations executed per echo.
 * work: DEVICE scratch of met2_tv_work_bytes() bytes (7 working copies of the volume), or NULL: allocated and freed inside, which
 * makes the call blocking.  n_te <= 127 for echo_major = 0. */
int64_t met2_tv_work_bytes(int32_t nx, int32_t ny, int32_t nz, int32_t n_te, int32_t echo_major);
int met2_tv_chambolle(int32_t device, int32_t nx, int32_t ny, int32_t nz, int32_t n_te, const double *data, int32_t echo_major,
                      const double *weight, double weight_factor, double eps, int32_t max_num_iter, int32_t poll_every, double *out,
                      double *sigma, int32_t *iters, void *work, int64_t work_bytes, void *stream);
/* For reports: HIP-event time (ms) from the first to the last Chambolle launch of the calling thread's most recent
 * met2_tv_chambolle (blocks until they finished) and the number of iterations that were enqueued. */
int met2_tv_last_timing(double *iter_ms, int32_t *launches);

/* motor/motor_recon_met2_real_data_ROI.py:405-420, the reduction of the ROI mode: for every ROI the mean signal over its
 * voxels and the mean EPG kernel, each voxel contributing the dictionary slice of its own flip angle
 * (total_signal / nv, total_Kernel / nv).  `src` holds the dictionary the flip angles index; `dst` is a plan of the same
 * n_te x n_t2 with n_fa = number of ROIs: its dictionary (and Gram matrices) become the per-ROI mean kernels, so that
 * met2_fit(dst, MET2_X2, nroi, mean_signal, fa_index = 0..nroi-1, ...) is the per-ROI fit of :419.
 * DEVICE pointers: data (echo e of voxel v at data[v * voxel_stride + e * echo_stride]), roi_index [nvox] int32 ROI
 * ordinal 0..nroi-1 (negative: voxel belongs to no ROI), fa_index [nvox] float64 (NULL = 0), out mean_signal
 * [nroi][n_te], out count [nroi] float64 voxels per ROI (0 -> that ROI's outputs are nan, as in the reference).
 * Deterministic (fixed summation order).  Blocking. */
int met2_roi_reduce(met2_plan *src, met2_plan *dst, int64_t nvox, const double *data, int64_t voxel_stride, int64_t echo_stride,
                    const int32_t *roi_index, const double *fa_index, double *mean_signal, double *count, void *stream);

/* motor:443-472 alone (fsol already on the device). */
int met2_metrics(met2_plan *plan, int64_t nvox, const double *fsol, const uint8_t *mask, double *maps,
                 void *stream);

/* Duration in ms of the solver kernel of the most recent met2_fit / met2_fa_bruteforce on
 * this plan, measured with HIP events on the launch stream (blocks until it finished). */
int met2_plan_last_kernel_ms(met2_plan *plan, double *ms);
/* NNLS/T2SPARC/X2/L-curve/GCV fits (and BayesReg at n_t2 > 64) give every wave an LDS region for a Cholesky factor of a reduced
 * passive-set capacity (the largest that lets 16 waves share a CU's LDS, never below 0.6 n_t2: 50 at n_t2 = 60; at two bins per
 * lane the largest that lets the 8 waves those kernels are compiled for share it: 71 at n_t2 = 120).  A voxel whose set outgrows
 * it (~1 % at 32 x 60, 5-10 % at 48 x 120) goes on in place with the factor's columns beyond the capacity in a per-wave slot in
 * device memory (allocated by the first such fit on the plan: 18 MB at n_t2 = 60, 77 MB at 120): ONE solver launch per fit.
 * met2_plan_last_spill_count: how many voxels of the most recent finished fit took that route (for reports and tests).
 * met2_plan_last_second_pass_ms: rounds 1-4 solved those voxels again in a second launch at full capacity; that ladder runs
 * only with MET2_TWO_PASS=1 in the environment (A/B test switch), otherwise this returns 0. */
int met2_plan_last_spill_count(met2_plan *plan, int64_t *count);
int met2_plan_last_second_pass_ms(met2_plan *plan, double *ms);

/* Launch geometry of the solver kernel (for reports): workgroups, threads per workgroup,
 * dynamic LDS bytes per workgroup. */
int met2_plan_launch_info(met2_plan *plan, int32_t method, int32_t *grid, int32_t *block, int32_t *lds_bytes);

/* Which form of the GCV trace (algorithms.py:285-296) met2_fit takes on this plan (for reports and tests): low_rank = 1 when every
 * flip angle's dictionary is of numerical rank <= 16 -- an orthonormal basis of 16 vectors leaves less than 1e-9 of its largest
 * column, `residual` = the largest such remainder over the flip angles -- and the trace is taken from the 17 x 17 matrix in that
 * basis (true for EPG dictionaries: ~1e-10 at 48 x 120, ~1e-12 at 32 x 60); 0: from the (n_te + 1) x (n_te + 1) matrix. */
int met2_plan_gcv_form(met2_plan *plan, int32_t *low_rank, double *residual);

#ifdef __cplusplus
}
#endif
#endif /* MET2_HIP_H */
