// met2_host.hip -- met2_fit_host (ABI 5): driver steps 2-4 (motor:349-373, 427-472) HOST TO HOST on one or several devices.
//
// The reference holds its volume in host memory (motor:167-182) and hands fitting_slice_T2 one image row at a time from one Python process
// (motor:427-441).  This entry is that loop for callers who own nothing but host arrays: the voxel list is cut into blocks, block b goes to
// plan b mod n_plans (interleaved: background, CSF and white matter cluster in space and differ 10x in iteration count),
// and every plan is driven by its own host thread through a three-stream pipeline on ITS device -- H2D of block c + 1 | [brute-force FA
// estimation and] the fit of block c | D2H of block c - 1 -- built from the library's own asynchronous entries (met2_fa_bruteforce_strided,
// met2_fit_enqueue_strided, met2_plan_finish).  There is no exchange between devices: every voxel is solved on its own, the outputs land in
// the caller's host arrays at the voxel's own index, so the result is bit for bit that of ONE met2_fit over the whole list whatever the
// number of plans, the block size or the mix of devices.  Host arrays in pinned (hipHostMalloc / hipHostRegister) memory are copied from and to
// in place; pageable ones are staged through pinned block buffers by the plan's thread while the device works.
//
// No kernels in this file: it is the host side of the boundary, in C++ because the boundary is a C ABI (no torch, no Python).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/met2_hip.h"
#include "abi_common.hpp"

namespace met2 { __attribute__((visibility("hidden"))) void spline_tables_release(); }

namespace {

// what one plan keeps between calls for this entry: streams, events, two block slots on the device and (when some host array is pageable)
// two pinned staging slots.  Grown on demand, freed by met2_plan_destroy (met2::host_release).
struct Slot {
    char *dev = nullptr;      // device slab
    char *pin = nullptr;      // pinned slab, same layout
    hipEvent_t ev_in = nullptr, ev_fit = nullptr, ev_out = nullptr;
};
struct Work {
    int device = -1;
    int64_t cap = 0;          // voxels per slot
    int nte = 0, nt2 = 0;
    int nlr = 0;              // the slots hold a [cap][nlr] residual table (spline FA method)
    bool fa_in = false;       // ... and a second input block (the volume the FA step sees)
    bool pinned = false;
    // the spline FA method's configuration (met2_plan_attach_fa_spline)
    met2_plan *plan_lr = nullptr;
    std::vector<double> alpha_lr, alpha_hr;
    hipStream_t s_in = nullptr, s_fit = nullptr, s_out = nullptr;
    Slot slot[2];
};

std::mutex g_work_mutex;
std::map<met2_plan *, Work *> g_work;

// slab layout of a slot with capacity `cap` voxels (byte offsets, every array 16-byte aligned)
struct Layout {
    size_t in, fa, fsol, sig, reg, lam, maps, status, mk, in_fa, resid, total;
    Layout(int64_t cap, int nte, int nt2, int nlr, bool fa_in)
    {
        size_t o = 0;
        auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 15) & ~(size_t)15; return at; };
        in = take(sizeof(double) * (size_t)cap * nte);
        fa = take(sizeof(double) * (size_t)cap);
        fsol = take(sizeof(double) * (size_t)cap * nt2);
        sig = take(sizeof(double) * (size_t)cap * nte);
        reg = take(sizeof(double) * (size_t)cap);
        lam = take(sizeof(double) * (size_t)cap);
        maps = take(sizeof(double) * (size_t)cap * 6);
        status = take(sizeof(int32_t) * (size_t)cap);
        mk = take((size_t)cap);
        in_fa = take(fa_in ? sizeof(double) * (size_t)cap * nte : 0);
        resid = take(sizeof(double) * (size_t)cap * nlr);
        total = o;
    }
};

void free_work(Work *w)
{
    if (!w) return;
    DevGuard guard(w->device);
    for (Slot &s : w->slot) {
        if (s.dev) (void)hipFree(s.dev);
        if (s.pin) (void)hipHostFree(s.pin);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_fit) (void)hipEventDestroy(s.ev_fit);
        if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    }
    if (w->s_in) (void)hipStreamDestroy(w->s_in);
    if (w->s_fit) (void)hipStreamDestroy(w->s_fit);
    if (w->s_out) (void)hipStreamDestroy(w->s_out);
    delete w;
}

// is this host pointer in memory the device can DMA from / to directly?
bool is_pinned(const void *ptr)
{
    if (!ptr) return true;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    const hipError_t e = hipPointerGetAttributes(&at, ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }      // ordinary (pageable) memory is unknown to the runtime
    return at.type == hipMemoryTypeHost;
}

// dst <- src on the calling thread plus up to three helpers (a block of 262 144 voxels is 67 MB in and 208 MB out; one thread's
// memcpy would stand next to a 37 ms fit)
void host_copy(void *dst, const void *src, size_t bytes)
{
    if (bytes < ((size_t)8 << 20)) { memcpy(dst, src, bytes); return; }
    const int parts = 4;
    const size_t step = ((bytes / parts) + 63) & ~(size_t)63;
    std::thread helpers[parts - 1];
    for (int i = 1; i < parts; ++i) {
        const size_t lo = std::min(bytes, step * i), hi = std::min(bytes, step * (i + 1));
        helpers[i - 1] = std::thread([=] { if (hi > lo) memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    memcpy(dst, src, std::min(bytes, step));
    for (auto &h : helpers) h.join();
}

struct Job {
    met2_plan *const *plans; int n_plans; int method; int64_t nvox;
    const double *data; const double *fa_data; int64_t vs, es;
    const double *fa_index; const uint8_t *mask; int estimate_fa;
    double *fsol, *sig, *reg, *lam, *maps; int32_t *status; double *fa_out;
    int64_t chunk; int nte, nt2; bool split;
    // which host arrays the device reaches directly
    bool pin_data, pin_fa_data, pin_fa, pin_mask, pin_fsol, pin_sig, pin_reg, pin_lam, pin_maps, pin_status, pin_fa_out;
    int in_case;              // 0: voxel-major rows (es == 1), 1: echo-major (vs == 1), 2: general strides (gathered on the host)
};

int ensure_work(met2_plan *plan, int device, const Job &J, bool need_pin, Work **out)
{
    Work *w;
    {
        std::lock_guard<std::mutex> lock(g_work_mutex);
        Work *&slot = g_work[plan];
        if (!slot) { slot = new Work(); slot->device = device; }
        w = slot;
    }
    USE_DEVICE(device);
    if (!w->s_in) {
        HIPCHK(hipStreamCreateWithFlags(&w->s_in, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&w->s_fit, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&w->s_out, hipStreamNonBlocking));
        for (Slot &s : w->slot) {
            HIPCHK(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&s.ev_fit, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
        }
    }
    const int need_lr = J.estimate_fa == 2 ? (int)w->alpha_lr.size() : 0;
    const bool need_fa_in = J.fa_data != nullptr;
    const bool grow = w->cap < J.chunk || w->nte != J.nte || w->nt2 != J.nt2 || w->nlr < need_lr || (need_fa_in && !w->fa_in);
    if (grow) {
        for (Slot &s : w->slot) {
            if (s.dev) { HIPCHK(hipFree(s.dev)); s.dev = nullptr; }
            if (s.pin) { HIPCHK(hipHostFree(s.pin)); s.pin = nullptr; }
        }
        w->pinned = false;
        w->cap = std::max(w->cap, J.chunk); w->nte = J.nte; w->nt2 = J.nt2; w->nlr = std::max(w->nlr, need_lr); w->fa_in = w->fa_in || need_fa_in;
        const Layout L(w->cap, w->nte, w->nt2, w->nlr, w->fa_in);
        for (Slot &s : w->slot) HIPCHK(hipMalloc((void **)&s.dev, L.total));
    }
    if (need_pin && !w->pinned) {
        const Layout L(w->cap, w->nte, w->nt2, w->nlr, w->fa_in);
        for (Slot &s : w->slot) HIPCHK(hipHostMalloc((void **)&s.pin, L.total, hipHostMallocDefault));
        w->pinned = true;
    }
    *out = w;
    return MET2_OK;
}

// one plan's share of the job: blocks t, t + n_plans, t + 2 n_plans, ...
int pipeline(const Job &J, int t, Work *w)
{
    met2_plan *plan = J.plans[t];
    const int nte = J.nte, nt2 = J.nt2;
    const int64_t chunk = J.chunk;
    const int64_t nblocks = (J.nvox + chunk - 1) / chunk;
    // this plan's pieces of the voxel list: its blocks, with the first one split 1/8 + 7/8 and the last one 7/8 + 1/8 -- the upload of the
    // very first piece and the download of the very last one are the only copies that do not run under a fit, so they are made short
    // (configs[1], one plan, four blocks of 262 144: 166.7 -> see DESIGN section 7; a piece costs ~1.9 ms of queue tail and sort passes)
    std::vector<std::pair<int64_t, int64_t>> seg;           // (first voxel, voxels)
    for (int64_t b = t; b < nblocks; b += J.n_plans) seg.emplace_back(b * chunk, std::min<int64_t>(chunk, J.nvox - b * chunk));
    if (J.split && !seg.empty()) {
        auto eighth = [](int64_t n) { return n >= 32768 ? std::max<int64_t>(4096, (n / 8) & ~(int64_t)4095) : 0; };
        {   const auto last = seg.back(); const int64_t e = eighth(last.second);
            if (e) { seg.back() = {last.first, last.second - e}; seg.emplace_back(last.first + last.second - e, e); } }
        {   const auto first = seg.front(); const int64_t e = eighth(first.second);
            if (e) { seg.front() = {first.first + e, first.second - e}; seg.insert(seg.begin(), {first.first, e}); } }
    }
    const int64_t mine = (int64_t)seg.size();
    const Layout L(w->cap, nte, nt2, w->nlr, w->fa_in);
    auto lo_of = [&](int64_t c) { return seg[(size_t)c].first; };
    auto n_of = [&](int64_t c) { return seg[(size_t)c].second; };
    const bool stage_in = !J.pin_data || J.in_case == 2;
    // the block on the device: voxel-major [n][nte] (cases 0 and 2) or echo-major [nte][n] (case 1); read in place either way
    const bool dev_echo_major = J.in_case == 1;

    // one input block (the volume, or the volume the FA step sees): host -> the slot's region at `off`, in the layout the kernels read
    auto put_block = [&](Slot &S, size_t off, const double *src, bool stage, int64_t lo, int64_t n) -> int {
        double *d_in = (double *)(S.dev + off);
        if (stage) {
            double *h = (double *)(S.pin + off);
            if (J.in_case == 0) {
                if (J.vs == nte) host_copy(h, src + lo * J.vs, sizeof(double) * (size_t)n * nte);
                else for (int64_t v = 0; v < n; ++v) memcpy(h + v * nte, src + (lo + v) * J.vs, sizeof(double) * nte);
            } else if (J.in_case == 1) {
                for (int e = 0; e < nte; ++e) memcpy(h + (size_t)e * n, src + e * J.es + lo, sizeof(double) * (size_t)n);
            } else {
                for (int64_t v = 0; v < n; ++v)
                    for (int e = 0; e < nte; ++e) h[v * nte + e] = src[(lo + v) * J.vs + e * J.es];
            }
            HIPCHK(hipMemcpyAsync(d_in, h, sizeof(double) * (size_t)n * nte, hipMemcpyHostToDevice, w->s_in));
        } else if (J.in_case == 0) {
            if (J.vs == nte) HIPCHK(hipMemcpyAsync(d_in, src + lo * J.vs, sizeof(double) * (size_t)n * nte, hipMemcpyHostToDevice, w->s_in));
            else HIPCHK(hipMemcpy2DAsync(d_in, sizeof(double) * nte, src + lo * J.vs, sizeof(double) * J.vs, sizeof(double) * nte, (size_t)n,
                                         hipMemcpyHostToDevice, w->s_in));
        } else {
            HIPCHK(hipMemcpy2DAsync(d_in, sizeof(double) * (size_t)n, src + lo, sizeof(double) * J.es, sizeof(double) * (size_t)n, (size_t)nte,
                                    hipMemcpyHostToDevice, w->s_in));
        }
        return MET2_OK;
    };
    const bool stage_fa_in = J.fa_data && (!J.pin_fa_data || J.in_case == 2);

    auto upload = [&](int64_t c) -> int {
        Slot &S = w->slot[c & 1];
        const int64_t lo = lo_of(c), n = n_of(c);
        if (stage_in || stage_fa_in || (J.fa_index && !J.pin_fa) || (J.mask && !J.pin_mask))
            if (c >= 2) HIPCHK(hipEventSynchronize(S.ev_in));        // the H2D of block c - 2 has left this pinned slot
        if (c >= 2) {                                                 // the device slot is free once block c - 2 has been fitted and its
            HIPCHK(hipStreamWaitEvent(w->s_in, S.ev_fit, 0));         // outputs (the FA indices live in it) copied out: waited for on the
            HIPCHK(hipStreamWaitEvent(w->s_in, S.ev_out, 0));         // GPU, not by this thread
        }
        int rc_ = put_block(S, L.in, J.data, stage_in, lo, n);
        if (rc_) return rc_;
        if (J.fa_data && (rc_ = put_block(S, L.in_fa, J.fa_data, stage_fa_in, lo, n))) return rc_;
        if (J.fa_index) {
            const double *src = J.fa_index + lo;
            if (!J.pin_fa) { memcpy(S.pin + L.fa, src, sizeof(double) * (size_t)n); src = (const double *)(S.pin + L.fa); }
            HIPCHK(hipMemcpyAsync(S.dev + L.fa, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, w->s_in));
        }
        if (J.mask) {
            const uint8_t *src = J.mask + lo;
            if (!J.pin_mask) { memcpy(S.pin + L.mk, src, (size_t)n); src = (const uint8_t *)(S.pin + L.mk); }
            HIPCHK(hipMemcpyAsync(S.dev + L.mk, src, (size_t)n, hipMemcpyHostToDevice, w->s_in));
        }
        HIPCHK(hipEventRecord(S.ev_in, w->s_in));
        return MET2_OK;
    };

    // D2H of one output array of block c: straight into the caller's (pinned) array, or into the pinned slot for drain()
    auto d2h = [&](Slot &S, size_t off, void *user, bool direct, size_t bytes) -> int {
        if (!user) return MET2_OK;
        HIPCHK(hipMemcpyAsync(direct ? user : (void *)(S.pin + off), S.dev + off, bytes, hipMemcpyDeviceToHost, w->s_out));
        return MET2_OK;
    };
    auto drain = [&](int64_t c) -> int {
        Slot &S = w->slot[c & 1];
        const int64_t lo = lo_of(c), n = n_of(c);
        HIPCHK(hipEventSynchronize(S.ev_out));
        if (!J.pin_fsol) host_copy(J.fsol + lo * nt2, S.pin + L.fsol, sizeof(double) * (size_t)n * nt2);
        if (J.sig && !J.pin_sig) host_copy(J.sig + lo * nte, S.pin + L.sig, sizeof(double) * (size_t)n * nte);
        if (!J.pin_reg) memcpy(J.reg + lo, S.pin + L.reg, sizeof(double) * (size_t)n);
        if (J.lam && !J.pin_lam) memcpy(J.lam + lo, S.pin + L.lam, sizeof(double) * (size_t)n);
        if (J.maps && !J.pin_maps)
            for (int i = 0; i < 6; ++i) memcpy(J.maps + (size_t)i * J.nvox + lo, S.pin + L.maps + sizeof(double) * (size_t)i * n, sizeof(double) * (size_t)n);
        if (J.status && !J.pin_status) memcpy(J.status + lo, S.pin + L.status, sizeof(int32_t) * (size_t)n);
        if (J.fa_out && !J.pin_fa_out && J.estimate_fa) memcpy(J.fa_out + lo, S.pin + L.fa, sizeof(double) * (size_t)n);
        return MET2_OK;
    };

    int rc;
    if (mine > 0 && (rc = upload(0))) return rc;
    for (int64_t c = 0; c < mine; ++c) {
        Slot &S = w->slot[c & 1];
        const int64_t lo = lo_of(c), n = n_of(c);
        HIPCHK(hipStreamWaitEvent(w->s_fit, S.ev_in, 0));
        if (c >= 2) HIPCHK(hipStreamWaitEvent(w->s_fit, S.ev_out, 0));            // the outputs of block c - 2 have left this slot
        const double *d_in = (const double *)(S.dev + L.in);
        const int64_t dvs = dev_echo_major ? 1 : nte, des = dev_echo_major ? n : 1;
        const uint8_t *d_mk = J.mask ? (const uint8_t *)(S.dev + L.mk) : nullptr;
        double *d_fa = (J.fa_index || J.estimate_fa) ? (double *)(S.dev + L.fa) : nullptr;
        const double *d_fa_in = J.fa_data ? (const double *)(S.dev + L.in_fa) : d_in;      // what the FA step sees (motor:337-343)
        if (J.estimate_fa == 1) {
            rc = met2_fa_bruteforce_strided(plan, n, d_fa_in, dvs, des, d_mk, d_fa, nullptr, nullptr, w->s_fit);
            if (rc) return rc;
        } else if (J.estimate_fa == 2) {
            // fa_estimation.py:35-70: plain-NNLS residuals on the coarse grid, cubic spline through them, its bounded minimum snapped to the fine grid
            double *d_res = (double *)(S.dev + L.resid);
            rc = met2_fa_bruteforce_strided(w->plan_lr, n, d_fa_in, dvs, des, d_mk, d_fa, nullptr, d_res, w->s_fit);
            if (rc) return rc;
            rc = met2_fa_spline_select_strided(w->device, n, (int32_t)w->alpha_lr.size(), w->alpha_lr.data(), d_res, (int32_t)w->alpha_hr.size(),
                                               w->alpha_hr.data(), nte, d_fa_in, dvs, des, d_mk, d_fa, nullptr, w->s_fit);
            if (rc) return rc;
        }
        rc = met2_fit_enqueue_strided(plan, J.method, n, d_in, dvs, des, d_fa, d_mk, (double *)(S.dev + L.fsol),
                                      J.sig ? (double *)(S.dev + L.sig) : nullptr, (double *)(S.dev + L.reg),
                                      J.lam ? (double *)(S.dev + L.lam) : nullptr, J.maps ? (double *)(S.dev + L.maps) : nullptr,
                                      J.status ? (int32_t *)(S.dev + L.status) : nullptr, w->s_fit);
        if (rc) return rc;
        HIPCHK(hipEventRecord(S.ev_fit, w->s_fit));
        HIPCHK(hipStreamWaitEvent(w->s_out, S.ev_fit, 0));
        if ((rc = d2h(S, L.fsol, J.fsol + lo * nt2, J.pin_fsol, sizeof(double) * (size_t)n * nt2))) return rc;
        if ((rc = d2h(S, L.sig, J.sig ? J.sig + lo * nte : nullptr, J.pin_sig, sizeof(double) * (size_t)n * nte))) return rc;
        if ((rc = d2h(S, L.reg, J.reg + lo, J.pin_reg, sizeof(double) * (size_t)n))) return rc;
        if ((rc = d2h(S, L.lam, J.lam ? J.lam + lo : nullptr, J.pin_lam, sizeof(double) * (size_t)n))) return rc;
        if (J.maps) {
            if (J.pin_maps) {
                for (int i = 0; i < 6; ++i)
                    HIPCHK(hipMemcpyAsync(J.maps + (size_t)i * J.nvox + lo, S.dev + L.maps + sizeof(double) * (size_t)i * n, sizeof(double) * (size_t)n,
                                          hipMemcpyDeviceToHost, w->s_out));
            } else if ((rc = d2h(S, L.maps, J.maps, false, sizeof(double) * (size_t)n * 6))) return rc;
        }
        if ((rc = d2h(S, L.status, J.status ? J.status + lo : nullptr, J.pin_status, sizeof(int32_t) * (size_t)n))) return rc;
        // (the estimated indices only: given ones are copied host to host below -- the pinned slot's FA array is the staging area of the
        //  NEXT block's given indices by the time this block is drained)
        if (J.fa_out && J.estimate_fa && (rc = d2h(S, L.fa, J.fa_out + lo, J.pin_fa_out, sizeof(double) * (size_t)n))) return rc;
        HIPCHK(hipEventRecord(S.ev_out, w->s_out));
        if (c + 1 < mine && (rc = upload(c + 1))) return rc;                       // staged and enqueued while the device works on block c
        if (c >= 1 && (rc = drain(c - 1))) return rc;
    }
    if (mine > 0 && (rc = drain(mine - 1))) return rc;
    if (J.fa_out && !J.estimate_fa)                                                // the given indices, or flip angle 0 for every voxel
        for (int64_t c = 0; c < mine; ++c) {
            if (J.fa_index) { if (J.fa_out != J.fa_index) memmove(J.fa_out + lo_of(c), J.fa_index + lo_of(c), sizeof(double) * (size_t)n_of(c)); }
            else memset(J.fa_out + lo_of(c), 0, sizeof(double) * (size_t)n_of(c));
        }
    rc = met2_plan_finish(plan, w->s_fit);                                         // waits for the fits; reports an FA index outside the dictionary
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(w->s_out));
    return MET2_OK;
}

struct Outcome { int rc = MET2_OK; std::string msg; double ms = 0.0; };

void run_plan(const Job &J, int t, bool need_pin, bool spawned, Outcome *out)
{
    const auto t0 = std::chrono::steady_clock::now();
    met2_options opt;
    int rc = met2_plan_get_options(J.plans[t], &opt);
    Work *w = nullptr;
    if (!rc) rc = ensure_work(J.plans[t], opt.device, J, need_pin, &w);
    if (!rc) {
        DevGuard guard(opt.device);
        rc = pipeline(J, t, w);
        if (rc) {
            // nothing of this call may still be in flight when it returns: the caller's arrays are the copies' targets, and the plan must
            // not keep a pending error word
            const std::string msg = met2_last_error();
            (void)hipStreamSynchronize(w->s_in); (void)hipStreamSynchronize(w->s_fit); (void)hipStreamSynchronize(w->s_out);
            (void)met2_plan_finish(J.plans[t], w->s_fit);
            (void)hipGetLastError();
            out->msg = msg;
        }
    } else out->msg = met2_last_error();
    if (spawned && J.estimate_fa == 2) met2::spline_tables_release();     // the spline step's per-thread device tables die with this thread
    out->rc = rc;
    out->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

extern "C" int met2_plan_attach_fa_spline(met2_plan *plan, met2_plan *plan_lr, int32_t n_lr, const double *alpha_lr, int32_t n_hr, const double *alpha_hr)
{
    if (!plan) return fail(MET2_E_INVALID, "NULL plan");
    Work *w;
    {
        std::lock_guard<std::mutex> lock(g_work_mutex);
        Work *&slot = g_work[plan];
        if (!slot) {
            met2_options opt;
            const int rc = met2_plan_get_options(plan, &opt);
            if (rc) { g_work.erase(plan); return rc; }
            slot = new Work(); slot->device = opt.device;
        }
        w = slot;
    }
    if (!plan_lr) { w->plan_lr = nullptr; w->alpha_lr.clear(); w->alpha_hr.clear(); return MET2_OK; }     // detach
    if (plan_lr == plan) return fail(MET2_E_INVALID, "the coarse plan must be another plan");
    if (!alpha_lr || !alpha_hr) return fail(MET2_E_INVALID, "NULL argument");
    int a = 0, b = 0, c = 0, a2 = 0, b2 = 0, c2 = 0;
    met2_options o1, o2;
    int rc = met2_plan_get_shape(plan, &a, &b, &c);
    if (!rc) rc = met2_plan_get_shape(plan_lr, &a2, &b2, &c2);
    if (!rc) rc = met2_plan_get_options(plan, &o1);
    if (!rc) rc = met2_plan_get_options(plan_lr, &o2);
    if (rc) return rc;
    if (a2 != a || b2 != b || o1.device != o2.device) return fail(MET2_E_INVALID, "the coarse plan must have the plan's n_te x n_t2 and live on its device");
    if (n_lr != c2 || n_hr != c) return fail(MET2_E_INVALID, "n_lr / n_hr must be the flip-angle counts of the coarse plan and of the plan");
    if (n_lr < 4 || n_lr > 32) return fail(MET2_E_UNSUPPORTED, "coarse FA grid must have 4..32 points");
    for (int i = 1; i < n_lr; ++i) if (!(alpha_lr[i] > alpha_lr[i - 1])) return fail(MET2_E_INVALID, "coarse FA grid must increase");
    w->plan_lr = plan_lr;
    w->alpha_lr.assign(alpha_lr, alpha_lr + n_lr);
    w->alpha_hr.assign(alpha_hr, alpha_hr + n_hr);
    return MET2_OK;
}

namespace met2 {
// called by met2_plan_destroy: the block buffers, streams and events this entry keeps with a plan
__attribute__((visibility("hidden"))) void host_release(met2_plan *plan)
{
    Work *w = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_work_mutex);
        auto it = g_work.find(plan);
        if (it != g_work.end()) { w = it->second; g_work.erase(it); }
    }
    free_work(w);
}
}  // namespace met2

extern "C" int met2_fit_host(met2_plan *const *plans, int32_t n_plans, int32_t method, int64_t nvox, const double *data, const double *fa_data,
                             int64_t voxel_stride, int64_t echo_stride, const double *fa_index, const uint8_t *mask, int32_t estimate_fa, double *fsol, double *sig,
                             double *reg, double *lam, double *maps, int32_t *status, double *fa_out, int64_t chunk, double *plan_ms)
{
    if (!plans || n_plans < 1 || n_plans > 64) return fail(MET2_E_INVALID, "met2_fit_host: 1 to 64 plans");
    int nte = 0, nt2 = 0, nfa = 0;
    for (int t = 0; t < n_plans; ++t) {
        if (!plans[t]) return fail(MET2_E_INVALID, "met2_fit_host: NULL plan");
        for (int u = 0; u < t; ++u) if (plans[u] == plans[t]) return fail(MET2_E_INVALID, "met2_fit_host: the same plan twice (one plan serves one stream at a time)");
        int a, b, c;
        const int rc = met2_plan_get_shape(plans[t], &a, &b, &c);
        if (rc) return rc;
        if (t == 0) { nte = a; nt2 = b; nfa = c; }
        else if (a != nte || b != nt2 || c != nfa) return fail(MET2_E_INVALID, "met2_fit_host: plans of different shapes");
    }
    if (plan_ms) for (int t = 0; t < n_plans; ++t) plan_ms[t] = 0.0;
    if (nvox == 0) return MET2_OK;
    if (nvox < 0) return fail(MET2_E_INVALID, "nvox out of range");
    if (!data || !fsol || !reg) return fail(MET2_E_INVALID, "NULL argument");
    if (voxel_stride <= 0 || echo_stride <= 0) return fail(MET2_E_INVALID, "met2_fit_host: strides must be positive");
    if (estimate_fa < 0 || estimate_fa > 2) return fail(MET2_E_INVALID, "estimate_fa: 0 (given / flip angle 0), 1 (brute force) or 2 (spline)");
    if (estimate_fa && fa_index) return fail(MET2_E_INVALID, "estimate_fa together with fa_index");
    if (fa_data && !estimate_fa) return fail(MET2_E_INVALID, "fa_data is what the FA estimation sees: it needs estimate_fa 1 or 2");
    if (estimate_fa == 2) {
        std::lock_guard<std::mutex> lock(g_work_mutex);
        for (int t = 0; t < n_plans; ++t) {
            auto it = g_work.find(plans[t]);
            if (it == g_work.end() || !it->second->plan_lr)
                return fail(MET2_E_STATE, "estimate_fa = 2 needs met2_plan_attach_fa_spline on every plan first");
        }
    }
    if (chunk < 0) return fail(MET2_E_INVALID, "chunk < 0");
    if (chunk == 0) {
        // four blocks per plan so that the copies of a plan's first and last block (the only ones not under a fit) are a quarter of its
        // share, in multiples of 4 096 voxels, at most 262 144 (0.29 GB per slot at 32 x 60)
        const int64_t per = (nvox + (int64_t)n_plans * 4 - 1) / ((int64_t)n_plans * 4);
        chunk = std::min<int64_t>(262144, std::max<int64_t>(4096, (per + 4095) / 4096 * 4096));
    }
    chunk = std::min<int64_t>(chunk, nvox);
    if (chunk > 0x7fffffff) return fail(MET2_E_INVALID, "chunk out of range");

    Job J;
    J.plans = plans; J.n_plans = n_plans; J.method = method; J.nvox = nvox; J.data = data; J.fa_data = fa_data; J.vs = voxel_stride; J.es = echo_stride;
    J.fa_index = fa_index; J.mask = mask; J.estimate_fa = estimate_fa; J.fsol = fsol; J.sig = sig; J.reg = reg; J.lam = lam; J.maps = maps;
    J.status = status; J.fa_out = fa_out; J.chunk = chunk; J.nte = nte; J.nt2 = nt2;
    J.split = getenv("MET2_HOST_NOSPLIT") == nullptr;            // test / A-B switch: whole blocks only
    J.in_case = echo_stride == 1 ? 0 : (voxel_stride == 1 ? 1 : 2);
    if (J.in_case == 0 && voxel_stride < nte) return fail(MET2_E_INVALID, "met2_fit_host: voxel_stride < n_te with echo_stride 1 (overlapping voxels)");
    J.pin_data = is_pinned(data); J.pin_fa_data = is_pinned(fa_data); J.pin_fa = is_pinned(fa_index); J.pin_mask = is_pinned(mask); J.pin_fsol = is_pinned(fsol);
    J.pin_sig = is_pinned(sig); J.pin_reg = is_pinned(reg); J.pin_lam = is_pinned(lam); J.pin_maps = is_pinned(maps);
    J.pin_status = is_pinned(status); J.pin_fa_out = is_pinned(fa_out);
    const bool need_pin = !(J.pin_data && J.pin_fa_data && J.in_case != 2 && J.pin_fa && J.pin_mask && J.pin_fsol && J.pin_sig && J.pin_reg && J.pin_lam && J.pin_maps &&
                            J.pin_status && J.pin_fa_out);

    std::vector<Outcome> res(n_plans);
    if (n_plans == 1) run_plan(J, 0, need_pin, false, &res[0]);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_plans; ++t) th.emplace_back(run_plan, std::cref(J), t, need_pin, true, &res[t]);
        for (auto &x : th) x.join();
    }
    if (plan_ms) for (int t = 0; t < n_plans; ++t) plan_ms[t] = res[t].ms;
    for (int t = 0; t < n_plans; ++t)
        if (res[t].rc) return fail(res[t].rc, "met2_fit_host, plan " + std::to_string(t) + ": " + res[t].msg);
    return MET2_OK;
}
