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

namespace met2 {
__attribute__((visibility("hidden"))) void spline_tables_release();
__attribute__((visibility("hidden"))) int plan_reserve(met2_plan *plan, int64_t nvox);      // the plan's per-voxel scratch for blocks of nvox voxels (met2_hip.hip)
}

namespace {

// motor:180-182 and :279 on a block: every echo of voxel v times mask_values[v], negative values clipped to 0 (in place).
// The block is [n][nte] (vs = nte, es = 1) or [nte][n] (vs = 1, es = n): element i belongs to voxel i / nte or i % n.
__global__ void host_prepare_kernel(double *d, const double *mv, int64_t n, int nte, int echo_major)
{
    const int64_t total = n * nte;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = echo_major ? i % n : i / nte;
        const double x = d[i] * mv[v];
        d[i] = x > 0.0 ? x : (x != x ? x : 0.0);                       // np.maximum-style clip that keeps nan (the fit flags it)
    }
}

// the gate of the FA step (fa_estimation.py:45): mask and a positive echo sum, as 1.0 / 0.0 per voxel
__global__ void host_gate_kernel(const double *d, const uint8_t *mk, int64_t n, int nte, int64_t vs, int64_t es, double *gate)
{
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        double t = 0.0;
        for (int e = 0; e < nte; ++e) t += d[v * vs + e * es];
        gate[v] = (t > 0.0 && (!mk || mk[v])) ? 1.0 : 0.0;
    }
}

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
    size_t pin_bytes = 0;     // size of each slot's pinned slab
    hipStream_t s_in = nullptr, s_fit = nullptr, s_out = nullptr;
    Slot slot[2];
};

// the spline FA method's configuration of a plan (met2_plan_attach_fa_spline)
struct Attach {
    met2_plan *plan_lr = nullptr;
    std::vector<double> alpha_lr, alpha_hr;
};

std::mutex g_work_mutex;
std::map<met2_plan *, Work *> g_work;
std::map<met2_plan *, Attach> g_attach;
// Buffers of destroyed plans wait here for the next plan on their device: a driver that builds its plans per call (recon_met2_arrays does)
// would otherwise allocate 0.6 GB of device memory and pin as much host memory per call -- tens of ms.  At most two per device;
// met2_host_trim() frees them.
std::vector<Work *> g_pool;

// which arrays a slab holds
struct Regions {
    bool in = true, fa = true, fsol = true, sig = true, reg = true, lam = true, maps = true, status = true, mk = true, mv = true, gate = true, in_fa = false;
    int nlr = 0;
};

// slab layout of a slot with capacity `cap` voxels (byte offsets, every array 16-byte aligned).  The device slab holds every array; the
// pinned slab only those the caller keeps in pageable memory (a driver with pinned outputs stages 256 of 1 100 bytes per voxel).
struct Layout {
    size_t in, fa, fsol, sig, reg, lam, maps, status, mk, mv, gate, in_fa, resid, total;
    Layout(int64_t cap, int nte, int nt2, const Regions &r)
    {
        size_t o = 0;
        auto take = [&](bool have, size_t bytes) { const size_t at = o; if (have) o += (bytes + 15) & ~(size_t)15; return at; };
        in = take(r.in, sizeof(double) * (size_t)cap * nte);
        fa = take(r.fa, sizeof(double) * (size_t)cap);
        fsol = take(r.fsol, sizeof(double) * (size_t)cap * nt2);
        sig = take(r.sig, sizeof(double) * (size_t)cap * nte);
        reg = take(r.reg, sizeof(double) * (size_t)cap);
        lam = take(r.lam, sizeof(double) * (size_t)cap);
        maps = take(r.maps, sizeof(double) * (size_t)cap * 6);
        status = take(r.status, sizeof(int32_t) * (size_t)cap);
        mk = take(r.mk, (size_t)cap);
        mv = take(r.mv, sizeof(double) * (size_t)cap);
        gate = take(r.gate, sizeof(double) * (size_t)cap);
        in_fa = take(r.in_fa, sizeof(double) * (size_t)cap * nte);
        resid = take(r.nlr > 0, sizeof(double) * (size_t)cap * (size_t)r.nlr);
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

// 0: ordinary (pageable) host memory, 1: pinned host memory (the device can DMA from / to it directly), 2: device memory
int mem_kind(const void *ptr)
{
    if (!ptr) return 1;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    const hipError_t e = hipPointerGetAttributes(&at, ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }          // pageable memory is unknown to the runtime
    if (at.type == hipMemoryTypeHost) return 1;
    if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeArray || at.type == hipMemoryTypeManaged) return 2;
    return 0;
}
bool is_pinned(const void *ptr) { return mem_kind(ptr) == 1; }

// dst <- src on the calling thread plus up to three helpers (a block of 262 144 voxels is 67 MB in and 208 MB out; one thread's
// memcpy would stand next to a 37 ms fit)
void host_copy(void *dst, const void *src, size_t bytes)
{
    if (bytes < ((size_t)8 << 20)) { memcpy(dst, src, bytes); return; }
    const int parts = 4;
    const size_t step = ((bytes / parts) + 63) & ~(size_t)63;
    std::thread helpers[parts - 1];
    int started = 0;
    for (int i = 1; i < parts; ++i) {
        const size_t lo = std::min(bytes, step * i), hi = std::min(bytes, step * (i + 1));
        try { helpers[i - 1] = std::thread([=] { if (hi > lo) memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); }); ++started; }
        catch (...) { break; }                  // no thread to be had (a pids limit, say): the calling thread copies the rest itself
    }
    memcpy(dst, src, std::min(bytes, step));
    for (int i = started + 1; i < parts; ++i) {
        const size_t lo = std::min(bytes, step * i), hi = std::min(bytes, step * (i + 1));
        if (hi > lo) memcpy((char *)dst + lo, (const char *)src + lo, hi - lo);
    }
    for (int i = 0; i < started; ++i) helpers[i].join();
}

// `rows` rows of row_bytes each, pitches in bytes: the rows are dealt to the calling thread and up to three helpers when there is enough to copy
void host_copy_rows(char *dst, size_t dst_pitch, const char *src, size_t src_pitch, size_t row_bytes, int64_t rows)
{
    auto run = [=](int64_t r0, int64_t r1) { for (int64_t r = r0; r < r1; ++r) memcpy(dst + r * dst_pitch, src + r * src_pitch, row_bytes); };
    if (row_bytes * (size_t)rows < ((size_t)8 << 20) || rows < 4) { run(0, rows); return; }
    const int parts = 4;
    std::thread helpers[parts - 1];
    int started = 0;
    for (int i = 1; i < parts; ++i) {
        try { helpers[i - 1] = std::thread(run, rows * i / parts, rows * (i + 1) / parts); ++started; }
        catch (...) { break; }                  // (as in host_copy)
    }
    run(0, rows / parts);
    for (int i = started + 1; i < parts; ++i) run(rows * i / parts, rows * (i + 1) / parts);
    for (int i = 0; i < started; ++i) helpers[i].join();
}

struct Job {
    met2_plan *const *plans; int n_plans; int method; int64_t nvox;
    const double *data; const double *fa_data; int64_t vs, es;
    const double *fa_index; const uint8_t *mask; const double *mask_values; int estimate_fa;
    double *fsol, *sig, *reg, *lam, *maps; int32_t *status; double *fa_out, *fa_gate;
    int64_t chunk; int64_t run; int nte, nt2; bool split;     // run: voxels per dealt run (0: one plan, contiguous blocks)
    // which host arrays the device reaches directly
    bool pin_data, pin_fa_data, pin_fa, pin_mask, pin_mv, pin_gate, pin_fsol, pin_sig, pin_reg, pin_lam, pin_maps, pin_status, pin_fa_out;
    Regions stage;            // the arrays that go through the pinned slab
    int in_case;              // 0: voxel-major rows (es == 1), 1: echo-major (vs == 1), 2: general strides (gathered on the host)
};

int ensure_work(met2_plan *plan, int device, const Job &J, bool need_pin, const Attach &at, Work **out)
{
    Work *w;
    {
        std::lock_guard<std::mutex> lock(g_work_mutex);
        Work *&slot = g_work[plan];
        if (!slot) {
            int best = -1;
            for (size_t i = 0; i < g_pool.size(); ++i)
                if (g_pool[i]->device == device && (best < 0 || g_pool[i]->cap > g_pool[(size_t)best]->cap)) best = (int)i;
            if (best >= 0) { slot = g_pool[(size_t)best]; g_pool.erase(g_pool.begin() + best); }
            else { slot = new Work(); slot->device = device; }
        }
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
    const int need_lr = J.estimate_fa == 2 ? (int)at.alpha_lr.size() : 0;
    const bool need_fa_in = J.fa_data != nullptr;
    const bool grow = w->cap < J.chunk || w->nte != J.nte || w->nt2 != J.nt2 || w->nlr < need_lr || (need_fa_in && !w->fa_in);
    // a failed allocation leaves the Work EMPTY (cap = 0, no slabs), never half-sized: the shape is committed only once every slab exists, so the
    // next call -- on this plan, or on the plan that takes this Work from the pool -- allocates again instead of running on a NULL slab
    auto drop_all = [&]() {
        for (Slot &s : w->slot) {
            if (s.dev) { (void)hipFree(s.dev); s.dev = nullptr; }
            if (s.pin) { (void)hipHostFree(s.pin); s.pin = nullptr; }
        }
        w->cap = 0; w->pin_bytes = 0;
    };
    if (grow) {
        const int64_t cap = std::max(w->cap, J.chunk);
        const int nlr = std::max(w->nlr, need_lr);
        const bool fa_in = w->fa_in || need_fa_in;
        drop_all();
        Regions all; all.in_fa = fa_in; all.nlr = nlr;
        const Layout L(cap, J.nte, J.nt2, all);
        for (Slot &s : w->slot) {
            const hipError_t e = hipMalloc((void **)&s.dev, L.total);
            if (e != hipSuccess) { (void)hipGetLastError(); s.dev = nullptr; drop_all(); return fail(MET2_E_HIP, std::string("hipMalloc of a block slot: ") + hipGetErrorString(e)); }
        }
        w->cap = cap; w->nte = J.nte; w->nt2 = J.nt2; w->nlr = nlr; w->fa_in = fa_in;
    }
    const size_t pin_need = need_pin ? Layout(w->cap, w->nte, w->nt2, J.stage).total : 0;
    if (pin_need > w->pin_bytes) {
        for (Slot &s : w->slot) if (s.pin) { (void)hipHostFree(s.pin); s.pin = nullptr; }
        w->pin_bytes = 0;
        for (Slot &s : w->slot) {
            const hipError_t e = hipHostMalloc((void **)&s.pin, pin_need, hipHostMallocDefault);
            if (e != hipSuccess) { (void)hipGetLastError(); s.pin = nullptr; drop_all(); return fail(MET2_E_HIP, std::string("hipHostMalloc of a staging slab: ") + hipGetErrorString(e)); }
        }
        w->pin_bytes = pin_need;
    }
    *out = w;
    return MET2_OK;
}

// One plan's share of the job.
// Dealing (J.run): with several plans the voxel list is dealt in RUNS of 4 096 voxels, run j -> plan j mod n_plans (the granularity of dist.py's
// interleave: background, CSF and white matter cluster along the slow axis and differ 10x in iteration count), and a plan's DMA block is `chunk`
// voxels of ITS runs -- chunk / 4 096 runs that lie n_plans runs apart in the caller's arrays, moved by one pitched (2-D) copy per array and
// direction.  With one plan (J.run == 0) a block is a contiguous range, as before.  Until round 5 whole blocks of 65 536 - 262 144 voxels were
// dealt round-robin: 16 - 64 times coarser than the distributed driver.
int pipeline(const Job &J, int t, Work *w, const Attach &at)
{
    met2_plan *plan = J.plans[t];
    const int nte = J.nte, nt2 = J.nt2;
    const int64_t chunk = J.chunk, run = J.run;
    const int NP = J.n_plans;
    // a segment: n voxels of this plan starting at voxel g0 of the caller's list -- contiguous (run == 0), or runs at g0, g0 + NP run, ...
    struct Seg { int64_t g0, n; };
    std::vector<Seg> seg;
    if (run == 0) {
        const int64_t nblocks = (J.nvox + chunk - 1) / chunk;
        for (int64_t b = t; b < nblocks; b += NP) seg.push_back({b * chunk, std::min<int64_t>(chunk, J.nvox - b * chunk)});
    } else {
        const int64_t nruns = (J.nvox + run - 1) / run, cr = chunk / run;            // (chunk is a multiple of run)
        const int64_t mine_runs = t < nruns ? (nruns - t + NP - 1) / NP : 0;
        for (int64_t r0 = 0; r0 < mine_runs; r0 += cr) {
            const int64_t r1 = std::min(mine_runs, r0 + cr), jl = t + (r1 - 1) * NP;  // jl: the block's last run in the caller's list
            seg.push_back({(t + r0 * NP) * run, (r1 - 1 - r0) * run + std::min<int64_t>(run, J.nvox - jl * run)});
        }
    }
    auto advance = [&](int64_t g0, int64_t e) { return run ? g0 + (e / run) * NP * run : g0 + e; };   // the voxel e places further on in this plan's share (e: whole runs)
    // the first block is split 1/8 + 7/8 and the last one 7/8 + 1/8 -- the upload of the very first piece and the download of the very last one
    // are the only copies that do not run under a fit, so they are made short (a piece costs ~1.5 ms of queue tail and spill-over kernel)
    if (J.split && !seg.empty()) {
        auto eighth = [](int64_t n) { return n >= 131072 ? std::max<int64_t>(4096, (n / 8) & ~(int64_t)4095) : 0; };      // (smaller blocks are not worth two more pieces)
        {   const Seg last = seg.back(); const int64_t e = eighth(last.n), keep = run ? ((last.n - e + run - 1) / run) * run : last.n - e;
            if (e && keep > 0 && keep < last.n) { seg.back() = {last.g0, keep}; seg.push_back({advance(last.g0, keep), last.n - keep}); } }
        {   const Seg first = seg.front(); const int64_t e = eighth(first.n);
            if (e && e < first.n) { seg.front() = {advance(first.g0, e), first.n - e}; seg.insert(seg.begin(), {first.g0, e}); } }
    }
    const int64_t mine = (int64_t)seg.size();
    Regions all_; all_.in_fa = w->fa_in; all_.nlr = w->nlr;
    const Layout L(w->cap, nte, nt2, all_);          // device slab
    const Layout P(w->cap, nte, nt2, J.stage);       // pinned slab
    const bool stage_in = !J.pin_data || J.in_case == 2;
    // the block on the device: voxel-major [n][nte] (cases 0 and 2) or echo-major [nte][n] (case 1); read in place either way
    const bool dev_echo_major = J.in_case == 1;

    // ---- a per-voxel array of the caller (rb bytes per voxel, voxel 0 at `host`) against the block's packed copy
    // the contiguous pieces of a segment: fn(first voxel in the caller's list, offset inside the block, voxels)
    auto pieces = [&](const Seg &sg, auto &&fn) {
        if (run == 0) { fn(sg.g0, (int64_t)0, sg.n); return; }
        for (int64_t off = 0, g = sg.g0; off < sg.n; off += run, g += (int64_t)NP * run) fn(g, off, std::min<int64_t>(run, sg.n - off));
    };
    // block <-> host array on the calling thread (and helpers when the block is large): staging of pageable arrays
    auto host_side = [&](char *packed, char *host, size_t rb, const Seg &sg, bool to_packed) {
        if (run == 0 || sg.n <= run) {
            if (to_packed) host_copy(packed, host + (size_t)sg.g0 * rb, (size_t)sg.n * rb); else host_copy(host + (size_t)sg.g0 * rb, packed, (size_t)sg.n * rb);
            return;
        }
        const int64_t nf = sg.n / run, tail = sg.n % run;
        if (to_packed) host_copy_rows(packed, (size_t)run * rb, host + (size_t)sg.g0 * rb, (size_t)NP * run * rb, (size_t)run * rb, nf);
        else host_copy_rows(host + (size_t)sg.g0 * rb, (size_t)NP * run * rb, packed, (size_t)run * rb, (size_t)run * rb, nf);
        if (tail) {
            char *hp = host + (size_t)(sg.g0 + nf * NP * run) * rb, *pp = packed + (size_t)nf * run * rb;
            if (to_packed) memcpy(pp, hp, (size_t)tail * rb); else memcpy(hp, pp, (size_t)tail * rb);
        }
    };
    // block <-> host array by the device's copy engines (pinned or device-resident host side): one pitched copy for the whole runs, one for a tail
    auto dma = [&](char *dev, char *host, size_t rb, const Seg &sg, bool to_dev, hipStream_t st) -> int {
        const hipMemcpyKind kind = to_dev ? hipMemcpyDefault : hipMemcpyDeviceToHost;      // (hipMemcpyDefault: a volume that a filter left on a device is a device pointer)
        if (run == 0 || sg.n <= run) {
            char *hp = host + (size_t)sg.g0 * rb;
            HIPCHK(hipMemcpyAsync(to_dev ? (void *)dev : (void *)hp, to_dev ? (const void *)hp : (const void *)dev, (size_t)sg.n * rb, kind, st));
            return MET2_OK;
        }
        const int64_t nf = sg.n / run, tail = sg.n % run;
        char *hp = host + (size_t)sg.g0 * rb;
        const size_t wd = (size_t)run * rb, hpitch = (size_t)NP * run * rb;
        if (to_dev) HIPCHK(hipMemcpy2DAsync(dev, wd, hp, hpitch, wd, (size_t)nf, kind, st));
        else HIPCHK(hipMemcpy2DAsync(hp, hpitch, dev, wd, wd, (size_t)nf, kind, st));
        if (tail) {
            char *ht = host + (size_t)(sg.g0 + nf * NP * run) * rb, *dt = dev + (size_t)nf * run * rb;
            HIPCHK(hipMemcpyAsync(to_dev ? (void *)dt : (void *)ht, to_dev ? (const void *)ht : (const void *)dt, (size_t)tail * rb, kind, st));
        }
        return MET2_OK;
    };
    // host array -> device block: through the pinned slab's room `stage` when the array is pageable (stage != NULL), else straight
    auto to_device = [&](char *dev, const void *host, size_t rb, const Seg &sg, char *stage) -> int {
        if (!stage) return dma(dev, (char *)host, rb, sg, true, w->s_in);
        host_side(stage, (char *)host, rb, sg, true);
        HIPCHK(hipMemcpyAsync(dev, stage, (size_t)sg.n * rb, hipMemcpyHostToDevice, w->s_in));
        return MET2_OK;
    };
    // device block -> host array, or -> the pinned slab's room for it (drain() copies on)
    auto from_device = [&](char *dev, void *host, size_t rb, const Seg &sg, char *stage) -> int {
        if (!host) return MET2_OK;
        if (!stage) return dma(dev, (char *)host, rb, sg, false, w->s_out);
        HIPCHK(hipMemcpyAsync(stage, dev, (size_t)sg.n * rb, hipMemcpyDeviceToHost, w->s_out));
        return MET2_OK;
    };

    // one input block (the volume, or the volume the FA step sees): host -> the slot's region at `off`, in the layout the kernels read
    auto put_block = [&](Slot &S, size_t off, size_t poff, const double *src, bool stage, const Seg &sg) -> int {
        char *d_in = S.dev + off, *h = stage ? S.pin + poff : nullptr;
        const int64_t n = sg.n;
        int rc_ = MET2_OK;
        if (J.in_case == 0 && J.vs == nte) return to_device(d_in, src, sizeof(double) * (size_t)nte, sg, h);
        if (J.in_case == 1) {                          // echo-major: every echo is a per-voxel array of its own
            for (int e = 0; e < nte && !rc_; ++e)
                rc_ = to_device(d_in + sizeof(double) * (size_t)e * n, src + (size_t)e * J.es, sizeof(double), sg, h ? h + sizeof(double) * (size_t)e * n : nullptr);
            return rc_;
        }
        // rows with a pitch, or general strides: contiguous blocks only (fit_host_impl deals whole blocks for these layouts)
        const int64_t lo = sg.g0;
        if (stage) {
            double *hd = (double *)h;
            if (J.in_case == 0) host_copy_rows((char *)hd, sizeof(double) * nte, (const char *)(src + lo * J.vs), sizeof(double) * (size_t)J.vs, sizeof(double) * nte, n);
            else
                for (int64_t v = 0; v < n; ++v)
                    for (int e = 0; e < nte; ++e) hd[v * nte + e] = src[(lo + v) * J.vs + e * J.es];
            HIPCHK(hipMemcpyAsync(d_in, hd, sizeof(double) * (size_t)n * nte, hipMemcpyHostToDevice, w->s_in));
        } else
            HIPCHK(hipMemcpy2DAsync(d_in, sizeof(double) * nte, src + lo * J.vs, sizeof(double) * J.vs, sizeof(double) * nte, (size_t)n, hipMemcpyDefault, w->s_in));
        return MET2_OK;
    };
    const bool stage_fa_in = J.fa_data && (!J.pin_fa_data || J.in_case == 2);

    auto upload = [&](int64_t c) -> int {
        Slot &S = w->slot[c & 1];
        const Seg &sg = seg[(size_t)c];
        if (stage_in || stage_fa_in || (J.fa_index && !J.pin_fa) || (J.mask && !J.pin_mask) || (J.mask_values && !J.pin_mv))
            if (c >= 2) HIPCHK(hipEventSynchronize(S.ev_in));        // the H2D of block c - 2 has left this pinned slot
        if (c >= 2) {                                                 // the device slot is free once block c - 2 has been fitted and its
            HIPCHK(hipStreamWaitEvent(w->s_in, S.ev_fit, 0));         // outputs (the FA indices live in it) copied out: waited for on the
            HIPCHK(hipStreamWaitEvent(w->s_in, S.ev_out, 0));         // GPU, not by this thread
        }
        int rc_ = put_block(S, L.in, P.in, J.data, stage_in, sg);
        if (rc_) return rc_;
        if (J.fa_data && (rc_ = put_block(S, L.in_fa, P.in_fa, J.fa_data, stage_fa_in, sg))) return rc_;
        if (J.fa_index && (rc_ = to_device(S.dev + L.fa, J.fa_index, sizeof(double), sg, J.pin_fa ? nullptr : S.pin + P.fa))) return rc_;
        if (J.mask && (rc_ = to_device(S.dev + L.mk, J.mask, 1, sg, J.pin_mask ? nullptr : S.pin + P.mk))) return rc_;
        if (J.mask_values && (rc_ = to_device(S.dev + L.mv, J.mask_values, sizeof(double), sg, J.pin_mv ? nullptr : S.pin + P.mv))) return rc_;
        HIPCHK(hipEventRecord(S.ev_in, w->s_in));
        return MET2_OK;
    };

    auto drain = [&](int64_t c) -> int {
        Slot &S = w->slot[c & 1];
        const Seg &sg = seg[(size_t)c];
        const int64_t n = sg.n;
        HIPCHK(hipEventSynchronize(S.ev_out));
        if (!J.pin_fsol) host_side(S.pin + P.fsol, (char *)J.fsol, sizeof(double) * (size_t)nt2, sg, false);
        if (J.sig && !J.pin_sig) host_side(S.pin + P.sig, (char *)J.sig, sizeof(double) * (size_t)nte, sg, false);
        if (!J.pin_reg) host_side(S.pin + P.reg, (char *)J.reg, sizeof(double), sg, false);
        if (J.lam && !J.pin_lam) host_side(S.pin + P.lam, (char *)J.lam, sizeof(double), sg, false);
        if (J.maps && !J.pin_maps)
            for (int i = 0; i < 6; ++i) host_side(S.pin + P.maps + sizeof(double) * (size_t)i * n, (char *)(J.maps + (size_t)i * J.nvox), sizeof(double), sg, false);
        if (J.status && !J.pin_status) host_side(S.pin + P.status, (char *)J.status, sizeof(int32_t), sg, false);
        if (J.fa_out && !J.pin_fa_out && J.estimate_fa) host_side(S.pin + P.fa, (char *)J.fa_out, sizeof(double), sg, false);
        if (J.fa_gate && !J.pin_gate) host_side(S.pin + P.gate, (char *)J.fa_gate, sizeof(double), sg, false);
        return MET2_OK;
    };

    int rc;
    if (mine > 0 && (rc = upload(0))) return rc;
    for (int64_t c = 0; c < mine; ++c) {
        Slot &S = w->slot[c & 1];
        const Seg &sg = seg[(size_t)c];
        const int64_t n = sg.n;
        HIPCHK(hipStreamWaitEvent(w->s_fit, S.ev_in, 0));
        if (c >= 2) HIPCHK(hipStreamWaitEvent(w->s_fit, S.ev_out, 0));            // the outputs of block c - 2 have left this slot
        const double *d_in = (const double *)(S.dev + L.in);
        const int64_t dvs = dev_echo_major ? 1 : nte, des = dev_echo_major ? n : 1;
        const uint8_t *d_mk = J.mask ? (const uint8_t *)(S.dev + L.mk) : nullptr;
        double *d_fa = (J.fa_index || J.estimate_fa) ? (double *)(S.dev + L.fa) : nullptr;
        if (J.mask_values) {
            const int64_t total = n * nte;
            const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 1 << 16);
            hipLaunchKernelGGL(host_prepare_kernel, dim3(blocks), dim3(256), 0, w->s_fit, (double *)(S.dev + L.in), (const double *)(S.dev + L.mv), n, nte,
                               dev_echo_major ? 1 : 0);
            HIPCHK(hipGetLastError());
        }
        const double *d_fa_in = J.fa_data ? (const double *)(S.dev + L.in_fa) : d_in;      // what the FA step sees (motor:337-343)
        if (J.fa_gate) {
            const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 1 << 16);
            hipLaunchKernelGGL(host_gate_kernel, dim3(blocks), dim3(256), 0, w->s_fit, d_fa_in, d_mk, n, nte, dvs, des, (double *)(S.dev + L.gate));
            HIPCHK(hipGetLastError());
        }
        if (J.estimate_fa == 1) {
            rc = met2_fa_bruteforce_strided(plan, n, d_fa_in, dvs, des, d_mk, d_fa, nullptr, nullptr, w->s_fit);
            if (rc) return rc;
        } else if (J.estimate_fa == 2) {
            // fa_estimation.py:35-70: plain-NNLS residuals on the coarse grid, cubic spline through them, its bounded minimum snapped to the fine grid
            double *d_res = (double *)(S.dev + L.resid);
            rc = met2_fa_bruteforce_strided(at.plan_lr, n, d_fa_in, dvs, des, d_mk, d_fa, nullptr, d_res, w->s_fit);
            if (rc) return rc;
            rc = met2_fa_spline_select_strided(w->device, n, (int32_t)at.alpha_lr.size(), at.alpha_lr.data(), d_res, (int32_t)at.alpha_hr.size(),
                                               at.alpha_hr.data(), nte, d_fa_in, dvs, des, d_mk, d_fa, nullptr, w->s_fit);
            if (rc) return rc;
        }
        rc = met2_fit_enqueue_strided(plan, J.method, n, d_in, dvs, des, d_fa, d_mk, (double *)(S.dev + L.fsol),
                                      J.sig ? (double *)(S.dev + L.sig) : nullptr, (double *)(S.dev + L.reg),
                                      J.lam ? (double *)(S.dev + L.lam) : nullptr, J.maps ? (double *)(S.dev + L.maps) : nullptr,
                                      J.status ? (int32_t *)(S.dev + L.status) : nullptr, w->s_fit);
        if (rc) return rc;
        HIPCHK(hipEventRecord(S.ev_fit, w->s_fit));
        HIPCHK(hipStreamWaitEvent(w->s_out, S.ev_fit, 0));
        if ((rc = from_device(S.dev + L.fsol, J.fsol, sizeof(double) * (size_t)nt2, sg, J.pin_fsol ? nullptr : S.pin + P.fsol))) return rc;
        if ((rc = from_device(S.dev + L.sig, J.sig, sizeof(double) * (size_t)nte, sg, J.pin_sig ? nullptr : S.pin + P.sig))) return rc;
        if ((rc = from_device(S.dev + L.reg, J.reg, sizeof(double), sg, J.pin_reg ? nullptr : S.pin + P.reg))) return rc;
        if ((rc = from_device(S.dev + L.lam, J.lam, sizeof(double), sg, J.pin_lam ? nullptr : S.pin + P.lam))) return rc;
        if (J.maps) {
            if (J.pin_maps) {
                for (int i = 0; i < 6; ++i)
                    if ((rc = from_device(S.dev + L.maps + sizeof(double) * (size_t)i * n, J.maps + (size_t)i * J.nvox, sizeof(double), sg, nullptr))) return rc;
            } else HIPCHK(hipMemcpyAsync(S.pin + P.maps, S.dev + L.maps, sizeof(double) * (size_t)n * 6, hipMemcpyDeviceToHost, w->s_out));
        }
        if ((rc = from_device(S.dev + L.status, J.status, sizeof(int32_t), sg, J.pin_status ? nullptr : S.pin + P.status))) return rc;
        // (the estimated indices only: given ones are copied host to host below -- the pinned slot's FA array is the staging area of the
        //  NEXT block's given indices by the time this block is drained)
        if (J.fa_out && J.estimate_fa && (rc = from_device(S.dev + L.fa, J.fa_out, sizeof(double), sg, J.pin_fa_out ? nullptr : S.pin + P.fa))) return rc;
        if ((rc = from_device(S.dev + L.gate, J.fa_gate, sizeof(double), sg, J.pin_gate ? nullptr : S.pin + P.gate))) return rc;
        HIPCHK(hipEventRecord(S.ev_out, w->s_out));
        if (c + 1 < mine && (rc = upload(c + 1))) return rc;                       // staged and enqueued while the device works on block c
        if (c >= 1 && (rc = drain(c - 1))) return rc;
    }
    if (mine > 0 && (rc = drain(mine - 1))) return rc;
    if (J.fa_out && !J.estimate_fa)                                                // the given indices, or flip angle 0 for every voxel
        for (int64_t c = 0; c < mine; ++c)
            pieces(seg[(size_t)c], [&](int64_t g, int64_t, int64_t cnt) {
                if (J.fa_index) { if (J.fa_out != J.fa_index) memmove(J.fa_out + g, J.fa_index + g, sizeof(double) * (size_t)cnt); }
                else memset(J.fa_out + g, 0, sizeof(double) * (size_t)cnt);
            });
    rc = met2_plan_finish(plan, w->s_fit);                                         // waits for the fits; reports an FA index outside the dictionary
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(w->s_out));
    return MET2_OK;
}

struct Outcome { int rc = MET2_OK; std::string msg; double ms = 0.0; };

void run_plan_body(const Job &J, int t, bool need_pin, bool spawned, Outcome *out);
// a C++ exception (std::bad_alloc, a thread that cannot be started) must not leave a plan's thread -- std::terminate would take the host process down
void run_plan(const Job &J, int t, bool need_pin, bool spawned, Outcome *out)
{
    try { run_plan_body(J, t, need_pin, spawned, out); }
    catch (const std::exception &e) { out->rc = MET2_E_HIP; out->msg = std::string("C++ exception: ") + e.what(); }
    catch (...) { out->rc = MET2_E_HIP; out->msg = "unknown C++ exception"; }
}
void run_plan_body(const Job &J, int t, bool need_pin, bool spawned, Outcome *out)
{
    const auto t0 = std::chrono::steady_clock::now();
    met2_options opt;
    int rc = met2_plan_get_options(J.plans[t], &opt);
    Work *w = nullptr;
    Attach at;
    {
        std::lock_guard<std::mutex> lock(g_work_mutex);
        auto it = g_attach.find(J.plans[t]);
        if (it != g_attach.end()) at = it->second;
    }
    if (!rc) rc = ensure_work(J.plans[t], opt.device, J, need_pin, at, &w);
    // the plan's sort scratch for the largest block, once: growing it block by block (the first piece is an eighth of a block) frees device
    // memory twice per call, and hipFree waits for every stream of the device
    if (!rc) rc = met2::plan_reserve(J.plans[t], J.chunk);
    if (!rc && J.estimate_fa == 2 && at.plan_lr) rc = met2::plan_reserve(at.plan_lr, J.chunk);
    if (!rc) {
        DevGuard guard(opt.device);
        rc = pipeline(J, t, w, at);
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
    if (!plan_lr) {                                                                                       // detach
        std::lock_guard<std::mutex> lock(g_work_mutex);
        g_attach.erase(plan);
        return MET2_OK;
    }
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
    Attach at;
    at.plan_lr = plan_lr;
    at.alpha_lr.assign(alpha_lr, alpha_lr + n_lr);
    at.alpha_hr.assign(alpha_hr, alpha_hr + n_hr);
    std::lock_guard<std::mutex> lock(g_work_mutex);
    g_attach[plan] = std::move(at);
    return MET2_OK;
}

// frees the block buffers that destroyed plans left for the next plan on their device
extern "C" int met2_host_trim(void)
{
    std::vector<Work *> take;
    {
        std::lock_guard<std::mutex> lock(g_work_mutex);
        take.swap(g_pool);
    }
    for (Work *w : take) free_work(w);
    return MET2_OK;
}

namespace met2 {
// called by met2_plan_destroy: the block buffers, streams and events this entry keeps with a plan
__attribute__((visibility("hidden"))) void host_release(met2_plan *plan)
{
    Work *w = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_work_mutex);
        g_attach.erase(plan);
        for (auto it = g_attach.begin(); it != g_attach.end();)                  // a coarse plan that goes away takes its attachments with it
            it = (it->second.plan_lr == plan) ? g_attach.erase(it) : std::next(it);
        auto it = g_work.find(plan);
        if (it != g_work.end()) { w = it->second; g_work.erase(it); }
        if (w && w->s_in) {
            int same = 0;
            for (Work *q : g_pool) same += q->device == w->device;
            if (same < 2 && g_pool.size() < 32) { g_pool.push_back(w); w = nullptr; }
        }
    }
    free_work(w);
}
}  // namespace met2

static int fit_host_impl(met2_plan *const *plans, int32_t n_plans, int32_t method, int64_t nvox, const double *data, const double *fa_data,
                             int64_t voxel_stride, int64_t echo_stride, const double *mask_values, const double *fa_index, const uint8_t *mask,
                             int32_t estimate_fa, double *fsol, double *sig, double *reg, double *lam, double *maps, int32_t *status, double *fa_out,
                             double *fa_gate, int64_t chunk, double *plan_ms)
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
        std::vector<met2_plan *> coarse;
        for (int t = 0; t < n_plans; ++t) {
            auto it = g_attach.find(plans[t]);
            if (it == g_attach.end() || !it->second.plan_lr)
                return fail(MET2_E_STATE, "estimate_fa = 2 needs met2_plan_attach_fa_spline on every plan first");
            // every plan's thread runs the FA walk on ITS coarse plan (queue word, scratch): one coarse plan per plan, none of them fitted here
            met2_plan *lr = it->second.plan_lr;
            for (met2_plan *q : coarse) if (q == lr) return fail(MET2_E_INVALID, "met2_fit_host: the same coarse plan attached to two plans (one plan serves one stream at a time)");
            for (int u = 0; u < n_plans; ++u) if (plans[u] == lr) return fail(MET2_E_INVALID, "met2_fit_host: a coarse plan of the spline FA step is itself in plans[]");
            coarse.push_back(lr);
        }
    }
    if (chunk < 0) return fail(MET2_E_INVALID, "chunk < 0");
    if (chunk == 0) {
        // four blocks per plan (each plan splits its first and last one further), in multiples of 4 096 voxels, at most 262 144 (0.29 GB per
        // slot at 32 x 60).  Every block ends with the tail of its persistent kernel and a small clean-up pass (~1.9 ms), which speaks for
        // large blocks -- configs[1] (X2 on every voxel, 150 ms of kernels against 42 ms of copies), one plan, blocks of 262 144 / 524 288 /
        // 1 048 576 voxels: 166.1 / 161.6 / 160.7 ms -- but a block's download only hides under the NEXT block's kernels, which speaks for
        // small ones where the device is fast: the driver on the half-masked phantom (75 ms of kernels, the same 42 ms of copies) 0.085 s at
        // 262 144 against 0.090 s at 524 288.  The driver's case decides.
        // With many plans the share of each is small; blocks are then not cut below 65 536 voxels (a 6.5 ms fit against ~2 ms per block).
        const int64_t share = (nvox + n_plans - 1) / n_plans;
        const int64_t per = (nvox + (int64_t)n_plans * 4 - 1) / ((int64_t)n_plans * 4);
        chunk = std::min<int64_t>(262144, std::max<int64_t>(std::min<int64_t>(65536, share), (per + 4095) / 4096 * 4096));
    }
    chunk = std::min<int64_t>(chunk, nvox);
    if (chunk > 0x7fffffff) return fail(MET2_E_INVALID, "chunk out of range");
    // several plans: the list is dealt in runs of 4 096 voxels (pipeline()); the layouts whose blocks are moved by pitched copies already (rows
    // with a pitch, general strides) keep whole blocks.  MET2_HOST_BLOCKS=1: whole blocks always (A/B switch).
    const bool interleave = n_plans > 1 && (voxel_stride == 1 || (echo_stride == 1 && voxel_stride == nte)) && getenv("MET2_HOST_BLOCKS") == nullptr;
    if (interleave) chunk = (chunk + 4095) / 4096 * 4096;

    Job J;
    J.plans = plans; J.n_plans = n_plans; J.method = method; J.nvox = nvox; J.data = data; J.fa_data = fa_data; J.vs = voxel_stride; J.es = echo_stride;
    J.fa_index = fa_index; J.mask = mask; J.estimate_fa = estimate_fa; J.fsol = fsol; J.sig = sig; J.reg = reg; J.lam = lam; J.maps = maps;
    J.status = status; J.fa_out = fa_out; J.fa_gate = fa_gate; J.mask_values = mask_values; J.chunk = chunk; J.nte = nte; J.nt2 = nt2;
    J.run = interleave ? 4096 : 0;
    J.split = getenv("MET2_HOST_NOSPLIT") == nullptr;            // test / A-B switch: whole blocks only
    J.in_case = echo_stride == 1 ? 0 : (voxel_stride == 1 ? 1 : 2);
    if (J.in_case == 0 && voxel_stride < nte) return fail(MET2_E_INVALID, "met2_fit_host: voxel_stride < n_te with echo_stride 1 (overlapping voxels)");
    {   // data and fa_data may also lie in DEVICE memory (the volume as a whole-volume filter left it): copied block by block like pinned memory
        const int kd = mem_kind(data), kf = mem_kind(fa_data);
        if ((kd == 2 || kf == 2) && J.in_case == 2) return fail(MET2_E_UNSUPPORTED, "met2_fit_host: a device-resident volume needs voxel_stride 1 or echo_stride 1");
        const void *host_only[] = {fa_index, mask, mask_values, fsol, sig, reg, lam, maps, status, fa_out, fa_gate, plan_ms};
        for (const void *q : host_only) if (q && mem_kind(q) == 2) return fail(MET2_E_INVALID, "met2_fit_host: only data and fa_data may be device pointers");
        J.pin_data = kd != 0; J.pin_fa_data = kf != 0;
    }
    J.pin_fa = is_pinned(fa_index); J.pin_mask = is_pinned(mask); J.pin_fsol = is_pinned(fsol);
    J.pin_sig = is_pinned(sig); J.pin_reg = is_pinned(reg); J.pin_lam = is_pinned(lam); J.pin_maps = is_pinned(maps);
    J.pin_status = is_pinned(status); J.pin_fa_out = is_pinned(fa_out); J.pin_mv = is_pinned(mask_values); J.pin_gate = is_pinned(fa_gate);
    Regions &R = J.stage;
    R.in = !J.pin_data || J.in_case == 2;
    R.in_fa = fa_data && (!J.pin_fa_data || J.in_case == 2);
    R.fa = (fa_index && !J.pin_fa) || (fa_out && estimate_fa && !J.pin_fa_out);
    R.fsol = !J.pin_fsol; R.sig = sig && !J.pin_sig; R.reg = !J.pin_reg; R.lam = lam && !J.pin_lam; R.maps = maps && !J.pin_maps;
    R.status = status && !J.pin_status; R.mk = mask && !J.pin_mask; R.mv = mask_values && !J.pin_mv; R.gate = fa_gate && !J.pin_gate; R.nlr = 0;
    const bool need_pin = R.in || R.in_fa || R.fa || R.fsol || R.sig || R.reg || R.lam || R.maps || R.status || R.mk || R.mv || R.gate;

    std::vector<Outcome> res(n_plans);
    if (n_plans == 1) run_plan(J, 0, need_pin, false, &res[0]);
    else {
        std::vector<std::thread> th;
        th.reserve((size_t)n_plans);
        try {
            for (int t = 0; t < n_plans; ++t) th.emplace_back(run_plan, std::cref(J), t, need_pin, true, &res[t]);
        } catch (...) {                       // a thread could not be started: the ones that run finish their share first
            for (auto &x : th) x.join();
            throw;
        }
        for (auto &x : th) x.join();
    }
    if (plan_ms) for (int t = 0; t < n_plans; ++t) plan_ms[t] = res[t].ms;
    for (int t = 0; t < n_plans; ++t)
        if (res[t].rc) return fail(res[t].rc, "met2_fit_host, plan " + std::to_string(t) + ": " + res[t].msg);
    return MET2_OK;
}

// the C boundary lets no C++ exception through (a thread that cannot be started, an allocation that fails)
extern "C" int met2_fit_host(met2_plan *const *plans, int32_t n_plans, int32_t method, int64_t nvox, const double *data, const double *fa_data,
                             int64_t voxel_stride, int64_t echo_stride, const double *mask_values, const double *fa_index, const uint8_t *mask,
                             int32_t estimate_fa, double *fsol, double *sig, double *reg, double *lam, double *maps, int32_t *status, double *fa_out,
                             double *fa_gate, int64_t chunk, double *plan_ms)
{
    try {
        return fit_host_impl(plans, n_plans, method, nvox, data, fa_data, voxel_stride, echo_stride, mask_values, fa_index, mask, estimate_fa, fsol, sig, reg, lam,
                             maps, status, fa_out, fa_gate, chunk, plan_ms);
    } catch (const std::exception &e) {
        return fail(MET2_E_HIP, std::string("met2_fit_host: ") + e.what());
    } catch (...) {
        return fail(MET2_E_HIP, "met2_fit_host: unknown C++ exception");
    }
}
