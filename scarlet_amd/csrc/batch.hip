// C ABI of libscarlet_amd.so (declared in include/scarlet_amd.h): a batch of
// independent blends resident on one GPU and the proximal-gradient loop over it.
//
// Per iteration (the body of the proxmin.adaprox loop called at
// scarlet/blend.py:165-180):
//   render      Blend.get_model                       blend.py:200-244
//   conv        ConvolutionRenderer (FFT, cached K^)  renderer.py:247-259, fft.py:368-396
//   residual    -logL and w (m - d)                   observation.py:147-170
//   finalize    loss history + convergence test       blend.py:273, 294-299
//   conv^T      same transforms with conj(K^)         (vjp of fft.py:316-331)
//   update      gradient gather + AMSGrad + prox      lite/parameters.py:274-305
//
// FFT path: rocFFT batched 2-D real transforms over (blend, band).  The image is
// placed at the origin of the zero-padded (Fy, Fx) buffer and the kernel stamp is
// wrapped with its centre on index 0, which is the same linear convolution the
// reference gets with _pad / ifftshift / fftshift / _centered, without the shift
// copies.  The kernel spectrum is computed once (the reference recomputes it in
// every call, fft.py:387-388).
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <utility>

#include "common.h"

namespace smi {

static thread_local std::string g_error;
// hardware queues the runtime has, as far as the host told us (smi_set_hw_queues)
static std::atomic<int> g_hw_queues{4};
static std::atomic<long long> g_observation_uploads{0};  // smi_observation_uploads
// rocFFT plans kept between batches (see PlanCache below)
static std::mutex g_plan_mutex;
void set_error(const std::string &msg) { g_error = msg; }

static int next_fast_len(int n) {
    for (;; ++n) {
        int m = n;
        for (int p : {2, 3, 5})
            while (m % p == 0) m /= p;
        if (m == 1) return n;
    }
}

// fft.py:116-167 for axes (1, 2) of a (C, H, W) cube and a (Ck, ph, pw) kernel
static void reference_fft_shape(int H, int W, int ph, int pw, int *Fy, int *Fx) {
    int fy = next_fast_len(H + ph + 3), fx = next_fast_len(W + pw + 3);
    while (fx % 2) fx = next_fast_len(fx + 1);
    if (ph % 2 == 0)
        while (fy % 2) fy = next_fast_len(fy + 1);
    *Fy = fy;
    *Fx = fx;
}

#define SMI_FFT(expr)                                                          \
    do {                                                                       \
        rocfft_status _s = (expr);                                             \
        if (_s != rocfft_status_success) {                                     \
            smi::set_error(std::string(#expr) + ": rocfft status " +           \
                           std::to_string((int)_s));                           \
            return SMI_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

template <typename T>
static hipError_t dev_alloc(T **p, size_t n) {
    return hipMalloc(reinterpret_cast<void **>(p), (n ? n : 1) * sizeof(T));
}

}  // namespace smi

using namespace smi;

// ---------------------------------------------------------------------------------------
// rocFFT plans.  Creating a plan costs ~0.3 s of run-time kernel compilation, and a fit
// with box resizing, or an initialisation that renders every source, builds many batches
// of the same FFT shape one after the other: plans are kept between batches.
//
// Only ONE FFT shape per device is kept, and idle plans of another shape are destroyed
// before new ones are made, because rocFFT 7.2 returns wrong transforms from a plan
// created while a plan with the transposed complex shape is alive -- (Fy, Fx) next to
// (Fx / 2, 2 Fy), e.g. 60 x 60 next to 30 x 120, 64 x 64 next to 32 x 128
// (tools/rocfft_repro/repro.cpp shows it with rocFFT alone).  Batches that are alive at
// the same time with different shapes get plans of their own, and an automatically chosen
// shape steps aside (next fast length) if it is the transposed partner of a live one.
// ---------------------------------------------------------------------------------------
struct PlanPair {
    rocfft_plan fwd = nullptr, inv = nullptr;
};

struct PlanCache {
    int Fy = 0, Fx = 0, refs = 0;        // shape of the kept plans, batches using them
    std::map<int, PlanPair> by_count;    // number of transforms -> plans
};

static std::map<int, PlanCache> g_plan_cache;                       // per device
static std::map<std::tuple<int, int, int>, int> g_live_shapes;      // (device, Fy, Fx) -> batches

static int make_plans(int Fy, int Fx, int count, PlanPair *p) {
    const size_t lengths[2] = {(size_t)Fx, (size_t)Fy};
    SMI_FFT(rocfft_plan_create(&p->fwd, rocfft_placement_notinplace,
                               rocfft_transform_type_real_forward, rocfft_precision_single, 2,
                               lengths, (size_t)count, nullptr));
    SMI_FFT(rocfft_plan_create(&p->inv, rocfft_placement_notinplace,
                               rocfft_transform_type_real_inverse, rocfft_precision_single, 2,
                               lengths, (size_t)count, nullptr));
    return SMI_OK;
}

static void destroy_plans(PlanPair *p) {
    if (p->fwd) rocfft_plan_destroy(p->fwd);
    if (p->inv) rocfft_plan_destroy(p->inv);
    p->fwd = p->inv = nullptr;
}

// a live batch on this device whose complex shape is the transpose of (Fy, Fx)'s?
static bool transposed_partner_alive(int device, int Fy, int Fx) {
    for (const auto &kv : g_live_shapes) {
        if (kv.second <= 0 || std::get<0>(kv.first) != device) continue;
        const int Ly = std::get<1>(kv.first), Lx = std::get<2>(kv.first);
        if (Fy == Lx / 2 && Fx / 2 == Ly) return true;
    }
    return false;
}

// Called with the FFT shape a new rocFFT-path batch wants.  Drops idle plans of other
// shapes, moves an automatically chosen shape out of the way of a live transposed partner,
// registers the batch as alive and says whether it may use the kept plans (`shared`).
static int plans_enter(int device, bool shape_is_fixed, int *Fy, int *Fx, bool *shared) {
    std::lock_guard<std::mutex> guard(g_plan_mutex);
    PlanCache &c = g_plan_cache[device];
    if (c.refs == 0 && (c.Fy != *Fy || c.Fx != *Fx)) {
        for (auto &kv : c.by_count) destroy_plans(&kv.second);
        c.by_count.clear();
        c.Fy = c.Fx = 0;
    }
    while (transposed_partner_alive(device, *Fy, *Fx)) {
        SMI_REQUIRE(!shape_is_fixed,
                    "this FFT shape cannot coexist with a live batch of the transposed shape "
                    "(rocFFT returns wrong transforms); close that batch or choose another shape");
        *Fy = next_fast_len(*Fy + 1);
    }
    if (c.Fy == 0) {
        c.Fy = *Fy;
        c.Fx = *Fx;
    }
    *shared = c.Fy == *Fy && c.Fx == *Fx;
    if (*shared) ++c.refs;
    ++g_live_shapes[std::make_tuple(device, *Fy, *Fx)];
    return SMI_OK;
}

static void plans_leave(int device, int Fy, int Fx, bool shared) {
    std::lock_guard<std::mutex> guard(g_plan_mutex);
    if (shared) --g_plan_cache[device].refs;
    auto it = g_live_shapes.find(std::make_tuple(device, Fy, Fx));
    if (it != g_live_shapes.end() && --it->second <= 0) g_live_shapes.erase(it);
}

// plans for `count` transforms: the kept ones (`shared`) or fresh ones the caller owns
static int plans_get(int device, int Fy, int Fx, int count, bool shared, PlanPair *out) {
    if (!shared) return make_plans(Fy, Fx, count, out);
    std::lock_guard<std::mutex> guard(g_plan_mutex);
    PlanPair &p = g_plan_cache[device].by_count[count];
    if (!p.fwd) {
        const int rc = make_plans(Fy, Fx, count, &p);
        if (rc) return rc;
    }
    *out = p;
    return SMI_OK;
}

struct smi_batch {
    smi_batch_desc d{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool null_renderer = false;
    bool fused = false;       // LDS-resident convolution kernel instead of rocFFT
    int Fy = 0, Fx = 0, Fxh = 0;  // FFT shape
    int Py = 0, Px = 0;           // row/plane strides of the P / Q cubes
    float2 *Kt = nullptr;         // kernel spectrum in the fused kernel's order
    long long *dbg = nullptr;     // optional stage time stamps of workgroup 0 (debug)
    // FFT work cubes
    float *P = nullptr, *Q = nullptr;
    float2 *S = nullptr, *Khat = nullptr;
    void *work = nullptr;
    rocfft_plan plan_fwd = nullptr, plan_inv = nullptr;
    bool plans_registered = false, plans_shared = false;  // see PlanCache
    rocfft_execution_info info = nullptr;
    // observation
    float *data = nullptr, *weights = nullptr;
    float4 *dw = nullptr;  // data, weights by row pairs: BatchView::dw
    bool own_obs = false;
    double *log_norm = nullptr;
    // components
    int32_t *comp_start = nullptr, *c_blend = nullptr, *c_oy = nullptr, *c_ox = nullptr,
            *c_h = nullptr, *c_w = nullptr, *c_flags = nullptr, *c_plan = nullptr;
    int64_t *c_moff = nullptr;
    float *c_sed_min_step = nullptr, *c_sed_rel = nullptr, *c_morph_step = nullptr,
          *c_morph_rel = nullptr, *c_min_grad = nullptr, *c_lthresh = nullptr;
    float *sed = nullptr, *morph = nullptr;
    float *mom[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float *g_sed = nullptr, *g_morph = nullptr;
    float *xp_tmp = nullptr;  // BatchView::xp_tmp
    int n_size_classes = 0;   // size classes of the register-resident update kernels in use
    // point sources: {offset, m, v, vhat} x (y, x) per component, gradient, PSF sigma
    double *pt = nullptr, *g_center = nullptr;
    float *c_sigma = nullptr, *c_beta = nullptr;
    int n_point = 0;
    std::vector<double> box_center;  // mean of the box bounds per component (y, x)
    std::vector<char> is_point;
    // free Fourier shifts
    int n_shift = 0, max_box_side = 1, max_box_w = 1;
    float *shift_scratch = nullptr;  // BatchView::shift_scratch
    bool inline_render = true;  // smi_batch_set_inline_render
    float *morph_param = nullptr, *c_shift_step = nullptr, *c_shift_rel = nullptr;
    int32_t *c_shift_fft = nullptr;
    std::vector<char> is_shift;
    std::vector<int64_t> h_moff;
    std::vector<int32_t> h_blend;  // owning blend of every component
    // scarlet.lite
    float *c_center_floor = nullptr, *c_bg_level = nullptr, *c_fista_step = nullptr;
    float *c_sym_strength = nullptr;
    int32_t *c_chain_repeat = nullptr;
    int32_t mono_mask = 0;  // a component carries SMI_PROX_MONO_MASK
    float *c_pos_floor = nullptr;
    double *fista_t = nullptr;
    // free shift of the difference kernel (smi_batch_set_kernel_shift); ks.stamp == nullptr: none
    smi::KernelShiftView ks{};
    float *ks_resid = nullptr;  // rendered cube / weighted residual [nb][C][H][W]
    double2 *ks_tmp = nullptr;  // scratch of launch_stamp_spectrum
    // smi_batch_save_state: device copies of everything a step changes
    // {sed, morph, morph_param, m/v/vhat x2, fista_t, pt, kernel shift} and their sizes in bytes
    void *saved[12] = {};
    size_t saved_bytes[12] = {};
    int scheme = SMI_SCHEME_AMSGRAD;
    bool include_log_norm = true;
    bool lite_flags = false;  // some component uses FIT_CENTER / BG_THRESH
    int32_t *have_prev = nullptr;
    float *scratch = nullptr;  // update-kernel state of boxes too large for the LDS
    int64_t n_morph = 0;
    bool have_components = false, have_obs = false, have_kernel = false;
    // further observations on coarser grids (one blend) and their loss terms (device)
    std::vector<smi::LowRes *> lowres;
    double *extra_terms = nullptr;
    // ranges of blends stepped on streams of their own (blends are independent): the tail
    // of one range's update kernel overlaps the next iteration's convolution of the others
    int n_sub = 0;  // 0 = automatic
    std::vector<int32_t> h_comp_start;
    // further observations of the batch on the model's pixel grid (smi_batch_add_observation):
    // own data, weights and kernel spectrum; their loss partials follow layer 0's in
    // loss_partial ([nb][(1 + n_layers) C]), their gradient images are added to Q via Q2
    struct ObsLayer {
        float *data = nullptr, *weights = nullptr;
        float2 *Kt = nullptr;
        float4 *dw = nullptr;
    };
    std::vector<ObsLayer> layers;
    float *Q2 = nullptr;
    // work items of the register-resident update kernels (common.h, BatchView::work)
    int32_t *work_items = nullptr;
    int32_t stage_plan[kNumUpdateClasses];
    std::vector<int32_t> h_work_start;  // [kNumUpdateClasses][n_blends + 1]
    std::vector<hipStream_t> sub_streams;
    std::vector<hipEvent_t> sub_events;  // [0] fork, [s] join of range s >= 1
    // per blend
    int32_t *state = nullptr, *zero_state = nullptr, *n_loss = nullptr, *status_out = nullptr;
    int32_t *it_base = nullptr;  // BatchView::it_base (smi_batch_set_iteration_base)
    int32_t *pause_at = nullptr, *conv_flag = nullptr;  // smi_batch_set_pause_at
    std::vector<char> plan_shared;  // plans[i] belongs to the plan cache (smi_batch_add_sweep_plan)
    int32_t *h_round = nullptr;  // pinned staging of smi_batch_set_round / get_round, 3 x n_blends
    double *loss_hist = nullptr, *last_loss = nullptr, *loss_partial = nullptr;
    // plans
    std::vector<SweepPlanDev> plans;
    SweepPlanDev *d_plans = nullptr;
    int max_levels = 0;
    BatchView view{};
    float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> events;
    double phase_ms[6] = {0, 0, 0, 0, 0, 0};
    int timed_iters = 0;
};

namespace {

void refresh_view(smi_batch *b) {
    BatchView &v = b->view;
    v.nb = b->d.n_blends;
    v.C = b->d.C;
    v.H = b->d.H;
    v.W = b->d.W;
    v.Fy = b->Py;
    v.Fx = b->Px;
    v.n_comp = b->d.n_components;
    v.n_comp_total = b->d.n_components;
    v.comp_start = b->comp_start;
    v.c_blend = b->c_blend;
    v.c_oy = b->c_oy;
    v.c_ox = b->c_ox;
    v.c_h = b->c_h;
    v.c_w = b->c_w;
    v.c_flags = b->c_flags;
    v.c_plan = b->c_plan;
    v.c_moff = b->c_moff;
    v.c_sed_min_step = b->c_sed_min_step;
    v.c_sed_rel = b->c_sed_rel;
    v.c_morph_step = b->c_morph_step;
    v.c_morph_rel = b->c_morph_rel;
    v.c_min_grad = b->c_min_grad;
    v.c_lthresh = b->c_lthresh;
    v.sed = b->sed;
    v.morph = b->morph;
    v.m_sed = b->mom[0];
    v.v_sed = b->mom[1];
    v.vh_sed = b->mom[2];
    v.m_morph = b->mom[3];
    v.v_morph = b->mom[4];
    v.vh_morph = b->mom[5];
    v.data = b->data;
    v.weights = b->weights;
    v.dw = b->dw;
    v.it_base = b->it_base;
    v.pause_at = b->pause_at;
    v.conv_flag = b->conv_flag;
    v.log_norm = b->log_norm;
    v.state = b->state;
    v.n_loss = b->n_loss;
    v.loss_hist = b->loss_hist;
    v.hist_cap = b->d.max_iter;
    v.last_loss = b->last_loss;
    v.have_prev = b->have_prev;
    v.scratch = b->scratch;
    v.loss_partial = b->loss_partial;
    v.n_partial = b->fused ? b->d.C * (1 + (int)b->layers.size()) : (b->d.H * b->d.W + 255) / 256;
    v.plans = b->d_plans;
    v.max_levels = b->max_levels;
    v.fast_plans = 1;
    v.mono_mask = b->mono_mask;
    v.b1 = b->b1;
    v.b2 = b->b2;
    v.eps = b->eps;
    v.n_point = b->n_point;
    v.pt = b->pt;
    v.c_sigma = b->c_sigma;
    v.c_beta = b->c_beta;
    v.n_shift = b->n_shift;
    v.max_box_side = b->max_box_side;
    v.shift_scratch = b->shift_scratch;
    v.morph_param = b->morph_param;
    v.g_sed_buf = b->g_sed;
    v.g_morph_buf = b->g_morph;
    v.xp_tmp = b->xp_tmp;
    v.n_morph_total = b->n_morph;
    v.c_shift_step = b->c_shift_step;
    v.c_shift_rel = b->c_shift_rel;
    v.c_shift_fft = b->c_shift_fft;
    v.c_center_floor = b->c_center_floor;
    v.c_sym_strength = b->c_sym_strength;
    v.c_chain_repeat = b->c_chain_repeat;
    v.c_pos_floor = b->c_pos_floor;
    v.c_bg_level = b->c_bg_level;
    v.scheme = b->scheme;
    v.lite = b->scheme == SMI_SCHEME_FISTA || b->lite_flags;
    v.c_fista_step = b->c_fista_step;
    v.fista_t = b->fista_t;
    v.extra_term = b->extra_terms;
    v.n_extra = (int32_t)b->lowres.size();
    v.blend0 = 0;
    v.comp0 = 0;
    v.work = b->work_items;
    v.work0 = 0;
    v.work_start = b->h_work_start.data();
    v.nb_total = b->d.n_blends;
    for (int cls = 0; cls < kNumUpdateClasses; ++cls) {
        const int p = b->have_components ? b->stage_plan[cls] : -1;
        // One-wavefront classes, one plane (boxes up to 47^2).  The two-plane schedule of
        // larger boxes is bit-exact too but slower than the level plan -- 61^2: 32.8 k clocks
        // per sweep against 28.4 k, 51^2: 28.5 k against 20.3 k (tools/sweep_cycles.py): a lane
        // issues two pixels per step while at most eleven of its sixteen rings are under way
        // -- so those boxes keep the level plan.
        // SMI_RING_LATE=0 (development aid): plans with late rings (boxes of 49^2 .. 63^2) keep
        // the level plan
        static const bool may_late = [] {
            const char *e = getenv("SMI_RING_LATE");
            return !e || atoi(e) != 0;
        }();
        const bool ok = p >= 0 && p < (int)b->plans.size() && b->plans[p].ring &&
                        kUpdateTeam[cls] == 64 && b->plans[p].ring_planes == 1 &&
                        (may_late || b->plans[p].ring_nat == b->plans[p].ring_pad);
        v.stage_plan[cls] = ok ? p : -1;
        v.stage_bytes[cls] = ok ? b->plans[p].ring_bytes : 0;
    }
    {
        const int slots = (b->max_box_w + 14) / 16 + 1;
        v.render_slots = slots <= 6 ? slots : 0;
    }
    for (const auto &pl : b->plans)
        if (!pl.slots) v.fast_plans = 0;
}

int ready(smi_batch *b) {
    SMI_REQUIRE(b != nullptr, "null batch");
    SMI_REQUIRE(b->have_components, "smi_batch_set_components has not been called");
    SMI_REQUIRE(b->have_obs, "smi_batch_set_observation has not been called");
    SMI_REQUIRE(b->null_renderer || b->have_kernel, "smi_batch_set_kernel has not been called");
    SMI_HIP(hipSetDevice(b->device));
    return SMI_OK;
}

int fft_exec(smi_batch *b, rocfft_plan plan, void *in, void *out) {
    void *ins[1] = {in}, *outs[1] = {out};
    SMI_FFT(rocfft_execute(plan, ins, outs, b->info));
    return SMI_OK;
}

// rendered = model (*) kernel, or its transpose; in: P, out: Q
int convolve(smi_batch *b, const BatchView &v, int conj) {
    if (b->null_renderer) return SMI_OK;
    if (b->fused) {
        set_error("internal: rocFFT convolution requested on the fused path");
        return SMI_ERR_INVALID;
    }
    int rc = fft_exec(b, b->plan_fwd, b->P, b->S);
    if (rc) return rc;
    launch_cmul(b->S, b->Khat, v.nb, v.C, (int64_t)b->Fy * b->Fxh, b->d.kernel_bands,
                b->d.kernel_per_blend, conj, v.state, b->stream);
    return fft_exec(b, b->plan_inv, b->S, b->Q);
}

BatchView unmasked_view(smi_batch *b) {
    BatchView v = b->view;
    v.state = b->zero_state;
    return v;
}

int upload_plans(smi_batch *b) {
    if (b->d_plans) {
        SMI_HIP(hipFree(b->d_plans));
        b->d_plans = nullptr;
    }
    if (!b->plans.empty()) {
        SMI_HIP(dev_alloc(&b->d_plans, b->plans.size()));
        SMI_HIP(hipMemcpy(b->d_plans, b->plans.data(), b->plans.size() * sizeof(SweepPlanDev),
                          hipMemcpyHostToDevice));
    }
    refresh_view(b);
    return SMI_OK;
}

template <typename T>
int upload(T **dst, const T *src, size_t n) {
    if (*dst) {
        SMI_HIP(hipFree(*dst));
        *dst = nullptr;
    }
    SMI_HIP(dev_alloc(dst, n));
    if (n) SMI_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return SMI_OK;
}

template <typename T>
int seam1_sweep(T *flat_img, const T *weights, const int32_t *offsets, int32_t n_off,
                const int32_t *dist_idx, int32_t n_idx, int32_t n_pix, T min_gradient) {
    SMI_REQUIRE(flat_img && weights && offsets && (dist_idx || n_idx == 0), "null argument");
    SMI_REQUIRE(n_pix > 0 && n_off > 0 && n_idx >= 0, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available");
        return SMI_ERR_NO_DEVICE;
    }
    std::vector<double> w64((size_t)n_off * n_pix);
    for (size_t i = 0; i < w64.size(); ++i) w64[i] = (double)weights[i];
    SweepPlanHost plan;
    if (!build_sweep_plan(n_pix, w64.data(), offsets, n_off, dist_idx, n_idx, &plan))
        return SMI_ERR_INVALID;
    return sweep_host_buffers<T>(flat_img, n_pix, plan, min_gradient);
}

template <typename T>
int seam1_sweep_many(int32_t n_img, T *images, int32_t n_pix, const T *weights,
                     const int32_t *offsets, int32_t n_off, const int32_t *dist_idx,
                     int32_t n_idx, T min_gradient) {
    SMI_REQUIRE(n_img >= 0 && n_pix > 0 && n_off > 0 && n_idx >= 0, "bad sizes");
    if (n_img == 0) return SMI_OK;
    SMI_REQUIRE(images && weights && offsets && (dist_idx || n_idx == 0), "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available");
        return SMI_ERR_NO_DEVICE;
    }
    // the plans on the host cores, a few images per thread (a 128 x 128 plan takes ~1 ms)
    std::vector<SweepPlanHost> plans(n_img);
    std::vector<char> ok(n_img, 0);
    const int n_threads = std::max(1, std::min<int>(n_img, std::min(8u, std::thread::hardware_concurrency())));
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t)
        pool.emplace_back([&, t] {
            std::vector<double> w64((size_t)n_off * n_pix);
            for (int i = t; i < n_img; i += n_threads) {
                const T *w = weights + (size_t)i * n_off * n_pix;
                for (size_t q = 0; q < w64.size(); ++q) w64[q] = (double)w[q];
                ok[i] = build_sweep_plan(n_pix, w64.data(), offsets, n_off,
                                         dist_idx + (size_t)i * n_idx, n_idx, &plans[i]);
            }
        });
    for (auto &th : pool) th.join();
    for (int i = 0; i < n_img; ++i)
        if (!ok[i]) {
            set_error("malformed monotonicity tables of image " + std::to_string(i));
            return SMI_ERR_INVALID;
        }
    return sweep_many_host_buffers<T>(images, n_img, n_pix, plans, min_gradient);
}

template <typename T>
int seam1_filter(const T *image, int32_t H, int32_t W, const T *values, int32_t n_taps,
                 const int32_t *ys, const int32_t *ye, const int32_t *xs, const int32_t *xe,
                 T *result) {
    SMI_REQUIRE(image && values && ys && ye && xs && xe && result, "null argument");
    SMI_REQUIRE(H > 0 && W > 0 && n_taps >= 0, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available");
        return SMI_ERR_NO_DEVICE;
    }
    return apply_filter_host_buffers<T>(image, H, W, values, n_taps, ys, ye, xs, xe, result);
}

}  // namespace

namespace {
template <typename T>
int seam1_mask_valid(int32_t i, int32_t j, const T *image, int32_t rows, int32_t cols,
                     uint8_t *unchecked, uint8_t *orphans, double variance, int32_t *bounds,
                     double thresh) {
    SMI_REQUIRE(image && unchecked && orphans && bounds, "null argument");
    SMI_REQUIRE(rows > 0 && cols > 0 && i >= 0 && i < rows && j >= 0 && j < cols, "bad indices");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available");
        return SMI_ERR_NO_DEVICE;
    }
    return mask_valid_host_buffers<T>(i, j, image, rows, cols, unchecked, orphans, variance,
                                      bounds, thresh);
}

template <typename T>
int seam1_mask_interpolate(const int32_t *ri, const int32_t *ci, int32_t n, uint8_t *unchecked,
                           T *model, int32_t rows, int32_t cols, uint8_t *orphans,
                           double variance, int32_t recursive, int32_t *bounds) {
    SMI_REQUIRE((ri && ci) || n == 0, "null index arrays");
    SMI_REQUIRE(model && unchecked && orphans && bounds && rows > 0 && cols > 0 && n >= 0,
                "null argument / bad sizes");
    for (int32_t k = 0; k < n; ++k)
        SMI_REQUIRE(ri[k] >= 0 && ri[k] < rows && ci[k] >= 0 && ci[k] < cols, "index out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available");
        return SMI_ERR_NO_DEVICE;
    }
    return mask_interpolate_host_buffers<T>(ri, ci, n, unchecked, model, rows, cols, orphans,
                                            variance, recursive, bounds);
}
}  // namespace

extern "C" {

const char *smi_last_error(void) { return g_error.c_str(); }

const char *smi_version(void) { return "scarlet_amd 0.1 gfx950"; }

int smi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int smi_prox_weighted_monotonic_f32(float *flat_img, const float *weights,
                                    const int32_t *offsets, int32_t n_off,
                                    const int32_t *dist_idx, int32_t n_idx, int32_t n_pix,
                                    float min_gradient) {
    return seam1_sweep<float>(flat_img, weights, offsets, n_off, dist_idx, n_idx, n_pix,
                              min_gradient);
}
int smi_prox_weighted_monotonic_f64(double *flat_img, const double *weights,
                                    const int32_t *offsets, int32_t n_off,
                                    const int32_t *dist_idx, int32_t n_idx, int32_t n_pix,
                                    double min_gradient) {
    return seam1_sweep<double>(flat_img, weights, offsets, n_off, dist_idx, n_idx, n_pix,
                               min_gradient);
}
int smi_prox_weighted_monotonic_many_f32(int32_t n_img, float *images, int32_t n_pix,
                                         const float *weights, const int32_t *offsets,
                                         int32_t n_off, const int32_t *dist_idx, int32_t n_idx,
                                         float min_gradient) {
    return seam1_sweep_many<float>(n_img, images, n_pix, weights, offsets, n_off, dist_idx, n_idx,
                                   min_gradient);
}
int smi_prox_weighted_monotonic_many_f64(int32_t n_img, double *images, int32_t n_pix,
                                         const double *weights, const int32_t *offsets,
                                         int32_t n_off, const int32_t *dist_idx, int32_t n_idx,
                                         double min_gradient) {
    return seam1_sweep_many<double>(n_img, images, n_pix, weights, offsets, n_off, dist_idx, n_idx,
                                    min_gradient);
}
int smi_apply_filter_f32(const float *image, int32_t H, int32_t W, const float *values,
                         int32_t n_taps, const int32_t *y_start, const int32_t *y_end,
                         const int32_t *x_start, const int32_t *x_end, float *result) {
    return seam1_filter<float>(image, H, W, values, n_taps, y_start, y_end, x_start, x_end,
                               result);
}
int smi_apply_filter_f64(const double *image, int32_t H, int32_t W, const double *values,
                         int32_t n_taps, const int32_t *y_start, const int32_t *y_end,
                         const int32_t *x_start, const int32_t *x_end, double *result) {
    return seam1_filter<double>(image, H, W, values, n_taps, y_start, y_end, x_start, x_end,
                                result);
}

struct smi_resampler {
    smi::Resampler *impl = nullptr;
};

int smi_resampler_create(const float *A, const float *Pt, int32_t C, int32_t n_a, int32_t n_b,
                         int32_t Fy, int32_t Fx, smi_resampler **out) {
    SMI_REQUIRE(A && Pt && out, "null argument");
    SMI_REQUIRE(C > 0 && n_a > 0 && n_b > 0 && Fy > 0 && Fx > 0, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available");
        return SMI_ERR_NO_DEVICE;
    }
    auto *h = new smi_resampler;
    const int rc = resampler_create(A, Pt, C, n_a, n_b, Fy, Fx, &h->impl);
    if (rc) {
        resampler_destroy(h->impl);
        delete h;
        return rc;
    }
    *out = h;
    return SMI_OK;
}

int smi_resampler_render(smi_resampler *r, const float *model, float *out) {
    SMI_REQUIRE(r && r->impl && model && out, "null argument");
    return resampler_render(r->impl, model, out);
}

int smi_resampler_time(smi_resampler *r, int32_t n_rep, double *ms_per_render) {
    SMI_REQUIRE(r && r->impl && ms_per_render && n_rep > 0, "bad argument");
    return resampler_time(r->impl, n_rep, ms_per_render);
}

int smi_resampler_get_path(smi_resampler *r, int32_t *path) {
    SMI_REQUIRE(r && r->impl && path, "null argument");
    *path = resampler_get_path(r->impl);
    return SMI_OK;
}

int smi_resampler_set_path(smi_resampler *r, int32_t path) {
    SMI_REQUIRE(r && r->impl, "null argument");
    return resampler_set_path(r->impl, path);
}

static constexpr int kMaxLowRes = 8;

int smi_batch_attach_lowres(smi_batch *b, smi_resampler *r, const int32_t *channels,
                            const float *data, const float *weights, double log_norm) {
    SMI_REQUIRE(b && r && channels && data && weights, "null argument");
    SMI_REQUIRE(b->d.n_blends == 1, "a low-resolution observation needs a batch of one blend");
    SMI_REQUIRE(r->impl, "resampler already destroyed");
    SMI_REQUIRE((int)b->lowres.size() < kMaxLowRes, "too many low-resolution observations");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    if (!b->extra_terms) {
        SMI_HIP(dev_alloc(&b->extra_terms, kMaxLowRes));
        SMI_HIP(hipMemset(b->extra_terms, 0, kMaxLowRes * sizeof(double)));
    }
    LowRes *l = nullptr;
    int rc = lowres_create(r->impl, channels, data, weights, log_norm, b->d.H, b->d.W,
                           b->extra_terms + b->lowres.size(), &l);
    if (rc) {
        lowres_destroy(l);
        return rc;
    }
    b->lowres.push_back(l);
    refresh_view(b);
    return SMI_OK;
}

int smi_batch_get_lowres_rendered(smi_batch *b, int32_t index, float *out) {
    SMI_REQUIRE(b && out, "null argument");
    SMI_REQUIRE(index >= 0 && index < (int)b->lowres.size(), "no such low-resolution observation");
    return lowres_get_rendered(b->lowres[index], out, b->stream);
}

int smi_resampler_destroy(smi_resampler *r) {
    if (!r) return SMI_OK;
    resampler_destroy(r->impl);
    delete r;
    return SMI_OK;
}

int smi_get_valid_monotonic_pixels_f32(int32_t i, int32_t j, const float *image, int32_t rows,
                                       int32_t cols, uint8_t *unchecked, uint8_t *orphans,
                                       double variance, int32_t *bounds, double thresh) {
    return seam1_mask_valid<float>(i, j, image, rows, cols, unchecked, orphans, variance, bounds, thresh);
}
int smi_get_valid_monotonic_pixels_f64(int32_t i, int32_t j, const double *image, int32_t rows,
                                       int32_t cols, uint8_t *unchecked, uint8_t *orphans,
                                       double variance, int32_t *bounds, double thresh) {
    return seam1_mask_valid<double>(i, j, image, rows, cols, unchecked, orphans, variance, bounds, thresh);
}
int smi_linear_interpolate_invalid_pixels_f32(const int32_t *ri, const int32_t *ci, int32_t n,
                                              uint8_t *unchecked, float *model, int32_t rows,
                                              int32_t cols, uint8_t *orphans, double variance,
                                              int32_t recursive, int32_t *bounds) {
    return seam1_mask_interpolate<float>(ri, ci, n, unchecked, model, rows, cols, orphans, variance,
                                         recursive, bounds);
}
int smi_linear_interpolate_invalid_pixels_f64(const int32_t *ri, const int32_t *ci, int32_t n,
                                              uint8_t *unchecked, double *model, int32_t rows,
                                              int32_t cols, uint8_t *orphans, double variance,
                                              int32_t recursive, int32_t *bounds) {
    return seam1_mask_interpolate<double>(ri, ci, n, unchecked, model, rows, cols, orphans,
                                          variance, recursive, bounds);
}

static int batch_create_impl(const smi_batch_desc *desc, int device, smi_batch **out,
                             smi_batch **partial) {
    SMI_REQUIRE(desc && out, "null argument");
    SMI_REQUIRE(desc->n_blends > 0 && desc->n_blends <= 65535, "n_blends must be in [1, 65535]");
    SMI_REQUIRE(desc->C > 0 && desc->C <= 64, "C must be in [1, 64]");
    SMI_REQUIRE(desc->H > 0 && desc->W > 0, "empty frame");
    SMI_REQUIRE(desc->n_components >= 0, "negative component count");
    SMI_REQUIRE(desc->max_iter > 0, "max_iter must be positive");
    SMI_REQUIRE((desc->kernel_h == 0) == (desc->kernel_w == 0), "kernel_h/kernel_w mismatch");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device available");
        return SMI_ERR_NO_DEVICE;
    }
    SMI_REQUIRE(device >= 0 && device < ndev, "device index out of range");
    SMI_HIP(hipSetDevice(device));

    smi_batch *b = new smi_batch();
    *partial = b;  // released by the caller if anything below fails
    b->d = *desc;
    b->device = device;
    b->null_renderer = desc->kernel_h == 0;
    const int nb = desc->n_blends, C = desc->C, H = desc->H, W = desc->W;
    if (b->null_renderer) {
        b->Fy = H;
        b->Fx = W;
    } else {
        SMI_REQUIRE(desc->kernel_bands == 1 || desc->kernel_bands == C,
                    "kernel_bands must be 1 or C");
        if (desc->fft_h > 0 && desc->fft_w > 0) {
            // alias-free 'same' output needs F >= N + P/2 (image at the origin)
            SMI_REQUIRE(desc->fft_h >= H + desc->kernel_h / 2 && desc->fft_w >= W + desc->kernel_w / 2,
                        "fft shape too small for an alias-free convolution");
            SMI_REQUIRE(desc->fft_w % 2 == 0, "fft_w must be even");
            b->Fy = desc->fft_h;
            b->Fx = desc->fft_w;
            b->fused = desc->conv_path != 1 && fused_conv_instantiated(b->Fy, b->Fx);
        } else {
            int fy = 0, fx = 0;
            if (desc->conv_path != 1 &&
                fused_conv_choose(H + desc->kernel_h / 2, W + desc->kernel_w / 2, &fy, &fx)) {
                b->fused = true;
                b->Fy = fy;
                b->Fx = fx;
            } else {
                reference_fft_shape(H, W, desc->kernel_h, desc->kernel_w, &b->Fy, &b->Fx);
            }
        }
        SMI_REQUIRE(desc->conv_path != 2 || b->fused, "fused convolution not available for this shape");
    }
    const bool padded = !b->null_renderer && !b->fused;
    if (padded) {
        const bool fixed = desc->fft_h > 0 && desc->fft_w > 0;
        const int rc_enter = plans_enter(device, fixed, &b->Fy, &b->Fx, &b->plans_shared);
        if (rc_enter) return rc_enter;
        b->plans_registered = true;
    }
    b->Fxh = b->Fx / 2 + 1;
    b->Py = padded ? b->Fy : H;
    b->Px = padded ? b->Fx : W;
    const size_t n_real = (size_t)nb * C * b->Py * b->Px;
    SMI_HIP(dev_alloc(&b->P, n_real));
    SMI_HIP(hipMemset(b->P, 0, n_real * sizeof(float)));
    if (b->null_renderer) {
        b->Q = b->P;
    } else {
        SMI_HIP(dev_alloc(&b->Q, n_real));
        SMI_HIP(hipMemset(b->Q, 0, n_real * sizeof(float)));
    }
    if (padded) {
        {
            // once per process: every call re-opens rocFFT's run-time kernel cache
            std::lock_guard<std::mutex> guard(g_plan_mutex);
            static bool rocfft_ready = false;
            if (!rocfft_ready) {
                SMI_FFT(rocfft_setup());
                rocfft_ready = true;
            }
        }
        const size_t n_cplx = (size_t)nb * C * b->Fy * b->Fxh;
        SMI_HIP(dev_alloc(&b->S, n_cplx));
        {
            PlanPair pp;
            const int rc_plans = plans_get(device, b->Fy, b->Fx, nb * C, b->plans_shared, &pp);
            b->plan_fwd = pp.fwd;
            b->plan_inv = pp.inv;
            if (rc_plans) return rc_plans;
        }
        size_t w1 = 0, w2 = 0;
        SMI_FFT(rocfft_plan_get_work_buffer_size(b->plan_fwd, &w1));
        SMI_FFT(rocfft_plan_get_work_buffer_size(b->plan_inv, &w2));
        const size_t wb = w1 > w2 ? w1 : w2;
        SMI_FFT(rocfft_execution_info_create(&b->info));
        if (wb) {
            SMI_HIP(hipMalloc(&b->work, wb));
            SMI_FFT(rocfft_execution_info_set_work_buffer(b->info, b->work, wb));
        }
    }
    SMI_HIP(dev_alloc(&b->state, nb));
    SMI_HIP(dev_alloc(&b->zero_state, nb));
    SMI_HIP(dev_alloc(&b->n_loss, nb));
    SMI_HIP(dev_alloc(&b->status_out, 2));
    SMI_HIP(dev_alloc(&b->loss_hist, (size_t)nb * desc->max_iter));
    SMI_HIP(dev_alloc(&b->last_loss, nb));
    SMI_HIP(dev_alloc(&b->have_prev, nb));
    SMI_HIP(hipMemset(b->have_prev, 0, nb * sizeof(int32_t)));
    SMI_HIP(dev_alloc(&b->log_norm, nb));
    SMI_HIP(dev_alloc(&b->loss_partial, (size_t)nb * std::max(C, (H * W + 255) / 256)));
    SMI_HIP(hipMemset(b->state, 0, nb * sizeof(int32_t)));
    SMI_HIP(hipMemset(b->zero_state, 0, nb * sizeof(int32_t)));
    SMI_HIP(hipMemset(b->n_loss, 0, nb * sizeof(int32_t)));
    SMI_HIP(hipMemset(b->last_loss, 0, nb * sizeof(double)));
    refresh_view(b);
    *out = b;
    *partial = nullptr;
    return SMI_OK;
}

int smi_batch_create(const smi_batch_desc *desc, int device, smi_batch **out) {
    smi_batch *partial = nullptr;
    const int rc = batch_create_impl(desc, device, out, &partial);
    if (rc != SMI_OK && partial) {
        const std::string msg = g_error;  // destroy() must not clobber the reason
        smi_batch_destroy(partial);
        g_error = msg;
    }
    return rc;
}

static void release_kernel_shift(smi_batch *b);
static int refresh_shifted_kernel(smi_batch *b, int respect_state);

int smi_batch_destroy(smi_batch *b) {
    if (!b) return SMI_OK;
    (void)hipSetDevice(b->device);
    (void)hipDeviceSynchronize();
    if (b->info) rocfft_execution_info_destroy(b->info);
    if (b->plans_registered) {
        if (!b->plans_shared) {
            PlanPair own;
            own.fwd = b->plan_fwd;
            own.inv = b->plan_inv;
            destroy_plans(&own);
        }
        plans_leave(b->device, b->Fy, b->Fx, b->plans_shared);
    }
    for (size_t i = 0; i < b->plans.size(); ++i) {
        auto &pl = b->plans[i];
        if (i < b->plan_shared.size() && b->plan_shared[i]) continue;  // (the plan cache's)
        (void)hipFree(pl.level_start);
        (void)hipFree(pl.pix);
        (void)hipFree(pl.cnt);
        (void)hipFree(pl.nbr);
        (void)hipFree(pl.wt);
        if (pl.slots) (void)hipFree(pl.slots);
        if (pl.ring) (void)hipFree(const_cast<void *>(pl.ring));
    }
    void *bufs[] = {b->P, b->null_renderer ? nullptr : (void *)b->Q, b->S, b->Khat, b->Kt, b->work,
                    b->own_obs ? b->data : nullptr, b->own_obs ? b->weights : nullptr, b->dw,
                    b->log_norm, b->comp_start, b->c_blend, b->c_oy, b->c_ox, b->c_h, b->c_w,
                    b->c_flags, b->c_plan, b->c_moff, b->c_sed_min_step, b->c_sed_rel,
                    b->c_morph_step, b->c_morph_rel, b->c_min_grad, b->c_lthresh, b->sed,
                    b->morph, b->mom[0], b->mom[1], b->mom[2], b->mom[3], b->mom[4], b->mom[5],
                    b->g_sed, b->g_morph, b->xp_tmp, b->pt, b->g_center, b->c_sigma, b->c_beta, b->morph_param,
                    b->c_shift_step, b->c_shift_rel, b->c_shift_fft, b->c_center_floor, b->c_sym_strength, b->c_chain_repeat, b->c_pos_floor, b->c_bg_level,
                    b->c_fista_step, b->fista_t, b->have_prev, b->scratch, b->state, b->zero_state, b->n_loss, b->status_out, b->it_base, b->pause_at, b->conv_flag, b->shift_scratch,
                    b->loss_hist, b->last_loss, b->loss_partial, b->d_plans, b->work_items};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    for (auto e : b->events) (void)hipEventDestroy(e);
    for (auto e : b->sub_events) (void)hipEventDestroy(e);
    for (auto st : b->sub_streams) (void)hipStreamDestroy(st);
    if (b->h_round) (void)hipHostFree(b->h_round);
    for (auto *l : b->lowres) lowres_destroy(l);
    for (auto &l : b->layers)
        for (void *p : {(void *)l.data, (void *)l.weights, (void *)l.Kt, (void *)l.dw})
            if (p) (void)hipFree(p);
    for (void *p : b->saved)
        if (p) (void)hipFree(p);
    release_kernel_shift(b);
    if (b->Q2) (void)hipFree(b->Q2);
    if (b->extra_terms) (void)hipFree(b->extra_terms);
    delete b;
    return SMI_OK;
}

// Sweep plans by content.  A plan -- level plan, slot entries, ring stream: ~1 MB for a 61 x 61
// box -- depends on the tables alone, and every fit of a scene asks for the same three or four
// (a scene's Blend.fit spent 1 ms of its 18 building and uploading them).  The device copies
// are therefore kept per device under a hash of the tables and shared by all batches; they are
// never freed (at most kPlanCacheLimit of them; beyond that a batch owns its plans again).
namespace {
constexpr size_t kPlanCacheLimit = 512;
struct PlanKey {
    int device, h, w, n_idx;
    uint64_t hash;
    bool operator<(const PlanKey &o) const {
        return std::tie(device, h, w, n_idx, hash) < std::tie(o.device, o.h, o.w, o.n_idx, o.hash);
    }
};
std::mutex g_sweep_plans_mu;
std::map<PlanKey, SweepPlanDev> g_sweep_plans;

uint64_t fnv1a(const void *p, size_t n, uint64_t h) {
    // (eight bytes at a time: the tables are a few hundred KB)
    const uint64_t *q = static_cast<const uint64_t *>(p);
    for (size_t i = 0; i < n / 8; ++i) h = (h ^ q[i]) * 1099511628211ull;
    const unsigned char *t = static_cast<const unsigned char *>(p) + (n / 8) * 8;
    for (size_t i = 0; i < n % 8; ++i) h = (h ^ t[i]) * 1099511628211ull;
    return h;
}
}  // namespace

int smi_batch_add_sweep_plan(smi_batch *b, int32_t h, int32_t w, const double *weights,
                             const int32_t *offsets, const int32_t *dist_idx, int32_t n_idx) {
    SMI_REQUIRE(b && weights && offsets, "null argument");
    SMI_REQUIRE(h > 0 && w > 0, "empty box");
    SMI_HIP(hipSetDevice(b->device));
    static const bool use_cache = [] {  // development aid: SMI_PLAN_CACHE=0
        const char *e = getenv("SMI_PLAN_CACHE");
        return !e || atoi(e) != 0;
    }();
    PlanKey key{b->device, h, w, n_idx, 1469598103934665603ull};
    if (use_cache) {
        key.hash = fnv1a(weights, (size_t)8 * h * w * sizeof(double), key.hash);
        key.hash = fnv1a(offsets, 8 * sizeof(int32_t), key.hash);
        if (n_idx > 0) key.hash = fnv1a(dist_idx, (size_t)n_idx * sizeof(int32_t), key.hash);
        std::lock_guard<std::mutex> lock(g_sweep_plans_mu);
        auto hit = g_sweep_plans.find(key);
        if (hit != g_sweep_plans.end()) {
            b->plans.push_back(hit->second);
            b->plan_shared.push_back(1);
            if (hit->second.n_levels > b->max_levels) b->max_levels = hit->second.n_levels;
            if (int rc = upload_plans(b)) return rc;
            return (int)b->plans.size() - 1;
        }
    }
    SweepPlanHost hp;
    if (!build_sweep_plan(h * w, weights, offsets, 8, dist_idx, n_idx, &hp)) return SMI_ERR_INVALID;
    SweepPlanDev dp;
    dp.h = h;
    dp.w = w;
    dp.n_entries = hp.n_entries;
    dp.max_terms = hp.max_terms;
    dp.n_levels = (int32_t)hp.level_start.size() - 1;
    std::vector<float> wt(hp.wt.size());
    for (size_t i = 0; i < wt.size(); ++i) wt[i] = (float)hp.wt[i];
    int rc;
    if ((rc = upload(&dp.level_start, hp.level_start.data(), hp.level_start.size()))) return rc;
    if ((rc = upload(&dp.pix, hp.pix.data(), hp.pix.size()))) return rc;
    if ((rc = upload(&dp.cnt, hp.cnt.data(), hp.cnt.size()))) return rc;
    if ((rc = upload(&dp.nbr, hp.nbr.data(), hp.nbr.size()))) return rc;
    if ((rc = upload(&dp.wt, wt.data(), wt.size()))) return rc;
    if (hp.max_terms <= 4 && h * w <= 16376) {
        // slot layout of the fast update kernel: every level padded to 64-lane steps
        const SweepSlotEntry idle{};  // zeros: the spare cell at address 0, weights 0
        std::vector<SweepSlotEntry> slots;
        std::vector<uint32_t> used;  // lanes in use per step
        for (int l = 0; l < dp.n_levels; ++l) {
            const int s0 = hp.level_start[l], s1 = hp.level_start[l + 1];
            for (int base = s0; base < s1; base += 64) {
                used.push_back((uint32_t)std::min(64, s1 - base));
                for (int lane = 0; lane < 64; ++lane) {
                    SweepSlotEntry e = idle;
                    const int q = base + lane;
                    if (q < s1) {
                        const uint32_t p = 16 + (uint32_t)hp.pix[q] * 4;
                        const int n = hp.cnt[q];
                        uint32_t nb4[4] = {p, p, p, p};
                        for (int j = 0; j < n; ++j) {
                            nb4[j] = 16 + (uint32_t)hp.nbr[(size_t)j * hp.n_entries + q] * 4;
                            e.w[j] = (float)hp.wt[(size_t)j * hp.n_entries + q];
                        }
                        e.p_n0 = p | (nb4[0] << 16);
                        e.n1_n2 = nb4[1] | (nb4[2] << 16);
                        e.n3 = nb4[3];
                    }
                    slots.push_back(e);
                }
            }
        }
        // the sweep processes four steps per loop iteration and requests plan entries
        // up to three steps beyond the current iteration: pad to a multiple of four
        // steps and append three idle steps that are read but never executed
        while ((slots.size() / 64) % 4)
            for (int lane = 0; lane < 64; ++lane) slots.push_back(idle);
        dp.n_slots = (int32_t)(slots.size() / 64);
        for (int extra = 0; extra < 3 * 64; ++extra) slots.push_back(idle);
        used.resize(slots.size() / 64 + 3, 0);
        for (size_t step = 0; step < slots.size() / 64; ++step)
            for (int lane = 0; lane < 64; ++lane) slots[step * 64 + lane].lanes_ahead = used[step + 3];
        if ((rc = upload(&dp.slots, slots.data(), slots.size()))) return rc;
    }
    // ring schedule of the radial tables (common.h); SMI_RING_SWEEP=0 keeps the slot plan
    // (development aid: A/B runs)
    static const bool use_ring = [] {
        const char *e = getenv("SMI_RING_SWEEP");
        return !e || atoi(e) != 0;
    }();
    RingPlanHost rp;
    std::vector<uint8_t> stream;
    if (use_ring && dp.slots && build_ring_plan(h, w, weights, offsets, 8, dist_idx, n_idx, &rp) &&
        ring_device_stream(rp, &stream)) {
        uint8_t *d_ring = nullptr;
        if ((rc = upload(&d_ring, stream.data(), stream.size()))) return rc;
        dp.ring = d_ring;
        dp.ring_planes = rp.planes;
        dp.ring_pad = rp.n_pad;
        dp.ring_nat = rp.n_nat;
        dp.ring_rmax = rp.rmax;
        dp.ring_centre = rp.centre;
        dp.ring_perm = rp.perm;
        dp.ring_bytes = (uint32_t)stream.size();
    }
    bool shared = false;
    if (use_cache) {
        std::lock_guard<std::mutex> lock(g_sweep_plans_mu);
        if (g_sweep_plans.size() < kPlanCacheLimit)
            shared = g_sweep_plans.emplace(key, dp).second;  // (the cache owns the device copies now)
    }
    b->plans.push_back(dp);
    b->plan_shared.push_back(shared ? 1 : 0);
    if (dp.n_levels > b->max_levels) b->max_levels = dp.n_levels;
    if ((rc = upload_plans(b))) return rc;
    return (int)b->plans.size() - 1;
}

int smi_sweep_ring_plan(int32_t h, int32_t w, const double *weights, const int32_t *offsets,
                        const int32_t *dist_idx, int32_t n_idx, int32_t info[8], float *wts,
                        uint16_t *addr, int64_t capacity) {
    SMI_REQUIRE(weights && offsets && dist_idx && info, "null argument");
    SMI_REQUIRE(h > 0 && w > 0, "empty box");
    RingPlanHost rp;
    if (!build_ring_plan(h, w, weights, offsets, 8, dist_idx, n_idx, &rp)) return 0;
    std::vector<uint8_t> stream;
    // (n_nat rides in the upper half of the first word: the caller's array has eight entries)
    const int32_t vals[8] = {rp.planes | (rp.n_nat << 8), rp.n_steps, rp.n_pad, rp.rmax, rp.centre,
                             (int32_t)rp.perm, (int32_t)rp.addr.size(),
                             ring_device_stream(rp, &stream) ? (int32_t)stream.size() : 0};
    memcpy(info, vals, sizeof(vals));
    if (wts && addr && capacity >= (int64_t)rp.addr.size()) {
        memcpy(wts, rp.wts.data(), rp.wts.size() * sizeof(float));
        memcpy(addr, rp.addr.data(), rp.addr.size() * sizeof(uint16_t));
    }
    return 1;
}

// the fused convolution kernel's copy of an observation (common.h: BatchView::dw).  Adopted
// device buffers may have been written by another stream: the device is idle first (once per
// adoption); arrays this library uploaded itself arrived by a blocking copy, and the batch
// stream has been synchronised by the caller -- no other stream or batch is stalled (a fit
// with noise_factor > 0 registers an observation every iteration).
static int interleave_observation(smi_batch *b, const float *d_data, const float *d_weights,
                                  float4 **dw, bool adopted = false) {
    if (!b->fused) return SMI_OK;
    const int H = b->d.H, W = b->d.W;
    const int64_t planes = (int64_t)b->d.n_blends * b->d.C;
    // (sixteen elements of slack, zero: the kernel's loads of the columns W .. W + 15 of the last
    // pair, which it multiplies by a weight of 0 -- they have to be finite)
    const size_t n = (size_t)planes * ((H + 1) / 2) * W;
    if (!*dw) {
        SMI_HIP(dev_alloc(dw, n + 16));
        SMI_HIP(hipMemset(*dw + n, 0, 16 * sizeof(float4)));
    }
    if (adopted) SMI_HIP(hipDeviceSynchronize());
    launch_interleave_obs(d_data, d_weights, *dw, planes, H, W, b->stream);
    return SMI_OK;
}

int smi_batch_set_observation(smi_batch *b, const float *data, const float *weights) {
    SMI_REQUIRE(b && data && weights, "null argument");
    // (log_norm is recomputed from this observation alone: the terms of observations added
    // with smi_batch_add_observation and of smi_batch_add_loss_constant would be lost)
    SMI_REQUIRE(b->layers.empty(), "the first observation cannot be replaced once further observations were added");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));  // pending steps may still read the old arrays
    const size_t n = (size_t)b->d.n_blends * b->d.C * b->d.H * b->d.W;
    if (!b->own_obs) b->data = b->weights = nullptr;
    int rc;
    if ((rc = upload(&b->data, data, n))) return rc;
    if ((rc = upload(&b->weights, weights, n))) return rc;
    g_observation_uploads.fetch_add(1);
    b->own_obs = true;
    if ((rc = interleave_observation(b, b->data, b->weights, &b->dw))) return rc;
    launch_log_norm(b->weights, b->log_norm, b->d.n_blends, (int64_t)b->d.C * b->d.H * b->d.W,
                    b->stream);
    if (!b->include_log_norm)
        SMI_HIP(hipMemsetAsync(b->log_norm, 0, b->d.n_blends * sizeof(double), b->stream));
    b->have_obs = true;
    refresh_view(b);
    return SMI_OK;
}

int smi_batch_set_previous_loss(smi_batch *b, const double *loss) {
    SMI_REQUIRE(b && loss, "null argument");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int nb = b->d.n_blends;
    SMI_HIP(hipMemcpy(b->last_loss, loss, nb * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int32_t> ones(nb, 1);
    SMI_HIP(hipMemcpy(b->have_prev, ones.data(), nb * sizeof(int32_t), hipMemcpyHostToDevice));
    return SMI_OK;
}

int smi_batch_add_loss_constant(smi_batch *b, const double *constant) {
    SMI_REQUIRE(b && constant, "null argument");
    SMI_REQUIRE(b->have_obs, "smi_batch_add_loss_constant follows smi_batch_set_observation");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int nb = b->d.n_blends;
    std::vector<double> ln(nb);
    SMI_HIP(hipMemcpy(ln.data(), b->log_norm, nb * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < nb; ++i) ln[i] += constant[i];
    SMI_HIP(hipMemcpy(b->log_norm, ln.data(), nb * sizeof(double), hipMemcpyHostToDevice));
    return SMI_OK;
}

int smi_batch_set_log_norm(smi_batch *b, int32_t include) {
    SMI_REQUIRE(b, "null batch");
    SMI_REQUIRE(!b->have_obs, "smi_batch_set_log_norm must precede smi_batch_set_observation");
    b->include_log_norm = include != 0;
    return SMI_OK;
}

int smi_batch_set_scheme(smi_batch *b, int32_t scheme) {
    SMI_REQUIRE(b, "null batch");
    SMI_REQUIRE(scheme == SMI_SCHEME_AMSGRAD || scheme == SMI_SCHEME_FISTA, "unknown scheme");
    SMI_REQUIRE(!b->have_components, "smi_batch_set_scheme must precede smi_batch_set_components");
    b->scheme = scheme;
    refresh_view(b);
    return SMI_OK;
}

int smi_batch_get_fista_state(smi_batch *b, float *z_sed, float *z_morph, double *t) {
    SMI_REQUIRE(b && b->have_components && b->scheme == SMI_SCHEME_FISTA, "not a FISTA batch");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const size_t n = b->d.n_components;
    if (z_sed && n) SMI_HIP(hipMemcpy(z_sed, b->mom[0], n * b->d.C * sizeof(float), hipMemcpyDeviceToHost));
    if (z_morph && b->n_morph)
        SMI_HIP(hipMemcpy(z_morph, b->mom[3], (size_t)b->n_morph * sizeof(float), hipMemcpyDeviceToHost));
    if (t && n) SMI_HIP(hipMemcpy(t, b->fista_t, n * 2 * sizeof(double), hipMemcpyDeviceToHost));
    return SMI_OK;
}

int smi_batch_set_fista_state(smi_batch *b, const float *z_sed, const float *z_morph,
                              const double *t) {
    SMI_REQUIRE(b && b->have_components && b->scheme == SMI_SCHEME_FISTA, "not a FISTA batch");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const size_t n = b->d.n_components;
    if (z_sed && n) SMI_HIP(hipMemcpy(b->mom[0], z_sed, n * b->d.C * sizeof(float), hipMemcpyHostToDevice));
    if (z_morph && b->n_morph)
        SMI_HIP(hipMemcpy(b->mom[3], z_morph, (size_t)b->n_morph * sizeof(float), hipMemcpyHostToDevice));
    if (t && n) SMI_HIP(hipMemcpy(b->fista_t, t, n * 2 * sizeof(double), hipMemcpyHostToDevice));
    return SMI_OK;
}

int smi_batch_set_observation_device(smi_batch *b, const float *d_data, const float *d_weights) {
    SMI_REQUIRE(b && d_data && d_weights, "null argument");
    // (log_norm is recomputed from this observation alone: the terms of observations added
    // with smi_batch_add_observation and of smi_batch_add_loss_constant would be lost)
    SMI_REQUIRE(b->layers.empty(), "the first observation cannot be replaced once further observations were added");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));  // pending steps may still read the old arrays
    if (b->own_obs) {
        (void)hipFree(b->data);
        (void)hipFree(b->weights);
        b->own_obs = false;
    }
    b->data = const_cast<float *>(d_data);
    b->weights = const_cast<float *>(d_weights);
    if (int rc = interleave_observation(b, b->data, b->weights, &b->dw, true)) return rc;
    launch_log_norm(b->weights, b->log_norm, b->d.n_blends, (int64_t)b->d.C * b->d.H * b->d.W,
                    b->stream);
    if (!b->include_log_norm)
        SMI_HIP(hipMemsetAsync(b->log_norm, 0, b->d.n_blends * sizeof(double), b->stream));
    b->have_obs = true;
    refresh_view(b);
    return SMI_OK;
}

// ---- ConvolutionRenderer(psf_shift=...): free shift of the difference kernel ----------
static void release_kernel_shift(smi_batch *b) {
    for (void *p : {(void *)b->ks.stamp, (void *)b->ks.shifted, (void *)b->ks.partial,
                    (void *)b->ks.state, (void *)b->ks_resid, (void *)b->ks_tmp})
        if (p) (void)hipFree(p);
    b->ks = smi::KernelShiftView{};
    b->ks_resid = nullptr;
    b->ks_tmp = nullptr;
}

// stamps at the current shift -> spectrum the fused kernel reads
static int refresh_shifted_kernel(smi_batch *b, int respect_state) {
    int rc;
    if ((rc = launch_psf_shift_forward(b->view, b->ks, respect_state, b->stream))) return rc;
    const double sc = 0.5 / ((double)b->Fy * (double)b->Fx);
    return launch_stamp_spectrum(b->ks.shifted, b->ks_tmp, b->Kt, b->ks.n_sets * b->ks.bands,
                                 b->ks.ph, b->ks.pw, b->Fy, b->Fx, sc, b->stream);
}

// gradient of -logL w.r.t. the kernel shift at the model cube in b->P (and, unless
// grad_only, the AMSGrad step of the shift; the spectrum is refreshed after the iteration's
// convolution, which still belongs to the old shift)
static int kernel_shift_backward(smi_batch *b, const BatchView &v, int it, int grad_only) {
    int rc;
    if ((rc = launch_fused_conv(v, b->Fy, b->Fx, b->P, b->Kt, b->d.kernel_bands,
                                b->d.kernel_per_blend, b->ks_resid, 1, nullptr, b->stream)))
        return rc;
    return launch_psf_shift_backward(v, b->ks, b->ks_resid, b->P, it, grad_only, b->stream);
}

int smi_batch_set_kernel_shift_relative_step(smi_batch *b, double factor) {
    SMI_REQUIRE(b != nullptr, "null batch");
    SMI_REQUIRE(b->ks.stamp != nullptr, "smi_batch_set_kernel_shift has not been called");
    SMI_REQUIRE(factor >= 0, "negative factor");
    b->ks.rel = factor;
    return SMI_OK;
}

int smi_batch_set_kernel_shift(smi_batch *b, const float *kernel, int32_t h0, int32_t w0,
                               const int32_t *fft_shape, const double *shift,
                               const double *moments, double step) {
    SMI_REQUIRE(b && kernel && fft_shape && shift, "null argument");
    SMI_REQUIRE(b->fused, "a free kernel shift needs the fused convolution path");
    SMI_REQUIRE(b->have_components && b->have_obs, "set the observation and the components first");
    // (further observations on the model's grid keep their fixed kernels: the free shift belongs
    // to the first observation's)
    SMI_REQUIRE(b->lowres.empty(), "a free kernel shift next to a low-resolution observation");
    SMI_REQUIRE(b->d.kernel_per_blend || b->d.n_blends == 1,
                "a free kernel shift belongs to one blend: use per-blend kernels");
    const int ph = b->d.kernel_h, pw = b->d.kernel_w;
    SMI_REQUIRE(h0 > 0 && w0 > 0 && h0 <= ph && w0 <= pw, "stamp larger than the batch's kernel stamp");
    SMI_REQUIRE(fft_shape[0] >= h0 && fft_shape[1] >= w0 && fft_shape[0] <= 512 && fft_shape[1] <= 512,
                "bad FFT shape of the shift");
    SMI_REQUIRE(step >= 0, "negative step");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    release_kernel_shift(b);
    smi::KernelShiftView &ks = b->ks;
    ks.n_sets = b->d.kernel_per_blend ? b->d.n_blends : 1;
    ks.bands = b->d.kernel_bands;
    ks.per_blend = b->d.kernel_per_blend;
    ks.h0 = h0;
    ks.w0 = w0;
    ks.ph = ph;
    ks.pw = pw;
    ks.oy = ph / 2 - h0 / 2;
    ks.ox = pw / 2 - w0 / 2;
    ks.Fy = fft_shape[0];
    ks.Fx = fft_shape[1];
    ks.slab = 8;
    ks.n_part = (ks.bands == 1 ? b->d.C : 1) * ((b->d.H + ks.slab - 1) / ks.slab);
    ks.step = step;
    ks.rel = 0.0;
    const size_t n_img = (size_t)ks.n_sets * ks.bands, n0 = (size_t)h0 * w0;
    int rc;
    float *stamp = nullptr;
    if ((rc = upload(&stamp, kernel, n_img * n0))) return rc;
    ks.stamp = stamp;
    SMI_HIP(dev_alloc(&ks.shifted, n_img * ph * pw));
    SMI_HIP(hipMemset(ks.shifted, 0, n_img * ph * pw * sizeof(float)));
    SMI_HIP(dev_alloc(&ks.partial, n_img * ks.n_part * n0));
    std::vector<double> st((size_t)ks.n_sets * 10, 0.0);
    for (int s = 0; s < ks.n_sets; ++s) {
        st[(size_t)s * 10] = shift[2 * s];
        st[(size_t)s * 10 + 1] = shift[2 * s + 1];
        for (int i = 0; moments && i < 6; ++i) st[(size_t)s * 10 + 2 + i] = moments[6 * s + i];
    }
    if ((rc = upload(&ks.state, st.data(), st.size()))) return rc;
    SMI_HIP(dev_alloc(&b->ks_resid, (size_t)b->d.n_blends * b->d.C * b->d.H * b->d.W));
    SMI_HIP(dev_alloc(&b->ks_tmp, n_img * ph * b->Fxh));
    if (!b->Kt) SMI_HIP(dev_alloc(&b->Kt, n_img * b->Fy * b->Fxh));
    if ((rc = refresh_shifted_kernel(b, 0))) return rc;
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipGetLastError());
    b->have_kernel = true;
    return SMI_OK;
}

int smi_batch_get_kernel_shift(smi_batch *b, double *shift, double *moments, double *gradient,
                               float *kernel) {
    SMI_REQUIRE(b && b->ks.stamp, "the batch has no free kernel shift");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int n = b->ks.n_sets;
    std::vector<double> st((size_t)n * 10);
    SMI_HIP(hipMemcpy(st.data(), b->ks.state, st.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int s = 0; s < n; ++s) {
        for (int a = 0; a < 2; ++a) {
            if (shift) shift[2 * s + a] = st[(size_t)s * 10 + a];
            if (gradient) gradient[2 * s + a] = st[(size_t)s * 10 + 8 + a];
        }
        for (int i = 0; moments && i < 6; ++i) moments[6 * s + i] = st[(size_t)s * 10 + 2 + i];
    }
    if (kernel)
        SMI_HIP(hipMemcpy(kernel, b->ks.shifted,
                          (size_t)n * b->ks.bands * b->ks.ph * b->ks.pw * sizeof(float),
                          hipMemcpyDeviceToHost));
    return SMI_OK;
}

int smi_batch_set_kernel(smi_batch *b, const float *kernel) {
    SMI_REQUIRE(b && kernel, "null argument");
    SMI_REQUIRE(!b->null_renderer, "batch was created without a kernel (NullRenderer)");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));  // pending steps may still read the old arrays
    release_kernel_shift(b);  // a fixed kernel from now on
    const int n_img = (b->d.kernel_per_blend ? b->d.n_blends : 1) * b->d.kernel_bands;
    const int ph = b->d.kernel_h, pw = b->d.kernel_w;
    SMI_REQUIRE(ph <= b->Fy && pw <= b->Fx, "kernel larger than the FFT shape");
    const size_t n_real = (size_t)n_img * b->Fy * b->Fx, n_cplx = (size_t)n_img * b->Fy * b->Fxh;
    float *d_kern = nullptr, *d_pad = nullptr;
    int rc;
    if ((rc = upload(&d_kern, kernel, (size_t)n_img * ph * pw))) return rc;
    if (b->fused) {
        // direct DFT of the stamp (no rocFFT plan): 0.5 = the 1/2 of the fused kernel's
        // Hermitian row separation, 1/(Fy Fx) = normalisation of its inverse transforms
        double2 *d_tmp = nullptr;
        SMI_HIP(dev_alloc(&d_tmp, (size_t)n_img * ph * b->Fxh));
        if (!b->Kt) SMI_HIP(dev_alloc(&b->Kt, n_cplx));
        const double sc = 0.5 / ((double)b->Fy * (double)b->Fx);
        if ((rc = launch_stamp_spectrum(d_kern, d_tmp, b->Kt, n_img, ph, pw, b->Fy, b->Fx, sc,
                                        b->stream)))
            return rc;
        SMI_HIP(hipStreamSynchronize(b->stream));
        SMI_HIP(hipGetLastError());
        (void)hipFree(d_tmp);
        (void)hipFree(d_kern);
        b->have_kernel = true;
        return SMI_OK;
    }
    SMI_HIP(dev_alloc(&d_pad, n_real));
    SMI_HIP(hipMemsetAsync(d_pad, 0, n_real * sizeof(float), b->stream));
    const float scale = 1.0f / ((float)b->Fy * (float)b->Fx);
    launch_wrap_kernel(d_kern, d_pad, n_img, ph, pw, b->Fy, b->Fx, scale, b->stream);
    if (!b->Khat) SMI_HIP(dev_alloc(&b->Khat, n_cplx));
    PlanPair kp;
    if ((rc = plans_get(b->device, b->Fy, b->Fx, n_img, b->plans_shared, &kp))) return rc;
    rocfft_plan plan = kp.fwd;
    size_t wb = 0;
    SMI_FFT(rocfft_plan_get_work_buffer_size(plan, &wb));
    rocfft_execution_info info = nullptr;
    SMI_FFT(rocfft_execution_info_create(&info));
    void *work = nullptr;
    if (wb) {
        SMI_HIP(hipMalloc(&work, wb));
        SMI_FFT(rocfft_execution_info_set_work_buffer(info, work, wb));
    }
    SMI_FFT(rocfft_execution_info_set_stream(info, b->stream));
    void *ins[1] = {d_pad}, *outs[1] = {b->Khat};
    SMI_FFT(rocfft_execute(plan, ins, outs, info));
    SMI_HIP(hipStreamSynchronize(b->stream));
    rocfft_execution_info_destroy(info);
    if (!b->plans_shared) destroy_plans(&kp);
    if (work) (void)hipFree(work);
    (void)hipFree(d_pad);
    (void)hipFree(d_kern);
    if (b->fused) {
        // the fused kernel wants [img][pos_y(ky)][kx]; the 1/2 of its Hermitian row
        // separation is folded into the spectrum as well
        if (!b->Kt) SMI_HIP(dev_alloc(&b->Kt, n_cplx));
        if ((rc = launch_permute_kernel_spectrum(b->Khat, b->Kt, n_img, b->Fy, b->Fx, 0.5f, b->stream)))
            return rc;
        SMI_HIP(hipStreamSynchronize(b->stream));
    }
    b->have_kernel = true;
    return SMI_OK;
}

// Q += Q2 (gradient images of further observations)
__global__ void add_images_kernel(float *Q, const float *Q2, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Q[i] += Q2[i];
}

// the further same-grid observations: loss partials (and with `backward` their gradient
// images, added to Q) for the model cube in b->P
static int layers_evaluate(smi_batch *b, const BatchView &v, int backward, hipStream_t s) {
    const int C = b->d.C;
    for (size_t l = 0; l < b->layers.size(); ++l) {
        BatchView vl = v;
        vl.data = b->layers[l].data;
        vl.weights = b->layers[l].weights;
        vl.dw = b->layers[l].dw;
        vl.loss_partial = v.loss_partial + (l + 1) * C;
        const int rc = launch_fused_conv(vl, b->Fy, b->Fx, b->P, b->layers[l].Kt, b->d.kernel_bands,
                                         b->d.kernel_per_blend, b->Q2, backward ? 0 : 1, nullptr, s);
        if (rc) return rc;
        if (backward) {
            const int64_t n = (int64_t)b->d.n_blends * C * b->d.H * b->d.W;
            hipLaunchKernelGGL(add_images_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                               b->Q, b->Q2, n);
        }
    }
    return SMI_OK;
}

int smi_batch_add_observation(smi_batch *b, const float *data, const float *weights,
                              const float *kernel) {
    SMI_REQUIRE(b && data && weights && kernel, "null argument");
    SMI_REQUIRE(b->have_obs && b->have_kernel, "add the first observation and its kernel first");
    SMI_REQUIRE(b->fused, "further observations need the fused convolution path");
    // (a free kernel shift stays with the first observation: smi_batch_set_kernel_shift)
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int nb = b->d.n_blends, C = b->d.C;
    const size_t n = (size_t)nb * C * b->d.H * b->d.W;
    smi_batch::ObsLayer layer;
    // whatever fails below, the layer's buffers go with it
    struct Guard {
        smi_batch::ObsLayer *l;
        ~Guard() {
            if (!l) return;
            for (void *p : {(void *)l->data, (void *)l->weights, (void *)l->Kt, (void *)l->dw})
                if (p) (void)hipFree(p);
        }
    } guard{&layer};
    int rc;
    if ((rc = upload(&layer.data, data, n))) return rc;
    if ((rc = upload(&layer.weights, weights, n))) return rc;
    if ((rc = interleave_observation(b, layer.data, layer.weights, &layer.dw))) return rc;
    g_observation_uploads.fetch_add(1);
    // kernel spectrum: the routine of smi_batch_set_kernel, into a buffer of its own
    // (smi_batch_set_kernel makes the batch's kernel a fixed one: the first observation's free
    // shift steps aside for the call)
    float2 *first = b->Kt;
    b->Kt = nullptr;
    const smi::KernelShiftView ks = b->ks;
    float *ks_resid = b->ks_resid;
    auto *ks_tmp = b->ks_tmp;
    b->ks = smi::KernelShiftView{};
    b->ks_resid = nullptr;
    b->ks_tmp = nullptr;
    rc = smi_batch_set_kernel(b, kernel);
    layer.Kt = b->Kt;
    b->Kt = first;
    b->ks = ks;
    b->ks_resid = ks_resid;
    b->ks_tmp = ks_tmp;
    if (rc) return rc;
    // log_norm of the layer joins the blend's (observation.py:172-186)
    if (b->include_log_norm) {
        struct Scratch {
            double *p = nullptr;
            ~Scratch() {
                if (p) (void)hipFree(p);
            }
        } d_ln;
        SMI_HIP(dev_alloc(&d_ln.p, (size_t)nb));
        launch_log_norm(layer.weights, d_ln.p, nb, (int64_t)C * b->d.H * b->d.W, b->stream);
        std::vector<double> ln(nb);
        SMI_HIP(hipMemcpyAsync(ln.data(), d_ln.p, nb * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        SMI_HIP(hipStreamSynchronize(b->stream));
        if ((rc = smi_batch_add_loss_constant(b, ln.data()))) return rc;
    }
    guard.l = nullptr;
    b->layers.push_back(layer);
    if (!b->Q2) SMI_HIP(dev_alloc(&b->Q2, n));
    if (b->loss_partial) SMI_HIP(hipFree(b->loss_partial));
    b->loss_partial = nullptr;
    SMI_HIP(dev_alloc(&b->loss_partial, (size_t)nb * C * (1 + b->layers.size())));
    const int keep = b->view.max_box_pixels;
    refresh_view(b);
    b->view.max_box_pixels = keep;
    return SMI_OK;
}

// smi_batch_set_components, and with `keep` smi_batch_update_components: the table of
// components replaced while the components with keep[k] != 0 hold on to their parameters and
// moments on the device (same number of components per blend; the others take theirs from
// the state records in `states`, in the order of k)
static int set_components_impl(smi_batch *b, const smi_components *c, const int32_t *keep,
                               const float *states) {
    SMI_REQUIRE(b && c, "null argument");
    SMI_REQUIRE(c->blend && c->origin_y && c->origin_x && c->box_h && c->box_w &&
                    (keep || (c->sed && c->morph)) && c->sed_min_step && c->morph_step && c->prox_flags,
                "missing component array");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));  // pending steps may still read the old arrays
    const int n = b->d.n_components, nb = b->d.n_blends, C = b->d.C;
    // what the kept components carry over: the four pixel arrays get a new packing (the
    // spectra and their moments stay where they are: [n][C] either way)
    std::vector<int64_t> old_moff;
    float *old_px[4] = {nullptr, nullptr, nullptr, nullptr};
    if (keep) {
        SMI_REQUIRE(b->have_components && states, "smi_batch_update_components follows smi_batch_set_components");
        SMI_REQUIRE(b->n_point == 0 && b->n_shift == 0 && b->scheme != SMI_SCHEME_FISTA && !b->ks.stamp,
                    "smi_batch_update_components: factorized image components under AMSGrad only");
        for (int k = 0; k < n; ++k) {
            SMI_REQUIRE(!(c->prox_flags[k] & (SMI_COMPONENT_POINT_SOURCE | SMI_COMPONENT_SHIFTING)),
                        "smi_batch_update_components: factorized image components only");
            SMI_REQUIRE(c->blend[k] == b->h_blend[k], "smi_batch_update_components: components moved between blends");
            if (keep[k] == 1)
                SMI_REQUIRE((int64_t)c->box_h[k] * c->box_w[k] == b->h_moff[k + 1] - b->h_moff[k],
                            "smi_batch_update_components: a kept component changed its box size");
            if (keep[k] >= 2) {  // resized on the device about its centre
                const int64_t n_old = b->h_moff[k + 1] - b->h_moff[k];
                const int ow = (int)std::lround(std::sqrt((double)n_old));
                SMI_REQUIRE(keep[k] <= 3 && (int64_t)ow * ow == n_old && c->box_h[k] == c->box_w[k] &&
                                c->box_h[k] <= 1024 && (c->box_h[k] - ow) % 2 == 0,
                            "smi_batch_update_components: device resize needs square boxes of at most "
                            "1024 pixels a side whose sides differ by an even number");
            }
        }
    }
    struct FreeOld {
        float **p;
        ~FreeOld() {
            for (int i = 0; i < 4; ++i)
                if (p[i]) (void)hipFree(p[i]);
        }
    } free_old{old_px};
    std::vector<int32_t> start(nb + 1, 0);
    std::vector<int64_t> moff(n + 1, 0);
    int max_pix = 1, prev = 0;
    for (int k = 0; k < n; ++k) {
        SMI_REQUIRE(c->blend[k] >= prev && c->blend[k] < nb, "components must be grouped by blend");
        SMI_REQUIRE(c->box_h[k] > 0 && c->box_w[k] > 0, "empty component box");
        prev = c->blend[k];
        start[c->blend[k] + 1]++;
        const int64_t np = (int64_t)c->box_h[k] * c->box_w[k];
        moff[k + 1] = moff[k] + np;
        if (np > max_pix) max_pix = (int)np;
        if (c->prox_flags[k] & SMI_COMPONENT_SHIFTING) {
            SMI_REQUIRE(c->center, "shifting component without its shift (center)");
            SMI_REQUIRE(!(c->prox_flags[k] & SMI_COMPONENT_POINT_SOURCE),
                        "a point source cannot carry a Fourier shift");
            // (the FFT lengths of fft.shift, 2 n + 10 rounded up, must fit the 512-entry tables)
            SMI_REQUIRE(c->box_h[k] <= 240 && c->box_w[k] <= 240,
                        "shifting component box larger than 240 pixels");
        }
        if (c->prox_flags[k] & SMI_COMPONENT_POINT_SOURCE) {
            SMI_REQUIRE(c->center && c->psf_sigma, "point source without center / psf_sigma");
            SMI_REQUIRE(c->box_h[k] < 64 && c->box_w[k] < 64, "point-source box larger than 63 pixels");
            SMI_REQUIRE(c->psf_sigma[k] > 0.f, "point source with psf_sigma <= 0");
            SMI_REQUIRE(!(c->prox_flags[k] & SMI_PROX_MONOTONIC), "point source with a morphology constraint");
        }
        if (c->prox_flags[k] & SMI_PROX_MONOTONIC) {
            const int pid = c->sweep_plan ? c->sweep_plan[k] : -1;
            const int last = pid + ((c->prox_flags[k] & SMI_PROX_FIT_CENTER) ? 8 : 0);
            SMI_REQUIRE(pid >= 0 && last < (int)b->plans.size(), "monotonic component without sweep plan");
            for (int q = pid; q <= last; ++q)
                SMI_REQUIRE(b->plans[q].h == c->box_h[k] && b->plans[q].w == c->box_w[k],
                            "sweep plan shape does not match the component box");
        }
        if (c->prox_flags[k] & SMI_PROX_BG_THRESH)
            SMI_REQUIRE(c->bg_level, "SMI_PROX_BG_THRESH without bg_level");
        SMI_REQUIRE(C <= 64, "more than 64 bands");
    }
    for (int i = 0; i < nb; ++i) start[i + 1] += start[i];
    b->h_comp_start = start;
    b->lite_flags = false;
    for (int k = 0; k < n; ++k)
        if (c->prox_flags[k] & (SMI_PROX_FIT_CENTER | SMI_PROX_BG_THRESH)) b->lite_flags = true;
    b->n_morph = moff[n];
    SMI_REQUIRE(b->n_morph < ((int64_t)1 << 31), "more than 2^31 morphology pixels in one batch");
    b->view.max_box_pixels = max_pix;

    // (every argument check is behind us: only now do the kept pixel arrays leave the batch, so a
    // refused table leaves the old one and its state in place)
    if (keep) {
        old_moff = b->h_moff;
        old_px[0] = b->morph;
        b->morph = nullptr;
        for (int i = 0; i < 3; ++i) {
            old_px[1 + i] = b->mom[3 + i];
            b->mom[3 + i] = nullptr;
        }
    }
    std::vector<float> zeros_n(n, 0.f), rel(n, 1e-2f);
    std::vector<int32_t> noplan(n, -1);
    int rc;
#define UP(field, src, count) \
    if ((rc = upload(&b->field, src, (size_t)(count)))) return rc
    UP(comp_start, start.data(), nb + 1);
    UP(c_blend, c->blend, n);
    UP(c_oy, c->origin_y, n);
    UP(c_ox, c->origin_x, n);
    UP(c_h, c->box_h, n);
    UP(c_w, c->box_w, n);
    UP(c_flags, c->prox_flags, n);
    UP(c_plan, c->sweep_plan ? c->sweep_plan : noplan.data(), n);
    UP(c_moff, moff.data(), n + 1);
    UP(c_sed_min_step, c->sed_min_step, (size_t)n * C);
    UP(c_sed_rel, c->sed_rel_step ? c->sed_rel_step : rel.data(), n);
    UP(c_morph_step, c->morph_step, n);
    UP(c_morph_rel, c->morph_rel_step ? c->morph_rel_step : zeros_n.data(), n);
    UP(c_min_grad, c->min_gradient ? c->min_gradient : zeros_n.data(), n);
    UP(c_lthresh, c->l_thresh ? c->l_thresh : zeros_n.data(), n);
    if (keep) {
        SMI_HIP(dev_alloc(&b->morph, (size_t)b->n_morph));
    } else {
        UP(sed, c->sed, (size_t)n * C);
        UP(morph, c->morph, (size_t)b->n_morph);
    }
    std::vector<float> cfloor(n, 1e-6f);
    UP(c_center_floor, c->center_floor ? c->center_floor : cfloor.data(), n);
    std::vector<float> full_strength(n, 1.f);
    UP(c_sym_strength, c->sym_strength ? c->sym_strength : full_strength.data(), n);
    UP(c_pos_floor, c->pos_floor ? c->pos_floor : zeros_n.data(), n);
    b->mono_mask = 0;
    for (int k = 0; k < n; ++k)
        if (c->prox_flags[k] & SMI_PROX_MONO_MASK) {
            SMI_REQUIRE(c->prox_flags[k] & SMI_PROX_MONOTONIC, "SMI_PROX_MONO_MASK needs SMI_PROX_MONOTONIC");
            b->mono_mask = 1;
        }
    bool repeats = false;
    for (int k = 0; k < n && c->chain_repeat; ++k) {
        SMI_REQUIRE(c->chain_repeat[k] >= 1, "chain_repeat must be >= 1");
        repeats = repeats || c->chain_repeat[k] > 1;
    }
    if (repeats) {
        UP(c_chain_repeat, c->chain_repeat, n);
    } else if (b->c_chain_repeat) {
        SMI_HIP(hipFree(b->c_chain_repeat));
        b->c_chain_repeat = nullptr;
    }
    if (c->bg_level) {
        UP(c_bg_level, c->bg_level, (size_t)n * C);
    } else if (b->c_bg_level) {
        SMI_HIP(hipFree(b->c_bg_level));
        b->c_bg_level = nullptr;
    }
    if (b->scheme == SMI_SCHEME_FISTA) {
        SMI_REQUIRE(c->fista_step, "SMI_SCHEME_FISTA needs fista_step");
        UP(c_fista_step, c->fista_step, n);
        std::vector<double> ones((size_t)n * 2, 1.0);
        UP(fista_t, ones.data(), (size_t)n * 2);
    }
#undef UP
    for (int i = keep ? 3 : 0; i < 6; ++i) {
        const size_t cnt = i < 3 ? (size_t)n * C : (size_t)b->n_morph;
        if (b->mom[i]) SMI_HIP(hipFree(b->mom[i]));
        SMI_HIP(dev_alloc(&b->mom[i], cnt));
        if (!keep) SMI_HIP(hipMemset(b->mom[i], 0, (cnt ? cnt : 1) * sizeof(float)));
    }
    if (b->scheme == SMI_SCHEME_FISTA) {
        // FistaParameter: z0 = x (lite/parameters.py:126-131)
        if (n) SMI_HIP(hipMemcpy(b->mom[0], b->sed, (size_t)n * C * sizeof(float), hipMemcpyDeviceToDevice));
        if (b->n_morph)
            SMI_HIP(hipMemcpy(b->mom[3], b->morph, (size_t)b->n_morph * sizeof(float),
                              hipMemcpyDeviceToDevice));
    }
    if (b->scratch) {
        SMI_HIP(hipFree(b->scratch));
        b->scratch = nullptr;
    }
    // x / psi / z of the generic update kernel fit the LDS up to ~100^2 pixels per box
    if ((4 + (b->mono_mask ? 1.25 : 0)) * ((max_pix + 3) & ~3) * sizeof(float) + 4096 > 160 * 1024)
        SMI_HIP(dev_alloc(&b->scratch, 3 * (size_t)b->n_morph));
    if (b->g_sed) SMI_HIP(hipFree(b->g_sed));
    if (b->g_morph) SMI_HIP(hipFree(b->g_morph));
    SMI_HIP(dev_alloc(&b->g_sed, (size_t)n * C));
    SMI_HIP(dev_alloc(&b->g_morph, (size_t)b->n_morph));
    if (b->xp_tmp) {
        SMI_HIP(hipFree(b->xp_tmp));
        b->xp_tmp = nullptr;
    }
    {
        // (only the four-wavefront teams keep x and psi there: boxes beyond the largest
        // one-wavefront class)
        bool teams = false;
        for (int k = 0; k < n && !teams; ++k)
            teams = update_class(c->box_h[k] * c->box_w[k]) >= kNumSmallClasses;
        if (teams) SMI_HIP(dev_alloc(&b->xp_tmp, 2 * (size_t)b->n_morph));
    }
    // point sources: offset of the centre from the mean of the box bounds
    // (morphology.py:503-507), moments zero
    std::vector<double> pt((size_t)n * 8, 0.0);
    std::vector<float> sigma(n, 0.f), beta(n, 0.f);
    b->box_center.assign((size_t)n * 2, 0.0);
    b->is_point.assign((size_t)n, 0);
    b->n_point = 0;
    b->n_shift = 0;
    b->max_box_side = 1;
    b->max_box_w = 1;
    b->is_shift.assign((size_t)n, 0);
    b->h_moff = moff;
    b->h_blend.assign(c->blend, c->blend + n);
    std::vector<float> shift_step(n, 1e-1f), shift_rel(n, 0.f);
    std::vector<int32_t> shift_fft((size_t)n * 2, 0);
    for (int k = 0; k < n; ++k) {
        b->max_box_side = std::max(b->max_box_side, std::max(c->box_h[k], c->box_w[k]));
        b->max_box_w = std::max(b->max_box_w, (int)c->box_w[k]);
        if (!(c->prox_flags[k] & SMI_COMPONENT_SHIFTING)) continue;
        b->n_shift++;
        b->is_shift[k] = 1;
        pt[8 * k] = c->center[2 * k];
        pt[8 * k + 1] = c->center[2 * k + 1];
        if (c->shift_step) shift_step[k] = c->shift_step[k];
        if (c->shift_rel_step) shift_rel[k] = c->shift_rel_step[k];
        // fft.shift: _get_fft_shape(image, image, padding=10) (fft.py:116-167, 401-403)
        int fy = next_fast_len(2 * c->box_h[k] + 10), fx = next_fast_len(2 * c->box_w[k] + 10);
        while (fx % 2) fx = next_fast_len(fx + 1);
        if (c->box_h[k] % 2 == 0)
            while (fy % 2) fy = next_fast_len(fy + 1);
        shift_fft[2 * k] = fy;
        shift_fft[2 * k + 1] = fx;
    }
    if ((rc = upload(&b->c_shift_step, shift_step.data(), (size_t)n))) return rc;
    if ((rc = upload(&b->c_shift_rel, shift_rel.data(), (size_t)n))) return rc;
    if ((rc = upload(&b->c_shift_fft, shift_fft.data(), (size_t)n * 2))) return rc;
    if (b->shift_scratch) {
        SMI_HIP(hipFree(b->shift_scratch));
        b->shift_scratch = nullptr;
    }
    if (b->n_shift && shift_needs_scratch(max_pix, b->max_box_side))
        SMI_HIP(dev_alloc(&b->shift_scratch, 4 * (size_t)b->n_morph + 16 * (size_t)n));
    if (b->n_shift) {
        // the uploaded images are the parameters; `morph` becomes the shifted image
        if ((rc = upload(&b->morph_param, c->morph, (size_t)b->n_morph))) return rc;
    } else if (b->morph_param) {
        SMI_HIP(hipFree(b->morph_param));
        b->morph_param = nullptr;
    }
    for (int k = 0; k < n; ++k) {
        if (!(c->prox_flags[k] & SMI_COMPONENT_POINT_SOURCE)) continue;
        b->n_point++;
        b->is_point[k] = 1;
        b->box_center[2 * k] = c->origin_y[k] + 0.5 * c->box_h[k];
        b->box_center[2 * k + 1] = c->origin_x[k] + 0.5 * c->box_w[k];
        pt[8 * k] = c->center[2 * k] - b->box_center[2 * k];
        pt[8 * k + 1] = c->center[2 * k + 1] - b->box_center[2 * k + 1];
        sigma[k] = c->psf_sigma[k];
        if (c->psf_beta) {
            SMI_REQUIRE(c->psf_beta[k] >= 0.f, "point source with psf_beta < 0");
            beta[k] = c->psf_beta[k];
        }
    }
    if ((rc = upload(&b->pt, pt.data(), pt.size()))) return rc;
    if ((rc = upload(&b->c_sigma, sigma.data(), (size_t)n))) return rc;
    if ((rc = upload(&b->c_beta, beta.data(), (size_t)n))) return rc;
    if (b->g_center) SMI_HIP(hipFree(b->g_center));
    SMI_HIP(dev_alloc(&b->g_center, (size_t)n * 2 + 2));  // never empty: a batch may hold no component
    SMI_HIP(hipMemset(b->g_center, 0, ((size_t)n * 2 + 2) * sizeof(double)));
    // work list of the register-resident update kernels: components by size class, then
    // blend (BatchView::work)
    {
        std::vector<std::vector<int32_t>> items(kNumUpdateClasses);
        b->h_work_start.assign((size_t)kNumUpdateClasses * (nb + 1), 0);
        for (int i = 0; i < nb; ++i) {
            for (int k = start[i]; k < start[i + 1]; ++k) {
                if (c->prox_flags[k] & SMI_COMPONENT_POINT_SOURCE) continue;
                const int cls = update_class(c->box_h[k] * c->box_w[k]);
                if (cls >= 0) items[cls].push_back(k);  // else: such a batch takes the general kernel
            }
            for (int cls = 0; cls < kNumUpdateClasses; ++cls)
                b->h_work_start[(size_t)cls * (nb + 1) + i + 1] = (int32_t)items[cls].size();
        }
        std::vector<int32_t> work;
        int32_t base = 0;
        for (int cls = 0; cls < kNumUpdateClasses; ++cls) {
            for (int i = 0; i <= nb; ++i) b->h_work_start[(size_t)cls * (nb + 1) + i] += base;
            work.insert(work.end(), items[cls].begin(), items[cls].end());
            base += (int32_t)items[cls].size();
        }
        b->n_size_classes = 0;
        for (int cls = 0; cls < kNumUpdateClasses; ++cls) b->n_size_classes += !items[cls].empty();
        work.push_back(-1);  // never empty
        if ((rc = upload(&b->work_items, work.data(), work.size()))) return rc;
        // the plan most monotonic components of a class use: the one its launches stage in LDS
        for (int cls = 0; cls < kNumUpdateClasses; ++cls) {
            std::map<int32_t, int> uses;
            for (int32_t k : items[cls])
                if ((c->prox_flags[k] & SMI_PROX_MONOTONIC) && !(c->prox_flags[k] & SMI_PROX_FIT_CENTER) &&
                    c->sweep_plan && c->sweep_plan[k] >= 0)
                    uses[c->sweep_plan[k]]++;
            int32_t best = -1;
            int most = 0;
            for (const auto &u : uses)
                if (u.second > most && u.first < (int)b->plans.size() && b->plans[u.first].ring) {
                    best = u.first;
                    most = u.second;
                }
            b->stage_plan[cls] = best;
        }
    }
    b->have_components = true;
    const int max_pixels = b->view.max_box_pixels;
    refresh_view(b);
    b->view.max_box_pixels = max_pixels;
    // the morphology of a point source is derived from its centre
    const BatchView v = unmasked_view(b);
    if (keep) {
        // kept components: pixel arrays from the old packing; the others: their records
        std::vector<int64_t> rec(n, -1);
        int64_t total = 0;
        std::vector<int32_t> kept(keep, keep + n);  // (the codes: carry_states_kernel)
        for (int k = 0; k < n; ++k) {
            if (kept[k]) continue;
            rec[k] = total;
            total += 4 * (int64_t)C + 4 * (moff[k + 1] - moff[k]);
        }
        int32_t *d_keep = nullptr;
        int64_t *d_old = nullptr, *d_rec = nullptr;
        float *d_states = nullptr;
        if ((rc = upload(&d_keep, kept.data(), (size_t)n))) return rc;
        if ((rc = upload(&d_old, old_moff.data(), (size_t)n + 1))) return rc;
        if ((rc = upload(&d_rec, rec.data(), (size_t)n))) return rc;
        if ((rc = upload(&d_states, states, (size_t)total))) return rc;
        float *new_px[4] = {b->morph, b->mom[3], b->mom[4], b->mom[5]};
        launch_carry_states(d_keep, d_old, b->c_moff, n, old_px, new_px, b->stream);
        launch_scatter_states(v, d_rec, d_states, b->stream);
        SMI_HIP(hipStreamSynchronize(b->stream));
        for (void *p : {(void *)d_keep, (void *)d_old, (void *)d_rec, (void *)d_states}) (void)hipFree(p);
        SMI_HIP(hipGetLastError());
        return SMI_OK;
    }
    if ((rc = launch_point_sources(v, nullptr, 0, 0.f, 0, nullptr, nullptr, 2, b->stream))) return rc;
    if ((rc = launch_shift_forward(v, 0, b->stream))) return rc;
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int smi_batch_set_components(smi_batch *b, const smi_components *c) {
    return set_components_impl(b, c, nullptr, nullptr);
}

int smi_batch_update_components(smi_batch *b, const smi_components *c, const int32_t *keep,
                                const float *states) {
    SMI_REQUIRE(keep, "null argument");
    return set_components_impl(b, c, keep, states);
}

int smi_batch_resize_test(smi_batch *b, int32_t *margin, double *edge_pull) {
    SMI_REQUIRE(b && b->have_components && margin && edge_pull, "components not set / null argument");
    SMI_HIP(hipSetDevice(b->device));
    const int n = b->d.n_components;
    int32_t *d_margin = nullptr;
    double *d_pull = nullptr;
    SMI_HIP(dev_alloc(&d_margin, (size_t)n));
    SMI_HIP(dev_alloc(&d_pull, (size_t)n));
    launch_resize_test(unmasked_view(b), d_margin, d_pull, b->stream);
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipMemcpy(margin, d_margin, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    SMI_HIP(hipMemcpy(edge_pull, d_pull, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d_margin);
    (void)hipFree(d_pull);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int smi_batch_get_component_states(smi_batch *b, const int32_t *components, int32_t n_sel,
                                   float *states) {
    SMI_REQUIRE(b && b->have_components && (n_sel == 0 || (components && states)),
                "components not set / null argument");
    SMI_REQUIRE(b->n_point == 0 && b->n_shift == 0, "state records: factorized image components only");
    if (n_sel == 0) return SMI_OK;
    SMI_HIP(hipSetDevice(b->device));
    const int n = b->d.n_components, C = b->d.C;
    std::vector<int64_t> off(n_sel);
    int64_t total = 0;
    for (int j = 0; j < n_sel; ++j) {
        SMI_REQUIRE(components[j] >= 0 && components[j] < n, "component index out of range");
        off[j] = total;
        total += 4 * (int64_t)C + 4 * (b->h_moff[components[j] + 1] - b->h_moff[components[j]]);
    }
    int32_t *d_sel = nullptr;
    int64_t *d_off = nullptr;
    float *d_states = nullptr;
    int rc;
    if ((rc = upload(&d_sel, components, (size_t)n_sel))) return rc;
    if ((rc = upload(&d_off, off.data(), (size_t)n_sel))) return rc;
    SMI_HIP(dev_alloc(&d_states, (size_t)total));
    launch_gather_states(unmasked_view(b), d_sel, d_off, n_sel, d_states, b->stream);
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipMemcpy(states, d_states, (size_t)total * sizeof(float), hipMemcpyDeviceToHost));
    for (void *p : {(void *)d_sel, (void *)d_off, (void *)d_states}) (void)hipFree(p);
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int smi_batch_set_iteration_base(smi_batch *b, const int32_t *base) {
    SMI_REQUIRE(b != nullptr, "null batch");
    SMI_REQUIRE(b->n_point == 0 && b->n_shift == 0 && !b->ks.stamp && b->scheme != SMI_SCHEME_FISTA,
                "smi_batch_set_iteration_base: factorized image components under AMSGrad only");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    if (!base) {
        if (b->it_base) (void)hipFree(b->it_base);
        b->it_base = nullptr;
    } else {
        if (!b->it_base) SMI_HIP(dev_alloc(&b->it_base, (size_t)b->d.n_blends));
        SMI_HIP(hipMemcpy(b->it_base, base, b->d.n_blends * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    refresh_view(b);
    return SMI_OK;
}

int smi_batch_set_pause_at(smi_batch *b, const int32_t *it) {
    SMI_REQUIRE(b != nullptr, "null batch");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const size_t nb = (size_t)b->d.n_blends;
    if (!it) {
        if (b->pause_at) (void)hipFree(b->pause_at);
        if (b->conv_flag) (void)hipFree(b->conv_flag);
        b->pause_at = b->conv_flag = nullptr;
    } else {
        if (!b->pause_at) {
            SMI_HIP(dev_alloc(&b->pause_at, nb));
            SMI_HIP(dev_alloc(&b->conv_flag, nb));
        }
        SMI_HIP(hipMemcpy(b->pause_at, it, nb * sizeof(int32_t), hipMemcpyHostToDevice));
        SMI_HIP(hipMemset(b->conv_flag, 0, nb * sizeof(int32_t)));
    }
    refresh_view(b);
    return SMI_OK;
}

int smi_batch_set_round(smi_batch *b, const int32_t *state, const int32_t *base,
                        const int32_t *pause_at) {
    SMI_REQUIRE(b != nullptr, "null batch");
    SMI_REQUIRE(!base || (b->n_point == 0 && b->n_shift == 0 && !b->ks.stamp && b->scheme != SMI_SCHEME_FISTA),
                "smi_batch_set_round: iteration bases for factorized image components under AMSGrad only");
    SMI_HIP(hipSetDevice(b->device));
    const size_t nb = (size_t)b->d.n_blends, bytes = nb * sizeof(int32_t);
    bool grew = false;
    if (!b->h_round) SMI_HIP(hipHostMalloc((void **)&b->h_round, 3 * bytes, hipHostMallocDefault));
    if (base && !b->it_base) {
        SMI_HIP(dev_alloc(&b->it_base, nb));
        grew = true;
    }
    if (pause_at && !b->pause_at) {
        SMI_HIP(dev_alloc(&b->pause_at, nb));
        SMI_HIP(dev_alloc(&b->conv_flag, nb));
        grew = true;
    }
    // (the staging buffer may still feed the copies of the last call)
    SMI_HIP(hipStreamSynchronize(b->stream));
    if (state) {
        memcpy(b->h_round, state, bytes);
        SMI_HIP(hipMemcpyAsync(b->state, b->h_round, bytes, hipMemcpyHostToDevice, b->stream));
    }
    if (base) {
        memcpy(b->h_round + nb, base, bytes);
        SMI_HIP(hipMemcpyAsync(b->it_base, b->h_round + nb, bytes, hipMemcpyHostToDevice, b->stream));
    }
    if (pause_at) {
        memcpy(b->h_round + 2 * nb, pause_at, bytes);
        SMI_HIP(hipMemcpyAsync(b->pause_at, b->h_round + 2 * nb, bytes, hipMemcpyHostToDevice, b->stream));
        SMI_HIP(hipMemsetAsync(b->conv_flag, 0, bytes, b->stream));
    }
    if (grew) refresh_view(b);
    return SMI_OK;
}

int smi_batch_get_round(smi_batch *b, int32_t *state, int32_t *n_loss, int32_t *converged) {
    SMI_REQUIRE(b && state && n_loss, "null argument");
    SMI_REQUIRE(!converged || b->conv_flag, "smi_batch_get_round: converged follows a pause_at");
    SMI_HIP(hipSetDevice(b->device));
    const size_t nb = (size_t)b->d.n_blends, bytes = nb * sizeof(int32_t);
    if (!b->h_round) SMI_HIP(hipHostMalloc((void **)&b->h_round, 3 * bytes, hipHostMallocDefault));
    SMI_HIP(hipMemcpyAsync(b->h_round, b->state, bytes, hipMemcpyDeviceToHost, b->stream));
    SMI_HIP(hipMemcpyAsync(b->h_round + nb, b->n_loss, bytes, hipMemcpyDeviceToHost, b->stream));
    if (converged)
        SMI_HIP(hipMemcpyAsync(b->h_round + 2 * nb, b->conv_flag, bytes, hipMemcpyDeviceToHost, b->stream));
    SMI_HIP(hipStreamSynchronize(b->stream));
    for (size_t i = 0; i < nb; ++i) state[i] = std::min(b->h_round[i], 3);
    memcpy(n_loss, b->h_round + nb, bytes);
    if (converged) memcpy(converged, b->h_round + 2 * nb, bytes);
    return SMI_OK;
}

int smi_batch_get_converged(smi_batch *b, int32_t *flag) {
    SMI_REQUIRE(b && flag, "null argument");
    SMI_REQUIRE(b->conv_flag != nullptr, "smi_batch_get_converged follows smi_batch_set_pause_at");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipMemcpy(flag, b->conv_flag, (size_t)b->d.n_blends * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SMI_OK;
}

int smi_batch_set_states(smi_batch *b, const int32_t *state) {
    SMI_REQUIRE(b && state, "null argument");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipMemcpy(b->state, state, b->d.n_blends * sizeof(int32_t), hipMemcpyHostToDevice));
    return SMI_OK;
}

int smi_batch_get_progress(smi_batch *b, int32_t *state, int32_t *n_loss) {
    SMI_REQUIRE(b && state && n_loss, "null argument");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const size_t nb = (size_t)b->d.n_blends;
    SMI_HIP(hipMemcpy(state, b->state, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    SMI_HIP(hipMemcpy(n_loss, b->n_loss, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nb; ++i) state[i] = std::min(state[i], 3);
    return SMI_OK;
}

int smi_batch_get_model_morphology(smi_batch *b, float *morph) {
    SMI_REQUIRE(b && b->have_components && morph, "components not set / null argument");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipMemcpy(morph, b->morph, (size_t)b->n_morph * sizeof(float), hipMemcpyDeviceToHost));
    return SMI_OK;
}

int smi_batch_get_centers(smi_batch *b, double *center, double *m, double *v, double *vhat,
                          double *gradient) {
    SMI_REQUIRE(b && b->have_components, "components not set");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int n = b->d.n_components;
    std::vector<double> pt((size_t)n * 8);
    if (n) SMI_HIP(hipMemcpy(pt.data(), b->pt, pt.size() * sizeof(double), hipMemcpyDeviceToHost));
    double *dst[4] = {center, m, v, vhat};
    for (int f = 0; f < 4; ++f) {
        if (!dst[f]) continue;
        for (int k = 0; k < n; ++k)
            for (int a = 0; a < 2; ++a) {
                double val = pt[8 * k + 2 * f + a];
                if (f == 0 && b->is_point[k]) val += b->box_center[2 * k + a];
                dst[f][2 * k + a] = val;
            }
    }
    if (gradient && n)
        SMI_HIP(hipMemcpy(gradient, b->g_center, (size_t)n * 2 * sizeof(double),
                          hipMemcpyDeviceToHost));
    return SMI_OK;
}

int smi_batch_set_centers(smi_batch *b, const double *center) {
    SMI_REQUIRE(b && b->have_components && center, "components not set / null argument");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int n = b->d.n_components;
    if (!n) return SMI_OK;
    std::vector<double> pt((size_t)n * 8);
    SMI_HIP(hipMemcpy(pt.data(), b->pt, pt.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < n; ++k) {
        const bool free2 = b->is_point[k] || b->is_shift[k];
        if (!free2) continue;  // (entries of other components are ignored)
        for (int a = 0; a < 2; ++a)
            pt[8 * k + a] = center[2 * k + a] - (b->is_point[k] ? b->box_center[2 * k + a] : 0.0);
    }
    SMI_HIP(hipMemcpy(b->pt, pt.data(), pt.size() * sizeof(double), hipMemcpyHostToDevice));
    // what enters the model follows: the PSF at the new centre, the image at the new shift
    const BatchView v = unmasked_view(b);
    int rc;
    if ((rc = launch_point_sources(v, nullptr, 0, 0.f, 0, nullptr, nullptr, 2, b->stream))) return rc;
    if ((rc = launch_shift_forward(v, 0, b->stream))) return rc;
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int smi_batch_set_center_moments(smi_batch *b, const double *m, const double *v,
                                 const double *vhat) {
    SMI_REQUIRE(b && b->have_components, "components not set");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int n = b->d.n_components;
    if (!n) return SMI_OK;
    std::vector<double> pt((size_t)n * 8);
    SMI_HIP(hipMemcpy(pt.data(), b->pt, pt.size() * sizeof(double), hipMemcpyDeviceToHost));
    const double *src[3] = {m, v, vhat};
    for (int f = 0; f < 3; ++f)
        for (int k = 0; k < n; ++k)
            for (int a = 0; a < 2; ++a) pt[8 * k + 2 * (f + 1) + a] = src[f] ? src[f][2 * k + a] : 0.0;
    SMI_HIP(hipMemcpy(b->pt, pt.data(), pt.size() * sizeof(double), hipMemcpyHostToDevice));
    return SMI_OK;
}

int smi_batch_set_moments(smi_batch *b, const float *m_sed, const float *v_sed,
                          const float *vhat_sed, const float *m_morph, const float *v_morph,
                          const float *vhat_morph) {
    SMI_REQUIRE(b && b->have_components, "components not set");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));  // pending steps may still read the old arrays
    const float *src[6] = {m_sed, v_sed, vhat_sed, m_morph, v_morph, vhat_morph};
    for (int i = 0; i < 6; ++i) {
        const size_t cnt = i < 3 ? (size_t)b->d.n_components * b->d.C : (size_t)b->n_morph;
        if (src[i])
            SMI_HIP(hipMemcpy(b->mom[i], src[i], cnt * sizeof(float), hipMemcpyHostToDevice));
        else
            SMI_HIP(hipMemset(b->mom[i], 0, (cnt ? cnt : 1) * sizeof(float)));
    }
    return SMI_OK;
}

int smi_batch_get_moments(smi_batch *b, float *m_sed, float *v_sed, float *vhat_sed,
                          float *m_morph, float *v_morph, float *vhat_morph) {
    SMI_REQUIRE(b && b->have_components, "components not set");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    float *dst[6] = {m_sed, v_sed, vhat_sed, m_morph, v_morph, vhat_morph};
    for (int i = 0; i < 6; ++i) {
        const size_t cnt = i < 3 ? (size_t)b->d.n_components * b->d.C : (size_t)b->n_morph;
        if (dst[i]) SMI_HIP(hipMemcpy(dst[i], b->mom[i], cnt * sizeof(float), hipMemcpyDeviceToHost));
    }
    return SMI_OK;
}

int smi_batch_get_parameters(smi_batch *b, float *sed, float *morph) {
    SMI_REQUIRE(b && b->have_components, "components not set");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    if (sed)
        SMI_HIP(hipMemcpy(sed, b->sed, (size_t)b->d.n_components * b->d.C * sizeof(float),
                          hipMemcpyDeviceToHost));
    if (morph) {
        SMI_HIP(hipMemcpy(morph, b->morph, (size_t)b->n_morph * sizeof(float),
                          hipMemcpyDeviceToHost));
        // the parameter of a shifting component is its unshifted image
        for (int k = 0; k < b->d.n_components && b->n_shift; ++k)
            if (b->is_shift[k])
                SMI_HIP(hipMemcpy(morph + b->h_moff[k], b->morph_param + b->h_moff[k],
                                  (size_t)(b->h_moff[k + 1] - b->h_moff[k]) * sizeof(float),
                                  hipMemcpyDeviceToHost));
    }
    return SMI_OK;
}

int smi_batch_set_parameters(smi_batch *b, const float *sed, const float *morph) {
    SMI_REQUIRE(b && b->have_components, "components not set");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    if (sed)
        SMI_HIP(hipMemcpy(b->sed, sed, (size_t)b->d.n_components * b->d.C * sizeof(float),
                          hipMemcpyHostToDevice));
    if (morph) {
        SMI_HIP(hipMemcpy(b->morph, morph, (size_t)b->n_morph * sizeof(float),
                          hipMemcpyHostToDevice));
        if (b->n_shift) {
            SMI_HIP(hipMemcpy(b->morph_param, morph, (size_t)b->n_morph * sizeof(float),
                              hipMemcpyHostToDevice));
            int rc = launch_shift_forward(unmasked_view(b), 0, b->stream);
            if (rc) return rc;
            SMI_HIP(hipStreamSynchronize(b->stream));
        }
    }
    return SMI_OK;
}

int smi_batch_set_optimizer(smi_batch *b, float b1, float b2, float eps) {
    SMI_REQUIRE(b, "null batch");
    SMI_REQUIRE(b1 >= 0.f && b1 < 1.f && b2 >= 0.f && b2 < 1.f && eps >= 0.f, "bad AMSGrad constants");
    b->b1 = b1;
    b->b2 = b2;
    b->eps = eps;
    const int keep = b->view.max_box_pixels;
    refresh_view(b);
    b->view.max_box_pixels = keep;
    return SMI_OK;
}

int smi_batch_set_stream(smi_batch *b, void *stream) {
    SMI_REQUIRE(b, "null batch");
    b->stream = reinterpret_cast<hipStream_t>(stream);
    if (b->info) SMI_FFT(rocfft_execution_info_set_stream(b->info, b->stream));
    return SMI_OK;
}

// every attached low-resolution observation: rendering, loss term and (backward) gradient
static int lowres_evaluate_all(smi_batch *b, int backward) {
    for (auto *l : b->lowres) {
        const int rc = lowres_evaluate(l, b->P, b->Py, b->Px, backward, b->stream);
        if (rc) return rc;
    }
    return SMI_OK;
}

static void lowres_add_all(smi_batch *b) {
    for (auto *l : b->lowres) lowres_add_gradient(l, b->Q, b->Py, b->Px, b->stream);
}

int smi_batch_forward(smi_batch *b, float *model, float *rendered, double *logL) {
    int rc = ready(b);
    if (rc) return rc;
    const BatchView v = unmasked_view(b);
    const int nb = v.nb, C = v.C, H = v.H, W = v.W;
    const size_t n_out = (size_t)nb * C * H * W;
    launch_render(v, b->P, b->stream);
    if ((rc = lowres_evaluate_all(b, 0))) return rc;
    float *tmp = nullptr;
    if (model || rendered) SMI_HIP(dev_alloc(&tmp, n_out));
    if (model) {
        launch_crop(b->P, tmp, nb * C, H, W, b->Py, b->Px, b->stream);
        SMI_HIP(hipMemcpyAsync(model, tmp, n_out * sizeof(float), hipMemcpyDeviceToHost, b->stream));
        SMI_HIP(hipStreamSynchronize(b->stream));
    }
    if (b->fused) {
        // mode 1: rendered cube to Q (compact) and the loss partials
        if ((rc = launch_fused_conv(v, b->Fy, b->Fx, b->P, b->Kt, b->d.kernel_bands,
                                    b->d.kernel_per_blend, b->Q, 1, nullptr, b->stream)))
            return rc;
        if ((rc = layers_evaluate(b, v, 0, b->stream))) return rc;
    } else if ((rc = convolve(b, v, 0))) {
        return rc;
    }
    if (rendered) {
        launch_crop(b->Q, tmp, nb * C, H, W, b->Py, b->Px, b->stream);
        SMI_HIP(hipMemcpyAsync(rendered, tmp, n_out * sizeof(float), hipMemcpyDeviceToHost, b->stream));
        SMI_HIP(hipStreamSynchronize(b->stream));
    }
    if (tmp) (void)hipFree(tmp);
    if (logL) {
        if (!b->fused) launch_residual(v, b->Q, b->P, b->stream);
        std::vector<double> part((size_t)nb * v.n_partial), ln(nb);
        SMI_HIP(hipMemcpyAsync(part.data(), b->loss_partial, part.size() * sizeof(double),
                               hipMemcpyDeviceToHost, b->stream));
        SMI_HIP(hipMemcpyAsync(ln.data(), b->log_norm, nb * sizeof(double), hipMemcpyDeviceToHost,
                               b->stream));
        std::vector<double> terms(b->lowres.size(), 0.0);
        if (!terms.empty())
            SMI_HIP(hipMemcpyAsync(terms.data(), b->extra_terms, terms.size() * sizeof(double),
                                   hipMemcpyDeviceToHost, b->stream));
        SMI_HIP(hipStreamSynchronize(b->stream));
        for (int i = 0; i < nb; ++i) {
            double t = 0.0;
            for (int j = 0; j < v.n_partial; ++j) t += part[(size_t)i * v.n_partial + j];
            for (double e : terms) t += 2.0 * e;
            logL[i] = -(ln[i] + 0.5 * t);
        }
    }
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

int smi_batch_gradient(smi_batch *b, float *g_sed, float *g_morph) {
    int rc = ready(b);
    if (rc) return rc;
    const BatchView v = unmasked_view(b);
    if (b->fused) {
        launch_render(v, b->P, b->stream);
        if ((rc = lowres_evaluate_all(b, 1))) return rc;
        if (b->ks.stamp && (rc = kernel_shift_backward(b, v, 0, 1))) return rc;
        if ((rc = launch_fused_conv(v, b->Fy, b->Fx, b->P, b->Kt, b->d.kernel_bands,
                                    b->d.kernel_per_blend, b->Q, 0, b->dbg, b->stream)))
            return rc;
        if ((rc = layers_evaluate(b, v, 1, b->stream))) return rc;
    } else {
        launch_render(v, b->P, b->stream);
        if ((rc = lowres_evaluate_all(b, 1))) return rc;
        if ((rc = convolve(b, v, 0))) return rc;
        launch_residual(v, b->Q, b->P, b->stream);
        if ((rc = convolve(b, v, 1))) return rc;
    }
    lowres_add_all(b);
    if ((rc = launch_update(v, b->Q, 0, 0.f, 0, b->g_sed, b->g_morph, 1, b->stream))) return rc;
    if ((rc = launch_point_sources(v, b->Q, 0, 0.f, 0, b->g_sed, b->g_center, 1, b->stream)))
        return rc;
    if ((rc = launch_shift_backward(v, b->Q, 0, b->g_center, 1, b->stream))) return rc;
    SMI_HIP(hipStreamSynchronize(b->stream));
    if (g_sed)
        SMI_HIP(hipMemcpy(g_sed, b->g_sed, (size_t)v.n_comp * v.C * sizeof(float),
                          hipMemcpyDeviceToHost));
    if (g_morph)
        SMI_HIP(hipMemcpy(g_morph, b->g_morph, (size_t)b->n_morph * sizeof(float),
                          hipMemcpyDeviceToHost));
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

// nothing but factorized components under one fused convolution
static bool plain_batch(const smi_batch *b) {
    return b->fused && b->n_point == 0 && b->n_shift == 0 && b->lowres.empty() &&
           b->layers.empty() && !b->ks.stamp;
}

// A plain batch needs the model only as the input rows of the convolution, and the
// convolution kernel renders them itself (fused_conv.hip, ModelGather): no model cube, no
// render launch per iteration.  SMI_INLINE_RENDER=0 keeps the cube (development aid).
static bool inline_render(const smi_batch *b) {
    static const bool allowed = [] {
        const char *e = getenv("SMI_INLINE_RENDER");
        return !(e && e[0] == '0');
    }();
    // (point sources and shifted images are morphologies in `morph` like any other by the time
    // the convolution runs; further observations, a low-resolution term and a free kernel shift
    // read the cube)
    return allowed && b->inline_render && b->fused && b->lowres.empty() && b->layers.empty() &&
           !b->ks.stamp && b->view.render_slots > 0;
}


// launch_update forks the launches per size class onto streams of their own (BatchView::class_streams)
static bool class_streams_pay(const smi_batch *b) {
    static const int forced = [] {  // development aid: 0 never, 1 whenever there are two classes
        const char *e = getenv("SMI_RANGE_CLASS_STREAMS");
        return e ? atoi(e) : -1;
    }();
    if (forced == 0 || b->n_size_classes < 2) return false;
    if (forced > 0) return true;
    // (512 blends of the quickstart scene: four ranges 1 527 k, two ranges with class streams
    // 1 374 k blend-it/s; 768: 1 541 / 1 507; 1024: 1 510 / 1 620; 2048: 1 670 / 1 758)
    return g_hw_queues.load(std::memory_order_relaxed) >= 8 && b->d.n_components >= 10000;
}

// number of blend ranges a step is split into
static int sub_ranges(const smi_batch *b) {
    if (!plain_batch(b)) return 1;
    const int nb = b->d.n_blends;
    // measured on MI355X (bench.py --blends N --sub-ranges n --steps 20, k blend-it/s for
    // n = 2 / 3 / 4): 128 blends 506 / 508 / 335, 256 blends 658 / 673 / 498, 512 blends
    // 760 / 785 / 662, 1024 blends 824 / 829 / 781.  Range 0 runs on the batch stream, so
    // three ranges sit on three of HIP's four hardware queues; a fourth shares one (with
    // GPU_MAX_HW_QUEUES=8 four ranges give 517 / 842 k at 128 / 1024 blends, six fewer).
    // Round 3: a small batch is bound by the serial chain of a range (convolution ~60 us, then
    // update ~140 us in the early iterations), not by the chip, so a fourth range adds
    // throughput where a fourth hardware queue exists -- GPU_MAX_HW_QUEUES = 8 (k blend-it/s,
    // 3 / 4 / 5 ranges): 128 blends 537 / 615 / 391, 256: 746 / 776 / 603, 512: 809 / 814 /
    // 740, 1024: 919 / 857 / 826.
    const int queues = g_hw_queues.load(std::memory_order_relaxed);  // smi_set_hw_queues
    int n = b->n_sub > 0 ? b->n_sub : (nb < 128 ? 1 : (queues >= 8 && nb < 768) ? 4 : 3);
    // boxes of several size classes (one latency-bound launch per class and range): two ranges
    // whose class launches run side by side on streams of their own -- 2 x 4 streams on eight
    // hardware queues (batch of 1024 quickstart blends, k blend-it/s over 100 iterations: three
    // ranges 1 511, four 1 570, two ranges with class streams 1 617; round 6)
    if (b->n_sub <= 0 && class_streams_pay(b)) n = 2;
    return std::max(1, std::min(n, nb));
}

// The fused path with the batch split into ranges of blends, each on its own stream.
static int step_sub_ranges(smi_batch *b, int n_sub, int32_t it0, int32_t n_iter, float e_rel,
                           int32_t min_iter, int32_t prox_max_iter, int check) {
    int rc;
    // range 0 stays on the batch stream, the others get streams of their own.  HIP maps
    // streams onto 4 hardware queues round-robin (GPU_MAX_HW_QUEUES); streams that share a
    // queue run one after the other, so more than 4 ranges do not overlap any further.
    while ((int)b->sub_streams.size() < n_sub - 1) {
        hipStream_t st;
        SMI_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        b->sub_streams.push_back(st);
    }
    while ((int)b->sub_events.size() < n_sub) {
        hipEvent_t e;
        SMI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        b->sub_events.push_back(e);
    }
    const bool timing = b->timing;
    const bool no_cube = inline_render(b);
    if (timing) {
        const size_t need = (size_t)n_iter * 6;
        while (b->events.size() < need) {
            hipEvent_t e;
            SMI_HIP(hipEventCreate(&e));
            b->events.push_back(e);
        }
    }
    const int nb = b->d.n_blends;
    std::vector<BatchView> views(n_sub, b->view);
    for (int s = 0; s < n_sub; ++s) {
        const int lo = (int)((int64_t)s * nb / n_sub), hi = (int)((int64_t)(s + 1) * nb / n_sub);
        views[s].blend0 = lo;
        views[s].range_slot = s;
        views[s].nb = hi - lo;
        views[s].comp0 = b->h_comp_start[lo];
        views[s].n_comp = b->h_comp_start[hi] - b->h_comp_start[lo];
    }
    SMI_HIP(hipEventRecord(b->sub_events[0], b->stream));
    for (int s = 1; s < n_sub; ++s)
        SMI_HIP(hipStreamWaitEvent(b->sub_streams[s - 1], b->sub_events[0], 0));
    for (int i = 0; i < n_iter; ++i) {
        const int it = it0 + i;
        for (int s = 0; s < n_sub; ++s) {
            views[s].fail_code = 3 + it;  // (state word of a blend that fails in this iteration)
            const BatchView &v = views[s];
            hipStream_t st = s == 0 ? b->stream : b->sub_streams[s - 1];
            // phase times are those of range 0 (its kernels overlap the other ranges')
            hipEvent_t *ev = timing && s == 0 ? &b->events[(size_t)i * 6] : nullptr;
            if (ev) SMI_HIP(hipEventRecord(ev[0], st));
            if (!no_cube) launch_render(v, b->P, st);
            if (ev) SMI_HIP(hipEventRecord(ev[1], st));
            if ((rc = launch_fused_conv(v, b->Fy, b->Fx, no_cube ? nullptr : b->P, b->Kt, b->d.kernel_bands,
                                        b->d.kernel_per_blend, b->Q, 0, nullptr, st)))
                return rc;
            if (ev) SMI_HIP(hipEventRecord(ev[2], st));
            if (ev) SMI_HIP(hipEventRecord(ev[3], st));
            if (ev) SMI_HIP(hipEventRecord(ev[4], st));
            // (the loss bookkeeping rides along with the updates: kernels.hip, finalize_blend)
            if ((rc = launch_update_finalize(v, b->Q, it, e_rel, min_iter, check, prox_max_iter, st)))
                return rc;
            if (check) launch_advance(v, st);
            if (ev) SMI_HIP(hipEventRecord(ev[5], st));
        }
    }
    for (int s = 1; s < n_sub; ++s) {
        SMI_HIP(hipEventRecord(b->sub_events[s], b->sub_streams[s - 1]));
        SMI_HIP(hipStreamWaitEvent(b->stream, b->sub_events[s], 0));
    }
    SMI_HIP(hipGetLastError());
    return SMI_OK;
}

static int collect_phase_times(smi_batch *b, int n_iter) {
    SMI_HIP(hipStreamSynchronize(b->stream));
    for (int p = 0; p < 6; ++p) b->phase_ms[p] = 0.0;
    for (int i = 0; i < n_iter; ++i) {
        hipEvent_t *ev = &b->events[(size_t)i * 6];
        for (int p = 0; p < 5; ++p) {
            float ms = 0.f;
            SMI_HIP(hipEventElapsedTime(&ms, ev[p], ev[p + 1]));
            b->phase_ms[p] += ms;
        }
        float tot = 0.f;
        SMI_HIP(hipEventElapsedTime(&tot, ev[0], ev[5]));
        b->phase_ms[5] += tot;
    }
    for (int p = 0; p < 6; ++p) b->phase_ms[p] /= n_iter;
    b->timed_iters = n_iter;
    return SMI_OK;
}

int smi_batch_step(smi_batch *b, int32_t it0, int32_t n_iter, float e_rel, int32_t min_iter,
                   int32_t prox_max_iter, int32_t check_convergence) {
    int rc = ready(b);
    if (rc) return rc;
    SMI_REQUIRE(n_iter >= 0 && it0 >= 0 && prox_max_iter >= 0, "bad iteration arguments");
    const BatchView &v = b->view;
    const int check = check_convergence != 0;
    const bool timing = b->timing;
    b->view.class_streams = class_streams_pay(b) ? 1 : 0;
    const int n_sub = sub_ranges(b);
    const bool no_cube = inline_render(b), plain = plain_batch(b);
    if (n_sub > 1) {
        if ((rc = step_sub_ranges(b, n_sub, it0, n_iter, e_rel, min_iter, prox_max_iter, check)))
            return rc;
        return timing && n_iter > 0 ? collect_phase_times(b, n_iter) : SMI_OK;
    }
    if (timing) {
        const size_t need = (size_t)n_iter * 6;
        while (b->events.size() < need) {
            hipEvent_t e;
            SMI_HIP(hipEventCreate(&e));
            b->events.push_back(e);
        }
    }
    for (int i = 0; i < n_iter; ++i) {
        const int it = it0 + i;
        b->view.fail_code = 3 + it;  // (state word of a blend that fails in this iteration)
        hipEvent_t *ev = timing ? &b->events[(size_t)i * 6] : nullptr;
        if (ev) SMI_HIP(hipEventRecord(ev[0], b->stream));
        if (b->fused) {
            if (!no_cube) launch_render(v, b->P, b->stream);
            if ((rc = lowres_evaluate_all(b, 1))) return rc;
            if (b->ks.stamp && (rc = kernel_shift_backward(b, v, it, 0))) return rc;
            // conv + residual/loss + conv^T in one LDS-resident kernel
            if (ev) SMI_HIP(hipEventRecord(ev[1], b->stream));
            if ((rc = launch_fused_conv(v, b->Fy, b->Fx, no_cube ? nullptr : b->P, b->Kt, b->d.kernel_bands,
                                        b->d.kernel_per_blend, b->Q, 0, b->dbg, b->stream)))
                return rc;
            if ((rc = layers_evaluate(b, v, 1, b->stream))) return rc;
            if (ev) SMI_HIP(hipEventRecord(ev[2], b->stream));
            if (!plain) launch_finalize(v, it, e_rel, min_iter, check, b->stream);
            if (ev) SMI_HIP(hipEventRecord(ev[3], b->stream));
            if (ev) SMI_HIP(hipEventRecord(ev[4], b->stream));
        } else {
            launch_render(v, b->P, b->stream);
            if ((rc = lowres_evaluate_all(b, 1))) return rc;
            if (ev) SMI_HIP(hipEventRecord(ev[1], b->stream));
            if ((rc = convolve(b, v, 0))) return rc;
            if (ev) SMI_HIP(hipEventRecord(ev[2], b->stream));
            launch_residual(v, b->Q, b->P, b->stream);
            launch_finalize(v, it, e_rel, min_iter, check, b->stream);
            if (ev) SMI_HIP(hipEventRecord(ev[3], b->stream));
            if ((rc = convolve(b, v, 1))) return rc;
            if (ev) SMI_HIP(hipEventRecord(ev[4], b->stream));
        }
        lowres_add_all(b);
        if ((rc = launch_shift_backward(v, b->Q, it, nullptr, 0, b->stream))) return rc;
        if (plain) {  // (the loss bookkeeping rides along with the updates)
            if ((rc = launch_update_finalize(v, b->Q, it, e_rel, min_iter, check, prox_max_iter,
                                             b->stream)))
                return rc;
        } else if ((rc = launch_update(v, b->Q, it, e_rel, prox_max_iter, nullptr, nullptr, 0,
                                       b->stream))) {
            return rc;
        }
        if ((rc = launch_point_sources(v, b->Q, it, e_rel, prox_max_iter, nullptr, nullptr, 0,
                                       b->stream)))
            return rc;
        if ((rc = launch_shift_forward(v, 1, b->stream))) return rc;
        if (b->ks.stamp && (rc = refresh_shifted_kernel(b, 1))) return rc;
        if (check) launch_advance(v, b->stream);
        if (ev) SMI_HIP(hipEventRecord(ev[5], b->stream));
    }
    SMI_HIP(hipGetLastError());
    if (timing && n_iter > 0) return collect_phase_times(b, n_iter);
    return SMI_OK;
}

int smi_batch_set_sub_ranges(smi_batch *b, int32_t n) {
    SMI_REQUIRE(b && n >= 0, "bad argument");
    b->n_sub = n;
    return SMI_OK;
}

int smi_batch_set_inline_render(smi_batch *b, int32_t on) {
    SMI_REQUIRE(b != nullptr, "null batch");
    b->inline_render = on != 0;
    return SMI_OK;
}

int smi_batch_get_sub_ranges(smi_batch *b, int32_t *n) {
    SMI_REQUIRE(b && n, "null argument");
    *n = sub_ranges(b);
    return SMI_OK;
}

int smi_batch_status(smi_batch *b, int32_t *n_active, int32_t *first_error) {
    SMI_REQUIRE(b, "null batch");
    SMI_HIP(hipSetDevice(b->device));
    launch_count_active(b->state, b->d.n_blends, b->status_out, b->stream);
    int32_t out[2] = {0, -1};
    SMI_HIP(hipMemcpyAsync(out, b->status_out, sizeof(out), hipMemcpyDeviceToHost, b->stream));
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipGetLastError());
    if (n_active) *n_active = out[0];
    if (first_error) *first_error = out[1];
    return SMI_OK;
}

int smi_batch_get_states(smi_batch *b, int32_t *state) {
    SMI_REQUIRE(b && state, "null argument");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipMemcpy(state, b->state, b->d.n_blends * sizeof(int32_t), hipMemcpyDeviceToHost));
    // (on the device a failed blend carries 3 + the iteration it failed in: finalize_blend)
    for (int32_t i = 0; i < b->d.n_blends; ++i) state[i] = std::min(state[i], 3);
    return SMI_OK;
}

int smi_batch_fit(smi_batch *b, int32_t max_iter, float e_rel, int32_t min_iter,
                  int32_t prox_max_iter, int32_t sync_every, int32_t *n_iter) {
    int rc = ready(b);
    if (rc) return rc;
    SMI_REQUIRE(max_iter >= 0, "negative max_iter");
    if (sync_every <= 0) sync_every = 10;
    int it = 0;
    while (it < max_iter) {
        const int chunk = std::min(sync_every, max_iter - it);
        if ((rc = smi_batch_step(b, it, chunk, e_rel, min_iter, prox_max_iter, 1))) return rc;
        it += chunk;
        int32_t active = 0, err = -1;
        if ((rc = smi_batch_status(b, &active, &err))) return rc;
        if (err >= 0) {
            set_error("blend " + std::to_string(err) + ": parameters are not finite");
            return SMI_ERR_ARITHMETIC;
        }
        if (active == 0) break;
    }
    if (n_iter) {
        SMI_HIP(hipMemcpy(n_iter, b->n_loss, b->d.n_blends * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return SMI_OK;
}

int smi_batch_get_loss(smi_batch *b, double *out, int32_t capacity, int32_t *n_iter) {
    SMI_REQUIRE(b, "null batch");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int nb = b->d.n_blends, cap = b->d.max_iter;
    std::vector<int32_t> n(nb);
    SMI_HIP(hipMemcpy(n.data(), b->n_loss, nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_iter) std::memcpy(n_iter, n.data(), nb * sizeof(int32_t));
    if (out) {
        SMI_REQUIRE(capacity > 0, "capacity must be positive");
        std::vector<double> hist((size_t)nb * cap);
        SMI_HIP(hipMemcpy(hist.data(), b->loss_hist, hist.size() * sizeof(double),
                          hipMemcpyDeviceToHost));
        const double nan = std::numeric_limits<double>::quiet_NaN();
        for (int i = 0; i < nb; ++i)
            for (int j = 0; j < capacity; ++j)
                out[(size_t)i * capacity + j] =
                    (j < n[i] && j < cap) ? hist[(size_t)i * cap + j] : nan;
    }
    return SMI_OK;
}

int smi_batch_reset(smi_batch *b) {
    SMI_REQUIRE(b, "null batch");
    SMI_HIP(hipSetDevice(b->device));
    SMI_HIP(hipStreamSynchronize(b->stream));
    const int nb = b->d.n_blends;
    SMI_HIP(hipMemset(b->state, 0, nb * sizeof(int32_t)));
    SMI_HIP(hipMemset(b->n_loss, 0, nb * sizeof(int32_t)));
    SMI_HIP(hipMemset(b->last_loss, 0, nb * sizeof(double)));
    SMI_HIP(hipMemset(b->have_prev, 0, nb * sizeof(int32_t)));
    return SMI_OK;
}

namespace {
// the arrays a step changes, in the order of smi_batch::saved
void mutable_state(smi_batch *b, void **ptr, size_t *bytes) {
    const size_t n = (size_t)b->d.n_components, nC = n * b->d.C, nm = (size_t)b->n_morph;
    void *p[12] = {b->sed, b->morph, b->morph_param, b->mom[0], b->mom[1], b->mom[2], b->mom[3],
                   b->mom[4], b->mom[5], b->fista_t, b->pt, b->ks.state};
    const size_t s[12] = {nC * 4, nm * 4, nm * 4, nC * 4, nC * 4, nC * 4, nm * 4, nm * 4, nm * 4,
                          n * 2 * sizeof(double), n * 8 * sizeof(double),
                          (size_t)b->ks.n_sets * 10 * sizeof(double)};
    for (int i = 0; i < 12; ++i) {
        ptr[i] = p[i];
        bytes[i] = p[i] ? s[i] : 0;
    }
}
}  // namespace

int smi_batch_save_state(smi_batch *b) {
    SMI_REQUIRE(b && b->have_components, "components not set");
    SMI_HIP(hipSetDevice(b->device));
    void *ptr[12];
    size_t bytes[12];
    mutable_state(b, ptr, bytes);
    for (int i = 0; i < 12; ++i) {
        if (b->saved[i] && b->saved_bytes[i] != bytes[i]) {
            SMI_HIP(hipFree(b->saved[i]));
            b->saved[i] = nullptr;
        }
        b->saved_bytes[i] = bytes[i];
        if (!bytes[i]) continue;
        if (!b->saved[i]) SMI_HIP(hipMalloc(&b->saved[i], bytes[i]));
        SMI_HIP(hipMemcpyAsync(b->saved[i], ptr[i], bytes[i], hipMemcpyDeviceToDevice, b->stream));
    }
    return SMI_OK;
}

int smi_batch_restore_state(smi_batch *b) {
    SMI_REQUIRE(b && b->have_components, "components not set");
    SMI_HIP(hipSetDevice(b->device));
    void *ptr[12];
    size_t bytes[12];
    mutable_state(b, ptr, bytes);
    for (int i = 0; i < 12; ++i) {
        SMI_REQUIRE(bytes[i] == b->saved_bytes[i] && (!bytes[i] || b->saved[i]),
                    "no saved state for these components (smi_batch_save_state)");
        if (bytes[i])
            SMI_HIP(hipMemcpyAsync(ptr[i], b->saved[i], bytes[i], hipMemcpyDeviceToDevice, b->stream));
    }
    const int nb = b->d.n_blends;
    SMI_HIP(hipMemsetAsync(b->state, 0, nb * sizeof(int32_t), b->stream));
    SMI_HIP(hipMemsetAsync(b->n_loss, 0, nb * sizeof(int32_t), b->stream));
    SMI_HIP(hipMemsetAsync(b->last_loss, 0, nb * sizeof(double), b->stream));
    SMI_HIP(hipMemsetAsync(b->have_prev, 0, nb * sizeof(int32_t), b->stream));
    return b->ks.stamp ? refresh_shifted_kernel(b, 0) : SMI_OK;
}

int smi_batch_enable_timing(smi_batch *b, int32_t on) {
    SMI_REQUIRE(b, "null batch");
    b->timing = on != 0;
    return SMI_OK;
}

int smi_batch_get_timing(smi_batch *b, double *ms_per_phase, int32_t n_phases) {
    SMI_REQUIRE(b && ms_per_phase, "null argument");
    for (int p = 0; p < n_phases && p < 6; ++p) ms_per_phase[p] = b->phase_ms[p];
    return SMI_OK;
}

// development aid (not in the public header): shader-clock stamps of the fused
// kernel's stages in workgroup 0 of the last launch
int smi_debug_fused_stamps(smi_batch *b, long long *out6) {
    SMI_REQUIRE(b && out6, "null argument");
    SMI_HIP(hipSetDevice(b->device));
    if (!b->dbg) {
        SMI_HIP(dev_alloc(&b->dbg, 16));
        SMI_HIP(hipMemset(b->dbg, 0, 16 * sizeof(long long)));
        std::memset(out6, 0, 16 * sizeof(long long));
        return SMI_OK;
    }
    SMI_HIP(hipStreamSynchronize(b->stream));
    SMI_HIP(hipMemcpy(out6, b->dbg, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return SMI_OK;
}

// development aid (not in the public header): shader clocks of n_rep monotonic sweeps per
// wavefront (mode 0: slot plan, 1: ring schedule), `waves` wavefronts per workgroup, and the
// swept images [groups * waves][h * w]
int smi_debug_sweep_cycles(smi_batch *b, int32_t plan_id, int32_t mode, int32_t n_rep,
                           float min_gradient, int32_t waves, int32_t groups, long long *cycles,
                           float *images) {
    SMI_REQUIRE(b && cycles && images, "null argument");
    SMI_REQUIRE(plan_id >= 0 && plan_id < (int)b->plans.size(), "no such plan");
    SMI_HIP(hipSetDevice(b->device));
    const SweepPlanDev &pl = b->plans[plan_id];
    const size_t nw = (size_t)waves * groups, n = (size_t)pl.h * pl.w;
    long long *d_cycles = nullptr;
    float *d_images = nullptr;
    SMI_HIP(dev_alloc(&d_cycles, nw));
    SMI_HIP(dev_alloc(&d_images, nw * n));
    int rc = launch_sweep_timing(b->d_plans, pl, plan_id, mode, n_rep, 1.f - min_gradient, waves,
                                 groups, d_cycles, d_images, b->stream);
    if (rc == SMI_OK) {
        hipError_t e = hipStreamSynchronize(b->stream);
        if (e == hipSuccess) e = hipMemcpy(cycles, d_cycles, nw * sizeof(long long), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(images, d_images, nw * n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_error(hipGetErrorString(e));
            rc = SMI_ERR_HIP;
        }
    }
    (void)hipFree(d_cycles);
    (void)hipFree(d_images);
    return rc;
}

int smi_batch_fft_shape(smi_batch *b, int32_t *fft_h, int32_t *fft_w) {
    SMI_REQUIRE(b, "null batch");
    if (fft_h) *fft_h = b->Fy;
    if (fft_w) *fft_w = b->Fx;
    return SMI_OK;
}

int64_t smi_observation_uploads(void) { return g_observation_uploads.load(); }

int smi_set_hw_queues(int32_t n) {
    return g_hw_queues.exchange(n > 0 ? n : 4, std::memory_order_relaxed);
}

int smi_batch_conv_path_used(smi_batch *b, int32_t *path) {
    SMI_REQUIRE(b && path, "null argument");
    *path = b->null_renderer ? 0 : b->fused ? 2 : 1;
    return SMI_OK;
}

}  // extern "C"
