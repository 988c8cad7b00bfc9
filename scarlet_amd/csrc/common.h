// Internal declarations shared by the translation units of libscarlet_amd.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/scarlet_amd.h"

namespace smi {

void set_error(const std::string &msg);

#define SMI_HIP(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            smi::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));     \
            return SMI_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define SMI_REQUIRE(cond, msg)                                                     \
    do {                                                                           \
        if (!(cond)) {                                                             \
            smi::set_error(std::string(msg) + " (" #cond ")");                     \
            return SMI_ERR_INVALID;                                                \
        }                                                                          \
    } while (0)

// Raise a kernel's dynamic-LDS limit to `lds` bytes on the CURRENT device.  Function
// attributes live per device, so one process driving several GPUs (fit_blends(devices=...))
// has to configure each of them; `configured` is the caller's static per-kernel table.
constexpr int kMaxDevices = 32;
inline int ensure_dynamic_lds(const void *kern, size_t lds, size_t (&configured)[kMaxDevices]) {
    int dev = 0;
    SMI_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    if (!tracked || lds > configured[dev]) {
        SMI_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (tracked) configured[dev] = lds;
    }
    return SMI_OK;
}

// ---------------------------------------------------------------------------
// Level-ordered plan of one monotonic operator (one box shape x weighting).
//
// The reference walks the pixels one after another in order of increasing
// radius and clips each against the weighted mean of its neighbours nearer the
// peak (operators_pybind11.cc:14-36).  Pixels whose inputs are all final can be
// processed together: the plan groups the sweep order into levels of the
// read/write dependency graph, so that a wavefront handles one level per step
// and still reproduces the sequential result bit for bit.
//
// Entry e (level major):  pix[e] target pixel, cnt[e] number of terms,
// nbr[j*E+e] / wt[j*E+e] the j-th term in ascending neighbour order.
// ---------------------------------------------------------------------------
struct SweepPlanHost {
    int32_t h = 0, w = 0;
    int32_t n_entries = 0;  // E
    int32_t max_terms = 0;  // D
    std::vector<int32_t> level_start;  // n_levels + 1
    std::vector<int32_t> pix;          // E
    std::vector<int32_t> cnt;          // E
    std::vector<int32_t> nbr;          // D * E
    std::vector<double> wt;            // D * E
};

// Build the plan from the tables the reference binds (operator.py:62-96).
// Returns false (with set_error) on malformed input.
bool build_sweep_plan(int32_t n_pix, const double *weights, const int32_t *offsets,
                      int32_t n_off, const int32_t *dist_idx, int32_t n_idx,
                      SweepPlanHost *out);

// One lane's work in one 64-wide step of the sweep (fast path: <= 4 terms, images of
// at most 16376 pixels): LDS byte addresses 16 + 4 * pixel, two per word -- p | n0 << 16,
// n1 | n2 << 16, n3 -- and the four weights.  Missing terms: own address, weight 0.  The
// pixels of a step sit in its first lanes; the other lanes are idle: all zeros, i.e. all
// five addresses = the spare cell at address 0 in front of the image, weights 0 (what a
// buffer load returns for an out-of-range offset: the sweep does not fetch idle entries).
// `lanes_ahead` = number of lanes in use three steps on, the same in every entry of a step.
struct alignas(16) SweepSlotEntry {
    uint32_t p_n0, n1_n2, n3, lanes_ahead;
    float w[4];
};

// Size classes of the register-resident update kernel: a box of N pixels runs in the
// instantiation with the smallest T * NPL >= N (thread t of the component's T threads owns
// pixels t, t + T, ...), so every box of a class has more than T * NPL_prev pixels.  One
// wavefront per component for the reference's standard boxes (21 + 10 k pixels a side,
// morphology.py / initialization.py:173-177: 21^2 .. 61^2 map onto one class each), four
// wavefronts for larger boxes up to 122^2 (the state of a 71^2 or 81^2 box does not fit the
// registers of one wavefront; the image in LDS and the sweep stay those of one wavefront).
constexpr int kNumSmallClasses = 5;
constexpr int kNumUpdateClasses = 9;
// up to this many components a range with several size classes is updated in one launch
// (update_kernel_mixed; measurements there)
constexpr int kMixedUpdateLimit = 3072;
// up to this many components of a class in a range, their workgroups are packed with as many
// wavefronts as a CU holds (update_kernel_reg; measurements there)
constexpr int kUpdatePackLimit = 1024;
constexpr int kUpdateNpl[kNumUpdateClasses] = {7, 16, 27, 42, 59, 16, 27, 42, 59};
constexpr int kUpdateTeam[kNumUpdateClasses] = {64, 64, 64, 64, 64, 256, 256, 256, 256};
constexpr int kMaxRegisterBox = 256 * 59;
inline int update_class(int n_pix) {
    for (int c = 0; c < kNumUpdateClasses; ++c)
        if (n_pix <= kUpdateTeam[c] * kUpdateNpl[c]) return c;
    return -1;
}
// LDS floats of one component's image in that kernel: the spare cell of the sweep, then all
// T * NPL pixel slots of the class (slots beyond the box hold zeros, so the loops need no
// bounds)
inline int update_image_stride(int n_pix) {
    const int c = update_class(n_pix);
    return kUpdateTeam[c] * kUpdateNpl[c] + 4;
}

// ---------------------------------------------------------------------------
// Ring schedule of the radial tables (operator.py:591-667), the plan the update kernels
// prefer.  With r = max(|Y|, |X|) the ring of a pixel around the peak and j = min(|Y|, |X|)
// its position along the ring inside its octant, the neighbours strictly nearer the peak are
//     A = (r-1, j-1)   B = (r-1, j)   C = (r-1, j+1) [j < r-1]   D = (r, j-1),
// so pixel (r, j) is final after level L = 2 r + j - 1: A is final at L - 3, B at L - 2, C and
// D at L - 1.  One lane per (octant, r mod 8 P) with P "planes" of 64 lanes: the lane walks
// along its ring, one pixel per level, then waits for ring r + 8 P.  Its operands are its own
// previous result (D) and the last three results of the lane of ring r - 1 (A, B, C), which
// arrive by DPP row rotations: the sweep reads the image only for the pixel itself.  Axis
// pixels (j = 0) take A from the octant across the axis, diagonal pixels (j = r) take B from
// the octant across the diagonal, and either octant computes them (same operands, same order).
// The sum runs in the order of the reference's neighbour table, a constant of the octant:
// A, B, C ascending or descending with D at position `pd` (3 bits per octant in `perm`).
//
// Lane layout: lane = row * 16 + half * 8 + (r mod 8); rows = octant pairs that share an axis
// (+x, +y, -x, -y), halves = sign along the minor axis ((-,+), (+,-), (+,-), (-,+)).
//
// Device stream, `n_pad` + kRingAhead steps (n_pad = steps rounded up to the unrolling of the
// loop, idle entries behind).  The radial tables of a centred odd square box are the same in
// all eight octants once the weights are sorted by role (checked bit for bit by
// ring_device_stream, which refuses anything else), so a step stores the float4 weights
// (A, B, C, D) once per ring: [step][plane][r mod 8] float4, 128 P bytes per step -- the eight
// lanes of a ring read one address, a broadcast -- followed by the uint16 LDS byte addresses
// [step][plane][lane] 16 + 4 * pixel (0 = idle: the spare cell in front of the image), another
// 128 P bytes per step: 17 KB for a 41^2 box, 49 KB for 61^2.  With two planes a lane handles
// the rings r = m + 8 (mod 16) and r = m (mod 16) in the same step; the lane below ring
// r = 8 k is m = 7 of the other plane.
// ---------------------------------------------------------------------------
constexpr int kRingUnroll = 6;   // steps per loop iteration (period of the axis / diagonal phases)
constexpr int kRingAhead = 6;    // steps the address stream is requested ahead
constexpr int kRingMaxPlanes = 2;
constexpr uint16_t kRingLate = 0x8000, kRingAxis = 1, kRingDiag = 2;  // flags of an address word
constexpr uint16_t kRingAddrMask = 0x7FFC;
// Rings beyond 23 on ONE plane (round 5).  Ring r + 8 needs the lane of ring r, which walks its
// r + 1 pixels during the levels 2 r - 1 .. 3 r - 1: from ring 24 on the next ring of a lane is
// due before the lane is free.  Such a ring simply starts late: level S(r) = max(2 r - 1,
// S(r - 1) + 2, S(r - 8) + r - 7), pixel (r, j) at level S(r) + j.  Its operands A, B, C are
// then lam(r) = S(r) - S(r - 1) - 2 levels older than usual (lam <= 1 up to ring 31: boxes up to
// 63^2, 96 instead of 89 steps for 61^2), which the lane program covers with one more entry of
// DPP history.  The stream says which entries are late, which are the axis pixel (j = 0) and
// the diagonal pixel (j = r) of their ring -- the level no longer does -- in bits 15, 0 and 1 of
// the address word, from step `n_nat` on (a multiple of the unrolling, at least four steps
// before the first late ring starts); the steps before it run the plain schedule.
struct RingPlanHost {
    int32_t planes = 0;   // 1: rings up to 31 (late starts from ring 24 on), 2: up to 47
    int32_t n_steps = 0;  // 3 rmax - 1, or S(rmax) + rmax with late rings
    int32_t n_nat = 0;    // steps of the plain schedule (== n_pad without late rings)
    int32_t n_pad = 0;
    int32_t rmax = 0;
    int32_t centre = 0;   // flat index of the peak
    uint32_t perm = 0;    // per octant (row * 2 + half): bit 0 = A, B, C ascending, bits 1-2 = pd
    std::vector<float> wts;      // [n_pad + kRingAhead][planes][64][4]
    std::vector<uint16_t> addr;  // [n_pad + kRingAhead][planes][64]
};
// false (no error set) when the tables do not have the radial structure or the box is too large
bool build_ring_plan(int32_t h, int32_t w, const double *weights, const int32_t *offsets,
                     int32_t n_off, const int32_t *dist_idx, int32_t n_idx, RingPlanHost *out);
// the stream the kernels read (above); false when the weights differ between octants or the
// octants do not hold the same pixels (off-centre peak, even or oblong box)
bool ring_device_stream(const RingPlanHost &rp, std::vector<uint8_t> *out);

struct SweepPlanDev {
    int32_t h = 0, w = 0, n_entries = 0, max_terms = 0, n_levels = 0;
    int32_t n_slots = 0;
    // ring schedule (nullptr: the tables are not radial, or rings beyond 8 * kRingMaxPlanes - 1)
    const void *ring = nullptr;
    int32_t ring_planes = 0, ring_pad = 0, ring_rmax = 0, ring_centre = 0, ring_nat = 0;
    uint32_t ring_perm = 0, ring_bytes = 0;
    SweepSlotEntry *slots = nullptr;  // nullptr when the plan does not fit the fast path
    int32_t *level_start = nullptr;
    int32_t *pix = nullptr;
    int32_t *cnt = nullptr;
    int32_t *nbr = nullptr;
    float *wt = nullptr;
};

// ---------------------------------------------------------------------------
// kernel launchers (kernels.hip)
// ---------------------------------------------------------------------------

struct BatchView {
    int32_t nb, C, H, W, Fy, Fx;  // Fy=H, Fx=W for the NullRenderer
    int32_t n_comp;
    int32_t n_comp_total;  // components of the whole batch (n_comp: of this range of blends)
    // components (device, SoA)
    const int32_t *comp_start;  // nb + 1
    const int32_t *c_blend, *c_oy, *c_ox, *c_h, *c_w, *c_flags, *c_plan;
    const int64_t *c_moff;  // n_comp + 1 offsets into the packed morph arrays
    const float *c_sed_min_step, *c_sed_rel, *c_morph_step, *c_morph_rel;
    const float *c_min_grad, *c_lthresh;
    float *sed, *morph;  // parameters
    float *m_sed, *v_sed, *vh_sed, *m_morph, *v_morph, *vh_morph;
    // observation
    const float *data, *weights;
    // data and weights of a pair of rows side by side, [nb][C][(H + 1) / 2][W] x {data[2j][x],
    // data[2j+1][x], weights[2j][x], weights[2j+1][x]} (zeros for row H of an odd frame): what the
    // fused convolution kernel reads -- one 16-byte load per pixel pair instead of four 4-byte
    // ones, and the operands of its packed arithmetic arrive as pairs (a copy made when the
    // observation is registered; nullptr on the rocFFT path)
    const float4 *dw;
    const double *log_norm;  // nb
    // per blend state
    int32_t *state;       // 0 active, 1 last iteration, 2 done, 3 non-finite
    int32_t *n_loss;      // losses recorded
    double *loss_hist;    // nb * hist_cap
    int32_t hist_cap;
    double *last_loss;    // nb
    const int32_t *have_prev;  // nb: last_loss was seeded by smi_batch_set_previous_loss
    double *loss_partial; // nb * n_partial
    int32_t n_partial;
    // sweep plans
    const SweepPlanDev *plans;  // device array
    int32_t max_box_pixels;
    int32_t max_levels;  // over all plans
    int32_t fast_plans;  // every plan has the slot layout
    int32_t mono_mask;   // some component carries SMI_PROX_MONO_MASK (general update kernel)
    float b1, b2, eps;   // AMSGrad constants (defaults 0.9, 0.999, 1e-8: lite/parameters.py:194)
    // point sources: per component {offset y, x, m y, x, v y, x, vhat y, x} with
    // offset = centre - mean(box bounds) (morphology.py:503-507), and the PSF sigma
    int32_t n_point;
    double *pt;
    const float *c_sigma;
    const float *c_beta;  // > 0: Moffat profile with alpha = c_sigma (point_source_kernel)
    // free Fourier shifts (shift.hip): pt holds {shift y, x, m, v, vhat}; `morph` is the
    // shifted image the model uses, `morph_param` the image parameter; the update
    // kernels read the pulled-back gradient from g_morph_buf / g_sed_buf
    int32_t n_shift, max_box_side;
    // boxes beyond the LDS (more than ~100 x 100 pixels): the four image-sized work arrays of
    // the shift kernels in global memory, 4 N floats per component at 4 c_moff[k] + 16 k
    float *shift_scratch;
    float *morph_param;
    float *g_sed_buf, *g_morph_buf;
    // register-resident update kernels: the pre-prox image x and the AMSGrad denominator psi of
    // every pixel, written once per iteration and read back once per proximal sub-iteration
    // (x at [0, n_morph), psi at [n_morph, 2 n_morph); L2-resident between the two)
    float *xp_tmp;
    int64_t n_morph_total;
    const float *c_shift_step;
    const float *c_shift_rel;  // relative_step factor of a free shift (0: constant step)
    const int32_t *c_shift_fft;  // (Fy, Fx) per component, fft.py:116-167 with padding 10
    // scarlet.lite: centre floor, background threshold levels [n_comp][C], FISTA
    const float *c_center_floor;
    const float *c_sym_strength;  // SymmetryConstraint(strength)
    const float *c_pos_floor;  // PositivityConstraint(zero) of the morphology
    const int32_t *c_chain_repeat;  // ConstraintChain(repeat); nullptr when every chain runs once
    const float *c_bg_level;
    float *scratch;              // 3 * n_morph floats when the largest box exceeds the LDS
    int32_t lite;                // FISTA, or a component with FIT_CENTER / BG_THRESH
    int32_t scheme;              // SMI_SCHEME_*; FISTA keeps z in m_sed / m_morph
    const float *c_fista_step;
    double *fista_t;             // [n_comp][2]
    // further observations of blend 0 (resample.hip): log_norm + chi^2 / 2 of each
    const double *extra_term;
    int32_t n_extra;
    // sub-range launches (one stream per range of blends): the grids cover `nb` blends /
    // `n_comp` components starting at these offsets; all arrays stay whole-batch
    int32_t blend0, comp0;
    // work list of the register-resident update kernels: the component of every workgroup,
    // sorted by size class, then blend (point sources have their own kernel and are not
    // listed).  `work_start` is a HOST array [kNumUpdateClasses][nb_total + 1]: first
    // entry of every blend within a class (prefix sums), for the launch ranges.
    const int32_t *work;
    int32_t work0;
    // fused_conv_kernel without a model cube (model == nullptr) renders its rows itself:
    // columns of one residue class (mod 16) a box can hold, (widest box + 14) / 16 + 1; 0 =
    // boxes too wide for that (the cube comes from render_kernel)
    int32_t render_slots;
    const int32_t *work_start;
    int32_t nb_total;
    // register-resident update kernels with the ring plan staged in LDS (update_kernel_reg):
    // per size class the plan most of its components use (-1: none has a ring schedule) and
    // the bytes of its stream
    int32_t stage_plan[kNumUpdateClasses];
    uint32_t stage_bytes[kNumUpdateClasses];
    // value a failing update stores in state[b]: 3 + the iteration (launch_update stamps it;
    // anything >= 3 means non-finite parameters, finalize_blend explains the iteration)
    int32_t fail_code = 3;
    // which of the concurrent ranges of blends this view is (step_sub_ranges): the launches per
    // size class of a range fork onto side streams of that range (launch_update)
    int32_t range_slot = 0;
    // several size classes in a batch too large for update_kernel_mixed, eight hardware queues:
    // two ranges whose class launches run side by side (batch.hip: sub_ranges)
    int32_t class_streams = 0;
    // per blend: the iteration counter at which its current adaprox call began (nullptr: 0 for
    // all).  A blend whose boxes were resized starts anew (blend.py:276-302) while its batch
    // mates go on: the kernels take the rules of a first step (alpha / 10, vhat = v) and
    // min_iter from `it - it_base[b]` (smi_batch_set_iteration_base).
    // smi_batch_set_pause_at: the iteration after whose update blend b pauses; conv_flag[b] = 1
    // when the stopping rule fired (finalize_blend)
    const int32_t *pause_at = nullptr;
    int32_t *conv_flag = nullptr;
    const int32_t *it_base = nullptr;
    __device__ __forceinline__ int local_it(int b, int it) const {
        return it_base ? it - it_base[b] : it;
    }
};

void launch_render(const BatchView &v, float *P, hipStream_t s);
void launch_residual(const BatchView &v, const float *Q, float *R, hipStream_t s);
void launch_finalize(const BatchView &v, int32_t it, float e_rel, int32_t min_iter,
                     int32_t check, hipStream_t s);
void launch_advance(const BatchView &v, hipStream_t s);
void launch_cmul(float2 *S, const float2 *K, int32_t nb, int32_t C, int64_t plane,
                 int32_t k_bands, int32_t k_per_blend, int32_t conj,
                 const int32_t *state, hipStream_t s);
int launch_update(const BatchView &v, const float *G, int32_t it, float e_rel,
                  int32_t prox_max_iter, float *g_sed_out, float *g_morph_out,
                  int32_t grad_only, hipStream_t s);
int launch_update_finalize(const BatchView &v, const float *G, int32_t it, float e_rel,
                           int32_t min_iter, int32_t check, int32_t prox_max_iter,
                           hipStream_t s);
// mode 0: one optimizer step of every point source (spectrum + centre), 1: gradients only
// (g_sed_out, g_center_out[n_comp][2]), 2: evaluate the morphologies from the centres
int launch_point_sources(const BatchView &v, const float *G, int32_t it, float e_rel,
                         int32_t prox_max_iter, float *g_sed_out, double *g_center_out,
                         int32_t mode, hipStream_t s);
bool shift_needs_scratch(int max_box_pixels, int max_box_side);
int launch_shift_backward(const BatchView &v, const float *G, int32_t it, double *g_shift_out,
                          int32_t grad_only, hipStream_t s);
int launch_shift_forward(const BatchView &v, int32_t respect_state, hipStream_t s);

// ConvolutionRenderer(psf_shift=...) (renderer.py:175-177, 215-228): the difference kernel
// carries a free sub-pixel Fourier shift, one per kernel set (the batch's kernel, or one
// per blend), shared by the bands of the set
struct KernelShiftView {
    int32_t n_sets, bands, per_blend;
    int32_t h0, w0;    // stamp as the renderer holds it
    int32_t ph, pw;    // device stamp (odd sides), the renderer's sits at (oy, ox)
    int32_t oy, ox;
    int32_t Fy, Fx;    // FFT lengths of fft.shift for an (h0, w0) image (fft.py:116-167, padding 10)
    int32_t n_part;    // partial sums per stamp pixel of the kernel gradient
    int32_t slab;      // frame rows per partial sum
    double step, rel;  // step = max(step, rel * mean(shift)) (parameter.py:126-129)
    const float *stamp;  // [n_sets bands][h0][w0] unshifted
    float *shifted;      // [n_sets bands][ph][pw] kernel at the current shift
    double *partial;     // [n_sets bands][n_part][h0 w0]
    double *state;       // [n_sets][10]: shift, m, v, vhat, last gradient, (y, x) each
};
// R: rendered cube [nb][C][H][W] on entry, w (rendered - data) on return; M: the model
// cube.  Gradient of -logL w.r.t. the shift into state[8..9]; unless grad_only the shift
// takes its AMSGrad step (the kernel itself is refreshed by launch_psf_shift_forward).
int launch_psf_shift_backward(const BatchView &v, const KernelShiftView &ks, float *R,
                              const float *M, int32_t it, int32_t grad_only, hipStream_t s);
int launch_psf_shift_forward(const BatchView &v, const KernelShiftView &ks, int32_t respect_state,
                             hipStream_t s);
// seam 1, monotonic mask operators (mask.hip)
template <typename T>
int mask_valid_host_buffers(int32_t i, int32_t j, const T *image, int32_t rows, int32_t cols,
                            uint8_t *unchecked, uint8_t *orphans, double variance,
                            int32_t *bounds, double thresh);
template <typename T>
int mask_interpolate_host_buffers(const int32_t *row_idx, const int32_t *col_idx, int32_t n_idx,
                                  uint8_t *unchecked, T *model, int32_t rows, int32_t cols,
                                  uint8_t *orphans, double variance, int32_t recursive,
                                  int32_t *bounds);
// multi-resolution rendering (resample.hip)
struct Resampler;
int resampler_create(const float *A, const float *Pt, int C, int n_a, int n_b, int Fy, int Fx,
                     Resampler **out);
int resampler_render(Resampler *r, const float *model, float *out);
int resampler_time(Resampler *r, int n_rep, double *ms_per_render);
int resampler_get_path(const Resampler *r);
int resampler_set_path(Resampler *r, int path);
void resampler_destroy(Resampler *r);
struct LowRes;
int lowres_create(Resampler *r, const int32_t *channels, const float *data, const float *weights,
                  double log_norm, int H, int W, double *term_slot, LowRes **out);
void lowres_destroy(LowRes *l);
int lowres_evaluate(LowRes *l, const float *P, int Py, int Px, int backward, hipStream_t s);
void lowres_add_gradient(LowRes *l, float *Q, int Py, int Px, hipStream_t s);
int lowres_get_rendered(LowRes *l, float *out, hipStream_t s);
int launch_sweep_timing(const SweepPlanDev *d_plans, const SweepPlanDev &host_plan, int plan_id,
                        int mode, int n_rep, float one_minus_g, int waves, int groups,
                        long long *cycles, float *images, hipStream_t s);
// box resizing (kernels.hip: resize_test_kernel and the state records)
void launch_resize_test(const BatchView &v, int32_t *margin, double *pull, hipStream_t s);
void launch_gather_states(const BatchView &v, const int32_t *sel, const int64_t *off, int32_t n_sel,
                          float *staging, hipStream_t s);
void launch_scatter_states(const BatchView &v, const int64_t *off, const float *staging,
                           hipStream_t s);
void launch_carry_states(const int32_t *keep, const int64_t *old_moff, const int64_t *new_moff,
                         int32_t n, float *const from[4], float *const to[4], hipStream_t s);
void launch_interleave_obs(const float *data, const float *weights, float4 *dw, int64_t planes,
                           int32_t H, int32_t W, hipStream_t s);
void launch_log_norm(const float *weights, double *log_norm, int32_t nb, int64_t n,
                     hipStream_t s);
void launch_wrap_kernel(const float *kern, float *out, int32_t n_img, int32_t ph,
                        int32_t pw, int32_t Fy, int32_t Fx, float scale, hipStream_t s);
void launch_crop(const float *P, float *out, int32_t n_img, int32_t H, int32_t W,
                 int32_t Fy, int32_t Fx, hipStream_t s);
void launch_count_active(const int32_t *state, int32_t nb, int32_t *out, hipStream_t s);

// fused LDS-resident convolution (fused_conv.hip)
bool fused_conv_supported(int Fy, int Fx);
bool fused_conv_instantiated(int Fy, int Fx);
bool fused_conv_choose(int ny, int nx, int *Fy, int *Fx);
int launch_fused_conv(const BatchView &v, int Fy, int Fx, const float *model, const float2 *Kt,
                      int k_bands, int k_per_blend, float *out, int mode, long long *dbg,
                      hipStream_t s);
int launch_fused_conv_short(const BatchView &v, int Fy, int Fx, const float *model,
                            const float2 *Kt, int k_bands, int k_per_blend, float *out, int mode,
                            long long *dbg, hipStream_t s);
int launch_stamp_spectrum(const float *d_kern, double2 *d_tmp, float2 *Kt, int n_img, int ph,
                          int pw, int Fy, int Fx, double scale, hipStream_t s);
int launch_permute_kernel_spectrum(const float2 *Khat, float2 *Kt, int n_img, int Fy, int Fx,
                                   float scale, hipStream_t s);

template <typename T>
int sweep_host_buffers(T *flat_img, int32_t n_pix, const SweepPlanHost &plan, T min_gradient);
// n_img images of n_pix pixels, one plan each, one launch (smi_prox_weighted_monotonic_many_*)
template <typename T>
int sweep_many_host_buffers(T *images, int32_t n_img, int32_t n_pix,
                            const std::vector<SweepPlanHost> &plans, T min_gradient);
template <typename T>
int apply_filter_host_buffers(const T *image, int32_t H, int32_t W, const T *values,
                              int32_t n_taps, const int32_t *ys, const int32_t *ye,
                              const int32_t *xs, const int32_t *xe, T *result);

}  // namespace smi
