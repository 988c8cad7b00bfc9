/*
 * scarlet_amd.h -- C ABI of libscarlet_amd.so, the MI355X (gfx950) back end for
 * the proximal-gradient fitting loop of pmelchior/scarlet.
 *
 * Plain C: pointers, sizes, POD structs.  All functions return 0 on success or a
 * negative smi_status; smi_last_error() gives the message of the last failure on
 * the calling thread.  Host pointers unless a parameter is named d_* (device).
 * Handles are not thread-safe; one smi_batch lives on one GPU.
 *
 * The entry points replace these reference interfaces (paths relative to the
 * reference checkout):
 *
 *   seam 1  scarlet/operators_pybind11.cc:234-260  (pybind11 module
 *           `scarlet.operators_pybind11`, called from scarlet/operator.py:54-58
 *           and scarlet/renderer.py:108-116)
 *             prox_weighted_monotonic  -> smi_prox_weighted_monotonic_{f32,f64}
 *             apply_filter             -> smi_apply_filter_{f32,f64}
 *
 *   seam 2  the optimizer call in scarlet/blend.py:165-180
 *           (proxmin.adaprox(X, grad, step, prox=..., scheme="amsgrad",
 *           prox_max_iter=10, M=, V=, Vhat=) together with the closures it is
 *           given: Blend._loss_func blend.py:259-274, Blend.get_model
 *           blend.py:200-244, Observation.get_log_likelihood
 *           observation.py:147-170, ConvolutionRenderer renderer.py:247-259,
 *           ConstraintChain constraint.py:76-80)
 *             -> smi_batch_* : the same loop for a batch of independent blends,
 *                resident on the device.
 */
#ifndef SCARLET_AMD_H
#define SCARLET_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum smi_status {
    SMI_OK = 0,
    SMI_ERR_INVALID = -1,   /* bad argument (shape, NULL, out of range)            */
    SMI_ERR_HIP = -2,       /* HIP runtime / rocFFT failure                        */
    SMI_ERR_NO_DEVICE = -3, /* no usable GPU                                       */
    SMI_ERR_ARITHMETIC = -4 /* a parameter became non-finite (model.py:153-165)    */
} smi_status;

const char *smi_last_error(void);
int smi_device_count(void);
/* "scarlet_amd <version> gfx950" */
const char *smi_version(void);

/* ------------------------------------------------------------------------- *
 * Seam 1: operators_pybind11 drop-ins.  Caller owns all buffers; flat_img and
 * result are modified in place, like the Eigen::Ref arguments of the reference.
 * ------------------------------------------------------------------------- */

/* operators_pybind11.cc:14-36.  flat_img[n_pix]; weights[n_off][n_pix] row major;
 * offsets[n_off]; dist_idx[n_idx] = sweep order (peak excluded).  Sequential
 * (Gauss-Seidel) semantics are preserved exactly: the device kernel walks the
 * levels of the dependency graph, which yields bit-identical results. */
int smi_prox_weighted_monotonic_f32(float *flat_img, const float *weights,
                                    const int32_t *offsets, int32_t n_off,
                                    const int32_t *dist_idx, int32_t n_idx,
                                    int32_t n_pix, float min_gradient);
int smi_prox_weighted_monotonic_f64(double *flat_img, const double *weights,
                                    const int32_t *offsets, int32_t n_off,
                                    const int32_t *dist_idx, int32_t n_idx,
                                    int32_t n_pix, double min_gradient);

/* The same sweep for n_img independent images of n_pix pixels, every one with tables of its
 * own, in ONE launch (one workgroup per image): what the initialisation of a scene needs --
 * each source's detection image made monotonic about that source's centre (source.py:312-333
 * calls operators_pybind11.cc:14-36 once per source; initialization.py:287-363 loops over the
 * sources).  images[n_img][n_pix] in place; weights[n_img][n_off][n_pix]; dist_idx[n_img][n_idx];
 * offsets[n_off] shared.  Image i comes out bit for bit as smi_prox_weighted_monotonic_*(
 * images[i], weights[i], offsets, n_off, dist_idx[i], n_idx, n_pix, min_gradient) leaves it. */
int smi_prox_weighted_monotonic_many_f32(int32_t n_img, float *images, int32_t n_pix,
                                         const float *weights, const int32_t *offsets,
                                         int32_t n_off, const int32_t *dist_idx, int32_t n_idx,
                                         float min_gradient);
int smi_prox_weighted_monotonic_many_f64(int32_t n_img, double *images, int32_t n_pix,
                                         const double *weights, const int32_t *offsets,
                                         int32_t n_off, const int32_t *dist_idx, int32_t n_idx,
                                         double min_gradient);

/* operators_pybind11.cc:39-56.  image[H][W], result[H][W]; taps given as in the
 * reference: values[n_taps] and the four slice-bound vectors. */
int smi_apply_filter_f32(const float *image, int32_t H, int32_t W,
                         const float *values, int32_t n_taps,
                         const int32_t *y_start, const int32_t *y_end,
                         const int32_t *x_start, const int32_t *x_end,
                         float *result);
int smi_apply_filter_f64(const double *image, int32_t H, int32_t W,
                         const double *values, int32_t n_taps,
                         const int32_t *y_start, const int32_t *y_end,
                         const int32_t *x_start, const int32_t *x_end,
                         double *result);

/* get_valid_monotonic_pixels / linear_interpolate_invalid_pixels
 * (operators_pybind11.cc:61-232, float32 and float64 overload sets; callers
 * operator.py:155-176).  Row-major (rows, cols) images; `unchecked` / `orphans` are the
 * NumPy bool maps as bytes; `bounds` = (min row, max row, min col, max col); all updated
 * in place like the Eigen::Ref arguments of the reference.  Bit-identical maps and
 * values to the C++ recursion (tests: test_mask_operators_bit_exact). */
int smi_get_valid_monotonic_pixels_f32(int32_t i, int32_t j, const float *image, int32_t rows,
                                       int32_t cols, uint8_t *unchecked, uint8_t *orphans,
                                       double variance, int32_t *bounds, double thresh);
int smi_get_valid_monotonic_pixels_f64(int32_t i, int32_t j, const double *image, int32_t rows,
                                       int32_t cols, uint8_t *unchecked, uint8_t *orphans,
                                       double variance, int32_t *bounds, double thresh);
int smi_linear_interpolate_invalid_pixels_f32(const int32_t *row_indices,
                                              const int32_t *column_indices, int32_t n_indices,
                                              uint8_t *unchecked, float *model, int32_t rows,
                                              int32_t cols, uint8_t *orphans, double variance,
                                              int32_t recursive, int32_t *bounds);
int smi_linear_interpolate_invalid_pixels_f64(const int32_t *row_indices,
                                              const int32_t *column_indices, int32_t n_indices,
                                              uint8_t *unchecked, double *model, int32_t rows,
                                              int32_t cols, uint8_t *orphans, double variance,
                                              int32_t recursive, int32_t *bounds);

/* ------------------------------------------------------------------------- *
 * Seam 2: batched proximal-gradient fit.
 * ------------------------------------------------------------------------- */

typedef struct smi_batch smi_batch;

/* prox chain of a morphology (morphology.py:644-670): bit flags */
enum {
    SMI_PROX_MONOTONIC = 1,  /* MonotonicityConstraint (constraint.py:183-234)  */
    SMI_PROX_SYMMETRY = 2,   /* SymmetryConstraint (constraint.py:262-273)      */
    SMI_PROX_POSITIVE = 4,   /* PositivityConstraint (constraint.py:83-92)      */
    SMI_PROX_CENTER_ON = 8,  /* CenterOnConstraint (constraint.py:276-287)      */
    SMI_PROX_NORM_MAX = 16,  /* NormalizationConstraint("max") (95-114)         */
    SMI_PROX_NORM_SUM = 32,  /* NormalizationConstraint("sum")                  */
    SMI_PROX_L1 = 64,        /* L1Constraint -> proxmin prox_soft (134-145)     */
    SMI_PROX_L0 = 128,       /* L0Constraint -> proxmin prox_hard (117-130)     */
    /* MonotonicityConstraint(fit_center_radius=1) (constraint.py:203-208,
     * operator.py:99-129): the sweep starts at the brightest pixel of the 3x3 block
     * around the box centre; `sweep_plan` is the first of 9 consecutive plans built
     * for the centres (cy-1..cy+1) x (cx-1..cx+1), row major */
    SMI_PROX_FIT_CENTER = 256,
    /* scarlet.lite background threshold (lite/models.py:222-228): a pixel is set to 0
     * when sed[c] * morph < bg_level[c] in every band (new sed); replaces positivity */
    SMI_PROX_BG_THRESH = 512,
    /* l_thresh of SMI_PROX_L1 / L0 is relative: multiplied by the step of the proximal
     * sub-iteration, gamma = alpha / max(psi) (type="relative", the reference's default,
     * constraint.py:117-145; lite/parameters.py:296-299) */
    SMI_PROX_L_RELATIVE = 1024,
    /* MonotonicityConstraint(use_mask=True) (constraint.py:228-232): pixels that a
     * strictly decreasing, positive 4-neighbour path connects to the centre
     * (get_valid_monotonic_pixels, operators_pybind11.cc:61-125, variance 0) keep their
     * value from before the sweep; needs SMI_PROX_MONOTONIC */
    SMI_PROX_MONO_MASK = 2048,
    /* not a constraint: the component is a PointSource (source.py:92-128) whose
     * morphology is the model PSF evaluated at a free sub-pixel centre
     * (PointSourceMorphology, morphology.py:476-513; GaussianPSF, psf.py:80-142) */
    SMI_COMPONENT_POINT_SOURCE = 1 << 16,
    /* the image morphology is moved by a free sub-pixel Fourier shift
     * (ExtendedSource(shifting=True): morphology.py:124-130, 673-676; fft.shift,
     * fft.py:399-428): `center` holds the initial shift (y, x), `shift_step` its step */
    SMI_COMPONENT_SHIFTING = 1 << 17,
    /* Parameter(fixed=True) (parameter.py:38-39, blend.py:107-115): the parameter stays in
     * X but its gradient is taken as zero; the step (nothing) and the proximal operator
     * are still applied, as proxmin does with the expanded zero gradient.  FIXED_MORPH
     * fixes the image of an extended source or the centre of a point source. */
    SMI_COMPONENT_FIXED_SED = 1 << 18,
    SMI_COMPONENT_FIXED_MORPH = 1 << 19
};
#define SMI_PROX_EXTENDED_SOURCE \
    (SMI_PROX_MONOTONIC | SMI_PROX_POSITIVE | SMI_PROX_CENTER_ON | SMI_PROX_NORM_MAX)

typedef struct smi_batch_desc {
    int32_t n_blends;     /* independent scenes in the batch                       */
    int32_t C, H, W;      /* model frame = observed frame (bands, rows, columns)   */
    int32_t n_components; /* total over all blends                                 */
    int32_t kernel_h;     /* difference-kernel stamp; 0 => NullRenderer            */
    int32_t kernel_w;
    int32_t kernel_bands; /* 1 (band shared) or C                                  */
    int32_t kernel_per_blend; /* 0: one kernel set for the whole batch, 1: per blend */
    int32_t fft_h;        /* FFT shape; 0 => reference rule fft.py:116-167         */
    int32_t fft_w;
    int32_t max_iter;     /* capacity of the per-blend loss history                */
    int32_t conv_path;    /* 0 auto: LDS-resident fused convolution kernel when the  */
                          /*   padded band fits the LDS, else rocFFT;                */
                          /* 1 rocFFT pipeline (reference FFT shape by default);     */
                          /* 2 fused kernel or fail                                  */
} smi_batch_desc;

/* per component, all arrays of length n_components unless noted */
typedef struct smi_components {
    const int32_t *blend;       /* owning blend, non-decreasing                     */
    const int32_t *origin_y;    /* box origin in frame pixels (may be negative)     */
    const int32_t *origin_x;
    const int32_t *box_h;
    const int32_t *box_w;
    const float *sed;           /* [n_components][C]                                */
    const float *morph;         /* boxes packed back to back, row major             */
    const float *sed_min_step;  /* [n_components][C]  (spectrum.py:56)              */
    const float *sed_rel_step;  /* relative_step factor, 1e-2 in the reference      */
    const float *morph_step;    /* constant step, 1e-2 (morphology.py:670)          */
    const int32_t *prox_flags;  /* SMI_PROX_* of the morphology                     */
    const int32_t *sweep_plan;  /* index into the plans added with                  */
                                /* smi_batch_add_sweep_plan, -1 if not monotonic    */
    const float *min_gradient;  /* MonotonicityConstraint.min_gradient              */
    const float *l_thresh;      /* threshold for SMI_PROX_L1/L0 (absolute unless     */
                                /* SMI_PROX_L_RELATIVE)                              */
    const float *morph_rel_step;/* step = max(morph_step, rel * mean(morph)); NULL=0 */
                                /* (relative_step, parameter.py:126-129)            */
    /* point sources (SMI_COMPONENT_POINT_SOURCE in prox_flags); both NULL if none.
     * For such a component the box is the PSF box at the rounded initial centre
     * (morphology.py:494-497), its `morph` input is ignored (the library evaluates
     * the pixel-integrated Gaussian), and `morph_step` is the step of the centre
     * (3e-2, source.py:115). */
    const double *center;       /* [n_components][2] (y, x) in frame pixels; for     */
                                /* SMI_COMPONENT_SHIFTING entries: the shift (y, x)  */
    const float *psf_sigma;     /* [n_components] model PSF sigma (all bands alike)  */
    const float *shift_step;    /* [n_components] step of a free shift (1e-1,        */
                                /* morphology.py:675); NULL = 1e-1                   */
    const float *center_floor;  /* [n_components] floor of the centre pixel for      */
                                /* SMI_PROX_CENTER_ON: 1e-6 (CenterOnConstraint,     */
                                /* constraint.py:276-287) or the `floor` of a lite   */
                                /* component (lite/models.py:233-236); NULL = 1e-6   */
    const float *bg_level;      /* [n_components][C] bg_rms * bg_thresh for          */
                                /* SMI_PROX_BG_THRESH; NULL if unused                */
    const float *fista_step;    /* [n_components] FistaParameter.step                */
                                /* (lite/initialization.py:308-312); NULL unless the */
                                /* batch runs SMI_SCHEME_FISTA                       */
    const float *sym_strength;  /* [n_components] strength of SMI_PROX_SYMMETRY         */
                                /* (SymmetryConstraint(strength), constraint.py:262-273,*/
                                /* operator.py:274-293); NULL = 1                        */
    const float *pos_floor;     /* [n_components] PositivityConstraint(zero) of the       */
                                /* morphology (constraint.py:83-92); NULL = 0              */
    const int32_t *chain_repeat;/* [n_components] ConstraintChain(repeat) (constraint.py:  */
                                /* 60-80): the whole chain applied that many times per     */
                                /* proximal evaluation; NULL = 1                           */
    const float *shift_rel_step;/* [n_components] relative_step of a free shift             */
                                /* (parameter.py:126-129): step = max(shift_step, rel *     */
                                /* mean(shift)); NULL = 0.  (The centre of a point source   */
                                /* takes its rule from morph_step / morph_rel_step: step =  */
                                /* max(morph_step, rel * mean(centre in frame pixels)).)    */
    const float *psf_beta;      /* [n_components] point sources on a MoffatPSF model PSF     */
                                /* (psf.py:145-202): (1 + r^2 / alpha^2)^-beta sampled at    */
                                /* the pixel centres, alpha in psf_sigma; 0 or NULL = the    */
                                /* pixel-integrated Gaussian of width psf_sigma              */
} smi_components;

int smi_batch_create(const smi_batch_desc *desc, int device, smi_batch **out);
int smi_batch_destroy(smi_batch *b);

/* Monotonic operator tables exactly as operator.prox_weighted_monotonic binds them
 * (operator.py:62-96): weights[8][h*w] (float64), offsets[8], dist_idx[n_idx].
 * Returns the plan index (>= 0) or a negative status. */
int smi_batch_add_sweep_plan(smi_batch *b, int32_t h, int32_t w,
                             const double *weights, const int32_t *offsets,
                             const int32_t *dist_idx, int32_t n_idx);

/* The ring schedule the update kernels derive from such tables when they are the radial ones
 * of operator.py:591-667 (host only, no GPU; csrc/common.h describes the schedule).  Returns
 * 1 and fills info = {planes | n_nat << 8, n_steps, n_pad, rmax, centre, perm, lanes,
 * stream_bytes} (n_nat: steps of the plain schedule; from there on the address words carry
 * flags: 0x8000 late ring, 1 axis pixel, 2 diagonal pixel) when the tables qualify, 0 when they do not (the kernels then keep the level plan), < 0 on bad
 * arguments.  stream_bytes = size of the device stream with the weights stored once per ring,
 * 0 when the octants differ (off-centre peak, even or oblong box): only tables with such a
 * stream run the ring schedule on the device.  With capacity >= lanes also wts[lanes][4]
 * (weights by role A, B, C, D) and addr[lanes] (16 + 4 * pixel | flags, 0 = idle);
 * lanes = (n_pad + 6) * planes * 64, step major. */
int smi_sweep_ring_plan(int32_t h, int32_t w, const double *weights, const int32_t *offsets,
                        const int32_t *dist_idx, int32_t n_idx, int32_t info[8], float *wts,
                        uint16_t *addr, int64_t capacity);

/* data, weights: [n_blends][C][H][W] float32 (observation.py:52-57).  The _device
 * form adopts device buffers (e.g. torch tensors) without copying them: log_norm, the loss of
 * the rocFFT path and further renderers read the caller's buffers, which must stay alive and
 * UNCHANGED while the batch uses them.  The fused convolution reads a SNAPSHOT taken by this
 * call -- data and weights interleaved by row pairs, one more copy of the observation in device
 * memory (8 bytes per pixel and band, also for adopted buffers) -- so an adopted tensor changed
 * in place is not seen by it: call smi_batch_set_observation_device again to register the new
 * contents.  The _device form waits for the whole device once (the buffers may have been
 * written on any stream); the host form only for the batch's stream. */
int smi_batch_set_observation(smi_batch *b, const float *data, const float *weights);
int smi_batch_set_observation_device(smi_batch *b, const float *d_data,
                                     const float *d_weights);
/* kernel: [kernel_per_blend ? n_blends : 1][kernel_bands][kernel_h][kernel_w] */
int smi_batch_set_kernel(smi_batch *b, const float *kernel);

/* ConvolutionRenderer(psf_shift=...) (renderer.py:175-177, 215-228): the difference kernel
 * is moved by a free sub-pixel Fourier shift (fft.shift, fft.py:399-428) that the fit
 * updates like any other parameter (no constraint, AMSGrad step `step`, 1e-2 in the
 * reference).  Call after smi_batch_set_observation and smi_batch_set_components instead
 * of smi_batch_set_kernel.
 *   kernel    [n_sets][kernel_bands][h0][w0]: the unshifted stamps as the renderer holds
 *             them (h0 <= kernel_h, w0 <= kernel_w; centred in the batch's odd stamp);
 *             n_sets = kernel_per_blend ? n_blends : 1 (a shared kernel needs n_blends = 1)
 *   fft_shape FFT lengths (y, x) fft.shift uses for an (h0, w0) image (fft.py:116-167 with
 *             padding 10): the periodic interpolation depends on them
 *   shift     [n_sets][2] initial (y, x); moments [n_sets][6] = m, v, vhat (y, x each) or NULL
 * Every smi_batch_step then evaluates d(-logL)/d(shift) = sum w (m - d) (model (*) dK/ds)
 * at the parameters of the iteration, steps the shift and rebuilds the kernel spectrum on
 * the device; smi_batch_gradient evaluates the gradient only.  smi_batch_set_kernel makes
 * the kernel fixed again. */
int smi_batch_set_kernel_shift(smi_batch *b, const float *kernel, int32_t h0, int32_t w0,
                               const int32_t *fft_shape, const double *shift,
                               const double *moments, double step);
/* shift [n_sets][2], moments [n_sets][6], gradient [n_sets][2] (last evaluation), kernel
 * [n_sets][kernel_bands][kernel_h][kernel_w] at the current shift; any may be NULL */
int smi_batch_get_kernel_shift(smi_batch *b, double *shift, double *moments, double *gradient,
                               float *kernel);
/* relative_step for the kernel shift (parameter.py:126-129): step = max(step of
 * smi_batch_set_kernel_shift, factor * mean(shift)); factor = 0 (default): constant step.
 * Call after smi_batch_set_kernel_shift. */
int smi_batch_set_kernel_shift_relative_step(smi_batch *b, double factor);
int smi_batch_set_components(smi_batch *b, const smi_components *comps);

/* AMSGrad moments (blend.py:153-163), same packing as sed / morph; NULL = zeros */
int smi_batch_set_moments(smi_batch *b, const float *m_sed, const float *v_sed,
                          const float *vhat_sed, const float *m_morph,
                          const float *v_morph, const float *vhat_morph);
int smi_batch_get_moments(smi_batch *b, float *m_sed, float *v_sed, float *vhat_sed,
                          float *m_morph, float *v_morph, float *vhat_morph);
int smi_batch_get_parameters(smi_batch *b, float *sed, float *morph);
/* point-source centres / free shifts [n_components][2] (entries of other components:
 * 0) and their AMSGrad moments; the gradient is valid after smi_batch_gradient.  Any
 * pointer may be NULL.  For a shifting component smi_batch_get_parameters returns the
 * image parameter; smi_batch_get_model_morphology returns what enters the model (the
 * shifted image; equal to the parameter for every other component). */
int smi_batch_get_centers(smi_batch *b, double *center, double *m, double *v, double *vhat,
                          double *gradient);
int smi_batch_set_center_moments(smi_batch *b, const double *m, const double *v,
                                 const double *vhat);
/* New values [n_components][2] for the point-source centres / free shifts (entries of other
 * components are ignored); the morphologies that enter the model follow.  For a 2-vector the
 * host steps itself -- a prior, a constraint or a step callable on it (blend.py:120-145 treats
 * every Parameter alike): its device step is 0, the host takes the step from the gradient of
 * smi_batch_get_centers and writes the result back here. */
int smi_batch_set_centers(smi_batch *b, const double *center);
int smi_batch_get_model_morphology(smi_batch *b, float *morph);
int smi_batch_set_parameters(smi_batch *b, const float *sed, const float *morph);

/* Update rule of the parameters.  SMI_SCHEME_AMSGRAD (default): proxmin.adaprox as used
 * by Blend.fit and lite's AdaproxParameter.  SMI_SCHEME_FISTA: lite's FistaParameter
 * (lite/parameters.py:92-165, Beck & Teboulle 2009): y = z - step/sum(other^2) * grad,
 * x' = prox(y), t' = (1 + sqrt(1 + 4 t^2))/2, z = x + (1 + (t-1)/t')(x' - x); the
 * proximal operator is applied once.  Call before smi_batch_set_components; the
 * state is z (same layout as sed / morph, initialised to x) and t per parameter. */
enum { SMI_SCHEME_AMSGRAD = 0, SMI_SCHEME_FISTA = 1 };
int smi_batch_set_scheme(smi_batch *b, int32_t scheme);
/* t: [n_components][2] (spectrum, morphology); any pointer may be NULL */
int smi_batch_get_fista_state(smi_batch *b, float *z_sed, float *z_morph, double *t);
int smi_batch_set_fista_state(smi_batch *b, const float *z_sed, const float *z_morph,
                              const double *t);

/* AMSGrad constants forwarded by Blend.fit(**alg_kwargs) to adaprox (blend.py:165-180);
 * defaults b1 = 0.9, b2 = 0.999, eps = 1e-8 (lite/parameters.py:194) */
int smi_batch_set_optimizer(smi_batch *b, float b1, float b2, float eps);

/* scarlet.lite's loss has no normalisation term (lite/models.py:541): with
 * include = 0 the recorded loss is 1/2 sum w (m - d)^2 only (default 1: + log_norm,
 * observation.py:147-186).  Call before smi_batch_set_observation. */
int smi_batch_set_log_norm(smi_batch *b, int32_t include);

/* A further observation of every blend on the model's pixel grid: Blend._loss_func sums
 * the log-likelihoods of all observations (blend.py:264-271), so two observations that
 * share model channels are two terms of the loss and of the gradient image.  data, weights:
 * [n_blends][C][H][W] over the MODEL's channels (zero weight where this observation has
 * none); kernel: like smi_batch_set_kernel (same stamp shape / band layout as the batch's).
 * Needs the fused convolution path; call after smi_batch_set_observation / _set_kernel. */
int smi_batch_add_observation(smi_batch *b, const float *data, const float *weights,
                              const float *kernel);

/* Add a constant to the loss of every blend (on top of log_norm + chi^2 / 2).  The
 * facade uses it for the part of an observation that lies outside the model frame:
 * the reference zero-fills the model there (renderer.py:130-161, match_shape), so those
 * pixels contribute sum w d^2 / 2 and their share of log_norm to the loss
 * (observation.py:147-186) and, through |L|, to the stopping rule, but not to the
 * gradient.  Call after smi_batch_set_observation (which resets it). */
int smi_batch_add_loss_constant(smi_batch *b, const double *constant /* [n_blends] */);

/* Seed the "previous loss" of the stopping rule |L - L_prev| < e_rel |L| (same sign
 * convention as smi_batch_get_loss) for a batch that continues an earlier one, e.g.
 * after scarlet.lite resized a box: LiteBlend.fit keeps counting and compares the first
 * new loss with the last old one (lite/models.py:617-619). */
int smi_batch_set_previous_loss(smi_batch *b, const double *loss /* [n_blends] */);

/* HIP stream the batch launches on (hipStream_t as void*); NULL = default stream */
int smi_batch_set_stream(smi_batch *b, void *stream);

/* Blend.get_model + Observation.render + get_log_likelihood for every blend.
 * Any output may be NULL.  model, rendered: [n_blends][C][H][W]; logL[n_blends]
 * (includes -log_norm, observation.py:170-186). */
int smi_batch_forward(smi_batch *b, float *model, float *rendered, double *logL);

/* Gradient of the loss (-logL) w.r.t. every sed and morphology at the current
 * parameters; layout as sed / morph. */
int smi_batch_gradient(smi_batch *b, float *g_sed, float *g_morph);

/* Run `n_iter` iterations of the adaprox loop on the device without host
 * synchronisation; `it0` is the iteration counter of the first one (it = 0 takes
 * a tenth of the step and sets vhat = v).  `e_rel` is the relative tolerance of
 * the proximal sub-iterations (lite/parameters.py:297-303) and, when
 * `check_convergence` != 0, of the per-blend stopping rule evaluated on the device:
 * a blend stops after the update of the iteration in which
 *   it > min_iter && |L[it] - L[it-1]| < e_rel |L[it]|        (blend.py:294-299).
 * Asynchronous on the batch stream. */
int smi_batch_step(smi_batch *b, int32_t it0, int32_t n_iter, float e_rel,
                   int32_t min_iter, int32_t prox_max_iter, int32_t check_convergence);

/* Blends are independent, so a step can run ranges of blends on streams of their own:
 * while one range's update kernel drains, the others' next convolution fills the chip.
 * n = 0 (default): automatic (with the fused convolution 3 ranges from 128 blends on, 4 below
 * 768 blends when smi_set_hw_queues said that eight hardware queues exist; else 1);
 * point sources, free shifts and a low-resolution observation keep it at 1.  Results are
 * identical for every n.  The caller's stream (smi_batch_set_stream) still brackets the
 * step: the ranges start after its pending work and it waits for all of them. */
int smi_batch_set_sub_ranges(smi_batch *b, int32_t n);

/* Hardware queues the HIP runtime of this process maps streams onto (GPU_MAX_HW_QUEUES when
 * the runtime started; HIP's default is 4).  The library cannot see that number and does not
 * read the environment: it assumes 4 until the host says otherwise, and runs four ranges of
 * blends side by side only with n >= 8 (streams that share a queue run one after the other).
 * Process-wide; returns the previous value. */
int smi_set_hw_queues(int32_t n);
int smi_batch_get_sub_ranges(smi_batch *b, int32_t *n);

/* A batch of nothing but factorized components under one fused convolution (no point
 * sources, free shifts, further observations) needs Blend.get_model (blend.py:200-244) only
 * as the input rows of the convolution; by default (on = 1) the convolution kernel renders
 * those rows itself and smi_batch_step neither launches the render kernel nor moves a model
 * cube through HBM.  on = 0 keeps the cube (render kernel per iteration).  The rows are the
 * same bit for bit either way; boxes wider than 81 pixels always take the cube. */
int smi_batch_set_inline_render(smi_batch *b, int32_t on);

/* Blocking.  n_active: blends still iterating; error: index of the first blend
 * whose parameters became non-finite, or -1. */
int smi_batch_status(smi_batch *b, int32_t *n_active, int32_t *first_error);
/* per-blend state: 0 iterating, 1 in its last iteration, 2 converged (stopping rule),
 * 3 stopped with non-finite parameters (Model.check_parameters, model.py:153-165) */
int smi_batch_get_states(smi_batch *b, int32_t *state /* [n_blends] */);

/* Blend.fit for the whole batch: step() in chunks of `sync_every` iterations until
 * max_iter or until every blend has converged.  n_iter[n_blends] receives the
 * number of loss evaluations per blend (= len(blend.loss)). */
int smi_batch_fit(smi_batch *b, int32_t max_iter, float e_rel, int32_t min_iter,
                  int32_t prox_max_iter, int32_t sync_every, int32_t *n_iter);

/* loss history: out[n_blends][capacity] (loss = -logL, blend.py:273); entries past
 * a blend's n_iter are NaN. */
int smi_batch_get_loss(smi_batch *b, double *out, int32_t capacity, int32_t *n_iter);

/* Re-arm all blends (active, loss history cleared); parameters are kept. */
int smi_batch_reset(smi_batch *b);

/* Keep / bring back a device-side copy of everything a step changes (parameters, m / v /
 * vhat, FISTA and point-source state): the warm restart of Blend.fit (X, M, V, Vhat carried
 * into a new adaprox call, blend.py:155-170) without a round trip over the host.
 * smi_batch_restore_state also re-arms the blends like smi_batch_reset.  Both are
 * asynchronous on the batch stream. */
int smi_batch_save_state(smi_batch *b);
int smi_batch_restore_state(smi_batch *b);

/* Mean device time in milliseconds of the dominant kernel family per iteration,
 * measured with hipEvents on the batch stream during the last smi_batch_step call
 * when timing was enabled with smi_batch_enable_timing(b, 1); with several ranges of
 * blends (smi_batch_set_sub_ranges) the times are those of the first range, whose
 * kernels share the chip with the other ranges'.
 * phase: 0 render, 1 forward conv, 2 residual, 3 adjoint conv, 4 update, 5 total */
int smi_batch_enable_timing(smi_batch *b, int32_t on);
int smi_batch_get_timing(smi_batch *b, double *ms_per_phase, int32_t n_phases);

/* FFT shape actually used (fft_h, fft_w) */
int smi_batch_fft_shape(smi_batch *b, int32_t *fft_h, int32_t *fft_w);

/* ---------------------------------------------------------------------------------
 * Box resizing (ImageMorphology.update, morphology.py:132-207; Blend.fit's hook every ten
 * iterations, blend.py:284-292) for a batch that stays on the device between hooks.
 *
 * smi_batch_resize_test: the two reductions update() decides on, per component --
 *   margin[k]: width of the frame of pixels <= 0 around the image (shrink_box; INT32_MAX for
 *   an image without a pixel > 0), edge_pull[k]: the largest of the four edge means of
 *   -m / sqrt(sqrt(v)) * step * (image > 0) over the pixels with v != 0 (-inf if there is
 *   none; summed in another order than NumPy does and with the float32 step: callers treat
 *   values within 1e-6 of the threshold 0.1 as "ask the host").  Point sources and shifted
 *   images report (-1, nan).
 * State record of a component, float32: [sed C][m C][v C][vhat C][image N][m N][v N][vhat N],
 *   N = box_h * box_w.  smi_batch_get_component_states packs the records of the listed
 *   components back to back into `states`.
 * smi_batch_update_components: a new component table (same components per blend, `sed` and
 *   `morph` of *c are not read) in which the components with keep[k] != 0 keep their device-
 *   resident parameters and moments (their boxes must not have changed) and the others take
 *   theirs from `states`, records in the order of k.  The observation, the kernel, the loss
 *   histories and the per-blend states stay.  Factorized image components under AMSGrad
 *   only.  A table that fails the argument checks leaves the old one and its state in place;
 *   after an allocation or transfer error the batch is without components.
 *   keep[k] = 2 / 3: the component keeps its device-resident state and its SQUARE box is resized
 *   about its centre on the device, the way ImageMorphology.update does it on the host
 *   (morphology.py:132-207): the new side is box_h[k] = box_w[k] (sides differ by an even
 *   number, at most 1024 pixels), a smaller box takes the centred slice of image and moments, a
 *   larger one pads the moments with zeros and the image with np.pad(mode="linear_ramp") in
 *   float64 rounded to float32 -- keep 2 when the host image is float32 (the rows added along
 *   axis 0 are rounded before the ramps along axis 1), keep 3 when it is float64.  origin and
 *   morph_step of such a row are the caller's (origin -/+ the inset, step / 2).
 * smi_batch_set_states / smi_batch_get_progress: per-blend state (0 iterating, 2 finished or
 *   paused -- its workgroups return at once --, 3 failed) and number of recorded losses.
 * smi_batch_set_iteration_base: per blend, the iteration counter at which its current adaprox
 *   call began (NULL: 0 for all).  smi_batch_step(it0, n) then runs the blends of a batch at
 *   different counters in one launch: a blend takes the rules of the first step (alpha / 10,
 *   vhat = v: the restart after a box resize, blend.py:276-302) and min_iter from
 *   it - base[b].  Factorized image components under AMSGrad only.
 * --------------------------------------------------------------------------------- */
int smi_batch_resize_test(smi_batch *b, int32_t *margin, double *edge_pull);
int smi_batch_get_component_states(smi_batch *b, const int32_t *components, int32_t n,
                                   float *states);
int smi_batch_update_components(smi_batch *b, const smi_components *c, const int32_t *keep,
                                const float *states);
int smi_batch_set_states(smi_batch *b, const int32_t *state);
int smi_batch_set_iteration_base(smi_batch *b, const int32_t *base);
int smi_batch_get_progress(smi_batch *b, int32_t *state, int32_t *n_loss);
/* Per blend: the iteration counter `it` (as passed to smi_batch_step) after whose update the
 * blend pauses -- it is skipped by the rest of the call like a blend that has stopped (state 2)
 * -- or NULL: nobody pauses.  One smi_batch_step(it0, n) then takes every blend to ITS next
 * resize hook (blend.py:196-198: every ten iterations of the blend's own adaprox call) or to
 * the end of its iteration budget, instead of all blends to the nearest hook of any of them
 * (a thousand blends whose calls restarted at different times: 12 calls instead of 32).
 * smi_batch_get_converged: 1 for the blends whose stopping rule (blend.py:294-299) fired since
 * the last smi_batch_set_pause_at -- what tells a blend that stopped from one that pauses. */
int smi_batch_set_pause_at(smi_batch *b, const int32_t *it);
int smi_batch_get_converged(smi_batch *b, int32_t *flag);
/* What a round of fit_blends sends and fetches, in one call each (stream-ordered copies through
 * a pinned staging buffer: two synchronisations per round instead of eight small blocking
 * copies -- a single scene's fit has eight rounds of ~2 ms).  smi_batch_set_round = set_states +
 * set_iteration_base + set_pause_at (a NULL array is left as it is on the device);
 * smi_batch_get_round = get_progress + get_converged. */
int smi_batch_set_round(smi_batch *b, const int32_t *state, const int32_t *base,
                        const int32_t *pause_at);
int smi_batch_get_round(smi_batch *b, int32_t *state, int32_t *n_loss, int32_t *converged);

/* Number of host-to-device uploads of observation cubes (smi_batch_set_observation and
 * smi_batch_add_observation) this process has made so far: lets a caller -- and the tests --
 * check that a driver which keeps its batch alive does not ship the data again. */
int64_t smi_observation_uploads(void);

/* The convolution path the batch runs (what conv_path = 0 "auto" resolved to):
 * 0 none (NullRenderer, no difference kernel), 1 rocFFT pipeline, 2 fused_conv_kernel. */
int smi_batch_conv_path_used(smi_batch *b, int32_t *path);

/* ---------------------------------------------------------------------------------
 * Multi-resolution rendering: the per-call part of ResolutionRenderer
 * (renderer.py:478-545) for unrotated pixel grids.  The host builds the two linear
 * operators once (scarlet_amd/renderer.py): A[C][n_a][Fy*Fx], the difference kernel
 * shifted along y to every low-resolution row (renderer.py:341-353), and
 * Pt[Fx][Fx*n_b], the transposed operator that shifts a padded model row to every
 * low-resolution column (renderer.py:498-505).  A rendering of a padded model cube
 * [C][Fy][Fx] is the linear map out[C][n_a][n_b] = A . (model . Pt) per band.
 *
 * The reference's shift is a phase ramp between a real transform and its inverse on the
 * padded grid (renderer.py:414-476), i.e. Pt is circulant along x: Pt[x'][x][b] =
 * s_b[(x - x') mod Fx].  smi_resampler_create checks that, and then evaluates the same map
 * through transforms along x of the model rows, the operator rows (made once, on the
 * device, in float64) and s_b -- path 1, "spectral": 1/25 of the arithmetic and 1/5 of the
 * HBM traffic of the two dense products (path 0), which stay for any other Pt. */
typedef struct smi_resampler smi_resampler;
int smi_resampler_create(const float *A, const float *Pt, int32_t C, int32_t n_a, int32_t n_b,
                         int32_t Fy, int32_t Fx, smi_resampler **out);
int smi_resampler_render(smi_resampler *r, const float *model, float *out);
/* mean device time (ms) of one rendering of the model last passed to smi_resampler_render,
 * over n_rep repetitions without host transfers (benchmark of BASELINE config 5) */
int smi_resampler_time(smi_resampler *r, int32_t n_rep, double *ms_per_render);
int smi_resampler_destroy(smi_resampler *r);
/* which evaluation the resampler uses: 0 = two dense products per band, 1 = spectral.
 * smi_resampler_set_path(r, 0) switches a spectral resampler to the dense products (the
 * parity tests compare the two); path 1 is refused for an operator that is not circulant. */
int smi_resampler_get_path(smi_resampler *r, int32_t *path);
int smi_resampler_set_path(smi_resampler *r, int32_t path);

/* The low-resolution observation as a further term of a fit (Blend._loss_func sums the
 * log-likelihoods of all observations, blend.py:265-271; Observation.get_log_likelihood,
 * observation.py:147-170).  `channels[C]` gives the model channel of every band of the
 * resampler, `data` / `weights` are [C][n_a][n_b], `log_norm` is Observation.log_norm of
 * that observation.  From then on every iteration renders the model cube through `r`,
 * adds log_norm + chi^2 / 2 to the loss and the transposed operator applied to
 * w (m - d) to the gradient image.  Every call adds one observation (at most 8).  Batches
 * of one blend only; `r` must outlive `b`. */
int smi_batch_attach_lowres(smi_batch *b, smi_resampler *r, const int32_t *channels,
                            const float *data, const float *weights, double log_norm);
/* rendering [C][n_a][n_b] of the index-th attached observation in the last forward /
 * gradient / step call */
int smi_batch_get_lowres_rendered(smi_batch *b, int32_t index, float *out);

#ifdef __cplusplus
}
#endif
#endif /* SCARLET_AMD_H */
