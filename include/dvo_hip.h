/*
 * dvo_hip.h — C ABI of the MI355X (gfx950) dense RGB-D direct-alignment core.
 *
 * This is the drop-in boundary for ONE path of pfontana96/dense-visual-odometry: the backend
 * hooks of BaseRobustDVO and the few lines of its Gauss-Newton loop that consume them.
 * Reference citations are relative to /root/reference/src/dense_visual_odometry/.
 *
 * Conventions
 *   - plain C symbols, plain pointers and sizes; no C++/torch types cross the boundary;
 *   - every function returns DVO_OK (0) or a negative dvo_status; nothing throws;
 *     dvo_last_error(ctx) returns a human-readable message for the last failure on that ctx
 *     (dvo_last_error(NULL): last failure of dvo_create on the calling thread);
 *   - host pointers are owned by the caller and only need to be valid for the duration of the
 *     call; the ctx owns all device memory and (unless dvo_set_stream is used) one HIP stream;
 *   - one ctx per GPU; a ctx is not thread-safe, different ctxs are independent;
 *   - there is NO CPU fallback: without a HIP device every entry point fails with
 *     DVO_ERR_NO_DEVICE / DVO_ERR_HIP.
 *
 * Data model (device resident, per frame "slot", per pyramid level L of size w_L x h_L):
 *   gray   u8  [h][w]            intensity (level 0 = caller's image, L>0 = median3 + 2x decimate)
 *   depth  u16 [h][w]            raw depth (same pyramid rule)
 *   packed u32 [h+1][pitch]      bits 0-7 intensity, bits 8-19 Sobel gx (signed 12 bit),
 *                                bits 20-31 Sobel gy (signed 12 bit); one zero pad row/column so
 *                                the 2x2 bilinear neighbourhood is always addressable
 *
 * Pose arguments `T` are row-major 3x4 float32 [R|t] (the top three rows of Se3.exp(),
 * utils/lie_algebra/special_euclidean_group.py:35-52), one per frame pair.
 *
 * Normal-equation record (`out` of dvo_normal_equations, float64, DVO_NE_STRIDE = 32 per pair):
 *   [0..20]  upper triangle of H = J^T W J, row-major (H00 H01 .. H05 H11 .. H55)
 *   [21..26] J^T W r      (the caller negates: b = -J^T W r, base_robust_dvo.py:188)
 *   [27]     sum w r^2    [28] number of valid pixels N    [29] sum r^2    [30..31] reserved (0)
 *
 * Environment (every variable is read ONCE, at dvo_create; a captured solve graph bakes the values in):
 *   DVO_GRAPH=0            plain launches instead of hipGraph replay of the device-side solve
 *   DVO_ST_EXACT           DVO_WEIGHT_STUDENT_T: no one-pass forms anywhere (passes as written; see dvo_solve_batch)
 *   DVO_ST_SOLVE_PHASES    ... passes as written with a control launch after every scale pass (the original flow)
 *   DVO_NO_RCACHE          ... the second scale pass recomputes the residuals instead of reading the first pass's
 *   DVO_INGEST_UNFUSED     separate ingest kernels at every level (the path odd image sizes take anyway)
 *   DVO_SMALL_MAX_PIX=n    pyramid levels of at most n pixels run in the persistent per-pair kernel (default 19200; 0: none)
 *   DVO_SMALL_THREADS=n    workgroup size of that kernel, 256 / 512 / 1024 (default: chosen from the batch size)
 *   DVO_MAX_GROUPS=n       a reduction workgroup runs through at most n groups of 1024 valid-point records (default 48; matters for
 *                          launches of more than ~16 VGA pairs).  Longer workgroups are faster and sum more records per lane in float32
 *                          before the float64 stages: 16 gives the accuracy of small launches (worst increment error on the reference
 *                          run's iterations 1.5e-6 instead of 5e-6 of the 1e-5 bound) for ~1.2 % of the throughput
 *   DVO_RCCL_LIB=path      RCCL library dvo_comm_create loads (default librccl.so)
 * These are tested configuration switches.  Tuning and diagnostic knobs (DVO_EXP_TAPS, DVO_SMALL_STAMPS, DVO_TARGET_BLOCKS, ...)
 * exist only in experiment builds of the library (-DDVO_EXPERIMENTS, bottom of this file).
 */
#ifndef DVO_HIP_H
#define DVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dvo_ctx dvo_ctx;

typedef enum dvo_status {
    DVO_OK = 0,
    DVO_ERR_INVALID_ARG = -1,
    DVO_ERR_NO_DEVICE = -2,
    DVO_ERR_HIP = -3,
    DVO_ERR_OUT_OF_MEMORY = -4,
    DVO_ERR_BAD_STATE = -5      /* e.g. slot not uploaded, camera not set */
} dvo_status;

/* which of the reference's three behavioural variants of the residual/Jacobian path */
typedef enum dvo_variant {
    DVO_VARIANT_V1 = 0,  /* CPU default: Sobel grad of I2 sampled at the warped point, J_w at the un-warped
                            point (cpu_robust_dense_visual_odometry.py:58,123-130,168-170) */
    DVO_VARIANT_V2 = 1,  /* approximate_image2_gradient=True: Sobel grad of I1 at the pixel (:60-77,:187-188) */
    DVO_VARIANT_V3 = 2,  /* numba.cuda kernel: central-difference grad of I1, J_w at the WARPED point, strict
                            bilinear bounds (cuda/residuals_kernel.py:46-109) */
    DVO_VARIANT_GN = 3   /* build-defined production mode: V1's residual, validity and sampled Sobel gradient of
                            I2, with the gradient normalised to a unit derivative (x 1/8) and J_w evaluated at the WARPED point:
                            a true Gauss-Newton step per iteration instead of V1's 1/8 step with the Jacobian of the un-warped
                            point (quirks Q1, Q2).  Same cost per iteration as V1; not a reference behaviour (parity unpinned) */
} dvo_variant;

typedef enum dvo_weight_mode {
    DVO_WEIGHT_NONE = 0,       /* use_weighter=False: w = 1 */
    DVO_WEIGHT_STUDENT_T = 1,  /* w = 6 / (5 + r^2 * lambda); weight_param[pair] = lambda
                                  (weighter/t_weighter.py:34) */
    DVO_WEIGHT_HUBER = 2,      /* build-defined: w = min(1, k/|r|); weight_param[pair] = k (grey levels) */
    DVO_WEIGHT_STUDENT_T_MEAN = 3  /* build-defined "proper" Student-t: the same weight, but the scale is the MEAN of the weighted
                                  r^2 (what t_weighter.py:41-45 intends; the reference's np.mean of a 1-element row is a sum, quirk
                                  Q3), IRLS to |d lambda| <= 1e-3 lambda, at most 4 passes per Gauss-Newton iteration, each iteration
                                  warm-started from the previous one's lambda.  weight_param[pair] = lambda. */
} dvo_weight_mode;

#define DVO_NE_STRIDE 32
#define DVO_MAX_LEVELS 8

/* ---- lifetime -------------------------------------------------------------------------------- */

/* Creates a context on HIP device `device` with `n_slots` frame slots sized for max_w x max_h images (at most
 * 4096 x 4096) and up to max_levels pyramid levels, and scratch for up to max_pairs frame pairs per call.
 * Counterpart of RobustDVOGPU.__init__ pre-allocating its pyramids
 * (core/robust_dense_visual_odometry/gpu_robust_dense_visual_odometry.py:17-47). */
int dvo_create(int device, int max_w, int max_h, int max_levels, int n_slots, int max_pairs, dvo_ctx** out);
void dvo_destroy(dvo_ctx* ctx);
const char* dvo_last_error(const dvo_ctx* ctx);

/* Launch all work of this ctx on an externally owned hipStream_t (e.g. torch's current stream).
 * NULL restores the ctx's own stream. */
int dvo_set_stream(dvo_ctx* ctx, void* hip_stream);
int dvo_synchronize(dvo_ctx* ctx);

/* ---- camera (camera_model.py:17-79) ------------------------------------------------------------ */

/* Level-0 intrinsics; level L intrinsics follow RGBDCameraModel.at (camera_model.py:62-79).
 * Call it BEFORE uploading frames (the reference takes the camera model in the constructor): the ingest stores
 * z = depth * depth_scale with every depth-valid pixel.  A frame uploaded earlier, or under another depth_scale, keeps
 * its pyramid but the alignment calls refuse it as the previous frame of a pair (DVO_ERR_BAD_STATE) until it is
 * uploaded again; changing only the intrinsics needs no re-upload (cached launch graphs of dvo_solve_batch are dropped,
 * so the next solve uses the new intrinsics). */
int dvo_set_camera(dvo_ctx* ctx, float fx, float fy, float cx, float cy, float skew, double depth_scale);
/* Writes the 3x4 float32 level intrinsics the kernels use (for tests / hosts that want to cross-check). */
int dvo_get_level_intrinsics(const dvo_ctx* ctx, int level, float* k3x4);

/* ---- frame ingest: replaces RobustDVOCPU._build_pyramids + _setup --------------------------------
 * (cpu_robust_dense_visual_odometry.py:44-77; utils/image_pyramid.py:19-54; utils/jacobian.py:47-73) */

/* Host images -> slot: H2D copy, then on device: (median3 + 2x decimate) per level for gray and depth,
 * Sobel3x3/BORDER_REFLECT of every gray level packed with the intensity, and per level the list of depth-valid
 * pixels (camera_model.py:195 mask, :200-201 z) the Gauss-Newton iterations work on. */
int dvo_upload_frame(dvo_ctx* ctx, int slot, const uint8_t* gray, const uint16_t* depth, int w, int h, int levels);
/* Same, but gray/depth are DEVICE pointers already resident in HBM (async on the ctx stream). */
int dvo_upload_frame_device(dvo_ctx* ctx, int slot, const void* d_gray, const void* d_depth, int w, int h, int levels);
/* Batched form: n_frames images, contiguous [n][h][w] DEVICE arrays, into the listed (distinct) slots; one fused
 * launch per pyramid level plus three for the valid-pixel lists, for the whole batch (async on the ctx stream). */
int dvo_upload_frames_device(dvo_ctx* ctx, int n_frames, const int* slots, const void* d_gray, const void* d_depth,
                             int w, int h, int levels);
/* Batched form for HOST images: contiguous [n][h][w] arrays, two large H2D copies + the batched build.  With
 * page-locked host memory (dvo_host_alloc) the copies are asynchronous: the arrays must then stay untouched until
 * the next blocking call on this ctx (dvo_synchronize, dvo_solve_batch_fetch, ...). */
int dvo_upload_frames(dvo_ctx* ctx, int n_frames, const int* slots, const uint8_t* gray, const uint16_t* depth,
                      int w, int h, int levels);
/* Stream ordering between two contexts of one GPU: work enqueued on `ctx` after this call starts only when
 * everything enqueued on `other` so far has finished (event record + stream wait; nothing blocks on the host).
 * Used to run the uploads of several contexts one after another on the PCIe link while their solves overlap. */
int dvo_wait_for(dvo_ctx* ctx, dvo_ctx* other);
/* Several contexts of one GPU fed from host memory: after this call the host->device copies of dvo_upload_frames(ctx)
 * are issued on `peer`'s copy stream (peer may be ctx itself; NULL restores copies on the ctx stream) and the ctx
 * stream waits for them by event.  With one peer for all contexts the uploads take turns on the PCIe link in call
 * order while each context's kernels run behind its own copies only: upload of one overlaps the solve of another.
 * `peer` must outlive `ctx`'s last upload. */
int dvo_set_upload_peer(dvo_ctx* ctx, dvo_ctx* peer);
/* Page-locked host memory for the upload paths (hipHostMalloc / hipHostFree). */
int dvo_host_alloc(size_t bytes, void** out);
int dvo_host_free(void* p);
/* Colour ingest: BGR u8 [h][w][3] -> gray (OpenCV 8-bit fixed point) and depth clamp
 * depth*depth_scale > max_distance -> 0, on device (core/base_dense_visual_odometry.py:58-59). */
int dvo_upload_frame_bgr(dvo_ctx* ctx, int slot, const uint8_t* bgr, const uint16_t* depth, int w, int h, int levels,
                         double max_distance);
/* Batched colour ingest: n_frames contiguous [n][h][w][3] BGR images + [n][h][w] depth images into the listed (distinct)
 * slots.  The host form moves the batch with two large copies like dvo_upload_frames (asynchronous for page-locked memory,
 * on the shared copy stream when an upload peer is set), then one device pass converts / clamps all frames into the level-0
 * planes and the batched pyramid build follows.  Replaces cvtColor + clamp of core/base_dense_visual_odometry.py:58-59 and
 * the all-in-RAM frame loop of src/test_dvo.py:180-191 for colour sequences. */
int dvo_upload_frames_bgr(dvo_ctx* ctx, int n_frames, const int* slots, const uint8_t* bgr, const uint16_t* depth,
                          int w, int h, int levels, double max_distance);
int dvo_upload_frames_bgr_device(dvo_ctx* ctx, int n_frames, const int* slots, const void* d_bgr, const void* d_depth,
                                 int w, int h, int levels, double max_distance);
/* Copies one level of a slot back to the host (any pointer may be NULL). gx/gy are the unpacked Sobel planes. */
int dvo_download_level(dvo_ctx* ctx, int slot, int level, uint8_t* gray, uint16_t* depth, int16_t* gx, int16_t* gy);
int dvo_slot_info(const dvo_ctx* ctx, int slot, int* w, int* h, int* levels);
/* Copies the valid-point list of one level back to the host, for tests and hosts that want the reference's deprojection
 * inputs: *count = number of depth-valid pixels (mask of camera_model.py:195), and for the first min(*count, capacity) of
 * them, in raster order, z[i] = (float)((double)depth * depth_scale) (camera_model.py:200-201) and
 * pix[i] = x | y << 12 | intensity << 24.  z / pix may be NULL.  DVO_ERR_BAD_STATE if the slot was uploaded before
 * dvo_set_camera. */
int dvo_download_points(dvo_ctx* ctx, int slot, int level, float* z, uint32_t* pix, int capacity, int* count);

/* ---- the hot path ------------------------------------------------------------------------------ */

/* Fused residual + Jacobian + weight + 6x6 reduction for n_pairs frame pairs at one level: what
 * compute_residuals_and_jacobian (cpu_robust...:134-200) followed by H = J^T W J, b = -J^T W r
 * (base_robust_dvo.py:168-188) computes, without materialising r or J.  Blocks until `out` is valid. */
int dvo_normal_equations(dvo_ctx* ctx, int level, int n_pairs, const int* prev_slot, const int* cur_slot,
                         const float* T, int variant, int weight_mode, const double* weight_param, double* out);

/* Student-t scale pass: out[pair] = sum_i r_i^2 * 6 / (5 + r_i^2 * lambda[pair])
 * (TDistributionWeighter._compute_scale, weighter/t_weighter.py:36-47 — a sum, reference quirk Q3). */
int dvo_weighted_r2_sum(dvo_ctx* ctx, int level, int n_pairs, const int* prev_slot, const int* cur_slot,
                        const float* T, int variant, const double* lambda, double* out);

/* The same pass returning both the sum and the number of residuals it ran over (depth-valid and in bounds):
 * stats[2 * pair] = sum, stats[2 * pair + 1] = count; the mean-based scale of DVO_WEIGHT_STUDENT_T_MEAN is sum / count. */
int dvo_weighted_r2_stats(dvo_ctx* ctx, int level, int n_pairs, const int* prev_slot, const int* cur_slot,
                          const float* T, int variant, const double* lambda, double* stats);

/* Dense outputs with the return contract of compute_residuals_and_jacobian (API parity / debugging):
 * r [h*w] (0 where invalid), J [h*w][6], valid [h*w] (depth != 0 AND warped point in bounds),
 * uv [h*w][2] warped coordinates (NaN where depth == 0).  Any output pointer may be NULL. */
int dvo_residuals_jacobian(dvo_ctx* ctx, int level, int prev_slot, int cur_slot, const float* T, int variant,
                           float* r, float* J, uint8_t* valid, float* uv);

/* Whole coarse-to-fine Gauss-Newton solve of n_pairs frame pairs on the device (BaseRobustDVO._step,
 * base_robust_dvo.py:137-236): no host round trip; stop rules evaluated per pair on the device.  Pyramid levels of at most
 * 160 x 120 pixels run in one persistent launch (a workgroup per pair does all their iterations); finer levels take one
 * reduction launch and one control launch per iteration.
 * Student-t IRLS (weighter/t_weighter.py:21-34), DVO_WEIGHT_STUDENT_T (the reference's weighter, summing quirk included):
 *   - persistent levels: the reference's loop as written (until |d lambda| < 1e-3, at most 50 passes);
 *   - finer levels (one launch per iteration): the TWO scale passes the reference's loop takes whenever the summed scale exceeds
 *     1000 (any image with more than a few hundred non-zero residuals: lambda is then below 1e-3 after the first pass, so the second
 *     difference is too) - for DVO_VARIANT_V1 at levels of 320 x 240 and more evaluated in ONE pass over the points: the scale sum of
 *     the second pass and the weighted normal equations are series in lambda r^2 / 5 (below ~1e-4 there, because the reference's
 *     scale is a SUM over the image) whose terms are moments of the same pass; every iteration bounds, on the increment itself
 *     (explicit H^-1, right-hand side and matrix perturbation), what the form drops, at half the parity tolerance of 1e-5;
 *   - FALLBACK, PER PAIR, before dvo_solve_batch / dvo_solve_batch_fetch return: a pair with an iteration whose bound was not met is
 *     solved again with the passes as written; a pair whose IRLS does not have that two-pass shape at some iteration of a finer level
 *     (detected by the one-pass forms and by the passes as written alike) is solved again with EVERY level in the persistent kernel,
 *     i.e. by the reference's loop itself.  Only the flagged pairs are solved again (one compacted batch per form); the other pairs
 *     of the batch keep their results.  dvo_student_t_fallbacks / dvo_student_t_fallback_pairs count them.
 *   The environment variable DVO_ST_EXACT (read at dvo_create) switches the one-pass forms off everywhere (passes as written, shape
 *   fallback as above).  DVO_VARIANT_V3 keeps per-iteration launches with two scale passes at every level and has no fallback.
 *   DVO_WEIGHT_STUDENT_T_MEAN (build-defined) runs at most four passes at the finer levels.
 * RULE for the split enqueue / fetch form with DVO_WEIGHT_STUDENT_T: the fallback reads the FRAMES at fetch time, so the slots of an
 * enqueued solve must not be uploaded again before its fetch (checked: DVO_ERR_BAD_STATE from the fetch if a pair that needs the
 * fallback lost its frames; pairs that do not need it are unaffected).  T_init, the prior and the level schedule are copied at
 * enqueue and may change freely.
 *   T_init  n_pairs x 12 or NULL (identity)          T_out   n_pairs x 12
 *   xi_out  n_pairs x 6 float64 log of the estimate (may be NULL)
 *   iters_out n_pairs (total GN iterations run, may be NULL)   err_out n_pairs (last accepted error, may be NULL) */
int dvo_solve_batch(dvo_ctx* ctx, int n_pairs, const int* prev_slot, const int* cur_slot, int levels,
                    int max_iterations, double tolerance, int max_increased_steps_allowed, int variant,
                    int weight_mode, double weight_param, const float* T_init, float* T_out, double* xi_out,
                    int* iters_out, double* err_out);
/* Asynchronous form: enqueue only; results land in device buffers, fetched by dvo_solve_batch_fetch - once per enqueue (a second
 * fetch of a DVO_WEIGHT_STUDENT_T solve whose first fetch had to solve pairs again is refused: DVO_ERR_BAD_STATE). */
int dvo_solve_batch_enqueue(dvo_ctx* ctx, int n_pairs, const int* prev_slot, const int* cur_slot, int levels,
                            int max_iterations, double tolerance, int max_increased_steps_allowed, int variant,
                            int weight_mode, double weight_param, const float* T_init);
int dvo_solve_batch_fetch(dvo_ctx* ctx, int n_pairs, float* T_out, double* xi_out, int* iters_out, double* err_out);
/* Diagnostic: number of fetched DVO_WEIGHT_STUDENT_T solves in which at least one pair had to be solved again (see above) ... */
int dvo_student_t_fallbacks(const dvo_ctx* ctx, int64_t* count);
/* ... and the number of pairs solved again over all of them. */
int dvo_student_t_fallback_pairs(const dvo_ctx* ctx, int64_t* pairs);

/* Motion prior of the reference (sigma, base_robust_dvo.py:69-71,:154,:190-194,:217-218) for subsequent
 * dvo_solve_batch calls: H += I/sigma, b += (I/sigma) log(old), err += 0.5 sigma |log(old)|, old <- inc^-1 old, with
 * `old` restarting at every level from the previous pair's transform last_qt[pair] = (qw qx qy qz tx ty tz).  With
 * last_qt == NULL there is no previous transform and the solve applies NO prior, as the reference does for the first
 * pair of a sequence (`old is None`, base_robust_dvo.py:154,:190).  sigma <= 0 switches the prior off.  The prior couples
 * consecutive pairs of a sequence, so with it a sequence is processed pair after pair (SURVEY 8e "when not"). */
int dvo_set_prior(dvo_ctx* ctx, double sigma, const double* last_qt, int n_pairs);

/* Optional iteration schedule for subsequent dvo_solve_batch calls: level l (0 = finest) runs at most iters[l] Gauss-Newton
 * iterations instead of the call's max_iterations (0 skips the level; levels beyond n_levels take iters[n_levels - 1]).
 * iters == NULL or n_levels <= 0 restores the reference's behaviour, one max_iterations for every level
 * (base_robust_dvo.py:161).  Build-defined: meant for DVO_VARIANT_GN, which converges in 1-3 iterations per level. */
int dvo_set_level_iterations(dvo_ctx* ctx, const int* iters, int n_levels);

/* Per-iteration trace of dvo_solve_batch (tests / debugging): enable with the number of iterations to keep
 * per pair (0 disables).  Each record is DVO_TRACE_STRIDE doubles: the normal-equation record [0..31], the
 * solved increment xi [32..37] and the 3x4 pose the iteration was evaluated at [38..49]. */
#define DVO_TRACE_STRIDE 50
int dvo_trace_enable(dvo_ctx* ctx, int max_iterations_traced);
int dvo_trace_read(dvo_ctx* ctx, int n_pairs, double* out /* n_pairs x cap x DVO_TRACE_STRIDE */);

/* Test hook: out[i] = the quotient a[i]/b[i] as the kernels' projection computes it (shared refined reciprocal + one residual
 * correction), ref[i] = the compiler's IEEE division on the device.  Host arrays of n floats.  a == NULL: out[i] = the refined
 * reciprocal of b[i] itself, ref[i] = 1 / b[i] (the quotient is correctly rounded for every numerator iff that reciprocal is the
 * float nearest to 1 / b: checked for all 2^23 mantissas in the GPU suite). */
int dvo_debug_divide(dvo_ctx* ctx, const float* a, const float* b, float* out, float* ref, int n);

/* ---- multi-GPU: the one collective of the path -------------------------------------------
 * One process per GPU; after a round every rank contributes n_records pose records of 16 floats (12 floats [R|t],
 * iterations, final error, 2 reserved) and receives all ranks' records in rank order: one RCCL all-gather over
 * xGMI (64 B per record: latency-bound).  RCCL is loaded with dlopen at first use.  No counterpart in the reference
 * (it has no distributed code); Python hosts may equally use torch.distributed (dvo_amd.batch.allgather_poses). */
typedef struct dvo_comm dvo_comm;
int dvo_comm_unique_id(void* id128 /* 128 bytes out: rank 0 creates it and ships it to the other ranks */);
int dvo_comm_create(int device, int rank, int nranks, const void* id128, dvo_comm** out);
int dvo_allgather_poses(dvo_comm* comm, const float* mine /* n_records x 16 */, int n_records,
                        float* all /* nranks x n_records x 16 */);
void dvo_comm_destroy(dvo_comm* comm);
const char* dvo_comm_last_error(void);

/* ---- measurement ------------------------------------------------------------------------------- */

/* When enabled, every launch of the reduction kernel is bracketed by HIP events on the ctx stream. */
int dvo_profile_enable(dvo_ctx* ctx, int on);
/* Sums since the last reset: per level, number of reduction launches, their total device time (ms)
 * and total pixels / depth-valid pixels... read with n = DVO_MAX_LEVELS entries each. Blocks (syncs). */
int dvo_profile_read(dvo_ctx* ctx, int64_t* launches, double* ms, int reset);
/* Issue-rate probe: what one VALU wave-instruction costs on this GPU right now, in ns per SIMD, with `waves_per_simd` waves
 * resident per SIMD.  Four instruction classes: simple (v_fmac / v_mul / v_add ..., operands in different register banks), simple
 * with both multiplicands in one VGPR bank, conversion / bit-field / compare / select / shift, v_rcp_f32.  out[0]: the composition
 * of one trip of the level-0 reduction's record loop (compiled into the probe; written to composition[4] when not NULL, so that the
 * caller can compare it with the kernel that was built: tools/isa_count.py), arranged as the compiled loop arranges it; out[1..4]:
 * each class alone; out[5]: the composition again, interleaved in small blocks (the order of the classes matters by ~8 %).  ALU-only
 * launches of `iterations` x ~500 instructions per wave.  bench.py multiplies out[0] by the kernel's instruction count to put the
 * issue-bound prediction beside the measured launch time (roofline.valu_issue). */
int dvo_debug_valu_rate(dvo_ctx* ctx, int waves_per_simd, int iterations, int* composition /* [4] or NULL */, double* ns_per_wave_instr /* [8]; an experiments build writes one value per probe mode: [128] */);
/* Event-pair stopwatch on the ctx stream. */
int dvo_timer_start(dvo_ctx* ctx);
int dvo_timer_stop(dvo_ctx* ctx, float* ms);

/* ---- experiment builds only ---------------------------------------------------------------------
 * `make EXTRA=-DDVO_EXPERIMENTS OUT=<elsewhere>` adds what the A/B scripts under tools/ use and the product library does not
 * contain: the rejected tap formulations of round 3 (csrc/dvo_exp_taps.inc, DVO_EXP_TAPS), the phase clocks of the persistent
 * kernel (DVO_SMALL_STAMPS), the launch-shape knobs DVO_XCD_REMAP / DVO_MAX_GROUPS / DVO_TARGET_BLOCKS and DVO_ST_DEBUG. */
#ifdef DVO_EXPERIMENTS
/* ctx created with DVO_SMALL_STAMPS set: shader-clock ticks thread 0 of each pair's workgroup of the persistent small-level
 * kernel spent in {record loops, wave reduction, cross-wave sum, control step, Student-t scale passes, whole kernel, -, -},
 * summed since the last reset: out[n_pairs][8]. */
int dvo_debug_small_stamps(dvo_ctx* ctx, int n_pairs, uint64_t* out, int reset);
#endif

#ifdef __cplusplus
}
#endif
#endif /* DVO_HIP_H */
