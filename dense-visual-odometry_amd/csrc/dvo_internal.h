// Internal structures shared by the C-ABI host code (dvo_abi.hip) and the kernels (dvo_kernels.hip).
// gfx950 only; not part of the public boundary (see include/dvo_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define DVO_BLOCK 256          // 4 wave64 per workgroup
#ifndef DVO_PX_PER_THREAD
#define DVO_PX_PER_THREAD 4    // pixels per lane per group (lane L takes pixels L, L+64, ... of the wave segment)
#endif
#define DVO_PX_PER_BLOCK (DVO_BLOCK * DVO_PX_PER_THREAD)
#define DVO_NACC 30            // 21 H + 6 J^T W r + sum w r^2 + N + sum r^2
#define DVO_STRIDE 32          // doubles per normal-equation record (== DVO_NE_STRIDE)

// One pyramid level of a frame slot: byte offsets inside the slot, image size, packed-plane pitch (pixels).
struct DvoLevelGeom {
    int w, h, pitch, npix;
    float wmax_f, hmax_f, pitch_f;      // (float)(w-1), (float)(h-1), (float)pitch: kernel arguments live in SGPRs and gfx950
                                        // has no scalar int->float conversion
    float pitch4_f;                     // 4 * pitch (bytes per packed row) as a float
    unsigned tap_bias;                  // 0x4B000000 where 4 * (h + 1) * pitch < 2^23 (tap byte offsets are then formed as the bit pattern
                                        // of a float32 in [2^23, 2^24): dvo_geo), else 0 (offsets by conversion and shift)
    int nseg;                           // ceil(npix / 1024): 1024-pixel segments of the valid-point list builder
    size_t off_gray, off_depth, off_packed;
    size_t off_list;                    // valid-point list: one 8-byte DvoPointRec per depth-valid pixel, raster order
};

// Valid-point list (built once per frame and level at ingest, read by every Gauss-Newton iteration): the pixels with
// depth != 0 (camera_model.py:195) in raster order.  z = (float)((double)depth * depth_scale) (camera_model.py:200-201)
// is iteration-invariant, so it is computed once here; pix = x | y << 12 | intensity << 24.
struct DvoPointRec {
    float z;
    uint32_t pix;
};
#define DVO_LIST_SLACK 2048             // bytes readable past the last record (a partly filled wave reads whole segments)
#define DVO_LIST_SEG 1024               // pixels per list-builder segment
#define DVO_LIST_LEVELS 8                // == DVO_MAX_LEVELS

// Level intrinsics K_L (RGBDCameraModel.at, camera_model.py:62-79) and the entries of inv(K_L)
// (camera_model.py:219) spelled out as in oracle/dvo_oracle.py:inverse_intrinsics.
struct DvoLevelCam {
    float fx, fy, cx, cy, s;
    float ia, ib, ic, id, ie;   // inv(K) = [[ia ib ic],[0 id ie],[0 0 1]]
    int skew0;                  // 1: skew == 0 (the zero terms of the reference's matrix products are skipped: exact)
};

struct DvoReduceParams {
    const uint8_t* slots;       // base of all frame slots
    size_t slot_stride;
    DvoLevelGeom g;
    DvoLevelCam cam;
    double depth_scale;
    const int* prev_slot;       // [n_pairs]
    const int* cur_slot;        // [n_pairs]
    const float* T;             // [n_pairs][12] row-major [R|t]
    const double* wparam;       // [n_pairs] lambda (student-t) or k (huber); may be null for WEIGHT_NONE
    const int* active;          // [n_pairs] device-mode: 0 => pair skipped at this level; null => all active
    double* partials;           // [n_pairs][blocks_per_pair][DVO_STRIDE]
    double* partials2;          // weight mode 4 (Student-t moment form): the r^2-weighted sums, same shape; else unused
    float lambda0;              // weight mode 5 (Student-t, predicted lambda): 1 / initial_sigma^2 of the first scale pass
    int blocks_per_pair;
    int xcd_remap;              // 1: XCD-aware logical block order
    int nt_records;             // 1: the valid-point records are loaded non-temporally (level 0: read once per launch)
    int groups_per_block;       // each workgroup reduces groups_per_block * DVO_PX_PER_BLOCK list records of one pair
    int n_pairs;
    const int* list_count;      // [n_slots][DVO_LIST_LEVELS] number of records in each slot's list
    int level;
    double* scale_out;          // Student-t device solve: the scale passes write one double per workgroup here ([n_pairs][blocks_per_pair])
    const double* scale1;       // ... and the following kernels derive lambda from those sums themselves (no control launch
    const double* scale2;       //     between the three passes of an iteration); all null: lambda comes from wparam
    float* rcache;              // Student-t scale passes: [n_pairs][g.npix] residual of every record of the current pose
                                // (NaN: out of bounds), written by the first pass and read by the second; or null
    const int* irls_done;       // Student-t scale passes with control launches: [n_pairs] 1 once a pair's IRLS has converged for this
                                // iteration (its workgroups exit at once); or null
};

struct DvoDenseParams {
    const uint8_t* slots;
    size_t slot_stride;
    DvoLevelGeom g;
    DvoLevelCam cam;
    double depth_scale;
    int prev_slot, cur_slot;
    float T[12];
    float* r;          // [npix] or null
    float* J;          // [npix][6] or null
    uint8_t* valid;    // [npix] or null
    float* uv;         // [npix][2] or null
};

// Per-pair Gauss-Newton state kept on the device by the fused solve (dvo_solve_batch).
struct DvoPairState {
    double q[4];        // estimate rotation, quaternion wxyz
    double t[3];        // estimate translation
    double err_prev;    // last accepted error (base_robust_dvo.py:158)
    double err_last;    // error of the last evaluated iteration
    double lambda;      // student-t: current lambda; huber: previous sigma (<=0: none yet)
    double lq[4], lt[3];   // motion prior: previous pair's transform (quaternion wxyz, translation)
    double oq[4], ot[3];   // motion prior: `old` of base_robust_dvo.py:154,:217-218
    int increased;      // err_increased_count (:159)
    int active;         // 1 while this level is still iterating for this pair
    int iters;          // total GN iterations evaluated
    int pad;
    double lam_hist[3]; // Student-t one-pass forms: the converged lambda of the last three iterations of this level ([0] newest)
    int lam_n;          // ... how many of them are valid (reset at every level)
    int pad2;
    double lam_err;     // ... and how far off (relative) the prediction of the last evaluated iteration's lambda was (1: none yet)
};

struct DvoSolveParams {
    const double* partials;     // [n_pairs][blocks_per_pair][DVO_STRIDE]
    const double* partials2;    // Student-t moment form: r^2-weighted sums of the same launch, or null
    int* inexact;               // Student-t one-pass forms: counter of (pair, iteration)s whose bound was not met, or null
    int predicted;              // 1: the reduction before this control launch was weighted with the predicted lambda (mode 5)
    double lambda0;             // Student-t: 1 / initial_sigma^2
    int blocks_per_pair;
    int n_pairs;
    DvoPairState* state;        // [n_pairs]
    float* T;                   // [n_pairs][12] updated in place
    int* active;                // [n_pairs]
    double* wparam;             // [n_pairs] next-iteration weight parameter (written for student-t / huber)
    double tolerance;
    int max_increased;
    int weight_mode;
    double weight_param;        // huber: k multiplier
    double inv_sigma;           // motion prior 1/sigma (base_robust_dvo.py:69-71), 0 = off
    int phase;                  // 0: normal-equation iteration; 1..: student-t scale passes
    int* irls_done;             // [n_pairs] see DvoReduceParams, or null
    double* trace;              // optional [n_pairs][max_trace][DVO_STRIDE+8] per-iteration records, or null
    int trace_stride, trace_cap;
    const int* trace_row;       // trace row of pair i (a re-solve of a compacted subset writes into its pairs' original rows), or null: i
    int* pair_flags;            // Student-t of the reference, per pair: |= 1 an iteration's one-pass bound was not met (the pair is solved
                                // again with the passes as written), |= 2 the IRLS of an iteration would not take exactly the two passes
                                // the per-iteration levels evaluate (the pair is solved again by the persistent kernel, which runs the
                                // reference's loop); or null
    const double* scale1;       // Student-t, passes as written without control launches between them: the per-workgroup sums of the two
    const double* scale2;       // scale passes of this iteration (the control step checks the two-pass shape from them), or null
    double st_tol;              // Student-t one-pass forms: an iteration is kept when the bound on what its form drops is within st_tol of
                                // the increment (2-norm); 5e-6 = half the parity tolerance
};

// Persistent small-level Gauss-Newton kernel (k_gn_small): ONE workgroup per frame pair runs every iteration of up to
// DVO_SMALL_LEVELS consecutive coarse pyramid levels — reduction, 6x6 solve, SE(3) update and stop rules — without
// leaving the kernel (no launch per iteration, no partial sums through HBM).
#define DVO_SMALL_LEVELS 8               // == DVO_MAX_LEVELS: the Student-t fallback runs EVERY level of its pairs here
struct DvoSmallParams {
    const uint8_t* slots;
    size_t slot_stride;
    const int* prev_slot;
    const int* cur_slot;
    const int* list_count;       // [n_slots][DVO_LIST_LEVELS]
    int n_levels;                // levels handled by this launch, coarse to fine: level_id[0] > level_id[1] > ...
    int level_id[DVO_SMALL_LEVELS];
    DvoLevelGeom g[DVO_SMALL_LEVELS];
    DvoLevelCam cam[DVO_SMALL_LEVELS];
    double depth_scale;
    int n_pairs;
    int level_iters[DVO_SMALL_LEVELS];   // iterations of level_id[k]
    DvoPairState* state;         // [n_pairs]
    float* T;                    // [n_pairs][12]
    int* active;                 // [n_pairs]
    double* wparam;              // [n_pairs]
    float* rcache;               // Student-t: [n_pairs][rcache_stride] residuals of the current pose, or null (recompute)
    size_t rcache_stride;
    double tolerance;
    int max_increased;
    int weight_mode;
    double weight_param;
    double inv_sigma;
    double lambda0;              // Student-t: 1 / initial_sigma^2
    int student_t_mean;          // Student-t scale = mean (proper) instead of the reference's sum (quirk Q3)
    int* irls_done;              // unused here (the IRLS loop runs inside the kernel); null
    double* trace;
    int trace_stride, trace_cap;
    const int* trace_row;        // see DvoSolveParams
    unsigned long long* stamps;  // diagnostic (DVO_SMALL_STAMPS): [n_pairs][8] shader-clock ticks thread 0 of the workgroup spent in
                                 // {record loops, wave reduction + barrier, cross-wave sum + barrier, control step + barrier,
                                 //  Student-t scale passes, whole kernel}; or null
};

struct DvoBatchIngestParams {
    uint8_t* slots;
    size_t slot_stride;
    const int* slot_ids;         // [n_frames] device
    const uint8_t* src_gray;     // [n_frames][h][w] device, or null if level 0 is already in the slots
    const uint16_t* src_depth;   // [n_frames][h][w] device
    int n_frames, levels;
    DvoLevelGeom g[8];
    double depth_scale;          // for the valid-point lists
    int* list_count;             // [n_slots][DVO_LIST_LEVELS]
    int* seg_counts;             // scratch [n_frames][sum of g[l].nseg]: per-segment counts, then exclusive offsets
    int allow_fused;             // 0: separate kernels for every level (DVO_INGEST_UNFUSED, tests)
};

// launch wrappers implemented in dvo_kernels.hip -------------------------------------------------
hipError_t dvo_launch_bgr_ingest(const uint8_t* bgr, const uint16_t* depth_in, int w, int h, uint8_t* gray,
                                 uint16_t* depth_out, double depth_scale, double max_distance, hipStream_t s);
hipError_t dvo_launch_batch_ingest(const DvoBatchIngestParams& p, hipStream_t s);
hipError_t dvo_launch_batch_bgr_ingest(const DvoBatchIngestParams& p, const uint8_t* bgr, const uint16_t* depth_in, unsigned d_max,
                                       hipStream_t s);
hipError_t dvo_launch_unpack(const uint32_t* packed, int w, int h, int pitch, int16_t* gx, int16_t* gy, hipStream_t s);
hipError_t dvo_launch_reduce(const DvoReduceParams& p, int variant, int weight_mode, int scale_pass, hipStream_t s);
hipError_t dvo_launch_scale_cached(const DvoReduceParams& p, hipStream_t s);
hipError_t dvo_launch_sum_partials(const double* partials, int blocks_per_pair, int n_pairs, double* out, hipStream_t s);
hipError_t dvo_launch_dense(const DvoDenseParams& p, int variant, hipStream_t s);
hipError_t dvo_launch_init_state(DvoPairState* st, float* T, const float* T_init, int n_pairs, double lambda0,
                                 const double* prior_qt, hipStream_t s);
hipError_t dvo_launch_level_begin(DvoPairState* st, int* active, double* wparam, int n_pairs, int weight_mode,
                                  double lambda0, int one_pass, hipStream_t s);
hipError_t dvo_launch_solve(const DvoSolveParams& p, hipStream_t s);
hipError_t dvo_launch_gn_small(const DvoSmallParams& p, int variant, int threads, hipStream_t s);
hipError_t dvo_launch_debug_divide(const float* a, const float* b, float* out, float* ref, int n, hipStream_t s);
hipError_t dvo_launch_finish(const DvoPairState* st, int n_pairs, double* xi_out, int* iters_out, double* err_out,
                             hipStream_t s);
hipError_t dvo_launch_valu_rate(int mode, int iterations, int n_cus, int waves_per_simd, float* sink, hipStream_t s);
int dvo_valu_rate_instructions(int mode, int* comp4);
#ifdef DVO_EXPERIMENTS
// experimental tap formulations of the level-0 reduction (dvo_exp_taps.inc; DVO_EXP_TAPS knob, tools/exp_taps.py)
hipError_t dvo_launch_exp_convert(const uint8_t* slots, size_t slot_stride, const DvoLevelGeom& g, const int* slot_ids, int n_slots,
                                  uint32_t* gradh, uint8_t* ipl, size_t plane_texels, hipStream_t s);
hipError_t dvo_launch_reduce_exp(const DvoReduceParams& p, int mode, const uint32_t* gradh, const uint8_t* ipl, size_t plane_texels,
                                 hipStream_t s);
#endif
