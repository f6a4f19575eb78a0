// C ABI of the MI355X direct-alignment core (see include/dvo_hip.h for the contract and the
// reference interfaces each entry point replaces).  Host-side only: context, device memory, launches.
#include "../../include/dvo_hip.h"
#include "dvo_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <climits>

static_assert(DVO_LIST_LEVELS == DVO_MAX_LEVELS, "the list-count table has one column per pyramid level");
static_assert(DVO_STRIDE == DVO_NE_STRIDE, "normal-equation record stride of the boundary");

namespace {

thread_local char g_create_error[512] = "";

struct SlotInfo {
    int w = 0, h = 0, levels = 0;
    bool ready = false;
    bool lists = false;     // valid-point lists built (needs the camera's depth_scale at upload time)
    uint64_t epoch = 0;     // bumped by every upload into the slot (the Student-t fallback checks its frames are still the enqueued ones)
};

struct HostStage {          // pinned staging for the small per-call arrays
    int* prev = nullptr;
    int* cur = nullptr;
    float* T = nullptr;
    double* wparam = nullptr;
    double* out = nullptr;          // [max_pairs][DVO_STRIDE]
    double* xi = nullptr;           // [max_pairs][6]
    int* iters = nullptr;
    double* err = nullptr;
};

}  // namespace

struct SolveKey {
    int n_pairs, w, h, levels, max_iterations, max_increased, variant, weight_mode, has_init, trace_cap, has_last;
    int st_exact;                       // Student-t of the reference: 0 = one-pass forms where they apply; 1 = the passes as written at the
                                        // per-iteration levels (two scale passes + the weighted pass, no one-pass form anywhere); 2 = EVERY
                                        // level in the persistent kernel, i.e. the reference's IRLS loop itself (the fallback for pairs
                                        // whose loop does not take exactly two passes)
    int trace_rows;                     // 1: trace rows through d_trace_row (re-solve of a compacted subset)
    double tolerance, weight_param, inv_sigma;
    const void* trace;
    const void* stream;
    int level_iters[DVO_MAX_LEVELS];    // iterations per level (dvo_set_level_iterations), or max_iterations at every level
};

struct dvo_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int max_w = 0, max_h = 0, max_levels = 0, n_slots = 0, max_pairs = 0;
    size_t slot_stride = 0;
    uint8_t* d_slots = nullptr;
    std::vector<SlotInfo> slots;
    // optional per-level iteration schedule (dvo_set_level_iterations)
    bool level_iters_set = false;
    int level_iters[DVO_MAX_LEVELS] = {0};
    // camera
    bool have_camera = false;
    float fx = 0, fy = 0, cx = 0, cy = 0, skew = 0;
    double depth_scale = 0;
    // scratch (device)
    int* d_prev = nullptr;
    int* d_cur = nullptr;
    float* d_T = nullptr;
    float* d_Tinit = nullptr;
    double* d_wparam = nullptr;
    int* d_active = nullptr;
    int* d_frame_slots = nullptr;
    int* h_frame_slots = nullptr;        // page-locked ring of DVO_FS_RING slot-id lists: an asynchronous upload reads its list later
    hipEvent_t ev_fs[4] = {nullptr, nullptr, nullptr, nullptr};
    bool fs_used[4] = {false, false, false, false};
    int fs_next = 0;
    int* d_list_count = nullptr;         // [n_slots][DVO_LIST_LEVELS] records in each slot's valid-point lists
    int* d_seg_counts = nullptr;         // list-builder scratch [n_slots][segments of all levels]
    int segs_per_frame = 0;
    double* d_partials = nullptr;
    size_t partials_doubles = 0;
    double* d_out = nullptr;
    DvoPairState* d_state = nullptr;
    double* d_xi = nullptr;
    int* d_iters = nullptr;
    double* d_err = nullptr;
    uint8_t* d_stage_img = nullptr;      // staging for host->device image uploads (bgr needs 3 B/px + depth)
    uint8_t* d_batch_stage = nullptr;    // staging of dvo_upload_frames (whole batches), grown on demand
    size_t batch_stage_bytes = 0;
    int* d_irls_done = nullptr;          // Student-t device solve with control launches: per pair, IRLS converged for this iteration
    double* d_scale1 = nullptr;          // Student-t device solve: per-workgroup sums of the two scale passes
    double* d_scale2 = nullptr;
    double* d_partials2 = nullptr;       // Student-t moment form: the r^2-weighted partial sums (same shape as d_partials)
    int* d_inexact = nullptr;            // ... iterations whose expansion bound was not met (device counter) and its page-locked copy
    int* h_inexact = nullptr;
    int* d_pair_flags = nullptr;         // Student-t of the reference: per pair, |= 1 one-pass bound not met, |= 2 IRLS not of the two-pass shape
    int* h_pair_flags = nullptr;         // (page-locked copy)
    int* d_trace_row = nullptr;          // trace row of each pair of a compacted re-solve
    uint64_t upload_epoch = 0;
    bool last_moment = false;            // the solve enqueued last is a Student-t solve of the reference (its per-pair flags and counters are read at fetch)
    SolveKey last_key;
    bool last_has_init = false;
    // what the last Student-t solve was enqueued with: dvo_solve_batch_fetch re-stages the flagged pairs of it (host copies, so that
    // T_init / the prior may change between enqueue and fetch; the frames themselves must not: epochs)
    struct LastSolve {
        std::vector<int> prev, cur;
        std::vector<uint64_t> ep_prev, ep_cur;
        std::vector<float> Tinit;
        std::vector<double> prior;
    } last;
    bool refetch_invalid = false;        // the last fetch solved flagged pairs again: the device buffers now hold THAT compacted solve
    int64_t st_fallbacks = 0;            // fetched solves in which at least one pair had to be solved again
    int64_t st_fallback_pairs = 0;       // ... and how many pairs that were
    unsigned long long* d_small_stamps = nullptr;   // DVO_SMALL_STAMPS: [max_pairs][8] phase clocks of k_gn_small (diagnostic)
    float* d_rcache = nullptr;           // Student-t device solve: residuals kept between the scale passes [max_pairs][max_w*max_h]
    uint32_t* d_exp_gradh = nullptr;     // DVO_EXP_TAPS side planes [n_slots][exp_plane_texels]: {f16 gx, f16 gy} / intensity
    uint8_t* d_exp_ipl = nullptr;
    size_t exp_plane_texels = 0;
    float* d_dense = nullptr;            // dense writer outputs: r[npix] J[6 npix] uv[2 npix] + valid
    double* d_trace = nullptr;
    int trace_cap = 0;
    double* d_prior = nullptr;           // [max_pairs][7] previous-pair transforms for the motion prior
    double* h_prior = nullptr;
    double prior_sigma = 0.0;            // <= 0: prior off
    bool prior_has_last = false;
    HostStage hs;
    // profiling
    bool profiling = false;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    hipEvent_t tm_a = nullptr, tm_b = nullptr;
    hipEvent_t ev_staging = nullptr;
    hipEvent_t ev_order = nullptr;       // dvo_wait_for: marks "everything enqueued on this ctx so far"
    // dvo_set_upload_peer: host->device copies of dvo_upload_frames go to the peer's copy stream (one in-order queue
    // for all contexts that share the PCIe link), the ctx stream waits for them by event
    hipStream_t copy_stream = nullptr;
    dvo_ctx* upload_peer = nullptr;
    hipEvent_t ev_uploaded = nullptr;    // recorded on the copy stream after this ctx's copies
    hipEvent_t ev_ingested = nullptr;    // recorded on the ctx stream after the ingest kernels that read the staging area
    bool ingested_recorded = false;
    bool staging_in_flight = false;
    struct PendingEv { hipEvent_t a, b; int level; };
    std::vector<PendingEv> pending;
    std::vector<hipEvent_t> ev_pool;
    int64_t prof_launches[DVO_MAX_LEVELS] = {0};
    double prof_ms[DVO_MAX_LEVELS] = {0};
    // hipGraph cache of the device-side solve
    struct GraphEntry { SolveKey key; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    bool graphs_enabled = true;
    // tuning / debugging knobs, read from the environment ONCE at dvo_create (never on the per-call path; a captured
    // solve graph bakes them in, so they must not change under it)
    struct Knobs {
        int xcd_remap = 0;          // DVO_XCD_REMAP=1: XCD-aware logical block order (measured slower)
        int max_groups = 48;        // DVO_MAX_GROUPS: record groups (1024 records) per reduction workgroup at most.  A trade of speed against the
                                    // length of the per-lane float32 sums (profiles/r04_experiments.txt [9], [14]; 256 VGA pairs per launch;
                                    // error = worst increment of the reference run's 89 level-0 iterations against its float64 sums, bound 1e-5):
                                    //   16: baseline, 1.5e-6     24: +0.5 %, 3.4e-6     32: +0.9 %, 3.9e-6     48: +1.1-1.5 %, 5.1e-6     96: 4.3e-6
                                    // (median 3.2e-7 -> 5.5e-7: it grows like the square root of the lane's length).  Each workgroup pays its
                                    // prologue and its 30-term float64 reduction once, and with four streams in flight nobody waits for a
                                    // launch's tail.  A second level for the six J^T W r sums (LDS, flushed every 8 / 16 trips) brings the error
                                    // to 2.1e-6 / 3.1e-6 and costs 1.4 % / 0.6 %: not kept.  tests: ..._production_launch_shape_...
        int target_blocks = 1024;   // DVO_TARGET_BLOCKS
        int nt_levels = 1;          // DVO_NT_LEVELS: bit l = level l's valid-point records are loaded non-temporally (re-checked in the job: level 0 only)
        bool st_solve_phases = false;   // DVO_ST_SOLVE_PHASES: Student-t device solve with control launches between the passes
        bool no_rcache = false;         // DVO_NO_RCACHE: second Student-t scale pass recomputes the residuals
        bool ingest_unfused = false;    // DVO_INGEST_UNFUSED: separate ingest kernels for every level (tests)
        int small_max_pix = 19200;      // DVO_SMALL_MAX_PIX: levels with at most this many pixels run in the persistent per-pair
                                        // kernel k_gn_small (0: every level takes the per-iteration launches)
        int small_threads = 0;          // DVO_SMALL_THREADS: workgroup size of k_gn_small (0: chosen from the batch size)
        bool st_exact = false;          // DVO_ST_EXACT: the reference's Student-t always as its three passes (no moment form)
        bool st_debug = false;          // DVO_ST_DEBUG (experiment builds): fallback decisions on stderr
        double st_tol = 5e-6;           // (experiment builds: DVO_ST_TOL) one-pass Student-t iterations are kept within this of the increment
        int exp_taps = 0;               // DVO_EXP_TAPS=1..4: dvo_normal_equations (V1, unweighted) runs an experimental tap
                                        // formulation of the reduction (dvo_exp_taps.inc; tools/exp_taps.py); never the solve path
    } knobs;
    char err[512] = "";
};

namespace {

int fail(dvo_ctx* ctx, int code, const char* fmt, ...)
{
    char* dst = ctx ? ctx->err : g_create_error;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(ctx, (_e == hipErrorOutOfMemory) ? DVO_ERR_OUT_OF_MEMORY : DVO_ERR_HIP,         \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);     \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Layout of one frame slot for a w x h image: per level gray | depth | packed, each 256-byte aligned with
// 64 bytes of slack so 8-byte vector loads of the last pixels stay inside the slot.
void level_geometry(int w, int h, int levels, DvoLevelGeom* g, size_t* total)
{
    size_t off = 0;
    int lw = w, lh = h;
    for (int l = 0; l < levels; ++l) {
        g[l].w = lw;
        g[l].h = lh;
        g[l].npix = lw * lh;
        g[l].pitch = (int)align_up((size_t)lw + 2, 16);
        g[l].wmax_f = (float)(lw - 1);
        g[l].hmax_f = (float)(lh - 1);
        g[l].pitch_f = (float)g[l].pitch;
        g[l].pitch4_f = 4.0f * (float)g[l].pitch;
        g[l].tap_bias = (4.0 * (double)(lh + 1) * (double)g[l].pitch < 8388608.0) ? 0x4B000000u : 0u;
        g[l].nseg = (lw * lh + DVO_LIST_SEG - 1) / DVO_LIST_SEG;
        g[l].off_gray = off;
        off = align_up(off + (size_t)lw * lh + 64, 256);
        g[l].off_depth = off;
        off = align_up(off + (size_t)lw * lh * 2 + 64, 256);
        g[l].off_packed = off;
        off = align_up(off + (size_t)(lh + 1) * g[l].pitch * 4 + 64, 256);
        g[l].off_list = off;
        off = align_up(off + (size_t)lw * lh * sizeof(DvoPointRec) + DVO_LIST_SLACK + 64, 256);
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
    }
    if (total) *total = off;
}

// RGBDCameraModel.at (camera_model.py:62-79) in float32, and inv(K_L) as in oracle inverse_intrinsics
DvoLevelCam level_camera(const dvo_ctx* c, int level)
{
    DvoLevelCam k;
    if (level == 0) {
        k.fx = c->fx; k.fy = c->fy; k.cx = c->cx; k.cy = c->cy; k.s = c->skew;
    } else {
        // scale_matrix (float32) @ K: entries sc*fx, sc*s, sc*cx + off*1, sc*fy, sc*cy + off
        const float sc = (float)std::ldexp(1.0, -level);
        const float off = (float)(std::ldexp(1.0, -level - 1) - 0.5);
        k.fx = sc * c->fx;
        k.s = sc * c->skew;
        k.cx = sc * c->cx + off;
        k.fy = sc * c->fy;
        k.cy = sc * c->cy + off;
    }
    const volatile float one = 1.0f;
    volatile float ia = one / k.fx;
    volatile float id = one / k.fy;
    volatile float ie = (-k.cy) / k.fy;
    volatile float sd = k.s * id;
    volatile float ib = (-sd) / k.fx;
    volatile float se = k.s * ie;
    volatile float t = (-se) - k.cx;
    volatile float ic = t / k.fx;
    k.ia = ia; k.ib = ib; k.ic = ic; k.id = id; k.ie = ie;
    k.skew0 = (k.s == 0.0f && k.ib == 0.0f) ? 1 : 0;
    return k;
}

// ... for the kernels: the fast instantiations (template argument SKEW = false) skip the zero-skew products AND form tap offsets as float
// bit patterns (dvo_geo), which needs the level's tap_bias; a level without it (4 (h + 1) pitch >= 2^23) takes the general ones
DvoLevelCam level_camera(const dvo_ctx* c, int level, const DvoLevelGeom& g)
{
    DvoLevelCam k = level_camera(c, level);
    if (g.tap_bias == 0u) k.skew0 = 0;
    return k;
}

int check_slot(dvo_ctx* ctx, int slot, const char* what)
{
    if (slot < 0 || slot >= ctx->n_slots) return fail(ctx, DVO_ERR_INVALID_ARG, "%s: slot %d out of range [0,%d)", what, slot, ctx->n_slots);
    if (!ctx->slots[slot].ready) return fail(ctx, DVO_ERR_BAD_STATE, "%s: slot %d holds no frame", what, slot);
    return DVO_OK;
}

int check_pairs(dvo_ctx* ctx, int n_pairs, const int* prev_slot, const int* cur_slot, int level_needed, int* w, int* h)
{
    if (!ctx->have_camera) return fail(ctx, DVO_ERR_BAD_STATE, "camera not set (dvo_set_camera)");
    if (n_pairs <= 0 || n_pairs > ctx->max_pairs) return fail(ctx, DVO_ERR_INVALID_ARG, "n_pairs %d outside [1,%d]", n_pairs, ctx->max_pairs);
    if (!prev_slot || !cur_slot) return fail(ctx, DVO_ERR_INVALID_ARG, "null slot arrays");
    for (int i = 0; i < n_pairs; ++i) {
        int rc = check_slot(ctx, prev_slot[i], "prev");
        if (rc == DVO_OK && !ctx->slots[prev_slot[i]].lists)
            rc = fail(ctx, DVO_ERR_BAD_STATE, "slot %d was uploaded before dvo_set_camera (or under another depth_scale): upload it again", prev_slot[i]);
        if (rc) return rc;
        rc = check_slot(ctx, cur_slot[i], "cur");
        if (rc) return rc;
        const SlotInfo& a = ctx->slots[prev_slot[i]];
        const SlotInfo& b = ctx->slots[cur_slot[i]];
        const SlotInfo& f = ctx->slots[prev_slot[0]];
        if (a.w != f.w || a.h != f.h || b.w != f.w || b.h != f.h)
            return fail(ctx, DVO_ERR_INVALID_ARG, "pair %d: frames of one call must share one size (%dx%d vs %dx%d)", i, a.w, a.h, f.w, f.h);
        if (a.levels <= level_needed || b.levels <= level_needed)
            return fail(ctx, DVO_ERR_INVALID_ARG, "pair %d: level %d not built (slots have %d/%d levels)", i, level_needed, a.levels, b.levels);
    }
    *w = ctx->slots[prev_slot[0]].w;
    *h = ctx->slots[prev_slot[0]].h;
    return DVO_OK;
}

int blocks_for(int npix) { return (npix + DVO_PX_PER_BLOCK - 1) / DVO_PX_PER_BLOCK; }

hipEvent_t pool_event(dvo_ctx* ctx)
{
    if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void drain_profile(dvo_ctx* ctx)
{
    for (auto& pe : ctx->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pe.b) == hipSuccess && hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            ctx->prof_launches[pe.level] += 1;
            ctx->prof_ms[pe.level] += ms;
        }
        ctx->ev_pool.push_back(pe.a);
        ctx->ev_pool.push_back(pe.b);
    }
    ctx->pending.clear();
}

// one launch of the reduction kernel, optionally bracketed by events
hipError_t launch_reduce_profiled(dvo_ctx* ctx, const DvoReduceParams& p, int level, int variant, int weight_mode, int scale_pass,
                                  int exp_mode = 0)
{
    hipEvent_t a = nullptr, b = nullptr;
    if (ctx->profiling) {
        if (ctx->pending.size() > 4096) drain_profile(ctx);
        a = pool_event(ctx);
        b = pool_event(ctx);
        if (a && b) (void)hipEventRecord(a, ctx->stream);
    }
#ifdef DVO_EXPERIMENTS
    hipError_t e = exp_mode ? dvo_launch_reduce_exp(p, exp_mode, ctx->d_exp_gradh, ctx->d_exp_ipl, ctx->exp_plane_texels, ctx->stream)
                            : dvo_launch_reduce(p, variant, weight_mode, scale_pass, ctx->stream);
#else
    (void)exp_mode;
    hipError_t e = dvo_launch_reduce(p, variant, weight_mode, scale_pass, ctx->stream);
#endif
    if (ctx->profiling && a && b) {
        (void)hipEventRecord(b, ctx->stream);
        ctx->pending.push_back({a, b, level});
    }
    return e;
}

int fill_reduce_params(dvo_ctx* ctx, int level, int n_pairs, int w, int h, DvoReduceParams* p)
{
    DvoLevelGeom g[DVO_MAX_LEVELS];
    level_geometry(w, h, level + 1, g, nullptr);
    p->slots = ctx->d_slots;
    p->slot_stride = ctx->slot_stride;
    p->g = g[level];
    p->cam = level_camera(ctx, level, g[level]);
    p->depth_scale = ctx->depth_scale;
    p->prev_slot = ctx->d_prev;
    p->cur_slot = ctx->d_cur;
    p->T = ctx->d_T;
    p->wparam = ctx->d_wparam;
    p->active = nullptr;
    p->nt_records = (ctx->knobs.nt_levels >> level) & 1;      // the finest level's records go around the caches (read once per launch)
    p->partials = ctx->d_partials;
    p->partials2 = nullptr;
    p->lambda0 = 0.04f;
    p->list_count = ctx->d_list_count;
    p->level = level;
    p->rcache = nullptr;
    p->irls_done = nullptr;
    p->scale_out = nullptr;
    p->scale1 = nullptr;
    p->scale2 = nullptr;
    p->xcd_remap = ctx->knobs.xcd_remap;
    // more pixels per workgroup on the big levels amortises the 30-term cross-lane reduction
    // groups so that the launch still has a few thousand workgroups: (pixels of the batch / 1024) / ~4096
    {
        const long long total_groups = (long long)n_pairs * ((g[level].npix + DVO_PX_PER_BLOCK - 1) / DVO_PX_PER_BLOCK);
        const int gmax = ctx->knobs.max_groups;
        const int target = ctx->knobs.target_blocks;   // measured on MI355X: small levels gain from fewer, longer workgroups (L2 27 -> 21 us)
        int gp = (int)(total_groups / target);
        if (gp < 1) gp = 1;
        if (gp > gmax) gp = gmax;
        p->groups_per_block = gp;
    }
    p->blocks_per_pair = (g[level].npix + DVO_PX_PER_BLOCK * p->groups_per_block - 1) / (DVO_PX_PER_BLOCK * p->groups_per_block);
    p->n_pairs = n_pairs;
    if ((size_t)n_pairs * p->blocks_per_pair * DVO_STRIDE > ctx->partials_doubles)
        return fail(ctx, DVO_ERR_INVALID_ARG, "internal: partial buffer too small");
    return DVO_OK;
}

// Slot ids of an upload go to the device with an asynchronous copy from a page-locked list; the lists form a ring so
// that a following upload on the same ctx does not overwrite one the copy engine has not read yet.
int stage_frame_slots(dvo_ctx* ctx, const int* slots, int n)
{
    const int k = ctx->fs_next;
    ctx->fs_next = (k + 1) & 3;
    if (ctx->fs_used[k]) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_fs[k]));
    int* hbuf = ctx->h_frame_slots + (size_t)k * ctx->n_slots;
    memcpy(hbuf, slots, (size_t)n * sizeof(int));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_frame_slots, hbuf, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fs[k], ctx->stream));
    ctx->fs_used[k] = true;
    return DVO_OK;
}

int build_pyramid_on_device(dvo_ctx* ctx, int slot, int w, int h, int levels)
{
    // level 0 is already in the slot; the batched ingest kernels build the rest (n_frames = 1)
    DvoBatchIngestParams p;
    p.slots = ctx->d_slots;
    p.slot_stride = ctx->slot_stride;
    p.slot_ids = ctx->d_frame_slots;
    p.src_gray = nullptr;
    p.src_depth = nullptr;
    p.n_frames = 1;
    p.levels = levels;
    p.depth_scale = ctx->depth_scale;
    p.list_count = ctx->d_list_count;
    p.seg_counts = ctx->d_seg_counts;
    p.allow_fused = ctx->knobs.ingest_unfused ? 0 : 1;
    level_geometry(w, h, levels, p.g, nullptr);
    int rc = stage_frame_slots(ctx, &slot, 1);
    if (rc) return rc;
    HIP_TRY(ctx, dvo_launch_batch_ingest(p, ctx->stream));
    return DVO_OK;
}

int prepare_slot(dvo_ctx* ctx, int slot, int w, int h, int levels)
{
    if (slot < 0 || slot >= ctx->n_slots) return fail(ctx, DVO_ERR_INVALID_ARG, "slot %d out of range [0,%d)", slot, ctx->n_slots);
    if (w <= 0 || h <= 0 || w > ctx->max_w || h > ctx->max_h)
        return fail(ctx, DVO_ERR_INVALID_ARG, "image %dx%d outside the ctx maximum %dx%d", w, h, ctx->max_w, ctx->max_h);
    if (levels < 1 || levels > ctx->max_levels) return fail(ctx, DVO_ERR_INVALID_ARG, "levels %d outside [1,%d]", levels, ctx->max_levels);
    if ((w >> (levels - 1)) < 1 || (h >> (levels - 1)) < 1) return fail(ctx, DVO_ERR_INVALID_ARG, "%d levels is too many for a %dx%d image", levels, w, h);
    SlotInfo& s = ctx->slots[slot];
    if (s.w != w || s.h != h || s.levels != levels) {
        // geometry of the slot changes: the zero pad row/column of the packed planes must be re-established
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_slots + (size_t)slot * ctx->slot_stride, 0, ctx->slot_stride, ctx->stream));
    }
    s.w = w; s.h = h; s.levels = levels; s.ready = false;
    s.epoch = ++ctx->upload_epoch;
    s.lists = ctx->have_camera;      // the valid-point lists built at ingest hold z = depth * depth_scale
    return DVO_OK;
}

}  // namespace

// ==================================================================================================
extern "C" {

const char* dvo_last_error(const dvo_ctx* ctx) { return ctx ? ctx->err : g_create_error; }

int dvo_create(int device, int max_w, int max_h, int max_levels, int n_slots, int max_pairs, dvo_ctx** out)
{
    if (!out) return fail(nullptr, DVO_ERR_INVALID_ARG, "dvo_create: out is null");
    *out = nullptr;
    if (max_w <= 0 || max_h <= 0 || max_levels < 1 || max_levels > DVO_MAX_LEVELS || n_slots < 2 || max_pairs < 1)
        return fail(nullptr, DVO_ERR_INVALID_ARG, "dvo_create: bad sizes (max %dx%d, levels %d, slots %d, pairs %d)", max_w, max_h, max_levels, n_slots, max_pairs);
    // the kernels form tap offsets y * pitch + x in float32 and pack pixel coordinates in 16 bits
    // the kernels form tap offsets y * pitch + x in float32 and the valid-point records hold x and y in 12 bits each
    if ((double)(max_h + 1) * (double)align_up((size_t)max_w + 2, 16) >= 16777216.0 || max_w > 4096 || max_h > 4096)
        return fail(nullptr, DVO_ERR_INVALID_ARG, "dvo_create: %dx%d exceeds the supported image size (at most 4096x4096 and (h+1)*pitch below 2^24)", max_w, max_h);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, DVO_ERR_NO_DEVICE, "dvo_create: no HIP device (%s); there is no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, DVO_ERR_INVALID_ARG, "dvo_create: device %d outside [0,%d)", device, ndev);
    dvo_ctx* ctx = new (std::nothrow) dvo_ctx();
    if (!ctx) return fail(nullptr, DVO_ERR_OUT_OF_MEMORY, "dvo_create: host allocation failed");
#define CREATE_TRY(expr)                                                                           \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            int rc = fail(nullptr, (_e == hipErrorOutOfMemory) ? DVO_ERR_OUT_OF_MEMORY : DVO_ERR_HIP, \
                          "dvo_create: %s failed: %s", #expr, hipGetErrorString(_e));              \
            dvo_destroy(ctx);                                                                      \
            return rc;                                                                             \
        }                                                                                          \
    } while (0)
    ctx->device = device;
    ctx->max_w = max_w; ctx->max_h = max_h; ctx->max_levels = max_levels; ctx->n_slots = n_slots; ctx->max_pairs = max_pairs;
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    DvoLevelGeom g[DVO_MAX_LEVELS];
    int lv = max_levels;
    while (lv > 1 && ((max_w >> (lv - 1)) < 1 || (max_h >> (lv - 1)) < 1)) --lv;
    level_geometry(max_w, max_h, lv, g, &ctx->slot_stride);
    ctx->slot_stride = align_up(ctx->slot_stride, 4096);
    ctx->slots.resize(n_slots);
    CREATE_TRY(hipMalloc((void**)&ctx->d_slots, ctx->slot_stride * (size_t)n_slots));
    CREATE_TRY(hipMemsetAsync(ctx->d_slots, 0, ctx->slot_stride * (size_t)n_slots, ctx->stream));
    const size_t np = (size_t)max_pairs;
    CREATE_TRY(hipMalloc((void**)&ctx->d_prev, np * sizeof(int)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_cur, np * sizeof(int)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_T, np * 12 * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_Tinit, np * 12 * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_wparam, np * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_active, np * sizeof(int)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_irls_done, np * sizeof(int)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_pair_flags, np * sizeof(int)));
    CREATE_TRY(hipMemset(ctx->d_pair_flags, 0, np * sizeof(int)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->h_pair_flags, np * sizeof(int)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_trace_row, np * sizeof(int)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_frame_slots, (size_t)n_slots * sizeof(int)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->h_frame_slots, 4 * (size_t)n_slots * sizeof(int)));
    for (int i = 0; i < 4; ++i) CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_fs[i], hipEventDisableTiming));
    for (int l = 0; l < lv; ++l) ctx->segs_per_frame += g[l].nseg;
    CREATE_TRY(hipMalloc((void**)&ctx->d_list_count, (size_t)n_slots * DVO_LIST_LEVELS * sizeof(int)));
    CREATE_TRY(hipMemsetAsync(ctx->d_list_count, 0, (size_t)n_slots * DVO_LIST_LEVELS * sizeof(int), ctx->stream));
    CREATE_TRY(hipMalloc((void**)&ctx->d_seg_counts, (size_t)n_slots * (size_t)ctx->segs_per_frame * sizeof(int)));
    ctx->partials_doubles = np * (size_t)blocks_for(max_w * max_h) * DVO_STRIDE;
    CREATE_TRY(hipMalloc((void**)&ctx->d_partials, ctx->partials_doubles * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_out, np * DVO_STRIDE * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_state, np * sizeof(DvoPairState)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_xi, np * 6 * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_iters, np * sizeof(int)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_err, np * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_prior, np * 7 * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->h_prior, np * 7 * sizeof(double)));
    CREATE_TRY(hipMalloc((void**)&ctx->d_stage_img, (size_t)max_w * max_h * 5 + 256));
    CREATE_TRY(hipMalloc((void**)&ctx->d_dense, (size_t)max_w * max_h * (9 * sizeof(float) + 1) + 256));
    CREATE_TRY(hipHostMalloc((void**)&ctx->hs.prev, np * sizeof(int)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->hs.cur, np * sizeof(int)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->hs.T, np * 12 * sizeof(float)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->hs.wparam, np * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->hs.out, np * DVO_STRIDE * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->hs.xi, np * 6 * sizeof(double)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->hs.iters, np * sizeof(int)));
    CREATE_TRY(hipHostMalloc((void**)&ctx->hs.err, np * sizeof(double)));
    CREATE_TRY(hipEventCreate(&ctx->tm_a));
    CREATE_TRY(hipEventCreate(&ctx->tm_b));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_staging, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_uploaded, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_ingested, hipEventDisableTiming));
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    CREATE_TRY(hipStreamSynchronize(ctx->stream));
    {
        // product knobs (listed in include/dvo_hip.h, "Environment"): read ONCE here, never on a per-call path
        const char* e = getenv("DVO_GRAPH");
        ctx->graphs_enabled = !(e && e[0] == '0');
        ctx->knobs.st_solve_phases = getenv("DVO_ST_SOLVE_PHASES") != nullptr;
        ctx->knobs.no_rcache = getenv("DVO_NO_RCACHE") != nullptr;
        ctx->knobs.ingest_unfused = getenv("DVO_INGEST_UNFUSED") != nullptr;
        ctx->knobs.st_exact = getenv("DVO_ST_EXACT") != nullptr;
        e = getenv("DVO_SMALL_MAX_PIX");
        if (e) ctx->knobs.small_max_pix = atoi(e);
        e = getenv("DVO_SMALL_THREADS");
        if (e && (atoi(e) == 256 || atoi(e) == 512 || atoi(e) == 1024)) ctx->knobs.small_threads = atoi(e);
        e = getenv("DVO_MAX_GROUPS");
        if (e && atoi(e) > 0) ctx->knobs.max_groups = atoi(e);
#ifdef DVO_EXPERIMENTS
        // tuning / diagnostic knobs of experiment builds (make EXTRA=-DDVO_EXPERIMENTS): not in the product library
        e = getenv("DVO_XCD_REMAP");
        ctx->knobs.xcd_remap = (e && e[0] == '1') ? 1 : 0;
        e = getenv("DVO_TARGET_BLOCKS");
        if (e && atoi(e) > 0) ctx->knobs.target_blocks = atoi(e);
        e = getenv("DVO_NT_LEVELS");
        if (e) ctx->knobs.nt_levels = atoi(e);
        ctx->knobs.st_debug = getenv("DVO_ST_DEBUG") != nullptr;
        e = getenv("DVO_ST_TOL");
        if (e && atof(e) > 0.0) ctx->knobs.st_tol = atof(e);
        if (getenv("DVO_SMALL_STAMPS") != nullptr) {
            CREATE_TRY(hipMalloc((void**)&ctx->d_small_stamps, (size_t)max_pairs * 8 * sizeof(unsigned long long)));
            CREATE_TRY(hipMemset(ctx->d_small_stamps, 0, (size_t)max_pairs * 8 * sizeof(unsigned long long)));
        }
        e = getenv("DVO_EXP_TAPS");
        if (e) ctx->knobs.exp_taps = atoi(e);
#endif
    }
#undef CREATE_TRY
    *out = ctx;
    return DVO_OK;
}

void dvo_destroy(dvo_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    drain_profile(ctx);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (auto& g : ctx->graphs) (void)hipGraphExecDestroy(g.exec);
    if (ctx->tm_a) (void)hipEventDestroy(ctx->tm_a);
    if (ctx->tm_b) (void)hipEventDestroy(ctx->tm_b);
    if (ctx->ev_staging) (void)hipEventDestroy(ctx->ev_staging);
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    for (int i = 0; i < 4; ++i) if (ctx->ev_fs[i]) (void)hipEventDestroy(ctx->ev_fs[i]);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->ev_uploaded) (void)hipEventDestroy(ctx->ev_uploaded);
    if (ctx->ev_ingested) (void)hipEventDestroy(ctx->ev_ingested);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    void* dev[] = {ctx->d_slots, ctx->d_prev, ctx->d_cur, ctx->d_T, ctx->d_Tinit, ctx->d_wparam, ctx->d_active, ctx->d_partials,
                   ctx->d_out, ctx->d_state, ctx->d_xi, ctx->d_iters, ctx->d_err, ctx->d_stage_img, ctx->d_dense, ctx->d_trace, ctx->d_frame_slots, ctx->d_prior, ctx->d_batch_stage,
                   ctx->d_list_count, ctx->d_seg_counts, ctx->d_rcache, ctx->d_scale1, ctx->d_scale2, ctx->d_irls_done, ctx->d_pair_flags, ctx->d_trace_row,
                   ctx->d_exp_gradh, ctx->d_exp_ipl, ctx->d_partials2, ctx->d_inexact, ctx->d_small_stamps};
    for (void* p : dev) if (p) (void)hipFree(p);
    void* host[] = {ctx->hs.prev, ctx->hs.cur, ctx->hs.T, ctx->hs.wparam, ctx->hs.out, ctx->hs.xi, ctx->hs.iters, ctx->hs.err, ctx->h_frame_slots, ctx->h_prior,
                    ctx->h_inexact, ctx->h_pair_flags};
    for (void* p : host) if (p) (void)hipHostFree(p);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int dvo_set_stream(dvo_ctx* ctx, void* hip_stream)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return DVO_OK;
}

int dvo_synchronize(dvo_ctx* ctx)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DVO_OK;
}

int dvo_set_camera(dvo_ctx* ctx, float fx, float fy, float cx, float cy, float skew, double depth_scale)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!(fx > 0.f) || !(fy > 0.f) || !(depth_scale >= 0.0) || !std::isfinite(cx) || !std::isfinite(cy) || !std::isfinite(skew))
        return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_set_camera: need fx, fy > 0 and depth_scale >= 0");
    if (!ctx->have_camera || depth_scale != ctx->depth_scale)
        for (SlotInfo& si : ctx->slots) si.lists = false;      // their valid-point lists hold z for the old scale
    // a captured solve graph holds the level intrinsics (and the skew / no-skew kernel choice) as kernel arguments:
    // graphs recorded under the old camera must not be replayed
    if (!ctx->graphs.empty()) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (auto& g : ctx->graphs) (void)hipGraphExecDestroy(g.exec);
        ctx->graphs.clear();
    }
    ctx->fx = fx; ctx->fy = fy; ctx->cx = cx; ctx->cy = cy; ctx->skew = skew; ctx->depth_scale = depth_scale;
    ctx->have_camera = true;
    return DVO_OK;
}

int dvo_get_level_intrinsics(const dvo_ctx* ctx, int level, float* k)
{
    if (!ctx || !k) return DVO_ERR_INVALID_ARG;
    if (!ctx->have_camera) return fail(const_cast<dvo_ctx*>(ctx), DVO_ERR_BAD_STATE, "camera not set");
    if (level < 0 || level >= 30) return fail(const_cast<dvo_ctx*>(ctx), DVO_ERR_INVALID_ARG, "level %d out of range", level);
    DvoLevelCam c = level_camera(ctx, level);
    const float m[12] = {c.fx, c.s, c.cx, 0.f, 0.f, c.fy, c.cy, 0.f, 0.f, 0.f, 1.f, 0.f};
    memcpy(k, m, sizeof(m));
    return DVO_OK;
}

int dvo_upload_frame(dvo_ctx* ctx, int slot, const uint8_t* gray, const uint16_t* depth, int w, int h, int levels)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!gray || !depth) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_upload_frame: null image");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = prepare_slot(ctx, slot, w, h, levels);
    if (rc) return rc;
    DvoLevelGeom g[DVO_MAX_LEVELS];
    level_geometry(w, h, levels, g, nullptr);
    uint8_t* base = ctx->d_slots + (size_t)slot * ctx->slot_stride;
    HIP_TRY(ctx, hipMemcpyAsync(base + g[0].off_gray, gray, (size_t)w * h, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(base + g[0].off_depth, depth, (size_t)w * h * 2, hipMemcpyHostToDevice, ctx->stream));
    rc = build_pyramid_on_device(ctx, slot, w, h, levels);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // host buffers are only valid for the call
    ctx->slots[slot].ready = true;
    return DVO_OK;
}

int dvo_upload_frame_device(dvo_ctx* ctx, int slot, const void* d_gray, const void* d_depth, int w, int h, int levels)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!d_gray || !d_depth) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_upload_frame_device: null image");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = prepare_slot(ctx, slot, w, h, levels);
    if (rc) return rc;
    DvoLevelGeom g[DVO_MAX_LEVELS];
    level_geometry(w, h, levels, g, nullptr);
    uint8_t* base = ctx->d_slots + (size_t)slot * ctx->slot_stride;
    HIP_TRY(ctx, hipMemcpyAsync(base + g[0].off_gray, d_gray, (size_t)w * h, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(base + g[0].off_depth, d_depth, (size_t)w * h * 2, hipMemcpyDeviceToDevice, ctx->stream));
    rc = build_pyramid_on_device(ctx, slot, w, h, levels);
    if (rc) return rc;
    ctx->slots[slot].ready = true;
    return DVO_OK;
}

int dvo_upload_frames_device(dvo_ctx* ctx, int n_frames, const int* slots, const void* d_gray, const void* d_depth,
                             int w, int h, int levels)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!slots || !d_gray || !d_depth) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_upload_frames_device: null argument");
    if (n_frames < 1 || n_frames > ctx->n_slots) return fail(ctx, DVO_ERR_INVALID_ARG, "n_frames %d outside [1,%d]", n_frames, ctx->n_slots);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n_frames; ++i) {
        for (int j = 0; j < i; ++j)
            if (slots[j] == slots[i]) return fail(ctx, DVO_ERR_INVALID_ARG, "slot %d listed twice", slots[i]);
        int rc = prepare_slot(ctx, slots[i], w, h, levels);
        if (rc) return rc;
    }
    DvoBatchIngestParams p;
    p.slots = ctx->d_slots;
    p.slot_stride = ctx->slot_stride;
    p.slot_ids = ctx->d_frame_slots;
    p.src_gray = static_cast<const uint8_t*>(d_gray);
    p.src_depth = static_cast<const uint16_t*>(d_depth);
    p.n_frames = n_frames;
    p.levels = levels;
    p.depth_scale = ctx->depth_scale;
    p.list_count = ctx->d_list_count;
    p.seg_counts = ctx->d_seg_counts;
    p.allow_fused = ctx->knobs.ingest_unfused ? 0 : 1;
    level_geometry(w, h, levels, p.g, nullptr);
    {
        int rc = stage_frame_slots(ctx, slots, n_frames);
        if (rc) return rc;
    }
    HIP_TRY(ctx, dvo_launch_batch_ingest(p, ctx->stream));
    for (int i = 0; i < n_frames; ++i) ctx->slots[slots[i]].ready = true;
    return DVO_OK;
}

int dvo_upload_frames(dvo_ctx* ctx, int n_frames, const int* slots, const uint8_t* gray, const uint16_t* depth,
                      int w, int h, int levels)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!slots || !gray || !depth) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_upload_frames: null argument");
    if (n_frames < 1 || n_frames > ctx->n_slots) return fail(ctx, DVO_ERR_INVALID_ARG, "n_frames %d outside [1,%d]", n_frames, ctx->n_slots);
    if (w <= 0 || h <= 0 || w > ctx->max_w || h > ctx->max_h) return fail(ctx, DVO_ERR_INVALID_ARG, "image %dx%d outside the ctx maximum", w, h);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one device staging area for the whole batch (grown on demand), two large copies instead of 2 per frame
    const size_t npx = (size_t)w * h, need = align_up(npx * n_frames, 256) + npx * n_frames * 2 + 256;
    if (need > ctx->batch_stage_bytes) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->upload_peer) HIP_TRY(ctx, hipStreamSynchronize(ctx->upload_peer->copy_stream));   // copies into the old area may be in flight
        if (ctx->d_batch_stage) (void)hipFree(ctx->d_batch_stage);
        ctx->d_batch_stage = nullptr;
        ctx->batch_stage_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_batch_stage, need));
        ctx->batch_stage_bytes = need;
    }
    uint8_t* sg = ctx->d_batch_stage;
    uint16_t* sd = reinterpret_cast<uint16_t*>(ctx->d_batch_stage + align_up(npx * n_frames, 256));
    if (ctx->upload_peer == nullptr) {
        HIP_TRY(ctx, hipMemcpyAsync(sg, gray, npx * n_frames, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(sd, depth, npx * n_frames * 2, hipMemcpyHostToDevice, ctx->stream));
        return dvo_upload_frames_device(ctx, n_frames, slots, sg, sd, w, h, levels);
    }
    // Copies on the shared in-order copy stream (uploads of different contexts take turns on the link without any
    // cross-stream wait in front of a copy: the runtime may resolve such a wait on the host, blocking the caller);
    // only kernels wait, by event.  The staging area is reused: the copy waits for the ingest kernels of the previous
    // upload of this ctx, an event that completed long ago in steady state.
    hipStream_t cs = ctx->upload_peer->copy_stream;
    if (ctx->ingested_recorded) HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->ev_ingested, 0));
    HIP_TRY(ctx, hipMemcpyAsync(sg, gray, npx * n_frames, hipMemcpyHostToDevice, cs));
    HIP_TRY(ctx, hipMemcpyAsync(sd, depth, npx * n_frames * 2, hipMemcpyHostToDevice, cs));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_uploaded, cs));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_uploaded, 0));
    int rc = dvo_upload_frames_device(ctx, n_frames, slots, sg, sd, w, h, levels);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_ingested, ctx->stream));
    ctx->ingested_recorded = true;
    return DVO_OK;
}

// largest raw depth the clamp of core/base_dense_visual_odometry.py:59 keeps: (double)d * depth_scale > max_distance is
// monotonic in d, so the per-pixel float64 product becomes one integer compare on the device, exactly
static unsigned depth_clamp_threshold(double depth_scale, double max_distance)
{
    unsigned lo = 0, hi = 65535;                   // invariant: every d <= lo is kept (d = 0 always is)
    if (!((double)hi * depth_scale > max_distance)) return hi;
    while (lo + 1 < hi) {
        const unsigned mid = (lo + hi) / 2;
        if ((double)mid * depth_scale > max_distance) hi = mid; else lo = mid;
    }
    return lo;
}

int dvo_upload_frames_bgr_device(dvo_ctx* ctx, int n_frames, const int* slots, const void* d_bgr, const void* d_depth,
                                 int w, int h, int levels, double max_distance)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!slots || !d_bgr || !d_depth) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_upload_frames_bgr_device: null argument");
    if (n_frames < 1 || n_frames > ctx->n_slots) return fail(ctx, DVO_ERR_INVALID_ARG, "n_frames %d outside [1,%d]", n_frames, ctx->n_slots);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n_frames; ++i) {
        for (int j = 0; j < i; ++j)
            if (slots[j] == slots[i]) return fail(ctx, DVO_ERR_INVALID_ARG, "slot %d listed twice", slots[i]);
        int rc = prepare_slot(ctx, slots[i], w, h, levels);
        if (rc) return rc;
    }
    DvoBatchIngestParams p;
    p.slots = ctx->d_slots;
    p.slot_stride = ctx->slot_stride;
    p.slot_ids = ctx->d_frame_slots;
    p.src_gray = nullptr;                        // level 0 is written into the slots by the colour kernel
    p.src_depth = nullptr;
    p.n_frames = n_frames;
    p.levels = levels;
    p.depth_scale = ctx->depth_scale;
    p.list_count = ctx->d_list_count;
    p.seg_counts = ctx->d_seg_counts;
    p.allow_fused = ctx->knobs.ingest_unfused ? 0 : 1;
    level_geometry(w, h, levels, p.g, nullptr);
    {
        int rc = stage_frame_slots(ctx, slots, n_frames);
        if (rc) return rc;
    }
    HIP_TRY(ctx, dvo_launch_batch_bgr_ingest(p, static_cast<const uint8_t*>(d_bgr), static_cast<const uint16_t*>(d_depth),
                                             depth_clamp_threshold(ctx->depth_scale, max_distance), ctx->stream));
    HIP_TRY(ctx, dvo_launch_batch_ingest(p, ctx->stream));
    for (int i = 0; i < n_frames; ++i) ctx->slots[slots[i]].ready = true;
    return DVO_OK;
}

int dvo_upload_frames_bgr(dvo_ctx* ctx, int n_frames, const int* slots, const uint8_t* bgr, const uint16_t* depth,
                          int w, int h, int levels, double max_distance)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!slots || !bgr || !depth) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_upload_frames_bgr: null argument");
    if (n_frames < 1 || n_frames > ctx->n_slots) return fail(ctx, DVO_ERR_INVALID_ARG, "n_frames %d outside [1,%d]", n_frames, ctx->n_slots);
    if (w <= 0 || h <= 0 || w > ctx->max_w || h > ctx->max_h) return fail(ctx, DVO_ERR_INVALID_ARG, "image %dx%d outside the ctx maximum", w, h);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one device staging area for the whole batch (grown on demand): colour (3 B/px) then depth (2 B/px), two large copies
    const size_t npx = (size_t)w * h, cbytes = align_up(npx * n_frames * 3, 256), need = cbytes + npx * n_frames * 2 + 256;
    if (need > ctx->batch_stage_bytes) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->upload_peer) HIP_TRY(ctx, hipStreamSynchronize(ctx->upload_peer->copy_stream));
        if (ctx->d_batch_stage) (void)hipFree(ctx->d_batch_stage);
        ctx->d_batch_stage = nullptr;
        ctx->batch_stage_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_batch_stage, need));
        ctx->batch_stage_bytes = need;
    }
    uint8_t* sc = ctx->d_batch_stage;
    uint16_t* sd = reinterpret_cast<uint16_t*>(ctx->d_batch_stage + cbytes);
    if (ctx->upload_peer == nullptr) {
        HIP_TRY(ctx, hipMemcpyAsync(sc, bgr, npx * n_frames * 3, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(sd, depth, npx * n_frames * 2, hipMemcpyHostToDevice, ctx->stream));
        return dvo_upload_frames_bgr_device(ctx, n_frames, slots, sc, sd, w, h, levels, max_distance);
    }
    // copies on the shared in-order copy stream, kernels wait by event (see dvo_upload_frames)
    hipStream_t cs = ctx->upload_peer->copy_stream;
    if (ctx->ingested_recorded) HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->ev_ingested, 0));
    HIP_TRY(ctx, hipMemcpyAsync(sc, bgr, npx * n_frames * 3, hipMemcpyHostToDevice, cs));
    HIP_TRY(ctx, hipMemcpyAsync(sd, depth, npx * n_frames * 2, hipMemcpyHostToDevice, cs));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_uploaded, cs));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_uploaded, 0));
    int rc = dvo_upload_frames_bgr_device(ctx, n_frames, slots, sc, sd, w, h, levels, max_distance);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_ingested, ctx->stream));
    ctx->ingested_recorded = true;
    return DVO_OK;
}

int dvo_set_upload_peer(dvo_ctx* ctx, dvo_ctx* peer)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (peer && peer->device != ctx->device) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_set_upload_peer: contexts are on different devices");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->upload_peer) HIP_TRY(ctx, hipStreamSynchronize(ctx->upload_peer->copy_stream));
    ctx->upload_peer = peer;
    return DVO_OK;
}

int dvo_wait_for(dvo_ctx* ctx, dvo_ctx* other)
{
    if (!ctx || !other) return DVO_ERR_INVALID_ARG;
    if (ctx == other) return DVO_OK;
    if (ctx->device != other->device) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_wait_for: contexts are on different devices");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(other->ev_order, other->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, other->ev_order, 0));
    return DVO_OK;
}

int dvo_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return DVO_ERR_INVALID_ARG;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(nullptr, (e == hipErrorOutOfMemory) ? DVO_ERR_OUT_OF_MEMORY : DVO_ERR_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return DVO_OK;
}

int dvo_host_free(void* p)
{
    if (!p) return DVO_OK;
    return hipHostFree(p) == hipSuccess ? DVO_OK : DVO_ERR_HIP;
}

int dvo_upload_frame_bgr(dvo_ctx* ctx, int slot, const uint8_t* bgr, const uint16_t* depth, int w, int h, int levels,
                         double max_distance)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!bgr || !depth) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_upload_frame_bgr: null image");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = prepare_slot(ctx, slot, w, h, levels);
    if (rc) return rc;
    DvoLevelGeom g[DVO_MAX_LEVELS];
    level_geometry(w, h, levels, g, nullptr);
    uint8_t* base = ctx->d_slots + (size_t)slot * ctx->slot_stride;
    const size_t n = (size_t)w * h;
    uint8_t* st_bgr = ctx->d_stage_img;
    uint16_t* st_depth = reinterpret_cast<uint16_t*>(ctx->d_stage_img + align_up(n * 3, 256));
    HIP_TRY(ctx, hipMemcpyAsync(st_bgr, bgr, n * 3, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(st_depth, depth, n * 2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, dvo_launch_bgr_ingest(st_bgr, st_depth, w, h, base + g[0].off_gray,
                                       reinterpret_cast<uint16_t*>(base + g[0].off_depth), ctx->depth_scale, max_distance, ctx->stream));
    rc = build_pyramid_on_device(ctx, slot, w, h, levels);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->slots[slot].ready = true;
    return DVO_OK;
}

int dvo_slot_info(const dvo_ctx* ctx, int slot, int* w, int* h, int* levels)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (slot < 0 || slot >= ctx->n_slots) return fail(const_cast<dvo_ctx*>(ctx), DVO_ERR_INVALID_ARG, "slot %d out of range", slot);
    const SlotInfo& s = ctx->slots[slot];
    if (w) *w = s.ready ? s.w : 0;
    if (h) *h = s.ready ? s.h : 0;
    if (levels) *levels = s.ready ? s.levels : 0;
    return DVO_OK;
}

int dvo_download_level(dvo_ctx* ctx, int slot, int level, uint8_t* gray, uint16_t* depth, int16_t* gx, int16_t* gy)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_slot(ctx, slot, "dvo_download_level");
    if (rc) return rc;
    const SlotInfo& s = ctx->slots[slot];
    if (level < 0 || level >= s.levels) return fail(ctx, DVO_ERR_INVALID_ARG, "level %d outside [0,%d)", level, s.levels);
    DvoLevelGeom g[DVO_MAX_LEVELS];
    level_geometry(s.w, s.h, s.levels, g, nullptr);
    const uint8_t* base = ctx->d_slots + (size_t)slot * ctx->slot_stride;
    const size_t n = (size_t)g[level].npix;
    if (gray) HIP_TRY(ctx, hipMemcpyAsync(gray, base + g[level].off_gray, n, hipMemcpyDeviceToHost, ctx->stream));
    if (depth) HIP_TRY(ctx, hipMemcpyAsync(depth, base + g[level].off_depth, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (gx || gy) {
        int16_t* dgx = reinterpret_cast<int16_t*>(ctx->d_dense);
        int16_t* dgy = dgx + n;
        HIP_TRY(ctx, dvo_launch_unpack(reinterpret_cast<const uint32_t*>(base + g[level].off_packed), g[level].w, g[level].h,
                                       g[level].pitch, dgx, dgy, ctx->stream));
        if (gx) HIP_TRY(ctx, hipMemcpyAsync(gx, dgx, n * 2, hipMemcpyDeviceToHost, ctx->stream));
        if (gy) HIP_TRY(ctx, hipMemcpyAsync(gy, dgy, n * 2, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DVO_OK;
}

int dvo_download_points(dvo_ctx* ctx, int slot, int level, float* z, uint32_t* pix, int capacity, int* count)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!count || capacity < 0) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_download_points: null count / negative capacity");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_slot(ctx, slot, "dvo_download_points");
    if (rc) return rc;
    const SlotInfo& s = ctx->slots[slot];
    if (!s.lists) return fail(ctx, DVO_ERR_BAD_STATE, "slot %d was uploaded before dvo_set_camera (or under another depth_scale): upload it again", slot);
    if (level < 0 || level >= s.levels) return fail(ctx, DVO_ERR_INVALID_ARG, "level %d outside [0,%d)", level, s.levels);
    DvoLevelGeom g[DVO_MAX_LEVELS];
    level_geometry(s.w, s.h, s.levels, g, nullptr);
    int n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, ctx->d_list_count + (size_t)slot * DVO_LIST_LEVELS + level, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *count = n;
    const int m = n < capacity ? n : capacity;
    if (m > 0 && (z || pix)) {
        std::vector<DvoPointRec> host((size_t)m);
        const uint8_t* base = ctx->d_slots + (size_t)slot * ctx->slot_stride;
        HIP_TRY(ctx, hipMemcpy(host.data(), base + g[level].off_list, (size_t)m * sizeof(DvoPointRec), hipMemcpyDeviceToHost));
        for (int i = 0; i < m; ++i) {
            if (z) z[i] = host[i].z;
            if (pix) pix[i] = host[i].pix;
        }
    }
    return DVO_OK;
}

static int stage_pairs(dvo_ctx* ctx, int n_pairs, const int* prev_slot, const int* cur_slot, const float* T,
                       const double* wparam)
{
    // the pinned staging buffers may still be the source of an in-flight async copy of an earlier enqueue
    if (ctx->staging_in_flight) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_staging));
        ctx->staging_in_flight = false;
    }
    memcpy(ctx->hs.prev, prev_slot, n_pairs * sizeof(int));
    memcpy(ctx->hs.cur, cur_slot, n_pairs * sizeof(int));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_prev, ctx->hs.prev, n_pairs * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_cur, ctx->hs.cur, n_pairs * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (T) {
        memcpy(ctx->hs.T, T, (size_t)n_pairs * 12 * sizeof(float));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_T, ctx->hs.T, (size_t)n_pairs * 12 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    if (wparam) {
        memcpy(ctx->hs.wparam, wparam, n_pairs * sizeof(double));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_wparam, ctx->hs.wparam, n_pairs * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_staging, ctx->stream));
    ctx->staging_in_flight = true;
    return DVO_OK;
}

static int check_variant(dvo_ctx* ctx, int variant)
{
    if (variant != DVO_VARIANT_V1 && variant != DVO_VARIANT_V2 && variant != DVO_VARIANT_V3 && variant != DVO_VARIANT_GN)
        return fail(ctx, DVO_ERR_INVALID_ARG, "variant %d unknown (V1=0, V2=1, V3=2, GN=3)", variant);
    return DVO_OK;
}

int dvo_normal_equations(dvo_ctx* ctx, int level, int n_pairs, const int* prev_slot, const int* cur_slot,
                         const float* T, int variant, int weight_mode, const double* weight_param, double* out)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!T || !out) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_normal_equations: null T/out");
    if (level < 0) return fail(ctx, DVO_ERR_INVALID_ARG, "level %d < 0", level);
    if (weight_mode < 0 || weight_mode > 3) return fail(ctx, DVO_ERR_INVALID_ARG, "weight_mode %d unknown", weight_mode);
    if (weight_mode != 0 && !weight_param) return fail(ctx, DVO_ERR_INVALID_ARG, "weight_param required for weight_mode %d", weight_mode);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_variant(ctx, variant);
    if (rc) return rc;
    int w, h;
    rc = check_pairs(ctx, n_pairs, prev_slot, cur_slot, level, &w, &h);
    if (rc) return rc;
    rc = stage_pairs(ctx, n_pairs, prev_slot, cur_slot, T, weight_mode ? weight_param : nullptr);
    if (rc) return rc;
    DvoReduceParams p;
    rc = fill_reduce_params(ctx, level, n_pairs, w, h, &p);
    if (rc) return rc;
    int exp_mode = 0;
#ifdef DVO_EXPERIMENTS
    if (ctx->knobs.exp_taps && variant == DVO_VARIANT_V1 && weight_mode == 0 && p.cam.skew0) {
        // experimental tap formulations: side planes of the current frames, rebuilt on every call (test hook, not timed)
        exp_mode = ctx->knobs.exp_taps;
        if (!ctx->d_exp_gradh) {
            DvoLevelGeom g0[DVO_MAX_LEVELS];
            level_geometry(ctx->max_w, ctx->max_h, 1, g0, nullptr);
            ctx->exp_plane_texels = align_up((size_t)(g0[0].h + 1) * g0[0].pitch + 64, 64);
            HIP_TRY(ctx, hipMalloc(&ctx->d_exp_gradh, (size_t)ctx->n_slots * ctx->exp_plane_texels * 4 + 8192));
            HIP_TRY(ctx, hipMalloc(&ctx->d_exp_ipl, (size_t)ctx->n_slots * ctx->exp_plane_texels + 8192));
        }
        HIP_TRY(ctx, dvo_launch_exp_convert(ctx->d_slots, ctx->slot_stride, p.g, ctx->d_cur, n_pairs, ctx->d_exp_gradh, ctx->d_exp_ipl,
                                            ctx->exp_plane_texels, ctx->stream));
    }
#endif
    HIP_TRY(ctx, launch_reduce_profiled(ctx, p, level, variant, weight_mode, 0, exp_mode));
    HIP_TRY(ctx, dvo_launch_sum_partials(ctx->d_partials, p.blocks_per_pair, n_pairs, ctx->d_out, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs.out, ctx->d_out, (size_t)n_pairs * DVO_STRIDE * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->hs.out, (size_t)n_pairs * DVO_STRIDE * sizeof(double));
    return DVO_OK;
}

int dvo_weighted_r2_sum(dvo_ctx* ctx, int level, int n_pairs, const int* prev_slot, const int* cur_slot,
                        const float* T, int variant, const double* lambda, double* out)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!T || !out || !lambda) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_weighted_r2_sum: null argument");
    if (level < 0) return fail(ctx, DVO_ERR_INVALID_ARG, "level %d < 0", level);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_variant(ctx, variant);
    if (rc) return rc;
    int w, h;
    rc = check_pairs(ctx, n_pairs, prev_slot, cur_slot, level, &w, &h);
    if (rc) return rc;
    rc = stage_pairs(ctx, n_pairs, prev_slot, cur_slot, T, lambda);
    if (rc) return rc;
    DvoReduceParams p;
    rc = fill_reduce_params(ctx, level, n_pairs, w, h, &p);
    if (rc) return rc;
    HIP_TRY(ctx, launch_reduce_profiled(ctx, p, level, variant, 1, 1));
    HIP_TRY(ctx, dvo_launch_sum_partials(ctx->d_partials, p.blocks_per_pair, n_pairs, ctx->d_out, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs.out, ctx->d_out, (size_t)n_pairs * DVO_STRIDE * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_pairs; ++i) out[i] = ctx->hs.out[(size_t)i * DVO_STRIDE];
    return DVO_OK;
}

int dvo_weighted_r2_stats(dvo_ctx* ctx, int level, int n_pairs, const int* prev_slot, const int* cur_slot,
                          const float* T, int variant, const double* lambda, double* stats)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!T || !stats || !lambda) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_weighted_r2_stats: null argument");
    if (level < 0) return fail(ctx, DVO_ERR_INVALID_ARG, "level %d < 0", level);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_variant(ctx, variant);
    if (rc) return rc;
    int w, h;
    rc = check_pairs(ctx, n_pairs, prev_slot, cur_slot, level, &w, &h);
    if (rc) return rc;
    rc = stage_pairs(ctx, n_pairs, prev_slot, cur_slot, T, lambda);
    if (rc) return rc;
    DvoReduceParams p;
    rc = fill_reduce_params(ctx, level, n_pairs, w, h, &p);
    if (rc) return rc;
    HIP_TRY(ctx, launch_reduce_profiled(ctx, p, level, variant, 1, 1));
    HIP_TRY(ctx, dvo_launch_sum_partials(ctx->d_partials, p.blocks_per_pair, n_pairs, ctx->d_out, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs.out, ctx->d_out, (size_t)n_pairs * DVO_STRIDE * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_pairs; ++i) {
        stats[2 * i] = ctx->hs.out[(size_t)i * DVO_STRIDE];
        stats[2 * i + 1] = ctx->hs.out[(size_t)i * DVO_STRIDE + 1];
    }
    return DVO_OK;
}

int dvo_residuals_jacobian(dvo_ctx* ctx, int level, int prev_slot, int cur_slot, const float* T, int variant,
                           float* r, float* J, uint8_t* valid, float* uv)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!T) return fail(ctx, DVO_ERR_INVALID_ARG, "dvo_residuals_jacobian: null T");
    if (level < 0) return fail(ctx, DVO_ERR_INVALID_ARG, "level %d < 0", level);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_variant(ctx, variant);
    if (rc) return rc;
    int w, h;
    rc = check_pairs(ctx, 1, &prev_slot, &cur_slot, level, &w, &h);
    if (rc) return rc;
    DvoLevelGeom g[DVO_MAX_LEVELS];
    level_geometry(w, h, level + 1, g, nullptr);
    const size_t n = (size_t)g[level].npix;
    DvoDenseParams p;
    p.slots = ctx->d_slots;
    p.slot_stride = ctx->slot_stride;
    p.g = g[level];
    p.cam = level_camera(ctx, level, g[level]);
    p.depth_scale = ctx->depth_scale;
    p.prev_slot = prev_slot;
    p.cur_slot = cur_slot;
    memcpy(p.T, T, sizeof(p.T));
    p.r = ctx->d_dense;
    p.J = ctx->d_dense + n;
    p.uv = ctx->d_dense + 7 * n;
    p.valid = reinterpret_cast<uint8_t*>(ctx->d_dense + 9 * n);
    HIP_TRY(ctx, dvo_launch_dense(p, variant, ctx->stream));
    if (r) HIP_TRY(ctx, hipMemcpyAsync(r, p.r, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (J) HIP_TRY(ctx, hipMemcpyAsync(J, p.J, 6 * n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (uv) HIP_TRY(ctx, hipMemcpyAsync(uv, p.uv, 2 * n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (valid) HIP_TRY(ctx, hipMemcpyAsync(valid, p.valid, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DVO_OK;
}

// Enqueues (or, under stream capture, records) the launch sequence of one batched coarse-to-fine solve.
static int record_solve(dvo_ctx* ctx, const SolveKey& k, const float* d_init)
{
    const int n_pairs = k.n_pairs, w = k.w, h = k.h, levels = k.levels;
    const int variant = k.variant, weight_mode = k.weight_mode, max_increased_steps_allowed = k.max_increased;
    const double tolerance = k.tolerance, weight_param = k.weight_param;
    int rc;
    const double lambda0 = 1.0 / 25.0;                      // TDistributionWeighter: initial_sigma = 5 (t_weighter.py:11-15)
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_irls_done, 0, (size_t)n_pairs * sizeof(int), ctx->stream));
    if (ctx->d_inexact) HIP_TRY(ctx, hipMemsetAsync(ctx->d_inexact, 0, 8 * sizeof(int), ctx->stream));
    if (weight_mode == DVO_WEIGHT_STUDENT_T) HIP_TRY(ctx, hipMemsetAsync(ctx->d_pair_flags, 0, (size_t)n_pairs * sizeof(int), ctx->stream));
    const int* trace_row = k.trace_rows ? ctx->d_trace_row : nullptr;
    int* pair_flags = (weight_mode == DVO_WEIGHT_STUDENT_T) ? ctx->d_pair_flags : nullptr;
    HIP_TRY(ctx, dvo_launch_init_state(ctx->d_state, ctx->d_T, d_init, n_pairs, lambda0,
                                       (k.inv_sigma > 0.0 && k.has_last) ? ctx->d_prior : nullptr, ctx->stream));
    // The coarsest levels (at most knobs.small_max_pix pixels each) run in ONE persistent launch, a workgroup per pair
    // doing every iteration of those levels (k_gn_small); the finer levels take one reduction launch + one control launch
    // per iteration.
    int level = levels - 1;
    if (variant != DVO_VARIANT_V3) {
        DvoLevelGeom g[DVO_MAX_LEVELS];
        level_geometry(w, h, levels, g, nullptr);
        DvoSmallParams sp;
        memset(&sp, 0, sizeof(sp));
        int n = 0;
        // a handful of pairs cannot fill the chip either way: there a 160x120 level (13 k records) is faster spread over its ~19
        // workgroups than inside one, so only levels up to 80x60 go to the persistent kernel (single pair: 0.87 -> 0.77 ms)
        // (not with a Student-t weighting: the persistent kernel runs the reference's IRLS loop as written, the per-iteration
        // launches a fixed number of passes, and a pair must not get a different lambda because of the size of its batch)
        const bool student_t = (weight_mode == DVO_WEIGHT_STUDENT_T || weight_mode == DVO_WEIGHT_STUDENT_T_MEAN);
        // (st_exact == 2, the Student-t fallback: EVERY level here, because this kernel runs the reference's IRLS loop as written)
        const int max_pix = (k.st_exact == 2) ? INT_MAX
                          : (n_pairs <= 16 && !student_t) ? (ctx->knobs.small_max_pix < 4800 ? ctx->knobs.small_max_pix : 4800) : ctx->knobs.small_max_pix;
        while (level - n >= 0 && n < DVO_SMALL_LEVELS && g[level - n].npix <= max_pix) {
            sp.level_id[n] = level - n;
            sp.level_iters[n] = k.level_iters[level - n];
            sp.g[n] = g[level - n];
            sp.cam[n] = level_camera(ctx, level - n, g[level - n]);
            ++n;
        }
        for (int i = 1; i < n; ++i)
            if (!sp.cam[i].skew0) sp.cam[0].skew0 = 0;          // one instantiation serves all the launch's levels (dvo_launch_gn_small reads cam[0])
        if (n > 0) {
            sp.slots = ctx->d_slots;
            sp.slot_stride = ctx->slot_stride;
            sp.prev_slot = ctx->d_prev;
            sp.cur_slot = ctx->d_cur;
            sp.list_count = ctx->d_list_count;
            sp.n_levels = n;
            sp.depth_scale = ctx->depth_scale;
            sp.n_pairs = n_pairs;
            sp.state = ctx->d_state;
            sp.T = ctx->d_T;
            sp.active = ctx->d_active;
            sp.wparam = ctx->d_wparam;
            sp.rcache = (weight_mode == DVO_WEIGHT_STUDENT_T || weight_mode == DVO_WEIGHT_STUDENT_T_MEAN) ? ctx->d_rcache : nullptr;
            sp.student_t_mean = (weight_mode == DVO_WEIGHT_STUDENT_T_MEAN) ? 1 : 0;
            sp.trace_row = trace_row;
            sp.rcache_stride = (size_t)ctx->max_w * ctx->max_h;
            sp.tolerance = tolerance;
            sp.max_increased = max_increased_steps_allowed;
            sp.weight_mode = weight_mode;
            sp.weight_param = weight_param;
            sp.inv_sigma = k.has_last ? k.inv_sigma : 0.0;
            sp.lambda0 = lambda0;
            sp.trace = ctx->d_trace;
            sp.trace_stride = DVO_STRIDE + 18;
            sp.trace_cap = ctx->trace_cap;
            sp.stamps = ctx->d_small_stamps;
            // workgroup size: the whole batch should be resident at once (<= 2048 threads per CU at <= 128 VGPRs); a single
            // pair gets the widest workgroup (latency), a large batch narrower ones (more pairs in flight per CU)
            int threads = ctx->knobs.small_threads;
            if (threads == 0) threads = (n_pairs <= 8) ? 1024 : (n_pairs <= 64) ? 512 : 256;   // measured: B=256 per stream: 256 > 512 > 1024
            hipEvent_t ea = nullptr, eb = nullptr;
            if (ctx->profiling) {
                ea = pool_event(ctx);
                eb = pool_event(ctx);
                if (ea && eb) (void)hipEventRecord(ea, ctx->stream);
            }
            HIP_TRY(ctx, dvo_launch_gn_small(sp, variant, threads, ctx->stream));
            if (ctx->profiling && ea && eb) {
                (void)hipEventRecord(eb, ctx->stream);
                ctx->pending.push_back({ea, eb, level});        // booked on the coarsest level it covers
            }
            level -= n;
        }
    }
    for (; level >= 0; --level) {
        DvoReduceParams p;
        rc = fill_reduce_params(ctx, level, n_pairs, w, h, &p);
        if (rc) return rc;
        p.active = ctx->d_active;
        DvoSolveParams sp;
        sp.partials = ctx->d_partials;
        sp.blocks_per_pair = p.blocks_per_pair;
        sp.n_pairs = n_pairs;
        sp.state = ctx->d_state;
        sp.T = ctx->d_T;
        sp.active = ctx->d_active;
        sp.wparam = ctx->d_wparam;
        sp.tolerance = tolerance;
        sp.max_increased = max_increased_steps_allowed;
        sp.weight_mode = weight_mode;
        sp.weight_param = weight_param;
        sp.inv_sigma = k.has_last ? k.inv_sigma : 0.0;   // no previous transform: the reference skips the prior (base_robust_dvo.py:190)
        sp.trace = ctx->d_trace;
        sp.trace_stride = DVO_STRIDE + 18;
        sp.trace_cap = ctx->trace_cap;
        sp.trace_row = trace_row;
        sp.st_tol = ctx->knobs.st_tol;
        sp.pair_flags = pair_flags;
        sp.scale1 = nullptr;
        sp.scale2 = nullptr;
        sp.irls_done = nullptr;
        sp.phase = 0;
        sp.partials2 = nullptr;
        sp.inexact = nullptr;
        sp.lambda0 = lambda0;
        sp.predicted = 0;
        // The reference's Student-t at the per-iteration levels: ONE pass per iteration (moment form, see student_t_moment_combine)
        // instead of scale pass + cached scale pass + weighted pass, wherever the level is large enough for its bound to hold on
        // ordinary images (>= 320x240); a solve whose bound fails is repeated with the three-pass form by dvo_solve_batch_fetch.
        const bool moment_form = (weight_mode == DVO_WEIGHT_STUDENT_T) && !k.st_exact && !ctx->knobs.st_solve_phases &&
                                 variant == DVO_VARIANT_V1 && ctx->d_partials2 != nullptr && p.g.npix >= 76800;
        HIP_TRY(ctx, dvo_launch_level_begin(ctx->d_state, ctx->d_active, ctx->d_wparam, n_pairs, weight_mode, lambda0, moment_form ? 1 : 0, ctx->stream));
        for (int it = 0; it < k.level_iters[level]; ++it) {
            if (moment_form) {
                // Both one-pass forms evaluate the weights under a guessed lambda and correct the sums to first order in what the
                // guess was off by (student_t_moment_combine / student_t_predicted_combine).  The CORRECTED form (second record:
                // derivatives of H and b) is second-order accurate whatever the guess: first three iterations of a level (no lambda
                // history to extrapolate from yet) and every iteration of the levels below 640x480, where the lambda of a few ten
                // thousand points is not predictable to the 1e-4 the cheaper form needs.  The PREDICTED form (only b corrected,
                // six extra sums instead of 30) from the fourth iteration on at the large levels.
                p.partials2 = ctx->d_partials2;
                p.lambda0 = (float)lambda0;
                sp.predicted = (it > 2 && p.g.npix >= 640 * 480) ? 1 : 0;
                HIP_TRY(ctx, launch_reduce_profiled(ctx, p, level, variant, sp.predicted ? 5 : 4, 0));
                sp.partials2 = ctx->d_partials2;
                sp.inexact = ctx->d_inexact;
                HIP_TRY(ctx, dvo_launch_solve(sp, ctx->stream));
                continue;
            }
            if (weight_mode == DVO_WEIGHT_STUDENT_T || weight_mode == DVO_WEIGHT_STUDENT_T_MEAN) {
                // IRLS scale passes (weighter/t_weighter.py:24-32): the first keeps the residuals of this pose, the later ones sum
                // them again under the new lambda
                p.rcache = ctx->d_rcache;
                if (weight_mode == DVO_WEIGHT_STUDENT_T && ctx->d_rcache != nullptr && ctx->d_scale1 != nullptr && !ctx->knobs.st_solve_phases) {
                    // the reference's form, two passes, no control launch between the three kernels: each derives its lambda from the
                    // per-workgroup sums of the passes before it (same summation order as k_solve => same lambda, bit for bit)
                    p.scale_out = ctx->d_scale1;
                    HIP_TRY(ctx, launch_reduce_profiled(ctx, p, level, variant, 1, 1));
                    p.scale_out = ctx->d_scale2;
                    p.scale1 = ctx->d_scale1;
                    HIP_TRY(ctx, dvo_launch_scale_cached(p, ctx->stream));
                    p.scale_out = nullptr;
                    p.scale2 = ctx->d_scale2;
                    sp.scale1 = ctx->d_scale1;          // the control launch checks the two-pass shape of the reference's loop
                    sp.scale2 = ctx->d_scale2;
                } else {
                    // a control launch after every pass (k_solve, phase > 0) updates lambda and the per-pair converged flag; the
                    // workgroups of a converged pair leave the remaining passes at once
                    const int npass = (weight_mode == DVO_WEIGHT_STUDENT_T_MEAN) ? 4 : 2;
                    p.irls_done = ctx->d_irls_done;
                    sp.irls_done = ctx->d_irls_done;
                    for (int pass = 1; pass <= npass; ++pass) {
                        if (pass == 1 || ctx->d_rcache == nullptr) HIP_TRY(ctx, launch_reduce_profiled(ctx, p, level, variant, 1, 1));
                        else HIP_TRY(ctx, dvo_launch_scale_cached(p, ctx->stream));
                        sp.phase = pass;
                        HIP_TRY(ctx, dvo_launch_solve(sp, ctx->stream));
                    }
                    p.irls_done = nullptr;
                }
                p.rcache = nullptr;
            }
            HIP_TRY(ctx, launch_reduce_profiled(ctx, p, level, variant, weight_mode, 0));
            p.scale1 = nullptr;
            p.scale2 = nullptr;
            sp.phase = 0;
            HIP_TRY(ctx, dvo_launch_solve(sp, ctx->stream));
            sp.scale1 = nullptr;
            sp.scale2 = nullptr;
            sp.irls_done = nullptr;
        }
    }
    HIP_TRY(ctx, dvo_launch_finish(ctx->d_state, n_pairs, ctx->d_xi, ctx->d_iters, ctx->d_err, ctx->stream));
    return DVO_OK;
}


static int enqueue_solve(dvo_ctx* ctx, const SolveKey& key, const float* d_init);

int dvo_solve_batch_enqueue(dvo_ctx* ctx, int n_pairs, const int* prev_slot, const int* cur_slot, int levels,
                            int max_iterations, double tolerance, int max_increased_steps_allowed, int variant,
                            int weight_mode, double weight_param, const float* T_init)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (levels < 1 || max_iterations < 0) return fail(ctx, DVO_ERR_INVALID_ARG, "levels %d / max_iterations %d invalid", levels, max_iterations);
    if (weight_mode < 0 || weight_mode > 3) return fail(ctx, DVO_ERR_INVALID_ARG, "weight_mode %d unknown", weight_mode);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = check_variant(ctx, variant);
    if (rc) return rc;
    int w, h;
    rc = check_pairs(ctx, n_pairs, prev_slot, cur_slot, levels - 1, &w, &h);
    if (rc) return rc;
    rc = stage_pairs(ctx, n_pairs, prev_slot, cur_slot, nullptr, nullptr);
    if (rc) return rc;
    if ((weight_mode == DVO_WEIGHT_STUDENT_T || weight_mode == DVO_WEIGHT_STUDENT_T_MEAN) && ctx->d_rcache == nullptr && !ctx->knobs.no_rcache) {
        // residual cache of the scale passes (allocated on first use, outside any graph capture); without it the second
        // pass recomputes the residuals
        if (hipMalloc((void**)&ctx->d_rcache, (size_t)ctx->max_pairs * ctx->max_w * ctx->max_h * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            ctx->d_rcache = nullptr;
        }
        const size_t nsc = ctx->partials_doubles / DVO_STRIDE;           // one double per (pair, workgroup)
        if (ctx->d_rcache != nullptr && (hipMalloc((void**)&ctx->d_scale1, nsc * sizeof(double)) != hipSuccess ||
                                         hipMalloc((void**)&ctx->d_scale2, nsc * sizeof(double)) != hipSuccess)) {
            (void)hipGetLastError();
            if (ctx->d_scale1) (void)hipFree(ctx->d_scale1);
            ctx->d_scale1 = ctx->d_scale2 = nullptr;
        }
    }
    if (weight_mode == DVO_WEIGHT_STUDENT_T && ctx->d_partials2 == nullptr && !ctx->knobs.st_solve_phases) {
        // moment form of the reference's Student-t: a second set of partial sums and the counter of unsafe iterations
        if (hipMalloc((void**)&ctx->d_partials2, ctx->partials_doubles * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&ctx->d_inexact, 8 * sizeof(int)) != hipSuccess ||
            hipHostMalloc((void**)&ctx->h_inexact, 8 * sizeof(int)) != hipSuccess) {
            (void)hipGetLastError();
            if (ctx->d_partials2) (void)hipFree(ctx->d_partials2);
            if (ctx->d_inexact) (void)hipFree(ctx->d_inexact);
            ctx->d_partials2 = nullptr;
            ctx->d_inexact = nullptr;                     // the three-pass form needs neither
        }
    }
    const float* d_init = nullptr;
    if (T_init) {
        memcpy(ctx->hs.T, T_init, (size_t)n_pairs * 12 * sizeof(float));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_Tinit, ctx->hs.T, (size_t)n_pairs * 12 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_staging, ctx->stream));
        d_init = ctx->d_Tinit;
    }
    // The launch sequence of the whole solve is fixed by (batch size, image size, levels, iterations, modes): it is
    // captured once into a hipGraph and replayed, which removes the per-launch host cost of the ~2*levels*iterations
    // dependent launches (the loop is launch-bound for small batches).  Slot ids / initial poses are read by the
    // kernels from fixed device buffers refreshed above, so they are not part of the key.
    SolveKey key;
    memset(&key, 0, sizeof(key));
    key.n_pairs = n_pairs; key.w = w; key.h = h; key.levels = levels; key.max_iterations = max_iterations;
    key.max_increased = max_increased_steps_allowed; key.variant = variant; key.weight_mode = weight_mode;
    key.tolerance = tolerance; key.weight_param = weight_param; key.has_init = T_init ? 1 : 0;
    key.trace = ctx->d_trace; key.trace_cap = ctx->trace_cap; key.stream = ctx->stream;
    key.inv_sigma = (ctx->prior_sigma > 0.0) ? 1.0 / ctx->prior_sigma : 0.0;
    key.has_last = ctx->prior_has_last ? 1 : 0;
    for (int l = 0; l < levels && l < DVO_MAX_LEVELS; ++l) key.level_iters[l] = ctx->level_iters_set ? ctx->level_iters[l] : max_iterations;
    key.st_exact = ctx->knobs.st_exact ? 1 : 0;
    key.trace_rows = 0;
    ctx->last_key = key;
    ctx->last_has_init = (T_init != nullptr);
    ctx->refetch_invalid = false;
    ctx->last_moment = (weight_mode == DVO_WEIGHT_STUDENT_T) && variant != DVO_VARIANT_V3;
    if (ctx->last_moment) {
        // host copies for the per-pair fallback of dvo_solve_batch_fetch
        dvo_ctx::LastSolve& L = ctx->last;
        L.prev.assign(prev_slot, prev_slot + n_pairs);
        L.cur.assign(cur_slot, cur_slot + n_pairs);
        L.ep_prev.resize(n_pairs);
        L.ep_cur.resize(n_pairs);
        for (int i = 0; i < n_pairs; ++i) { L.ep_prev[i] = ctx->slots[prev_slot[i]].epoch; L.ep_cur[i] = ctx->slots[cur_slot[i]].epoch; }
        if (T_init) L.Tinit.assign(T_init, T_init + (size_t)n_pairs * 12); else L.Tinit.clear();
        if (key.inv_sigma > 0.0 && key.has_last) L.prior.assign(ctx->h_prior, ctx->h_prior + (size_t)n_pairs * 7); else L.prior.clear();
    }
    return enqueue_solve(ctx, key, d_init);
}

// the launch sequence of `key`: a cached hipGraph replay, a capture + replay, or plain launches (profiling / DVO_GRAPH=0)
static int enqueue_solve(dvo_ctx* ctx, const SolveKey& key, const float* d_init)
{
    int rc;
    const bool use_graph = ctx->graphs_enabled && !ctx->profiling;
    if (use_graph) {
        for (auto& g : ctx->graphs)
            if (memcmp(&g.key, &key, sizeof(key)) == 0) {
                HIP_TRY(ctx, hipGraphLaunch(g.exec, ctx->stream));
                return DVO_OK;
            }
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        rc = record_solve(ctx, key, d_init);
        hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIP_TRY(ctx, e);
        HIP_TRY(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        if (ctx->graphs.size() >= 16) { (void)hipGraphExecDestroy(ctx->graphs.front().exec); ctx->graphs.erase(ctx->graphs.begin()); }
        ctx->graphs.push_back({key, exec});
        HIP_TRY(ctx, hipGraphLaunch(exec, ctx->stream));
        return DVO_OK;
    }
    return record_solve(ctx, key, d_init);
}

// copies the device-side results of the solve enqueued last into the page-locked staging buffers and waits for them
static int fetch_results(dvo_ctx* ctx, int n_pairs, bool student_t)
{
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs.T, ctx->d_T, (size_t)n_pairs * 12 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs.xi, ctx->d_xi, (size_t)n_pairs * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs.iters, ctx->d_iters, (size_t)n_pairs * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs.err, ctx->d_err, (size_t)n_pairs * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (student_t) {
        if (ctx->d_inexact) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_inexact, ctx->d_inexact, 8 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pair_flags, ctx->d_pair_flags, (size_t)n_pairs * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DVO_OK;
}

// Student-t of the reference: the pairs `idx` of the solve enqueued last, solved again as a compacted batch in the form `st_exact`
// (1: passes as written at the per-iteration levels; 2: every level in the persistent kernel = the reference's loop itself), results
// scattered into the caller's arrays.  On return h_pair_flags[0..k) holds the flags of the re-solve (compacted order).
static int resolve_pairs(dvo_ctx* ctx, const std::vector<int>& idx, int st_exact, float* T_out, double* xi_out, int* iters_out, double* err_out)
{
    const dvo_ctx::LastSolve& L = ctx->last;
    const int k = (int)idx.size();
    std::vector<int> prev(k), cur(k);
    for (int i = 0; i < k; ++i) {
        const int j = idx[i];
        prev[i] = L.prev[j];
        cur[i] = L.cur[j];
        // the frames must still be the ones the solve was enqueued on (include/dvo_hip.h, dvo_solve_batch_enqueue)
        if (ctx->slots[prev[i]].epoch != L.ep_prev[j] || ctx->slots[cur[i]].epoch != L.ep_cur[j] || !ctx->slots[prev[i]].ready || !ctx->slots[cur[i]].ready)
            return fail(ctx, DVO_ERR_BAD_STATE, "Student-t solve: pair %d must be solved again with the passes as written, but slot %d or %d was uploaded "
                        "again between dvo_solve_batch_enqueue and dvo_solve_batch_fetch", j, prev[i], cur[i]);
    }
    int rc = stage_pairs(ctx, k, prev.data(), cur.data(), nullptr, nullptr);
    if (rc) return rc;
    const float* d_init = nullptr;
    if (!L.Tinit.empty()) {
        for (int i = 0; i < k; ++i) memcpy(ctx->hs.T + (size_t)i * 12, L.Tinit.data() + (size_t)idx[i] * 12, 12 * sizeof(float));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_Tinit, ctx->hs.T, (size_t)k * 12 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        d_init = ctx->d_Tinit;
    }
    // (page-locked scratch for the small compacted arrays: hs.wparam holds max_pairs doubles, unused by the solve path)
    if (!L.prior.empty()) {
        std::vector<double> pr((size_t)k * 7);
        for (int i = 0; i < k; ++i) memcpy(pr.data() + (size_t)i * 7, L.prior.data() + (size_t)idx[i] * 7, 7 * sizeof(double));
        HIP_TRY(ctx, hipMemcpy(ctx->d_prior, pr.data(), pr.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    SolveKey key = ctx->last_key;
    key.n_pairs = k;
    key.st_exact = st_exact;
    if (ctx->d_trace) {
        key.trace_rows = 1;
        HIP_TRY(ctx, hipMemcpy(ctx->d_trace_row, idx.data(), (size_t)k * sizeof(int), hipMemcpyHostToDevice));
        // the re-solved pairs' rows start again (iterations beyond the new count would otherwise keep the first solve's records)
        const size_t row = (size_t)ctx->trace_cap * (DVO_STRIDE + 18) * sizeof(double);
        for (int i = 0; i < k; ++i) HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char*>(ctx->d_trace) + (size_t)idx[i] * row, 0, row, ctx->stream));
    }
    rc = enqueue_solve(ctx, key, d_init);
    if (rc) return rc;
    rc = fetch_results(ctx, k, true);
    if (rc) return rc;
    for (int i = 0; i < k; ++i) {
        const int j = idx[i];
        if (T_out) memcpy(T_out + (size_t)j * 12, ctx->hs.T + (size_t)i * 12, 12 * sizeof(float));
        if (xi_out) memcpy(xi_out + (size_t)j * 6, ctx->hs.xi + (size_t)i * 6, 6 * sizeof(double));
        if (iters_out) iters_out[j] = ctx->hs.iters[i];
        if (err_out) err_out[j] = ctx->hs.err[i];
    }
    if (!L.prior.empty())           // leave the whole batch's prior where dvo_set_prior put it
        HIP_TRY(ctx, hipMemcpy(ctx->d_prior, L.prior.data(), L.prior.size() * sizeof(double), hipMemcpyHostToDevice));
    return DVO_OK;
}

int dvo_solve_batch_fetch(dvo_ctx* ctx, int n_pairs, float* T_out, double* xi_out, int* iters_out, double* err_out)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (n_pairs <= 0 || n_pairs > ctx->max_pairs) return fail(ctx, DVO_ERR_INVALID_ARG, "n_pairs %d outside [1,%d]", n_pairs, ctx->max_pairs);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->refetch_invalid)
        return fail(ctx, DVO_ERR_BAD_STATE, "dvo_solve_batch_fetch: this solve was already fetched and some of its pairs were solved again (Student-t "
                    "fallback); the device buffers hold that partial solve - fetch once per enqueue");
    const bool student_t = ctx->last_moment && n_pairs == ctx->last_key.n_pairs;
    int rc = fetch_results(ctx, n_pairs, student_t);
    if (rc) return rc;
    if (T_out) memcpy(T_out, ctx->hs.T, (size_t)n_pairs * 12 * sizeof(float));
    if (xi_out) memcpy(xi_out, ctx->hs.xi, (size_t)n_pairs * 6 * sizeof(double));
    if (iters_out) memcpy(iters_out, ctx->hs.iters, (size_t)n_pairs * sizeof(int));
    if (err_out) memcpy(err_out, ctx->hs.err, (size_t)n_pairs * sizeof(double));
    if (!student_t) return DVO_OK;
    ctx->last_moment = false;           // the counters below are added once per enqueued solve, whatever the caller fetches again
    // Per-pair fallback.  A pair with an iteration whose one-pass bound was not met (flag 1) is solved again with the passes as
    // written at the per-iteration levels; a pair whose IRLS does not have the two-pass shape those levels evaluate (flag 2, from
    // either solve) is solved again by the persistent kernel at every level, which runs the reference's loop (until |d lambda| <
    // 1e-3, at most 50 passes).  Only the flagged pairs are solved again, as one compacted batch per form; the others keep their
    // results.
    std::vector<int> bound_only, shape;
    for (int i = 0; i < n_pairs; ++i) {
        if (ctx->h_pair_flags[i] & 2) shape.push_back(i);
        else if (ctx->h_pair_flags[i] & 1) bound_only.push_back(i);
    }
    if (bound_only.empty() && shape.empty()) return DVO_OK;
    ctx->refetch_invalid = true;
    ctx->st_fallbacks += 1;
    ctx->st_fallback_pairs += (int64_t)(bound_only.size() + shape.size());
    if (ctx->knobs.st_debug)
        fprintf(stderr, "dvo: Student-t: %zu pair(s) of %d solved again with the passes as written, %zu by the reference's loop (iterations flagged: %d; "
                        "IRLS shape %d, moment bound %d, predicted bound %d)\n", bound_only.size(), n_pairs, shape.size(),
                ctx->h_inexact ? ctx->h_inexact[0] : 0, ctx->h_inexact ? ctx->h_inexact[1] : 0, ctx->h_inexact ? ctx->h_inexact[2] : 0, ctx->h_inexact ? ctx->h_inexact[3] : 0);
    if (!bound_only.empty()) {
        rc = resolve_pairs(ctx, bound_only, 1, T_out, xi_out, iters_out, err_out);
        if (rc) return rc;
        for (size_t i = 0; i < bound_only.size(); ++i)
            if (ctx->h_pair_flags[i] & 2) shape.push_back(bound_only[i]);        // ... and the passes as written found a third pass
    }
    if (!shape.empty()) {
        rc = resolve_pairs(ctx, shape, 2, T_out, xi_out, iters_out, err_out);
        if (rc) return rc;
    }
    return DVO_OK;
}

#ifdef DVO_EXPERIMENTS
int dvo_debug_small_stamps(dvo_ctx* ctx, int n_pairs, uint64_t* out, int reset)
{
    if (!ctx || !out || n_pairs <= 0 || n_pairs > ctx->max_pairs) return DVO_ERR_INVALID_ARG;
    if (!ctx->d_small_stamps) return fail(ctx, DVO_ERR_BAD_STATE, "dvo_debug_small_stamps: needs a library built with -DDVO_EXPERIMENTS and a ctx created under DVO_SMALL_STAMPS");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_small_stamps, (size_t)n_pairs * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(ctx, hipMemset(ctx->d_small_stamps, 0, (size_t)ctx->max_pairs * 8 * sizeof(unsigned long long)));
    return DVO_OK;
}
#endif

int dvo_student_t_fallbacks(const dvo_ctx* ctx, int64_t* count)
{
    if (!ctx || !count) return DVO_ERR_INVALID_ARG;
    *count = ctx->st_fallbacks;
    return DVO_OK;
}

int dvo_student_t_fallback_pairs(const dvo_ctx* ctx, int64_t* pairs)
{
    if (!ctx || !pairs) return DVO_ERR_INVALID_ARG;
    *pairs = ctx->st_fallback_pairs;
    return DVO_OK;
}

int dvo_solve_batch(dvo_ctx* ctx, int n_pairs, const int* prev_slot, const int* cur_slot, int levels,
                    int max_iterations, double tolerance, int max_increased_steps_allowed, int variant,
                    int weight_mode, double weight_param, const float* T_init, float* T_out, double* xi_out,
                    int* iters_out, double* err_out)
{
    int rc = dvo_solve_batch_enqueue(ctx, n_pairs, prev_slot, cur_slot, levels, max_iterations, tolerance,
                                     max_increased_steps_allowed, variant, weight_mode, weight_param, T_init);
    if (rc) return rc;
    return dvo_solve_batch_fetch(ctx, n_pairs, T_out, xi_out, iters_out, err_out);
}

int dvo_debug_divide(dvo_ctx* ctx, const float* a, const float* b, float* out, float* ref, int n)
{
    if (!ctx || !b || !out || !ref || n <= 0) return DVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, (size_t)n * 4 * sizeof(float)));
    hipError_t e = a ? hipMemcpyAsync(d, a, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpyAsync(d + n, b, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = dvo_launch_debug_divide(a ? d : nullptr, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, n, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ref, d + 3 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    HIP_TRY(ctx, e);
    return DVO_OK;
}

int dvo_debug_valu_rate(dvo_ctx* ctx, int waves_per_simd, int iterations, int* composition, double* ns_per_wave_instr)
{
    if (!ctx || !ns_per_wave_instr) return DVO_ERR_INVALID_ARG;
    if (waves_per_simd < 1 || waves_per_simd > 8 || iterations < 1) return fail(ctx, DVO_ERR_INVALID_ARG, "waves_per_simd in [1,8], iterations >= 1");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount;
#ifdef DVO_EXPERIMENTS
    const int n_modes = 72;             // (experiment builds: out[6..13] = other orders and the v_rcp placement cases of tools/ubench/probe_orders.py)
#else
    const int n_modes = 6;
#endif
    for (int mode = 0; mode < n_modes; ++mode) {
        const double per_iter = (double)dvo_valu_rate_instructions(mode, (mode == 0) ? composition : nullptr);
        const int it = (mode == 4) ? (iterations + 1) / 2 : iterations;                              // (128 reciprocals ~ 512 simple instructions)
        HIP_TRY(ctx, dvo_launch_valu_rate(mode, 4, cus, waves_per_simd, nullptr, ctx->stream));      // warm-up: code in the instruction cache
        HIP_TRY(ctx, hipEventRecord(ctx->tm_a, ctx->stream));
        HIP_TRY(ctx, dvo_launch_valu_rate(mode, it, cus, waves_per_simd, nullptr, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->tm_b, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->tm_b));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->tm_a, ctx->tm_b));
        // every SIMD holds waves_per_simd waves, each issuing per_iter instructions per iteration
        ns_per_wave_instr[mode] = 1e6 * (double)ms / ((double)it * per_iter * waves_per_simd);
    }
    return DVO_OK;
}

int dvo_set_prior(dvo_ctx* ctx, double sigma, const double* last_qt, int n_pairs)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!(sigma > 0.0)) { ctx->prior_sigma = 0.0; ctx->prior_has_last = false; return DVO_OK; }
    ctx->prior_sigma = sigma;
    ctx->prior_has_last = false;
    if (last_qt) {
        if (n_pairs <= 0 || n_pairs > ctx->max_pairs) return fail(ctx, DVO_ERR_INVALID_ARG, "n_pairs %d outside [1,%d]", n_pairs, ctx->max_pairs);
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // h_prior may still be in flight
        memcpy(ctx->h_prior, last_qt, (size_t)n_pairs * 7 * sizeof(double));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_prior, ctx->h_prior, (size_t)n_pairs * 7 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        ctx->prior_has_last = true;
    }
    return DVO_OK;
}

int dvo_set_level_iterations(dvo_ctx* ctx, const int* iters, int n_levels)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    if (!iters || n_levels <= 0) { ctx->level_iters_set = false; return DVO_OK; }
    if (n_levels > DVO_MAX_LEVELS) return fail(ctx, DVO_ERR_INVALID_ARG, "n_levels %d exceeds %d", n_levels, DVO_MAX_LEVELS);
    for (int l = 0; l < n_levels; ++l)
        if (iters[l] < 0) return fail(ctx, DVO_ERR_INVALID_ARG, "iterations of level %d negative", l);
    for (int l = 0; l < DVO_MAX_LEVELS; ++l) ctx->level_iters[l] = (l < n_levels) ? iters[l] : iters[n_levels - 1];
    ctx->level_iters_set = true;
    return DVO_OK;
}

int dvo_trace_enable(dvo_ctx* ctx, int max_iterations_traced)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_trace) { (void)hipFree(ctx->d_trace); ctx->d_trace = nullptr; ctx->trace_cap = 0; }
    if (max_iterations_traced <= 0) return DVO_OK;
    const size_t n = (size_t)ctx->max_pairs * max_iterations_traced * (DVO_STRIDE + 18);
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trace, n * sizeof(double)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_trace, 0, n * sizeof(double), ctx->stream));
    ctx->trace_cap = max_iterations_traced;
    return DVO_OK;
}

int dvo_trace_read(dvo_ctx* ctx, int n_pairs, double* out)
{
    if (!ctx || !out) return DVO_ERR_INVALID_ARG;
    if (!ctx->d_trace) return fail(ctx, DVO_ERR_BAD_STATE, "trace not enabled");
    if (n_pairs <= 0 || n_pairs > ctx->max_pairs) return fail(ctx, DVO_ERR_INVALID_ARG, "n_pairs out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)n_pairs * ctx->trace_cap * (DVO_STRIDE + 18);
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_trace, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DVO_OK;
}

int dvo_profile_enable(dvo_ctx* ctx, int on)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    ctx->profiling = on != 0;
    return DVO_OK;
}

int dvo_profile_read(dvo_ctx* ctx, int64_t* launches, double* ms, int reset)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_profile(ctx);
    for (int i = 0; i < DVO_MAX_LEVELS; ++i) {
        if (launches) launches[i] = ctx->prof_launches[i];
        if (ms) ms[i] = ctx->prof_ms[i];
        if (reset) { ctx->prof_launches[i] = 0; ctx->prof_ms[i] = 0.0; }
    }
    return DVO_OK;
}

int dvo_timer_start(dvo_ctx* ctx)
{
    if (!ctx) return DVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->tm_a, ctx->stream));
    return DVO_OK;
}

int dvo_timer_stop(dvo_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return DVO_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->tm_b, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->tm_b));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->tm_a, ctx->tm_b));
    return DVO_OK;
}

}  // extern "C"
