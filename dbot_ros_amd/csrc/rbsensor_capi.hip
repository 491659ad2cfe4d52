// rbsensor_capi.hip -- host side of librbsensor_mi355x.so: the C-ABI declared in
// include/rbsensor_mi355x.h over the kernels in rbsensor_kernels.hip.
//
// Device-resident state per handle (sizes at 640x480, N slots):
//   soup      9 * n_tri doubles, SoA triangle soup                     (368 KB @5 120 tris)
//   frame     rows*cols float, current observation                     (1.2 MB)
//   occ[2]    2 * N * rows*cols float, double-buffered occlusion planes (2.46 MB * N)
//   win[2]    2 * N int4 + one float: a plane is explicit inside its window, the background elsewhere
//   poses / indices / out   per-call staging for the host-pointer API
// There is no CPU path in this library.
#include "rbsensor_kernels.hip"
#include "rbsensor_tracker.hip"
#include "rbsensor_peers.hip"

#include "../../include/rbsensor_mi355x.h"

#include <dlfcn.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>   // (the frame's double -> float staging: AVX2 / AVX-512 where the host has them; other hosts take the scalar loop)
#endif

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <array>
#include <map>
#include <new>
#include <string>
#include <utility>
#include <vector>

using rbs::DevParams;

static_assert(sizeof(rbs_config) == 216 && offsetof(rbs_config, state_slab_px) == 208 && offsetof(rbs_config, likelihood_precision) == 184 &&
                  offsetof(rbs_config, device_ids) == 200,
              "rbs_config layout (ABI 2) is mirrored by dbot_ros_amd/_capi.py and tests/test_capi_cpu.py");

#ifndef RBS_TRACKER_SPLIT_MAX_DEFAULT
#define RBS_TRACKER_SPLIT_MAX_DEFAULT 5000   // (measured, tests/cpp/host_bench --tracker: frame by frame +12-16 % at 1 000-2 000 particles, +6-10 % at 4 000, nothing from 6 000 up)
#endif

struct rbs_handle {
    int device = 0;
    int rows = 0, cols = 0, npx = 0;
    int max_particles = 0;
    int n_bodies = 0;
    DevParams base{};           // static part of the kernel parameters
    double p_ov = 0, p_oo = 0, init_occ = 0, delta_time = 0;
    double* d_soup = nullptr;
    float* d_frame = nullptr;
    double* d_aux = nullptr;    // per-frame-pixel model terms, [npx][4] (precision F64), of a frame ingested from device memory
    double* d_aux_slot[2] = {nullptr, nullptr};   // ... of the host frame uploaded into d_fin[k] (computed on the upload stream)
    const double* cur_aux = nullptr;              // what the kernels read: d_aux, or d_aux_slot[cur_slot]
    float* d_pbg = nullptr;
    float* d_occ[2] = {nullptr, nullptr};
    int cur = 0;
    int pending_frames = 0;     // set_observation calls since the last updating loglikes
    const float* lazy_frame = nullptr;   // frame handed over by rbs_set_observation_device whose ingest
    hipStream_t lazy_stream = nullptr;   //   kernel has not been launched yet (it rides on the next loglikes)
    double* d_poses = nullptr;
    int* d_indices = nullptr;
    double* d_out = nullptr;
    int* d_rects[2] = {nullptr, nullptr};  // [max_particles][4], alternating per call: the previous
                                // call's copy kernel may still be reading its rectangles
    rbs::Groups* d_groups[2] = {nullptr, nullptr};   // [max_particles] per-group rectangles (several bodies), alternating like d_rects
    rbs::Strips* d_strips[2] = {nullptr, nullptr};   // [max_particles] the copy kernel's cells outside the groups' rectangles (windowed planes), alternating alike
    int tracker_split_max = RBS_TRACKER_SPLIT_MAX_DEFAULT;   // rbs_tracker_*: see RBS_OPT_TRACKER_SPLIT_MAX
    bool copy_walk = false;                          // RBS_COPY_WALK=1 (tooling / A-B): several bodies take the walk over the whole region instead
    int* d_parents[2] = {nullptr, nullptr};   // [max_particles] snapshot of the caller's indices, alternating like d_rects
    int4* d_win[2] = {nullptr, nullptr};   // [max_particles] window of each plane, per buffer
    int4* d_win_used = nullptr; // [max_particles] region the copy kernel writes this call
    // window-sized slabs (rbs_config.state_slab_px): a slot holds slab_px floats and the stored region
    // of its plane; 0 = whole planes
    int slab_px = 0;
    size_t plane_stride = 0;    // floats per slot: npx or slab_px
    int4* d_reg[2] = {nullptr, nullptr};   // [max_particles] stored region of each plane, per buffer
    int* d_err = nullptr;       // [2] device: [0] a region did not fit its slab, [1] the largest region asked for so far (px)
    int* h_err = nullptr;       // pinned copy, fetched with the log-likelihoods ([2], [3]: the flags as they were BEFORE the call)
    bool slab_auto = false;     // the slab size is the library's choice (rbs_config.state_slab_px == 0 with many particles)
    bool slab_probed = false;   // ... and has been checked against a first call's regions (rbs_loglikes_device, probe_auto_slabs)
    int* d_bbox = nullptr;      // [4] scratch of rbs_import_plane / rbs_set_occlusion
    bool windowed = true;       // planes valid inside their window only (state_layout dense: whole plane)
    bool many_clusters = false; // a body of more than 256 clusters: the rbs_raster_kernel_many_* instantiations (shared cluster cull)
    // windowed planes whose windows have grown to a large part of the frame are served like whole
    // planes (streaming copy kernel beside two raster blocks per CU); the stored area is sampled
    // on the device every timing_every-th updating call and read back without blocking
    unsigned char* d_wide_flags = nullptr;   // [max_particles][copy blocks per plane], allocated on first use
    unsigned long long* d_area = nullptr;
    unsigned long long* h_area = nullptr;   // pinned
    hipEvent_t ev_area = nullptr;
    bool area_pending = false;
    int area_n = 0;
    bool wide = false;
    double area_frac = 0.0;     // last sampled stored-window area / frame area
    double mid_enter = 0.15;    // above it: two raster blocks per CU, so the windowed copy runs beside them
    // Raster / copy balance: the windowed copy kernel runs beside the persistent raster blocks and
    // a call ends when the later of the two does.  When the copy kernel outlasts the raster kernel
    // (several bodies far apart: one window spans them and the copy fills the gaps) the next calls
    // run a few raster blocks fewer, and take them back when the raster kernel is the later one
    // again.  Decided from the kernel timers of the last timed call; the results do not depend on
    // the number of blocks.  RBS_BALANCE=0 pins the grid.
    bool balance = true;
    int balance_blocks = 0;     // current grid of the raster kernel (0: raster_blocks)
    long balance_seen = 0;      // timed_calls the rule has looked at
    double wide_enter = 0.50, wide_leave = 0.35;   // measured: the streaming copy wins from about half the frame
    int cu_count = 256;
    int smalln_target = 768;    // few particles: aim at about this many work items per call
    int rect_align = 4;         // windowed planes: rectangles move in float4 columns
    int win_chunks = 8;         // row chunks (blocks) per particle of the windowed copy kernel (several bodies: it walks the whole window)
    int win_chunks_single = 2;  // ... one body: it enumerates only the cells outside the rectangle, a tenth of the window
    int upload_chunks = 2;      // pieces a caller's host frame is staged and sent in (RBS_UPLOAD_CHUNKS) ...
    bool quiet = false;         // ... while nothing long of this handle's is running on the device (set by the entry points that
                                // wait for results, cleared by every launch): behind a running raster kernel the second piece's
                                // hand-over waits for that kernel (measured: pipelined tracker 3 750 -> 1 880 frames/s)
    bool async_outstanding = false;   // rbs_loglikes_device calls since the last rbs_synchronize (possibly on the caller's streams)
    float background = 0.f;     // never-covered occlusion level of the current buffer
    int2* d_item_range = nullptr;   // [max_particles] work items of each particle
    int* d_item_particle = nullptr; // [partial_cap] owner of each work item
    int* d_ctr = nullptr;           // [4] work-item counters of even / odd calls (+ [2] the split launch's likelihood-kernel tickets)
    // Split launch (round 5): geometry kernel -> depth tiles in memory -> likelihood kernel, each with its own occupancy
    // (rbsensor_kernels.hip, rbs_depth_kernel / rbs_eval_kernel).  rbs_config has no field for it: RBS_SPLIT in the
    // environment at rbs_create (tooling / A-B), otherwise the library's choice below.
    // Shared trail (round 5): a plane equals a handle-wide background PLANE outside its window instead of a scalar level.  A
    // pixel stays in a window ~730 frames after the object left it, and every child inherits that trail from its parent: with
    // resampling the particles soon share it from a common ancestor, and it was stored, read and stepped once per PARTICLE per
    // frame (a sweeping object: windows of 85 % of the frame, 2.2 M particle-likelihoods/s where the headline has 10.5).  Once
    // the sampled window area exceeds stp_enter the handle switches: the shared plane starts as the scalar background everywhere
    // and is RE-BASED on one particle's plane (inside that plane's window its values), every child of that call is re-measured
    // against it (its window shrinks to where it really differs), and so again every stp_every updating calls while windows
    // are large.  Values are unchanged bit for bit (the plane steps with the same float operations as any stored value); only
    // what is stored changes.  Single-device handles (whole planes or slabs), binary64 likelihood; RBS_SHARED_TRAIL=0 disables.
    bool stp = false, stp_allowed = true;
    float* d_bgp[2] = {nullptr, nullptr};   // [npx] the shared plane of either buffer (indexed like d_occ: by `cur`)
    double stp_enter = 0.10;
    int stp_every = 32;
    long stp_last_rebase = -1000000;
    long stp_rebases = 0;
    long stp_block_until = 0;               // re-basing did not shrink the windows (particles that share no ancestor): not again before this call
    // Round 6: the shared trail on handles whose planes OTHERS read in place.  The planes of every shard of a group (every rank of
    // an attached job) are stored against ONE shared plane, of which every device keeps its own identical copy: it starts as the
    // scalar level everywhere, steps with the same operations on every device, and is re-based by all of them in the SAME call on
    // the SAME global slot (a shard reads that slot's plane from its owner, in place) -- so "outside its window" means the same
    // values whichever device asks.  A group takes ONE decision per call for all its shards (group_begin_call, from the largest
    // window fraction any shard has sampled); attached ranks are told by their caller (rbs_shared_trail_rebase: the same call on
    // every rank before the same step), who has the collective to agree on it.
    struct StpNow { int rebase = -1; bool entering = false; } stp_now;   // group: this call's directive to the shards
    bool stp_leave_pending = false;         // group: the call in flight leaves the shared trail (the group's own state follows at the next call)
    int stp_request = -1;                   // rbs_shared_trail_rebase: >= 0 re-base on that GLOBAL slot at the next updating call (entering if need be), -2: leave
    bool ipc_exported = false;              // other processes may map this handle's planes (rbs_ipc_export): it no longer switches on its own
    int4* d_bgp_box = nullptr;              // [1] bounding box of the shared plane's values that differ from the scalar background
    bool split = false;
    unsigned* d_depth = nullptr;    // [depth_items][kDepthTilePx]
    size_t depth_items = 0;
    int depth_blocks = 0, eval_blocks = 0;   // the two persistent grids
    int* d_done = nullptr;      // [max_particles] finished work items per particle
    double* d_partial = nullptr; // [partial_cap] per-item partial sums
    unsigned long long* d_phase = nullptr;  // RBS_PHASE_TIMING builds
    size_t partial_cap = 0;
    float* d_cluster_sphere = nullptr;
    float* d_cluster_cone = nullptr;
    double* d_cluster_vtx = nullptr;   // [clusters][3][64] unique vertices of each cluster (vertex sharing)
    int* d_cluster_nv = nullptr;       // [clusters]
    unsigned* d_tri_local = nullptr;   // [n_tri] positions of a triangle's vertices in its cluster's list
    float* d_vtx = nullptr;         // [sum of vertex counts][4] float32 vertices (screen rectangles)
    float* d_tri_plane = nullptr;   // [n_tri][4] model-space plane of each triangle (float32 pre-cull)
    int precision = RBS_PRECISION_F64;
    // Stamped planes (rbs_config.occlusion_mode = RBS_OCC_REFERENCE; DevParams::exact): a slot is plane_px floats + plane_px
    // 16-bit ages, plane_stride floats in all.
    bool one_body_kernel = true;    // single-body models take rbs_raster_kernel_one_f64 (RBS_ONE_BODY=0, tooling: the general kernel)
    bool exact = false;
    int age_max = 0;            // ages beyond it are background
    double* d_ptab = nullptr;   // [age_max + 1][2] the propagation table
    std::vector<double> ptab;   // ... its host copy
    long update_clock = 0;      // frames between rbs_reset and the last updating call: the planes' epoch
    float* d_tmp_plane = nullptr;   // [npx] scratch of the plane hooks (the shared plane's effective values)
    float* d_render = nullptr;
    // pinned staging of the host-pointer API: poses | indices in one block (one H2D copy), the
    // log-likelihoods in another, and the event the caller waits on (the out copy alone -- the
    // planes' copy kernel on the second stream is joined by the next call, as in the device API)
    unsigned char* h_in = nullptr;
    double* h_out = nullptr;
    unsigned char* d_in = nullptr;     // device image of h_in
    unsigned char* h_in_dev = nullptr; // h_in / h_out as the device addresses them (kernels read / write them in place)
    double* h_out_dev = nullptr;
    size_t in_idx_off = 0;             // byte offset of the indices inside h_in / d_in
    hipEvent_t ev_out = nullptr;
    float* h_frames[2] = {nullptr, nullptr};   // pinned frame staging, alternating
    float* h_frames_dev[2] = {nullptr, nullptr};   // ... as the device addresses them
    size_t frame_pull_bytes = 128 * 1024;      // frames up to this size are read from the staging buffer by a kernel (RBS_FRAME_PULL_BYTES)
    hipEvent_t ev_frame[2] = {nullptr, nullptr};   // the upload out of h_frames[k] into d_fin[k] has finished
    int frame_slot = 0;
    // Host frames travel on their own stream into one of two device staging buffers, and the launch
    // stream only waits for the upload where it ingests the frame: a frame handed over while the
    // previous frame's kernels are still running (rbs_tracker_submit) is uploaded beside them.
    hipStream_t up_stream = nullptr;
    float* d_fin[2] = {nullptr, nullptr};
    hipEvent_t ev_used[2] = {nullptr, nullptr};    // the ingest kernel that read d_fin[k] has run
    int lazy_slot = -1;                            // d_fin slot the pending (lazy) frame sits in
    // A host frame needs no ingest on the launch stream (precision F64: its per-pixel terms are
    // computed behind the copy on the upload stream, into d_aux_slot[k]): the staging
    // image d_fin[k] IS the observation until the next frame replaces it, and only the raster
    // kernel (not the rectangles kernel before it) waits for the upload.
    const float* cur_frame = nullptr;              // what the kernels read: d_frame, or d_fin[cur_slot]
    int cur_slot = -1;
    int frame_wait = -1;                           // ev_frame[] the raster launches wait for
    hipEvent_t ev_reader = nullptr;                // orders the handle's stream after a reader on a caller's stream
    // the older routes, kept behind environment switches read at rbs_create (A/B runs, differential tests):
    bool frame_ingest = false;                     // RBS_FRAME_INGEST=1: host frames are copied into d_frame by the ingest kernel
    bool host_copies = false;                      // RBS_HOST_STAGED_COPIES=1: poses / results travel as H2D / D2H copies
    float* h_frame = nullptr;   // = h_frames[frame_slot]
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // the copy kernel runs here, beside the raster kernel
    hipEvent_t ev_fork = nullptr;
    int copy_blocks = 1 << 30;  // cap on the copy grid (one block per (particle, band) below it)
    int raster_blocks = 768;    // persistent raster grid: 3 per CU
    int copy_rows = 2;          // rows per block of rbs_copy_rows_kernel (0: banded kernel)
    int copy_tpb = 64;          // threads per copy block: one wave = 1 KB of a row (0: a block spans a row)
    const char* tile_override = nullptr;  // RBS_TILE env (tuning)
    // timing ring: HIP events around the whole call (on the launch stream) and around the copy
    // kernel (on the copy stream) for the last kRing loglikes calls
    static constexpr int kRing = 256;
    hipEvent_t ev_start[kRing] = {}, ev_stop[kRing] = {}, ev_copy_start[kRing] = {}, ev_join[kRing] = {};
    hipEvent_t ev_raster_start[kRing] = {}, ev_raster_stop[kRing] = {}, ev_copy_stop[kRing] = {};
    bool ring_update[kRing] = {};
    // every event is a packet the stream must retire between two kernels: only every
    // timing_every-th call carries the timing events (ring slots count TIMED calls); ev_join, which
    // orders later work after the copy kernel, is recorded on every updating call
    int timing_every = 8;
    long timed_calls = 0;
    long calls = 0;
    int join_pending = -1;      // ring slot whose copy kernel later work on the planes must wait for
    std::string err;
    bool frame_acquired = false;       // rbs_acquire_frame_buffer without its rbs_commit_frame_buffer yet
    int prefetched_slot = -1;          // staging slot holding a frame uploaded ahead of its turn (rbs_loglikes_prefetch), -1: none
    const float* borrowed_f32 = nullptr;   // ... the same for a float frame (the device tracker's frames at small particle counts)
    const double* borrowed = nullptr;  // rbs_set_observation_borrowed: the caller's frame, staged by the next host-pointer likelihood call
                                       // while that call's geometry kernel runs (or by whatever else needs the observation first)
    // A call that failed half-way through a fan-out (some shards enqueued, others not) or between a
    // tracker's buffer swaps leaves the handle's double buffers out of step: every later call is
    // refused with this message until rbs_reset (rbs_tracker_initialize) re-establishes a known state.
    bool poisoned = false;
    std::string poison_msg;
    // test hook, builds with -DRBS_TEST_HOOKS only (tests/test_gpu_multidevice.py): RBS_TEST_FAULT="<shard>:<call>" read at rbs_create makes
    // the group's <call>-th rbs_loglikes fail on shard <shard> after the shards before it were enqueued
    int fault_shard = -1;
    long fault_call = -1, group_calls = 0;
    // ---- several devices in one handle (rbs_config.n_devices > 1) ----
    // A GROUP handle owns one single-device handle ("shard") per device; global slot g lives on
    // shard g / shard_cap.  Every call is fanned out to the shards by the calling thread; a
    // shard's kernels read parents that live on other shards in place (peer access).
    std::vector<rbs_handle*> shards;   // group: its shards
    rbs_handle* group = nullptr;       // shard: its group
    int shard_index = 0;
    int shard_cap = 0;                 // global slots per device
    hipEvent_t ev_done = nullptr;      // shard: its last loglikes call has finished, planes included
    struct Rccl* rccl = nullptr;       // group: communicators for the log-likelihood all-gather (device tracker)
    // ---- particle sharding across PROCESSES (rbs_ipc_attach): the other ranks' plane buffers, window and
    // region tables mapped into this process; rank r owns global slots [r * max_particles, ...)
    int peer_world = 0, peer_rank = 0;
    const float* peer_occ[rbs::kMaxDevices][2] = {};
    const int4* peer_win[rbs::kMaxDevices][2] = {};
    const int4* peer_reg[rbs::kMaxDevices][2] = {};
    void* peer_mapped[rbs::kMaxDevices][6] = {};    // what hipIpcCloseMemHandle gets back
    void* d_peer_scratch = nullptr;                 // rbs_peer_resample: cdf [N] + 2 x [tiles] doubles + 2 x [n] ints
    size_t peer_scratch_bytes = 0;
    const float* snap_occ[rbs::kMaxDevices] = {};   // group: every shard's CURRENT planes / windows as of the start
    const int4* snap_win[rbs::kMaxDevices] = {};    //   of the call being fanned out (shards flip buffers one by one)
    const int4* snap_reg[rbs::kMaxDevices] = {};
};

// RCCL, bound at run time (dlopen): a single-device handle never needs it, and a process that
// already carries torch's copy of the library keeps exactly one.
struct Rccl {
    void* lib = nullptr;
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::vector<comm_t> comms;
};

namespace {

thread_local std::string g_create_error;

std::string fmt(const char* f, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

#define RBS_HIP(h, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (void)hipGetLastError(); /* HIP's last error is sticky: it must not fail a later call */ \
            (h)->err = fmt("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,   \
                           __LINE__);                                                         \
            return e_ == hipErrorOutOfMemory ? RBS_ERR_OUT_OF_MEMORY : RBS_ERR_HIP;           \
        }                                                                                     \
    } while (0)

int32_t fail(rbs_handle* h, int32_t code, const std::string& msg)
{
    h->err = msg;
    return code;
}

// alpha/beta of the occlusion process over n frames (affine form of SURVEY A.5), rounded
// once to float -- the same rule as oracle orc_eager_coeffs().
void occlusion_coeffs(const rbs_handle* h, int n_frames, float* alpha, float* beta)
{
    const double c = h->p_oo - h->p_ov;
    const double a = std::exp(((double)n_frames * h->delta_time) * std::log(c));
    const double g = (1.0 - h->p_oo) * (a - 1.0) / (c - 1.0);
    *alpha = (float)a;
    *beta = (float)((1.0 - a) - g);
}

// OcclusionModel propagate as the oracle evaluates it (oracle orc_propagate): the same expressions, the same libm.
double host_propagate(const rbs_handle* h, double occ, double dt)
{
    const double c = h->p_oo - h->p_ov;
    const double pow_c = std::exp(dt * std::log(c));
    const double new_visible = pow_c * (1.0 - occ) + (1.0 - h->p_oo) * (pow_c - 1.0) / (c - 1.0);
    return 1.0 - new_visible;
}
// Stamped planes: the prior of a never-covered pixel `frames` frames after rbs_reset (the oracle's slot of initial values, stamp 0).
float exact_background(const rbs_handle* h, long frames)
{
    return (float)host_propagate(h, (double)(float)h->init_occ, (double)frames * h->delta_time);
}
// Floats per slot of `px` pixels: the values, then the ages (two per float), rounded to whole float4s.
size_t exact_stride(size_t px) { return (px + (px + 1) / 2 + 3) & ~(size_t)3; }
constexpr double kExactTau = 0x1p-40;   // c^(age_max dt) <= this: what a pixel declared background can still differ by

int copy_bands_for(int rows, int cols)
{
    // ~3840 float4 (60 KB) per copy block; an EVEN band count keeps 1+bands odd so raster
    // blocks (block index = particle*(1+bands)) rotate over all 8 XCDs.
    const long n4 = ((long)rows * cols + 3) / 4;
    long b = (n4 + 3839) / 3840;
    if (b < 2) b = 2;
    if (b & 1) ++b;
    if (b > rows) b = (rows & 1) ? rows + 1 : rows;
    return (int)b;
}

// Most work items any rectangle inside the frame can split into (sizes the partial-sum buffer).
size_t tiles_upper_bound(int cols, int rows, int max_w, int cap_px)
{
    size_t worst = 1;
    for (int rw = 16; rw <= ((cols + 15) & ~15); rw += 16) {
        const rbs::TileGrid g = rbs::tile_grid(rw, rows, max_w, cap_px);
        worst = std::max(worst, (size_t)g.nx * g.ny);
    }
    return worst;
}

// Launch the ingest of a frame rbs_set_observation_device left pending, on the stream it was
// given for; `then` (if not that stream) is ordered after it.
int32_t flush_lazy_frame(rbs_handle* h, hipStream_t then)
{
    if (!h->lazy_frame) return RBS_OK;
    const size_t n = (size_t)h->npx;
    // (precision F32 keeps no per-pixel terms: a frame already in the handle's buffer needs nothing)
    if (h->d_aux || h->lazy_frame != h->d_frame)
    hipLaunchKernelGGL(rbs::frame_aux_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->lazy_stream,
                       h->lazy_frame, h->d_aux, h->d_pbg, h->npx, h->base.tw, h->base.ms, h->base.sf,
                       h->base.lambda, h->lazy_frame == h->d_frame ? (float*)nullptr : h->d_frame);
    RBS_HIP(h, hipGetLastError());
    h->lazy_frame = nullptr;
    if (h->lazy_slot >= 0) { RBS_HIP(h, hipEventRecord(h->ev_used[h->lazy_slot], h->lazy_stream)); h->lazy_slot = -1; }
    if (then != h->lazy_stream) {
        RBS_HIP(h, hipEventRecord(h->ev_fork, h->lazy_stream));
        RBS_HIP(h, hipStreamWaitEvent(then, h->ev_fork, 0));
    }
    return RBS_OK;
}

// The instantiations of the raster kernel: updating or read-only call, likelihood precision,
// whole planes or slabs, and the same eight again for object models with a body of many clusters.
void launch_raster(const rbs_handle* h, bool update, dim3 grid, dim3 block, size_t smem, hipStream_t s, const DevParams& P)
{
    if (h->exact && h->one_body_kernel && !h->many_clusters && h->n_bodies == 1 && !P.groups) {   // stamped planes, one body
#define RBS_X(U, S, T) hipLaunchKernelGGL((rbs::rbs_raster_kernel_exact_one_f64<U, S, T>), grid, block, smem, s, P); break
        switch ((update ? 4 : 0) | (h->slab_px ? 2 : 0) | (P.bgp_src ? 1 : 0)) {
            case 0: RBS_X(false, false, false);  case 1: RBS_X(false, false, true);
            case 2: RBS_X(false, true, false);   case 3: RBS_X(false, true, true);
            case 4: RBS_X(true, false, false);   case 5: RBS_X(true, false, true);
            case 6: RBS_X(true, true, false);    default: RBS_X(true, true, true);
        }
#undef RBS_X
        return;
    }
    if (h->exact) {   // stamped planes (binary64): updating or not, whole planes or slabs, many clusters, shared background plane
#define RBS_X(U, S, M, T) hipLaunchKernelGGL((rbs::rbs_raster_kernel_exact_f64<U, S, M, T>), grid, block, smem, s, P); break
        switch ((update ? 8 : 0) | (h->slab_px ? 4 : 0) | (h->many_clusters ? 2 : 0) | (P.bgp_src ? 1 : 0)) {
            case 0: RBS_X(false, false, false, false);  case 1: RBS_X(false, false, false, true);
            case 2: RBS_X(false, false, true, false);   case 3: RBS_X(false, false, true, true);
            case 4: RBS_X(false, true, false, false);   case 5: RBS_X(false, true, false, true);
            case 6: RBS_X(false, true, true, false);    case 7: RBS_X(false, true, true, true);
            case 8: RBS_X(true, false, false, false);   case 9: RBS_X(true, false, false, true);
            case 10: RBS_X(true, false, true, false);   case 11: RBS_X(true, false, true, true);
            case 12: RBS_X(true, true, false, false);   case 13: RBS_X(true, true, false, true);
            case 14: RBS_X(true, true, true, false);    default: RBS_X(true, true, true, true);
        }
#undef RBS_X
        return;
    }
    if (P.bgp_src && h->precision == RBS_PRECISION_F32) {   // the shared background plane, float32 likelihood
#define RBS_X(U, S, M) hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f32<U, S, M>), grid, block, smem, s, P); break
        switch ((update ? 4 : 0) | (h->slab_px ? 2 : 0) | (h->many_clusters ? 1 : 0)) {
            case 0: RBS_X(false, false, false);  case 1: RBS_X(false, false, true);
            case 2: RBS_X(false, true, false);   case 3: RBS_X(false, true, true);
            case 4: RBS_X(true, false, false);   case 5: RBS_X(true, false, true);
            case 6: RBS_X(true, true, false);    default: RBS_X(true, true, true);
        }
#undef RBS_X
        return;
    }
    if (P.bgp_src) {   // the shared background plane (binary64)
        switch ((update ? 4 : 0) | (h->slab_px ? 2 : 0) | (h->many_clusters ? 1 : 0)) {
            case 0: hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f64<false, false, false>), grid, block, smem, s, P); break;
            case 1: hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f64<false, false, true>), grid, block, smem, s, P); break;
            case 2: hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f64<false, true, false>), grid, block, smem, s, P); break;
            case 3: hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f64<false, true, true>), grid, block, smem, s, P); break;
            case 4: hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f64<true, false, false>), grid, block, smem, s, P); break;
            case 5: hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f64<true, false, true>), grid, block, smem, s, P); break;
            case 6: hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f64<true, true, false>), grid, block, smem, s, P); break;
            default: hipLaunchKernelGGL((rbs::rbs_raster_kernel_stp_f64<true, true, true>), grid, block, smem, s, P); break;
        }
        return;
    }
    if (h->one_body_kernel && !h->many_clusters && h->precision == RBS_PRECISION_F64 && h->n_bodies == 1 && !P.groups) {
        switch ((update ? 2 : 0) | (h->slab_px ? 1 : 0)) {
            case 0: hipLaunchKernelGGL((rbs::rbs_raster_kernel_one_f64<false, false>), grid, block, smem, s, P); break;
            case 1: hipLaunchKernelGGL((rbs::rbs_raster_kernel_one_f64<false, true>), grid, block, smem, s, P); break;
            case 2: hipLaunchKernelGGL((rbs::rbs_raster_kernel_one_f64<true, false>), grid, block, smem, s, P); break;
            default: hipLaunchKernelGGL((rbs::rbs_raster_kernel_one_f64<true, true>), grid, block, smem, s, P); break;
        }
        return;
    }
    const int key = (h->many_clusters ? 8 : 0) | (update ? 4 : 0) | (h->precision == RBS_PRECISION_F32 ? 2 : 0) | (h->slab_px ? 1 : 0);
    switch (key) {
        case 0: hipLaunchKernelGGL((rbs::rbs_raster_kernel_f64<false, false>), grid, block, smem, s, P); break;
        case 1: hipLaunchKernelGGL((rbs::rbs_raster_kernel_f64<false, true>), grid, block, smem, s, P); break;
        case 2: hipLaunchKernelGGL((rbs::rbs_raster_kernel_f32<false, false>), grid, block, smem, s, P); break;
        case 3: hipLaunchKernelGGL((rbs::rbs_raster_kernel_f32<false, true>), grid, block, smem, s, P); break;
        case 4: hipLaunchKernelGGL((rbs::rbs_raster_kernel_f64<true, false>), grid, block, smem, s, P); break;
        case 5: hipLaunchKernelGGL((rbs::rbs_raster_kernel_f64<true, true>), grid, block, smem, s, P); break;
        case 6: hipLaunchKernelGGL((rbs::rbs_raster_kernel_f32<true, false>), grid, block, smem, s, P); break;
        case 7: hipLaunchKernelGGL((rbs::rbs_raster_kernel_f32<true, true>), grid, block, smem, s, P); break;
        case 8: hipLaunchKernelGGL((rbs::rbs_raster_kernel_many_f64<false, false>), grid, block, smem, s, P); break;
        case 9: hipLaunchKernelGGL((rbs::rbs_raster_kernel_many_f64<false, true>), grid, block, smem, s, P); break;
        case 10: hipLaunchKernelGGL((rbs::rbs_raster_kernel_many_f32<false, false>), grid, block, smem, s, P); break;
        case 11: hipLaunchKernelGGL((rbs::rbs_raster_kernel_many_f32<false, true>), grid, block, smem, s, P); break;
        case 12: hipLaunchKernelGGL((rbs::rbs_raster_kernel_many_f64<true, false>), grid, block, smem, s, P); break;
        case 13: hipLaunchKernelGGL((rbs::rbs_raster_kernel_many_f64<true, true>), grid, block, smem, s, P); break;
        case 14: hipLaunchKernelGGL((rbs::rbs_raster_kernel_many_f32<true, false>), grid, block, smem, s, P); break;
        default: hipLaunchKernelGGL((rbs::rbs_raster_kernel_many_f32<true, true>), grid, block, smem, s, P); break;
    }
}

int32_t stage_borrowed(rbs_handle* h);

// The shared plane's own step on stream s: bgp[1 - cur] = step(bgp[cur]), re-based first on local slot `rebase` of the planes
// (occ, win, reg) -- its owner's, wherever they live -- or (-2) reset to the scalar level.
int32_t launch_bgp_step(rbs_handle* h, const DevParams& P, const float* occ, const int4* win, const int4* reg, int rebase, hipStream_t s)
{
    if (h->exact)
        hipLaunchKernelGGL(rbs::rbs_bgp_step_exact_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s, h->d_bgp[h->cur], h->d_bgp[1 - h->cur],
                           occ, win, reg, P.plane_stride, P.plane_px, rebase, h->rows, h->cols, P.elapsed2, P.bg_new);
    else
        hipLaunchKernelGGL(rbs::rbs_bgp_step_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s, h->d_bgp[h->cur], h->d_bgp[1 - h->cur],
                           occ, win, reg, P.plane_stride, rebase, h->rows, h->cols, P.alpha, P.beta, P.bg_new);
    RBS_HIP(h, hipGetLastError());
    return RBS_OK;
}

// The shared trail's policy (rbs_handle::stp), one decision per call: `o` owns the state -- a handle of its own, or the GROUP for
// all its shards.  Returns this call's directive: rebase >= 0 (re-base on that slot's plane; entering: the shared plane starts as
// the scalar level first), -2 (back to the scalar background), -1 (nothing).
rbs_handle::StpNow stp_decide(rbs_handle* o, bool update, double area_frac)
{
    rbs_handle::StpNow d;
    const long since = o->calls - o->stp_last_rebase;
    if (update && o->stp && area_frac > o->wide_enter && since >= 16 && since < o->stp_every) {
        // re-basing did not help: the windows are still most of the frame a sample or two later -- the particles share no
        // ancestor (a synthetic permutation of parents; a filter that never resamples).  Back to the scalar background and
        // the whole-plane machinery that serves such windows best: this call re-measures every child against the scalar
        // level over the bounding box of the shared plane's own values.
        d.rebase = -2;
    } else if (update && !o->stp && area_frac > o->stp_enter && o->calls > 0 && o->calls >= o->stp_block_until) {
        d.entering = true;
        d.rebase = 0;
        o->stp = true;
    } else if (update && o->stp && area_frac > o->stp_enter && since >= o->stp_every) {
        d.rebase = 0;
    }
    if (o->stp && d.rebase >= 0) { o->stp_last_rebase = o->calls; o->stp_rebases += 1; }
    return d;
}

int32_t enqueue_loglikes(rbs_handle* h, const double* d_poses, const int* d_indices, int n,
                         bool update, double* d_out, hipStream_t s, const double* host_poses = nullptr, const double* host_deltas = nullptr)
{
    h->quiet = false;
    // a borrowed frame (rbs_set_observation_borrowed) is staged BETWEEN the two kernels of the split launch, while the
    // geometry kernel -- which needs no frame -- runs; a handle that cannot launch that way stages it here and now
    if ((h->borrowed || h->borrowed_f32) && !(h->precision == RBS_PRECISION_F64 && h->windowed && s == h->stream))
        if (int32_t rc = stage_borrowed(h)) return rc;
    DevParams P = h->base;
    occlusion_coeffs(h, h->pending_frames, &P.alpha, &P.beta);
    P.bg_old = h->background;
    P.bg_new = std::fmaf(P.alpha, h->background, P.beta);
    P.exact = h->exact ? 1 : 0;
    P.plane_px = h->slab_px ? h->slab_px : h->npx;
    if (h->exact) {   // stamped planes: nothing is stepped; a never-covered pixel's prior comes from the model clock
        const unsigned e = (unsigned)std::min(h->pending_frames, 0xffff);
        P.elapsed2 = e | (e << 16);
        P.age_max = h->age_max;
        P.ptab = h->d_ptab;
        P.bg_new = exact_background(h, h->update_clock + h->pending_frames);
        P.bg_old = P.bg_new;
    }
    P.windowed = h->windowed ? 1 : 0;
    P.rect_align = h->windowed ? h->rect_align : rbs::kRectAlign;
    P.win_src = h->d_win[h->cur];
    P.win_dst = h->d_win[1 - h->cur];
    P.win_used = h->d_win_used;
    P.frame = h->cur_frame;
    P.aux = h->cur_aux;
    P.pbg = h->d_pbg;
    P.occ_src = h->d_occ[h->cur];
    P.occ_dst = h->d_occ[1 - h->cur];
    P.poses = d_poses;
    P.poses_src = host_poses;   // pinned host memory the rectangles kernel pulls the poses from (into d_poses)
    P.deltas_src = host_deltas; // ... or composes them from (rbs_loglikes_deltas)
    P.indices = d_indices;
    P.slots = h->max_particles;
    P.n_dev = 1;
    P.shard_cap = h->max_particles;
    P.slab_px = h->slab_px;
    P.plane_stride = (int)h->plane_stride;
    P.reg_src = h->d_reg[h->cur];
    P.reg_dst = h->d_reg[1 - h->cur];
    P.err = h->d_err;
    if (h->group) {   // the planes and windows of every shard of the group: parents are global slots
        rbs_handle* g = h->group;
        P.n_dev = (int)g->shards.size();
        P.shard_cap = g->shard_cap;
        P.slots = P.n_dev * g->shard_cap;
        for (int k = 0; k < P.n_dev; ++k) {
            P.occ_src_dev[k] = g->snap_occ[k];
            P.win_src_dev[k] = g->snap_win[k];
            P.reg_src_dev[k] = g->snap_reg[k];
        }
    }
    if (h->peer_world > 1) {   // the same table, the other devices being other PROCESSES' handles (every rank flips its buffers in step)
        P.n_dev = h->peer_world;
        P.shard_cap = h->max_particles;
        P.slots = h->peer_world * h->max_particles;
        for (int k = 0; k < P.n_dev; ++k) {
            P.occ_src_dev[k] = k == h->peer_rank ? h->d_occ[h->cur] : h->peer_occ[k][h->cur];
            P.win_src_dev[k] = k == h->peer_rank ? h->d_win[h->cur] : h->peer_win[k][h->cur];
            P.reg_src_dev[k] = k == h->peer_rank ? h->d_reg[h->cur] : h->peer_reg[k][h->cur];
        }
    }
    P.out = d_out;
    P.n = n;
    const int slot = (int)(h->calls % rbs_handle::kRing);          // ev_join
    const bool timed = h->timing_every <= 1 || h->calls < 4 || h->calls % h->timing_every == 0;   // and the first few, so short sessions have a warm sample
    const int tslot = (int)(h->timed_calls % rbs_handle::kRing);    // timing events
    if (timed) h->ring_update[tslot] = update;
    // tile limits: as large as the LDS tile allows (a particle's rectangle is then usually ONE
    // work item, whatever its aspect); with fewer particles than persistent blocks the pixel
    // budget is divided so that a rectangle splits into about blocks/n row bands
    if (h->windowed && update && h->area_pending && hipEventQuery(h->ev_area) == hipSuccess) {
        const double frac = (double)*h->h_area / ((double)h->area_n * (double)h->npx);   // sampled a few calls ago
        h->area_frac = frac;
        if (frac > h->wide_enter && !h->slab_px && !h->exact) h->wide = true;   // (slabs: regions that large do not fit anyway; stamped planes: the windowed copy serves every size)
        else if (frac < h->wide_leave) h->wide = false;
        h->area_pending = false;
    }
    // shared trail: enter once the windows have grown, re-base while they stay large (see rbs_handle::stp)
    int rebase = -1;          // >= 0: re-base on that (global) slot's plane; -2: back to the scalar background
    bool stp_leaving = false;
    P.bgp_src = nullptr; P.bgp_dst = nullptr; P.rebase_box = nullptr;
    rbs_handle* const so = h->group ? h->group : h;   // who owns the shared trail's state: the group for all its shards
    if (h->windowed && so->stp_allowed) {
        rbs_handle::StpNow d;
        if (h->group) {
            d = so->stp_now;                               // (decided once for every shard: group_begin_call)
        } else if (h->peer_world > 1 || h->ipc_exported || h->stp_request != -1) {   // the caller's directive (rbs_shared_trail_rebase): attached ranks always
            if (update && h->stp_request != -1) {
                if (h->stp_request >= 0) {
                    d.rebase = h->stp_request;
                    d.entering = !h->stp;
                    h->stp_last_rebase = h->calls;
                    h->stp_rebases += 1;
                } else if (h->stp) {
                    d.rebase = -2;
                }
                h->stp_request = -1;
            }
        } else {
            d = stp_decide(h, update, h->area_frac);
        }
        rebase = d.rebase;
        stp_leaving = rebase == -2;
        if (d.entering || (h->group && so->stp && !h->stp && !stp_leaving)) {   // this device's copy of the shared plane: the scalar level everywhere
            for (int k = 0; k < 2; ++k)
                if (!h->d_bgp[k]) RBS_HIP(h, hipMalloc(&h->d_bgp[k], sizeof(float) * (h->exact ? exact_stride((size_t)h->npx) : (size_t)h->npx)));
            if (!h->d_bgp_box) RBS_HIP(h, hipMalloc(&h->d_bgp_box, sizeof(int4)));
            hipLaunchKernelGGL(rbs::rbs_fill_kernel, dim3(256), dim3(256), 0, s, h->d_bgp[h->cur], (size_t)h->npx, h->background);
            RBS_HIP(h, hipGetLastError());
            if (h->exact)   // (every pixel background: ages 0xffff)
                RBS_HIP(h, hipMemsetAsync(h->d_bgp[h->cur] + h->npx, 0xff, sizeof(unsigned short) * (size_t)h->npx, s));
            h->stp = true;
            h->wide = false;
        }
        if (stp_leaving && h->stp) {
            hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_bgp_box, 1, make_int4(h->cols, h->rows, 0, 0));
            if (h->exact)
                hipLaunchKernelGGL(rbs::rbs_bbox_age_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s, h->d_bgp[h->cur], h->rows, h->cols,
                                   (unsigned)h->age_max, reinterpret_cast<int*>(h->d_bgp_box));
            else
            hipLaunchKernelGGL(rbs::rbs_bbox_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s, h->d_bgp[h->cur], h->rows, h->cols,
                               h->background, reinterpret_cast<int*>(h->d_bgp_box));
            RBS_HIP(h, hipGetLastError());
        }
        if (h->stp) {
            P.bgp_src = h->d_bgp[h->cur];
            P.bgp_dst = h->d_bgp[1 - h->cur];
            h->wide = false;   // (the whole-plane machinery knows the scalar background only)
            if (rebase >= 0) h->area_frac = 0.0;   // (judged again from the next sample)
        } else {
            rebase = -1; stp_leaving = false;
        }
    }
    // the whole-plane layout always runs two raster blocks per CU and can afford the larger LDS
    // tile (fewer rectangles split into two work items).  Windowed planes keep ONE tile size
    // whatever the number of blocks of a call: the split decides the order in which a
    // particle's partial sums are added, and the switch between the modes depends on when an
    // asynchronous read-back lands -- results must not.
    const bool f64 = h->precision == RBS_PRECISION_F64;   // (its math tables take a little of the LDS tile)
    // (round 5, C4 slice: the 16 384-px tile with two blocks per CU -- 6 work items per particle instead of 10 -- 7.62 ms against 6.32 with
    // three blocks and the small tile; two blocks with the small tile 8.16: the larger tile is worth 7 %, the third wave per SIMD 25 %)
    P.tile_px = !h->windowed && h->raster_blocks <= 2 * h->cu_count ? (f64 ? rbs::kTilePxBigF64 : rbs::kTilePxBig)
                                                                     : (h->exact ? rbs::kTilePxExact : f64 ? rbs::kTilePxF64 : rbs::kTilePx);
    // split launch: binary64 likelihood on windowed planes (the tile size is the handle's for its whole life, like the
    // monolith's: the split of a rectangle into items decides the order in which a particle's partial sums are added)
    const bool have_borrowed = h->borrowed != nullptr || h->borrowed_f32 != nullptr;
    bool split = (h->split || have_borrowed) && f64 && h->windowed;   // (the same tile, the same items, the same bits either way)
    if (split && !h->exact && !(!h->windowed && h->raster_blocks <= 2 * h->cu_count)) P.tile_px = rbs::kDepthTilePx;   // (= kTilePxF64 already; stamped planes: their own, smaller)
    P.tile_w = 256;
    P.tile_h = std::max(4, P.tile_px / 256 / std::max(1, h->smalln_target / std::max(1, n)));
    if (const char* m = h->tile_override) { P.tile_w = std::max(16, std::atoi(m) / 16 * 16); P.tile_h = std::max(1, std::atoi(m)); }
    const size_t tiles_max = tiles_upper_bound(h->cols, h->rows, P.tile_w, std::min(P.tile_w * P.tile_h, P.tile_px));
    const size_t need = (size_t)n * tiles_max * (h->d_groups[0] ? rbs::kMaxGroups : 1);
    if (need > h->partial_cap) {
        RBS_HIP(h, hipStreamSynchronize(s));
        RBS_HIP(h, hipStreamSynchronize(h->copy_stream));
        (void)hipFree(h->d_partial);
        (void)hipFree(h->d_item_particle);
        h->d_partial = nullptr;
        h->d_item_particle = nullptr;
        h->partial_cap = 0;
        RBS_HIP(h, hipMalloc(&h->d_partial, sizeof(double) * need));
        RBS_HIP(h, hipMalloc(&h->d_item_particle, sizeof(int) * need));
        h->partial_cap = need;
    }
    if (split) {
        // the hand-over buffer: one tile per work item, sized for the WORST case (every rectangle the whole frame: 36 tiles per
        // particle at 640x480) so that no item can ever lie beyond it -- 2.7 GB at 2 000 particles.  Where that exceeds
        // kDepthBudget (20 000 particles) the call is launched as ONE kernel instead (a borrowed frame is staged first).
        // RBS_SPLIT_ITEMS_PER_PARTICLE (tooling / tests): a buffer of n x that many tiles (+ 1 024); an item beyond it is
        // contained (its particle's log-likelihood is NaN), never a wild access.
        constexpr size_t kDepthBudget = (size_t)8 << 30;
        const char* pe = std::getenv("RBS_SPLIT_ITEMS_PER_PARTICLE");
        const size_t want = pe ? std::min(need, (size_t)n * (size_t)std::max(1, std::atoi(pe)) + 1024) : need;
        const size_t bytes = sizeof(unsigned) * (size_t)rbs::kDepthTilePx * want;
        //  the buffer must also FIT: no more than half of what the device has free once the old buffer is returned --
        // the planes of max_particles, other handles and the caller's own allocations live there too -- and an allocation that
        // fails all the same takes the one-kernel launch instead of failing the call
        bool fits = bytes <= kDepthBudget;
        if (fits && want > h->depth_items) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t mine = sizeof(unsigned) * (size_t)rbs::kDepthTilePx * h->depth_items;
                fits = bytes <= (free_b + mine) / 2;
            } else {
                (void)hipGetLastError();
            }
        }
        if (want > h->depth_items && fits) {
            RBS_HIP(h, hipStreamSynchronize(s));
            (void)hipFree(h->d_depth);
            h->d_depth = nullptr;
            h->depth_items = 0;
            if (hipMalloc(&h->d_depth, bytes) == hipSuccess) {
                h->depth_items = want;
            } else {
                (void)hipGetLastError();
                h->d_depth = nullptr;
                fits = false;
            }
        }
        if (want > h->depth_items && !fits) {   // one kernel (a borrowed frame is staged first: nothing of this call is enqueued yet)
            split = false;
            if (have_borrowed) {
                if (int32_t rc = stage_borrowed(h)) return rc;
                P.frame = h->cur_frame;
                P.aux = h->cur_aux;
            }
        }
    }
    if (split) {
        P.depth = h->d_depth;
        P.depth_items = (int)std::min<size_t>(h->depth_items, 0x7fffffff);
        P.ctrb_this = h->d_ctr + 4 + (int)(h->calls & 1);
        P.ctrb_next = h->d_ctr + 4 + (int)((h->calls + 1) & 1);
    }
    int* const d_rects = h->d_rects[h->calls & 1];
    P.rects = d_rects;
    P.groups = h->d_groups[h->calls & 1];
    P.strips = update && h->windowed && !h->copy_walk ? h->d_strips[h->calls & 1] : nullptr;
    P.parents = h->d_parents[h->calls & 1];
    P.item_range = h->d_item_range;
    P.item_particle = h->d_item_particle;
    P.ctr_this = h->d_ctr + 2 * (int)(h->calls & 1);
    P.ctr_next = h->d_ctr + 2 * (int)((h->calls + 1) & 1);
    P.done = h->d_done;
    P.partial = h->d_partial;
#ifdef RBS_PHASE_TIMING
    if (!h->d_phase) { RBS_HIP(h, hipMalloc(&h->d_phase, 256)); RBS_HIP(h, hipMemset(h->d_phase, 0, 256)); }
    P.phase = h->d_phase;
#endif
    if (timed) RBS_HIP(h, hipEventRecord(h->ev_start[tslot], s));
    const dim3 block(rbs::kBlock);
    const dim3 pgrid((unsigned)((n + rbs::kPrepPerBlock - 1) / rbs::kPrepPerBlock));
    // dense planes: prep + scan read only the poses and run ahead of the previous call's copy
    // kernel (which reads the other rectangle buffer); windowed planes: prep reads the windows
    // that copy kernel is still growing, so the join comes first (that copy is short)
    if (h->windowed && h->join_pending >= 0) {
        RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
        h->join_pending = -1;
    }
    // the plane the shared plane is re-based on: GLOBAL slot `rebase`, wherever it lives (a shard / an attached rank reads it from its owner)
    const float* rb_occ = P.occ_src; const int4* rb_win = P.win_src; const int4* rb_reg = h->slab_px ? P.reg_src : (const int4*)nullptr;
    int rb_local = rebase;
    if (rebase >= 0 && P.n_dev > 1) {
        const int owner = rebase / P.shard_cap;
        rb_local = rebase - owner * P.shard_cap;
        rb_occ = P.occ_src_dev[owner]; rb_win = P.win_src_dev[owner]; rb_reg = h->slab_px ? P.reg_src_dev[owner] : (const int4*)nullptr;
    }
    if (P.bgp_src) P.rebase_box = rebase >= 0 ? rb_win + rb_local : rebase == -2 ? h->d_bgp_box : nullptr;
    if (P.bgp_src && update)   // the shared plane's own step (and re-basing): reads the current planes, complete after the join above
        if (int32_t rc = launch_bgp_step(h, P, rb_occ, rb_win, rb_reg, rebase >= 0 ? rb_local : rebase, s)) return rc;
    bool sample_area = false;
    if (h->windowed && update) {
        sample_area = timed && !h->area_pending;
        if (sample_area) RBS_HIP(h, hipMemsetAsync(h->d_area, 0, sizeof(unsigned long long), s));
    }
    P.area_sum = sample_area ? h->d_area : nullptr;
    const bool wide = h->windowed && update && h->wide;
    if (h->lazy_frame && h->lazy_stream == s) {
        constexpr int ptb = 64 * rbs::kPrepPerBlock;
        const int aux_blocks = (h->d_aux || h->lazy_frame != h->d_frame) ? (h->npx + ptb - 1) / ptb : 0;
        hipLaunchKernelGGL(rbs::rbs_frame_prep_kernel, dim3((unsigned)(aux_blocks + (n + rbs::kPrepPerBlock - 1) / rbs::kPrepPerBlock)), dim3(ptb), 0, s, P,
                           d_rects, update ? 1 : 0, h->lazy_frame, h->d_aux, h->d_pbg,
                           h->lazy_frame == h->d_frame ? (float*)nullptr : h->d_frame, aux_blocks);
        h->lazy_frame = nullptr;
        if (h->lazy_slot >= 0) { RBS_HIP(h, hipEventRecord(h->ev_used[h->lazy_slot], s)); h->lazy_slot = -1; }
    } else {
        if (int32_t rc = flush_lazy_frame(h, s)) return rc;
        if (host_deltas) hipLaunchKernelGGL(rbs::rbs_prep_deltas_kernel, pgrid, dim3(64 * rbs::kPrepPerBlock), 0, s, P, d_rects, update ? 1 : 0);
        else hipLaunchKernelGGL(rbs::rbs_prep_kernel, pgrid, dim3(64 * rbs::kPrepPerBlock), 0, s, P, d_rects, update ? 1 : 0);
    }
    RBS_HIP(h, hipGetLastError());
    if (sample_area) {
        RBS_HIP(h, hipMemcpyAsync(h->h_area, h->d_area, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        RBS_HIP(h, hipEventRecord(h->ev_area, s));
        h->area_pending = true;
        h->area_n = (n + 7) / 8;    // (the rectangles kernel adds every 8th particle's region)
    }
    if (update) {
        // fork: the copy kernel runs on the handle's second stream, concurrently with the
        // persistent raster kernel; it needs this call's rectangles only
        RBS_HIP(h, hipEventRecord(h->ev_fork, s));
        // (the windowed copy kernel itself is launched BEHIND the raster kernel's launch, below: its one-wave blocks
        // (32 VGPRs) run beside the persistent raster blocks (3 x 160 VGPRs per SIMD) whichever of the two reaches the
        // device first, and the three runtime calls it takes -- 10-15 us of host time -- are then not in front of the
        // raster kernel when the host is what the device waits for: a tracker frame by frame, the host-pointer calls)
        if (!(h->windowed && !wide)) RBS_HIP(h, hipStreamWaitEvent(h->copy_stream, h->ev_fork, 0));
    }
    // deferred join: the raster kernel reads planes the previous updating call's copy kernel
    // may still be writing; everything before this point overlapped with that copy's tail
    if (h->join_pending >= 0) {
        RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
        h->join_pending = -1;
    }
    // wide windows: two raster blocks per CU leave the streaming copy its registers
    const bool mid = h->windowed && update && !wide && h->area_frac > h->mid_enter;
    if (h->balance && h->windowed && update && !wide && !mid && h->timed_calls > h->balance_seen && h->raster_blocks == 3 * h->cu_count) {
        const int last = (int)((h->timed_calls - 1) % rbs_handle::kRing);
        if (h->ring_update[last] && hipEventQuery(h->ev_raster_stop[last]) == hipSuccess && hipEventQuery(h->ev_copy_stop[last]) == hipSuccess) {
            float r_ms = 0.f, c_ms = 0.f;
            if (hipEventElapsedTime(&r_ms, h->ev_raster_start[last], h->ev_raster_stop[last]) == hipSuccess &&
                hipEventElapsedTime(&c_ms, h->ev_copy_start[last], h->ev_copy_stop[last]) == hipSuccess && r_ms > 0.f) {
                const int step = std::max(1, h->cu_count / 8), lo = 5 * h->cu_count / 2;
                int cur = h->balance_blocks ? h->balance_blocks : h->raster_blocks;
                if (c_ms > 1.04f * r_ms) cur = std::max(lo, cur - step);
                else if (c_ms < 0.92f * r_ms) cur = std::min(h->raster_blocks, cur + step);
                h->balance_blocks = cur;
            } else {
                (void)hipGetLastError();
            }
            h->balance_seen = h->timed_calls;
        }
    }
    const int full_grid = h->balance && h->balance_blocks && h->windowed && update ? h->balance_blocks : h->raster_blocks;
    const dim3 rgrid((unsigned)(wide || mid ? std::min(h->raster_blocks, 2 * h->cu_count) : full_grid));
    // a host frame that is read where it was uploaded: only now does the stream wait for the upload
    // (split launch: the geometry kernel needs no frame -- the wait sits between the two kernels, and a host frame
    // travels while the depth tiles are rasterized)
    if (!split && h->frame_wait >= 0) RBS_HIP(h, hipStreamWaitEvent(s, h->ev_frame[h->frame_wait], 0));
    // the windowed copy kernel (one-wave blocks beside the persistent kernels, on the handle's second stream).  A call that stages a
    // borrowed frame between its two kernels launches it BEFORE the staging: it then runs beside the geometry kernel instead of
    // beside the likelihood kernel the caller is waiting for (plugin step +0.7 %, tests/cpp/host_bench --plugin)
    bool window_copy_launched = false;
    auto launch_window_copy = [&]() -> int32_t {
        RBS_HIP(h, hipStreamWaitEvent(h->copy_stream, h->ev_fork, 0));
        if (timed) RBS_HIP(h, hipEventRecord(h->ev_copy_start[tslot], h->copy_stream));
        const int ny = std::min(n, 32768);
        // one body: only the cells outside its rectangle are enumerated (1); several bodies: the strip list the rectangles kernel
        // made (2: round 6); RBS_COPY_STRIPS=0 builds / RBS_COPY_WALK=1: the walk over the whole region (0)
        const int strips = !RBS_COPY_STRIPS ? 0 : !P.groups ? 1 : (P.strips ? 2 : 0);
        const dim3 wg((unsigned)(strips == 1 ? h->win_chunks_single : h->win_chunks), (unsigned)ny, (unsigned)((n + ny - 1) / ny));
#define RBS_X3(S, B, E)                                                                                                             \
        do {                                                                                                                        \
            if (strips == 2) hipLaunchKernelGGL((rbs::rbs_copy_window_kernel<S, 2, B, E>), wg, dim3(64), 0, h->copy_stream, P);      \
            else if (strips == 1) hipLaunchKernelGGL((rbs::rbs_copy_window_kernel<S, 1, B, E>), wg, dim3(64), 0, h->copy_stream, P); \
            else hipLaunchKernelGGL((rbs::rbs_copy_window_kernel<S, 0, B, E>), wg, dim3(64), 0, h->copy_stream, P);                  \
        } while (0)
        switch ((h->exact ? 4 : 0) | (h->slab_px ? 2 : 0) | (P.bgp_src ? 1 : 0)) {
            case 0: RBS_X3(false, false, false); break;  case 1: RBS_X3(false, true, false); break;
            case 2: RBS_X3(true, false, false); break;   case 3: RBS_X3(true, true, false); break;
            case 4: RBS_X3(false, false, true); break;   case 5: RBS_X3(false, true, true); break;
            case 6: RBS_X3(true, false, true); break;    default: RBS_X3(true, true, true); break;
        }
#undef RBS_X3
        RBS_HIP(h, hipGetLastError());
        if (timed) RBS_HIP(h, hipEventRecord(h->ev_copy_stop[tslot], h->copy_stream));
        RBS_HIP(h, hipEventRecord(h->ev_join[slot], h->copy_stream));
        window_copy_launched = true;
        return RBS_OK;
    };
    if (timed) RBS_HIP(h, hipEventRecord(h->ev_raster_start[tslot], s));
    if (split) {
        const bool two = wide || mid;   // (wide windows: fewer blocks, so that the streaming copy keeps its registers)
        const dim3 dgrid((unsigned)(two ? std::min(h->depth_blocks, 2 * h->cu_count) : h->depth_blocks));
        const dim3 egrid((unsigned)(two ? std::min(h->eval_blocks, 3 * h->cu_count) : h->eval_blocks));
        const size_t dsm = rbs::smem_bytes_depth(P.tile_px, h->many_clusters), esm = rbs::smem_bytes(0, true, false, h->exact);
        if (h->many_clusters) hipLaunchKernelGGL((rbs::rbs_depth_kernel<true>), dgrid, block, dsm, s, P);
        else hipLaunchKernelGGL((rbs::rbs_depth_kernel<false>), dgrid, block, dsm, s, P);
        RBS_HIP(h, hipGetLastError());
        static const bool copy_first = [] { const char* e = std::getenv("RBS_SPLIT_COPY_FIRST"); return e ? std::atoi(e) != 0 : true; }();
        if (copy_first && have_borrowed && update && h->windowed && !wide)
            if (int32_t rc = launch_window_copy()) return rc;
        if (have_borrowed) {   // the host converts and sends the caller's frame while the depth tiles are rasterized
            if (int32_t rc = stage_borrowed(h)) {
                // half a call is enqueued (rectangles + geometry kernel: the work-item counters of this parity are spent, no plane
                // has been written): every later call is refused until rbs_reset re-establishes a known state
                h->poisoned = true;
                h->poison_msg = h->err;
                return rc;
            }
            P.frame = h->cur_frame;
            P.aux = h->cur_aux;
        }
        if (h->frame_wait >= 0) RBS_HIP(h, hipStreamWaitEvent(s, h->ev_frame[h->frame_wait], 0));
        if (h->exact) {
#define RBS_X(U, S, T) hipLaunchKernelGGL((rbs::rbs_eval_kernel<U, S, T, true>), egrid, block, esm, s, P); break
            switch ((update ? 4 : 0) | (h->slab_px ? 2 : 0) | (P.bgp_src ? 1 : 0)) {
                case 0: RBS_X(false, false, false);  case 1: RBS_X(false, false, true);
                case 2: RBS_X(false, true, false);   case 3: RBS_X(false, true, true);
                case 4: RBS_X(true, false, false);   case 5: RBS_X(true, false, true);
                case 6: RBS_X(true, true, false);    default: RBS_X(true, true, true);
            }
#undef RBS_X
        } else
        if (P.bgp_src) {
            switch ((update ? 2 : 0) | (h->slab_px ? 1 : 0)) {
                case 0: hipLaunchKernelGGL((rbs::rbs_eval_kernel<false, false, true>), egrid, block, esm, s, P); break;
                case 1: hipLaunchKernelGGL((rbs::rbs_eval_kernel<false, true, true>), egrid, block, esm, s, P); break;
                case 2: hipLaunchKernelGGL((rbs::rbs_eval_kernel<true, false, true>), egrid, block, esm, s, P); break;
                default: hipLaunchKernelGGL((rbs::rbs_eval_kernel<true, true, true>), egrid, block, esm, s, P); break;
            }
        } else
        switch ((update ? 2 : 0) | (h->slab_px ? 1 : 0)) {
            case 0: hipLaunchKernelGGL((rbs::rbs_eval_kernel<false, false>), egrid, block, esm, s, P); break;
            case 1: hipLaunchKernelGGL((rbs::rbs_eval_kernel<false, true>), egrid, block, esm, s, P); break;
            case 2: hipLaunchKernelGGL((rbs::rbs_eval_kernel<true, false>), egrid, block, esm, s, P); break;
            default: hipLaunchKernelGGL((rbs::rbs_eval_kernel<true, true>), egrid, block, esm, s, P); break;
        }
        RBS_HIP(h, hipGetLastError());
    }
    if (update) {
        if (!split) launch_raster(h, true, rgrid, block, rbs::smem_bytes(P.tile_px, f64, h->many_clusters, h->exact), s, P);
        RBS_HIP(h, hipGetLastError());
        if (timed) RBS_HIP(h, hipEventRecord(h->ev_raster_stop[tslot], s));
        const dim3 cgrid((unsigned)std::min<long>((long)h->copy_blocks, (long)n * P.bands));
        if (timed && (!h->windowed || wide)) RBS_HIP(h, hipEventRecord(h->ev_copy_start[tslot], h->copy_stream));
        if (h->windowed && !wide) {
            if (!window_copy_launched) if (int32_t rc = launch_window_copy()) return rc;
        } else if (wide) {
            const int W4 = P.cols >> 2;
            const int nseg = (W4 + 63) / 64;
            const int ny = std::min(n, 32768);
            const dim3 rg((unsigned)(((P.rows + 1) / 2) * nseg), (unsigned)ny, (unsigned)((n + ny - 1) / ny));
            const int nblk = ((P.rows + 1) / 2) * nseg;
            if (!h->d_wide_flags)
                RBS_HIP(h, hipMalloc(&h->d_wide_flags, (size_t)h->max_particles * nblk));
            P.wide_flags = h->d_wide_flags;
            hipLaunchKernelGGL((rbs::rbs_copy_rows_kernel<2, true>), rg, dim3(64), 0, h->copy_stream, P, nseg);
            hipLaunchKernelGGL(rbs::rbs_wide_window_kernel, dim3((unsigned)n), dim3(64), 0, h->copy_stream, P, nseg, nblk);
        } else if ((P.cols & 3) == 0 && h->copy_rows > 0) {
            const int W4 = P.cols >> 2;
            const int tpb = h->copy_tpb > 0 ? h->copy_tpb : std::min(1024, (W4 + 63) / 64 * 64);
            const int nseg = (W4 + tpb - 1) / tpb;
            const dim3 rblock((unsigned)tpb);
            const int ny = std::min(n, 32768);
            const dim3 rg((unsigned)(((P.rows + h->copy_rows - 1) / h->copy_rows) * nseg), (unsigned)ny,
                          (unsigned)((n + ny - 1) / ny));
            switch (h->copy_rows) {
                case 1: hipLaunchKernelGGL((rbs::rbs_copy_rows_kernel<1, false>), rg, rblock, 0, h->copy_stream, P, nseg); break;
                case 2: hipLaunchKernelGGL((rbs::rbs_copy_rows_kernel<2, false>), rg, rblock, 0, h->copy_stream, P, nseg); break;
                case 4: hipLaunchKernelGGL((rbs::rbs_copy_rows_kernel<4, false>), rg, rblock, 0, h->copy_stream, P, nseg); break;
                default: hipLaunchKernelGGL((rbs::rbs_copy_rows_kernel<8, false>), rg, rblock, 0, h->copy_stream, P, nseg); break;
            }
        } else if ((P.cols & 3) == 0)
            hipLaunchKernelGGL((rbs::rbs_copy_kernel<4>), cgrid, block, 0, h->copy_stream, P);
        else
            hipLaunchKernelGGL((rbs::rbs_copy_kernel<1>), cgrid, block, 0, h->copy_stream, P);
        RBS_HIP(h, hipGetLastError());
        if (!h->windowed || wide) {
            if (timed) RBS_HIP(h, hipEventRecord(h->ev_copy_stop[tslot], h->copy_stream));
            RBS_HIP(h, hipEventRecord(h->ev_join[slot], h->copy_stream));
        }
    } else {
        if (!split) launch_raster(h, false, rgrid, block, rbs::smem_bytes(P.tile_px, f64, h->many_clusters, h->exact), s, P);
        RBS_HIP(h, hipGetLastError());
        if (timed) RBS_HIP(h, hipEventRecord(h->ev_raster_stop[tslot], s));
    }
    if (timed) { RBS_HIP(h, hipEventRecord(h->ev_stop[tslot], s)); h->timed_calls += 1; }
    if (h->cur_slot >= 0 && s != h->stream) {
        // the staging image is released on the handle's stream: keep that stream behind its readers
        RBS_HIP(h, hipEventRecord(h->ev_reader, s));
        RBS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_reader, 0));
    }
    if (stp_leaving) {   // from the next call on the planes are measured against the scalar background again
        h->stp = false;
        h->stp_block_until = h->calls + 4000;
        h->area_frac = 1.0;   // (what the last sample said: the next calls take the whole-plane machinery at once)
        h->wide = !h->slab_px && !h->exact;
    }
    if (update) h->join_pending = slot;   // joined lazily: by the next call, or by drain()
    if (h->group) {
        // other shards read this shard's planes: one event that covers both streams of this call
        if (h->join_pending >= 0) {
            RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
            h->join_pending = -1;
        }
        RBS_HIP(h, hipEventRecord(h->ev_done, s));
    }
    h->calls += 1;
    if (update) {
        h->cur = 1 - h->cur;
        h->update_clock += h->pending_frames;
        h->pending_frames = 0;
        h->background = P.bg_new;
    }
    return RBS_OK;
}

// Windowed planes: make slot's current plane dense in place (background outside its window) and
// mark its window full, on stream s.  Dense planes: nothing to do.
int32_t materialize(rbs_handle* h, int slot, hipStream_t s)
{
    if (!h->windowed) return RBS_OK;
    if (h->exact) return fail(h, RBS_ERR_UNSUPPORTED, "a slot of stamped planes (occlusion_mode REFERENCE) is not a float plane: use rbs_export_plane / rbs_get_occlusion");
    if (h->slab_px) return fail(h, RBS_ERR_UNSUPPORTED, "a slab cannot be made dense in place (state_slab_px)");
    hipLaunchKernelGGL(rbs::rbs_materialize_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s,
                       h->d_occ[h->cur] + (size_t)slot * h->plane_stride, h->d_win[h->cur] + slot, h->rows, h->cols,
                       h->background, h->stp ? (const float*)h->d_bgp[h->cur] : (const float*)nullptr);
    RBS_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_win[h->cur] + slot, 1,
                       make_int4(0, 0, h->cols, h->rows));
    RBS_HIP(h, hipGetLastError());
    return RBS_OK;
}

// Slabs: a slot's plane as a whole plane in `d_full` (device, npx floats), on stream s.
int32_t slab_expand(rbs_handle* h, int slot, float* d_full, hipStream_t s)
{
    hipLaunchKernelGGL(rbs::rbs_expand_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s,
                       h->d_occ[h->cur] + (size_t)slot * h->plane_stride, h->d_reg[h->cur] + slot, h->d_win[h->cur] + slot,
                       h->rows, h->cols, h->background, d_full, h->stp ? (const float*)h->d_bgp[h->cur] : (const float*)nullptr);
    RBS_HIP(h, hipGetLastError());
    return RBS_OK;
}

// Slabs: store a whole plane (device, npx floats) into a slot: its stored region and window become
// the float4-aligned bounding box of the values that differ from the background.  Synchronises
// (the box is needed on the host to check that it fits).
int32_t drain(rbs_handle* h, bool host_sync);
int32_t grow_slabs(rbs_handle* h, int new_slab);
int slab_for(const rbs_handle* h, int need);
int32_t slab_store(rbs_handle* h, int slot, const float* d_full, hipStream_t s)
{
    const int init[4] = {h->cols, h->rows, 0, 0};
    RBS_HIP(h, hipMemcpyAsync(h->d_bbox, init, sizeof(init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(rbs::rbs_bbox_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s, d_full, h->rows,
                       h->cols, h->background, h->d_bbox, h->stp ? (const float*)h->d_bgp[h->cur] : (const float*)nullptr);
    RBS_HIP(h, hipGetLastError());
    int box[4];
    RBS_HIP(h, hipMemcpyAsync(box, h->d_bbox, sizeof(box), hipMemcpyDeviceToHost, s));
    RBS_HIP(h, hipStreamSynchronize(s));
    int4 r = make_int4(box[0], box[1], std::min(box[2], h->cols), box[3]);
    if (r.z <= r.x || r.w <= r.y) r = make_int4(h->cols, h->rows, 0, 0);   // all background
    const long area = r.z > r.x ? (long)(r.z - r.x) * (r.w - r.y) : 0;
    if (area > (long)h->slab_px) {
        // the slabs grow for a plane handed in from outside as they do for a region a call asks for -- on
        // a handle of its own; the shards of a group and attached ranks must keep one size among them
        if (h->group || h->peer_world > 1)
            return fail(h, RBS_ERR_OUT_OF_MEMORY,
                        fmt("a plane whose values differ from the background over %ld px does not fit a slab of %d px (state_slab_px)", area, h->slab_px));
        if (int32_t rc = drain(h, true)) return rc;
        if (int32_t rc = grow_slabs(h, slab_for(h, (int)std::min<long>(area, h->npx)))) return rc;
        if (area > (long)h->slab_px)
            return fail(h, RBS_ERR_OUT_OF_MEMORY, fmt("a plane of %ld px does not fit a slab of %d px (state_slab_px)", area, h->slab_px));
    }
    if (area > 0) {
        hipLaunchKernelGGL(rbs::rbs_pack_kernel, dim3((unsigned)((area + 255) / 256)), dim3(256), 0, s, d_full, r, h->cols,
                           h->d_occ[h->cur] + (size_t)slot * h->plane_stride);
        RBS_HIP(h, hipGetLastError());
    }
    hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_win[h->cur] + slot, 1, r);
    hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_reg[h->cur] + slot, 1, r);
    RBS_HIP(h, hipGetLastError());
    return RBS_OK;
}

int32_t drain(rbs_handle* h, bool host_sync);

// Stamped planes (occlusion_mode REFERENCE): the hooks' two conversions.  A slot -> its plane of EFFECTIVE values as of the
// epoch (device, npx floats); slot < 0: the shared background plane itself.
int32_t exact_expand(rbs_handle* h, int slot, float* d_full, hipStream_t s)
{
    DevParams P = h->base;
    P.ptab = h->d_ptab;
    P.age_max = h->age_max;
    const float* base = slot >= 0 ? h->d_occ[h->cur] + (size_t)slot * h->plane_stride : nullptr;
    hipLaunchKernelGGL(rbs::rbs_expand_exact_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s, base, h->slab_px ? h->slab_px : h->npx,
                       slot >= 0 && h->slab_px ? (const int4*)(h->d_reg[h->cur] + slot) : (const int4*)nullptr,
                       slot >= 0 ? (const int4*)(h->d_win[h->cur] + slot) : (const int4*)nullptr, h->rows, h->cols, h->background,
                       h->stp ? (const float*)h->d_bgp[h->cur] : (const float*)nullptr, P, d_full);
    RBS_HIP(h, hipGetLastError());
    return RBS_OK;
}
// A plane of effective values handed in from outside -> a slot: the values as given with age 0 ("as of now", the rule of the
// oracle's set_occlusion), stored over the bounding box of what differs from the background (the scalar level, or the shared
// plane's effective values).  Synchronises (the box is needed on the host: slabs must fit it).
int32_t exact_store(rbs_handle* h, int slot, const float* d_full, hipStream_t s)
{
    const float* bgref = nullptr;
    if (h->stp) {
        if (int32_t rc = exact_expand(h, -1, h->d_tmp_plane, s)) return rc;
        bgref = h->d_tmp_plane;
    }
    const int init[4] = {h->cols, h->rows, 0, 0};
    RBS_HIP(h, hipMemcpyAsync(h->d_bbox, init, sizeof(init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(rbs::rbs_bbox_kernel, dim3((unsigned)((h->npx + 255) / 256)), dim3(256), 0, s, d_full, h->rows, h->cols, h->background, h->d_bbox, bgref);
    RBS_HIP(h, hipGetLastError());
    int box[4];
    RBS_HIP(h, hipMemcpyAsync(box, h->d_bbox, sizeof(box), hipMemcpyDeviceToHost, s));
    RBS_HIP(h, hipStreamSynchronize(s));
    int4 r = make_int4(box[0], box[1], std::min(box[2], h->cols), box[3]);
    if (r.z <= r.x || r.w <= r.y) r = make_int4(h->cols, h->rows, 0, 0);   // all background
    const long area = r.z > r.x ? (long)(r.z - r.x) * (r.w - r.y) : 0;
    if (h->slab_px && area > (long)h->slab_px) {
        if (h->group || h->peer_world > 1)
            return fail(h, RBS_ERR_OUT_OF_MEMORY,
                        fmt("a plane whose values differ from the background over %ld px does not fit a slab of %d px (state_slab_px)", area, h->slab_px));
        if (int32_t rc = drain(h, true)) return rc;
        if (int32_t rc = grow_slabs(h, slab_for(h, (int)std::min<long>(area, h->npx)))) return rc;
        if (area > (long)h->slab_px)
            return fail(h, RBS_ERR_OUT_OF_MEMORY, fmt("a plane of %ld px does not fit a slab of %d px (state_slab_px)", area, h->slab_px));
    }
    if (area > 0) {
        hipLaunchKernelGGL(rbs::rbs_pack_exact_kernel, dim3((unsigned)((area + 255) / 256)), dim3(256), 0, s, d_full, bgref, h->background, r, h->cols,
                           h->slab_px ? r.x : 0, h->slab_px ? r.y : 0, h->slab_px ? r.z - r.x : h->cols,
                           h->d_occ[h->cur] + (size_t)slot * h->plane_stride, h->slab_px ? h->slab_px : h->npx);
        RBS_HIP(h, hipGetLastError());
    }
    hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_win[h->cur] + slot, 1, r);
    if (h->slab_px) hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_reg[h->cur] + slot, 1, r);
    RBS_HIP(h, hipGetLastError());
    RBS_HIP(h, hipStreamSynchronize(s));
    return RBS_OK;
}

// "A region did not fit its slab" (h_err[0], fetched by the synchronising entry points): the message
// of the calls that cannot repair it themselves (asynchronous calls, the device tracker).
int32_t check_slab_error(rbs_handle* h)
{
    if (!h->slab_px || !h->h_err[0]) return RBS_OK;
    return fail(h, RBS_ERR_OUT_OF_MEMORY,
                fmt("a particle's occlusion window (with its screen rectangle) exceeded the slab of %d px "
                    "(rbs_config.state_slab_px) in a call that had already returned: its log-likelihood is NaN and its "
                    "plane was reset to the background; the slabs have been enlarged for the calls that follow", h->slab_px));
}

// Every slot gets `new_slab` floats (> the current size).  A slot stores its region row-major from
// its first float, so the planes move with one strided copy; the regions' tables stay as they are.
// Only the CURRENT buffer holds state (the other one is written from scratch by the next updating
// call).  The caller has drained the handle.
// Bytes to ask hipMalloc for when `bytes` are needed for a plane buffer.  A buffer whose size has bit 31 set
// (2-4 GiB, 6-8 GiB, ...) cannot be imported by another process on ROCm 7.2: hipIpcOpenMemHandle never
// returns (measured, tools/dbg/ipc_attach_notorch.py: 13 900 x 38 400 px opens at once, 14 500 x 38 400 hangs,
// 30 000 x 38 400 opens, 50 000 x 38 400 hangs).  Such sizes are rounded up to the next multiple of 4 GiB.
size_t occ_alloc_bytes(size_t bytes)
{
    return (bytes & 0x80000000ull) ? ((bytes >> 32) + 1) << 32 : bytes;
}

int32_t grow_slabs(rbs_handle* h, int new_slab)
{
    new_slab = std::min(h->npx, (new_slab + 1023) & ~1023);
    if (new_slab <= h->slab_px) return RBS_OK;
    if (h->peer_world > 1)   // the other ranks address these slabs with a fixed stride through their mappings
        return h->h_err[0] ? fail(h, RBS_ERR_OUT_OF_MEMORY,
                                  fmt("a region of %d px does not fit the slab of %d px, and slabs cannot grow once the handle is "
                                      "attached to other ranks (rbs_ipc_attach): create the handles with a larger state_slab_px",
                                      h->h_err[1], h->slab_px))
                           : RBS_OK;
    float* nb[2] = {nullptr, nullptr};
    const size_t new_stride = h->exact ? exact_stride((size_t)new_slab) : (size_t)new_slab;
    const size_t bytes = sizeof(float) * new_stride * h->max_particles;
    for (int k = 0; k < 2; ++k)
        if (hipMalloc(&nb[k], occ_alloc_bytes(bytes)) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(nb[0]);
            return fail(h, RBS_ERR_OUT_OF_MEMORY, fmt("enlarging the occlusion slabs from %d to %d px per slot needs 2 x %zu bytes more", h->slab_px, new_slab, bytes));
        }
    RBS_HIP(h, hipMemcpy2DAsync(nb[h->cur], sizeof(float) * new_stride, h->d_occ[h->cur], sizeof(float) * h->plane_stride,
                                sizeof(float) * (size_t)h->slab_px, (size_t)h->max_particles, hipMemcpyDeviceToDevice, h->stream));
    if (h->exact)   // ... and the slots' ages, which follow the values of a slot
        RBS_HIP(h, hipMemcpy2DAsync(nb[h->cur] + new_slab, sizeof(float) * new_stride, h->d_occ[h->cur] + h->slab_px, sizeof(float) * h->plane_stride,
                                    sizeof(unsigned short) * (size_t)h->slab_px, (size_t)h->max_particles, hipMemcpyDeviceToDevice, h->stream));
    RBS_HIP(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_occ[0]);
    (void)hipFree(h->d_occ[1]);
    h->d_occ[0] = nb[0];
    h->d_occ[1] = nb[1];
    h->slab_px = new_slab;
    h->plane_stride = new_stride;
    return RBS_OK;
}

// What a call must put back to be run again: an updating call only flips the buffers and moves the
// background level on -- the planes it read are intact.
// What a call that is taken back (a region did not fit its slab) must find as it was.  The shared trail's mode belongs to it:
// the planes the repeated call reads are measured against whatever background -- shared plane or scalar -- the FIRST attempt
// found, and an attempt that left the shared trail (or entered it, or re-based) has already switched the handle's view.
struct CallState { int cur; int pending_frames; float background; bool stp; long stp_last_rebase, stp_rebases, stp_block_until; double area_frac; bool wide; long update_clock; };
CallState save_call_state(const rbs_handle* h)
{
    return {h->cur, h->pending_frames, h->background, h->stp, h->stp_last_rebase, h->stp_rebases, h->stp_block_until, h->area_frac, h->wide, h->update_clock};
}
void restore_call_state(rbs_handle* h, const CallState& c)
{
    h->cur = c.cur; h->pending_frames = c.pending_frames; h->background = c.background; h->update_clock = c.update_clock;
    h->stp = c.stp; h->stp_last_rebase = c.stp_last_rebase; h->stp_rebases = c.stp_rebases; h->stp_block_until = c.stp_block_until;
    h->area_frac = c.area_frac; h->wide = c.wide;
}

// Slab size that holds a region of `need` px with room to move.
// (1.5 x: a slab enlarged for `need` is at most two thirds full, below the three quarters that trigger the next
// enlargement -- with 1.25 x the trigger stayed true after growing and every synchronising call drained the handle: ADVICE r3)
int slab_for(const rbs_handle* h, int need) { return (int)std::min<long>(h->npx, (long)need + need / 2 + 1024); }

// After a synchronising call on an idle handle: enlarge the slabs BEFORE a region fills one (the
// regions move a few pixels per frame; three quarters full is the trigger).  Clears the flag.
int32_t slab_housekeeping(rbs_handle* h)
{
    if (!h->slab_px) return RBS_OK;
    const bool overflowed = h->h_err[0] != 0;
    // drain (a host synchronisation of both streams) only when the slabs will really be reallocated: h_err[1] is a
    // sticky maximum, and a slab that is already the whole frame, or large enough for it, has nothing to gain
    const int target = (int)std::min<long>(h->npx, ((long)slab_for(h, h->h_err[1]) + 1023) & ~1023L);
    const bool grow = (long)h->h_err[1] * 4 > (long)h->slab_px * 3 && target > h->slab_px && h->peer_world <= 1;
    if (overflowed || grow) {
        if (int32_t rc = drain(h, true)) return rc;
        if (int32_t rc = grow_slabs(h, slab_for(h, h->h_err[1]))) return rc;
    }
    if (overflowed) {
        RBS_HIP(h, hipMemsetAsync(h->d_err, 0, sizeof(int), h->stream));   // (the flag; the maximum stays)
        h->h_err[0] = 0;
    }
    return RBS_OK;
}

// Make stream `s` (and the host, if sync) see the planes of the last updating call complete.
int32_t drain(rbs_handle* h, bool host_sync)
{
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    if (h->join_pending >= 0) {
        RBS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join[h->join_pending], 0));
        if (host_sync) RBS_HIP(h, hipEventSynchronize(h->ev_join[h->join_pending]));
        h->join_pending = -1;
    }
    if (host_sync) {
        RBS_HIP(h, hipStreamSynchronize(h->stream));
        RBS_HIP(h, hipStreamSynchronize(h->copy_stream));
    }
    return RBS_OK;
}

void release(rbs_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    for (auto& rank_bufs : h->peer_mapped)
        for (void*& p : rank_bufs)
            if (p) { (void)hipIpcCloseMemHandle(p); p = nullptr; }
    (void)hipFree(h->d_soup);
    (void)hipFree(h->d_frame);
    (void)hipFree(h->d_aux);
    (void)hipFree(h->d_aux_slot[0]);
    (void)hipFree(h->d_aux_slot[1]);
    (void)hipFree(h->d_pbg);
    (void)hipFree(h->d_occ[0]);
    (void)hipFree(h->d_occ[1]);
    (void)hipFree(h->d_poses);
    (void)hipFree(h->d_indices);
    (void)hipFree(h->d_out);
    (void)hipFree(h->d_rects[0]);
    (void)hipFree(h->d_rects[1]);
    (void)hipFree(h->d_groups[0]);
    (void)hipFree(h->d_groups[1]);
    (void)hipFree(h->d_strips[0]);
    (void)hipFree(h->d_strips[1]);
    (void)hipFree(h->d_parents[0]);
    (void)hipFree(h->d_parents[1]);
    (void)hipFree(h->d_win[0]);
    (void)hipFree(h->d_win[1]);
    (void)hipFree(h->d_win_used);
    (void)hipFree(h->d_reg[0]);
    (void)hipFree(h->d_reg[1]);
    (void)hipFree(h->d_err);
    (void)hipFree(h->d_bbox);
    (void)hipFree(h->d_peer_scratch);
    if (h->h_err) (void)hipHostFree(h->h_err);
    (void)hipFree(h->d_item_range);
    (void)hipFree(h->d_item_particle);
    (void)hipFree(h->d_ctr);
    (void)hipFree(h->d_depth);
    (void)hipFree(h->d_bgp[0]);
    (void)hipFree(h->d_bgp[1]);
    (void)hipFree(h->d_bgp_box);
    (void)hipFree(h->d_ptab);
    (void)hipFree(h->d_tmp_plane);
    (void)hipFree(h->d_area);
    (void)hipFree(h->d_wide_flags);
    if (h->h_area) (void)hipHostFree(h->h_area);
    if (h->ev_area) (void)hipEventDestroy(h->ev_area);
    (void)hipFree(h->d_done);
    (void)hipFree(h->d_partial);
    (void)hipFree(h->d_cluster_sphere);
    (void)hipFree(h->d_cluster_cone);
    (void)hipFree(h->d_cluster_vtx);
    (void)hipFree(h->d_cluster_nv);
    (void)hipFree(h->d_tri_local);
    (void)hipFree(h->d_tri_plane);
    (void)hipFree(h->d_vtx);
    (void)hipFree(h->d_render);
    if (h->up_stream) (void)hipStreamSynchronize(h->up_stream);
    for (int k = 0; k < 2; ++k) {
        if (h->h_frames[k]) (void)hipHostFree(h->h_frames[k]);
        if (h->ev_frame[k]) (void)hipEventDestroy(h->ev_frame[k]);
        if (h->ev_used[k]) (void)hipEventDestroy(h->ev_used[k]);
        if (k == 0 && h->ev_reader) (void)hipEventDestroy(h->ev_reader);
        (void)hipFree(h->d_fin[k]);
    }
    if (h->up_stream) (void)hipStreamDestroy(h->up_stream);
    if (h->h_in) (void)hipHostFree(h->h_in);
    if (h->h_out) (void)hipHostFree(h->h_out);
    (void)hipFree(h->d_in);
    if (h->ev_out) (void)hipEventDestroy(h->ev_out);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int i = 0; i < rbs_handle::kRing; ++i) {
        if (h->ev_start[i]) (void)hipEventDestroy(h->ev_start[i]);
        if (h->ev_stop[i]) (void)hipEventDestroy(h->ev_stop[i]);
        if (h->ev_copy_start[i]) (void)hipEventDestroy(h->ev_copy_start[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
        if (h->ev_raster_start[i]) (void)hipEventDestroy(h->ev_raster_start[i]);
        if (h->ev_raster_stop[i]) (void)hipEventDestroy(h->ev_raster_stop[i]);
        if (h->ev_copy_stop[i]) (void)hipEventDestroy(h->ev_copy_stop[i]);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// The staging image that served as the observation stops doing so (a new frame is coming): it may
// be overwritten once everything launched so far has run.
int32_t release_frame_slot(rbs_handle* h)
{
    if (h->cur_slot >= 0) {
        RBS_HIP(h, hipEventRecord(h->ev_used[h->cur_slot], h->stream));
        h->cur_slot = -1;
    }
    h->cur_frame = h->d_frame;
    h->cur_aux = h->d_aux;
    h->frame_wait = -1;
    return RBS_OK;
}

// H2D of a frame staged in pinned host memory (`src`, normally h->h_frame) on the upload stream into
// the device staging buffer of the current slot; the launch stream waits for it, and the ingest
// kernel (copy into d_frame + per-frame model terms) rides on the next loglikes launch
// (flush_lazy_frame otherwise).
// `pageable`: the caller's own frame, still to be staged into `src` (= h->h_frame) -- staged and sent
// in h->upload_chunks pieces, so that the copy engine carries one piece while the host copies the next
// (a 640x480 frame: 41 us of memcpy + 29 us of transfer one after the other, ~20 us less interleaved).
// dbot hands the image over as a vector of DOUBLES (R:source/dbot_ros/util/ros_interface.h:152-168): double -> float while
// staging.  The scalar loop converts 2 values per instruction on a baseline x86-64 build; AVX2 (every host this library
// meets; checked at run time) does 4 per instruction and halves the time of a 640x480 frame.
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void convert_f64_f32_avx2(float* __restrict__ dst, const double* __restrict__ src, size_t n)
{
    size_t p = 0;
    for (; p + 8 <= n; p += 8) {
        const __m128 a = _mm256_cvtpd_ps(_mm256_loadu_pd(src + p)), b = _mm256_cvtpd_ps(_mm256_loadu_pd(src + p + 4));
        _mm256_storeu_ps(dst + p, _mm256_set_m128(b, a));
    }
    for (; p < n; ++p) dst[p] = (float)src[p];
}
// AVX-512 hosts: 8 values per instruction and streaming stores (the staging image is read next by the copy engine, not by
// this core: no line of it needs to be fetched to be overwritten) -- 42 us against 50 per 640x480 frame out of L3 (EPYC 9575F,
// tools/dbg/convert_bench.cpp); the plugin path gains ~1 %
__attribute__((target("avx512f"))) static void convert_f64_f32_avx512(float* __restrict__ dst, const double* __restrict__ src, size_t n)
{
    size_t p = 0;
    for (; p + 16 <= n; p += 16) {
        const __m256 a = _mm512_cvtpd_ps(_mm512_loadu_pd(src + p)), b = _mm512_cvtpd_ps(_mm512_loadu_pd(src + p + 8));
        _mm512_stream_ps(dst + p, _mm512_castpd_ps(_mm512_insertf64x4(_mm512_castpd256_pd512(_mm256_castps_pd(a)), _mm256_castps_pd(b), 1)));
    }
    _mm_sfence();
    for (; p < n; ++p) dst[p] = (float)src[p];
}
#endif
static void convert_f64_f32(float* __restrict__ dst, const double* __restrict__ src, size_t n)
{
#if defined(__x86_64__)
    static const bool avx512 = __builtin_cpu_supports("avx512f") && !std::getenv("RBS_NO_AVX512");
    if (avx512 && (reinterpret_cast<uintptr_t>(dst) & 63) == 0) return convert_f64_f32_avx512(dst, src, n);
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return convert_f64_f32_avx2(dst, src, n);
#endif
    for (size_t p = 0; p < n; ++p) dst[p] = (float)src[p];
}

int32_t upload_frame(rbs_handle* h, const float* src, const float* pageable = nullptr, const double* pageable_f64 = nullptr)
{
    const int k = h->frame_slot;
    const size_t n = (size_t)h->npx;
    h->prefetched_slot = -1;   // (a frame uploaded ahead of its turn that another frame overtakes is abandoned)
    h->borrowed = nullptr;     // (... and so is a borrowed frame nobody evaluated)
    h->borrowed_f32 = nullptr;
    // (the two staging images alternate, so k is never the image that serves as the observation --
    // should it be, its readers are recorded first: the wait below must not be on a stale event)
    if (k == h->cur_slot)
        if (int32_t rc = release_frame_slot(h)) return rc;
    RBS_HIP(h, hipStreamWaitEvent(h->up_stream, h->ev_used[k], 0));   // d_fin[k]: read by the ingest kernel two frames ago
    // A SMALL frame (the reference's default operating point is 80x60: 19 KB) is not worth a transfer: a copy-engine
    // job costs ~11 us before it moves a byte and ~9 more until a kernel behind it starts.  The kernel that computes
    // the frame's model terms reads it where the host staged it (this handle's pinned buffer, over PCIe) and leaves
    // the observation in d_fin[k] as it goes (precision F32: that copy is all it does).
    const bool pull = h->frame_pull_bytes > 0 && n * sizeof(float) <= h->frame_pull_bytes && h->h_frames_dev[k] && !h->frame_ingest &&
                      (pageable || pageable_f64 || src == h->h_frames[k]);
    if (pageable || pageable_f64) {
        float* stage = h->h_frames[k];
        // (a frame of doubles takes the host ~4x as long to stage as a frame of floats: twice the pieces)
        const size_t want = (size_t)h->upload_chunks * (pageable_f64 && h->upload_chunks > 1 ? 2 : 1);
        const int pieces = h->quiet ? (int)std::max<size_t>(1, std::min<size_t>(want, n / 16384)) : 1;
        const size_t per = ((n + pieces - 1) / pieces + 1023) / 1024 * 1024;
        for (size_t off = 0; off < n; off += per) {
            const size_t len = std::min(per, n - off);
            if (pageable) std::memcpy(stage + off, pageable + off, len * sizeof(float));
            else convert_f64_f32(stage + off, pageable_f64 + off, len);   // dbot hands a vector of doubles
            if (!pull) RBS_HIP(h, hipMemcpyAsync(h->d_fin[k] + off, stage + off, len * sizeof(float), hipMemcpyHostToDevice, h->up_stream));
        }
    } else if (!pull) {
        RBS_HIP(h, hipMemcpyAsync(h->d_fin[k], src, n * sizeof(float), hipMemcpyHostToDevice, h->up_stream));
    }
    if (pull) {
        hipLaunchKernelGGL(rbs::frame_aux_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->up_stream,
                           h->h_frames_dev[k], h->d_aux ? h->d_aux_slot[k] : (double*)nullptr, h->d_pbg, h->npx, h->base.tw,
                           h->base.ms, h->base.sf, h->base.lambda, h->d_fin[k]);
        RBS_HIP(h, hipGetLastError());
    } else if (h->d_aux && !h->frame_ingest) {
        // precision F64: the frame's per-pixel model terms are computed right behind the copy, on the
        // upload stream, into the slot's own table -- the launch stream does not wait for the frame
        // until the raster kernel needs it, so a caller's transition and the rectangles kernel run
        // while the frame is still travelling (a tracker frame: 35 us of 400)
        hipLaunchKernelGGL(rbs::frame_aux_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->up_stream,
                           h->d_fin[k], h->d_aux_slot[k], h->d_pbg, h->npx, h->base.tw, h->base.ms, h->base.sf,
                           h->base.lambda, (float*)nullptr);
        RBS_HIP(h, hipGetLastError());
    }
    RBS_HIP(h, hipEventRecord(h->ev_frame[k], h->up_stream));
    if (int32_t rc = release_frame_slot(h)) return rc;
    if (!h->frame_ingest) {   // (callers have flushed any pending frame)
        // the staging image IS the observation (and, F64, its slot's table the model terms) until the
        // next frame replaces it
        h->cur_frame = h->d_fin[k];
        h->cur_aux = h->d_aux ? h->d_aux_slot[k] : nullptr;
        h->cur_slot = k;
        h->frame_wait = k;
        return RBS_OK;
    }
    RBS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_frame[k], 0));
    h->lazy_frame = h->d_fin[k];
    h->lazy_stream = h->stream;
    h->lazy_slot = k;
    return RBS_OK;
}

// The pinned frame buffer the next host frame is staged in: the other one of the two, once the
// upload that last used it has finished (two frames ago: in practice never a wait).
int32_t next_frame_staging(rbs_handle* h)
{
    h->frame_slot ^= 1;
    h->h_frame = h->h_frames[h->frame_slot];
    RBS_HIP(h, hipEventSynchronize(h->ev_frame[h->frame_slot]));
    return RBS_OK;
}

// rbs_set_observation_borrowed's frame becomes the observation now: staged (double -> float) into pinned memory and sent.
int32_t stage_borrowed(rbs_handle* h)
{
    if (!h->borrowed && !h->borrowed_f32) return RBS_OK;
    const double* d = h->borrowed;
    const float* f = h->borrowed_f32;
    h->borrowed = nullptr;
    h->borrowed_f32 = nullptr;
    if (int32_t rc = next_frame_staging(h)) return rc;
    return d ? upload_frame(h, h->h_frame, nullptr, d) : upload_frame(h, h->h_frame, f);
}

// A borrowed frame is the caller's memory until the likelihood call behind it RETURNS -- however it returns.  A call that is
// refused or fails before its kernels staged the frame must not leave the pointer behind for a later call to read: on the way
// out the frame is copied after all (it becomes the observation, as rbs_set_observation would have made it), or dropped where
// the handle takes no more work.
struct BorrowedFrameGuard {
    rbs_handle* h;
    ~BorrowedFrameGuard()
    {
        if (!h->borrowed && !h->borrowed_f32) return;
        if (h->poisoned || hipSetDevice(h->device) != hipSuccess) { h->borrowed = nullptr; h->borrowed_f32 = nullptr; return; }
        const std::string keep = h->err;      // (the message of the failure that brought us here stays rbs_last_error's)
        const int32_t rc = stage_borrowed(h);
        h->borrowed = nullptr;
        h->borrowed_f32 = nullptr;
        if (rc == RBS_OK) h->err = keep;
    }
};

// The NEXT frame, uploaded while the current one is still the observation (rbs_loglikes_prefetch: called between
// the launch of a call's kernels and the wait for its results, so the host's staging copy and the transfer pass
// behind the raster kernel): staged into the other pinned image, sent on the upload stream, its model terms
// (precision F64) computed behind it -- everything upload_frame does except making it the observation.
// What rbs_loglikes_prefetch needs, checked BEFORE the call's kernels are enqueued (a prefetch that failed
// behind an updating call left the caller with a failed call whose planes had already flipped).
int32_t prefetch_check(rbs_handle* h)
{
    if (h->frame_ingest)   // (frames ingested on the launch stream: RBS_FRAME_INGEST tooling mode)
        return fail(h, RBS_ERR_UNSUPPORTED, "loglikes_prefetch: not with RBS_FRAME_INGEST");
    if (h->prefetched_slot >= 0)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "loglikes_prefetch: the frame uploaded ahead by the previous rbs_loglikes_prefetch has not been "
                                                 "installed yet (rbs_set_observation_prefetched)");
    if ((h->frame_slot ^ 1) == h->cur_slot)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "loglikes_prefetch: the staging image the next frame would use is the current observation "
                                                 "(the staging images did not alternate)");
    return RBS_OK;
}

int32_t prefetch_frame(rbs_handle* h, const float* depth)
{
    if (int32_t rc = prefetch_check(h)) return rc;   // (before the staging image is switched: a refused call changes nothing)
    if (int32_t rc = next_frame_staging(h)) return rc;
    const int k = h->frame_slot;
    const size_t n = (size_t)h->npx;
    RBS_HIP(h, hipStreamWaitEvent(h->up_stream, h->ev_used[k], 0));   // d_fin[k]: its readers were enqueued two frames ago
    std::memcpy(h->h_frames[k], depth, n * sizeof(float));
    RBS_HIP(h, hipMemcpyAsync(h->d_fin[k], h->h_frames[k], n * sizeof(float), hipMemcpyHostToDevice, h->up_stream));
    if (h->d_aux) {
        hipLaunchKernelGGL(rbs::frame_aux_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->up_stream,
                           h->d_fin[k], h->d_aux_slot[k], h->d_pbg, h->npx, h->base.tw, h->base.ms, h->base.sf,
                           h->base.lambda, (float*)nullptr);
        RBS_HIP(h, hipGetLastError());
    }
    RBS_HIP(h, hipEventRecord(h->ev_frame[k], h->up_stream));
    h->prefetched_slot = k;
    return RBS_OK;
}

int32_t create_impl(const rbs_config* cfg, rbs_handle* h)
{
    if (cfg->abi_version != RBS_ABI_VERSION)
        return fail(h, RBS_ERR_INVALID_ARGUMENT,
                    fmt("abi_version %d, library is %d", cfg->abi_version, RBS_ABI_VERSION));
    if (cfg->rows <= 0 || cfg->cols <= 0 || cfg->cols > 8192 || cfg->rows > 8192)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("bad resolution %dx%d", cfg->cols, cfg->rows));
    if (cfg->max_particles <= 0)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "max_particles must be positive");
    if (cfg->n_objects <= 0 || cfg->n_objects > rbs::kMaxBodies)
        return fail(h, RBS_ERR_INVALID_ARGUMENT,
                    fmt("n_objects %d outside 1..%d", cfg->n_objects, rbs::kMaxBodies));
    if (!cfg->vertices || !cfg->vertex_counts || !cfg->triangles || !cfg->triangle_counts)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "null mesh pointer");
    const double* K = cfg->K;
    if (!(K[0] > 0.0) || !(K[4] > 0.0) || K[1] != 0.0 || K[3] != 0.0 || K[6] != 0.0 ||
        K[7] != 0.0 || K[8] != 1.0)
        return fail(h, RBS_ERR_UNSUPPORTED,
                    "camera matrix must be [fx 0 cx; 0 fy cy; 0 0 1] with fx, fy > 0");
    const double c = cfg->p_occluded_occluded - cfg->p_occluded_visible;
    if (!(c > 0.0) || !(c < 1.0))
        return fail(h, RBS_ERR_INVALID_ARGUMENT,
                    "need 0 < p_occluded_occluded - p_occluded_visible < 1");
    if (!(cfg->delta_time > 0.0)) return fail(h, RBS_ERR_INVALID_ARGUMENT, "delta_time must be > 0");
    // the pixel model: a density needs a positive noise level and a mixture weight in [0, 1) (with
    // these the per-pixel arithmetic never meets a NaN of its own making: sigma > 0 for every finite o)
    if (!(cfg->model_sigma > 0.0) || !(cfg->sigma_factor >= 0.0) || !std::isfinite(cfg->model_sigma) || !std::isfinite(cfg->sigma_factor))
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "need kinect.model_sigma > 0 and kinect.sigma_factor >= 0");
    // tail_weight: the reference hands the ROS parameter through unchecked, 0 included, so 0 <= tw < 1 is
    // accepted.  The library's erfc / exp are accurate in ABSOLUTE terms (1e-16), which is all the mixture can see next to
    // tw / max_depth -- with tw = 0 the far tails would be priced by that floor instead of the decaying value --
    // so a weight below kTailWeightFloor is EVALUATED as kTailWeightFloor (tw / max_depth = 1.7e-10: the tables' 1e-16 is
    // then 6e-7 relative in the deepest tail, far inside the 1e-5 bar; a log-likelihood that the exact tw = 0 model
    // drives to -inf becomes a large finite negative number instead).  Stated in the header and INTEGRATION.md.
    if (!(cfg->tail_weight >= 0.0) || !(cfg->tail_weight < 1.0))
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "need 0 <= kinect.tail_weight < 1");
    if (!(cfg->initial_occlusion_prob >= 0.0) || !(cfg->initial_occlusion_prob <= 1.0))
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "need 0 <= occlusion.initial_occlusion_prob <= 1");

    (void)hipGetLastError();   // a stale error left by somebody else on this thread is not ours
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(h, RBS_ERR_NO_DEVICE,
                    "no HIP device visible: librbsensor_mi355x has no CPU path");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(h, RBS_ERR_NO_DEVICE, fmt("device_id %d of %d", cfg->device_id, ndev));
    h->device = cfg->device_id;
    RBS_HIP(h, hipSetDevice(h->device));

    h->rows = cfg->rows;
    h->cols = cfg->cols;
    h->npx = cfg->rows * cfg->cols;
    h->max_particles = cfg->max_particles;
    h->n_bodies = cfg->n_objects;
    h->p_ov = cfg->p_occluded_visible;
    h->p_oo = cfg->p_occluded_occluded;
    h->init_occ = cfg->initial_occlusion_prob;
    h->delta_time = cfg->delta_time;
    switch (cfg->likelihood_precision) {
        case RBS_PRECISION_DEFAULT:
            // the caller leaves it open: the library's default (F64, the reference CPU model's arithmetic),
            // unless the environment names one (tooling: A/B runs of an unchanged caller); a precision
            // the caller names is never overridden
            h->precision = RBS_PRECISION_LIBRARY_DEFAULT;
            if (const char* m = std::getenv("RBS_PRECISION")) {
                if (!std::strcmp(m, "f64")) h->precision = RBS_PRECISION_F64;
                else if (!std::strcmp(m, "f32")) h->precision = RBS_PRECISION_F32;
            }
            break;
        case RBS_PRECISION_F64: case RBS_PRECISION_F32: h->precision = cfg->likelihood_precision; break;
        default: return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("likelihood_precision %d", cfg->likelihood_precision));
    }
    if (cfg->state_layout < RBS_STATE_DEFAULT || cfg->state_layout > RBS_STATE_DENSE)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("state_layout %d", cfg->state_layout));
    if (cfg->state_slab_px < RBS_SLAB_WHOLE_PLANES)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("state_slab_px %d", cfg->state_slab_px));
    if (cfg->occlusion_mode < RBS_OCC_DEFAULT || cfg->occlusion_mode > RBS_OCC_REFERENCE)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("occlusion_mode %d", cfg->occlusion_mode));

    DevParams& B = h->base;
    B.rows = h->rows; B.cols = h->cols; B.npx = h->npx;
    B.n_bodies = h->n_bodies;
    B.fx = K[0]; B.fy = K[4]; B.cx = K[2]; B.cy = K[5];
    B.tw = std::max(cfg->tail_weight, RBS_TAIL_WEIGHT_FLOOR); B.ms = cfg->model_sigma; B.sf = cfg->sigma_factor;
    B.lambda = -std::log(0.5) / rbs::kHalfLifeDepth;
    B.cv0 = (1.0 - B.tw) / std::sqrt(M_PI);
    B.bands = copy_bands_for(h->rows, h->cols);
    B.band_rows = (h->rows + B.bands - 1) / B.bands;

    // triangle soup (SoA) + bounding spheres.  Per body the triangles are ordered along a
    // bisection of positions and normals (below) and padded to a multiple of 64 with NaN triangles, so
    // that every aligned run of 64 is a compact surface patch ("cluster") one wave rasterizes.
    long n_tri = 0;
    B.tri_begin[0] = 0;
    for (int b = 0; b < h->n_bodies; ++b) {
        if (cfg->vertex_counts[b] <= 0 || cfg->triangle_counts[b] < 0)
            return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("object %d: bad mesh counts", b));
        n_tri += ((long)cfg->triangle_counts[b] + 63) / 64 * 64;
        B.tri_begin[b + 1] = (int)n_tri;
    }
    for (int b = h->n_bodies; b < rbs::kMaxBodies; ++b) B.tri_begin[b + 1] = (int)n_tri;
    for (int b = 0; b < rbs::kMaxBodies; ++b) B.tri_end[b] = B.tri_begin[b];
    for (int b = 0; b < h->n_bodies; ++b) {
        const int clusters = (B.tri_begin[b + 1] - B.tri_begin[b]) >> 6;
        h->many_clusters |= clusters > 64 * (rbs::kBlock / 64);   // (more steps of 64 clusters than the block has waves)
    }
    if (const char* e = std::getenv("RBS_SHARED_CULL")) h->many_clusters = std::atoi(e) != 0;   // (A/B: the other set of kernels)
    if (n_tri > (1L << 30)) return fail(h, RBS_ERR_INVALID_ARGUMENT, "too many triangles");
    B.n_tri = (int)n_tri;
    const size_t n_alloc = (size_t)(n_tri > 0 ? n_tri : 64);
    std::vector<double> soup((size_t)9 * n_alloc, std::nan(""));
    std::vector<float> cluster_sphere(4 * (n_alloc / 64), 0.f);
    std::vector<float> cluster_cone(4 * (n_alloc / 64), -2.f);   // min cos -2: never culled
    std::vector<float> tri_plane(4 * n_alloc, std::nanf(""));    // NaN: never pre-culled
    std::vector<double> cluster_vtx((size_t)192 * (n_alloc / 64), 0.0);
    std::vector<int> cluster_nv(n_alloc / 64, 0);
    std::vector<unsigned> tri_local(n_alloc, 0xffffffffu);
    const bool allow_cull = !(std::getenv("RBS_NO_CULL") && std::atoi(std::getenv("RBS_NO_CULL")));
    size_t voff = 0, toff = 0;
    for (int b = 0; b < h->n_bodies; ++b) {
        const int nv = cfg->vertex_counts[b], nt = cfg->triangle_counts[b];
        const double* V = cfg->vertices + 3 * voff;
        const int32_t* T = cfg->triangles + 3 * toff;
        double lo[3] = {V[0], V[1], V[2]}, hi[3] = {V[0], V[1], V[2]};
        for (int i = 0; i < nv; ++i)
            for (int c3 = 0; c3 < 3; ++c3) {
                const double x = V[3 * i + c3];
                if (!std::isfinite(x))
                    return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("object %d: non-finite vertex", b));
                lo[c3] = std::fmin(lo[c3], x);
                hi[c3] = std::fmax(hi[c3], x);
            }
        double ctr[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
        double r2 = 0.0;
        for (int i = 0; i < nv; ++i) {
            const double dx = V[3 * i] - ctr[0], dy = V[3 * i + 1] - ctr[1], dz = V[3 * i + 2] - ctr[2];
            r2 = std::fmax(r2, dx * dx + dy * dy + dz * dz);
        }
        B.sphere[b][0] = ctr[0]; B.sphere[b][1] = ctr[1]; B.sphere[b][2] = ctr[2];
        B.sphere[b][3] = std::sqrt(r2) * (1.0 + 1e-9) + 1e-12;
        for (int t = 0; t < nt; ++t)
            for (int k = 0; k < 3; ++k)
                if (T[3 * t + k] < 0 || T[3 * t + k] >= nv)
                    return fail(h, RBS_ERR_INVALID_ARGUMENT,
                                fmt("object %d triangle %d: vertex index %d out of range", b, t, T[3 * t + k]));
        // Back-face culling is exact only for a closed, consistently oriented surface: after
        // welding vertices by position and dropping degenerate triangles every directed edge
        // must occur exactly once, and its reverse exactly once.  The sign of the signed volume
        // tells which winding is outward.
        B.body_cull[b] = 0;
        if (allow_cull && nt >= 4) {
            std::map<std::array<double, 3>, int> weld;
            std::vector<int> wid(nv);
            for (int i = 0; i < nv; ++i) {
                const std::array<double, 3> key = {V[3 * i] + 0.0, V[3 * i + 1] + 0.0, V[3 * i + 2] + 0.0};  // -0 -> +0
                wid[i] = weld.emplace(key, (int)weld.size()).first->second;
            }
            std::map<std::pair<int, int>, int> edges;
            double vol6 = 0.0, avol6 = 0.0;
            bool ok = true;
            // connected components (shells) over the welded vertices: every shell must be wound the
            // same way -- an inside-out shell beside an outward one would pass the edge test and
            // still show the camera its "back" faces first
            std::vector<int> comp(weld.size());
            for (size_t i = 0; i < comp.size(); ++i) comp[i] = (int)i;
            auto find = [&](int x) { while (comp[x] != x) { comp[x] = comp[comp[x]]; x = comp[x]; } return x; };
            std::vector<double> tri_vol(nt, 0.0);
            for (int t = 0; t < nt && ok; ++t) {
                const int a = wid[T[3 * t]], bb = wid[T[3 * t + 1]], c3 = wid[T[3 * t + 2]];
                if (a == bb || bb == c3 || a == c3) continue;
                for (const auto& e : {std::make_pair(a, bb), std::make_pair(bb, c3), std::make_pair(c3, a)})
                    if (++edges[e] > 1) ok = false;
                const double* p0 = V + 3 * T[3 * t]; const double* p1 = V + 3 * T[3 * t + 1]; const double* p2 = V + 3 * T[3 * t + 2];
                const double d = (p0[0] - ctr[0]) * ((p1[1] - ctr[1]) * (p2[2] - ctr[2]) - (p1[2] - ctr[2]) * (p2[1] - ctr[1])) -
                                 (p0[1] - ctr[1]) * ((p1[0] - ctr[0]) * (p2[2] - ctr[2]) - (p1[2] - ctr[2]) * (p2[0] - ctr[0])) +
                                 (p0[2] - ctr[2]) * ((p1[0] - ctr[0]) * (p2[1] - ctr[1]) - (p1[1] - ctr[1]) * (p2[0] - ctr[0]));
                vol6 += d;
                avol6 += std::fabs(d);
                tri_vol[t] = d;
                comp[find(a)] = find(bb);
                comp[find(bb)] = find(c3);
            }
            if (ok) {
                std::map<int, std::pair<double, double>> shell;   // root -> (signed, absolute) volume * 6
                for (int t = 0; t < nt; ++t) {
                    if (tri_vol[t] == 0.0) continue;
                    auto& sv = shell[find(wid[T[3 * t]])];
                    sv.first += tri_vol[t];
                    sv.second += std::fabs(tri_vol[t]);
                }
                for (const auto& sv : shell)
                    if (!(std::fabs(sv.second.first) > 1e-6 * sv.second.second) || (sv.second.first > 0.0) != (vol6 > 0.0)) ok = false;
            }
            if (ok)
                for (const auto& e : edges)
                    if (edges.find({e.first.second, e.first.first}) == edges.end()) { ok = false; break; }
            if (ok && !edges.empty() && std::fabs(vol6) > 1e-6 * avol6) B.body_cull[b] = vol6 > 0.0 ? 1 : -1;
        }
        // Cluster order: recursive median bisection of the triangles in (centroid / extent,
        // 0.5 * unit normal) space, always along the widest of the six axes, left halves a whole
        // number of 64-triangle clusters.  Every aligned run of 64 is then a compact surface
        // patch with a narrow normal cone (a plain Morton order of the centroids mixes the two
        // sides of thin parts and gives cones too wide to cull by).
        std::vector<std::pair<uint32_t, int>> order(nt);
        {
            const double ext = std::fmax(std::fmax(hi[0] - lo[0], hi[1] - lo[1]), std::fmax(hi[2] - lo[2], 1e-300));
            std::vector<std::array<double, 6>> feat(nt);
            for (int t = 0; t < nt; ++t) {
                const double* p0 = V + 3 * T[3 * t]; const double* p1 = V + 3 * T[3 * t + 1]; const double* p2 = V + 3 * T[3 * t + 2];
                const double e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
                const double e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
                const double n3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const double len = std::sqrt(n3[0] * n3[0] + n3[1] * n3[1] + n3[2] * n3[2]);
                for (int c3 = 0; c3 < 3; ++c3) {
                    feat[t][c3] = (p0[c3] + p1[c3] + p2[c3]) / (3.0 * ext);
                    feat[t][3 + c3] = len > 0.0 ? 0.5 * n3[c3] / len : 0.0;
                }
            }
            std::vector<int> idx(nt);
            for (int t = 0; t < nt; ++t) idx[t] = t;
            std::vector<std::pair<int, int>> stack;   // [begin, end) ranges still to split
            stack.push_back({0, nt});
            while (!stack.empty()) {
                const auto rg = stack.back();
                stack.pop_back();
                const int cnt = rg.second - rg.first;
                if (cnt <= 64) continue;
                int dim = 0;
                double best = -1.0;
                for (int d = 0; d < 6; ++d) {
                    double mn = 1e300, mx = -1e300;
                    for (int i = rg.first; i < rg.second; ++i) { mn = std::fmin(mn, feat[idx[i]][d]); mx = std::fmax(mx, feat[idx[i]][d]); }
                    if (mx - mn > best) { best = mx - mn; dim = d; }
                }
                std::stable_sort(idx.begin() + rg.first, idx.begin() + rg.second,
                                 [&](int x, int y) { return feat[x][dim] < feat[y][dim]; });
                const int left = ((cnt + 63) / 64 / 2) * 64;
                stack.push_back({rg.first, rg.first + left});
                stack.push_back({rg.first + left, rg.second});
            }
            for (int j = 0; j < nt; ++j) order[j] = {0u, idx[j]};
        }
        const size_t base = (size_t)B.tri_begin[b];
        for (int j = 0; j < nt; ++j) {
            const int t = order[j].second;
            for (int k = 0; k < 3; ++k)
                for (int c3 = 0; c3 < 3; ++c3)
                    soup[(size_t)(3 * k + c3) * n_alloc + base + j] = V[3 * T[3 * t + k] + c3];
        }
        B.tri_end[b] = (int)base + nt;
        // vertex sharing: the unique vertices (by index) of every cluster of 64 and, per triangle, where
        // its three sit in that list; a cluster with more than 64 of them is set up per triangle
        for (size_t c = base / 64; c < (size_t)B.tri_begin[b + 1] / 64; ++c) {
            const size_t j0 = c * 64 - base, j1 = std::min<size_t>(j0 + 64, (size_t)nt);
            std::map<int, int> local;
            bool fits = true;
            for (size_t j = j0; j < j1 && fits; ++j)
                for (int k = 0; k < 3; ++k) {
                    const int vi = T[3 * order[j].second + k];
                    if (local.find(vi) == local.end()) {
                        if (local.size() == 64) { fits = false; break; }
                        const int pos = (int)local.size();
                        local[vi] = pos;
                        for (int c3 = 0; c3 < 3; ++c3) cluster_vtx[c * 192 + 64 * c3 + pos] = V[3 * vi + c3];
                    }
                }
            if (!fits || local.empty()) continue;
            cluster_nv[c] = (int)local.size();
            for (size_t j = j0; j < j1; ++j) {
                unsigned pk = 0;
                for (int k = 0; k < 3; ++k) pk |= (unsigned)local[T[3 * order[j].second + k]] << (8 * k);
                tri_local[base + j] = pk;
            }
        }
        // model-space plane of each triangle, unit normal of its winding (float32 pre-cull only)
        for (int j = 0; j < nt; ++j) {
            double p[3][3];
            for (int k = 0; k < 3; ++k)
                for (int c3 = 0; c3 < 3; ++c3) p[k][c3] = soup[(size_t)(3 * k + c3) * n_alloc + base + j];
            const double e1[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
            const double e2[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
            const double n3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const double len = std::sqrt(n3[0] * n3[0] + n3[1] * n3[1] + n3[2] * n3[2]);
            if (!(len > 0.0) || !std::isfinite(len)) continue;   // zero area: stays NaN = kept (the setup rejects it)
            // offset from the triangle's centroid (the three vertices give the same plane up to rounding)
            const double cx3 = (p[0][0] + p[1][0] + p[2][0]) / 3.0, cy3 = (p[0][1] + p[1][1] + p[2][1]) / 3.0,
                         cz3 = (p[0][2] + p[1][2] + p[2][2]) / 3.0;
            float* pl = &tri_plane[4 * (base + j)];
            pl[0] = (float)(n3[0] / len); pl[1] = (float)(n3[1] / len); pl[2] = (float)(n3[2] / len);
            pl[3] = (float)(-(n3[0] * cx3 + n3[1] * cy3 + n3[2] * cz3) / len);
        }
        // bounding sphere of each cluster of 64 (centre = bbox centre of its vertices)
        for (size_t c = base / 64; c < (size_t)B.tri_begin[b + 1] / 64; ++c) {
            double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
            const size_t j0 = c * 64 - base, j1 = std::min<size_t>(j0 + 64, (size_t)nt);
            for (size_t j = j0; j < j1; ++j)
                for (int k = 0; k < 3; ++k)
                    for (int c3 = 0; c3 < 3; ++c3) {
                        const double x = soup[(size_t)(3 * k + c3) * n_alloc + base + j];
                        clo[c3] = std::fmin(clo[c3], x); chi[c3] = std::fmax(chi[c3], x);
                    }
            double cc[3] = {0.5 * (clo[0] + chi[0]), 0.5 * (clo[1] + chi[1]), 0.5 * (clo[2] + chi[2])}, cr2 = 0.0;
            for (size_t j = j0; j < j1; ++j)
                for (int k = 0; k < 3; ++k) {
                    double d2 = 0.0;
                    for (int c3 = 0; c3 < 3; ++c3) {
                        const double d = soup[(size_t)(3 * k + c3) * n_alloc + base + j] - cc[c3];
                        d2 += d * d;
                    }
                    cr2 = std::fmax(cr2, d2);
                }
            for (int c3 = 0; c3 < 3; ++c3) cluster_sphere[4 * c + c3] = (float)cc[c3];
            cluster_sphere[4 * c + 3] = (float)(std::sqrt(cr2) * 1.0001 + 1e-6);
            if (B.body_cull[b] != 0) {   // cone of the cluster's outward unit normals
                std::vector<std::array<double, 3>> nrm;
                double ax[3] = {0, 0, 0};
                for (size_t j = j0; j < j1; ++j) {
                    double p[3][3];
                    for (int k = 0; k < 3; ++k)
                        for (int c3 = 0; c3 < 3; ++c3) p[k][c3] = soup[(size_t)(3 * k + c3) * n_alloc + base + j];
                    const double e1[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
                    const double e2[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
                    double n3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                    const double len = std::sqrt(n3[0] * n3[0] + n3[1] * n3[1] + n3[2] * n3[2]);
                    if (!(len > 0.0)) continue;   // zero area: never rendered
                    for (int c3 = 0; c3 < 3; ++c3) { n3[c3] *= (double)B.body_cull[b] / len; ax[c3] += n3[c3]; }
                    nrm.push_back({n3[0], n3[1], n3[2]});
                }
                const double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
                if (al > 1e-9 && !nrm.empty()) {
                    double mindp = 1.0;
                    for (const auto& n3 : nrm) mindp = std::fmin(mindp, (n3[0] * ax[0] + n3[1] * ax[1] + n3[2] * ax[2]) / al);
                    for (int c3 = 0; c3 < 3; ++c3) cluster_cone[4 * c + c3] = (float)(ax[c3] / al);
                    cluster_cone[4 * c + 3] = (float)(mindp - 1e-4);   // <= 0: the cone is too wide to cull by
                }
            }
        }
        voff += nv;
        toff += nt;
    }
    B.n_tri = (int)n_alloc;

    const size_t plane = (size_t)h->npx * sizeof(float);
    RBS_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    RBS_HIP(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    RBS_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < rbs_handle::kRing; ++i) {
        RBS_HIP(h, hipEventCreate(&h->ev_start[i]));
        RBS_HIP(h, hipEventCreate(&h->ev_stop[i]));
        RBS_HIP(h, hipEventCreate(&h->ev_copy_start[i]));
        RBS_HIP(h, hipEventCreate(&h->ev_join[i]));
        RBS_HIP(h, hipEventCreate(&h->ev_raster_start[i]));
        RBS_HIP(h, hipEventCreate(&h->ev_raster_stop[i]));
        RBS_HIP(h, hipEventCreate(&h->ev_copy_stop[i]));
    }
    {
        hipDeviceProp_t prop;
        RBS_HIP(h, hipGetDeviceProperties(&prop, h->device));
        h->cu_count = std::max(1, prop.multiProcessorCount);
        h->raster_blocks = 3 * h->cu_count;
        // tuning overrides (defaults are the measured best on MI355X)
        if (const char* m = std::getenv("RBS_RASTER_BLOCKS")) h->raster_blocks = std::max(1, std::atoi(m));
        if (const char* m = std::getenv("RBS_SHARED_TRAIL")) h->stp_allowed = std::atoi(m) != 0;
        if (const char* m = std::getenv("RBS_ONE_BODY")) h->one_body_kernel = std::atoi(m) != 0;
        if (const char* m = std::getenv("RBS_TRACKER_SPLIT_MAX")) h->tracker_split_max = std::atoi(m);
        if (const char* m = std::getenv("RBS_STP_ENTER")) h->stp_enter = std::atof(m);
        if (const char* m = std::getenv("RBS_STP_EVERY")) h->stp_every = std::max(1, std::atoi(m));
        h->split = RBS_SPLIT_DEFAULT != 0;
        if (const char* m = std::getenv("RBS_SPLIT")) h->split = std::atoi(m) != 0;
        h->depth_blocks = RBS_DEPTH_MINWAVES * h->cu_count;
        h->eval_blocks = RBS_EVAL_MINWAVES * h->cu_count;
        if (const char* m = std::getenv("RBS_DEPTH_BLOCKS")) h->depth_blocks = std::max(1, std::atoi(m));
        if (const char* m = std::getenv("RBS_EVAL_BLOCKS")) h->eval_blocks = std::max(1, std::atoi(m));
        h->tile_override = std::getenv("RBS_TILE");
        if (const char* m = std::getenv("RBS_COPY_ROWS")) h->copy_rows = std::atoi(m);
        if (const char* m = std::getenv("RBS_COPY_TPB")) h->copy_tpb = std::max(64, std::atoi(m) / 64 * 64);
        // the layout is the caller's choice (rbs_config.state_layout); the environment is consulted
        // only when the caller leaves it open (tools comparing both layouts of an unchanged caller)
        if (cfg->state_layout == RBS_STATE_DENSE) h->windowed = false;
        else if (cfg->state_layout == RBS_STATE_DEFAULT)
            if (const char* m = std::getenv("RBS_STATE")) h->windowed = std::strcmp(m, "dense") != 0;
        h->smalln_target = 2 * std::max(1, prop.multiProcessorCount);   // measured best at 64..500 particles
        if (const char* m = std::getenv("RBS_SMALLN_TARGET")) h->smalln_target = std::max(1, std::atoi(m));
        if (const char* m = std::getenv("RBS_MID_ENTER")) h->mid_enter = std::atof(m);
        if (const char* m = std::getenv("RBS_BALANCE")) h->balance = std::atoi(m) != 0;
        if (const char* m = std::getenv("RBS_WIDE_ENTER")) { h->wide_enter = std::atof(m); h->wide_leave = h->wide_enter * 0.67; }
        if (const char* m = std::getenv("RBS_RECT_ALIGN")) h->rect_align = std::atoi(m) >= 16 ? 16 : (std::atoi(m) >= 8 ? 8 : 4);
        if (const char* m = std::getenv("RBS_TIMING_EVERY")) h->timing_every = std::max(1, std::atoi(m));
        if (const char* m = std::getenv("RBS_WIN_CHUNKS")) h->win_chunks = h->win_chunks_single = std::min(1024, std::max(1, std::atoi(m)));
        if (const char* m = std::getenv("RBS_UPLOAD_CHUNKS")) h->upload_chunks = std::min(16, std::max(1, std::atoi(m)));
        if (h->cols & 3) h->windowed = false;   // windows move whole float4s
        h->base.rect_align = h->windowed ? h->rect_align : rbs::kRectAlign;
        // whole planes: the big copy kernel must run BESIDE the raster kernel, and three raster
        // blocks per CU hold 504 of a SIMD's 512 VGPRs -- two leave room for the copy waves
        if (!h->windowed && !std::getenv("RBS_RASTER_BLOCKS")) h->raster_blocks = 2 * std::max(1, prop.multiProcessorCount);
    }
    RBS_HIP(h, hipMalloc(&h->d_soup, soup.size() * sizeof(double)));
    RBS_HIP(h, hipMemcpy(h->d_soup, soup.data(), soup.size() * sizeof(double), hipMemcpyHostToDevice));
    B.soup = h->d_soup;
    RBS_HIP(h, hipMalloc(&h->d_frame, plane));
    h->cur_frame = h->d_frame;
    if (h->precision == RBS_PRECISION_F64) {   // F32 derives the per-pixel terms from the observation on the fly
        RBS_HIP(h, hipMalloc(&h->d_aux, sizeof(double) * rbs::AUX_PLANES * (size_t)h->npx));
        RBS_HIP(h, hipMalloc(&h->d_aux_slot[0], sizeof(double) * rbs::AUX_PLANES * (size_t)h->npx));
        RBS_HIP(h, hipMalloc(&h->d_aux_slot[1], sizeof(double) * rbs::AUX_PLANES * (size_t)h->npx));
        h->cur_aux = h->d_aux;
        RBS_HIP(h, hipMalloc(&h->d_pbg, plane));
    }
    RBS_HIP(h, hipMalloc(&h->d_render, plane));
    {   // the occlusion bookkeeping: the caller's choice; left open, the library's (tooling: RBS_OCC=reference|device in the
        // environment replaces DEFAULT only, never a mode the caller named)
        int mode = cfg->occlusion_mode;
        // (a pixel's age must fit 16 bits: c^(65 534 delta_time) <= 2^-40, i.e. the process forgets within the counter's range -- with the
        // reference's constants after 1 628 frames; a process that nearly never forgets, c = p_oo - p_ov above ~0.987 at 30 frames/s, cannot
        // be kept exactly this way)
        const bool ages_fit = std::exp((65534.0 * h->delta_time) * std::log(h->p_oo - h->p_ov)) <= kExactTau;
        const bool can = h->precision == RBS_PRECISION_F64 && h->windowed && ages_fit;
        if (mode == RBS_OCC_REFERENCE && h->precision == RBS_PRECISION_F64 && h->windowed && !ages_fit)
            return fail(h, RBS_ERR_UNSUPPORTED, fmt("occlusion_mode REFERENCE keeps a 16-bit age per pixel: with p_occluded_occluded - p_occluded_visible = %.6g and "
                                                    "delta_time = %.6g a value has not decayed to 2^-40 of itself after 65 534 frames", h->p_oo - h->p_ov, h->delta_time));
        if (mode == RBS_OCC_DEFAULT) {
            mode = RBS_OCC_LIBRARY_DEFAULT;
            if (const char* m = std::getenv("RBS_OCC")) mode = !std::strcmp(m, "reference") ? RBS_OCC_REFERENCE : !std::strcmp(m, "device") ? RBS_OCC_DEVICE_RULE : mode;
            if (mode == RBS_OCC_REFERENCE && !can) mode = RBS_OCC_DEVICE_RULE;
        } else if (mode == RBS_OCC_REFERENCE && !can) {
            return fail(h, RBS_ERR_UNSUPPORTED, "occlusion_mode REFERENCE needs the binary64 likelihood (likelihood_precision F64) and the windowed "
                                                "state layout (cols a multiple of 4)");
        }
        h->exact = mode == RBS_OCC_REFERENCE;
    }
    if (h->exact) {
        // the propagation table: entry a = the two terms of oracle orc_propagate that depend on the elapsed time a * delta_time alone
        const double cc = h->p_oo - h->p_ov, lc = std::log(cc);
        int K = 0;
        while (K < 65534 && std::exp(((double)K * h->delta_time) * lc) > kExactTau) ++K;
        h->age_max = K;
        h->ptab.resize(2 * (size_t)(K + 1));
        for (int a = 0; a <= K; ++a) {
            const double dt = (double)a * h->delta_time;
            const double pow_c = std::exp(dt * lc);
            h->ptab[2 * a] = pow_c;
            h->ptab[2 * a + 1] = (1.0 - h->p_oo) * (pow_c - 1.0) / (cc - 1.0);
        }
        RBS_HIP(h, hipMalloc(&h->d_ptab, sizeof(double) * h->ptab.size()));
        RBS_HIP(h, hipMemcpy(h->d_ptab, h->ptab.data(), sizeof(double) * h->ptab.size(), hipMemcpyHostToDevice));
        RBS_HIP(h, hipMalloc(&h->d_tmp_plane, plane));
    }
    // slabs apply to windowed planes only; a slab as large as a plane is a plane
    h->slab_px = 0;
    if (h->windowed && cfg->state_slab_px > 0 && cfg->state_slab_px < h->npx)
        h->slab_px = std::max(1024, (cfg->state_slab_px + 3) & ~3);
    else if (h->windowed && cfg->state_slab_px == 0 && h->max_particles > 8192 && (h->cols & 3) == 0 && h->npx / 8 >= 4096) {
        // many particles, nothing asked for: an eighth of a plane per slot (8x the particles in the same
        // memory, the same numbers); the slabs grow when a region comes close to filling one
        h->slab_px = (h->npx / 8 + 3) & ~3;
        h->slab_auto = true;
    }
    if (h->slab_px >= h->npx) h->slab_px = 0;
    h->plane_stride = h->slab_px ? (size_t)h->slab_px : (size_t)h->npx;
    if (h->exact) h->plane_stride = exact_stride(h->plane_stride);   // (the slots' ages follow their values)
    RBS_HIP(h, hipMalloc(&h->d_occ[0], occ_alloc_bytes(sizeof(float) * h->plane_stride * h->max_particles)));
    RBS_HIP(h, hipMalloc(&h->d_occ[1], occ_alloc_bytes(sizeof(float) * h->plane_stride * h->max_particles)));
    RBS_HIP(h, hipMalloc(&h->d_err, 2 * sizeof(int)));
    RBS_HIP(h, hipMemset(h->d_err, 0, 2 * sizeof(int)));
    RBS_HIP(h, hipHostMalloc(&h->h_err, 4 * sizeof(int), hipHostMallocDefault));
    h->h_err[0] = h->h_err[1] = h->h_err[2] = h->h_err[3] = 0;
    RBS_HIP(h, hipMalloc(&h->d_bbox, sizeof(int) * 4));
    if (h->slab_px) {
        RBS_HIP(h, hipMalloc(&h->d_reg[0], sizeof(int4) * (size_t)h->max_particles));
        RBS_HIP(h, hipMalloc(&h->d_reg[1], sizeof(int4) * (size_t)h->max_particles));
    }
    RBS_HIP(h, hipMalloc(&h->d_poses, sizeof(double) * 12 * h->n_bodies * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_indices, sizeof(int) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_out, sizeof(double) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_rects[0], sizeof(int) * 4 * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_rects[1], sizeof(int) * 4 * (size_t)h->max_particles));
    if (h->n_bodies > 1 && !(std::getenv("RBS_NO_GROUPS") && std::atoi(std::getenv("RBS_NO_GROUPS")))) {
        RBS_HIP(h, hipMalloc(&h->d_groups[0], sizeof(rbs::Groups) * (size_t)h->max_particles));
        RBS_HIP(h, hipMalloc(&h->d_groups[1], sizeof(rbs::Groups) * (size_t)h->max_particles));
        if (const char* e = std::getenv("RBS_COPY_WALK")) h->copy_walk = std::atoi(e) != 0;
        if (h->windowed)
            for (int k = 0; k < 2; ++k) RBS_HIP(h, hipMalloc(&h->d_strips[k], sizeof(rbs::Strips) * (size_t)h->max_particles));
    }
    RBS_HIP(h, hipMalloc(&h->d_parents[0], sizeof(int) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_parents[1], sizeof(int) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_win[0], sizeof(int4) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_win[1], sizeof(int4) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_win_used, sizeof(int4) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_item_range, sizeof(int2) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_area, sizeof(unsigned long long)));
    RBS_HIP(h, hipHostMalloc(&h->h_area, sizeof(unsigned long long), hipHostMallocDefault));
    RBS_HIP(h, hipEventCreateWithFlags(&h->ev_area, hipEventDisableTiming));
    RBS_HIP(h, hipMalloc(&h->d_ctr, sizeof(int) * 8));
    RBS_HIP(h, hipMemset(h->d_ctr, 0, sizeof(int) * 8));
    RBS_HIP(h, hipMalloc(&h->d_done, sizeof(int) * (size_t)h->max_particles));
    RBS_HIP(h, hipMalloc(&h->d_cluster_sphere, sizeof(float) * cluster_sphere.size()));
    RBS_HIP(h, hipMemcpy(h->d_cluster_sphere, cluster_sphere.data(), sizeof(float) * cluster_sphere.size(),
                         hipMemcpyHostToDevice));
    B.cluster_sphere = h->d_cluster_sphere;
    RBS_HIP(h, hipMalloc(&h->d_cluster_cone, sizeof(float) * cluster_cone.size()));
    RBS_HIP(h, hipMemcpy(h->d_cluster_cone, cluster_cone.data(), sizeof(float) * cluster_cone.size(),
                         hipMemcpyHostToDevice));
    B.cluster_cone = h->d_cluster_cone;
    RBS_HIP(h, hipMalloc(&h->d_cluster_vtx, sizeof(double) * cluster_vtx.size()));
    RBS_HIP(h, hipMemcpy(h->d_cluster_vtx, cluster_vtx.data(), sizeof(double) * cluster_vtx.size(), hipMemcpyHostToDevice));
    B.cluster_vtx = h->d_cluster_vtx;
    RBS_HIP(h, hipMalloc(&h->d_cluster_nv, sizeof(int) * cluster_nv.size()));
    RBS_HIP(h, hipMemcpy(h->d_cluster_nv, cluster_nv.data(), sizeof(int) * cluster_nv.size(), hipMemcpyHostToDevice));
    B.cluster_nv = h->d_cluster_nv;
    RBS_HIP(h, hipMalloc(&h->d_tri_local, sizeof(unsigned) * tri_local.size()));
    RBS_HIP(h, hipMemcpy(h->d_tri_local, tri_local.data(), sizeof(unsigned) * tri_local.size(), hipMemcpyHostToDevice));
    B.tri_local = h->d_tri_local;
    RBS_HIP(h, hipMalloc(&h->d_tri_plane, sizeof(float) * tri_plane.size()));
    RBS_HIP(h, hipMemcpy(h->d_tri_plane, tri_plane.data(), sizeof(float) * tri_plane.size(), hipMemcpyHostToDevice));
    B.tri_plane = reinterpret_cast<const rbs::floatx4*>(h->d_tri_plane);
    {   // float32 copy of the vertices, per body (screen rectangles)
        std::vector<float> vtx;
        size_t vo = 0;
        B.vtx_begin[0] = 0;
        for (int b = 0; b < h->n_bodies; ++b) {
            for (int i = 0; i < cfg->vertex_counts[b]; ++i) {
                for (int c3 = 0; c3 < 3; ++c3) vtx.push_back((float)cfg->vertices[3 * (vo + i) + c3]);
                vtx.push_back(0.f);
            }
            vo += (size_t)cfg->vertex_counts[b];
            B.vtx_begin[b + 1] = (int)vo;
        }
        for (int b = h->n_bodies; b < rbs::kMaxBodies; ++b) B.vtx_begin[b + 1] = (int)vo;
        RBS_HIP(h, hipMalloc(&h->d_vtx, sizeof(float) * std::max<size_t>(vtx.size(), 4)));
        RBS_HIP(h, hipMemcpy(h->d_vtx, vtx.data(), sizeof(float) * vtx.size(), hipMemcpyHostToDevice));
        B.vtx = reinterpret_cast<const rbs::floatx4*>(h->d_vtx);
    }
    if (const char* e = std::getenv("RBS_FRAME_INGEST")) h->frame_ingest = std::atoi(e) != 0;
    if (const char* e = std::getenv("RBS_FRAME_PULL_BYTES")) h->frame_pull_bytes = (size_t)std::max(0L, std::atol(e));
    if (const char* e = std::getenv("RBS_HOST_STAGED_COPIES")) h->host_copies = std::atoi(e) != 0;
    RBS_HIP(h, hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        RBS_HIP(h, hipHostMalloc(&h->h_frames[k], plane, hipHostMallocPortable));   // (every device of a group uploads from shard 0's)
        if (hipHostGetDevicePointer((void**)&h->h_frames_dev[k], h->h_frames[k], 0) != hipSuccess) { (void)hipGetLastError(); h->h_frames_dev[k] = nullptr; }
        RBS_HIP(h, hipEventCreateWithFlags(&h->ev_frame[k], hipEventDisableTiming));
        RBS_HIP(h, hipEventCreateWithFlags(&h->ev_used[k], hipEventDisableTiming));
        if (k == 0) RBS_HIP(h, hipEventCreateWithFlags(&h->ev_reader, hipEventDisableTiming));
        RBS_HIP(h, hipMalloc(&h->d_fin[k], plane));
    }
    h->h_frame = h->h_frames[0];
    {
        const size_t pose_bytes = sizeof(double) * 12 * h->n_bodies * (size_t)h->max_particles;
        h->in_idx_off = pose_bytes;
        const size_t in_bytes = pose_bytes + sizeof(int) * (size_t)h->max_particles;
        RBS_HIP(h, hipHostMalloc(&h->h_in, in_bytes, hipHostMallocDefault));
        RBS_HIP(h, hipHostMalloc(&h->h_out, sizeof(double) * (size_t)h->max_particles, hipHostMallocDefault));
        RBS_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->h_in_dev), h->h_in, 0));
        RBS_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->h_out_dev), h->h_out, 0));
        RBS_HIP(h, hipMalloc(&h->d_in, in_bytes));
        RBS_HIP(h, hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
    }

    // the raster / render kernels carve the LDS depth tile from dynamic shared memory
    {
        const void* kernels[] = {
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_f64<false, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_f64<false, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_f32<false, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_f32<false, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_f64<true, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_f64<true, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_f32<true, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_f32<true, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_many_f64<false, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_many_f64<false, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_many_f32<false, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_many_f32<false, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_many_f64<true, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_many_f64<true, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_many_f32<true, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_many_f32<true, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_one_f64<false, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_one_f64<false, true>),
            reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_one_f64<true, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_one_f64<true, true>)};
        for (const void* k : kernels)
            RBS_HIP(h, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rbs::smem_bytes(rbs::kTilePxBig, false, true)));
        if (h->exact) {
            const void* xk[] = {
#define RBS_X(U, S) reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_exact_f64<U, S, false, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_exact_f64<U, S, false, true>), \
                    reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_exact_f64<U, S, true, false>), reinterpret_cast<const void*>(&rbs::rbs_raster_kernel_exact_f64<U, S, true, true>)
                RBS_X(false, false), RBS_X(false, true), RBS_X(true, false), RBS_X(true, true)};
#undef RBS_X
            for (const void* k : xk)
                RBS_HIP(h, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rbs::smem_bytes(rbs::kTilePxBig, false, true)));
        }
    }
    RBS_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&rbs::rbs_render_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)rbs::smem_bytes(rbs::kTilePxBig, false)));

    {   // per-item partial sums: sized for the default tiling so no call ever allocates
        const size_t gmul = h->d_groups[0] ? rbs::kMaxGroups : 1;   // every group of bodies tiles its own rectangle
        size_t need = gmul * (size_t)h->max_particles * tiles_upper_bound(h->cols, h->rows, 256, h->exact ? rbs::kTilePxExact : h->split ? rbs::kDepthTilePx : rbs::kTilePx);
        for (int nn = 1; nn < std::min(h->max_particles, h->raster_blocks); nn *= 2) {
            const int th = std::max(4, rbs::kTilePx / 256 / std::max(1, std::max(h->smalln_target, h->raster_blocks) / nn));
            need = std::max(need, gmul * (size_t)std::min(2 * nn, h->max_particles) * tiles_upper_bound(h->cols, h->rows, 256, 256 * th));
        }
        RBS_HIP(h, hipMalloc(&h->d_partial, sizeof(double) * need));
        RBS_HIP(h, hipMalloc(&h->d_item_particle, sizeof(int) * need));
        h->partial_cap = need;
    }
    // HIP creates a stream's hardware queue at its first submission: do that now, not inside the
    // first updating call
    hipLaunchKernelGGL(rbs::rbs_fill_kernel, dim3(1), dim3(64), 0, h->copy_stream, h->d_render, (size_t)1, 0.f);
    RBS_HIP(h, hipGetLastError());
    RBS_HIP(h, hipEventRecord(h->ev_fork, h->copy_stream));
    RBS_HIP(h, hipStreamSynchronize(h->copy_stream));
    // no observation yet: every pixel "no reading"
    for (int p = 0; p < h->npx; ++p) h->h_frame[p] = NAN;
    if (int32_t rc = upload_frame(h, h->h_frame)) return rc;
    return rbs_reset(h);
}


// ----------------------------------------------------------------------------- several devices
// Copy a shard's error message up to the group handle the caller holds.
int32_t gfail(rbs_handle* g, rbs_handle* shard, int32_t rc)
{
    g->err = fmt("device %d: %s", shard->device, shard->err.c_str());
    return rc;
}

// A failure after part of a call was enqueued: wait for everything that was enqueued (nothing may
// still be writing when the caller reacts), and refuse further calls until rbs_reset.
int32_t poison(rbs_handle* h, int32_t rc)
{
    const std::string why = h->err;
    std::vector<rbs_handle*> all(h->shards.begin(), h->shards.end());
    if (all.empty()) all.push_back(h);
    for (rbs_handle* sh : all) {
        (void)hipSetDevice(sh->device);
        if (sh->stream) (void)hipStreamSynchronize(sh->stream);
        if (sh->copy_stream) (void)hipStreamSynchronize(sh->copy_stream);
        if (sh->up_stream) (void)hipStreamSynchronize(sh->up_stream);
    }
    (void)hipGetLastError();
    h->poisoned = true;
    h->poison_msg = why;
    h->err = why;
    return rc;
}
#define RBS_REFUSE_POISONED(h)                                                                \
    do {                                                                                      \
        if ((h)->poisoned)                                                                    \
            return fail(h, RBS_ERR_HIP, "the handle is in an undefined state after a failed call (" + (h)->poison_msg + \
                                            "): call rbs_reset / rbs_tracker_initialize");    \
    } while (0)

// Every shard orders its next call after EVERY shard's previous one: a call reads parents from
// the other shards' current planes (complete only when their previous updating call is) and
// overwrites the buffer the other shards' previous call was still reading.  All waits are
// issued before any shard enqueues (an event waited on is the one last recorded).
int32_t group_begin_call(rbs_handle* g, hipStream_t const* streams, bool update)
{
    const int nd = (int)g->shards.size();
    // the shared trail: ONE decision for every shard of this call (rbs_handle::stp_now), from the largest window fraction any
    // shard has sampled so far
    if (g->stp_leave_pending) {   // the previous call left the shared trail
        g->stp = false;
        g->stp_block_until = g->calls + 4000;
        g->stp_leave_pending = false;
    }
    g->stp_now = rbs_handle::StpNow();
    if (g->windowed && g->stp_allowed) {
        double frac = 0.0;
        for (rbs_handle* h : g->shards) frac = std::max(frac, h->area_frac);
        g->stp_now = stp_decide(g, update, frac);
        if (g->stp_now.rebase >= 0) for (rbs_handle* h : g->shards) h->area_frac = 0.0;
        if (g->stp_now.rebase == -2) { g->stp_leave_pending = true; for (rbs_handle* h : g->shards) h->area_frac = 1.0; }
    }
    g->calls += 1;
    for (int k = 0; k < nd; ++k) {
        g->snap_occ[k] = g->shards[k]->d_occ[g->shards[k]->cur];
        g->snap_win[k] = g->shards[k]->d_win[g->shards[k]->cur];
        g->snap_reg[k] = g->shards[k]->d_reg[g->shards[k]->cur];
    }
    for (int a = 0; a < nd; ++a) {
        rbs_handle* A = g->shards[a];
        RBS_HIP(g, hipSetDevice(A->device));
        for (int b = 0; b < nd; ++b)
            if (b != a) RBS_HIP(g, hipStreamWaitEvent(streams ? streams[a] : A->stream, g->shards[b]->ev_done, 0));
    }
    return RBS_OK;
}

// A shard that evaluates no particle in an updating call still moves its state on with the group -- its copy of the shared
// plane included (entering, stepping, re-basing, leaving: every device's copy stays the same plane).
int32_t advance_empty(rbs_handle* h, bool update)
{
    if (!update) return RBS_OK;
    float alpha, beta;
    occlusion_coeffs(h, h->pending_frames, &alpha, &beta);
    const float bg_new = h->exact ? exact_background(h, h->update_clock + h->pending_frames) : std::fmaf(alpha, h->background, beta);
    rbs_handle* g = h->group;
    if (g && h->windowed && g->stp_allowed) {
        const rbs_handle::StpNow d = g->stp_now;
        if (d.entering || (g->stp && !h->stp && d.rebase != -2)) {
            for (int k = 0; k < 2; ++k)
                if (!h->d_bgp[k]) RBS_HIP(h, hipMalloc(&h->d_bgp[k], sizeof(float) * (h->exact ? exact_stride((size_t)h->npx) : (size_t)h->npx)));
            if (!h->d_bgp_box) RBS_HIP(h, hipMalloc(&h->d_bgp_box, sizeof(int4)));
            hipLaunchKernelGGL(rbs::rbs_fill_kernel, dim3(256), dim3(256), 0, h->stream, h->d_bgp[h->cur], (size_t)h->npx, h->background);
            RBS_HIP(h, hipGetLastError());
            if (h->exact) RBS_HIP(h, hipMemsetAsync(h->d_bgp[h->cur] + h->npx, 0xff, sizeof(unsigned short) * (size_t)h->npx, h->stream));
            h->stp = true;
            h->wide = false;
        }
        if (h->stp) {
            DevParams P = h->base;
            P.alpha = alpha; P.beta = beta; P.bg_new = bg_new;
            P.plane_stride = (int)h->plane_stride;
            P.plane_px = h->slab_px ? h->slab_px : h->npx;
            const unsigned e = (unsigned)std::min(h->pending_frames, 0xffff);
            P.elapsed2 = e | (e << 16);
            int rb = d.rebase;
            const float* occ = nullptr; const int4* win = nullptr; const int4* reg = nullptr;
            if (rb >= 0) {
                const int owner = rb / g->shard_cap;
                occ = g->snap_occ[owner]; win = g->snap_win[owner]; reg = h->slab_px ? g->snap_reg[owner] : (const int4*)nullptr;
                rb -= owner * g->shard_cap;
            }
            if (int32_t rc = launch_bgp_step(h, P, occ, win, reg, rb, h->stream)) return rc;
            if (d.rebase == -2) { h->stp = false; h->area_frac = 1.0; }
        }
    }
    h->background = bg_new;
    h->cur = 1 - h->cur;
    h->update_clock += h->pending_frames;
    h->pending_frames = 0;
    // (h->calls stays: it selects the work-item counters, and a raster kernel zeroes the counters of
    // the call that FOLLOWS it -- a skipped call must not change which pair is next)
    return RBS_OK;
}

int32_t group_load_rccl(rbs_handle* g, const std::vector<int>& devs)
{
    Rccl* r = new Rccl;
    g->rccl = r;
    // a copy the process has loaded already (PyTorch ships its own librccl) before a second one from the ROCm tree
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        r->lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        if (r->lib) break;
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (r->lib) break;
        r->lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r->lib) return fail(g, RBS_ERR_UNSUPPORTED, fmt("several devices need RCCL: dlopen(librccl.so.1) failed: %s", dlerror()));
    r->CommInitAll = reinterpret_cast<decltype(r->CommInitAll)>(dlsym(r->lib, "ncclCommInitAll"));
    r->CommDestroy = reinterpret_cast<decltype(r->CommDestroy)>(dlsym(r->lib, "ncclCommDestroy"));
    r->GroupStart = reinterpret_cast<decltype(r->GroupStart)>(dlsym(r->lib, "ncclGroupStart"));
    r->GroupEnd = reinterpret_cast<decltype(r->GroupEnd)>(dlsym(r->lib, "ncclGroupEnd"));
    r->AllGather = reinterpret_cast<decltype(r->AllGather)>(dlsym(r->lib, "ncclAllGather"));
    r->GetErrorString = reinterpret_cast<decltype(r->GetErrorString)>(dlsym(r->lib, "ncclGetErrorString"));
    if (!r->CommInitAll || !r->CommDestroy || !r->GroupStart || !r->GroupEnd || !r->AllGather || !r->GetErrorString)
        return fail(g, RBS_ERR_UNSUPPORTED, "librccl lacks an ncclCommInitAll / ncclAllGather / ncclGroup* symbol");
    r->comms.assign(devs.size(), nullptr);
    const int rc = r->CommInitAll(r->comms.data(), (int)devs.size(), devs.data());
    if (rc != 0) {
        const std::string msg = fmt("ncclCommInitAll over %zu devices failed: %s", devs.size(), r->GetErrorString(rc));
        r->comms.clear();
        return fail(g, RBS_ERR_HIP, msg);
    }
    return RBS_OK;
}

// Every shard gets every shard's `count` doubles: buf[k * count .. ) of shard k's buffer is what
// shard k produced.  RCCL all-gather over xGMI when the devices are distinct (the exchange of the
// log-likelihoods before resampling); plain device-to-device copies when a device appears twice
// in the handle (RCCL refuses that: functional tests on a one-GPU box).
int32_t group_allgather(rbs_handle* g, double* const* bufs, size_t count, hipStream_t const* streams)
{
    const int nd = (int)g->shards.size();
    if (g->rccl && !g->rccl->comms.empty()) {
        Rccl* r = g->rccl;
        int rc = r->GroupStart();
        for (int k = 0; k < nd && rc == 0; ++k)
            rc = r->AllGather(bufs[k] + (size_t)k * count, bufs[k], count, /*ncclDouble*/ 8, r->comms[k], streams[k]);
        const int rc2 = r->GroupEnd();
        if (rc != 0 || rc2 != 0) return fail(g, RBS_ERR_HIP, fmt("ncclAllGather failed: %s", r->GetErrorString(rc ? rc : rc2)));
        return RBS_OK;
    }
    // one event per producer, then every consumer copies every other shard's piece on its own stream
    for (int k = 0; k < nd; ++k) {
        RBS_HIP(g, hipSetDevice(g->shards[k]->device));
        RBS_HIP(g, hipEventRecord(g->shards[k]->ev_fork, streams[k]));
    }
    for (int a = 0; a < nd; ++a) {
        RBS_HIP(g, hipSetDevice(g->shards[a]->device));
        for (int b = 0; b < nd; ++b) {
            if (b == a) continue;
            RBS_HIP(g, hipStreamWaitEvent(streams[a], g->shards[b]->ev_fork, 0));
            RBS_HIP(g, hipMemcpyPeerAsync(bufs[a] + (size_t)b * count, g->shards[a]->device, bufs[b] + (size_t)b * count,
                                          g->shards[b]->device, sizeof(double) * count, streams[a]));
        }
    }
    return RBS_OK;
}

int32_t create_group(const rbs_config* cfg, rbs_handle* g)
{
    const int nd = cfg->n_devices;
    if (nd > rbs::kMaxDevices) return fail(g, RBS_ERR_INVALID_ARGUMENT, fmt("n_devices %d > %d", nd, rbs::kMaxDevices));
    if (!cfg->device_ids) return fail(g, RBS_ERR_INVALID_ARGUMENT, "n_devices > 1 but device_ids is NULL");
    if (cfg->max_particles <= 0) return fail(g, RBS_ERR_INVALID_ARGUMENT, "max_particles must be positive");
    g->max_particles = cfg->max_particles;
    g->shard_cap = (cfg->max_particles + nd - 1) / nd;
    g->rows = cfg->rows; g->cols = cfg->cols; g->npx = cfg->rows * cfg->cols;
    g->n_bodies = cfg->n_objects;
    g->device = cfg->device_ids[0];
    std::vector<int> devs(cfg->device_ids, cfg->device_ids + nd);
    bool distinct = true;
    for (int a = 0; a < nd; ++a)
        for (int b = a + 1; b < nd; ++b)
            if (devs[a] == devs[b]) distinct = false;
    for (int k = 0; k < nd; ++k) {
        rbs_config sub = *cfg;
        sub.device_id = devs[k];
        sub.n_devices = 0;
        sub.device_ids = nullptr;
        sub.max_particles = g->shard_cap;
        rbs_handle* sh = new (std::nothrow) rbs_handle;
        if (!sh) return fail(g, RBS_ERR_OUT_OF_MEMORY, "out of host memory");
        g->shards.push_back(sh);
        sh->group = g;
        sh->shard_index = k;
        sh->shard_cap = g->shard_cap;
        if (int32_t rc = create_impl(&sub, sh)) return gfail(g, sh, rc);
        RBS_HIP(g, hipEventCreateWithFlags(&sh->ev_done, hipEventDisableTiming));
        RBS_HIP(g, hipEventRecord(sh->ev_done, sh->stream));
    }
    g->windowed = g->shards[0]->windowed;
    g->precision = g->shards[0]->precision;
    g->stp_allowed = g->shards[0]->stp_allowed;      // (the shared trail's policy is the group's: one decision per call for every shard)
    g->stp_enter = g->shards[0]->stp_enter;
    g->stp_every = g->shards[0]->stp_every;
    g->wide_enter = g->shards[0]->wide_enter;
#ifdef RBS_TEST_HOOKS   // (librbsensor_mi355x_hooks.so, `make hooks`: the release library carries no fault injection -- ADVICE r3)
    if (const char* f = std::getenv("RBS_TEST_FAULT")) {
        int a = -1; long c = -1;
        if (std::sscanf(f, "%d:%ld", &a, &c) == 2) { g->fault_shard = a; g->fault_call = c; }
    }
#endif
    // parents on another device are read in place
    for (int a = 0; a < nd; ++a) {
        RBS_HIP(g, hipSetDevice(devs[a]));
        for (int b = 0; b < nd; ++b) {
            if (devs[a] == devs[b]) continue;
            int can = 0;
            RBS_HIP(g, hipDeviceCanAccessPeer(&can, devs[a], devs[b]));
            if (!can) return fail(g, RBS_ERR_UNSUPPORTED, fmt("device %d cannot access device %d's memory (peer access)", devs[a], devs[b]));
            const hipError_t e = hipDeviceEnablePeerAccess(devs[b], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                (void)hipGetLastError();
                return fail(g, RBS_ERR_HIP, fmt("hipDeviceEnablePeerAccess(%d -> %d): %s", devs[a], devs[b], hipGetErrorString(e)));
            }
            (void)hipGetLastError();
        }
    }
    bool want_rccl = distinct;
#ifdef RBS_TEST_HOOKS   // (hooks library only) RBS_TEST_FORCE_RCCL=1: take the RCCL branch on a one-GPU box too -- ncclCommInitAll refuses a
                        // device that appears twice, which is the failure the fall-back below exists for
    if (const char* e = std::getenv("RBS_TEST_FORCE_RCCL")) want_rccl = want_rccl || std::atoi(e) != 0;
#endif
    if (want_rccl) {
        if (int32_t rc = group_load_rccl(g, devs)) {
            // First contact with a node must not end at the communicator: the exchange is n doubles per block, and the devices can
            // read each other (checked above) -- the peer-copy form of group_allgather does the same job.  RBS_REQUIRE_RCCL=1: fatal.
            const char* req = std::getenv("RBS_REQUIRE_RCCL");
            if (req && std::atoi(req) != 0) return rc;
            std::fprintf(stderr, "[rbsensor_mi355x] RCCL is not used for this handle (%s): log-likelihoods are exchanged by peer copies\n", g->err.c_str());
            if (g->rccl) {
                for (auto c : g->rccl->comms)
                    if (c) (void)g->rccl->CommDestroy(c);
                delete g->rccl;
                g->rccl = nullptr;
            }
            g->err.clear();
        }
    }
    return RBS_OK;
}

void release_group(rbs_handle* g)
{
    if (g->rccl) {
        for (auto c : g->rccl->comms)
            if (c) (void)g->rccl->CommDestroy(c);
        // the library stays loaded: other handles (or torch) may be using it
        delete g->rccl;
        g->rccl = nullptr;
    }
    for (rbs_handle* sh : g->shards) {
        if (sh->ev_done) { (void)hipSetDevice(sh->device); (void)hipStreamSynchronize(sh->stream); (void)hipEventDestroy(sh->ev_done); sh->ev_done = nullptr; }
        release(sh);
    }
    g->shards.clear();
    delete g;
}

// The host-pointer call on one device: poses and parent slots are staged in pinned memory, and
// -- no copy engine involved -- the rectangles kernel pulls them from there over PCIe (writing the
// device copy of the poses the raster kernel reads) and the raster kernel stores the
// log-likelihoods straight into the pinned result buffer.  A separate H2D copy of 200 KB queued
// behind the frame's upload on the same engine, and each copy <-> kernel hand-over costs ~10 us:
// together 30 us of a 300 us step.  RBS_HOST_STAGED_COPIES=1 restores the copies.
// rbs_loglikes_deltas: state deltas + default poses instead of absolute poses (packed on the way into pinned memory,
// composed on the device by the rectangles kernel, rbs_prep_deltas_kernel).
struct DeltaArgs { const double* deltas; const double* deflt; int stride; };
#ifndef RBS_TRACKER_SPLIT_MAX_DEFAULT
#define RBS_TRACKER_SPLIT_MAX_DEFAULT 5000   // (measured, tests/cpp/host_bench --tracker: frame by frame +12-16 % at 1 000-2 000 particles, +6-10 % at 4 000, nothing from 6 000 up)
#endif

int32_t host_call(rbs_handle* h, const double* poses, const int32_t* indices, int n, bool update, const DeltaArgs* da = nullptr)
{
    const bool copies = h->host_copies && !da;
    const size_t pose_bytes = sizeof(double) * 12 * h->n_bodies * (size_t)n;
    // slabs: the overflow flag as it stands BEFORE this call -- set, it belongs to an asynchronous call
    // that has not been reported yet
    if (h->slab_px) RBS_HIP(h, hipMemcpyAsync(h->h_err + 2, h->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (da) {
        // [n][bodies][6] deltas, then [bodies][6] default poses: half the bytes of the absolute poses they stand for
        // (they fit the staging block sized for those: (n + 1) * 6 <= max_particles * 12)
        const int B = h->n_bodies;
        double* st = reinterpret_cast<double*>(h->h_in);
        const size_t nb = (size_t)n * B;
        if (da->deltas == st) {}   // (rbs_deltas_buffer: the caller gathered straight into the staging block)
        else if (da->stride == 6) std::memcpy(st, da->deltas, sizeof(double) * 6 * nb);
        else for (size_t k = 0; k < nb; ++k) std::memcpy(st + 6 * k, da->deltas + (size_t)da->stride * k, sizeof(double) * 6);
        for (int b = 0; b < B; ++b) std::memcpy(st + 6 * nb + 6 * b, da->deflt + (size_t)da->stride * b, sizeof(double) * 6);
        std::memcpy(h->h_in + h->in_idx_off, indices, sizeof(int) * (size_t)n);
        // (the rectangles kernel composes: the wave that owns a particle reads its 48 bytes per body from here)
        if (h->lazy_frame) if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;   // (a pending device frame: its ingest has no deltas variant)
        if (int32_t rc = enqueue_loglikes(h, reinterpret_cast<const double*>(h->d_in),
                                          reinterpret_cast<const int*>(h->h_in_dev + h->in_idx_off), n, update,
                                          reinterpret_cast<double*>(h->h_out_dev), h->stream, nullptr,
                                          reinterpret_cast<const double*>(h->h_in_dev)))
            return rc;
        if (h->slab_px) RBS_HIP(h, hipMemcpyAsync(h->h_err, h->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        RBS_HIP(h, hipEventRecord(h->ev_out, h->stream));
        return RBS_OK;
    }
    std::memcpy(h->h_in, poses, pose_bytes);
    std::memcpy(h->h_in + h->in_idx_off, indices, sizeof(int) * (size_t)n);
    if (copies) {
        if (pose_bytes == h->in_idx_off) {
            RBS_HIP(h, hipMemcpyAsync(h->d_in, h->h_in, pose_bytes + sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
        } else {
            RBS_HIP(h, hipMemcpyAsync(h->d_in, h->h_in, pose_bytes, hipMemcpyHostToDevice, h->stream));
            RBS_HIP(h, hipMemcpyAsync(h->d_in + h->in_idx_off, h->h_in + h->in_idx_off, sizeof(int) * (size_t)n,
                                      hipMemcpyHostToDevice, h->stream));
        }
        if (int32_t rc = enqueue_loglikes(h, reinterpret_cast<const double*>(h->d_in),
                                          reinterpret_cast<const int*>(h->d_in + h->in_idx_off), n, update, h->d_out, h->stream))
            return rc;
        RBS_HIP(h, hipMemcpyAsync(h->h_out, h->d_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    } else {
        if (int32_t rc = enqueue_loglikes(h, reinterpret_cast<const double*>(h->d_in),
                                          reinterpret_cast<const int*>(h->h_in_dev + h->in_idx_off), n, update,
                                          reinterpret_cast<double*>(h->h_out_dev), h->stream,
                                          reinterpret_cast<const double*>(h->h_in_dev)))
            return rc;
    }
    if (h->slab_px) RBS_HIP(h, hipMemcpyAsync(h->h_err, h->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    RBS_HIP(h, hipEventRecord(h->ev_out, h->stream));
    return RBS_OK;
}

// Slabs of a group: every shard keeps the SAME slab size (a shard addresses its neighbours' planes
// with its own stride).  before != nullptr: a call overflowed -- take it back on every shard first.
// Otherwise housekeeping: enlarge when the largest region asked for on any shard has filled three
// quarters of a slab.  Drains every shard when it acts.
int32_t group_grow_slabs(rbs_handle* g, const std::vector<CallState>* before)
{
    rbs_handle* s0 = g->shards[0];
    if (!s0->slab_px) return RBS_OK;
    int need = 0;
    bool overflowed = false;
    for (rbs_handle* h : g->shards) { need = std::max(need, h->h_err[1]); overflowed = overflowed || h->h_err[0] != 0; }
    if (!before && !overflowed) {   // housekeeping: act (and drain every shard) only when the slabs will really be reallocated
        const int would = (int)std::min<long>(s0->npx, ((long)slab_for(s0, need) + 1023) & ~1023L);
        if ((long)need * 4 <= (long)s0->slab_px * 3 || would <= s0->slab_px) return RBS_OK;
    }
    for (size_t k = 0; k < g->shards.size(); ++k) {
        rbs_handle* h = g->shards[k];
        RBS_HIP(g, hipSetDevice(h->device));
        if (int32_t rc = drain(h, true)) return gfail(g, h, rc);
        if (before) restore_call_state(h, (*before)[k]);
    }
    const int target = slab_for(s0, need);
    for (rbs_handle* h : g->shards) {
        RBS_HIP(g, hipSetDevice(h->device));
        if (int32_t rc = grow_slabs(h, target)) return gfail(g, h, rc);
        if (h->h_err[0]) {
            RBS_HIP(g, hipMemsetAsync(h->d_err, 0, sizeof(int), h->stream));
            h->h_err[0] = 0;
        }
        RBS_HIP(g, hipStreamSynchronize(h->stream));
        RBS_HIP(g, hipEventRecord(h->ev_done, h->stream));
    }
    return RBS_OK;
}

// rbs_loglikes on a group: particle i is evaluated by shard i / shard_cap and (update) written
// to global slot i; `indices` are global parent slots.
int32_t group_loglikes(rbs_handle* g, const double* poses, int32_t* indices, int32_t n, int32_t update, double* out, const DeltaArgs* da = nullptr)
{
    const int nd = (int)g->shards.size();
    const int cap = g->shard_cap;
    for (int32_t i = 0; i < n; ++i)
        if (indices[i] < 0 || indices[i] >= nd * cap)
            return fail(g, RBS_ERR_INVALID_ARGUMENT, fmt("loglikes: indices[%d] = %d outside 0..%d", i, indices[i], nd * cap - 1));
    std::vector<CallState> before;
    for (rbs_handle* h : g->shards) before.push_back(save_call_state(h));
    const CallState gbefore = save_call_state(g);      // (the shared trail's state is the group's)
    const bool gleave = g->stp_leave_pending;
    const size_t stride = (size_t)12 * g->n_bodies;
    bool stale_overflow = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // from here on a failure leaves some shards advanced and others not: the group is poisoned
        if (int32_t rc = group_begin_call(g, nullptr, update != 0)) return poison(g, rc);
        for (int k = 0; k < nd; ++k) {
            rbs_handle* h = g->shards[k];
            const int lo = std::min(n, k * cap), cnt = std::min(n, (k + 1) * cap) - lo;
            if (hipSetDevice(h->device) != hipSuccess) return poison(g, fail(g, RBS_ERR_HIP, fmt("hipSetDevice(%d) failed", h->device)));
            if (cnt <= 0) {
                if (int32_t rc = advance_empty(h, update != 0)) return poison(g, gfail(g, h, rc));
                if (hipEventRecord(h->ev_done, h->stream) != hipSuccess) return poison(g, fail(g, RBS_ERR_HIP, "hipEventRecord failed"));
                continue;
            }
#ifdef RBS_TEST_HOOKS
            if (k == g->fault_shard && g->group_calls == g->fault_call) {   // (test hook)
                g->group_calls += 1;
                return poison(g, gfail(g, h, fail(h, RBS_ERR_HIP, "injected fault (RBS_TEST_FAULT)")));
            }
#endif
            DeltaArgs dk = {nullptr, nullptr, 0};
            if (da) dk = DeltaArgs{da->deltas + (size_t)da->stride * g->n_bodies * (size_t)lo, da->deflt, da->stride};
            if (int32_t rc = host_call(h, da ? nullptr : poses + stride * (size_t)lo, indices + lo, cnt, update != 0, da ? &dk : nullptr))
                return poison(g, gfail(g, h, rc));
        }
        g->group_calls += 1;
        bool overflow = false;
        for (int k = 0; k < nd; ++k) {
            rbs_handle* h = g->shards[k];
            const int lo = std::min(n, k * cap), cnt = std::min(n, (k + 1) * cap) - lo;
            if (cnt <= 0) continue;
            if (hipSetDevice(h->device) != hipSuccess || hipEventSynchronize(h->ev_out) != hipSuccess)
                return poison(g, fail(g, RBS_ERR_HIP, fmt("device %d: waiting for the log-likelihoods failed", h->device)));
            std::memcpy(out + lo, h->h_out, sizeof(double) * (size_t)cnt);
            overflow = overflow || (h->slab_px && h->h_err[0]);
            // the flag as it stood BEFORE this call (host_call fetches it): set, it belongs to an asynchronous call whose
            // results were NaN and which has not been reported yet -- repaired below like this call's own, but REPORTED
            if (attempt == 0 && h->slab_px && h->h_err[2]) stale_overflow = true;
        }
        if (!overflow) break;
        if (attempt == 1) return gfail(g, g->shards[0], check_slab_error(g->shards[0]));
        // a region did not fit its slab on some shard: the whole call is taken back on every shard
        // (the planes it read are intact), every shard's slabs are enlarged alike -- a shard reads
        // its neighbours' planes with its own stride -- and the call runs again
        if (int32_t rc = group_grow_slabs(g, &before)) return poison(g, rc);
        restore_call_state(g, gbefore);
        g->stp_leave_pending = gleave;
    }
    if (update)
        for (int32_t i = 0; i < n; ++i) indices[i] = i;
    if (int32_t rc = group_grow_slabs(g, nullptr)) return rc;   // (housekeeping: enlarge before a region fills a slab)
    if (stale_overflow) {   // this call's numbers are good; an earlier asynchronous call's were not (single-device path: the same, once)
        rbs_handle* s0 = g->shards[0];
        s0->h_err[0] = 1;
        const int32_t rc = check_slab_error(s0);
        s0->h_err[0] = 0;
        return gfail(g, s0, rc);
    }
    return RBS_OK;
}

// rbs_set_observation_device on a group: `d_depth` lives on the handle's FIRST device; every shard
// ingests it from there (its ingest kernel reads the 1.2 MB over xGMI, peer access) once the work
// enqueued on the caller's stream so far -- whatever produced the frame -- has run.
int32_t group_set_observation_device(rbs_handle* g, const float* d_depth, hipStream_t stream)
{
    rbs_handle* s0 = g->shards[0];
    RBS_HIP(g, hipSetDevice(s0->device));
    hipStream_t cs = stream ? stream : s0->stream;
    RBS_HIP(g, hipEventRecord(s0->ev_reader, cs));
    for (size_t k = 0; k < g->shards.size(); ++k) {
        rbs_handle* h = g->shards[k];
        if (hipSetDevice(h->device) != hipSuccess) return poison(g, fail(g, RBS_ERR_HIP, fmt("hipSetDevice(%d) failed", h->device)));
        if (int32_t rc = flush_lazy_frame(h, h->stream)) return k == 0 ? gfail(g, h, rc) : poison(g, gfail(g, h, rc));
        if (int32_t rc = release_frame_slot(h)) return k == 0 ? gfail(g, h, rc) : poison(g, gfail(g, h, rc));
        if (hipStreamWaitEvent(h->stream, s0->ev_reader, 0) != hipSuccess) return poison(g, fail(g, RBS_ERR_HIP, "hipStreamWaitEvent failed"));
        h->lazy_frame = d_depth;
        h->lazy_stream = h->stream;
        h->pending_frames += 1;
    }
    return RBS_OK;
}

// rbs_loglikes_device on a group: poses / parent slots / results are arrays on the handle's FIRST
// device in global particle order; shard k evaluates particles [k cap, (k+1) cap) on its own stream
// -- its rectangles kernel pulls its slice of the poses and parent slots over xGMI (the route the
// host-pointer call uses for pinned memory), its raster kernel stores its log-likelihoods into the
// caller's array in place -- after the work enqueued on the caller's stream so far, and the caller's
// stream is ordered after every shard: d_out is complete in `stream` order, nothing synchronises
// with the host.
int32_t group_loglikes_device(rbs_handle* g, const double* d_poses, const int32_t* d_indices, int32_t n, int32_t update,
                              double* d_out, hipStream_t stream)
{
    const int nd = (int)g->shards.size(), cap = g->shard_cap;
    rbs_handle* s0 = g->shards[0];
    RBS_HIP(g, hipSetDevice(s0->device));
    hipStream_t cs = stream ? stream : s0->stream;
    RBS_HIP(g, hipEventRecord(s0->ev_reader, cs));           // the producers of poses / indices on the caller's stream
    for (int k = 0; k < nd; ++k) {
        RBS_HIP(g, hipSetDevice(g->shards[k]->device));
        RBS_HIP(g, hipStreamWaitEvent(g->shards[k]->stream, s0->ev_reader, 0));
    }
    // from here on a failure leaves some shards advanced and others not: the group is poisoned
    if (int32_t rc = group_begin_call(g, nullptr, update != 0)) return poison(g, rc);
    const size_t stride = (size_t)12 * g->n_bodies;
    for (int k = 0; k < nd; ++k) {
        rbs_handle* h = g->shards[k];
        const int lo = std::min(n, k * cap), cnt = std::min(n, (k + 1) * cap) - lo;
        if (hipSetDevice(h->device) != hipSuccess) return poison(g, fail(g, RBS_ERR_HIP, fmt("hipSetDevice(%d) failed", h->device)));
        if (cnt <= 0) {
            if (int32_t rc = advance_empty(h, update != 0)) return poison(g, gfail(g, h, rc));
            if (hipEventRecord(h->ev_done, h->stream) != hipSuccess) return poison(g, fail(g, RBS_ERR_HIP, "hipEventRecord failed"));
            continue;
        }
        if (int32_t rc = enqueue_loglikes(h, reinterpret_cast<const double*>(h->d_in), d_indices + lo, cnt, update != 0, d_out + lo,
                                          h->stream, d_poses + stride * (size_t)lo))
            return poison(g, gfail(g, h, rc));
    }
    if (hipSetDevice(s0->device) != hipSuccess) return poison(g, fail(g, RBS_ERR_HIP, "hipSetDevice failed"));
    for (int k = 0; k < nd; ++k)
        if (hipStreamWaitEvent(cs, g->shards[k]->ev_done, 0) != hipSuccess) return poison(g, fail(g, RBS_ERR_HIP, "hipStreamWaitEvent failed"));
    return RBS_OK;
}

// Route a global slot to its shard.
rbs_handle* shard_of(rbs_handle* g, int32_t slot, int32_t* local)
{
    const int k = slot / g->shard_cap;
    *local = slot - k * g->shard_cap;
    return g->shards[k];
}
#define RBS_GROUP_SLOT(h, slot, call)                                                         \
    do {                                                                                      \
        if (!(h)->shards.empty()) {                                                           \
            if ((slot) < 0 || (slot) >= (int)(h)->shards.size() * (h)->shard_cap)             \
                return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("bad slot %d", (int)(slot)));    \
            int32_t l_;                                                                       \
            rbs_handle* sh_ = shard_of(h, slot, &l_);                                         \
            const int32_t rc_ = call;                                                         \
            return rc_ ? gfail(h, sh_, rc_) : RBS_OK;                                         \
        }                                                                                     \
    } while (0)
#define RBS_GROUP_ALL(h, call)                                                                \
    do {                                                                                      \
        if (!(h)->shards.empty()) {                                                           \
            for (rbs_handle* sh_ : (h)->shards) {                                             \
                const int32_t rc_ = call;                                                     \
                if (rc_) return gfail(h, sh_, rc_);                                           \
            }                                                                                 \
            return RBS_OK;                                                                    \
        }                                                                                     \
    } while (0)
// A call that changes every shard's state: a failure on the first shard (argument errors land
// there) changed nothing; one on a later shard leaves the shards out of step -> poisoned.
#define RBS_GROUP_ALL_MUT(h, call)                                                            \
    do {                                                                                      \
        if (!(h)->shards.empty()) {                                                           \
            for (size_t k_ = 0; k_ < (h)->shards.size(); ++k_) {                              \
                rbs_handle* sh_ = (h)->shards[k_];                                            \
                const int32_t rc_ = call;                                                     \
                if (rc_) return k_ == 0 ? gfail(h, sh_, rc_) : poison(h, gfail(h, sh_, rc_)); \
            }                                                                                 \
            return RBS_OK;                                                                    \
        }                                                                                     \
    } while (0)
#define RBS_GROUP_FIRST(h, call)                                                              \
    do {                                                                                      \
        if (!(h)->shards.empty()) {                                                           \
            rbs_handle* sh_ = (h)->shards[0];                                                 \
            const int32_t rc_ = call;                                                         \
            return rc_ ? gfail(h, sh_, rc_) : RBS_OK;                                         \
        }                                                                                     \
    } while (0)

}  // namespace

extern "C" {

int32_t rbs_abi_version(void) { return RBS_ABI_VERSION; }

int32_t rbs_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* rbs_last_error(const rbs_handle* h)
{
    return h ? h->err.c_str() : g_create_error.c_str();
}

int32_t rbs_create(const rbs_config* cfg, rbs_handle** out)
{
    if (!out) { g_create_error = "rbs_create: out is NULL"; return RBS_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    if (!cfg) { g_create_error = "rbs_create: cfg is NULL"; return RBS_ERR_INVALID_ARGUMENT; }
    rbs_handle* h = new (std::nothrow) rbs_handle;
    if (!h) { g_create_error = "rbs_create: out of host memory"; return RBS_ERR_OUT_OF_MEMORY; }
    int32_t rc;
    const bool group = cfg->n_devices > 1 || (cfg->n_devices == 1 && cfg->device_ids && std::getenv("RBS_GROUP_SINGLE"));
    try {
        rc = group ? (cfg->abi_version != RBS_ABI_VERSION
                          ? fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("abi_version %d, library is %d", cfg->abi_version, RBS_ABI_VERSION))
                          : create_group(cfg, h))
                   : create_impl(cfg, h);
    } catch (const std::exception& e) {
        h->err = std::string("rbs_create: ") + e.what();
        rc = RBS_ERR_OUT_OF_MEMORY;
    }
    if (rc != RBS_OK) {
        g_create_error = h->err;
        if (group) release_group(h); else release(h);
        return rc;
    }
    *out = h;
    return RBS_OK;
}

void rbs_destroy(rbs_handle* h)
{
    if (h && !h->shards.empty()) release_group(h);
    else release(h);
}

int32_t rbs_reset(rbs_handle* h)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) {
        for (rbs_handle* sh_ : h->shards)
            if (int32_t rc_ = rbs_reset(sh_)) return gfail(h, sh_, rc_);
        for (rbs_handle* sh_ : h->shards) {   // idle streams: a clean slate for the group's call ordering
            RBS_HIP(h, hipSetDevice(sh_->device));
            RBS_HIP(h, hipEventRecord(sh_->ev_done, sh_->stream));
        }
        h->poisoned = false;
        h->frame_acquired = false;
        h->stp = false;
        h->stp_leave_pending = false;
        h->stp_now = rbs_handle::StpNow();
        h->stp_last_rebase = -1000000;
        h->stp_block_until = 0;
        h->calls = 0;
        return RBS_OK;
    }
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = drain(h, true)) return rc;   // no copy kernel may still be writing planes
    // (a call that failed between its kernels may have left the work-item counters of either parity
    // in use: both pairs start from zero again)
    RBS_HIP(h, hipMemsetAsync(h->d_ctr, 0, 8 * sizeof(int), h->stream));
    h->poisoned = false;
    h->frame_acquired = false;
    h->prefetched_slot = -1;   // (a frame uploaded ahead belongs to the session that ended)
    h->borrowed = nullptr;
    h->borrowed_f32 = nullptr;
    h->stp = false;            // (every plane is all background again: the scalar says it all)
    h->stp_request = -1;
    h->stp_last_rebase = -1000000;
    h->stp_block_until = 0;
    h->area_frac = 0.0;
    h->cur = 0;
    h->pending_frames = 0;
    h->update_clock = 0;
    h->background = (float)h->init_occ;
    {   // windowed: every plane is all background (empty window); dense: full windows for ever
        const int4 w0 = h->windowed ? make_int4(h->cols, h->rows, 0, 0) : make_int4(0, 0, h->cols, h->rows);
        const unsigned g = (unsigned)((h->max_particles + 255) / 256);
        for (int b = 0; b < 2; ++b)
            hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(g), dim3(256), 0, h->stream, h->d_win[b],
                               h->max_particles, w0);
    }
    if (h->slab_px) {
        const unsigned g = (unsigned)((h->max_particles + 255) / 256);
        for (int b = 0; b < 2; ++b)
            hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(g), dim3(256), 0, h->stream, h->d_reg[b],
                               h->max_particles, make_int4(h->cols, h->rows, 0, 0));
    }
    RBS_HIP(h, hipMemsetAsync(h->d_err, 0, 2 * sizeof(int), h->stream));
    h->h_err[0] = h->h_err[1] = 0;
    const size_t n = h->plane_stride * h->max_particles;
    hipLaunchKernelGGL(rbs::rbs_fill_kernel, dim3(2048), dim3(256), 0, h->stream, h->d_occ[0], n,
                       (float)h->init_occ);
    // the second buffer too: touches every page now instead of inside the first updating call
    hipLaunchKernelGGL(rbs::rbs_fill_kernel, dim3(2048), dim3(256), 0, h->stream, h->d_occ[1], n,
                       (float)h->init_occ);
    RBS_HIP(h, hipGetLastError());
    if (int32_t rc = drain(h, true)) return rc;
    return RBS_OK;
}

int32_t rbs_set_observation(rbs_handle* h, const double* depth, size_t n)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_REFUSE_POISONED(h);
    h->frame_acquired = false;   // (an acquired buffer that was never committed is abandoned)
    RBS_GROUP_ALL_MUT(h, rbs_set_observation(sh_, depth, n));
    if (!depth || n != (size_t)h->npx)
        return fail(h, RBS_ERR_INVALID_ARGUMENT,
                    fmt("set_observation: expected %d pixels, got %zu", h->npx, n));
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    if (int32_t rc = next_frame_staging(h)) return rc;
    if (int32_t rc = upload_frame(h, h->h_frame, nullptr, depth)) return rc;   // (converted and sent piece by piece)
    h->pending_frames += 1;
    return RBS_OK;
}

int32_t rbs_set_observation_borrowed(rbs_handle* h, const double* depth, size_t n)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    // handles that cannot stage between two kernels (several devices; float32 likelihood; whole planes) copy at once
    if (!h->shards.empty() || h->group || h->precision != RBS_PRECISION_F64 || !h->windowed || h->frame_ingest)
        return rbs_set_observation(h, depth, n);
    RBS_REFUSE_POISONED(h);
    h->frame_acquired = false;
    if (!depth || n != (size_t)h->npx)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("set_observation_borrowed: expected %d pixels, got %zu", h->npx, n));
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    h->prefetched_slot = -1;
    h->borrowed_f32 = nullptr;
    h->borrowed = depth;          // (a borrowed frame nobody evaluated is simply replaced)
    h->pending_frames += 1;
    return RBS_OK;
}

int32_t rbs_set_observation_borrowed_f32(rbs_handle* h, const float* depth, size_t n)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty() || h->group || h->precision != RBS_PRECISION_F64 || !h->windowed || h->frame_ingest)
        return rbs_set_observation_f32(h, depth, n);
    RBS_REFUSE_POISONED(h);
    h->frame_acquired = false;
    if (!depth || n != (size_t)h->npx)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("set_observation_borrowed_f32: expected %d pixels, got %zu", h->npx, n));
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    h->prefetched_slot = -1;
    h->borrowed = nullptr;
    h->borrowed_f32 = depth;
    h->pending_frames += 1;
    return RBS_OK;
}

int32_t rbs_set_observation_f32(rbs_handle* h, const float* depth, size_t n)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_REFUSE_POISONED(h);
    h->frame_acquired = false;   // (an acquired buffer that was never committed is abandoned)
    RBS_GROUP_ALL_MUT(h, rbs_set_observation_f32(sh_, depth, n));
    if (!depth || n != (size_t)h->npx)
        return fail(h, RBS_ERR_INVALID_ARGUMENT,
                    fmt("set_observation_f32: expected %d pixels, got %zu", h->npx, n));
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    if (int32_t rc = next_frame_staging(h)) return rc;
    if (int32_t rc = upload_frame(h, h->h_frame, depth)) return rc;   // (staged and sent piece by piece)
    h->pending_frames += 1;
    return RBS_OK;
}

int32_t rbs_acquire_frame_buffer(rbs_handle* h, float** buf)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!buf) return fail(h, RBS_ERR_INVALID_ARGUMENT, "acquire_frame_buffer: null pointer");
    RBS_REFUSE_POISONED(h);
    if (h->frame_acquired)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "acquire_frame_buffer: the buffer handed out last has not been committed (rbs_commit_frame_buffer)");
    if (!h->shards.empty()) {
        // one staging buffer (shard 0's) feeds every device: it is free once every shard's upload
        // of the frame before last has finished
        rbs_handle* s0 = h->shards[0];
        s0->frame_slot ^= 1;
        s0->h_frame = s0->h_frames[s0->frame_slot];
        for (rbs_handle* sh : h->shards) {
            RBS_HIP(h, hipSetDevice(sh->device));
            sh->frame_slot = s0->frame_slot;
            RBS_HIP(h, hipEventSynchronize(sh->ev_frame[sh->frame_slot]));
        }
        *buf = s0->h_frame;
        h->frame_acquired = true;
        return RBS_OK;
    }
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = next_frame_staging(h)) return rc;
    *buf = h->h_frame;
    h->frame_acquired = true;
    return RBS_OK;
}

int32_t rbs_commit_frame_buffer(rbs_handle* h)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_REFUSE_POISONED(h);
    // a commit uploads the buffer the matching acquire handed out: without one (or after another
    // rbs_set_observation* call took the acquire's place) there is nothing defined to upload, and the
    // upload would land in the image the kernels in flight are reading
    if (!h->frame_acquired)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "commit_frame_buffer: no buffer acquired (rbs_acquire_frame_buffer comes first, once per frame)");
    h->frame_acquired = false;
    if (!h->shards.empty()) {
        const float* src = h->shards[0]->h_frame;
        for (rbs_handle* sh : h->shards) {
            if (hipSetDevice(sh->device) != hipSuccess) return poison(h, fail(h, RBS_ERR_HIP, fmt("hipSetDevice(%d) failed", sh->device)));
            if (int32_t rc = flush_lazy_frame(sh, sh->stream)) return poison(h, gfail(h, sh, rc));
            if (int32_t rc = upload_frame(sh, src)) return poison(h, gfail(h, sh, rc));
            sh->pending_frames += 1;
        }
        return RBS_OK;
    }
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    if (int32_t rc = upload_frame(h, h->h_frame)) return rc;
    h->pending_frames += 1;
    return RBS_OK;
}

int32_t rbs_set_observation_native_f32(rbs_handle* h, const float* native, int32_t width,
                                       int32_t height, int32_t f)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_REFUSE_POISONED(h);
    h->frame_acquired = false;
    RBS_GROUP_ALL_MUT(h, rbs_set_observation_native_f32(sh_, native, width, height, f));
    if (!native || f <= 0 || width <= 0 || height <= 0 || height / f != h->rows || width / f != h->cols)
        return fail(h, RBS_ERR_INVALID_ARGUMENT,
                    fmt("set_observation_native: %dx%d / %d does not give the evaluated %dx%d", width,
                        height, f, h->cols, h->rows));
    if (f == 1) return rbs_set_observation_f32(h, native, (size_t)h->npx);
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    if (int32_t rc = next_frame_staging(h)) return rc;
    // The reference's rule (R:source/dbot_ros/util/ros_interface.h:152-168): evaluated(r, c) =
    // native(r f, c f) -- applied while the frame is staged, so that only the rows*cols values the
    // sensor evaluates are copied and sent (1/f^2 of the driver's frame: 19 KB of 1.2 MB at the
    // reference's default factor 8) and the frame then travels like any other host frame.
    for (int r = 0; r < h->rows; ++r) {
        const float* src = native + (size_t)r * f * width;
        float* dst = h->h_frame + (size_t)r * h->cols;
        for (int c = 0; c < h->cols; ++c) dst[c] = src[(size_t)c * f];
    }
    if (int32_t rc = upload_frame(h, h->h_frame)) return rc;
    h->pending_frames += 1;
    return RBS_OK;
}

int32_t rbs_set_observation_device(rbs_handle* h, const float* d_depth, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_REFUSE_POISONED(h);
    h->frame_acquired = false;
    if (!d_depth) return fail(h, RBS_ERR_INVALID_ARGUMENT, "set_observation_device: null pointer");
    if (!h->shards.empty()) return group_set_observation_device(h, d_depth, static_cast<hipStream_t>(stream));
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    // the ingest kernel (copy into the handle's buffer + per-pixel model terms) is not launched
    // here: it shares a launch with the next rbs_loglikes* call's rectangles kernel when that
    // call comes on the same stream, and is launched on `s` by whatever needs the frame otherwise
    if (int32_t rc = flush_lazy_frame(h, s)) return rc;   // an earlier frame nobody evaluated
    if (int32_t rc = release_frame_slot(h)) return rc;
    h->prefetched_slot = -1;   // (a frame uploaded ahead of its turn that another frame overtakes is abandoned: ADVICE r4)
    h->borrowed = nullptr;
    h->borrowed_f32 = nullptr;
    h->lazy_frame = d_depth;
    h->lazy_stream = s;
    h->pending_frames += 1;
    return RBS_OK;
}

int32_t rbs_get_observation(rbs_handle* h, float* out)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_FIRST(h, rbs_get_observation(sh_, out));
    if (!out) return fail(h, RBS_ERR_INVALID_ARGUMENT, "get_observation: null pointer");
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = stage_borrowed(h)) return rc;
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    RBS_HIP(h, hipStreamSynchronize(h->stream));
    if (h->frame_wait >= 0) RBS_HIP(h, hipEventSynchronize(h->ev_frame[h->frame_wait]));
    RBS_HIP(h, hipMemcpy(out, h->cur_frame, sizeof(float) * h->npx, hipMemcpyDeviceToHost));
    return RBS_OK;
}

static int32_t loglikes_impl(rbs_handle* h, const double* poses, int32_t* indices, int32_t n, int32_t update, double* out_loglik,
                             const float* next_depth, const DeltaArgs* da = nullptr);

int32_t rbs_loglikes(rbs_handle* h, const double* poses, int32_t* indices, int32_t n,
                     int32_t update, double* out_loglik)
{
    return loglikes_impl(h, poses, indices, n, update, out_loglik, nullptr);
}

int32_t rbs_loglikes_deltas(rbs_handle* h, const double* deltas, const double* default_poses, int32_t body_stride,
                            int32_t* indices, int32_t n, int32_t update, double* out_loglik)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (body_stride < 6) return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("loglikes_deltas: body_stride = %d, need >= 6 (position, rotation vector)", body_stride));
    if (n > 0 && (!deltas || !default_poses)) return fail(h, RBS_ERR_INVALID_ARGUMENT, "loglikes_deltas: null pointer");
    const DeltaArgs da = {deltas, default_poses, body_stride};
    return loglikes_impl(h, deltas, indices, n, update, out_loglik, nullptr, &da);
}

int32_t rbs_deltas_buffer(rbs_handle* h, double** buf)
{
    if (!h || !buf) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) return fail(h, RBS_ERR_UNSUPPORTED, "deltas_buffer: single-device handles");
    *buf = reinterpret_cast<double*>(h->h_in);
    return RBS_OK;
}

int32_t rbs_get_poses(rbs_handle* h, double* out, int32_t n)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) return fail(h, RBS_ERR_UNSUPPORTED, "get_poses: single-device handles");
    if (!out || n < 0 || n > h->max_particles) return fail(h, RBS_ERR_INVALID_ARGUMENT, "get_poses: bad arguments");
    RBS_HIP(h, hipSetDevice(h->device));
    RBS_HIP(h, hipStreamSynchronize(h->stream));
    RBS_HIP(h, hipMemcpy(out, h->d_in, sizeof(double) * 12 * h->n_bodies * (size_t)n, hipMemcpyDeviceToHost));
    return RBS_OK;
}

int32_t rbs_loglikes_prefetch(rbs_handle* h, const double* poses, int32_t* indices, int32_t n, int32_t update, double* out_loglik,
                              const float* next_depth, size_t next_n)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) return fail(h, RBS_ERR_UNSUPPORTED, "loglikes_prefetch: single-device handles (a handle over several devices: rbs_acquire/commit_frame_buffer)");
    if (!next_depth || next_n != (size_t)h->npx)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("loglikes_prefetch: expected a next frame of %d pixels, got %zu", h->npx, next_n));
    if (n <= 0) return fail(h, RBS_ERR_INVALID_ARGUMENT, "loglikes_prefetch: n must be positive");
    return loglikes_impl(h, poses, indices, n, update, out_loglik, next_depth);
}

int32_t rbs_set_observation_prefetched(rbs_handle* h)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_REFUSE_POISONED(h);
    if (h->prefetched_slot < 0)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "set_observation_prefetched: no frame was uploaded ahead (rbs_loglikes_prefetch), or another frame has been set since");
    RBS_HIP(h, hipSetDevice(h->device));
    h->frame_acquired = false;
    if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
    const int k = h->prefetched_slot;
    h->prefetched_slot = -1;
    if (int32_t rc = release_frame_slot(h)) return rc;   // the readers of the frame it replaces are on the launch stream by now
    h->cur_frame = h->d_fin[k];
    h->cur_aux = h->d_aux ? h->d_aux_slot[k] : nullptr;
    h->cur_slot = k;
    h->frame_wait = k;
    h->pending_frames += 1;
    return RBS_OK;
}

static int32_t loglikes_impl(rbs_handle* h, const double* poses, int32_t* indices, int32_t n, int32_t update, double* out_loglik,
                             const float* next_depth, const DeltaArgs* da)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    const BorrowedFrameGuard borrowed_guard{h};   // (a frame borrowed for this call is not the library's any longer once it returns)
    RBS_REFUSE_POISONED(h);
    if (n < 0 || n > h->max_particles)
        return fail(h, RBS_ERR_INVALID_ARGUMENT,
                    fmt("loglikes: n = %d outside 0..max_particles = %d", n, h->max_particles));
    if (n == 0) return RBS_OK;
    if (!poses || !indices || !out_loglik)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "loglikes: null pointer");
    if (!h->shards.empty()) return group_loglikes(h, poses, indices, n, update, out_loglik, da);
    const int32_t slots_total = h->peer_world > 1 ? h->peer_world * h->max_particles : h->max_particles;   // (attached: parents are global slots)
    for (int32_t i = 0; i < n; ++i)
        if (indices[i] < 0 || indices[i] >= slots_total)
            return fail(h, RBS_ERR_INVALID_ARGUMENT,
                        fmt("loglikes: indices[%d] = %d outside 0..%d", i, indices[i],
                            slots_total - 1));
    RBS_HIP(h, hipSetDevice(h->device));
    // the call waits for the log-likelihoods only -- the occlusion planes are finished by the second
    // stream and joined by the next call
    const CallState before = save_call_state(h);
    if (next_depth) {   // refused before anything is enqueued: the planes, the indices and the staging images stay as they are
        if (int32_t rc = stage_borrowed(h)) return rc;   // (a borrowed frame and a look-ahead frame: the borrowed one is staged first)
        if (int32_t rc = prefetch_check(h)) return rc;
    }
    if (int32_t rc = host_call(h, poses, indices, n, update != 0, da)) return rc;
    // the next frame travels while this call's kernels run; should the upload itself fail (a runtime error), the
    // likelihood call is still completed -- results copied, indices rewritten -- and the error reported after it
    const int32_t prefetch_rc = next_depth ? prefetch_frame(h, next_depth) : RBS_OK;
    RBS_HIP(h, hipEventSynchronize(h->ev_out));
    const bool stale_overflow = h->slab_px && h->h_err[2] != 0;   // of an earlier asynchronous call: reported below, once
    if (h->slab_px && h->h_err[0]) {
        // a region did not fit its slab: the planes this call read are intact (double buffer), so the
        // call is taken back, the slabs are enlarged to hold the largest region asked for, and the
        // call runs again -- the caller sees the numbers of whole planes, later
        if (int32_t rc = drain(h, true)) return rc;
        restore_call_state(h, before);
        if (int32_t rc = slab_housekeeping(h)) return rc;
        if (int32_t rc = host_call(h, poses, indices, n, update != 0, da)) return rc;
        RBS_HIP(h, hipEventSynchronize(h->ev_out));
        if (h->h_err[0]) return check_slab_error(h);   // (cannot happen: a region is never larger than the frame)
    }
    std::memcpy(out_loglik, h->h_out, sizeof(double) * (size_t)n);
    if (update) {
        const int32_t first = h->peer_world > 1 ? h->peer_rank * h->max_particles : 0;
        for (int32_t i = 0; i < n; ++i) indices[i] = first + i;
    }
    if (int32_t rc = slab_housekeeping(h)) return rc;
    h->quiet = !h->async_outstanding;   // (everything on the handle's stream has run; the copy kernel's last blocks are short)
    if (stale_overflow) {   // this call's results are good; an earlier rbs_loglikes_device's were not
        h->h_err[0] = 1;
        const int32_t rc = check_slab_error(h);
        h->h_err[0] = 0;
        return rc;
    }
    return prefetch_rc;   // (its message is rbs_last_error's: nothing above overwrote it on the way here)
}

int32_t rbs_loglikes_device(rbs_handle* h, const double* d_poses, const int32_t* d_indices,
                            int32_t n, int32_t update, double* d_out_loglik, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_REFUSE_POISONED(h);
    if (n < 0 || n > h->max_particles)
        return fail(h, RBS_ERR_INVALID_ARGUMENT,
                    fmt("loglikes_device: n = %d outside 0..max_particles = %d", n, h->max_particles));
    if (n == 0) return RBS_OK;
    if (!d_poses || !d_indices || !d_out_loglik)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "loglikes_device: null pointer");
    if (!h->shards.empty()) return group_loglikes_device(h, d_poses, d_indices, n, update, d_out_loglik, static_cast<hipStream_t>(stream));
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (int32_t rc = stage_borrowed(h)) return rc;   // (only the host-pointer calls stage a borrowed frame behind their own kernels)
    if (h->slab_auto && !h->slab_probed && update && h->peer_world <= 1) {
        // slabs the LIBRARY chose (state_slab_px = 0, more than 8 192 particles): an asynchronous call cannot be taken
        // back, so the first one is preceded by a look at the regions it will store -- one small kernel and ONE host
        // synchronisation in the handle's lifetime -- and the slabs are enlarged first if they would be three
        // quarters full (a default-config handle must not answer its first call with NaNs)
        h->slab_probed = true;
        DevParams P = h->base;
        P.poses = d_poses; P.indices = d_indices; P.n = n;
        P.slots = h->max_particles; P.n_dev = 1; P.shard_cap = h->max_particles;
        P.win_src = h->d_win[h->cur];
        P.rect_align = h->windowed ? h->rect_align : rbs::kRectAlign;
        if (int32_t rc = drain(h, false)) return rc;
        int* d_max = h->d_err + 1;   // (the sticky maximum the rectangles kernel keeps: the probe only raises it early)
        hipLaunchKernelGGL(rbs::rbs_region_probe_kernel, dim3((unsigned)((n + rbs::kPrepPerBlock - 1) / rbs::kPrepPerBlock)),
                           dim3(64 * rbs::kPrepPerBlock), 0, s, P, d_max);
        RBS_HIP(h, hipGetLastError());
        RBS_HIP(h, hipMemcpyAsync(h->h_err, h->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        RBS_HIP(h, hipStreamSynchronize(s));
        if (int32_t rc = slab_housekeeping(h)) return rc;
    }
    h->async_outstanding = true;
    return enqueue_loglikes(h, d_poses, d_indices, n, update != 0, d_out_loglik, s);
}

int32_t rbs_synchronize(rbs_handle* h)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) {
        int32_t first = RBS_OK;
        for (rbs_handle* sh_ : h->shards) {
            RBS_HIP(h, hipSetDevice(sh_->device));
            if (int32_t rc = drain(sh_, true)) return gfail(h, sh_, rc);
            if (sh_->slab_px) RBS_HIP(h, hipMemcpy(sh_->h_err, sh_->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost));
            if (first == RBS_OK && check_slab_error(sh_) != RBS_OK) first = gfail(h, sh_, RBS_ERR_OUT_OF_MEMORY);
        }
        if (int32_t rc = group_grow_slabs(h, nullptr)) return rc;
        return first;
    }
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = stage_borrowed(h)) return rc;   // (the caller's frame is released when this returns)
    if (int32_t rc = drain(h, true)) return rc;
    if (h->slab_px) RBS_HIP(h, hipMemcpy(h->h_err, h->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost));
    // an asynchronous call whose region did not fit is reported here, ONCE; the slabs are enlarged
    // either way so that the calls that follow fit
    const int32_t rc = check_slab_error(h);
    const std::string msg = h->err;
    if (int32_t rc2 = slab_housekeeping(h)) return rc2;
    if (rc) h->err = msg;
    h->async_outstanding = false;
    h->quiet = true;
    return rc;
}

int32_t rbs_get_occlusion(rbs_handle* h, int32_t slot, float* out)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_get_occlusion(sh_, l_, out));
    if (slot < 0 || slot >= h->max_particles || !out)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("get_occlusion: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = drain(h, true)) return rc;
    if (h->slab_px || h->exact) {
        if (int32_t rc = h->exact ? exact_expand(h, slot, h->d_render, h->stream) : slab_expand(h, slot, h->d_render, h->stream)) return rc;
        RBS_HIP(h, hipStreamSynchronize(h->stream));
        RBS_HIP(h, hipMemcpy(out, h->d_render, sizeof(float) * h->npx, hipMemcpyDeviceToHost));
        return RBS_OK;
    }
    if (int32_t rc = materialize(h, slot, h->stream)) return rc;
    RBS_HIP(h, hipStreamSynchronize(h->stream));
    RBS_HIP(h, hipMemcpy(out, h->d_occ[h->cur] + (size_t)slot * h->npx, sizeof(float) * h->npx,
                         hipMemcpyDeviceToHost));
    return RBS_OK;
}

int32_t rbs_set_occlusion(rbs_handle* h, int32_t slot, const float* plane)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_set_occlusion(sh_, l_, plane));
    if (slot < 0 || slot >= h->max_particles || !plane)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("set_occlusion: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = drain(h, true)) return rc;
    if (h->slab_px || h->exact) {
        RBS_HIP(h, hipMemcpy(h->d_render, plane, sizeof(float) * h->npx, hipMemcpyHostToDevice));
        return h->exact ? exact_store(h, slot, h->d_render, h->stream) : slab_store(h, slot, h->d_render, h->stream);
    }
    RBS_HIP(h, hipMemcpy(h->d_occ[h->cur] + (size_t)slot * h->npx, plane, sizeof(float) * h->npx,
                         hipMemcpyHostToDevice));
    if (h->windowed) {   // the whole plane is explicit now; the next updating call tightens it again
        hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, h->stream, h->d_win[h->cur] + slot,
                           1, make_int4(0, 0, h->cols, h->rows));
        RBS_HIP(h, hipGetLastError());
        RBS_HIP(h, hipStreamSynchronize(h->stream));
    }
    return RBS_OK;
}

int32_t rbs_occlusion_device_ptr(rbs_handle* h, int32_t slot, void** out)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_occlusion_device_ptr(sh_, l_, out));
    if (slot < 0 || slot >= h->max_particles || !out)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("occlusion_device_ptr: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    if (h->exact) return fail(h, RBS_ERR_UNSUPPORTED, "occlusion_device_ptr: a slot of stamped planes (occlusion_mode REFERENCE) is not a float plane");
    if (int32_t rc = drain(h, true)) return rc;   // planes complete before the caller touches them
    if (int32_t rc = materialize(h, slot, h->stream)) return rc;   // and dense, whatever the caller does next
    RBS_HIP(h, hipStreamSynchronize(h->stream));
    *out = h->d_occ[h->cur] + (size_t)slot * h->npx;
    return RBS_OK;
}

int32_t rbs_occlusion_next_device_ptr(rbs_handle* h, int32_t slot, void** out)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_occlusion_next_device_ptr(sh_, l_, out));
    if (slot < 0 || slot >= h->max_particles || !out)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("occlusion_next_device_ptr: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    if (h->slab_px) return fail(h, RBS_ERR_UNSUPPORTED, "occlusion_next_device_ptr: slots are slabs, not planes (state_slab_px)");
    if (h->exact) return fail(h, RBS_ERR_UNSUPPORTED, "occlusion_next_device_ptr: a slot of stamped planes (occlusion_mode REFERENCE) is not a float plane");
    if (int32_t rc = drain(h, true)) return rc;   // planes complete before the caller touches them
    *out = h->d_occ[1 - h->cur] + (size_t)slot * h->npx;
    return RBS_OK;
}

int32_t rbs_export_plane(rbs_handle* h, int32_t slot, void* d_dst, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_export_plane(sh_, l_, d_dst, stream));
    if (slot < 0 || slot >= h->max_particles || !d_dst)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("export_plane: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (h->join_pending >= 0) RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
    if (h->exact) return exact_expand(h, slot, static_cast<float*>(d_dst), s);
    if (h->slab_px) return slab_expand(h, slot, static_cast<float*>(d_dst), s);
    if (int32_t rc = materialize(h, slot, s)) return rc;
    RBS_HIP(h, hipMemcpyAsync(d_dst, h->d_occ[h->cur] + (size_t)slot * h->npx, sizeof(float) * h->npx,
                              hipMemcpyDeviceToDevice, s));
    return RBS_OK;
}

int32_t rbs_import_plane(rbs_handle* h, int32_t slot, const void* d_src, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_import_plane(sh_, l_, d_src, stream));
    if (slot < 0 || slot >= h->max_particles || !d_src)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("import_plane: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (h->join_pending >= 0) RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
    if (h->exact) return exact_store(h, slot, static_cast<const float*>(d_src), s);   // (synchronises)
    if (h->slab_px) return slab_store(h, slot, static_cast<const float*>(d_src), s);   // (synchronises: the box must fit)
    RBS_HIP(h, hipMemcpyAsync(h->d_occ[h->cur] + (size_t)slot * h->npx, d_src, sizeof(float) * h->npx,
                              hipMemcpyDeviceToDevice, s));
    if (h->windowed) {
        hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_win[h->cur] + slot, 1,
                           make_int4(0, 0, h->cols, h->rows));
        RBS_HIP(h, hipGetLastError());
    }
    return RBS_OK;
}

int32_t rbs_stream_join(rbs_handle* h, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) {
        for (rbs_handle* sh_ : h->shards)
            if (int32_t rc = rbs_stream_join(sh_, stream)) return gfail(h, sh_, rc);
        return RBS_OK;
    }
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (h->join_pending >= 0) RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
    return RBS_OK;
}

// A slot's window as the host needs it for the window-sized transport (one 16-byte read-back).
static int32_t window_of(rbs_handle* h, int slot, hipStream_t s, int box[4])
{
    if (!h->windowed) { box[0] = 0; box[1] = 0; box[2] = h->cols; box[3] = h->rows; return RBS_OK; }
    RBS_HIP(h, hipMemcpyAsync(box, h->d_win[h->cur] + slot, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    RBS_HIP(h, hipStreamSynchronize(s));
    if (box[2] <= box[0] || box[3] <= box[1]) { box[0] = h->cols; box[1] = h->rows; box[2] = 0; box[3] = 0; }
    return RBS_OK;
}

int32_t rbs_export_window(rbs_handle* h, int32_t slot, int32_t rect_out[4], void* d_payload, size_t capacity_floats, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_export_window(sh_, l_, rect_out, d_payload, capacity_floats, stream));
    if (slot < 0 || slot >= h->max_particles || !rect_out)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("export_window: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (h->join_pending >= 0) RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
    if (h->exact) {
        // stamped planes travel as EFFECTIVE values (what the receiver's rbs_import_window takes "as of now"): the slot's plane is
        // assembled beside the slots and its window cut out of it (shared trail: the whole plane, as below)
        if (int32_t rc = exact_expand(h, slot, h->d_render, s)) return rc;
        int box[4] = {0, 0, h->cols, h->rows};
        if (!h->stp) if (int32_t rc = window_of(h, slot, s, box)) return rc;
        for (int k = 0; k < 4; ++k) rect_out[k] = box[k];
        const int w = box[2] - box[0], hh = box[3] - box[1];
        if (w <= 0 || hh <= 0) return RBS_OK;
        if ((size_t)w * (size_t)hh > capacity_floats || !d_payload)
            return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("export_window: the window of slot %d holds %d x %d values, the buffer %zu", slot, w, hh, capacity_floats));
        RBS_HIP(h, hipMemcpy2DAsync(d_payload, sizeof(float) * (size_t)w, h->d_render + (size_t)box[1] * h->cols + box[0], sizeof(float) * (size_t)h->cols,
                                    sizeof(float) * (size_t)w, (size_t)hh, hipMemcpyDeviceToDevice, s));
        return RBS_OK;
    }
    // (shared trail: outside its window the plane is the handle's background PLANE, which the receiver does not have -- the
    // slot is made dense first and travels whole)
    if (h->stp && h->slab_px) {   // (a slab cannot be made dense in place: the whole plane is assembled beside it and travels from there)
        for (int k = 0; k < 4; ++k) rect_out[k] = k < 2 ? 0 : (k == 2 ? h->cols : h->rows);
        if ((size_t)h->npx > capacity_floats || !d_payload)
            return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("export_window: the plane of slot %d travels whole (%d values), the buffer holds %zu", slot, h->npx, capacity_floats));
        return slab_expand(h, slot, static_cast<float*>(d_payload), s);
    }
    if (h->stp) if (int32_t rc = materialize(h, slot, s)) return rc;
    int box[4];
    if (int32_t rc = window_of(h, slot, s, box)) return rc;
    for (int k = 0; k < 4; ++k) rect_out[k] = box[k];
    const int w = box[2] - box[0], hh = box[3] - box[1];
    if (w <= 0 || hh <= 0) return RBS_OK;   // all background: nothing to carry
    if ((size_t)w * (size_t)hh > capacity_floats || !d_payload)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("export_window: the window of slot %d holds %d x %d values, the buffer %zu", slot, w, hh, capacity_floats));
    // where the window's first value is stored, and the stored row length
    const float* base = h->d_occ[h->cur] + (size_t)slot * h->plane_stride;
    size_t src_pitch = sizeof(float) * (size_t)h->cols;
    if (h->slab_px) {
        int reg[4];
        RBS_HIP(h, hipMemcpyAsync(reg, h->d_reg[h->cur] + slot, sizeof(reg), hipMemcpyDeviceToHost, s));
        RBS_HIP(h, hipStreamSynchronize(s));
        src_pitch = sizeof(float) * (size_t)(reg[2] - reg[0]);
        base += (size_t)(box[1] - reg[1]) * (size_t)(reg[2] - reg[0]) + (size_t)(box[0] - reg[0]);
    } else {
        base += (size_t)box[1] * h->cols + box[0];
    }
    RBS_HIP(h, hipMemcpy2DAsync(d_payload, sizeof(float) * (size_t)w, base, src_pitch, sizeof(float) * (size_t)w, (size_t)hh,
                                hipMemcpyDeviceToDevice, s));
    return RBS_OK;
}

int32_t rbs_import_window(rbs_handle* h, int32_t slot, const int32_t rect[4], const void* d_payload, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_import_window(sh_, l_, rect, d_payload, stream));
    if (slot < 0 || slot >= h->max_particles || !rect)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("import_window: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (h->join_pending >= 0) RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
    int4 r = make_int4(rect[0], rect[1], rect[2], rect[3]);
    const bool empty = r.z <= r.x || r.w <= r.y;
    if (empty) r = make_int4(h->cols, h->rows, 0, 0);
    else if (r.x < 0 || r.y < 0 || r.z > h->cols || r.w > h->rows || (h->windowed && ((r.x | r.z) & 3)) || !d_payload)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("import_window: bad rectangle (%d, %d, %d, %d)", rect[0], rect[1], rect[2], rect[3]));
    const int w = r.z - r.x, hh = r.w - r.y;
    float* dst = h->d_occ[h->cur] + (size_t)slot * h->plane_stride;
    if ((h->stp && h->slab_px) || h->exact) {
        // the sender's plane = its scalar background outside rect: assembled whole beside the slabs, stored like any plane handed in
        hipLaunchKernelGGL(rbs::rbs_fill_kernel, dim3(256), dim3(256), 0, s, h->d_render, (size_t)h->npx, h->background);
        RBS_HIP(h, hipGetLastError());
        if (!empty)
            RBS_HIP(h, hipMemcpy2DAsync(h->d_render + (size_t)r.y * h->cols + r.x, sizeof(float) * (size_t)h->cols, d_payload, sizeof(float) * (size_t)w,
                                        sizeof(float) * (size_t)w, (size_t)hh, hipMemcpyDeviceToDevice, s));
        return h->exact ? exact_store(h, slot, h->d_render, s) : slab_store(h, slot, h->d_render, s);
    }
    if (!h->windowed || h->stp) {
        // whole planes without windows (or a handle whose implicit background is a plane of its own): the sender's scalar
        // background has to be written out
        hipLaunchKernelGGL(rbs::rbs_fill_kernel, dim3(256), dim3(256), 0, s, dst, (size_t)h->npx, h->background);
        RBS_HIP(h, hipGetLastError());
        if (!empty)
            RBS_HIP(h, hipMemcpy2DAsync(dst + (size_t)r.y * h->cols + r.x, sizeof(float) * (size_t)h->cols, d_payload, sizeof(float) * (size_t)w,
                                        sizeof(float) * (size_t)w, (size_t)hh, hipMemcpyDeviceToDevice, s));
        if (h->stp) {
            hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_win[h->cur] + slot, 1, make_int4(0, 0, h->cols, h->rows));
            RBS_HIP(h, hipGetLastError());
        }
        return RBS_OK;
    }
    if (h->slab_px) {
        if (!empty && (long)w * hh > (long)h->slab_px) {
            // a window handed in from a handle whose slabs have already grown (every rank enlarges its slabs on its own
            // schedule: ADVICE r4): the slabs grow here as they do behind rbs_import_plane / rbs_set_occlusion (slab_store)
            // -- on a handle of its own; the shards of a group and attached ranks keep one size among them
            if (h->group || h->peer_world > 1)
                return fail(h, RBS_ERR_OUT_OF_MEMORY, fmt("import_window: a window of %d x %d values does not fit a slab of %d px (state_slab_px)", w, hh, h->slab_px));
            if (int32_t rc = drain(h, true)) return rc;
            RBS_HIP(h, hipStreamSynchronize(s));   // (earlier imports on the caller's stream wrote into the buffers about to be replaced)
            if (int32_t rc = grow_slabs(h, slab_for(h, (int)std::min<long>((long)w * hh, h->npx)))) return rc;
            if ((long)w * hh > (long)h->slab_px)
                return fail(h, RBS_ERR_OUT_OF_MEMORY, fmt("import_window: a window of %d x %d values does not fit a slab of %d px (state_slab_px)", w, hh, h->slab_px));
            dst = h->d_occ[h->cur] + (size_t)slot * h->plane_stride;
        }
        if (!empty)   // the slot's stored region becomes the window itself, packed
            RBS_HIP(h, hipMemcpyAsync(dst, d_payload, sizeof(float) * (size_t)w * (size_t)hh, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_reg[h->cur] + slot, 1, r);
    } else if (!empty) {
        RBS_HIP(h, hipMemcpy2DAsync(dst + (size_t)r.y * h->cols + r.x, sizeof(float) * (size_t)h->cols, d_payload, sizeof(float) * (size_t)w,
                                    sizeof(float) * (size_t)w, (size_t)hh, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(rbs::rbs_set_window_kernel, dim3(1), dim3(64), 0, s, h->d_win[h->cur] + slot, 1, r);
    RBS_HIP(h, hipGetLastError());
    return RBS_OK;
}

// ---- one process per GPU: the handles of the other ranks, mapped (include/rbsensor_mi355x.h) ----
namespace {
struct IpcBlob {
    uint32_t magic;
    int32_t device, max_particles, rows, cols, slab_px, windowed, cur, exact;
    int64_t plane_stride;
    hipIpcMemHandle_t mem[6];   // occ[0], occ[1], win[0], win[1], reg[0], reg[1] (reg: slabs only)
};
static_assert(sizeof(IpcBlob) <= RBS_IPC_BLOB_BYTES, "RBS_IPC_BLOB_BYTES");
constexpr uint32_t kIpcMagic = 0x52425331u;
}

int32_t rbs_ipc_export(rbs_handle* h, void* blob_out)
{
    if (!h || !blob_out) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) return fail(h, RBS_ERR_UNSUPPORTED, "ipc_export: a handle over several devices already shares its planes in-process");
    RBS_HIP(h, hipSetDevice(h->device));
    if (h->stp) return fail(h, RBS_ERR_UNSUPPORTED, "ipc_export: the handle already stores its planes against a shared background plane of its own "
                                                    "(windows grew past the shared-trail threshold): rbs_reset first, then export");
    h->ipc_exported = true;   // (other ranks read these planes in place: from now on the shared trail is entered / re-based / left only when the
                              //  caller says so, on every rank alike -- rbs_shared_trail_rebase)
    if (int32_t rc = drain(h, true)) return rc;
    std::memset(blob_out, 0, RBS_IPC_BLOB_BYTES);
    IpcBlob b{};
    b.magic = kIpcMagic;
    b.device = h->device; b.max_particles = h->max_particles; b.rows = h->rows; b.cols = h->cols;
    b.slab_px = h->slab_px; b.windowed = h->windowed ? 1 : 0; b.cur = h->cur; b.plane_stride = (int64_t)h->plane_stride; b.exact = h->exact ? 1 : 0;
    void* bufs[6] = {h->d_occ[0], h->d_occ[1], h->d_win[0], h->d_win[1], h->d_reg[0], h->d_reg[1]};
    for (int k = 0; k < 6; ++k)
        if (bufs[k]) RBS_HIP(h, hipIpcGetMemHandle(&b.mem[k], bufs[k]));
    std::memcpy(blob_out, &b, sizeof(b));
    return RBS_OK;
}

int32_t rbs_ipc_attach(rbs_handle* h, int32_t rank, int32_t world, const void* blobs)
{
    if (!h || !blobs) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) return fail(h, RBS_ERR_UNSUPPORTED, "ipc_attach: a handle over several devices already shares its planes in-process");
    if (world < 1 || world > rbs::kMaxDevices || rank < 0 || rank >= world)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("ipc_attach: rank %d of %d (at most %d ranks)", rank, world, rbs::kMaxDevices));
    if (h->peer_world > 1) return fail(h, RBS_ERR_INVALID_ARGUMENT, "ipc_attach: already attached");
    if (h->stp) return fail(h, RBS_ERR_UNSUPPORTED, "ipc_attach: the handle stores its planes against a shared background plane already: rbs_reset first");
#ifdef RBS_TEST_HOOKS   // (librbsensor_mi355x_hooks.so only) RBS_TEST_ATTACH_HANG=1: the call never returns -- what hipIpcOpenMemHandle was
                        // seen to do for some buffer sizes; tests/test_gpu_fullsize.py checks that bench.py --gpus N still prints its line
    if (const char* e = std::getenv("RBS_TEST_ATTACH_HANG"))
        if (std::atoi(e) != 0) for (;;) ::usleep(1000000);
#endif
    if ((long)world * h->max_particles > 0x7fffffffL) return fail(h, RBS_ERR_INVALID_ARGUMENT, "ipc_attach: too many global slots");
    RBS_HIP(h, hipSetDevice(h->device));
    if (std::getenv("RBS_DEBUG_IPC")) std::fprintf(stderr, "[rbs ipc] rank %d attach: draining\n", rank);
    if (int32_t rc = drain(h, true)) return rc;
    const unsigned char* raw = static_cast<const unsigned char*>(blobs);
    for (int k = 0; k < world; ++k) {
        IpcBlob b;
        std::memcpy(&b, raw + (size_t)k * RBS_IPC_BLOB_BYTES, sizeof(b));
        if (b.magic != kIpcMagic) return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("ipc_attach: blob %d is not an rbs_ipc_export", k));
        if (b.max_particles != h->max_particles || b.rows != h->rows || b.cols != h->cols || b.slab_px != h->slab_px ||
            b.windowed != (h->windowed ? 1 : 0) || b.plane_stride != (int64_t)h->plane_stride || b.cur != h->cur || b.exact != (h->exact ? 1 : 0))
            return fail(h, RBS_ERR_INVALID_ARGUMENT,
                        fmt("ipc_attach: rank %d's handle differs (max_particles %d, %d x %d, slab %d px, %s planes, buffer %d) from this one "
                            "(%d, %d x %d, %d, %s, %d): every rank must create the same handle and be at the same point of its call sequence",
                            k, b.max_particles, b.cols, b.rows, b.slab_px, b.windowed ? "windowed" : "whole", b.cur, h->max_particles, h->cols,
                            h->rows, h->slab_px, h->windowed ? "windowed" : "whole", h->cur));
        if (k == rank) continue;
        if (b.device != h->device) {   // another GPU of the node: read over xGMI
            int can = 0;
            RBS_HIP(h, hipDeviceCanAccessPeer(&can, h->device, b.device));
            if (!can) return fail(h, RBS_ERR_UNSUPPORTED, fmt("ipc_attach: device %d cannot access device %d's memory (peer access)", h->device, b.device));
            const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                (void)hipGetLastError();
                return fail(h, RBS_ERR_HIP, fmt("hipDeviceEnablePeerAccess(%d -> %d): %s", h->device, b.device, hipGetErrorString(e)));
            }
            (void)hipGetLastError();
        }
        for (int m = 0; m < 6; ++m) {
            if (m >= 4 && !h->slab_px) break;
            void* p = nullptr;
            if (std::getenv("RBS_DEBUG_IPC")) std::fprintf(stderr, "[rbs ipc] rank %d opens rank %d buffer %d\n", rank, k, m);
            const hipError_t e = hipIpcOpenMemHandle(&p, b.mem[m], hipIpcMemLazyEnablePeerAccess);
            if (std::getenv("RBS_DEBUG_IPC")) std::fprintf(stderr, "[rbs ipc] rank %d opened rank %d buffer %d -> %p (%d)\n", rank, k, m, p, (int)e);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(h, RBS_ERR_HIP, fmt("hipIpcOpenMemHandle(rank %d, buffer %d): %s", k, m, hipGetErrorString(e)));
            }
            h->peer_mapped[k][m] = p;
        }
        for (int c = 0; c < 2; ++c) {
            h->peer_occ[k][c] = static_cast<const float*>(h->peer_mapped[k][c]);
            h->peer_win[k][c] = static_cast<const int4*>(h->peer_mapped[k][2 + c]);
            h->peer_reg[k][c] = static_cast<const int4*>(h->peer_mapped[k][4 + c]);
        }
    }
    h->peer_world = world;
    h->peer_rank = rank;
    return RBS_OK;
}

int32_t rbs_stage_windows(rbs_handle* h, const int32_t* d_src_global, const int32_t* d_dst_local, int32_t n, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) return fail(h, RBS_ERR_UNSUPPORTED, "stage_windows: a handle over several devices reads every parent in place");
    if (!h->windowed || (h->cols & 3))
        return fail(h, RBS_ERR_UNSUPPORTED, "stage_windows: windowed planes only (state_layout dense moves whole planes: rbs_export/import_plane)");
    if (n < 0 || n > h->max_particles || (n > 0 && (!d_src_global || !d_dst_local)))
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("stage_windows: n = %d outside 0..%d, or a null pointer", n, h->max_particles));
    if (n == 0) return RBS_OK;
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    if (h->join_pending >= 0) RBS_HIP(h, hipStreamWaitEvent(s, h->ev_join[h->join_pending], 0));
    DevParams P = h->base;
    P.slots = h->max_particles; P.n_dev = 1; P.shard_cap = h->max_particles;
    P.slab_px = h->slab_px; P.plane_stride = (int)h->plane_stride;
    P.exact = h->exact ? 1 : 0; P.plane_px = h->slab_px ? h->slab_px : h->npx;
    P.occ_src = h->d_occ[h->cur]; P.win_src = h->d_win[h->cur]; P.reg_src = h->d_reg[h->cur];
    if (h->peer_world > 1) {
        P.n_dev = h->peer_world;
        P.slots = h->peer_world * h->max_particles;
        for (int k = 0; k < P.n_dev; ++k) {
            P.occ_src_dev[k] = k == h->peer_rank ? h->d_occ[h->cur] : h->peer_occ[k][h->cur];
            P.win_src_dev[k] = k == h->peer_rank ? h->d_win[h->cur] : h->peer_win[k][h->cur];
            P.reg_src_dev[k] = k == h->peer_rank ? h->d_reg[h->cur] : h->peer_reg[k][h->cur];
        }
    }
    hipLaunchKernelGGL(rbs::rbs_stage_kernel, dim3((unsigned)n, 4), dim3(256), 0, s, P, d_src_global, d_dst_local, h->d_occ[h->cur],
                       h->d_win[h->cur], h->d_reg[h->cur]);
    RBS_HIP(h, hipGetLastError());
    return RBS_OK;
}

int32_t rbs_peer_resample(rbs_handle* h, const double* d_loglik_all, const double* d_uniforms_sorted, int32_t n_total, int32_t n_local,
                          int32_t rank, int32_t min_share, double temperature, int32_t* d_parent_idx, int32_t* d_stage_src,
                          int32_t* d_stage_dst, int32_t* d_parents_local, int64_t* d_counts, void* stream)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) return fail(h, RBS_ERR_UNSUPPORTED, "peer_resample: a handle over several devices resamples inside rbs_tracker_*");
    if (n_local <= 0 || n_total < n_local || n_total % n_local != 0 || rank < 0 || rank >= n_total / n_local)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("peer_resample: n_total = %d is not world x n_local = %d with rank %d inside it", n_total, n_local, rank));
    if (2 * (long)n_local > (long)h->max_particles)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("peer_resample: n_local = %d needs max_particles >= %d (own slots + staging slots), have %d",
                                                      n_local, 2 * n_local, h->max_particles));
    if (h->peer_world > 1 && (h->peer_world != n_total / n_local || h->peer_rank != rank))
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("peer_resample: rank %d of %d, but the handle is attached as rank %d of %d", rank, n_total / n_local,
                                                      h->peer_rank, h->peer_world));
    if (!d_loglik_all || !d_uniforms_sorted || !d_parent_idx || !d_stage_src || !d_stage_dst || !d_counts)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "peer_resample: null pointer");
    if (!(temperature > 0.0) || min_share < 1) return fail(h, RBS_ERR_INVALID_ARGUMENT, "peer_resample: temperature must be > 0, min_share >= 1");
    RBS_HIP(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    const int tiles = (n_total + rbp::kTile - 1) / rbp::kTile;
    if (tiles > rbp::kMaxTiles) return fail(h, RBS_ERR_UNSUPPORTED, fmt("peer_resample: n_total = %d exceeds %d particles", n_total, rbp::kMaxTiles * rbp::kTile));
    const size_t need = sizeof(double) * ((size_t)n_total + 2 * (size_t)tiles) + 2 * sizeof(int) * (size_t)n_local;
    if (need > h->peer_scratch_bytes) {   // (first call, or a larger job: the only synchronising path)
        RBS_HIP(h, hipStreamSynchronize(s));
        (void)hipFree(h->d_peer_scratch);
        h->d_peer_scratch = nullptr;
        h->peer_scratch_bytes = 0;
        RBS_HIP(h, hipMalloc(&h->d_peer_scratch, need));
        h->peer_scratch_bytes = need;
    }
    rbp::PeerPlan Q;
    Q.ll_all = d_loglik_all; Q.uniforms = d_uniforms_sorted;
    Q.N = n_total; Q.n = n_local; Q.rank = rank; Q.cap = h->max_particles; Q.min_share = min_share;
    Q.temperature = temperature;
    Q.cdf = static_cast<double*>(h->d_peer_scratch);
    Q.tile_max = Q.cdf + n_total;
    Q.tile_total = Q.tile_max + tiles;
    Q.mine = reinterpret_cast<int*>(Q.tile_total + tiles);
    Q.aux = Q.mine + n_local;
    Q.parent_idx = d_parent_idx; Q.stage_src = d_stage_src; Q.stage_dst = d_stage_dst;
    Q.parents_local = d_parents_local;
    Q.counts = reinterpret_cast<long long*>(d_counts);
    // (every launch is checked where it is made: a refused launch must name its kernel, not surface at the last one -- VERDICT r4 #7)
    if (tiles > rbp::kOwnMaxTiles) {
        hipLaunchKernelGGL(rbp::peer_max_kernel, dim3(tiles), dim3(rbp::kThreads), 0, s, Q);
        RBS_HIP(h, hipGetLastError());
    }
    hipLaunchKernelGGL(rbp::peer_weights_kernel, dim3(tiles), dim3(rbp::kThreads), 0, s, Q);
    RBS_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(rbp::peer_search_kernel, dim3((unsigned)((n_local + rbp::kThreads - 1) / rbp::kThreads)), dim3(rbp::kThreads), 0, s, Q);
    RBS_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(rbp::peer_resample_kernel, dim3(1), dim3(rbp::kThreads), 0, s, Q);
    RBS_HIP(h, hipGetLastError());
    return RBS_OK;
}

int32_t rbs_render_depth(rbs_handle* h, const double* pose, float* out)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_FIRST(h, rbs_render_depth(sh_, pose, out));
    if (!pose || !out) return fail(h, RBS_ERR_INVALID_ARGUMENT, "render_depth: null pointer");
    RBS_HIP(h, hipSetDevice(h->device));
    RBS_HIP(h, hipMemcpyAsync(h->d_poses, pose, sizeof(double) * 12 * h->n_bodies,
                              hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(rbs::rbs_fill_kernel, dim3(256), dim3(256), 0, h->stream, h->d_render,
                       (size_t)h->npx, INFINITY);
    DevParams P = h->base;
    P.poses = h->d_poses;
    P.n = 1;
    P.tile_w = 256;
    P.tile_px = rbs::kTilePxBig;
    P.tile_h = P.tile_px / 256;
    hipLaunchKernelGGL(rbs::rbs_render_kernel, dim3(1), dim3(rbs::kBlock), rbs::smem_bytes(P.tile_px, false),
                       h->stream, P, h->d_render);
    RBS_HIP(h, hipGetLastError());
    RBS_HIP(h, hipMemcpyAsync(out, h->d_render, sizeof(float) * h->npx, hipMemcpyDeviceToHost,
                              h->stream));
    RBS_HIP(h, hipStreamSynchronize(h->stream));
    return RBS_OK;
}

#ifdef RBS_PHASE_TIMING
int32_t rbs_debug_phase_cycles(rbs_handle* h, unsigned long long* out8)
{
    if (!h || !h->d_phase) return RBS_ERR_INVALID_ARGUMENT;
    RBS_HIP(h, hipDeviceSynchronize());
    RBS_HIP(h, hipMemcpy(out8, h->d_phase, 256, hipMemcpyDeviceToHost));   // 16 cycle counters + 16 event counters
    RBS_HIP(h, hipMemset(h->d_phase, 0, 256));
    return RBS_OK;
}
#endif

int32_t rbs_last_kernel_ms(rbs_handle* h, float* ms)
{
    float copy_ms = 0.f;
    int32_t used = 0;
    return rbs_timing_summary(h, 1, ms, &copy_ms, &used);
}

int32_t rbs_timing_summary(rbs_handle* h, int32_t last_n, float* call_ms, float* copy_kernel_ms,
                           int32_t* n_used)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_FIRST(h, rbs_timing_summary(sh_, last_n, call_ms, copy_kernel_ms, n_used));
    if (!call_ms || !copy_kernel_ms || !n_used || last_n <= 0)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, "timing_summary: bad argument");
    if (h->calls == 0) return fail(h, RBS_ERR_INVALID_ARGUMENT, "timing_summary: no loglikes launched yet");
    RBS_HIP(h, hipSetDevice(h->device));
    // the timed calls among the last last_n calls (at least the most recent timed one)
    const long n = std::max<long>(1, std::min<long>({((long)last_n + h->timing_every - 1) / h->timing_every,
                                                      h->timed_calls, (long)rbs_handle::kRing}));
    double tot = 0.0, cpy = 0.0;
    int n_copy = 0;
    for (long k = 0; k < n; ++k) {
        const int slot = (int)((h->timed_calls - 1 - k) % rbs_handle::kRing);
        float ms = 0.f;
        RBS_HIP(h, hipEventSynchronize(h->ev_stop[slot]));
        RBS_HIP(h, hipEventElapsedTime(&ms, h->ev_start[slot], h->ev_stop[slot]));
        tot += ms;
        if (h->ring_update[slot]) {
            RBS_HIP(h, hipEventSynchronize(h->ev_copy_stop[slot]));
            RBS_HIP(h, hipEventElapsedTime(&ms, h->ev_copy_start[slot], h->ev_copy_stop[slot]));
            cpy += ms;
            ++n_copy;
        }
    }
    *call_ms = (float)(tot / (double)n);
    *copy_kernel_ms = n_copy ? (float)(cpy / n_copy) : 0.f;
    *n_used = (int32_t)n;
    return RBS_OK;
}

int32_t rbs_set_timing_every(rbs_handle* h, int32_t every)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_ALL(h, rbs_set_timing_every(sh_, every));
    if (every < 1) return fail(h, RBS_ERR_INVALID_ARGUMENT, "set_timing_every: every must be >= 1");
    h->timing_every = every;
    return RBS_OK;
}

int32_t rbs_raster_kernel_ms(rbs_handle* h, int32_t last_n, float* raster_kernel_ms)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_FIRST(h, rbs_raster_kernel_ms(sh_, last_n, raster_kernel_ms));
    if (!raster_kernel_ms || last_n <= 0) return fail(h, RBS_ERR_INVALID_ARGUMENT, "raster_kernel_ms: bad argument");
    if (h->calls == 0) return fail(h, RBS_ERR_INVALID_ARGUMENT, "raster_kernel_ms: no loglikes launched yet");
    RBS_HIP(h, hipSetDevice(h->device));
    const long n = std::max<long>(1, std::min<long>({((long)last_n + h->timing_every - 1) / h->timing_every,
                                                      h->timed_calls, (long)rbs_handle::kRing}));
    double tot = 0.0;
    for (long k = 0; k < n; ++k) {
        const int slot = (int)((h->timed_calls - 1 - k) % rbs_handle::kRing);
        float ms = 0.f;
        RBS_HIP(h, hipEventSynchronize(h->ev_raster_stop[slot]));
        RBS_HIP(h, hipEventElapsedTime(&ms, h->ev_raster_start[slot], h->ev_raster_stop[slot]));
        tot += ms;
    }
    *raster_kernel_ms = (float)(tot / (double)n);
    return RBS_OK;
}

int32_t rbs_get_window(rbs_handle* h, int32_t slot, int32_t out[4])
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_SLOT(h, slot, rbs_get_window(sh_, l_, out));
    if (slot < 0 || slot >= h->max_particles || !out)
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("get_window: bad slot %d", slot));
    RBS_HIP(h, hipSetDevice(h->device));
    if (int32_t rc = drain(h, true)) return rc;
    RBS_HIP(h, hipMemcpy(out, h->d_win[h->cur] + slot, sizeof(int32_t) * 4, hipMemcpyDeviceToHost));
    return RBS_OK;
}

int32_t rbs_set_option(rbs_handle* h, int32_t option, double value)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!std::isfinite(value)) return fail(h, RBS_ERR_INVALID_ARGUMENT, "set_option: the value is not finite");
    std::vector<rbs_handle*> all(h->shards.begin(), h->shards.end());
    all.push_back(h);      // (a group keeps the shared trail's policy itself; its shards the rest)
    for (rbs_handle* x : all) {
        switch (option) {
            case RBS_OPT_SHARED_TRAIL: x->stp_allowed = value != 0.0; break;
            case RBS_OPT_SHARED_TRAIL_ENTER:
                if (!(value > 0.0 && value <= 1.0)) return fail(h, RBS_ERR_INVALID_ARGUMENT, "set_option: RBS_OPT_SHARED_TRAIL_ENTER must be in (0, 1]");
                x->stp_enter = value; break;
            case RBS_OPT_SHARED_TRAIL_EVERY:
                if (!(value >= 1.0)) return fail(h, RBS_ERR_INVALID_ARGUMENT, "set_option: RBS_OPT_SHARED_TRAIL_EVERY must be >= 1");
                x->stp_every = (int)value; break;
            case RBS_OPT_TRACKER_SPLIT_MAX:
                if (!(value >= 0.0)) return fail(h, RBS_ERR_INVALID_ARGUMENT, "set_option: RBS_OPT_TRACKER_SPLIT_MAX must be >= 0");
                x->tracker_split_max = (int)std::min(value, 2e9); break;
            case RBS_OPT_TIMING_EVERY:
                if (!(value >= 1.0)) return fail(h, RBS_ERR_INVALID_ARGUMENT, "set_option: RBS_OPT_TIMING_EVERY must be >= 1");
                x->timing_every = (int)std::min(value, 1e9); break;
            default: return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("set_option: unknown option %d", option));
        }
    }
    return RBS_OK;
}

int32_t rbs_window_fraction(rbs_handle* h, double* out)
{
    if (!h || !out) return RBS_ERR_INVALID_ARGUMENT;
    double f = h->area_frac;
    for (rbs_handle* sh : h->shards) f = std::max(f, sh->area_frac);
    *out = f;
    return RBS_OK;
}

int32_t rbs_shared_trail_rebase(rbs_handle* h, int32_t global_slot)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    if (!h->shards.empty()) return fail(h, RBS_ERR_UNSUPPORTED, "shared_trail_rebase: a handle over several devices decides for its shards itself");
    if (!h->windowed) return fail(h, RBS_ERR_UNSUPPORTED, "shared_trail_rebase: windowed planes only");
    const int slots = std::max(1, h->peer_world) * h->max_particles;
    if (global_slot != -2 && (global_slot < 0 || global_slot >= slots))
        return fail(h, RBS_ERR_INVALID_ARGUMENT, fmt("shared_trail_rebase: slot %d outside 0..%d (or -2: leave)", global_slot, slots - 1));
    h->stp_request = global_slot;
    return RBS_OK;
}

int32_t rbs_shared_trail_state(rbs_handle* h, int32_t* active, int32_t* rebases)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    // (a handle over several devices: the group's own state -- one decision for all shards)
    if (active) *active = h->stp ? 1 : 0;
    if (rebases) *rebases = (int32_t)h->stp_rebases;
    return RBS_OK;
}

int32_t rbs_get_background(rbs_handle* h, float* out)
{
    if (!h) return RBS_ERR_INVALID_ARGUMENT;
    RBS_GROUP_FIRST(h, rbs_get_background(sh_, out));
    if (!out) return fail(h, RBS_ERR_INVALID_ARGUMENT, "get_background: null pointer");
    *out = h->background;
    return RBS_OK;
}

// ---------------------------------------------------------------------------- device tracker
}  // extern "C"

struct rbs_tracker {
    const double* frame64 = nullptr;   // rbs_tracker_submit_f64: this submit's frame as doubles (tracker_submit_impl reads it instead of `frame`)
    std::vector<float> frame_tmp;      // ... converted here where a float frame is needed (a sensor over several devices)
    rbs_handle* s = nullptr;
    rbt::TrackerDev T{};
    std::vector<void*> allocs;
    double* d_normals = nullptr;   // staging for host-supplied randomness
    double* d_uniforms = nullptr;
    std::string err;
    // a sensor over several devices: one replica (all particle states + the filter's kernels) per
    // shard; `s` is then the group handle and T is unused
    std::vector<rbs_tracker*> reps;
    // pipelining (rbs_tracker_submit / rbs_tracker_result): up to two frames in flight; per slot the
    // host-supplied randomness (pinned staging + its device image) and the frame's result (pinned)
    double* h_normals[2] = {nullptr, nullptr};
    double* h_uniforms[2] = {nullptr, nullptr};
    double* d_normals2[2] = {nullptr, nullptr};
    double* d_uniforms2[2] = {nullptr, nullptr};
    double* h_state[2] = {nullptr, nullptr};
    int* h_flags[2] = {nullptr, nullptr};
    int* h_serr[2] = {nullptr, nullptr};
    hipEvent_t ev_res[2] = {nullptr, nullptr};
    double* h_state_dev[2] = {nullptr, nullptr};   // h_state / h_flags as the device addresses them
    int* h_flags_dev[2] = {nullptr, nullptr};
    int32_t res_rc[2] = {0, 0};       // (a handle over several devices runs submit synchronously)
    int res_seq[2] = {0, 0};          // the frame number the kernel that finishes slot k's frame stores into h_flags[k][2]
    long submitted = 0, collected = 0;
    bool recentre_pending = false;    // T.part_old still holds the particles before the last frame's re-centring
    bool poisoned = false;            // a frame failed half-way (buffers partly swapped): rbs_tracker_initialize first
};

namespace {
int32_t tfail(rbs_tracker* t, int32_t code, const std::string& msg) { t->err = msg; t->s->err = msg; return code; }

#define RBT_HIP(t, call)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                           \
            return tfail(t, e_ == hipErrorOutOfMemory ? RBS_ERR_OUT_OF_MEMORY : RBS_ERR_HIP,   \
                         fmt("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__)); \
        }                                                                                      \
    } while (0)

template <typename X>
int32_t talloc(rbs_tracker* t, X** p, size_t count)
{
    RBT_HIP(t, hipMalloc(p, sizeof(X) * (count ? count : 1)));
    t->allocs.push_back(*p);
    return RBS_OK;
}
}  // namespace

extern "C" void rbs_tracker_destroy(rbs_tracker* t);

namespace {
// The filter step's weights and the tracker's mean: single-block kernels for few particles (the
// launch chain is what counts there), grid kernels from kMultiBlockFrom particles on.
void launch_weights(const rbt::TrackerDev& T, int updated, hipStream_t s)
{
    if (T.n < rbt::kMultiBlockFrom) {
        hipLaunchKernelGGL(rbt::weights_kernel, dim3(1), dim3(1024), 0, s, T, updated);
        return;
    }
    const int blocks = (T.n + rbt::kChunk - 1) / rbt::kChunk;
    hipLaunchKernelGGL(rbt::weights_w1_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, T, updated);
    hipLaunchKernelGGL(rbt::weights_w2_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, T);
    hipLaunchKernelGGL(rbt::weights_w3_kernel, dim3(1), dim3(64), 0, s, T, blocks);
    hipLaunchKernelGGL(rbt::weights_w4_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, T);
}
void launch_mean(const rbt::TrackerDev& T, hipStream_t s)
{
    if (T.n < rbt::kMultiBlockFrom) {
        hipLaunchKernelGGL(rbt::mean_kernel, dim3(1), dim3(1024), 0, s, T);
        return;
    }
    const int blocks = (T.n + rbt::kChunk - 1) / rbt::kChunk;
    hipLaunchKernelGGL(rbt::mean_m1_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, T);
    hipLaunchKernelGGL(rbt::mean_m2_kernel, dim3((unsigned)blocks), dim3(1024), 0, s, T);
    hipLaunchKernelGGL(rbt::mean_m3_kernel, dim3(1), dim3(64), 0, s, T, blocks);
}

// One device's tracker state.  cap > 0: the sensor is a shard of a group with `cap` slots per device.
int32_t tracker_create_one(rbs_handle* sensor, const rbs_tracker_params* p, int n_dev, int cap, rbs_tracker** out)
{
    RBS_HIP(sensor, hipSetDevice(sensor->device));
    rbs_tracker* t = new (std::nothrow) rbs_tracker;
    if (!t) return fail(sensor, RBS_ERR_OUT_OF_MEMORY, "tracker_create: out of host memory");
    t->s = sensor;
    rbt::TrackerDev& T = t->T;
    T.n = p->n_particles;
    T.parts = sensor->n_bodies;
    T.D = T.parts * rbt::kBody;
    for (int k = 0; k < 3; ++k) { T.sigma[k] = p->linear_sigma[k]; T.sigma[3 + k] = p->angular_sigma[k]; }
    T.vf = p->velocity_factor;
    T.max_kl = p->max_kl_divergence;
    const size_t n = (size_t)T.n, D = (size_t)T.D, P6 = (size_t)T.parts * 6;
    int32_t rc = RBS_OK;
    if ((rc = talloc(t, &T.part_old, n * D)) || (rc = talloc(t, &T.part_new, n * D)) ||
        (rc = talloc(t, &T.part_old2, n * D)) || (rc = talloc(t, &T.part_new2, n * D)) ||
        (rc = talloc(t, &T.noise, n * P6)) || (rc = talloc(t, &T.noise2, n * P6)) ||
        (rc = talloc(t, &T.logw, n)) || (rc = talloc(t, &T.ll, n)) || (rc = talloc(t, &T.ll2, n)) ||
        (rc = talloc(t, &T.ll_new, n)) || (rc = talloc(t, &T.idx, n)) || (rc = talloc(t, &T.idx2, n)) ||
        (rc = talloc(t, &T.parents, n)) || (rc = talloc(t, &T.cdf, n)) || (rc = talloc(t, &T.deflt, D)) ||
        (rc = talloc(t, &T.mean, D + (size_t)T.parts * 9)) || (rc = talloc(t, &T.poses, n * (size_t)T.parts * 12)) ||
        (rc = talloc(t, &T.flag, 2)) || (rc = talloc(t, &T.red, (size_t)rbt::kRedBlocks * (3 + D))) ||
        (rc = talloc(t, &t->d_normals, n * P6)) ||
        (rc = talloc(t, &t->d_uniforms, n * (size_t)T.parts)) ||
        (cap > 0 && ((rc = talloc(t, &T.layout, n)) || (rc = talloc(t, &T.ll_sorted, (size_t)n_dev * cap)) ||
                     (rc = talloc(t, &T.poses_sorted, (size_t)cap * T.parts * 12)) || (rc = talloc(t, &T.idx_sorted, (size_t)cap))))) {
        rbs_tracker_destroy(t);
        return rc;
    }
    if (hipMemsetAsync(T.deflt, 0, sizeof(double) * D, sensor->stream) != hipSuccess) {
        (void)hipGetLastError();
        rbs_tracker_destroy(t);
        return fail(sensor, RBS_ERR_HIP, "tracker_create: hipMemsetAsync failed");
    }
    t->d_normals2[0] = t->d_normals;
    t->d_uniforms2[0] = t->d_uniforms;
    if ((rc = talloc(t, &t->d_normals2[1], n * P6)) || (rc = talloc(t, &t->d_uniforms2[1], n * (size_t)T.parts))) {
        rbs_tracker_destroy(t);
        return rc;
    }
    for (int k = 0; k < 2; ++k) {
        if (hipHostMalloc(&t->h_normals[k], sizeof(double) * n * P6, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&t->h_uniforms[k], sizeof(double) * n * T.parts, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&t->h_state[k], sizeof(double) * D, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&t->h_flags[k], sizeof(int) * 4, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc(&t->h_serr[k], 2 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&t->ev_res[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            rbs_tracker_destroy(t);
            return fail(sensor, RBS_ERR_OUT_OF_MEMORY, "tracker_create: pinned host memory");
        }
        t->h_serr[k][0] = t->h_serr[k][1] = 0;
        t->h_flags[k][2] = 0;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&t->h_state_dev[k]), t->h_state[k], 0) != hipSuccess ||
            hipHostGetDevicePointer(reinterpret_cast<void**>(&t->h_flags_dev[k]), t->h_flags[k], 0) != hipSuccess) {
            (void)hipGetLastError();
            rbs_tracker_destroy(t);
            return fail(sensor, RBS_ERR_HIP, "tracker_create: hipHostGetDevicePointer failed");
        }
    }
    *out = t;
    return RBS_OK;
}
}  // namespace

extern "C" {

int32_t rbs_tracker_create(rbs_handle* sensor, const rbs_tracker_params* p, rbs_tracker** out)
{
    if (!sensor || !out) return RBS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const int total = sensor->shards.empty() ? sensor->max_particles : (int)sensor->shards.size() * sensor->shard_cap;
    if (!p || p->n_particles <= 0 || p->n_particles > total)
        return fail(sensor, RBS_ERR_INVALID_ARGUMENT, "tracker_create: n_particles outside 1..max_particles");
    if (p->n_particles > rbt::kRedBlocks * rbt::kChunk)
        return fail(sensor, RBS_ERR_INVALID_ARGUMENT, fmt("tracker_create: at most %d particles", rbt::kRedBlocks * rbt::kChunk));
    if (sensor->shards.empty()) return tracker_create_one(sensor, p, 1, 0, out);
    rbs_tracker* g = new (std::nothrow) rbs_tracker;
    if (!g) return fail(sensor, RBS_ERR_OUT_OF_MEMORY, "tracker_create: out of host memory");
    g->s = sensor;
    for (rbs_handle* sh : sensor->shards) {
        rbs_tracker* r = nullptr;
        if (int32_t rc = tracker_create_one(sh, p, (int)sensor->shards.size(), sensor->shard_cap, &r)) {
            sensor->err = sh->err;
            rbs_tracker_destroy(g);
            return rc;
        }
        g->reps.push_back(r);
    }
    *out = g;
    return RBS_OK;
}

void rbs_tracker_destroy(rbs_tracker* t)
{
    if (!t) return;
    if (!t->reps.empty() || !t->s->shards.empty()) {
        for (rbs_tracker* r : t->reps) rbs_tracker_destroy(r);
        delete t;
        return;
    }
    (void)hipSetDevice(t->s->device);
    (void)hipStreamSynchronize(t->s->stream);
    for (void* p : t->allocs) (void)hipFree(p);
    for (int k = 0; k < 2; ++k) {
        if (t->h_normals[k]) (void)hipHostFree(t->h_normals[k]);
        if (t->h_uniforms[k]) (void)hipHostFree(t->h_uniforms[k]);
        if (t->h_state[k]) (void)hipHostFree(t->h_state[k]);
        if (t->h_flags[k]) (void)hipHostFree(t->h_flags[k]);
        if (t->h_serr[k]) (void)hipHostFree(t->h_serr[k]);
        if (t->ev_res[k]) (void)hipEventDestroy(t->ev_res[k]);
    }
    delete t;
}

int32_t rbs_tracker_initialize(rbs_tracker* t, const double* default_state)
{
    if (!t) return RBS_ERR_INVALID_ARGUMENT;
    if (!default_state) return tfail(t, RBS_ERR_INVALID_ARGUMENT, "tracker_initialize: null state");
    t->submitted = t->collected = 0;
    t->recentre_pending = false;
    t->poisoned = false;
    if (!t->reps.empty()) {
        for (rbs_tracker* r : t->reps)
            if (int32_t rc = rbs_tracker_initialize(r, default_state)) { t->s->err = r->s->err; return rc; }
        for (rbs_handle* sh : t->s->shards) {   // the shards' resets left their streams idle: a clean slate for the group's ordering
            RBT_HIP(t, hipSetDevice(sh->device));
            RBT_HIP(t, hipEventRecord(sh->ev_done, sh->stream));
        }
        t->s->poisoned = false;
        t->s->frame_acquired = false;
        return RBS_OK;
    }
    rbt::TrackerDev& T = t->T;
    RBT_HIP(t, hipSetDevice(t->s->device));
    if (int32_t rc = rbs_reset(t->s)) return rc;
    hipStream_t s = t->s->stream;
    RBT_HIP(t, hipMemcpyAsync(T.deflt, default_state, sizeof(double) * T.D, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(rbt::init_kernel, dim3((unsigned)((T.n + 255) / 256)), dim3(256), 0, s, T);
    RBT_HIP(t, hipGetLastError());
    RBT_HIP(t, hipStreamSynchronize(s));
    T.frame = 0;
    for (int k = 0; k < 2; ++k) { t->h_flags[k][2] = 0; t->res_seq[k] = 0; }   // frame numbers start over
    if (t->s->slab_px && !t->s->group) {
        // Window-sized slabs: a tracker frame that overflows cannot be repeated, so the slabs are sized
        // BEFORE the first frame -- one updating probe call of a single particle at the default pose
        // measures the object's screen rectangle, the housekeeping enlarges the slabs to hold it with
        // room to move, and the sensor is reset again.  From there on the regions grow a few pixels
        // per frame and the housekeeping at every frame's result keeps ahead of them.
        rbs_handle* h = t->s;
        hipLaunchKernelGGL(rbt::default_pose_kernel, dim3(1), dim3(64), 0, s, T);
        RBT_HIP(t, hipGetLastError());
        if (int32_t rc = enqueue_loglikes(h, T.poses, T.idx, 1, true, T.ll_new, s)) return rc;
        if (int32_t rc = drain(h, true)) return rc;
        RBT_HIP(t, hipMemcpy(h->h_err, h->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost));
        if (int32_t rc = slab_housekeeping(h)) return rc;
        if (int32_t rc = rbs_reset(h)) return rc;
    }
    return RBS_OK;
}

}  // extern "C" (group_tracker_track below needs C++ linkage)

namespace {
// One frame on a sensor over several devices.  Every device runs the whole filter on all the
// particle states (identical inputs, identical code); the sensor call alone is sharded: device k
// evaluates the particles laid out at slots [k cap, (k+1) cap) (layout_kernel), and the
// log-likelihoods are exchanged with ONE all-gather per sampling block (RCCL over xGMI).
int32_t group_tracker_track(rbs_tracker* t, const float* frame, const double* normals, const double* uniforms,
                            uint64_t seed, double* out_state, int32_t* out_resamplings)
{
    rbs_handle* g = t->s;
    const int nd = (int)g->shards.size(), cap = g->shard_cap;
    if (frame)
        if (int32_t rc = rbs_set_observation_f32(g, frame, (size_t)g->npx)) return rc;
    std::vector<hipStream_t> streams(nd);
    std::vector<double*> llbufs(nd);
    for (int k = 0; k < nd; ++k) {
        rbs_tracker* r = t->reps[k];
        rbt::TrackerDev& T = r->T;
        rbs_handle* h = r->s;
        streams[k] = h->stream;
        llbufs[k] = T.ll_sorted;
        RBT_HIP(t, hipSetDevice(h->device));
        const size_t n = (size_t)T.n;
        T.normals = nullptr;
        T.uniforms = nullptr;
        if (normals) {
            RBT_HIP(t, hipMemcpyAsync(r->d_normals, normals, sizeof(double) * n * T.parts * 6, hipMemcpyHostToDevice, h->stream));
            T.normals = r->d_normals;
        }
        if (uniforms) {
            RBT_HIP(t, hipMemcpyAsync(r->d_uniforms, uniforms, sizeof(double) * n * T.parts, hipMemcpyHostToDevice, h->stream));
            T.uniforms = r->d_uniforms;
        }
        T.seed = seed;
    }
    const int parts = t->reps[0]->T.parts, n = t->reps[0]->T.n;
    const dim3 g256((unsigned)((n + 255) / 256)), b256(256);
    for (int b = 0; b < parts; ++b) {
        const bool last = b == parts - 1;
        for (int k = 0; k < nd; ++k) {
            rbs_tracker* r = t->reps[k];
            RBT_HIP(t, hipSetDevice(r->s->device));
            hipLaunchKernelGGL(rbt::propagate_kernel, g256, b256, 0, streams[k], r->T, b, 0);
            hipLaunchKernelGGL(rbt::layout_kernel, dim3(1), dim3(1024), 0, streams[k], r->T, nd, cap);
            const int lo = std::min(n, k * cap), cnt = std::min(n, (k + 1) * cap) - lo;
            if (cnt > 0)
                hipLaunchKernelGGL(rbt::shard_gather_kernel, dim3((unsigned)((cnt + 255) / 256)), b256, 0, streams[k], r->T, lo, cnt);
            RBT_HIP(t, hipGetLastError());
        }
        if (int32_t rc = group_begin_call(g, nullptr, last)) return rc;
        for (int k = 0; k < nd; ++k) {
            rbs_tracker* r = t->reps[k];
            rbs_handle* h = r->s;
            RBT_HIP(t, hipSetDevice(h->device));
            const int lo = std::min(n, k * cap), cnt = std::min(n, (k + 1) * cap) - lo;
            if (cnt > 0) {
                if (int32_t rc = enqueue_loglikes(h, r->T.poses_sorted, r->T.idx_sorted, cnt, last, r->T.ll_sorted + lo, h->stream))
                    return gfail(g, h, rc);
            } else {
                if (int32_t rc = advance_empty(h, last)) return gfail(g, h, rc);
                RBT_HIP(t, hipEventRecord(h->ev_done, h->stream));
            }
        }
        if (int32_t rc = group_allgather(g, llbufs.data(), (size_t)cap, streams.data())) return rc;
        for (int k = 0; k < nd; ++k) {
            rbs_tracker* r = t->reps[k];
            rbt::TrackerDev& T = r->T;
            RBT_HIP(t, hipSetDevice(r->s->device));
            hipStream_t s = streams[k];
            hipLaunchKernelGGL(rbt::shard_scatter_kernel, g256, b256, 0, s, T, last ? 1 : 0);
            launch_weights(T, 0, s);
            hipLaunchKernelGGL(rbt::resample_kernel, g256, b256, 0, s, T, b);
            hipLaunchKernelGGL(rbt::gather_kernel, dim3((unsigned)T.n), dim3(64), 0, s, T);
            RBT_HIP(t, hipGetLastError());
            std::swap(T.part_old, T.part_old2);
            std::swap(T.part_new, T.part_new2);
            std::swap(T.noise, T.noise2);
            std::swap(T.ll, T.ll2);
            std::swap(T.idx, T.idx2);
        }
    }
    int flags[2] = {0, 0};
    for (int k = 0; k < nd; ++k) {
        rbs_tracker* r = t->reps[k];
        rbt::TrackerDev& T = r->T;
        RBT_HIP(t, hipSetDevice(r->s->device));
        launch_mean(T, streams[k]);
        hipLaunchKernelGGL(rbt::recentre_kernel, g256, b256, 0, streams[k], T, T.part_new);
        RBT_HIP(t, hipGetLastError());
        std::swap(T.part_old, T.part_new);   // this frame's particles are the next frame's old ones
        if (k == 0) {
            RBT_HIP(t, hipMemcpyAsync(out_state, T.deflt, sizeof(double) * T.D, hipMemcpyDeviceToHost, streams[k]));
            RBT_HIP(t, hipMemcpyAsync(flags, T.flag, sizeof(flags), hipMemcpyDeviceToHost, streams[k]));
        }
        T.frame += 1;
    }
    for (int k = 0; k < nd; ++k) {   // the host frame / randomness buffers may be reused after this call
        RBT_HIP(t, hipSetDevice(t->reps[k]->s->device));
        RBT_HIP(t, hipStreamSynchronize(streams[k]));
    }
    if (out_resamplings) *out_resamplings = flags[1];
    int32_t slab_rc = RBS_OK;
    for (rbs_handle* sh : g->shards) {   // slabs: a region that did not fit, on any shard
        if (!sh->slab_px) break;
        RBT_HIP(t, hipSetDevice(sh->device));
        RBT_HIP(t, hipMemcpy(sh->h_err, sh->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost));
        if (slab_rc == RBS_OK && check_slab_error(sh) != RBS_OK) slab_rc = gfail(g, sh, RBS_ERR_OUT_OF_MEMORY);
    }
    const std::string msg = g->err;
    if (int32_t rc = group_grow_slabs(g, nullptr)) return rc;   // (every stream is idle here: enlarge before a region fills a slab)
    if (slab_rc) g->err = msg;
    return slab_rc;
}
}  // namespace

extern "C" {

// The body of rbs_tracker_submit; a failure in here happens after some of the frame's work was
// enqueued and some of the tracker's buffers were swapped.
static int32_t tracker_submit_impl(rbs_tracker* t, const float* frame, const double* normals, const double* uniforms, uint64_t seed);

int32_t rbs_tracker_submit(rbs_tracker* t, const float* frame, const double* normals, const double* uniforms, uint64_t seed)
{
    if (!t) return RBS_ERR_INVALID_ARGUMENT;
    if (t->poisoned || t->s->poisoned)
        return tfail(t, RBS_ERR_HIP, "tracker_submit: an earlier frame failed half-way (" + (t->s->poisoned ? t->s->poison_msg : t->err) +
                                         "): call rbs_tracker_initialize");
    const long before = t->submitted;
    const int32_t rc = tracker_submit_impl(t, frame, normals, uniforms, seed);
    if (rc != RBS_OK && !(rc == RBS_ERR_INVALID_ARGUMENT && t->submitted == before)) {
        // drain what was enqueued and refuse further frames: the particle buffers are out of step
        const std::string why = t->err.empty() ? t->s->err : t->err;
        if (t->reps.empty()) { (void)hipSetDevice(t->s->device); (void)hipStreamSynchronize(t->s->stream); (void)hipStreamSynchronize(t->s->copy_stream); }
        else for (rbs_tracker* r : t->reps) { (void)hipSetDevice(r->s->device); (void)hipStreamSynchronize(r->s->stream); (void)hipStreamSynchronize(r->s->copy_stream); }
        (void)hipGetLastError();
        t->poisoned = true;
        t->err = why;
        t->s->err = why;
    }
    return rc;
}

static int32_t tracker_submit_impl(rbs_tracker* t, const float* frame, const double* normals, const double* uniforms, uint64_t seed)
{
    if (t->submitted - t->collected >= 2)
        return tfail(t, RBS_ERR_INVALID_ARGUMENT, "tracker_submit: two frames are in flight already (call rbs_tracker_result)");
    const int slot = (int)(t->submitted & 1);
    const double* frame64 = t->frame64;
    t->frame64 = nullptr;
    if (frame64 && !t->reps.empty()) {   // (the sharded form uploads float frames: converted once here)
        t->frame_tmp.resize((size_t)t->reps[0]->s->npx);
        convert_f64_f32(t->frame_tmp.data(), frame64, t->frame_tmp.size());
        frame = t->frame_tmp.data();
        frame64 = nullptr;
    }
    if (!t->reps.empty()) {   // several devices: the frame runs to completion here, its result waits in the slot
        rbs_tracker* r0 = t->reps[0];
        t->res_rc[slot] = group_tracker_track(t, frame, normals, uniforms, seed, r0->h_state[slot], &r0->h_flags[slot][1]);
        t->submitted += 1;
        return t->res_rc[slot];
    }
    rbt::TrackerDev& T = t->T;
    rbs_handle* h = t->s;
    const BorrowedFrameGuard borrowed_guard{h};   // (the caller's frame is the caller's again when this returns, whatever happened)
    RBT_HIP(t, hipSetDevice(h->device));
    if ((frame || frame64) && h->npx <= 0) return tfail(t, RBS_ERR_INVALID_ARGUMENT, "tracker_submit: bad sensor");
    hipStream_t s = h->stream;
    const size_t n = (size_t)T.n;
    T.normals = nullptr;
    T.uniforms = nullptr;
    // host-supplied randomness goes through pinned staging: the caller's buffers are free on return
    if (normals) {
        std::memcpy(t->h_normals[slot], normals, sizeof(double) * n * T.parts * 6);
        RBT_HIP(t, hipMemcpyAsync(t->d_normals2[slot], t->h_normals[slot], sizeof(double) * n * T.parts * 6, hipMemcpyHostToDevice, s));
        T.normals = t->d_normals2[slot];
    }
    if (uniforms) {
        std::memcpy(t->h_uniforms[slot], uniforms, sizeof(double) * n * T.parts);
        RBT_HIP(t, hipMemcpyAsync(t->d_uniforms2[slot], t->h_uniforms[slot], sizeof(double) * n * T.parts, hipMemcpyHostToDevice, s));
        T.uniforms = t->d_uniforms2[slot];
    }
    T.seed = seed;
    T.host_state = t->h_state_dev[slot];
    T.host_flags = t->h_flags_dev[slot];
    const dim3 g256((unsigned)((T.n + 255) / 256)), b256(256);
    const char* nf = std::getenv("RBS_TRACKER_FUSED");
    const bool fused = T.n <= rbt::kFusedFilterMax && !(nf && std::atoi(nf) == 0);
    static const bool tail_on = [] { const char* e = std::getenv("RBS_TRACKER_TAIL"); return !(e && std::atoi(e) == 0); }();
    const bool tail = !fused && tail_on && T.n < rbt::kMultiBlockFrom;   // (RBS_TRACKER_TAIL=0: the three separate launches, A/B)
    for (int b = 0; b < T.parts; ++b) {
        const bool last = b == T.parts - 1;
        // (the transition fused into the sensor's rectangles kernel -- one launch less -- measured no
        // gain: 3 987 against 3 976 frames/s at 2 000 particles)
        hipLaunchKernelGGL(rbt::propagate_kernel, g256, b256, 0, s, T, b, b == 0 && t->recentre_pending ? 1 : 0);
        if (b == 0) t->recentre_pending = false;
        RBT_HIP(t, hipGetLastError());
        // the frame is handed over AFTER the first transition launch: the transition (and the
        // sensor's rectangles kernel behind it) do not depend on it, and run while the host copies
        // the frame into pinned memory and the copy engine uploads it
        if (b == 0 && (frame || frame64)) {
            // Few particles: the frame's own journey (staging copy, transfer, model terms: ~80 us) is most of a frame, and the sensor's
            // GEOMETRY kernel does not need it -- the frame is handed to the sensor as a borrowed one and staged by enqueue_loglikes
            // between the two kernels of the split launch (inside this call: the caller's buffer is free on return as before).  Above
            // kTrackerSplitMax evaluations the one-kernel launch's shorter kernel time wins (RBS_TRACKER_SPLIT_MAX, 0: never).
            const int split_max = h->tracker_split_max;
            const bool idle = t->submitted == t->collected;   // (a frame already in flight -- look-ahead -- hides the journey by itself, and the one-kernel launch is the shorter)
            if (idle && T.n * T.parts <= split_max && h->precision == RBS_PRECISION_F64 && h->windowed && !h->frame_ingest && !h->group) {
                RBS_REFUSE_POISONED(h);
                h->frame_acquired = false;
                if (int32_t rc = flush_lazy_frame(h, h->stream)) return rc;
                h->prefetched_slot = -1;
                h->borrowed = frame64;          // (doubles, as dbot hands images over: converted while they are staged)
                h->borrowed_f32 = frame64 ? nullptr : frame;
                h->pending_frames += 1;
            } else if (int32_t rc = frame64 ? rbs_set_observation(h, frame64, (size_t)h->npx) : rbs_set_observation_f32(h, frame, (size_t)h->npx)) return rc;
        }
        if (int32_t rc = enqueue_loglikes(h, T.poses, T.idx, T.n, last, T.ll_new, s)) return rc;
        if (fused) {
            hipLaunchKernelGGL(rbt::filter_step_kernel, dim3(1), dim3(1024), 0, s, T, b, last ? 1 : 0, last ? 1 : 0);
        } else if (tail && last) {
            // the estimate first (one launch, the result event right behind it), the gather after
            hipLaunchKernelGGL(rbt::filter_tail_kernel, dim3(1), dim3(1024), 0, s, T, b, 1);
            RBT_HIP(t, hipGetLastError());
            if (h->slab_px) RBT_HIP(t, hipMemcpyAsync(t->h_serr[slot], h->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
            RBT_HIP(t, hipEventRecord(t->ev_res[slot], s));
            hipLaunchKernelGGL(rbt::gather_kernel, dim3((unsigned)T.n), dim3(64), 0, s, T);
        } else {
            launch_weights(T, last ? 1 : 0, s);
            hipLaunchKernelGGL(rbt::resample_gather_kernel, dim3((unsigned)T.n), dim3(64), 0, s, T, b);
        }
        RBT_HIP(t, hipGetLastError());
        std::swap(T.part_old, T.part_old2);
        std::swap(T.part_new, T.part_new2);
        std::swap(T.noise, T.noise2);
        std::swap(T.ll, T.ll2);
        std::swap(T.idx, T.idx2);
    }
    if (!fused) {
        // (re-centring inside the single mean block was tried: its two rounds of rotations per thread
        // on one CU take longer than the separate launch, 25 us against 16.5 + 5)
        if (!tail) launch_mean(T, s);
        // the re-centring itself rides in the next frame's first transition launch (the thread that
        // moves particle i re-centres it first); rbs_tracker_get applies it on demand
        static const bool now = [] { const char* e = std::getenv("RBS_TRACKER_RECENTRE_NOW"); return e && std::atoi(e) != 0; }();
        if (now) hipLaunchKernelGGL(rbt::recentre_kernel, g256, b256, 0, s, T, T.part_new);
        else t->recentre_pending = true;
    }
    RBT_HIP(t, hipGetLastError());
    std::swap(T.part_old, T.part_new);   // this frame's particles are the next frame's old ones
    // (the estimate and the flags were stored into h_state[slot] / h_flags[slot] by the kernel that
    // finished them: no copies behind the last kernel)
    if (!tail) {
        if (h->slab_px) RBT_HIP(t, hipMemcpyAsync(t->h_serr[slot], h->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        RBT_HIP(t, hipEventRecord(t->ev_res[slot], s));
    }
    t->res_seq[slot] = (int)(T.frame + 1);
    T.frame += 1;
    t->res_rc[slot] = RBS_OK;
    t->submitted += 1;
    return RBS_OK;
}

int32_t rbs_tracker_result(rbs_tracker* t, double* out_state, int32_t* out_resamplings)
{
    if (!t) return RBS_ERR_INVALID_ARGUMENT;
    if (!out_state) return tfail(t, RBS_ERR_INVALID_ARGUMENT, "tracker_result: null output");
    if (t->collected >= t->submitted) return tfail(t, RBS_ERR_INVALID_ARGUMENT, "tracker_result: no frame in flight");
    const int slot = (int)(t->collected & 1);
    t->collected += 1;
    rbs_tracker* r = t->reps.empty() ? t : t->reps[0];
    const int D = r->T.D;
    if (t->reps.empty()) {
        RBT_HIP(t, hipSetDevice(t->s->device));
        // The kernel that finishes the frame stores the estimate into pinned (host-coherent) memory and
        // the frame's number behind it with a system-scope release: spin on that number -- the signalling
        // of the kernel's completion and the wake-up behind it cost several microseconds more.  The
        // event is the fall-back (and the only way with slabs, whose error word arrives by a copy).
        static const bool spin = [] { const char* e = std::getenv("RBS_TRACKER_SPIN"); return !(e && std::atoi(e) == 0); }();
        bool seen = false;
        if (spin && !t->s->slab_px && t->res_seq[slot] > 0) {
            volatile int* seq = t->h_flags[slot] + 2;
            for (long it = 0; it < 4000000 && !seen; ++it) {
                seen = __atomic_load_n(seq, __ATOMIC_ACQUIRE) == t->res_seq[slot];
                if (!seen && (it & 255) == 255 && hipEventQuery(t->ev_res[slot]) == hipSuccess) break;
                if (!seen) __builtin_ia32_pause();
            }
        }
        if (!seen) RBT_HIP(t, hipEventSynchronize(t->ev_res[slot]));
    }
    std::memcpy(out_state, r->h_state[slot], sizeof(double) * D);
    if (out_resamplings) *out_resamplings = r->h_flags[slot][1];
    if (!t->reps.empty()) return t->res_rc[slot];
    if (t->submitted == t->collected) t->s->quiet = !t->s->async_outstanding;   // frame by frame: the next frame may travel in pieces
    if (t->s->slab_px) {
        // the frame has run: a region that did not fit cannot be repaired here (the filter has resampled
        // on the contained particle's NaN) and is an error -- but the slabs are enlarged BEFORE that,
        // whenever the largest region asked for has filled three quarters of one, at a frame boundary
        // with nothing in flight (regions move a few pixels per frame)
        rbs_handle* h = t->s;
        h->h_err[0] = t->h_serr[slot][0];
        h->h_err[1] = t->h_serr[slot][1];
        const int32_t rc = check_slab_error(h);
        const std::string msg = h->err;
        if (t->submitted == t->collected)
            if (int32_t rc2 = slab_housekeeping(h)) return rc2;
        if (rc) { h->err = msg; t->err = msg; t->poisoned = true; }
        return rc;
    }
    return RBS_OK;
}

int32_t rbs_tracker_submit_f64(rbs_tracker* t, const double* frame, const double* normals, const double* uniforms, uint64_t seed)
{
    if (!t) return RBS_ERR_INVALID_ARGUMENT;
    t->frame64 = frame;
    const int32_t rc = rbs_tracker_submit(t, nullptr, normals, uniforms, seed);
    t->frame64 = nullptr;
    return rc;
}

int32_t rbs_tracker_track_f64(rbs_tracker* t, const double* frame, const double* normals, const double* uniforms, uint64_t seed,
                              double* out_state, int32_t* out_resamplings)
{
    if (!t) return RBS_ERR_INVALID_ARGUMENT;
    t->frame64 = frame;
    const int32_t rc = rbs_tracker_track(t, nullptr, normals, uniforms, seed, out_state, out_resamplings);
    t->frame64 = nullptr;
    return rc;
}

int32_t rbs_tracker_track(rbs_tracker* t, const float* frame, const double* normals,
                          const double* uniforms, uint64_t seed, double* out_state,
                          int32_t* out_resamplings)
{
    if (!t) return RBS_ERR_INVALID_ARGUMENT;
    if (!out_state) return tfail(t, RBS_ERR_INVALID_ARGUMENT, "tracker_track: null output");
    if (t->submitted != t->collected)
        return tfail(t, RBS_ERR_INVALID_ARGUMENT, "tracker_track: frames submitted with rbs_tracker_submit are still in flight");
    if (int32_t rc = rbs_tracker_submit(t, frame, normals, uniforms, seed)) {
        if (t->submitted != t->collected) t->collected = t->submitted;   // (a group's frame that failed: nothing to collect)
        return rc;
    }
    return rbs_tracker_result(t, out_state, out_resamplings);
}

int32_t rbs_tracker_get(rbs_tracker* t, double* particles, double* log_weights, int32_t* indices)
{
    if (!t) return RBS_ERR_INVALID_ARGUMENT;
    if (!t->reps.empty()) {
        for (rbs_tracker* r : t->reps) { RBT_HIP(t, hipSetDevice(r->s->device)); RBT_HIP(t, hipStreamSynchronize(r->s->stream)); }
        return rbs_tracker_get(t->reps[0], particles, log_weights, indices);
    }
    rbt::TrackerDev& T = t->T;
    RBT_HIP(t, hipSetDevice(t->s->device));
    if (t->recentre_pending) {   // (deferred into the next frame's transition launch: apply it now)
        hipLaunchKernelGGL(rbt::recentre_kernel, dim3((unsigned)((T.n + 255) / 256)), dim3(256), 0, t->s->stream, T, T.part_old);
        RBT_HIP(t, hipGetLastError());
        t->recentre_pending = false;
    }
    RBT_HIP(t, hipStreamSynchronize(t->s->stream));
    if (particles) RBT_HIP(t, hipMemcpy(particles, T.part_old, sizeof(double) * (size_t)T.n * T.D, hipMemcpyDeviceToHost));
    if (log_weights) RBT_HIP(t, hipMemcpy(log_weights, T.logw, sizeof(double) * (size_t)T.n, hipMemcpyDeviceToHost));
    if (indices) RBT_HIP(t, hipMemcpy(indices, T.idx, sizeof(int) * (size_t)T.n, hipMemcpyDeviceToHost));
    return RBS_OK;
}

}  // extern "C"
