// rbsensor_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the RbSensor likelihood evaluator.
//
// One RbSensor::loglikes(deltas, indices, update) call (made once per sampling block inside
// tracker_->track, R:source/dbot_ros/object_tracker_ros.hpp:49) is:
//
//   rbs_frame_prep_kernel  two independent jobs in one launch: ingest of a newly handed-over
//                       frame (precision F64: + its per-pixel model terms), and per particle (one
//                       wave each) the screen rectangle of its bodies from the projected
//                       vertices -- one rectangle per group of overlapping bodies --, its work
//                       items (one per <= 11 008-px tile, one atomicAdd per block of 8
//                       particles), the snapshot of its parent index and, on windowed planes, the
//                       region the copy kernel writes.  (rbs_prep_kernel: without a frame.)
//   rbs_raster_kernel<UPDATE, PREC, SLAB>  PERSISTENT, 3 blocks per CU, pulling work items from
//                       an atomic queue: software depth rasterizer (64-triangle clusters culled
//                       against the tile frustum and, for closed bodies, by their normal cone;
//                       triangles that clearly face away dropped by a float32 plane test and the
//                       survivors compacted through an LDS ring, set up in binary64 with full
//                       lanes; LDS depth tile, ds_min_u32 z-min), then the pixel pass: occlusion
//                       process, covered pixels compacted through a second LDS ring, the Kinect
//                       likelihood + occlusion posterior 64 pixels at a time (PREC 0: binary64
//                       with the reference's float rounding points; PREC 1: float32 on the
//                       exp2/log2/rcp units, every per-pixel term recomputed from the
//                       observation), block reduce -> the particle's log-likelihood.
//                       Bound by VALU issue (bench.py measures the fraction of the chip's issue peak live).
//   rbs_copy_window_kernel  (update only, second stream) the child's window outside its
//                       rectangle: the parent's values advanced by the occlusion process
//                       occ' = snap(fma(alpha, occ, beta)), the background where the parent
//                       stores nothing; re-tightens the child's window.
//   rbs_copy_rows_kernel  whole planes (state_layout dense, or windows grown past half the frame):
//                       streams the parent's plane into the child's slot outside the
//                       rectangle.  HBM bound: 2*4*rows*cols bytes per particle-likelihood.
//
// Occlusion planes are stored as a window + a background level: the
// numbers are those of whole planes, only the bytes moved differ.
//
// Arithmetic contract (tests/ compare against oracle/): the geometry is individually rounded
// binary64 in a fixed operation order (compile with -ffp-contract=off), the stored depth is
// one rounding to float, z-min is order independent, so coverage and depth are bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rbs_math.h"

namespace rbs {

// tuning knobs (overridable with -D for A/B experiments; defaults are the measured best)
// LDS budget: THREE raster blocks per CU.  3 waves/SIMD from three independent work items is
// worth far more than a tile that always holds a whole rectangle (0.327 -> 0.255 ms on C1; the
// rectangles that no longer fit split into two row bands).  The CU grants LDS in 1 280-byte
// steps: 11 520 pixels (45 KB) + the small arrays = 53 296 B is the largest block that still
// fits three times (11 648 pixels does not).
#ifndef RBS_TILE_PX
#define RBS_TILE_PX 11008
#endif
#ifndef RBS_COPY_UNROLL
#define RBS_COPY_UNROLL 8
#endif
#ifndef RBS_NT
#define RBS_NT 1
#endif
#ifndef RBS_BIG_THRESH
#define RBS_BIG_THRESH 96
#endif
// float32 filter of the coverage test in the per-lane sample loops (tri_filter): bit-exact by
// construction and green on every depth test, but MEASURED SLOWER on C1 (raster kernel F32 0.146 ->
// 0.154 ms, F64 0.184 -> 0.187): the loop's instruction count barely moves (36 -> 29 per sample, +25
// per triangle for the filter's coefficients), and its binary64 edge functions were filling the
// bubbles of the division's dependent chain for free.  Two samples per trip (two divisions in
// flight) measured no better either (0.146 -> 0.149).  Off by default; kept as the documented experiment.
#ifndef RBS_EDGE_FILTER
#define RBS_EDGE_FILTER 0
#endif
#ifndef RBS_PIN_EDGES
#define RBS_PIN_EDGES 1        // raster_lane_samples: the edge components stay in registers across the sample loops
#endif
#ifndef RBS_SHARE_VERTICES
#define RBS_SHARE_VERTICES 1   // raster_shared_cluster (0: every triangle transforms its own three vertices)
#endif

#ifndef RBS_BLOCK
#define RBS_BLOCK 256
#endif
constexpr int kBlock = RBS_BLOCK;          // threads per raster block (a multiple of 64)
constexpr int kTilePx = RBS_TILE_PX;       // LDS depth tile (u32 per pixel) of a launch with 3 blocks per CU
constexpr int kTilePxBig = 16384;          // ... of a launch with 2 blocks per CU (whole planes / wide windows)
// Precision F64 keeps the erfc and log tables of rbs_math.h in LDS (per-lane table reads: 5.9 KB
// per block), taken from the depth tile so that the same number of blocks stays resident.
#ifndef RBS_MATH_LDS
#define RBS_MATH_LDS 1
#endif
constexpr int kMathTabDoubles = RBS_MATH_LDS ? rbsm::kErfcIntervals * rbsm::kErfcCoefs + rbsm::kLogIntervals * 2 : 0;
constexpr int kMathTabPx = (kMathTabDoubles * 8 + 63) / 64 * 16;   // the tables' size in tile pixels, a multiple of 16
constexpr int kTilePxF64 = kTilePx - kMathTabPx, kTilePxBigF64 = kTilePxBig - kMathTabPx;
#ifndef RBS_BIG_CAP
#define RBS_BIG_CAP 256
#endif
constexpr int kBigCap = RBS_BIG_CAP;       // triangles deferred to the cooperative path per chunk
constexpr int kBigThresh = RBS_BIG_THRESH; // bbox pixels above which a triangle is "big"
constexpr int kCopyUnroll = RBS_COPY_UNROLL; // float4 loads in flight per lane in copy blocks
#ifndef RBS_SCAN_UNROLL
#define RBS_SCAN_UNROLL 2
#endif
#ifndef RBS_SCAN_UNROLL_F64
#define RBS_SCAN_UNROLL_F64 2
#endif
constexpr int kQPlanes_ = 4;
constexpr int kEvalQueue = 128;             // per-wave ring of covered pixels awaiting evaluation (a power of two)
constexpr int kTq = kQPlanes_ * kEvalQueue;  // the same LDS during the raster phase: a per-wave ring of triangle indices
// Three raster waves of 160 registers leave a SIMD the 32 that one copy wave needs (both
// precisions): the budget is stated to the compiler, which otherwise only aims at "three waves" = 168.
#ifndef RBS_RASTER_VGPRS
#define RBS_RASTER_VGPRS 80   // (the attribute counts architectural registers and the compiler doubles it on gfx90a+: 160 in all)
#endif
#ifndef RBS_PRETEST_CLUSTERS
#define RBS_PRETEST_CLUSTERS 4
#endif
constexpr int kPre = RBS_PRETEST_CLUSTERS;  // clusters pre-tested per step
static_assert(kPre * 64 + 63 <= kTq && (kTq & (kTq - 1)) == 0, "the triangle ring must hold a step's survivors");
// Meshes of many clusters (M4: 795; the rbs_raster_kernel_many_* instantiations, MANY): the cluster cull -- the same for
// every wave of the block -- is dealt to the block's waves by steps of 64 clusters and the verdicts shared through LDS
// (two masks per step): C4 slice +6 %.  Kernels of their own: the same code compiled into the one kernel cost C1 0.8 %.
constexpr int kCullSteps = 28;              // steps of 64 clusters culled between two barriers: 448 B of verdicts, what the LDS block had
                                            // left below the next 1 280-byte step (three blocks per CU, RBS_TILE_PX)
constexpr int kQPlanes = kQPlanes_;                 // ints per queued pixel: index, depth, prior, observation (F32 precision)
constexpr int kRectAlign = 16;              // whole planes: rectangle x-alignment in pixels (64 B)
constexpr float kSnapTau = 0x1p-18f;         // background snap of the occlusion process (oracle ORC_SNAP_TAU)
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr unsigned kInfBits = 0x7f800000u;
constexpr int kMaxBodies = 16;
constexpr int kMaxDevices = 8;
constexpr double kMaxDepth = 6.0;    // KinectPixelModel max_depth
constexpr double kHalfLifeDepth = 1.0;

// Several bodies: the screen rectangles of the bodies of a particle, overlapping ones merged, at
// most kMaxGroups of them (more: one union rectangle).  Every group is rasterized on its own --
// its own work items, its own bodies only -- so the gaps between objects that stand apart are
// neither scanned nor culled against; where rectangles overlap the bodies share a z-buffer as
// before.  The copy kernel writes what lies inside the union rectangle but in no group.
constexpr int kMaxGroups = 4;
struct Groups {
    int n, pad0, pad1, pad2;
    int4 rect[kMaxGroups];
    int first[kMaxGroups];        // first work item of the group, relative to the particle's first
    unsigned mask[kMaxGroups];    // bodies of the group
};

// Several bodies, windowed planes: what the copy kernel writes of a particle's region is the region MINUS its groups' rectangles.
// The rectangles kernel cuts that into STRIPS -- bands between the rectangles' top and bottom edges, within a band the intervals no
// rectangle covers -- so that the copy kernel enumerates the cells it writes instead of walking the whole region and skipping the
// rectangles' share (C2: 55 % of the cells).  At most (2 kMaxGroups + 1) bands of (kMaxGroups + 1) intervals.
constexpr int kMaxStrips = (2 * kMaxGroups + 1) * (kMaxGroups + 1);
struct Strips {
    int n, pad0, pad1, pad2;
    int first[kMaxStrips + 1];       // first cell (float4) of each strip in the particle's enumeration; [n] = their number
    ushort4 box[kMaxStrips];         // x0 / 4, x1 / 4, y0, y1 (columns and rows <= 8 192)
};

struct DevParams {
    int rows, cols, npx;
    int n_bodies;
    int n_tri;                     // soup length: every body padded to a multiple of 64
    int tri_begin[kMaxBodies + 1]; // triangle range per body (multiples of 64)
    int tri_end[kMaxBodies];       // end of the body's REAL triangles (the rest of its last cluster is NaN padding)
    const floatx4* tri_plane;      // [n_tri] model-space plane of each triangle: unit normal of its winding, offset
                                   //   (n.x + d = signed distance of x from the plane); NaN for padding
    const floatx4* vtx;            // vertices of every body (x, y, z, 0), float32: the screen rectangle only
    int vtx_begin[kMaxBodies + 1]; // vertex range per body
    const float* cluster_sphere;   // [n_tri/64][4] model-space bounding sphere of each cluster
    const float* cluster_cone;     // [n_tri/64][4] outward-normal cone of each cluster: unit axis, min cos
                                   //   (min cos <= -1: never cull this cluster)
    // Vertex sharing inside a cluster (raster_shared_cluster): the cluster's UNIQUE vertices and,
    // per triangle, the three positions of its vertices in that list.
    const double* cluster_vtx;     // [n_tri/64][3][64] x / y / z of the cluster's unique vertices (the soup's own values)
    const int* cluster_nv;         // [n_tri/64] number of unique vertices, 0: more than 64 (the cluster is set up per triangle)
    const unsigned* tri_local;     // [n_tri] i0 | i1 << 8 | i2 << 16, 0xffffffff for padding
    int body_cull[kMaxBodies];     // 0: keep every triangle; +1/-1: the body is a closed, consistently
                                   //   oriented surface (sign of its signed volume): back faces may go
    int tile_px;                   // pixels of this launch's LDS depth tile (its dynamic shared memory is sized for it)
    int rect_align;                // rectangle x-alignment in pixels: 16 whole planes, 4 windowed (float4)
    int tile_w, tile_h;            // work-item tile limits: width <= tile_w, pixels <= min(tile_w*tile_h, kTilePx)
    double fx, fy, cx, cy;
    double sphere[kMaxBodies][4];  // model-space bounding sphere: centre xyz, radius
    const double* soup;            // SoA [9][n_tri]: v0.xyz v1.xyz v2.xyz
    const float* frame;            // observation, float metres
    const double* aux;             // per-frame-pixel terms, [npx][4] binary64 (frame_aux_kernel)
    const float* pbg;              // per-frame-pixel background density, rounded to float
    double tw, ms, sf, lambda;     // tail_weight, model_sigma, sigma_factor, ln2/half_life
    double cv0;                    // (1 - tail_weight) / sqrt(pi): c_v = cv0 / (sqrt2 sigma)
    float alpha, beta;             // occlusion process over the elapsed frames
    float bg_old, bg_new;          // never-covered level before / after this call's step
    int windowed;                  // planes are valid inside their window only (else implicit bg)
    const int4* win_src;           // [slots] window of each parent plane (x0,y0,x1,y1)
    int4* win_dst;                 // [slots] window of each child plane: tight bbox, grown atomically
    int4* win_used;                // [n] region the copy kernel writes: bbox(parent window, rect)
    unsigned long long* area_sum;  // sampled calls: sum over particles of |win_used| in pixels (else null)
    unsigned char* wide_flags;     // wide windows: [n][blocks per plane] "this copy block wrote a non-background value"
    // Several devices in one handle (rbs_config.n_devices > 1): parent slots are GLOBAL, slot g lives
    // on device g / shard_cap at local slot g % shard_cap, and a parent on another device is read
    // in place over xGMI (peer access) -- its window's ~4 % of a plane, once.  n_dev <= 1: the
    // tables are unused.
    int n_dev, shard_cap;
    const float* occ_src_dev[kMaxDevices];   // current planes of every device of the handle
    const int4* win_src_dev[kMaxDevices];    // ... and their windows
    const int4* reg_src_dev[kMaxDevices];    // ... and (slabs) their stored regions
    // Window-sized slabs (rbs_config.state_slab_px > 0; windowed planes only): a slot holds
    // slab_px floats instead of a whole plane and stores its REGION reg = (x0, y0, x1, y1) row-major
    // with stride x1 - x0 -- the region an updating call writes, bbox(parent window, rectangle),
    // which contains the plane's window.  A child whose region does not fit its slab is contained
    // like a bad parent slot (log-likelihood NaN, empty plane) and raises the sticky err flag.
    int slab_px;                   // 0: whole planes (a slot = npx floats, stride cols)
    int plane_stride;              // floats per slot: npx or slab_px
    const int4* reg_src;           // [slots] stored region of each parent plane
    int4* reg_dst;                 // [slots] stored region of each child plane (= win_used of this call)
    int* err;                      // [2] sticky: [0] 1 = a particle's region did not fit its slab, [1] the largest region asked for (px)
    const float* occ_src;          // [slots][npx]
    float* occ_dst;                // [slots][npx]
    const double* poses;           // [n][n_bodies][12]
    const double* poses_src;       // host-pointer calls: pinned host memory the rectangles kernel copies the poses from (into `poses`)
    const double* deltas_src;      // rbs_loglikes_deltas: pinned [n][bodies][6] state deltas followed by [bodies][6] default poses (position,
                                   //   rotation vector) -- the rectangles kernel COMPOSES the absolute poses from them (into `poses`)
    const int* indices;            // [n] parent slot as the caller passed it: read by the rectangles kernel ONLY,
                                   //   in the caller's stream order (the caller may rewrite it right after the call)
    int* parents;                  // [n] the rectangles kernel's snapshot of `indices`, double-buffered across calls
                                   //   like `rects`: what the raster and the (side-stream) copy kernels read
    int slots;                     // occlusion slots allocated (valid parents: 0..slots-1)
    double* out;                   // [n]
    int n;
    int bands, band_rows;          // copy blocks per particle, rows per band
    const int* rects;              // [n][4] screen rectangles (rbs_prep_kernel): the union over the bodies
    Groups* groups;                // [n] per-group rectangles, several bodies only (null for one body)
    Strips* strips;                // [n] the copy kernel's cells, several bodies on windowed planes (null otherwise)
    int2* item_range;              // [n] (first work item, number of work items) of each particle
    int* item_particle;            // [items] owner of each work item
    int* ctr_this;                 // [2] this call's (items allotted, items taken) counters ...
    int* ctr_next;                 // [2] ... and the next call's, zeroed by this call's raster kernel
    double* partial;               // [items] partial log-likelihood per work item
    int* done;                     // [n] finished work items per particle (the last one sums them)
    // Split launch (rbs_depth_kernel -> rbs_eval_kernel, round 5): the depth tile of work item k, handed from the
    // geometry kernel to the likelihood kernel through memory (L2 / MALL sized: ~23 KB per C1 item).
    // Shared background PLANE (round 5; handles whose windows have grown, see rbsensor_capi.hip "shared trail"): outside its
    // window a plane equals bgp_src[pixel] instead of the scalar bg_old -- the trail every particle inherited from a common
    // ancestor is stored ONCE.  bgp_dst = the same plane after this call's step (what the children's windows are measured
    // against).  Null: the scalar background.
    const float* bgp_src;
    const float* bgp_dst;
    const int4* rebase_box;        // null, or the rectangle every child's region additionally covers in this call: the window of the
                                   // plane the shared plane is re-based on -- or, when the handle goes back to the scalar background,
                                   // the bounding box of the shared plane's own values
    unsigned* depth;               // [depth_items][tile_px] order-preserving float bits, kInfBits = not covered
    int depth_items;               // items the buffer holds (an item beyond it is contained: its particle's sum is NaN)
    int* ctrb_this;                // [1] the likelihood kernel's ticket counter of this call ...
    int* ctrb_next;                // ... and the next call's, zeroed by this call's geometry kernel
    // STAMPED planes (round 6; rbs_config.occlusion_mode = RBS_OCC_REFERENCE: the reference CPU model's own occlusion bookkeeping,
    // SURVEY A.4 / A.5, oracle mode LAZY).  A slot holds plane_px floats -- the posterior a pixel was last UPDATED to, never
    // stepped -- followed by plane_px 16-bit AGES: frames between that update and the slot's epoch (the last updating call).
    // The prior of a pixel at a call `elapsed` frames after the epoch is propagate(value, (age + elapsed) dt) in binary64 from
    // the table ptab (the oracle's operations in the oracle's order), rounded once to float; an age beyond age_max IS the
    // background -- a never-covered pixel, prior bg_new = (float)propagate(initial_occlusion_prob, clock dt) -- so a window
    // still follows the object (c^(age_max dt) <= 2^-40: the value would differ from the background by less than that).
    // Ages saturate at 0xffff.  Everything above (windows, slabs, shared plane, peers) addresses slots as before.
    int exact;
    int plane_px;                  // pixels per slot: npx, or slab_px (the ages of a slot start at its float plane_px)
    unsigned elapsed2;             // frames since the epoch, saturated to 16 bits, in both halves of a dword (packed adds)
    int age_max;                   // ages above it are background
    const double* ptab;            // [age_max + 1][2]: c^(age dt), ((1 - p_oo) (c^(age dt) - 1)) / (c - 1)
#ifdef RBS_PHASE_TIMING
    unsigned long long* phase;     // [32] accumulated wave-0 cycles per phase and event counts (profiling builds only)
#endif
};
#ifdef RBS_PHASE_TIMING
// (accumulated in LDS, flushed once per block: a global atomic per tick serialised the blocks)
__shared__ unsigned long long g_phase_lds[32];   // [0, 16): cycles per phase, [16, 32): event counts (RBS_COUNT)
#define RBS_TICK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); g_phase_lds[k] += t_ - tick_; tick_ = t_; } } while (0)
#define RBS_COUNT(k, v) do { if (threadIdx.x == 0) g_phase_lds[k] += (unsigned long long)(v); } while (0)
#define RBS_TICK_DECL unsigned long long tick_ = clock64()
#define RBS_TICK_PARAM , unsigned long long& tick_
#define RBS_TICK_ARG , tick_
#else
#define RBS_TICK(k) do {} while (0)
#define RBS_COUNT(k, v) do {} while (0)
#define RBS_TICK_DECL do {} while (0)
#define RBS_TICK_PARAM
#define RBS_TICK_ARG
#endif

struct Rect { int x0, y0, x1, y1; };

// How a rectangle of rw x rh pixels splits into work-item tiles: nx columns of equal 16-aligned
// width <= max_w, then as few equal rows as keep a tile within cap_px pixels.  A rectangle that
// fits one tile -- the usual case -- is one item whatever its aspect.
struct TileGrid { int tw, th, nx, ny; };
__host__ __device__ inline TileGrid tile_grid(int rw, int rh, int max_w, int cap_px)
{
    TileGrid g;
    g.nx = (rw + max_w - 1) / max_w;
    if (g.nx < 1) g.nx = 1;
    g.tw = ((rw + g.nx - 1) / g.nx + 15) & ~15;
    if (g.tw < 16) g.tw = 16;
    int hmax = cap_px / g.tw;
    if (hmax < 1) hmax = 1;
    g.ny = (rh + hmax - 1) / hmax;
    if (g.ny < 1) g.ny = 1;
    g.th = (rh + g.ny - 1) / g.ny;
    if (g.th < 1) g.th = 1;
    return g;
}

// The occlusion process on one stored value: affine step, then the background snap (same rule
// and constant as oracle orc_eager_prior).
__device__ inline float occ_step(float alpha, float beta, float v, float bg_new)
{
    const float x = fmaf(alpha, v, beta);
    return fabsf(x - bg_new) <= kSnapTau ? bg_new : x;
}

// Stamped planes: two 16-bit ages in a dword advanced by the call's elapsed frames, saturating at 0xffff (v_pk_add_u16 clamp).
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
constexpr unsigned kAgeBg2 = 0xffffffffu;    // two background ages
__device__ inline unsigned age_add2(unsigned packed, unsigned e2)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(ushort2v, packed), __builtin_bit_cast(ushort2v, e2)));
}
// The reference's prior of a pixel (oracle orc_propagate): the stored posterior v, last updated `age` frames ago
// (this call's elapsed frames included).  pow_c = exp(age dt log c) and the second summand come from the host's table, computed
// with the oracle's expressions; the three operations left are the oracle's, in its order, individually rounded.
__device__ inline float exact_prior(const DevParams& P, float v, int age)
{
    typedef double doublex2 __attribute__((ext_vector_type(2)));
    const doublex2 t = reinterpret_cast<const doublex2*>(P.ptab)[min(age, P.age_max)];
    const double new_visible = t.x * (1.0 - (double)v) + t.y;
    const float pr = (float)(1.0 - new_visible);
    return age > P.age_max ? P.bg_new : pr;
}
// Does pixel (v, a) of a plane differ from pixel (bv, ba) of the shared background plane?  (Background ages carry no value.)
__device__ inline bool exact_differs(float v, unsigned a, float bv, unsigned ba, unsigned age_max)
{
    return a > age_max ? ba <= age_max : (a != ba || v != bv);
}

// Where a parent slot's plane and window are: this device's buffers, or a peer's (see DevParams).
__device__ inline const float* parent_plane(const DevParams& P, int parent)
{
    if (P.n_dev <= 1) return P.occ_src + (size_t)parent * P.plane_stride;
    const int d = parent / P.shard_cap;
    return P.occ_src_dev[d] + (size_t)(parent - d * P.shard_cap) * P.plane_stride;
}
// A plane as the kernels address it: the value of pixel (x, y) is base[(y - y0) * stride + (x - x0)].
// Whole planes: y0 = x0 = 0, stride = cols.
struct PlaneRef { int x0, y0, stride; };
__device__ inline PlaneRef parent_ref(const DevParams& P, int parent)
{
    PlaneRef r = {0, 0, P.cols};
    if (P.slab_px) {
        int4 g;
        if (P.n_dev <= 1) g = P.reg_src[parent];
        else { const int d = parent / P.shard_cap; g = P.reg_src_dev[d][parent - d * P.shard_cap]; }
        r.x0 = g.x; r.y0 = g.y; r.stride = max(g.z - g.x, 0);
    }
    return r;
}
__device__ inline PlaneRef child_ref(const DevParams& P, int particle)
{
    PlaneRef r = {0, 0, P.cols};
    if (P.slab_px) { const int4 g = P.win_used[particle]; r.x0 = g.x; r.y0 = g.y; r.stride = max(g.z - g.x, 0); }
    return r;
}
__device__ inline int4 parent_window(const DevParams& P, int parent)
{
    if (P.n_dev <= 1) return P.win_src[parent];
    const int d = parent / P.shard_cap;
    return P.win_src_dev[d][parent - d * P.shard_cap];
}

// ------------------------------------------------------------------ screen rectangle
// Pixel rectangle containing every pixel the particle's bodies can cover: the bounding box of the
// projected VERTICES (a triangle wholly in front of the camera projects inside the hull of its
// projected vertices), computed by one wave per particle, lanes striding over the vertices, in
// float32 with a margin that covers the float rounding (a few 1e-3 px) -- the rasterizer's own
// binary64 bounding boxes therefore lie inside it.  x-aligned to rect_align pixels so raster and
// copy blocks split rows at float4 / 64-byte boundaries.  The result only decides WHO writes a
// pixel, never its value.  (Up to round 1 the rectangle came from the bounding sphere and the
// corners of the bounding box: 9 200 px for the 5 120-triangle ellipsoid at 0.7 m where this
// one has 5 700 -- the pixel pass, the tile and the stored windows shrink by as much.)
// Called by all 64 lanes of a wave; every lane returns the same rectangle.
__device__ inline Rect bodies_rect(const DevParams& P, const double* __restrict__ pose, int b_first, int b_last)
{
    const int lane = threadIdx.x & 63, stride = 64;
    const float fx = (float)P.fx, fy = (float)P.fy, cx = (float)P.cx, cy = (float)P.cy;
    float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY, zmin = INFINITY, tabs = 0.f;
    for (int b = b_first; b < b_last; ++b) {
        const double* Rt = pose + 12 * b;
        const float r0 = (float)Rt[0], r1 = (float)Rt[1], r2 = (float)Rt[2], r3 = (float)Rt[3], r4 = (float)Rt[4],
                    r5 = (float)Rt[5], r6 = (float)Rt[6], r7 = (float)Rt[7], r8 = (float)Rt[8];
        const float tx = (float)Rt[9], ty = (float)Rt[10], tz = (float)Rt[11];
        tabs = fmaxf(tabs, fabsf(tx) + fabsf(ty) + fabsf(tz));
        // eight vertex loads in flight per lane (the vertices sit in L2; one dependent load per
        // trip would leave this kernel, which the raster kernel waits for, latency bound)
#ifndef RBS_RECT_LOADS
#define RBS_RECT_LOADS 8
#endif
        constexpr int kV = RBS_RECT_LOADS;
        const int v1 = P.vtx_begin[b + 1];
        for (int i0 = P.vtx_begin[b] + lane; i0 < v1; i0 += stride * kV) {
            floatx4 pv[kV];
#pragma unroll
            for (int k = 0; k < kV; ++k) pv[k] = P.vtx[min(i0 + stride * k, v1 - 1)];   // the tail repeats the last vertex
#pragma unroll
            for (int k = 0; k < kV; ++k) {
                const floatx4 p = pv[k];
                const float X = fmaf(r0, p.x, fmaf(r1, p.y, fmaf(r2, p.z, tx)));
                const float Y = fmaf(r3, p.x, fmaf(r4, p.y, fmaf(r5, p.z, ty)));
                const float Z = fmaf(r6, p.x, fmaf(r7, p.y, fmaf(r8, p.z, tz)));
                const float iz = __builtin_amdgcn_rcpf(Z);
                const float u = fmaf(fx, X * iz, cx), v = fmaf(fy, Y * iz, cy);
                zmin = fminf(zmin, Z);     // NaN poses: fminf/fmaxf drop the NaN, zmin stays +inf ...
                umin = fminf(umin, u); umax = fmaxf(umax, u);
                vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        umin = fminf(umin, __shfl_xor(umin, off)); umax = fmaxf(umax, __shfl_xor(umax, off));
        vmin = fminf(vmin, __shfl_xor(vmin, off)); vmax = fmaxf(vmax, __shfl_xor(vmax, off));
        zmin = fminf(zmin, __shfl_xor(zmin, off));
        tabs = fmaxf(tabs, __shfl_xor(tabs, off));
    }
    Rect r;
    // a vertex at or behind the camera plane (the rasterizer drops such triangles, but their
    // neighbours may project anywhere), or a pose that is not finite: the whole frame
    const bool full = !(zmin > 1e-4f) || !(zmin < INFINITY) || !(tabs < INFINITY) ||
                      !(umax - umin < INFINITY) || !(vmax - vmin < INFINITY);   // ... and the extents are -inf
    if (full) {
        r.x0 = 0; r.y0 = 0; r.x1 = P.cols; r.y1 = P.rows;
    } else {
        // float rounding of X, Y, Z (inputs rounded to float + three fmas each), of the reciprocal
        // (1 ulp) and of the projection
        const float mx = 4e-6f * fx * (tabs + 1.0f) / zmin + 3e-7f * (fabsf(umin - cx) + fabsf(umax - cx)) + 1e-3f;
        const float my = 4e-6f * fy * (tabs + 1.0f) / zmin + 3e-7f * (fabsf(vmin - cy) + fabsf(vmax - cy)) + 1e-3f;
        const float W = (float)P.cols, H = (float)P.rows;
        r.x0 = (int)fminf(fmaxf(floorf(umin - mx), 0.0f), W);
        r.x1 = (int)fminf(fmaxf(ceilf(umax + mx) + 1.0f, 0.0f), W);
        r.y0 = (int)fminf(fmaxf(floorf(vmin - my), 0.0f), H);
        r.y1 = (int)fminf(fmaxf(ceilf(vmax + my) + 1.0f, 0.0f), H);
    }
    r.x0 &= ~(P.rect_align - 1);
    r.x1 = min(P.cols, (r.x1 + P.rect_align - 1) & ~(P.rect_align - 1));
    if (r.x1 <= r.x0 || r.y1 <= r.y0) { r.x0 = r.x1 = r.y0 = r.y1 = 0; }
    return r;
}
__device__ inline Rect particle_rect(const DevParams& P, const double* __restrict__ pose)
{
    return bodies_rect(P, pose, 0, P.n_bodies);
}

// Binary64 division as the compiler expands it -- hardware reciprocal, two Newton steps, quotient,
// exact remainder, one correcting FMA (Markstein) -- WITHOUT the v_div_scale / v_div_fmas /
// v_div_fixup wrapping that rescales operands near the ends of the exponent range and patches
// infinities, zeros and NaNs: 9 instructions instead of 13, the same bits whenever the scale
// factors would have been 1 (operands and quotient comfortably normal: the rasterizer's depths,
// focal lengths and plane coefficients).  At the extremes (a denominator that is zero, denormal
// or beyond 2^1000) the result may be inf / NaN / off where IEEE gives a finite huge or tiny
// number: every use below rejects those values on both sides (a depth must be a positive finite
// float, a bounding box must not be empty or NaN), so outcomes stay identical to the oracle's.
// RBS_IEEE_DIV=1 restores the plain operator (A/B builds).
#ifndef RBS_IEEE_DIV
#define RBS_IEEE_DIV 0
#endif
__device__ inline double div_f64(double a, double b)
{
#if RBS_IEEE_DIV
    return a / b;
#else
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
#endif
}

// ------------------------------------------------------------------ triangle setup
struct Tri {
    double u0, v0, u1, v1, u2, v2;
    double e01u, e01v, e12u, e12v, e20u, e20v;
    double pa, pb, pc, nv0;
    int xlo, xhi, ylo, yhi;
    // float32 FILTER of the three edge functions over the bounding box (tri_filter): sample (i, j)
    // of the box has E_k ~ fB[k] i + fA[k] j + fC[k] with |error| < thr
    float fA[3], fB[3], fC[3], thr;
};

// The coverage rule is a sign test on three binary64 edge functions, 21 binary64 instructions per
// sample -- and a binary64 instruction holds the VALU 1.65x as long as a float32 one on this chip
// (profiles/r03_valu_issue_microbench.txt).  An exact floating-point FILTER decides almost every
// sample in float32 instead: the edge functions are evaluated in box-local coordinates (vertex
// minus the box's corner, rounded to float: magnitudes of a few pixels, not hundreds) with a
// rigorous bound `thr` on their distance from the oracle's binary64 values; all three clearly
// positive or all clearly negative -> covered, one clearly positive and one clearly negative ->
// not covered, anything else (a sample within `thr` of an edge: about one in 1e4, or a non-finite
// intermediate) -> the oracle's own binary64 expression decides (tri_inside_exact).  Coverage is
// therefore bit-exact by construction.
// Bound: with L = max(box width, box height) + 1 every local coordinate, edge component and sample
// offset is <= 2L in magnitude; the local vertices carry a relative rounding u = 2^-24, the edges
// 2 more, each product and sum one more: |E_float - E_exact| <= 32 u L^2 (generous), and the
// oracle's binary64 value is within 1e-13 L^2 of E_exact.  thr = 64 u L^2.
__device__ inline void tri_filter(Tri& T, double xlo_d, double ylo_d)
{
    const float lu0 = (float)(T.u0 - xlo_d), lv0 = (float)(T.v0 - ylo_d);
    const float lu1 = (float)(T.u1 - xlo_d), lv1 = (float)(T.v1 - ylo_d);
    const float lu2 = (float)(T.u2 - xlo_d), lv2 = (float)(T.v2 - ylo_d);
    // E_k(i, j) = eu_k (j - lv_k) - ev_k (i - lu_k)
    const float eu0 = lu1 - lu0, ev0 = lv1 - lv0;
    const float eu1 = lu2 - lu1, ev1 = lv2 - lv1;
    const float eu2 = lu0 - lu2, ev2 = lv0 - lv2;
    T.fA[0] = eu0; T.fB[0] = -ev0; T.fC[0] = fmaf(ev0, lu0, -(eu0 * lv0));
    T.fA[1] = eu1; T.fB[1] = -ev1; T.fC[1] = fmaf(ev1, lu1, -(eu1 * lv1));
    T.fA[2] = eu2; T.fB[2] = -ev2; T.fC[2] = fmaf(ev2, lu2, -(eu2 * lv2));
    const float L = (float)(max(T.xhi - T.xlo, T.yhi - T.ylo) + 2);
    T.thr = (64.0f * 0x1p-24f) * L * L;
}

// Same operations, same order as oracle/rbsensor_oracle.c raster_triangle().  The clip window
// [wx0,wx1) x [wy0,wy1) is a sub-rectangle of the image, so clipping to it instead of to the
// image changes nothing inside the window.
// cullsign != 0: the body is a closed surface wholly in front of the camera and cullsign is its
// orientation; a triangle whose projected area has that sign faces away from the camera and is
// dropped -- every sample it covers is covered, no farther away, by a front face (see
// raster_window).
// The second half of the setup, from the camera-space vertices X, Y, Z and their projections u, v.
__device__ inline bool tri_finish(const DevParams& P, const double* X, const double* Y, const double* Z, const double* u,
                                  const double* v, int wx0, int wy0, int wx1, int wy1, int cullsign, Tri& T)
{
    if (!(Z[0] > 0.0 && Z[1] > 0.0 && Z[2] > 0.0)) return false;
    // bbox first: most sub-pixel triangles contain no integer sample point and leave here
    // (pure reordering of independent operations -- values are unchanged)
    const double umin = fmin(fmin(u[0], u[1]), u[2]), umax = fmax(fmax(u[0], u[1]), u[2]);
    const double vmin = fmin(fmin(v[0], v[1]), v[2]), vmax = fmax(fmax(v[0], v[1]), v[2]);
    const double xlo_d = fmax(ceil(umin), (double)wx0), xhi_d = fmin(floor(umax), (double)(wx1 - 1));
    const double ylo_d = fmax(ceil(vmin), (double)wy0), yhi_d = fmin(floor(vmax), (double)(wy1 - 1));
    if (!(xlo_d <= xhi_d) || !(ylo_d <= yhi_d)) return false;

    T.u0 = u[0]; T.v0 = v[0]; T.u1 = u[1]; T.v1 = v[1]; T.u2 = u[2]; T.v2 = v[2];
    T.e01u = u[1] - u[0]; T.e01v = v[1] - v[0];
    T.e12u = u[2] - u[1]; T.e12v = v[2] - v[1];
    T.e20u = u[0] - u[2]; T.e20v = v[0] - v[2];
    const double area2 = T.e01u * (v[2] - v[0]) - T.e01v * (u[2] - u[0]);
    if (!(area2 != 0.0) || !(fabs(area2) < INFINITY)) return false;
    if (cullsign != 0 && (cullsign > 0 ? area2 > 0.0 : area2 < 0.0)) return false;

    const double ax = X[1] - X[0], ay = Y[1] - Y[0], az = Z[1] - Z[0];
    const double bx = X[2] - X[0], by = Y[2] - Y[0], bz = Z[2] - Z[0];
    const double nx = ay * bz - az * by;
    const double ny = az * bx - ax * bz;
    const double nz = ax * by - ay * bx;
    T.nv0 = (nx * X[0] + ny * Y[0]) + nz * Z[0];
    T.pa = div_f64(nx, P.fx);
    T.pb = div_f64(ny, P.fy);
    T.pc = (nz - T.pa * P.cx) - T.pb * P.cy;
    T.xlo = (int)xlo_d; T.xhi = (int)xhi_d; T.ylo = (int)ylo_d; T.yhi = (int)yhi_d;
#if RBS_EDGE_FILTER
    tri_filter(T, xlo_d, ylo_d);
#endif
    return true;
}

// One vertex into camera space and onto the image plane: the oracle's operations in the oracle's
// order (oracle/rbsensor_oracle.c raster_triangle()).  The division is carried out whatever Z is:
// tri_finish rejects a triangle with a vertex at Z <= 0 before it looks at u, v.
__device__ inline void vertex_project(const DevParams& P, const double* __restrict__ Rt, double vx, double vy, double vz,
                                      double& X, double& Y, double& Z, double& u, double& v)
{
    X = ((Rt[0] * vx + Rt[1] * vy) + Rt[2] * vz) + Rt[9];
    Y = ((Rt[3] * vx + Rt[4] * vy) + Rt[5] * vz) + Rt[10];
    Z = ((Rt[6] * vx + Rt[7] * vy) + Rt[8] * vz) + Rt[11];
    const double iz = div_f64(1.0, Z);
    u = P.fx * (X * iz) + P.cx;
    v = P.fy * (Y * iz) + P.cy;
}

__device__ inline bool tri_setup(const DevParams& P, int t, const double* __restrict__ Rt,
                                 int wx0, int wy0, int wx1, int wy1, int cullsign, Tri& T)
{
    const double* __restrict__ s = P.soup;
    const size_t n = (size_t)P.n_tri;
    // all nine coalesced loads in flight together (one memory round trip per triangle, not three)
    double vv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) vv[k] = s[k * n + t];
    __builtin_amdgcn_sched_barrier(0);
    double X[3], Y[3], Z[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vx = vv[3 * k + 0], vy = vv[3 * k + 1], vz = vv[3 * k + 2];
        X[k] = ((Rt[0] * vx + Rt[1] * vy) + Rt[2] * vz) + Rt[9];
        Y[k] = ((Rt[3] * vx + Rt[4] * vy) + Rt[5] * vz) + Rt[10];
        Z[k] = ((Rt[6] * vx + Rt[7] * vy) + Rt[8] * vz) + Rt[11];
    }
    if (!(Z[0] > 0.0 && Z[1] > 0.0 && Z[2] > 0.0)) return false;
    double u[3], v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double iz = div_f64(1.0, Z[k]);
        u[k] = P.fx * (X[k] * iz) + P.cx;
        v[k] = P.fy * (Y[k] * iz) + P.cy;
    }
    return tri_finish(P, X, Y, Z, u, v, wx0, wy0, wx1, wy1, cullsign, T);
}

// The oracle's coverage rule on one integer sample point (px, py as doubles): all three edge
// functions >= 0 or all three <= 0.
__device__ inline bool tri_inside_exact(const Tri& T, double px, double py)
{
    const double E0 = T.e01u * (py - T.v0) - T.e01v * (px - T.u0);
    const double E1 = T.e12u * (py - T.v1) - T.e12v * (px - T.u1);
    double mn = fmin(E0, E1), mx = fmax(E0, E1);   // (folded as they come: two values live, not three)
    const double E2 = T.e20u * (py - T.v2) - T.e20v * (px - T.u2);
    mn = fmin(mn, E2);
    mx = fmax(mx, E2);
    // all three >= 0 or all three <= 0, as min / max (E is finite: the setup rejected anything else):
    // four instructions and one branch instead of two compare chains with a branch between them.
    return ((int)(mn >= 0.0) | (int)(mx <= 0.0)) != 0;
}
// Depth of a covered sample; z-min into the LDS tile at `at`.
__device__ inline void tri_depth(const Tri& T, double px, double py, unsigned* at)
{
    const double den = (T.pa * px + T.pb * py) + T.pc;
    const float zf = (float)div_f64(T.nv0, den);
    // a depth must be a positive finite float (zf > 0 && zf < inf): one class test, +denormal | +normal
    if (__builtin_amdgcn_classf(zf, 0x180)) atomicMin(at, __float_as_uint(zf));
}
// Coverage test + depth for one integer pixel (px, py: its coordinates as doubles); z-min into the
// LDS tile at `at`.
__device__ inline void tri_sample(const Tri& T, double px, double py, unsigned* at)
{
    // With three waves per SIMD a branch in this loop costs the wave more than the instructions it
    // guards: 3 % of the kernel.
    if (!tri_inside_exact(T, px, py)) return;
    const double den = (T.pa * px + T.pb * py) + T.pc;
    const float zf = (float)div_f64(T.nv0, den);
    // a depth must be a positive finite float (zf > 0 && zf < inf): one class test, +denormal | +normal
    if (__builtin_amdgcn_classf(zf, 0x180)) atomicMin(at, __float_as_uint(zf));
}
__device__ inline void tri_pixel(const Tri& T, int col, int row, unsigned* tile, int tw, int wx0, int wy0)
{
    tri_sample(T, (double)col, (double)row, &tile[(row - wy0) * tw + (col - wx0)]);
}

__device__ inline int body_of(const DevParams& P, int t)
{
    int b = 0;
    while (b + 1 < P.n_bodies && t >= P.tri_begin[b + 1]) ++b;
    return b;
}

// The cluster tests below are conservative with margins of 1e-3: the hardware square root (1 ulp)
// serves them; the correctly rounded one the build asks for elsewhere costs ~15 instructions.
__device__ inline float cone_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Conservative float32 test: can any triangle of the cluster (model-space bounding sphere
// c, rho) touch the pixel window [wx0,wx1) x [wy0,wy1)?  The window's four frustum planes
// pass through the camera centre; the sphere is culled only if it lies entirely outside one
// of them, with the window widened by 2 px and the radius by 1 mm + 0.1 % so float rounding
// can never cull a cluster the binary64 rasterizer would have touched.  Wave-uniform.
__device__ inline bool cluster_may_touch(const DevParams& P, const double* __restrict__ Rt,
                                         const floatx4 sph, int wx0, int wy0, int wx1,
                                         int wy1)
{
    const float sx = sph.x, sy = sph.y, sz = sph.z;
    const float rho = sph.w * 1.001f + 1e-3f;
    const float X = (float)Rt[0] * sx + (float)Rt[1] * sy + (float)Rt[2] * sz + (float)Rt[9];
    const float Y = (float)Rt[3] * sx + (float)Rt[4] * sy + (float)Rt[5] * sz + (float)Rt[10];
    const float Z = (float)Rt[6] * sx + (float)Rt[7] * sy + (float)Rt[8] * sz + (float)Rt[11];
    if (!(Z - rho > 1e-3f)) return Z + rho > 0.0f;  // straddles the camera plane: keep unless fully behind
    const float fx = (float)P.fx, fy = (float)P.fy, cx = (float)P.cx, cy = (float)P.cy;
    // u >= a  <=>  fx*X + (cx-a)*Z >= 0  (Z > 0); signed distance to that plane = (.)/|n|
    const float al = cx - ((float)wx0 - 2.0f), ar = cx - ((float)wx1 + 1.0f);
    const float at = cy - ((float)wy0 - 2.0f), ab = cy - ((float)wy1 + 1.0f);
    // signed distance to a plane < -rho  <=>  s < 0 and s^2 > rho^2 |n|^2  (no square root)
    const float r2 = rho * rho;
    bool out = false;
    { const float s_ = fx * X + al * Z; out |= s_ < 0.0f && s_ * s_ > r2 * (fx * fx + al * al); }   // left of the window
    { const float s_ = fx * X + ar * Z; out |= s_ > 0.0f && s_ * s_ > r2 * (fx * fx + ar * ar); }   // right
    { const float s_ = fy * Y + at * Z; out |= s_ < 0.0f && s_ * s_ > r2 * (fy * fy + at * at); }   // above
    { const float s_ = fy * Y + ab * Z; out |= s_ > 0.0f && s_ * s_ > r2 * (fy * fy + ab * ab); }   // below
    if (out) return false;
    return true;
}

// Conservative float32 test: does every triangle of the cluster face away from the camera?
// The cluster's outward normals lie within angle phi (cos phi = cone[3]) of the axis, its points
// within rho of the centre c; for a point p of a triangle with outward normal n,
//   n.(p - eye) >= n.(c - eye) - rho >= |c| cos(psi + phi) - rho,   psi = angle(axis, c - eye),
// so the cluster is back-facing when |c| (cos psi cos phi - sin psi sin phi) - rho > 0; a
// margin of 1e-3 (|c| + rho) + 1e-4 absorbs the float rounding.  One cluster per lane.
// Returns +1: every triangle faces away (cull the cluster); -1: every triangle faces the camera
// (nothing of it will be culled: no need to pre-test its triangles); 0: mixed or unknown.  The -1
// verdict only chooses a code path -- the exact projected-area test still sees every triangle.
__device__ inline int cluster_facing(const double* __restrict__ Rt, const floatx4 sph, const floatx4 cone4)
{
    const float cone[4] = {cone4.x, cone4.y, cone4.z, cone4.w};
    const float m = cone[3];
    if (!(m > 0.0f)) return 0;
    const float r0 = (float)Rt[0], r1 = (float)Rt[1], r2 = (float)Rt[2], r3 = (float)Rt[3], r4 = (float)Rt[4],
                r5 = (float)Rt[5], r6 = (float)Rt[6], r7 = (float)Rt[7], r8 = (float)Rt[8];
    const float X = r0 * sph.x + r1 * sph.y + r2 * sph.z + (float)Rt[9];
    const float Y = r3 * sph.x + r4 * sph.y + r5 * sph.z + (float)Rt[10];
    const float Z = r6 * sph.x + r7 * sph.y + r8 * sph.z + (float)Rt[11];
    const float ax = r0 * cone[0] + r1 * cone[1] + r2 * cone[2];
    const float ay = r3 * cone[0] + r4 * cone[1] + r5 * cone[2];
    const float az = r6 * cone[0] + r7 * cone[1] + r8 * cone[2];
    const float rho = sph.w * 1.001f + 1e-3f;
    const float D = cone_sqrt(X * X + Y * Y + Z * Z);
    const float ad = ax * X + ay * Y + az * Z;           // |c| cos psi
    const float sp = cone_sqrt(fmaxf(D * D - ad * ad, 0.0f)); // |c| sin psi
    const float spread = sp * cone_sqrt(fmaxf(1.0f - m * m, 0.0f)) + rho;
    const float margin = 1e-3f * (D + rho) + 1e-4f;
    if (ad * m - spread > margin) return 1;
    if (-ad * m - spread > margin) return -1;    // the same bound with the axis reversed
    return 0;
}
__device__ inline bool cluster_faces_away(const double* __restrict__ Rt, const floatx4 sph, const floatx4 cone4)
{
    return cluster_facing(Rt, sph, cone4) > 0;
}

// Back-face culling.  A body whose mesh is a closed, consistently oriented surface, every shell
// of it wound the same way (checked at create time; the surface is taken to be embedded, i.e.
// not passing through itself inside-out) and whose bounding sphere lies wholly in front of the camera plane (so no
// triangle is dropped by the Z <= 0 rule and the camera is outside it) cannot show a back face:
// the ray through a sample enters the solid through a front face no farther than any back face
// it meets, so the z-min -- the only thing the tile keeps -- is decided by front faces alone.
// Dropping back faces (whole clusters by their normal cone, single triangles by the sign of
// their projected area) therefore leaves every depth unchanged; in binary64 the two can differ
// only for a sample within rounding distance of a silhouette EDGE itself.  Bodies that fail the
// create-time check keep every triangle.
__device__ inline int body_cullsign(const DevParams& P, const double* __restrict__ Rt, int b)
{
    if (P.body_cull[b] == 0) return 0;
    const double sx = P.sphere[b][0], sy = P.sphere[b][1], sz = P.sphere[b][2];
    const double Z = ((Rt[6] * sx + Rt[7] * sy) + Rt[8] * sz) + Rt[11];
    return (Z - P.sphere[b][3] > 1e-6) ? P.body_cull[b] : 0;
}

// A set-up triangle's sample points (or the cooperative queue when it is big).
__device__ inline void raster_lane_samples(const Tri& T, int t, int wx0, int wy0, unsigned* tile, int tw, int* big, int* nbig)
{
#ifdef RBS_EXP_SKIP_PIXELS   // profiling builds (tools/phase_timing.py): triangle setup only
    if (T.nv0 != 12345.678) return;
#endif
    const int bw = T.xhi - T.xlo + 1, bh = T.yhi - T.ylo + 1;
    if (bw * bh > kBigThresh) {
        const int slot = atomicAdd(nbig, 1);
        if (slot < kBigCap) { big[slot] = t; return; }
    }
    // (one flat loop over the box's samples instead of rows x columns: fewer trips, 3 % slower)
#if RBS_EDGE_FILTER
    const float thr = T.thr, nthr = -T.thr;
    float fj = 0.0f;
    for (int row = T.ylo; row <= T.yhi; ++row, fj += 1.0f) {
        const float r0 = fmaf(T.fA[0], fj, T.fC[0]), r1 = fmaf(T.fA[1], fj, T.fC[1]), r2 = fmaf(T.fA[2], fj, T.fC[2]);
        unsigned* trow = tile + (row - wy0) * tw - wx0;
        float fi = 0.0f;
        for (int col = T.xlo; col <= T.xhi; ++col, fi += 1.0f) {
            const float e0 = fmaf(T.fB[0], fi, r0), e1 = fmaf(T.fB[1], fi, r1), e2 = fmaf(T.fB[2], fi, r2);
            const float mn = __builtin_fminf(__builtin_fminf(e0, e1), e2), mx = __builtin_fmaxf(__builtin_fmaxf(e0, e1), e2);
            bool in = mn > thr || mx < nthr;                          // clearly inside (either winding)
            const bool out = mn < nthr && mx > thr;                   // clearly outside
            if (__builtin_expect(__ballot(!in && !out) != 0, 0)) {    // within thr of an edge (or not finite): the oracle's expression
                if (!in && !out) {
                    // (opaque to the optimiser: the row-invariant half of the binary64 expression must not be
                    // hoisted into the row loop -- it is needed about once in 1e4 samples)
                    double px = (double)col, py = (double)row;
                    asm volatile("" : "+v"(px), "+v"(py));
                    in = tri_inside_exact(T, px, py);
                }
            }
            if (in) tri_depth(T, (double)col, (double)row, trow + col);
        }
    }
#else
#if RBS_PIN_EDGES
    // (the six edge components opaque to the optimiser, so that none of them is re-derived from the vertices inside the
    // loop: left alone the compiler re-materialises one of them per trip -- a binary64 subtraction it deems cheaper than two
    // registers; C2 / C4 +0.5-0.7 %, C1 +0.2 %, same bits)
    Tri U = T;
    asm volatile("" : "+v"(U.e01u), "+v"(U.e01v), "+v"(U.e12u), "+v"(U.e12v), "+v"(U.e20u), "+v"(U.e20v));
    for (int row = U.ylo; row <= U.yhi; ++row)
        for (int col = U.xlo; col <= U.xhi; ++col)
            tri_pixel(U, col, row, tile, tw, wx0, wy0);
#else
    for (int row = T.ylo; row <= T.yhi; ++row)
        for (int col = T.xlo; col <= T.xhi; ++col)
            tri_pixel(T, col, row, tile, tw, wx0, wy0);
#endif
#endif
}

// One lane's triangle: setup, then its sample points.
__device__ inline void raster_lane_triangle(const DevParams& P, int t, const double* __restrict__ Rt, int wx0,
                                            int wy0, int wx1, int wy1, int cullsign, unsigned* tile, int tw,
                                            int* big, int* nbig RBS_TICK_PARAM)
{
    Tri T;
    RBS_TICK(10);
    const bool ok_ = tri_setup(P, t, Rt, wx0, wy0, wx1, wy1, cullsign, T);
    RBS_TICK(11);
    if (!ok_) return;
    raster_lane_samples(T, t, wx0, wy0, tile, tw, big, nbig);
}

// A whole cluster whose triangles all go to the setup (a body that is not culled, or a cluster
// whose normal cone faces the camera), with its VERTICES shared: a triangle's transform and
// projection -- 99 of a setup's 175 instructions, three binary64 divisions among them -- are the
// same for every triangle around a vertex, six on a closed mesh.  Lane v transforms and projects
// the cluster's unique vertex v once (the oracle's operations in the oracle's order: the values
// are those tri_setup computes), lane t fetches its triangle's three vertices from the lanes that
// hold them (ds_bpermute: 30 cross-lane reads, no LDS space, no VALU) and finishes the setup.
// nv = the cluster's unique vertices (<= 64).
__device__ inline void raster_shared_cluster(const DevParams& P, int c, int nv, int t_end, const double* __restrict__ Rt,
                                             int wx0, int wy0, int wx1, int wy1, int cullsign, unsigned* tile, int tw,
                                             int* big, int* nbig RBS_TICK_PARAM)
{
    const int lane = threadIdx.x & 63;
    RBS_TICK(10);
    const double* __restrict__ cv = P.cluster_vtx + (size_t)c * 192;
    const int vl = min(lane, nv - 1);                      // (lanes past the list repeat its last vertex: nobody fetches them)
    const double vx = cv[vl], vy = cv[64 + vl], vz = cv[128 + vl];
    const int t = (c << 6) + lane;
    const unsigned packed = P.tri_local[t];
    __builtin_amdgcn_sched_barrier(0);
    double Xv, Yv, Zv, uv, vv;
    vertex_project(P, Rt, vx, vy, vz, Xv, Yv, Zv, uv, vv);
    double X[3], Y[3], Z[3], u[3], v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int src = (int)((packed >> (8 * k)) & 63u);
        X[k] = __shfl(Xv, src, 64); Y[k] = __shfl(Yv, src, 64); Z[k] = __shfl(Zv, src, 64);
        u[k] = __shfl(uv, src, 64); v[k] = __shfl(vv, src, 64);
    }
    Tri T;
    const bool ok_ = t < t_end && packed != 0xffffffffu && tri_finish(P, X, Y, Z, u, v, wx0, wy0, wx1, wy1, cullsign, T);
    RBS_TICK(11);
    if (ok_) raster_lane_samples(T, t, wx0, wy0, tile, tw, big, nbig);
}
// (phase-timing builds: the time between a lane leaving its sample loop and the next tick is the
// wave's sample phase -- lane 0 may leave early, so the caller ticks after reconvergence)

// Rasterize every body of one particle into the LDS tile covering window
// [wx0,wx1) x [wy0,wy1).  One wave takes one 64-triangle cluster at a time (triangles were
// grouped into compact surface patches at create time): wave-uniform frustum / normal-cone cull
// of whole clusters, then one lane per triangle.
// Triangle compaction: the expensive part -- binary64 transform, three divisions, projection,
// plane setup, the per-sample loops -- runs in lockstep, so a lane whose triangle turns out to
// face away idles through all of it.  For bodies that may be culled (closed, oriented, in front
// of the camera) a float32 test of the camera centre against the triangle's model-space plane
// (four floats, three FMAs) drops the triangles that CLEARLY face away before the setup, and the
// survivors' indices are compacted through a per-wave LDS queue (tq) and set up 64 at a time with
// full lanes.  The pre-test only removes triangles the exact projected-area test inside
// tri_setup would remove as well (it keeps everything within 10 um + 1e-5 |eye| of edge-on), so
// depths are unchanged bit for bit.
// Small triangles are rasterized by their lane; triangles whose clipped bbox exceeds kBigThresh
// pixels are queued in LDS and rasterized by the whole block, pixel-parallel.
// Clears the tile first; on return the tile is complete and synchronised.
// ONE: the object model has a single body (rbs_raster_kernel_one_f64): no body loop, no body mask, every per-body table entry a
// fixed kernel argument.
template <bool MANY = false, bool ONE = false>
__device__ inline void raster_window(const DevParams& P, const double* __restrict__ pose,
                                     int wx0, int wy0, int wx1, int wy1, bool cull, unsigned* tile,
                                     int* big, int* nbig, int* tq, unsigned body_mask, unsigned long long* cullm = nullptr)
{
    const int tw = wx1 - wx0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // The first body's pose is the item's first dependent load (4 000 of an item's 150 000 cycles
    // when waited for): its two cache lines are touched before the tile is cleared, so that the
    // loads that follow the clear hit the scalar cache.  (Holding the twelve values themselves
    // across the clear costs the registers the setup needs: 32 spills.)
    int b0 = 0;
    if (!ONE) while (b0 < P.n_bodies && !((body_mask >> b0) & 1u)) ++b0;
    double touch0 = 0.0, touch1 = 0.0;
    if (ONE || b0 < P.n_bodies) { touch0 = pose[12 * b0]; touch1 = pose[12 * b0 + 11]; }
    const int npx = tw * (wy1 - wy0);
    for (int p = threadIdx.x; p < npx; p += kBlock) tile[p] = kInfBits;
    if (threadIdx.x == 0) *nbig = 0;
    __syncthreads();
    asm volatile("" ::"s"(touch0), "s"(touch1));   // (the touches are not dead code)
#ifdef RBS_EXP_NO_RASTER   // profiling builds: per-item overheads + tile clear + pixel scan only
    return;
#endif
    RBS_TICK_DECL;
    // surviving clusters so far: dealt round-robin to the block's waves (an LDS ticket per
    // cluster instead measured no better: the waves of a block finish within a few percent)
    int taken = 0;
#pragma unroll
    for (int b = 0; b < (ONE ? 1 : P.n_bodies); ++b) {
        if (!ONE && !((body_mask >> b) & 1u)) continue;    // a body of another group: its rectangle is elsewhere
        const double* Rt = pose + 12 * b;
        const int c0 = P.tri_begin[b] >> 6, c1 = P.tri_begin[b + 1] >> 6;
        const int t_end = P.tri_end[b];
        const int cullsign = body_cullsign(P, Rt, b);
        // camera centre in model coordinates, e = -R^T t (float32: only the conservative pre-test uses it)
        const float tx = (float)Rt[9], ty = (float)Rt[10], tz = (float)Rt[11];
        const float ex = -((float)Rt[0] * tx + (float)Rt[3] * ty + (float)Rt[6] * tz);
        const float ey = -((float)Rt[1] * tx + (float)Rt[4] * ty + (float)Rt[7] * tz);
        const float ez = -((float)Rt[2] * tx + (float)Rt[5] * ty + (float)Rt[8] * tz);
        const float eps = 1e-5f * (1.0f + sqrtf(ex * ex + ey * ey + ez * ez));
        const float fsign = (float)cullsign;
        int qh = 0, qn = 0;   // ring queue of surviving triangle indices: head, count (wave-uniform)
#ifdef RBS_PHASE_TIMING
        if (threadIdx.x == 0 && eps + fsign == -12345.f) P.out[0] = 0.0;   // (waits for the pose)
        RBS_TICK(0);    // the body's pose and what is derived from it
#endif
        // MANY: the cull of up to kCullSteps steps of 64 clusters is dealt to the block's waves, the verdicts meet in
        // LDS (two barriers per stretch -- the first keeps a wave that still reads the previous verdicts safe);
        // otherwise one stretch = the whole body, every wave culling every step for itself.
        for (int cbase = c0, cend = c1; cbase < c1; cbase = cend) {
        cend = MANY ? min(c1, cbase + 64 * kCullSteps) : c1;
        if (MANY) {
            __syncthreads();
            for (int it0 = wave; cbase + (it0 << 6) < cend; it0 += 4 * (kBlock / 64)) {   // four of the wave's steps at a time:
                floatx4 sph[4], cone[4];                                                      // one memory latency, not four
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ci = cbase + ((it0 + u * (kBlock / 64)) << 6) + lane;
                    sph[u] = floatx4{0.f, 0.f, 0.f, 0.f}; cone[u] = sph[u];
                    if (ci < cend) {
                        sph[u] = reinterpret_cast<const floatx4*>(P.cluster_sphere)[ci];
                        cone[u] = reinterpret_cast<const floatx4*>(P.cluster_cone)[ci];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int it = it0 + u * (kBlock / 64), ci = cbase + (it << 6) + lane;
                    if (cbase + (it << 6) >= cend) break;   // wave-uniform
                    const int facing = cullsign != 0 && ci < cend ? cluster_facing(Rt, sph[u], cone[u]) : 0;
                    const bool hit = ci < cend && (!cull || cluster_may_touch(P, Rt, sph[u], wx0, wy0, wx1, wy1)) && facing <= 0;
                    const unsigned long long hitm = __ballot(hit), toward = __ballot(facing < 0);
                    if (lane == 0) { cullm[2 * it] = hitm; cullm[2 * it + 1] = toward; }
                }
            }
            __syncthreads();
        }
        for (int base = cbase; base < cend; base += 64) {
            // 64 clusters culled at once, one per lane (every wave computes the same mask)
            const int ci = base + lane;
            int facing;
            bool hit;
#ifndef RBS_NV_EARLY
#define RBS_NV_EARLY 1   // the cluster_nv read issued with the sphere and cone reads, not behind the cull (C1 +0.4 %)
#endif
            int nvc_early = 0;
            if (MANY) {
                const int it = (base - cbase) >> 6;
                hit = (cullm[2 * it] >> lane) & 1ull;
                facing = (cullm[2 * it + 1] >> lane) & 1ull ? -1 : 0;
            } else {
                floatx4 sph = floatx4{0.f, 0.f, 0.f, 0.f}, cone = sph;
                if (ci < c1) {
                    sph = reinterpret_cast<const floatx4*>(P.cluster_sphere)[ci];
                    cone = reinterpret_cast<const floatx4*>(P.cluster_cone)[ci];
#if RBS_NV_EARLY
                    nvc_early = P.cluster_nv[ci];
#endif
                }
                facing = cullsign != 0 && ci < c1 ? cluster_facing(Rt, sph, cone) : 0;
                hit = ci < c1 && (!cull || cluster_may_touch(P, Rt, sph, wx0, wy0, wx1, wy1)) && facing <= 0;
            }
            const unsigned long long mask = __ballot(hit);
            RBS_TICK(8);
            RBS_COUNT(16, min(64, c1 - base)); RBS_COUNT(17, __popcll(mask));
            // this wave's share: the surviving clusters are dealt round-robin by their rank
            const int rank = taken + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            unsigned long long mine = __ballot(hit && (rank % (kBlock / 64)) == wave);
            taken += __popcll(mask);
            if (MANY && mine == 0) continue;   // (nine clusters in ten of a large body are culled: most steps leave this wave nothing)
#if RBS_SHARE_VERTICES
            // Clusters all of whose triangles go to the setup -- every cluster of a body that is not
            // culled, and the clusters whose normal cone faces the camera (4/5 of a closed body's
            // surviving triangles) -- are set up with their vertices shared, straight from the
            // cluster (no pre-test, no ring); the others (silhouette clusters, clusters with more
            // than 64 unique vertices) take the per-triangle route below.
            const int nvc = RBS_NV_EARLY && !MANY ? nvc_early : ci < c1 ? P.cluster_nv[ci] : 0;
            unsigned long long whole = mine & __ballot(hit && nvc > 0 && (cullsign == 0 || facing < 0));
            mine &= ~whole;
            while (whole) {
                const int bit = __builtin_ctzll(whole);
                whole &= whole - 1;
                const int nv = __shfl(nvc, bit, 64);
                raster_shared_cluster(P, base + bit, __builtin_amdgcn_readfirstlane(nv), t_end, Rt, wx0, wy0, wx1, wy1, cullsign,
                                      tile, tw, big, nbig RBS_TICK_ARG);
                __builtin_amdgcn_wave_barrier(); RBS_TICK(12); RBS_COUNT(18, 1); RBS_COUNT(19, 1);
            }
#endif
            if (cullsign == 0) {   // nothing to pre-test: the clusters' lanes go straight to the setup
                while (mine) {
                    const int bit = __builtin_ctzll(mine);
                    mine &= mine - 1;
                    raster_lane_triangle(P, ((base + bit) << 6) + lane, Rt, wx0, wy0, wx1, wy1, 0, tile, tw, big, nbig RBS_TICK_ARG);
                    __builtin_amdgcn_wave_barrier(); RBS_TICK(12);
                }
                continue;
            }
            while (mine) {
                // pre-test kPre clusters per step: their planes are loaded together (one exposed
                // latency per step, not per cluster; three waves per SIMD do not hide it)
                int tt[kPre];
                floatx4 pl[kPre];
#pragma unroll
                for (int u = 0; u < kPre; ++u) {
                    tt[u] = -1;
                    if (mine) {   // wave-uniform
                        const int bit = __builtin_ctzll(mine);
                        mine &= mine - 1;
                        tt[u] = ((base + bit) << 6) + lane;
                    }
                    pl[u] = floatx4{0.f, 0.f, 0.f, 0.f};
                    if (tt[u] >= 0 && tt[u] < t_end) pl[u] = P.tri_plane[tt[u]];
                }
#pragma unroll
                for (int u = 0; u < kPre; ++u) {
                    if (tt[u] < 0) break;   // wave-uniform
                    const float sd = pl[u].x * ex + pl[u].y * ey + pl[u].z * ez + pl[u].w;   // eye's signed distance, winding side
                    const bool keep = tt[u] < t_end && !(fsign * sd < -eps);   // outward normal = cullsign * winding normal; NaN keeps
                    const unsigned long long km = __ballot(keep);
                    if (keep) {
                        const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0));
                        tq[(qh + pos) & (kTq - 1)] = tt[u];
                    }
                    qn += __popcll(km);
                }
                while (qn >= 64) {
                    __builtin_amdgcn_wave_barrier();
                    const int t64 = tq[(qh + lane) & (kTq - 1)];
                    __builtin_amdgcn_wave_barrier();
                    qh = (qh + 64) & (kTq - 1);
                    qn -= 64;
                    raster_lane_triangle(P, t64, Rt, wx0, wy0, wx1, wy1, cullsign, tile, tw, big, nbig RBS_TICK_ARG);
                    __builtin_amdgcn_wave_barrier(); RBS_TICK(12); RBS_COUNT(18, 1);
                }
            }
        }
        }
        if (qn > 0) {   // the body's last, partly filled batch
            __builtin_amdgcn_wave_barrier();
            if (lane < qn) {
                const int tt = tq[(qh + lane) & (kTq - 1)];
                raster_lane_triangle(P, tt, Rt, wx0, wy0, wx1, wy1, cullsign, tile, tw, big, nbig RBS_TICK_ARG);
            }
            __builtin_amdgcn_wave_barrier();
            RBS_TICK(12); RBS_COUNT(18, 1); RBS_COUNT(23, qn);
        }
    }
    RBS_TICK(9);   // pre-test, compaction, loop bookkeeping (what 8, 10, 11, 12 do not claim)
    __syncthreads();
    RBS_TICK(13);  // waiting for the block's other waves
    const int nb = min(*nbig, kBigCap);
    RBS_COUNT(20, nb); RBS_COUNT(21, 1);
    if (nb == 0) return;   // block-uniform; the usual case (a barrier costs an item about 1 %)
    for (int e = 0; e < nb; ++e) {
        const int t = big[e];
        const double* Rt = pose + 12 * (ONE ? 0 : body_of(P, t));
        Tri T;
        if (!tri_setup(P, t, Rt, wx0, wy0, wx1, wy1, 0, T)) continue;  // uniform across the block; culled before queueing
        const int bw = T.xhi - T.xlo + 1, bh = T.yhi - T.ylo + 1;
        for (int k = threadIdx.x; k < bw * bh; k += kBlock) {
            const int r = k / bw;
            tri_pixel(T, T.xlo + (k - r * bw), T.ylo + r, tile, tw, wx0, wy0);
        }
    }
    __syncthreads();
    RBS_TICK(14);  // big triangles
}

// ------------------------------------------------------------------ pixel model
// KinectPixelModel / OcclusionModel restated; expression text kept identical to the
// specification in oracle/rbsensor_oracle.c so both sides perform the same binary64
// operations (transcendentals differ by <= a few ulp of double between libm and ocml).
// Terms of the pixel model that depend on the observed depth only are computed once per
// frame (frame_aux_kernel) instead of once per particle-pixel.  With sigma = ms + sf*o^2,
// lam = ln2/half_life, d = r - o, w = d/(sqrt2 sigma):
//   p_vis(o|r) = tw/D + c_v * exp(-w^2)                      c_v = (1-tw)/(sqrt(2pi) sigma)
//   p_occ(o|r) = tw/D + e_o * E1/(E1-1) * (1 + erf(w + k))   E1 = exp(lam r), k = lam sigma/sqrt2,
//                                                            e_o = (1-tw) lam/2 * exp(lam/2 (lam sigma^2 - 2o))
//   p_bg(o)    = tw/D + (1-tw) lam exp(lam/2 (lam sigma^2 - 2o))        (rounded to float)
// Algebraically identical to SURVEY A.3 / oracle orc_prob_*; binary64 results differ from the
// oracle's expression order by a few ulp, far below the float rounding of a, b that follows.
enum { AUX_INV_S2S = 0, AUX_K = 1, AUX_OBS = 2, AUX_EO = 3, AUX_PLANES = 4 };

// keep != nullptr: `frame` is the caller's buffer and is also copied into the handle's own.
__device__ inline void frame_aux_pixel(int i, const float* __restrict__ frame, double* __restrict__ aux,
                                       float* __restrict__ pbg, int npx, double tw, double ms, double sf,
                                       double lam, float* __restrict__ keep)
{
    const float of = frame[i];
    if (keep) keep[i] = of;
    if (!aux) return;   // likelihood precision F32 derives these terms from the observation on the fly
    // the four terms of a pixel side by side (32 B): one cache line per evaluated pixel, not four
    double t4[4];
    rbsm::frame_terms((double)of, tw, ms, sf, lam, t4);
    double* a4 = aux + (size_t)AUX_PLANES * i;
    a4[0] = t4[0]; a4[1] = t4[1]; a4[2] = t4[2]; a4[3] = t4[3];
    (void)pbg;   // (p_bg = tw/D + 2 e_o is formed from the entry where it is needed: no plane of its own)
}

__global__ void frame_aux_kernel(const float* __restrict__ frame, double* __restrict__ aux,
                                 float* __restrict__ pbg, int npx, double tw, double ms, double sf,
                                 double lam, float* __restrict__ keep)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npx) frame_aux_pixel(i, frame, aux, pbg, npx, tw, ms, sf, lam, keep);
}

// log-likelihood ratio of one covered, observed pixel + posterior occlusion.
// Rounding points as in the oracle: a, b, p_bg -> float; a+b and the ratios in float (correctly
// rounded divisions); log in double.  The transcendentals are rbs_math.h's fixed-length binary64
// sequences (exp to 4e-16 relative, erfc to 1e-16 absolute, log to 1e-16 absolute + its own
// rounding) instead of ocml's general-purpose exp / erf / log: about 125 VALU instructions per 64
// pixels instead of 250, no divergent ranges, nothing spilled.  The float roundings of a and b then
// agree with the oracle's (libm) except where a term lies within ~1e-15 relative of a rounding
// boundary -- about one pixel in 1e7 -- which tests/ bound (planes: <= 1e-4 of pixels at 1 ulp;
// log-likelihoods: 1e-9).
struct MathTabs { const double* erfc; const double* logt; const double* ages; };
// Stamped planes: the first kLdsAges entries of the propagation table sit in LDS (a pixel the object covered within the last
// two seconds -- nearly every evaluated pixel of a tracked object); an older one takes the table in memory.
#ifndef RBS_EXACT_LDS_AGES
#define RBS_EXACT_LDS_AGES 64
#endif
constexpr int kLdsAges = RBS_EXACT_LDS_AGES;

#ifdef RBS_NOINLINE_EVAL
#define RBS_EVAL_INLINE __attribute__((noinline))
#else
#define RBS_EVAL_INLINE inline
#endif
// EXACT (stamped planes): `prior` is the pixel's stored VALUE and `age` the frames since its update -- the table entry of that
// age is requested WITH the frame terms, and the prior is formed once both have arrived (exact_prior's operations).
template <bool EXACT = false>
__device__ RBS_EVAL_INLINE double pixel_loglik(const DevParams& P, const MathTabs& M, int gi, float r, float prior, float& posterior, int age = 0)
{
    // the per-frame terms of this pixel, observation included: one 32-byte entry, one memory round trip
    typedef double doublex2 __attribute__((ext_vector_type(2)));
    // (with every lane reading one fixed entry instead -- no gather latency at all -- the kernel is 2.5 %
    // faster: prefetching the next batch's entries across the scan loop is not worth its registers)
    const doublex2* a4 = reinterpret_cast<const doublex2*>(P.aux + (size_t)AUX_PLANES * gi);
    const doublex2 a01 = a4[0], a23 = a4[1];
    doublex2 pt = {0.0, 0.0};
    if (EXACT) {
        const int ai = age > P.age_max ? 0 : age;   // (a background pixel's prior is bg_new whatever is computed here)
        // (the LDS pointer in its own address space: left generic, the two loads are merged into ONE flat load through a selected
        // pointer -- which waits on both memory counters and serialises with the frame-term gather)
        typedef const doublex2 __attribute__((address_space(3)))* lds_table;
        if (kLdsAges > 0 && __builtin_expect(__ballot(ai >= kLdsAges) == 0, 1)) pt = ((lds_table)M.ages)[ai];
        else pt = reinterpret_cast<const doublex2*>(P.ptab)[ai];
    }
    // what depends on the rendered depth alone runs while that entry travels
    const rbsm::PixelConsts C = {P.lambda, P.tw / kMaxDepth, P.cv0};
    const double rd = (double)r;
    const double g = rbsm::depth_term(C, rd);
    __builtin_amdgcn_sched_barrier(0);
#ifdef RBS_EXP_SKIP_EVAL     // profiling builds: the loads, none of the transcendental work
    posterior = prior;
    return a01.x + a01.y + a23.x + a23.y + g;
#endif
    if (EXACT) {
        const double new_visible = pt.x * (1.0 - (double)prior) + pt.y;
        const float pr = (float)(1.0 - new_visible);
        prior = age > P.age_max ? P.bg_new : pr;
    }
    return rbsm::pixel_loglik_f64(C, g, a01.x, a01.y, a23.x, a23.y, rd, prior, M.erfc, M.logt, posterior);
}

// Likelihood precision F32 (rbs_config.likelihood_precision = RBS_PRECISION_F32): the same model
// with the per-pixel transcendental work in float32 on the hardware's exp2 / log2 / rcp units --
// coverage, depth and the occlusion process are untouched (binary64 geometry, float state), only
// exp / erf / log and the mixture algebra change.  Written for RELATIVE accuracy: 1 + erf(x) as
// erfc(-x) (no cancellation in the lower tail) through t exp(-z^2 + poly(t)), t = 1/(1 + z/2)
// (Numerical Recipes' erfcc, fractional error 1.2e-7 everywhere); E1/(E1-1) as
// 1/(1 - exp(-lam r)); log(sum / p_bg) as log(sum) - log(p_bg), log(p_bg) recomputed per evaluated
// pixel in float32 like every other term that depends on the observation only.  About 50 VALU instructions per 64 pixels instead of 125 (ocml's
// expf / erfcf / logf and correctly rounded divisions) or 250 (binary64).  Per-pixel error of
// the log term: 1.3e-7 mean, 1.7e-6 max, bias -3e-8 (numpy float32 emulation over 4e5 random
// pixels); the particle's sum is accumulated in binary64.  tests/: <= 1e-5 relative against the
// reference-semantics (LAZY) oracle, the north_star tolerance.
#ifndef RBS_F32_EXP
#define RBS_F32_EXP 0
#endif
#ifndef RBS_F32_LOG
#define RBS_F32_LOG 0
#endif
#ifndef RBS_F32_DIV
#define RBS_F32_DIV 0
#endif
__device__ inline float fast_exp(float x) { return RBS_F32_EXP ? expf(x) : __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ inline float fast_log(float x) { return RBS_F32_LOG ? logf(x) : 0.6931471805599453f * __builtin_amdgcn_logf(x); }
// hardware reciprocal (1 ulp) + one Newton step: half an ulp for operands in the normal range
__device__ inline float fast_rcp(float x)
{
    if (RBS_F32_DIV) return 1.0f / x;
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(r, fmaf(-x, r, 1.0f), r);
}
__device__ inline float fast_erfc(float x)
{
    const float z = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.0f));
    float p = 0.17087277f;
    p = fmaf(p, t, -0.82215223f); p = fmaf(p, t, 1.48851587f); p = fmaf(p, t, -1.13520398f);
    p = fmaf(p, t, 0.27886807f);  p = fmaf(p, t, -0.18628806f); p = fmaf(p, t, 0.09678418f);
    p = fmaf(p, t, 0.37409196f);  p = fmaf(p, t, 1.00002368f);  p = fmaf(p, t, -1.26551223f);
    const float a = t * fast_exp(p - z * z);
    return x >= 0.0f ? a : 2.0f - a;
}
// Everything the likelihood needs beyond (rendered depth, prior, observation) is a function of
// the observation: recomputing those terms costs about 15 float instructions per pixel, fetching
// them (per-frame planes, as the binary64 path does) costs two gathers whose latency three waves
// per SIMD do not hide.  No memory access at all here.
__device__ inline double pixel_loglik_f32(const DevParams& P, float r, float prior, float o, float& posterior)
{
#ifdef RBS_EXP_SKIP_EVAL     // profiling builds: the queue traffic, none of the likelihood arithmetic
    posterior = prior;
    return (double)(r + o);
#endif
    const float lam = (float)P.lambda, twD = (float)(P.tw / kMaxDepth), omt = (float)(1.0 - P.tw);
    const float sigma = fmaf((float)P.sf * o, o, (float)P.ms);
    const float is = fast_rcp(sigma);
    const float inv_s2s = 0.7071067811865476f * is;                       // 1/(sqrt2 sigma)
    const float kk = (0.7071067811865476f * lam) * sigma;                 // lam sigma/sqrt2
    const float cv = (0.3989422804014327f * omt) * is;                    // (1-tw)/(sqrt(2 pi) sigma)
    const float eo = (0.5f * omt * lam) * fast_exp((0.5f * lam) * fmaf(lam * sigma, sigma, -2.0f * o));
    const float lpbg = fast_log(fmaf(2.0f, eo, twD));
    const float w = (r - o) * inv_s2s;
    const float pv = fmaf(cv, fast_exp(-(w * w)), twD);
    const float ratio = fast_rcp(1.0f - fast_exp(-(r * lam)));
    const float po = fmaf(eo * ratio, fast_erfc(-(w + kk)), twD);
    const float av = pv * (1.0f - prior);
    const float bv = po * prior;
    const float sum = av + bv;
    // the posterior is STATE (its error is carried into the next frames): one Newton step on the
    // reciprocal and a residual correction of the quotient -- within half an ulp of bv / sum for
    // these operands (normal range), 6 instructions instead of the IEEE division's 12
    const float rs = fast_rcp(sum);
    const float qd = bv * rs;
    posterior = fmaf(rs, fmaf(-qd, sum, bv), qd);
    return (double)(fast_log(sum) - lpbg);
}

__device__ inline double block_reduce_sum(double v, double* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < kBlock / 64; ++w) s += red[w];
    return s;
}

// ------------------------------------------------------------------ raster work item
struct Smem {
    unsigned* tile; int* big; double* red; int* nbig; int* item; int* evalq;
    double* mtab;   // precision F64: the erfc and log tables of rbs_math.h (kMathTabDoubles doubles)
    unsigned long long* cull;   // MANY: [kCullSteps][2] verdicts of the shared cluster cull (raster_window)
    int* ageq;                  // stamped planes: per wave kEvalQueue ages of the queued pixels (a fifth plane of the ring, behind everything else)
    double* agetab;             // ... and the first kLdsAges entries of the propagation table
};
__device__ inline Smem carve(unsigned char* smem, int kTilePx, bool math_tables)
{
    Smem m;
    m.tile = reinterpret_cast<unsigned*>(smem);
    m.big = reinterpret_cast<int*>(smem + sizeof(unsigned) * kTilePx);
    m.red = reinterpret_cast<double*>(smem + sizeof(unsigned) * kTilePx + sizeof(int) * kBigCap);
    m.nbig = reinterpret_cast<int*>(m.red + kBlock / 64);
    m.item = m.nbig + 1;
    m.evalq = m.nbig + 4;   // per wave: kQPlanes planes of kEvalQueue ints
    m.mtab = reinterpret_cast<double*>(m.evalq + kQPlanes * (kBlock / 64) * kEvalQueue);   // (16-byte aligned: everything before it is)
    // (last, so that nothing else moves: shifting the rings and the tables by these 448 bytes cost C1 0.6 %)
    m.cull = reinterpret_cast<unsigned long long*>(m.mtab + (math_tables ? kMathTabDoubles : 0));
    m.ageq = nullptr;
    m.agetab = nullptr;
    return m;
}

// One (particle, tile) work item: rasterize the tile window, evaluate its pixels, return the
// block-reduced partial log-likelihood (valid in thread 0).
template <bool UPDATE, int PREC, bool SLAB, bool MANY = false, int PHASE = 0, bool STP = false, bool EXACT = false, bool ONE = false>
__device__ inline double raster_eval_tile(const DevParams& P, int particle, Rect r, int tile_id,
                                          const Smem& m, unsigned body_mask, bool draw, int& ticket, const unsigned* gtile = nullptr)
{
    const TileGrid tg = tile_grid(r.x1 - r.x0, r.y1 - r.y0, P.tile_w, min(P.tile_w * P.tile_h, P.tile_px));
    const int ty = tile_id / tg.nx, tx = tile_id - ty * tg.nx;
    const int wx0 = r.x0 + tx * tg.tw, wy0 = r.y0 + ty * tg.th;
    const int wx1 = min(r.x1, wx0 + tg.tw), wy1 = min(r.y1, wy0 + tg.th);
    const int tw = wx1 - wx0, npx = tw * (wy1 - wy0);
    const bool whole = (wx0 == r.x0 && wy0 == r.y0 && wx1 == r.x1 && wy1 == r.y1);

    const double* pose = P.poses + (size_t)particle * 12 * P.n_bodies;
    const int parent = P.parents[particle];
    // a parent slot outside the allocation (only possible through the unchecked device-pointer
    // API) must not turn into a wild read: the particle's likelihood becomes NaN instead
    if ((unsigned)parent >= (unsigned)P.slots) return NAN;
    const float* __restrict__ src = parent_plane(P, parent);
    float* __restrict__ dst = UPDATE ? P.occ_dst + (size_t)particle * P.plane_stride : nullptr;
    const int4 pw = parent_window(P, parent);   // outside it the parent's plane is implicitly bg_old
    // stamped planes: a slot's ages follow its values
    const unsigned short* __restrict__ asrc = EXACT ? reinterpret_cast<const unsigned short*>(src + P.plane_px) : nullptr;
    unsigned short* __restrict__ adst = EXACT && UPDATE ? reinterpret_cast<unsigned short*>(dst + P.plane_px) : nullptr;
    const unsigned short* __restrict__ abgp = EXACT && STP ? reinterpret_cast<const unsigned short*>(P.bgp_src + P.npx) : nullptr;

    RBS_TICK_DECL;
    // a rectangle that is a single tile was sized from the same spheres: nothing to cull
    if (PHASE == 0)
    raster_window<MANY, ONE>(P, pose, wx0, wy0, wx1, wy1, !whole || (!ONE && (body_mask & (body_mask - 1u)) != 0u), m.tile, m.big, m.nbig,
                  m.evalq + (threadIdx.x >> 6) * kQPlanes * kEvalQueue, body_mask, m.cull);   // the eval queue is idle during the raster phase
    RBS_TICK(2);

    // (SLAB is a template parameter: whole planes pay neither the registers nor the index arithmetic;
    // fetched here, after the raster phase, so that they are not live through it)
    const PlaneRef sref = SLAB ? parent_ref(P, parent) : PlaneRef{0, 0, P.cols};
    const PlaneRef dref = SLAB && UPDATE ? child_ref(P, particle) : sref;
    // Pixel pass.  Only ~1/3 of a tile's pixels are covered by the object, in runs that leave
    // most lanes of a wave idle in the expensive likelihood code, so each wave compacts its
    // covered+observed pixels into an LDS queue and evaluates them 64 at a time with full lanes.
    // Everything else (the occlusion process on uncovered pixels) is finished in the scan.
    double ll = 0.0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int* q = m.evalq + wave * kQPlanes * kEvalQueue;   // planes: pixel index, depth bits, prior, observation
    int* aq = EXACT ? m.ageq + wave * kEvalQueue : nullptr;   // stamped planes: plane 2 holds the stored VALUE, this ring its age
    const MathTabs mt = RBS_MATH_LDS ? MathTabs{m.mtab, m.mtab + rbsm::kErfcIntervals * rbsm::kErfcCoefs, m.agetab}
                                     : MathTabs{rbsm::kErfcTab, rbsm::kLogTab, m.agetab};
    int qh = 0, qn = 0;                                // ring: head, count (wave-uniform)
    // evaluate the 64 (or, at the end, `cnt_`) oldest queued pixels
#define RBS_EVAL_BATCH(cnt_)                                                                            \
    do {                                                                                                \
        if (lane < (cnt_)) {                                                                            \
            const int at_ = (qh + lane) & (kEvalQueue - 1);                                             \
            const int eg_ = q[at_];                                                                     \
            const float ed_ = __int_as_float(q[kEvalQueue + at_]);                                      \
            float ep_ = __int_as_float(q[2 * kEvalQueue + at_]);                                        \
            float post_;                                                                                \
            if (PREC) ll += pixel_loglik_f32(P, ed_, ep_, __int_as_float(q[3 * kEvalQueue + at_]), post_); \
            else ll += pixel_loglik<EXACT>(P, mt, eg_, ed_, ep_, post_, EXACT ? aq[at_] : 0);           \
            /* plane 0 holds the child-plane offset in precision F32 (which needs no frame index), */   \
            /* plane 3 in F64 */                                                                        \
            if (UPDATE) dst[(PREC || !SLAB) ? eg_ : q[3 * kEvalQueue + at_]] = post_;                   \
        }                                                                                               \
    } while (0)
    // push this lane's pixel if `active`; evaluate 64 queued pixels as soon as there are 64
#define RBS_PUSH_EVAL(active, gidx, didx, depthbits, prior, obs, age)                                   \
    do {                                                                                                \
        const unsigned long long mask_ = __ballot(active);                                              \
        if (mask_) {                                                                                    \
            if (active) {                                                                               \
                const int pos_ = (qh + qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mask_ >> 32),          \
                                                __builtin_amdgcn_mbcnt_lo((unsigned)mask_, 0))) & (kEvalQueue - 1); \
                q[pos_] = PREC ? (didx) : (gidx);                                                       \
                q[kEvalQueue + pos_] = (int)(depthbits);                                                \
                q[2 * kEvalQueue + pos_] = __float_as_int(prior);                                       \
                if (EXACT) aq[pos_] = (int)(age);                                                       \
                if (PREC) q[3 * kEvalQueue + pos_] = __float_as_int(obs);                               \
                else if (SLAB) q[3 * kEvalQueue + pos_] = (didx);                                       \
            }                                                                                           \
            qn += __popcll(mask_);                                                                      \
            if (qn >= 64) {                                                                             \
                __builtin_amdgcn_wave_barrier();                                                        \
                RBS_EVAL_BATCH(64);                                                                     \
                __builtin_amdgcn_wave_barrier();                                                        \
                qh = (qh + 64) & (kEvalQueue - 1);                                                      \
                qn -= 64;                                                                               \
            }                                                                                           \
        }                                                                                               \
    } while (0)

#ifdef RBS_EXP_NO_SCAN     // profiling builds: no pixel pass at all
    if (P.cols < 0)
#endif
    if ((P.cols & 3) == 0) {
        // Four pixels per lane: rows of the tile, of both planes and of the frame are 16-byte
        // aligned (rectangles and windows move in float4 columns), so one ds_read_b128 + two
        // dwordx4 loads + one dwordx4 store serve four pixels, and the row/column walk is
        // incremental.  An active pixel's prior is stored with its quad and overwritten by its
        // posterior when the same wave evaluates it later (same wave, same address: in order).
        const int tq = tw >> 2, nq = npx >> 2;
        const int qstep = kBlock / tq, rstep = kBlock - qstep * tq;
        int lr = (int)threadIdx.x / tq;
        int qc = (int)threadIdx.x - lr * tq;
        const uint4* __restrict__ tile4 = reinterpret_cast<const uint4*>(PHASE == 2 ? gtile : m.tile);
        // kScanUnroll quads per lane per trip: all their loads (the parent's values come from HBM --
        // another call wrote them -- and a dependent load per trip left the phase latency bound:
        // 13 % of the kernel) are issued before the first is used
        // (binary64 likelihood: two quads per trip as well since it fits the budget without spilling in the loops:
        // raster kernel 0.179 -> 0.175 ms)
#ifndef RBS_SCAN_UNROLL_EVAL
#define RBS_SCAN_UNROLL_EVAL 4   // (the likelihood kernel of the split launch: its tile reads are memory round trips, and it has the registers --
                                 //  four quads' loads in flight per lane: 0.1945 -> 0.1873 ms for both kernels on C1; six: 0.1880; eight waves per
                                 //  SIMD at one quad, 64 registers, every item resident at once: 0.1933.  The order in which a wave queues its
                                 //  pixels does not depend on it: same bits)
#endif
#ifndef RBS_SCAN_UNROLL_ONE
#define RBS_SCAN_UNROLL_ONE 3    // (the single-body kernel has the registers for a third quad in flight: see rbs_raster_kernel_one_f64)
#endif
#ifndef RBS_SCAN_UNROLL_EXACT
#define RBS_SCAN_UNROLL_EXACT RBS_SCAN_UNROLL_F64
#endif
#ifndef RBS_SCAN_UNROLL_EXACT_ONE
#define RBS_SCAN_UNROLL_EXACT_ONE RBS_SCAN_UNROLL_EXACT
#endif
        constexpr int kScanUnroll = PHASE == 2 ? RBS_SCAN_UNROLL_EVAL : PREC ? RBS_SCAN_UNROLL : EXACT ? (ONE ? RBS_SCAN_UNROLL_EXACT_ONE : RBS_SCAN_UNROLL_EXACT)
                                    : ONE ? RBS_SCAN_UNROLL_ONE : RBS_SCAN_UNROLL_F64;
        for (int q0 = wave * 64; q0 < nq; q0 += kBlock * kScanUnroll) {
            uint4 d4[kScanUnroll];
            floatx4 s4[kScanUnroll], o4[kScanUnroll];
            uint2 a4[EXACT ? kScanUnroll : 1];                       // stamped planes: the quad's four ages
            int gb[kScanUnroll], sb[kScanUnroll], db[kScanUnroll];   // offsets into the frame, the parent's plane, the child's
            bool vl[kScanUnroll], ac[kScanUnroll];
#pragma unroll
            for (int u = 0; u < kScanUnroll; ++u) {
                const int qd = q0 + u * kBlock + lane;
                vl[u] = qd < nq;
                d4[u] = make_uint4(kInfBits, kInfBits, kInfBits, kInfBits);
                s4[u] = floatx4{P.bg_old, P.bg_old, P.bg_old, P.bg_old};
                o4[u] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (EXACT) a4[EXACT ? u : 0] = make_uint2(kAgeBg2, kAgeBg2);
                gb[u] = 0; sb[u] = 0; db[u] = 0;
                ac[u] = false;
                if (vl[u]) {
                    const int gy = wy0 + lr, gx = wx0 + (qc << 2);
                    gb[u] = gy * P.cols + gx;
                    sb[u] = SLAB ? (gy - sref.y0) * sref.stride + (gx - sref.x0) : gb[u];
                    db[u] = SLAB ? (gy - dref.y0) * dref.stride + (gx - dref.x0) : gb[u];
                    d4[u] = tile4[qd];
                    ac[u] = (d4[u].x & d4[u].y & d4[u].z & d4[u].w) != kInfBits;   // a finite depth lacks an exponent bit
                    const bool stored = gx >= pw.x && gx < pw.z && gy >= pw.y && gy < pw.w;
#ifndef RBS_EXP_NO_SRCLOAD
                    if (stored && (UPDATE || ac[u])) {
                        s4[u] = *reinterpret_cast<const floatx4*>(src + sb[u]);
                        if (EXACT) a4[EXACT ? u : 0] = *reinterpret_cast<const uint2*>(asrc + sb[u]);
                    } else if (STP && (UPDATE || ac[u])) {
                        s4[u] = *reinterpret_cast<const floatx4*>(P.bgp_src + gb[u]);   // the shared plane's value
                        if (EXACT) a4[EXACT ? u : 0] = *reinterpret_cast<const uint2*>(abgp + gb[u]);
                    }
#endif
                    // (the split launch's likelihood kernel, whose tile read is a memory round trip: requesting the frame's quad WITH it
                    // instead of behind it was measured -- 0.1945 ms either way)
                    if (ac[u]) o4[u] = *reinterpret_cast<const floatx4*>(P.frame + gb[u]);
                }
                qc += rstep; lr += qstep;
                if (qc >= tq) { qc -= tq; ++lr; }
            }
#pragma unroll
            for (int u = 0; u < kScanUnroll; ++u) {
                if (q0 + u * kBlock >= nq) break;   // wave-uniform
                if (EXACT) {
                    // stamped planes: nothing is stepped -- the values travel as they are, the ages advance by the call's elapsed
                    // frames, and an updated pixel's age restarts (its posterior overwrites the value when its batch is evaluated)
                    const uint2 ag = make_uint2(age_add2(a4[EXACT ? u : 0].x, P.elapsed2), age_add2(a4[EXACT ? u : 0].y, P.elapsed2));
                    const bool p0 = d4[u].x != kInfBits && isfinite(o4[u].x), p1 = d4[u].y != kInfBits && isfinite(o4[u].y);
                    const bool p2 = d4[u].z != kInfBits && isfinite(o4[u].z), p3 = d4[u].w != kInfBits && isfinite(o4[u].w);
                    if (UPDATE && vl[u]) {
                        *reinterpret_cast<floatx4*>(dst + db[u]) = s4[u];
                        uint2 w = ag;
                        if (p0) w.x &= 0xffff0000u;
                        if (p1) w.x &= 0x0000ffffu;
                        if (p2) w.y &= 0xffff0000u;
                        if (p3) w.y &= 0x0000ffffu;
                        // (These eight bytes per quad are what the mode costs: the ages' lines are as cold as the values' -- a child's slot was
                        // last written two calls ago -- and the pixel pass is bound by the lines it has in flight.  Measured on C1, raster
                        // kernel 0.166 ms with the device rule: 0.161 without any age traffic (no stepping to do), + 0.009 for the age
                        // loads, + 0.022 for these stores, + 0.006 for the binary64 prior.  Tried and no better: the ages interleaved with
                        // the values in 24-byte cells (0.200: 16-byte accesses at a 24-byte stride), non-temporal stores (0.195 - 0.205),
                        // the stores deferred to one burst behind the pixel pass through the depth tile (0.197 against 0.197).
                        // profiles/r06_exact_occlusion_cost.txt)
                        *reinterpret_cast<uint2*>(adst + db[u]) = w;
                    }
                    if (__ballot(ac[u]) == 0) continue;
                    RBS_PUSH_EVAL(p0, gb[u] + 0, db[u] + 0, d4[u].x, s4[u].x, o4[u].x, ag.x & 0xffffu);
                    RBS_PUSH_EVAL(p1, gb[u] + 1, db[u] + 1, d4[u].y, s4[u].y, o4[u].y, ag.x >> 16);
                    RBS_PUSH_EVAL(p2, gb[u] + 2, db[u] + 2, d4[u].z, s4[u].z, o4[u].z, ag.y & 0xffffu);
                    RBS_PUSH_EVAL(p3, gb[u] + 3, db[u] + 3, d4[u].w, s4[u].w, o4[u].w, ag.y >> 16);
                    continue;
                }
                floatx4 pr;
                pr.x = occ_step(P.alpha, P.beta, s4[u].x, P.bg_new);
                pr.y = occ_step(P.alpha, P.beta, s4[u].y, P.bg_new);
                pr.z = occ_step(P.alpha, P.beta, s4[u].z, P.bg_new);
                pr.w = occ_step(P.alpha, P.beta, s4[u].w, P.bg_new);
#ifndef RBS_EXP_NO_DSTSTORE
                if (UPDATE && vl[u]) *reinterpret_cast<floatx4*>(dst + db[u]) = pr;
#endif
#ifdef RBS_EXP_NO_PUSH
                continue;
#endif
                if (__ballot(ac[u]) == 0) continue;   // wave-uniform: nothing of the object in these 256 pixels
                RBS_PUSH_EVAL(d4[u].x != kInfBits && isfinite(o4[u].x), gb[u] + 0, db[u] + 0, d4[u].x, pr.x, o4[u].x, 0);
                RBS_PUSH_EVAL(d4[u].y != kInfBits && isfinite(o4[u].y), gb[u] + 1, db[u] + 1, d4[u].y, pr.y, o4[u].y, 0);
                RBS_PUSH_EVAL(d4[u].z != kInfBits && isfinite(o4[u].z), gb[u] + 2, db[u] + 2, d4[u].z, pr.z, o4[u].z, 0);
                RBS_PUSH_EVAL(d4[u].w != kInfBits && isfinite(o4[u].w), gb[u] + 3, db[u] + 3, d4[u].w, pr.w, o4[u].w, 0);
            }
        }
    } else {
        // any width: one pixel per lane
        for (int p0 = wave * 64; p0 < npx; p0 += kBlock) {
            const int p = p0 + lane;
            const bool valid = p < npx;
            int gi = 0;
            unsigned dbits = kInfBits;
            float sv = P.bg_old, ov = 0.f;
            if (valid) {
                const int lr = p / tw;
                const int gy = wy0 + lr, gx = wx0 + (p - lr * tw);
                gi = gy * P.cols + gx;
                dbits = PHASE == 2 ? gtile[p] : m.tile[p];
                const bool stored = gx >= pw.x && gx < pw.z && gy >= pw.y && gy < pw.w;
                if (stored && (UPDATE || dbits != kInfBits)) sv = src[gi];
                else if (STP && (UPDATE || dbits != kInfBits)) sv = P.bgp_src[gi];
                if (dbits != kInfBits) ov = P.frame[gi];
            }
            const float prior = occ_step(P.alpha, P.beta, sv, P.bg_new);
            const bool active = dbits != kInfBits && isfinite(ov);
            if (UPDATE && valid && !active) dst[gi] = prior;
            RBS_PUSH_EVAL(active, gi, gi, dbits, prior, ov, 0);   // whole planes only on this path (slabs and stamped planes need cols % 4 == 0)
        }
    }
#undef RBS_PUSH_EVAL
    __builtin_amdgcn_wave_barrier();
    RBS_EVAL_BATCH(qn);
#undef RBS_EVAL_BATCH
    RBS_TICK(3);
    // the next item's ticket is drawn here: its round trip (~2 us) passes while the block's waves
    // gather at the reduction's barrier, instead of after it with everybody waiting
    if (draw && threadIdx.x == 0) ticket = atomicAdd(&P.ctr_this[1], 1);
    const double total = block_reduce_sum(ll, m.red);
    RBS_TICK(4);
    return total;
}

// Is (row, col) written by the raster kernel (inside the particle's rectangle; several bodies:
// inside one of its groups' rectangles)?  q = the particle's union rectangle.
__device__ inline bool raster_writes(const DevParams& P, int particle, const int4& q, int row, int col)
{
    if (!(row >= q.y && row < q.w && col >= q.x && col < q.z)) return false;
    if (P.groups == nullptr) return true;
    const Groups* G = P.groups + particle;
    const int ng = G->n;
    for (int g = 0; g < ng; ++g) {
        const int4 r = G->rect[g];
        if (row >= r.y && row < r.w && col >= r.x && col < r.z) return true;
    }
    return false;
}

// ------------------------------------------------------------------ copy block
// Rows [band*band_rows, ...) of the parent's plane -> the child's slot, skipping the raster
// block's rectangle.  VEC == 4: float4 non-temporal stream; VEC == 1: any cols.
template <int VEC>
__device__ inline void copy_band(const DevParams& P, int particle, int band, Rect r)
{
    const int row0 = band * P.band_rows;
    const int row1 = min(P.rows, row0 + P.band_rows);
    if (row0 >= row1) return;
    const int parent = P.parents[particle];
    if ((unsigned)parent >= (unsigned)P.slots) return;
    const float* __restrict__ src = parent_plane(P, parent) + (size_t)row0 * P.cols;
    float* __restrict__ dst = P.occ_dst + (size_t)particle * P.plane_stride + (size_t)row0 * P.cols;
    const float alpha = P.alpha, beta = P.beta, bg_new = P.bg_new;

    if (VEC == 4) {
        const int W4 = P.cols >> 2;
        const int n4 = (row1 - row0) * W4;
        const floatx4* __restrict__ s4 = reinterpret_cast<const floatx4*>(src);
        floatx4* __restrict__ d4 = reinterpret_cast<floatx4*>(dst);
        const int qstep = kBlock / W4, rstep = kBlock - qstep * W4;
        int row = row0 + (int)threadIdx.x / W4;
        int c4 = (int)threadIdx.x - ((int)threadIdx.x / W4) * W4;
        for (int base = threadIdx.x; base < n4; base += kBlock * kCopyUnroll) {
            floatx4 v[kCopyUnroll];
            bool ok[kCopyUnroll];
#pragma unroll
            for (int k = 0; k < kCopyUnroll; ++k) {
                const int idx = base + k * kBlock;
                const int col = c4 << 2;
                ok[k] = idx < n4 && !raster_writes(P, particle, make_int4(r.x0, r.y0, r.x1, r.y1), row, col);
#if RBS_NT
                if (ok[k]) v[k] = __builtin_nontemporal_load(&s4[idx]);
#else
                if (ok[k]) v[k] = s4[idx];
#endif
                c4 += rstep; row += qstep;
                if (c4 >= W4) { c4 -= W4; ++row; }
            }
#pragma unroll
            for (int k = 0; k < kCopyUnroll; ++k) {
                if (!ok[k]) continue;
                floatx4 w;
                w.x = occ_step(alpha, beta, v[k].x, bg_new);
                w.y = occ_step(alpha, beta, v[k].y, bg_new);
                w.z = occ_step(alpha, beta, v[k].z, bg_new);
                w.w = occ_step(alpha, beta, v[k].w, bg_new);
#if RBS_NT
                __builtin_nontemporal_store(w, &d4[base + k * kBlock]);
#else
                d4[base + k * kBlock] = w;
#endif
            }
        }
    } else {
        const int W = P.cols;
        const int n1 = (row1 - row0) * W;
        for (int idx = threadIdx.x; idx < n1; idx += kBlock) {
            const int lr = idx / W;
            const int row = row0 + lr, col = idx - lr * W;
            if (raster_writes(P, particle, make_int4(r.x0, r.y0, r.x1, r.y1), row, col)) continue;
            dst[idx] = occ_step(alpha, beta, src[idx], bg_new);
        }
    }
}

// ------------------------------------------------------------------ kernels
// One thread per particle: the screen rectangle both the raster and the copy kernel use, and
// the number of tiles it splits into.
// An updating call on windowed planes also fixes the region the copy kernel writes -- the
// bounding box of the parent's window and the rectangle -- and seeds the child's window with the
// rectangle (the copy kernel grows it over every value it writes that differs from the
// background).  An empty window is (cols, rows, 0, 0), so unions are plain min/max.
constexpr int kPrepPerBlock = 8;   // particles (= waves) per block of the rectangles kernel

// Absolute pose of one body from a state delta and the body's default pose:
//   R = R(delta rotation vector) R(default rotation vector),  t = t(delta) + t(default),
// rotation vector -> matrix through the unit quaternion.  The operations and their order are those of
// oracle/tracker_oracle.c orc_compose_poses (and of rbsensor_tracker.hip propagate_body); sin / cos / sqrt are the device library's.
__device__ inline void rotvec_to_matrix_(const double* rv, double* R)
{
    const double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    const double half = 0.5 * angle;
    const double k = angle < 1e-9 ? 0.5 - angle * angle / 48.0 : sin(half) / angle;
    const double w = cos(half), x = rv[0] * k, y = rv[1] * k, z = rv[2] * k;
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
}
__device__ inline void compose_pose(const double* __restrict__ d, const double* __restrict__ d0, double* __restrict__ out)
{
    double Rd[9], R0[9];
    rotvec_to_matrix_(d + 3, Rd);
    rotvec_to_matrix_(d0 + 3, R0);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            out[3 * r + c] = Rd[3 * r] * R0[c] + Rd[3 * r + 1] * R0[3 + c] + Rd[3 * r + 2] * R0[6 + c];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[9 + k] = d[k] + d0[k];
}

// One wave per particle, kPrepPerBlock particles per block.  The work items of the block's
// particles are allotted with ONE atomicAdd per block (any free range will do: a particle's
// items are summed in their own order, so no scan kernel is needed -- but two thousand blocks
// bumping one counter serialise at 11-13 ns each, 26 us at 2 000 particles).
template <bool DELTAS = false>
__device__ inline void prep_particles(const DevParams& P, int block, int* __restrict__ rects, int update)
{
    __shared__ int cnts[kPrepPerBlock + 1];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = block * kPrepPerBlock + w;
    const bool live = i < P.n;
    // (the parent slot is needed last but asked for first: a host-pointer call's indices sit in pinned HOST memory, a PCIe round
    // trip of ~2 us that would otherwise follow the rectangle's arithmetic instead of passing beneath it)
    const int parent_early = live ? P.indices[i] : -1;
    Rect r = {0, 0, 0, 0};
    int cnt = 0;
    // (the per-body rectangles and the particle's groups live in LDS, one set per wave: indexed at run time, arrays of the
    // function's own went to scratch -- 384 bytes per lane, and every step of the merge below a round trip to memory:
    // the rectangles kernel of C2's three bodies took 76 us)
    __shared__ int4 gr_s[kPrepPerBlock][kMaxBodies];
    __shared__ unsigned gm_s[kPrepPerBlock][kMaxBodies];
    __shared__ Groups G_s[kPrepPerBlock];
    __shared__ int ys_s[kPrepPerBlock][2 * kMaxGroups + 4];   // (the strips' band edges: run-time indexed, so not a local array)
    int4* const gr = gr_s[w];
    unsigned* const gm = gm_s[w];
    Groups& G = G_s[w];
    if (lane == 0) G.n = 0;
    const int cap_px = min(P.tile_w * P.tile_h, P.tile_px);
    if (DELTAS && live) {
        // rbs_loglikes_deltas: lane b composes body b's absolute pose from its delta (48 bytes read from pinned host memory) and the
        // body's default pose, into the device array everything else reads -- this wave included, once its own stores have landed
        const int B = P.n_bodies;
        double* dstp = const_cast<double*>(P.poses) + (size_t)i * 12 * B;
        for (int b = lane; b < B; b += 64) {
            double d[6], d0[6], out[12];
#pragma unroll
            for (int k = 0; k < 6; ++k) { d[k] = P.deltas_src[((size_t)i * B + b) * 6 + k]; d0[k] = P.deltas_src[((size_t)P.n * B + b) * 6 + k]; }
            compose_pose(d, d0, out);
#pragma unroll
            for (int k = 0; k < 12; ++k) dstp[12 * b + k] = out[k];
        }
        __threadfence_block();
    } else if (P.poses_src && live) {
        // a host-pointer call: the particle's pose is pulled from pinned host memory (one PCIe read of
        // 96 B per body instead of a copy-engine transfer ahead of this kernel) into the device
        // array everything else reads -- this wave included, once its own stores have landed
        const int m = 12 * P.n_bodies;
        double* dstp = const_cast<double*>(P.poses) + (size_t)i * m;
        const double* srcp = P.poses_src + (size_t)i * m;
        for (int k = lane; k < m; k += 64) dstp[k] = srcp[k];
        __threadfence_block();
    }
    if (live && P.groups == nullptr) {
        r = particle_rect(P, P.poses + (size_t)i * 12 * P.n_bodies);
        const TileGrid tg = tile_grid(r.x1 - r.x0, r.y1 - r.y0, P.tile_w, cap_px);
        cnt = r.x1 > r.x0 ? tg.nx * tg.ny : 1;   // an empty rectangle still owns one (empty) item
    } else if (live) {
        // several bodies: one rectangle per body, overlapping ones merged (all lanes hold the same
        // values and take the same branches)
        // (int4 = x0, y0, x1, y1; every lane of the wave runs the same steps on the same values: LDS reads are broadcasts,
        // writes go through lane 0 with the wave's other lanes waiting at the barrier behind them)
        int ng = 0;
        for (int b = 0; b < P.n_bodies; ++b) {
            const Rect br = bodies_rect(P, P.poses + (size_t)i * 12 * P.n_bodies, b, b + 1);
            if (br.x1 > br.x0) {
                if (lane == 0) { gr[ng] = make_int4(br.x0, br.y0, br.x1, br.y1); gm[ng] = 1u << b; }
                ++ng;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (bool changed = true; changed;) {
            changed = false;
            for (int a = 0; a < ng && !changed; ++a)
                for (int c = a + 1; c < ng && !changed; ++c) {
                    const int4 A = gr[a], C = gr[c];
                    if (A.x < C.z && C.x < A.z && A.y < C.w && C.y < A.w) {
                        const int4 last = gr[ng - 1];
                        const unsigned ma = gm[a], mc = gm[c], ml = gm[ng - 1];
                        __builtin_amdgcn_wave_barrier();          // (every lane has read before lane 0 writes)
                        if (lane == 0) {
                            gr[a] = make_int4(min(A.x, C.x), min(A.y, C.y), max(A.z, C.z), max(A.w, C.w));
                            gm[a] = ma | mc;
                            if (c != ng - 1) { gr[c] = last; gm[c] = ml; }
                        }
                        __builtin_amdgcn_wave_barrier();
                        --ng;
                        changed = true;
                    }
                }
        }
        if (ng > kMaxGroups) {   // more separate objects than groups: one union rectangle
            int4 U = gr[0];
            unsigned mu = gm[0];
            for (int c = 1; c < ng; ++c) {
                const int4 C = gr[c];
                U = make_int4(min(U.x, C.x), min(U.y, C.y), max(U.z, C.z), max(U.w, C.w));
                mu |= gm[c];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { gr[0] = U; gm[0] = mu; }
            __builtin_amdgcn_wave_barrier();
            ng = 1;
        }
        if (lane == 0) G.n = ng;
        if (ng > 0) { const int4 g0 = gr[0]; r = Rect{g0.x, g0.y, g0.z, g0.w}; }
        for (int g = 0; g < ng; ++g) {
            const int4 R = gr[g];
            const TileGrid tg = tile_grid(R.z - R.x, R.w - R.y, P.tile_w, cap_px);
            if (lane == 0) { G.rect[g] = R; G.mask[g] = gm[g]; G.first[g] = cnt; }
            cnt += tg.nx * tg.ny;
            r.x0 = min(r.x0, R.x); r.y0 = min(r.y0, R.y);
            r.x1 = max(r.x1, R.z); r.y1 = max(r.y1, R.w);
        }
        if (ng == 0) cnt = 1;
    }
    if (lane == 0) cnts[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int k = 0; k < kPrepPerBlock; ++k) { const int c = cnts[k]; cnts[k] = total; total += c; }
        cnts[kPrepPerBlock] = total ? atomicAdd(&P.ctr_this[0], total) : 0;
    }
    __syncthreads();
    if (!live || lane != 0) return;
    reinterpret_cast<int4*>(rects)[i] = make_int4(r.x0, r.y0, r.x1, r.y1);
    if (P.groups) {
        Groups* out = P.groups + i;
        out->n = G.n;
        for (int g = 0; g < G.n; ++g) { out->rect[g] = G.rect[g]; out->first[g] = G.first[g]; out->mask[g] = G.mask[g]; }
    }
    P.done[i] = 0;
    const int parent = parent_early;
    P.parents[i] = parent;
    const int first = cnts[kPrepPerBlock] + cnts[w];
    P.item_range[i] = make_int2(first, cnt);
    for (int k = 0; k < cnt; ++k) P.item_particle[first + k] = i;
    if (update && P.windowed) {
        const int4 rw = r.x1 > r.x0 ? make_int4(r.x0, r.y0, r.x1, r.y1) : make_int4(P.cols, P.rows, 0, 0);
        int4 pw = make_int4(P.cols, P.rows, 0, 0);
        if ((unsigned)parent < (unsigned)P.slots) pw = parent_window(P, parent);
        int4 u = make_int4(min(pw.x, rw.x), min(pw.y, rw.y), max(pw.z, rw.z), max(pw.w, rw.w));
        if (P.rebase_box) {   // the shared plane is re-based: every child is re-measured over the rectangle in which it changes
            const int4 bw = *P.rebase_box;
            u = make_int4(min(u.x, bw.x), min(u.y, bw.y), max(u.z, bw.z), max(u.w, bw.w));
        }
        int4 seed = rw;
        if (P.slab_px) {
            // the child's slab stores exactly the region this call writes
            const long area = u.z > u.x && u.w > u.y ? (long)(u.z - u.x) * (long)(u.w - u.y) : 0;
            // the largest region any particle has asked for (err[1]): the host grows the slabs before one
            // overflows.  A plain read first: once the maximum has settled nobody touches the atomic.
            if (area > (long)P.err[1]) atomicMax(P.err + 1, (int)min(area, 0x7fffffffL));
            if (area > (long)P.slab_px) {
                // does not fit: contained like a bad parent slot (the raster and copy kernels skip the
                // particle, its log-likelihood is NaN), its plane becomes all background, and the
                // handle reports the error at its next synchronising call
                P.parents[i] = -1;
                u = make_int4(P.cols, P.rows, 0, 0);
                seed = u;
                atomicExch(P.err, 1);
            }
            P.reg_dst[i] = u;
        }
        P.win_used[i] = u;
        P.win_dst[i] = seed;
        if (P.strips && P.groups) {   // (lane 0 of the particle's wave: at most kMaxGroups rectangles, a few dozen scalar steps)
            Strips* S = P.strips + i;
            int* ys = ys_s[w];
            int ny = 0, ns = 0, total = 0;
            const int ng_ = G.n;
            if (u.z > u.x && u.w > u.y) {
                ys[ny++] = u.y;
                for (int g = 0; g < ng_; ++g) {
                    const int4 R = G.rect[g];
                    for (int e = 0; e < 2; ++e) {
                        const int y = min(max(e ? R.w : R.y, u.y), u.w);
                        int k = ny;                      // insert sorted, drop duplicates
                        while (k > 0 && ys[k - 1] > y) --k;
                        if (k > 0 && ys[k - 1] == y) continue;
                        for (int m = ny; m > k; --m) ys[m] = ys[m - 1];
                        ys[k] = y; ++ny;
                    }
                }
                if (ys[ny - 1] != u.w) ys[ny++] = u.w;
                for (int bnd = 0; bnd + 1 < ny; ++bnd) {
                    const int ya = ys[bnd], yb = ys[bnd + 1];
                    int cur = u.x;
                    for (;;) {     // the rectangles that span this band, left to right (they do not overlap: overlapping ones were merged)
                        int best = -1, bx = u.z;
                        for (int g = 0; g < ng_; ++g) {
                            const int4 R = G.rect[g];
                            if (R.z > R.x && R.y <= ya && R.w >= yb && R.z > cur && max(R.x, u.x) < bx) { best = g; bx = max(R.x, u.x); }
                        }
                        const int x1 = best < 0 ? u.z : bx;
                        if (x1 > cur) {
                            S->first[ns] = total;
                            S->box[ns] = make_ushort4((unsigned short)(cur >> 2), (unsigned short)(x1 >> 2), (unsigned short)ya, (unsigned short)yb);
                            total += ((x1 - cur) >> 2) * (yb - ya);
                            ++ns;
                        }
                        if (best < 0) break;
                        cur = max(cur, min(G.rect[best].z, u.z));
                        if (cur >= u.z) break;
                    }
                }
            }
            S->first[ns] = total;
            S->n = ns;
        }
        // (every 8th particle: the sum only feeds an estimate of the stored fraction of a plane, and one atomic per
        // particle on one address serialises at ~12 ns each -- 24 us of a sampled call's rectangles kernel at 2 000
        // particles, 250 us at 20 000)
        if (P.area_sum && (i & 7) == 0 && u.z > u.x && u.w > u.y)
            atomicAdd(P.area_sum, (unsigned long long)(u.z - u.x) * (unsigned long long)(u.w - u.y));
    }
}


__global__ __launch_bounds__(64 * kPrepPerBlock) void rbs_prep_kernel(const DevParams P, int* __restrict__ rects, int update)
{
    prep_particles(P, (int)blockIdx.x, rects, update);
}
// ... composing the poses from state deltas first (rbs_loglikes_deltas): a kernel of its own, so that the sin / cos it carries
// cost the other calls' rectangles kernel neither registers nor occupancy
__global__ __launch_bounds__(64 * kPrepPerBlock) void rbs_prep_deltas_kernel(const DevParams P, int* __restrict__ rects, int update)
{
    prep_particles<true>(P, (int)blockIdx.x, rects, update);
}

// A frame handed over right before this call: its per-pixel terms (aux_blocks blocks) and the
// particles' rectangles (the other blocks) are independent, one launch.
__global__ __launch_bounds__(64 * kPrepPerBlock) void rbs_frame_prep_kernel(const DevParams P, int* __restrict__ rects, int update,
                                      const float* __restrict__ frame_src, double* __restrict__ aux,
                                      float* __restrict__ pbg, float* __restrict__ keep, int aux_blocks)
{
    // the rectangle blocks come first: their threads run a long serial FP64 chain and should start
    // at once, the short per-pixel blocks fill in around them
    const int prep_blocks = (int)gridDim.x - aux_blocks;
    if ((int)blockIdx.x < prep_blocks) {
        prep_particles(P, (int)blockIdx.x, rects, update);
    } else {
        const int i = ((int)blockIdx.x - prep_blocks) * (int)blockDim.x + (int)threadIdx.x;
        if (i < P.npx) frame_aux_pixel(i, frame_src, aux, pbg, P.npx, P.tw, P.ms, P.sf, P.lambda, keep);
    }
}

// Three 4-wave blocks per CU = 3 waves/SIMD = 168 VGPRs (a few spilled dwords).
#ifndef RBS_RASTER_MINWAVES_F64
#define RBS_RASTER_MINWAVES_F64 3
#endif
#ifndef RBS_RASTER_MINWAVES
#define RBS_RASTER_MINWAVES 3
#endif
// Persistent: a fixed number of raster blocks per CU keeps its LDS/VGPR share for the whole
// launch (never starved by the many small copy blocks) and pulls (particle, tile) items from
// an atomic queue.
constexpr size_t smem_bytes(int tile_px, bool math_tables, bool many = false, bool exact = false);
template <bool UPDATE, int PREC, bool SLAB, bool MANY = false, bool STP = false, bool EXACT = false, bool ONE = false>
__device__ __forceinline__ void raster_kernel_body(const DevParams& P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Smem m = carve(smem, P.tile_px, PREC == 0 && RBS_MATH_LDS);
    if (EXACT) {
        m.ageq = reinterpret_cast<int*>(smem + smem_bytes(P.tile_px, PREC == 0 && RBS_MATH_LDS, MANY));
        m.agetab = reinterpret_cast<double*>(m.ageq + (kBlock / 64) * kEvalQueue);
        for (int i = threadIdx.x; i < 2 * min(kLdsAges, P.age_max + 1); i += kBlock) m.agetab[i] = P.ptab[i];
    }
    const int total = P.ctr_this[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { P.ctr_next[0] = 0; P.ctr_next[1] = 0; }
    if (PREC == 0 && RBS_MATH_LDS) {   // once per persistent block (the first item's tile clear ends in a barrier)
        constexpr int ne = rbsm::kErfcIntervals * rbsm::kErfcCoefs, nl = rbsm::kLogIntervals * 2;
        for (int i = threadIdx.x; i < ne; i += kBlock) m.mtab[i] = rbsm::kErfcTab[i];
        for (int i = threadIdx.x; i < nl; i += kBlock) m.mtab[ne + i] = rbsm::kLogTab[i];
    }
#ifdef RBS_PHASE_TIMING
    const unsigned long long c0_ = clock64(), w0_ = wall_clock64();
    if (threadIdx.x < 32) g_phase_lds[threadIdx.x] = 0;
#endif
    RBS_TICK_DECL;
    // The first round is dealt statically (block b takes item b): 768 blocks drawing their first
    // ticket at the same instant serialise on the counter, up to 10 us.  Everything after goes
    // through the ticket counter -- dealing further whole rounds statically as well (every block
    // runs that many items under any schedule) gains nothing at 2 000 particles and costs 8-25 %
    // where items differ (several tiles per particle, several bodies: C2, C4, 20 000 particles).
    const int grid = (int)gridDim.x;
    const int static_end = grid;
    int item = (int)blockIdx.x;
    for (;;) {
        if (item >= total) break;
        RBS_TICK(7);
        // wave-uniform: tell the compiler, so the pose / rectangle / parent index become scalar
        // loads held in SGPRs instead of per-lane vector loads in every loop
        const int particle = __builtin_amdgcn_readfirstlane(P.item_particle[item]);
        const int4 q = reinterpret_cast<const int4*>(P.rects)[particle];
        const Rect r = {q.x, q.y, q.z, q.w};
        const int2 range = P.item_range[particle];
        const int first = range.x;
        double part = 0.0;
        const bool draw = item + grid >= static_end;   // the next item comes from the ticket counter
        int ticket = -1;
#ifdef RBS_PHASE_TIMING
        if (threadIdx.x == 0 && q.x + range.y == -12345) P.out[0] = 0.0;   // (waits for the descriptor's loads)
        RBS_TICK(15);   // the item's descriptor
#endif
        if (ONE || P.groups == nullptr) {
            if (r.x1 > r.x0) part = raster_eval_tile<UPDATE, PREC, SLAB, MANY, 0, STP, EXACT, ONE>(P, particle, r, item - first, m, 0xffffffffu, draw, ticket);
        } else {   // several bodies: the item belongs to one group of bodies with its own rectangle
            const Groups* G = P.groups + particle;
            const int k = item - first, ng = __builtin_amdgcn_readfirstlane(G->n);
            int g = 0;
            for (int c = 1; c < ng; ++c) g += (k >= __builtin_amdgcn_readfirstlane(G->first[c])) ? 1 : 0;
            if (ng > 0) {
                const int4 gq = G->rect[g];
                const Rect gr = {__builtin_amdgcn_readfirstlane(gq.x), __builtin_amdgcn_readfirstlane(gq.y),
                                 __builtin_amdgcn_readfirstlane(gq.z), __builtin_amdgcn_readfirstlane(gq.w)};
                part = raster_eval_tile<UPDATE, PREC, SLAB, MANY, 0, STP, EXACT>(P, particle, gr, k - __builtin_amdgcn_readfirstlane(G->first[g]), m,
                                                      (unsigned)__builtin_amdgcn_readfirstlane((int)G->mask[g]), draw, ticket);
            }
        }
        if (threadIdx.x == 0) {
            // the particle's log-likelihood: its only item's sum, or -- by whichever block
            // finishes the particle's last item -- the items' sums added in item order.  The
            // partial sums travel through device-scope atomics (performed at the memory side, so
            // they need no fence: a release fence here would write the whole L2 back, and this
            // kernel keeps the planes it is writing there); the exchange has returned before the
            // counter is bumped.
            const int cnt = range.y;
            if (cnt == 1) {
                P.out[particle] = part;
            } else {
                unsigned long long* pp = reinterpret_cast<unsigned long long*>(P.partial);
                const unsigned long long old = atomicExch(pp + item, (unsigned long long)__double_as_longlong(part));
                const int bump = old == 0xfff8deadbeef0000ull ? 2 : 1;   // never 2: ties the counter to the returned value
                if (atomicAdd(&P.done[particle], bump) == cnt - 1) {
                    double sum = 0.0;
                    for (int k = 0; k < cnt; ++k) sum += __longlong_as_double((long long)atomicOr(pp + first + k, 0ull));
                    P.out[particle] = sum;
                }
            }
        }
        if (draw) {
            if (threadIdx.x == 0) {
                if (ticket < 0) ticket = atomicAdd(&P.ctr_this[1], 1);   // (an item that ended before its reduction)
                *m.item = static_end + ticket;
            }
            __syncthreads();
            item = __builtin_amdgcn_readfirstlane(*m.item);   // (written again at the end of the next item, many barriers away)
        } else {
            item += grid;
            __syncthreads();
        }
#ifdef RBS_PHASE_TIMING
        tick_ = clock64();   // (the item's own phases were clocked inside)
#endif
    }
#ifdef RBS_PHASE_TIMING
    if (threadIdx.x == 0) {   // block lifetime in shader cycles (clock64) and in 100 MHz wall ticks
        for (int k = 0; k < 32; ++k)
            if (k != 5 && k != 6) atomicAdd(&P.phase[k], g_phase_lds[k]);
        atomicAdd(&P.phase[5], clock64() - c0_);
        atomicAdd(&P.phase[6], wall_clock64() - w0_);
    }
#endif
}

// The two precisions are two kernels: the register budget is an attribute and cannot depend on a
// template parameter.
template <bool UPDATE, bool SLAB>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES) __attribute__((amdgpu_num_vgpr(RBS_RASTER_VGPRS)))
void rbs_raster_kernel_f32(const DevParams P)
{
    raster_kernel_body<UPDATE, 1, SLAB>(P);
}
// The same budget for the binary64 likelihood: on rbs_math.h's functions it fits 160 registers
// (a few kernel-lifetime dwords in scratch at most), so the windowed copy kernel runs beside it as well (C1 9.27 -> 9.72 M/s, C2 3.24
// -> 3.45; ocml's exp / erf / log needed 168 + 72 spilled).  RBS_RASTER_VGPRS_F64=0: no budget.
#ifndef RBS_RASTER_VGPRS_F64
#define RBS_RASTER_VGPRS_F64 80
#endif
#if RBS_RASTER_VGPRS_F64
#define RBS_F64_BUDGET __attribute__((amdgpu_num_vgpr(RBS_RASTER_VGPRS_F64)))
#else
#define RBS_F64_BUDGET
#endif
template <bool UPDATE, bool SLAB>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES_F64) RBS_F64_BUDGET void rbs_raster_kernel_f64(const DevParams P)
{
    raster_kernel_body<UPDATE, 0, SLAB>(P);
}
// ... an object model of ONE body (round 6, VERDICT r5 #7: the bounded experiment -- body loop, body masks, groups and the per-body
// tables' run-time indexing gone from the kernel).  Measured on C1 (tools/dbg/ab_one_body.sh, same box, two rounds each): spilled
// scalar registers 59 -> 48, kernel 0.1662 -> 0.1654 ms; with the registers that frees spent on the pixel pass' loads in flight
// (RBS_SCAN_UNROLL_ONE): three quads 0.1640 (44 spilled SGPRs, nothing in scratch), four 0.1647.  The target was 0.158: the kernel's
// waiting is not the scalar spills'.  Kept at three quads for its 1.3 % (parity suite green on it); the C1 kernel is finished here.
template <bool UPDATE, bool SLAB>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES_F64) RBS_F64_BUDGET void rbs_raster_kernel_one_f64(const DevParams P)
{
    raster_kernel_body<UPDATE, 0, SLAB, false, false, false, true>(P);
}
// ... with the shared background plane (STP: binary64; whole planes or slabs -- the shared plane itself is always a whole plane,
// addressed by frame offsets), kernels of their own so that the others stay as they are
template <bool UPDATE, bool SLAB, bool MANY>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES_F64) RBS_F64_BUDGET void rbs_raster_kernel_stp_f64(const DevParams P)
{
    raster_kernel_body<UPDATE, 0, SLAB, MANY, true>(P);
}
// ... the same for the float32 likelihood (round 6: the shared trail no longer needs the binary64 likelihood)
template <bool UPDATE, bool SLAB, bool MANY>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES) __attribute__((amdgpu_num_vgpr(RBS_RASTER_VGPRS)))
void rbs_raster_kernel_stp_f32(const DevParams P)
{
    raster_kernel_body<UPDATE, 1, SLAB, MANY, true>(P);
}
// ... on STAMPED planes (EXACT: the reference's occlusion bookkeeping; binary64; all of SLAB / MANY / STP), kernels of their own again
template <bool UPDATE, bool SLAB, bool MANY, bool STP>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES_F64) RBS_F64_BUDGET void rbs_raster_kernel_exact_f64(const DevParams P)
{
    raster_kernel_body<UPDATE, 0, SLAB, MANY, STP, true>(P);
}
// ... stamped planes, object model of ONE body (the specialisation of rbs_raster_kernel_one_f64)
template <bool UPDATE, bool SLAB, bool STP>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES_F64) RBS_F64_BUDGET void rbs_raster_kernel_exact_one_f64(const DevParams P)
{
    raster_kernel_body<UPDATE, 0, SLAB, false, STP, true, true>(P);
}
// ... and the same two for object models with a body of more than 256 clusters (MANY: the shared cluster cull).
template <bool UPDATE, bool SLAB>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES) __attribute__((amdgpu_num_vgpr(RBS_RASTER_VGPRS)))
void rbs_raster_kernel_many_f32(const DevParams P)
{
    raster_kernel_body<UPDATE, 1, SLAB, true>(P);
}
template <bool UPDATE, bool SLAB>
__global__ __launch_bounds__(kBlock, RBS_RASTER_MINWAVES_F64) RBS_F64_BUDGET void rbs_raster_kernel_many_f64(const DevParams P)
{
    raster_kernel_body<UPDATE, 0, SLAB, true>(P);
}

// ------------------------------------------------------------------ split launch (round 5)
// The same work as rbs_raster_kernel_f64 in TWO kernels, each with a register budget and an occupancy of its own:
//   rbs_depth_kernel   persistent; per work item cluster cull + triangle setup + sample loops into the LDS depth tile
//                      (raster_window, unchanged: depth stays bit-exact), then the tile is written to P.depth.  Carries
//                      neither the likelihood's registers nor its math tables nor the evaluation rings.  Needs no
//                      frame: a host frame is staged and travels while it runs.
//   rbs_eval_kernel    per work item the pixel pass of raster_eval_tile with the tile read from P.depth (L2 / MALL):
//                      occlusion process, compaction of covered + observed pixels, the binary64 likelihood 64 pixels
//                      at a time, block reduce, plane write.  15 KB of LDS and a small register budget: 5-6 waves per
//                      SIMD hide the 32-byte gather and the exp / erfc / log chains that three waves do not.
#ifndef RBS_SPLIT_DEFAULT
#define RBS_SPLIT_DEFAULT 0         // what a handle does when RBS_SPLIT is not in the environment
#endif
// Measured (round 5, profiles/r05_split_*): four geometry blocks per CU need a 7 680-px tile and a 120-register budget that
// spills 55 registers -- 126 us; three blocks per CU at the one-kernel launch's own tile and budget: 105 us.  The tile of the
// one-kernel launch also makes the two launches interchangeable call by call: the same work items, the same summation order,
// the same bits.
#ifndef RBS_DEPTH_TILE_PX
#define RBS_DEPTH_TILE_PX kTilePxF64
#endif
#ifndef RBS_DEPTH_MINWAVES
#define RBS_DEPTH_MINWAVES 3
#endif
#ifndef RBS_DEPTH_VGPRS
#define RBS_DEPTH_VGPRS 80          // 160 registers: three waves per SIMD and 32 left for one wave of the windowed copy kernel
#endif
#ifndef RBS_EVAL_MINWAVES
#define RBS_EVAL_MINWAVES 4         // (5 and 6 waves per SIMD run it no faster -- it waits for memory, section 4 of DESIGN.md -- and 128 registers leave the compiler room)
#endif
constexpr int kDepthTilePx = RBS_DEPTH_TILE_PX;
static_assert(kDepthTilePx == kTilePxF64 || RBS_DEPTH_MINWAVES != 3, "the split launch shares the one-kernel launch's tile");
struct SmemDepth { unsigned* tile; int* big; int* nbig; int* item; int* tq; unsigned long long* cull; };
__device__ inline SmemDepth carve_depth(unsigned char* smem, int tile_px)
{
    SmemDepth m;
    m.tile = reinterpret_cast<unsigned*>(smem);
    m.big = reinterpret_cast<int*>(smem + sizeof(unsigned) * tile_px);
    m.nbig = m.big + kBigCap;
    m.item = m.nbig + 1;
    m.tq = m.nbig + 4;                                   // per wave: kTq triangle indices
    m.cull = reinterpret_cast<unsigned long long*>(m.tq + kTq * (kBlock / 64));
    return m;
}
constexpr size_t smem_bytes_depth(int tile_px, bool many)
{
    return sizeof(unsigned) * (size_t)tile_px + sizeof(int) * kBigCap + 16 + sizeof(int) * kTq * (kBlock / 64) + (many ? 16 * kCullSteps : 0);
}
static_assert(RBS_DEPTH_MINWAVES * ((smem_bytes_depth(kDepthTilePx, true) + 1279) / 1280 * 1280) <= 160 * 1024, "the geometry kernel's blocks per CU");

template <bool MANY>
__device__ __forceinline__ void depth_kernel_body(const DevParams& P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SmemDepth m = carve_depth(smem, P.tile_px);
    const int total = P.ctr_this[0];
    // (the likelihood kernel's ticket counter is zeroed for THIS call, which runs behind this kernel on the same stream and is
    // the only one to use it: a handle mixes the two launch forms call by call, and a one-kernel call in between would leave a
    // counter that only "the next split call" zeroes holding a stale count -- ADVICE r5)
    if (blockIdx.x == 0 && threadIdx.x == 0) { P.ctr_next[0] = 0; P.ctr_next[1] = 0; *P.ctrb_this = 0; *P.ctrb_next = 0; }
    const int grid = (int)gridDim.x;
    int item = (int)blockIdx.x;   // the first round is dealt statically, the rest through the ticket counter (as in the monolith)
    for (;;) {
        if (item >= total) break;
        const int particle = __builtin_amdgcn_readfirstlane(P.item_particle[item]);
        const int first = P.item_range[particle].x;
        Rect r;
        unsigned body_mask = 0xffffffffu;
        int tile_id = item - first;
        bool have;
        if (P.groups == nullptr) {
            const int4 q = reinterpret_cast<const int4*>(P.rects)[particle];
            r = Rect{q.x, q.y, q.z, q.w};
            have = r.x1 > r.x0;
        } else {
            const Groups* G = P.groups + particle;
            const int ng = __builtin_amdgcn_readfirstlane(G->n);
            int g = 0;
            for (int c = 1; c < ng; ++c) g += (tile_id >= __builtin_amdgcn_readfirstlane(G->first[c])) ? 1 : 0;
            have = ng > 0;
            r = Rect{0, 0, 0, 0};
            if (have) {
                const int4 gq = G->rect[g];
                r = Rect{__builtin_amdgcn_readfirstlane(gq.x), __builtin_amdgcn_readfirstlane(gq.y),
                         __builtin_amdgcn_readfirstlane(gq.z), __builtin_amdgcn_readfirstlane(gq.w)};
                tile_id -= __builtin_amdgcn_readfirstlane(G->first[g]);
                body_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)G->mask[g]);
            }
        }
        const int parent = P.parents[particle];
        if (have && (unsigned)parent < (unsigned)P.slots && item < P.depth_items) {
            const TileGrid tg = tile_grid(r.x1 - r.x0, r.y1 - r.y0, P.tile_w, min(P.tile_w * P.tile_h, P.tile_px));
            const int ty = tile_id / tg.nx, tx = tile_id - ty * tg.nx;
            const int wx0 = r.x0 + tx * tg.tw, wy0 = r.y0 + ty * tg.th;
            const int wx1 = min(r.x1, wx0 + tg.tw), wy1 = min(r.y1, wy0 + tg.th);
            const int npx = (wx1 - wx0) * (wy1 - wy0);
            const bool whole = (wx0 == r.x0 && wy0 == r.y0 && wx1 == r.x1 && wy1 == r.y1);
            const double* pose = P.poses + (size_t)particle * 12 * P.n_bodies;
            raster_window<MANY>(P, pose, wx0, wy0, wx1, wy1, !whole || (body_mask & (body_mask - 1u)) != 0u, m.tile, m.big, m.nbig,
                                m.tq + (threadIdx.x >> 6) * kTq, body_mask, m.cull);
            // the tile leaves through 16-byte stores (rectangles move in float4 columns whenever cols % 4 == 0)
            unsigned* __restrict__ out = P.depth + (size_t)item * P.tile_px;
            if ((npx & 3) == 0) {
                const uint4* __restrict__ t4 = reinterpret_cast<const uint4*>(m.tile);
                uint4* __restrict__ o4 = reinterpret_cast<uint4*>(out);
                for (int k = threadIdx.x; k < (npx >> 2); k += kBlock) o4[k] = t4[k];
            } else {
                for (int k = threadIdx.x; k < npx; k += kBlock) out[k] = m.tile[k];
            }
        }
        if (threadIdx.x == 0) *m.item = grid + atomicAdd(&P.ctr_this[1], 1);
        __syncthreads();   // (also: the tile has been read before the next item clears it)
        item = __builtin_amdgcn_readfirstlane(*m.item);
        __syncthreads();   // (... and read by everybody before thread 0 writes the next one)
    }
}
template <bool MANY>
__global__ __launch_bounds__(kBlock, RBS_DEPTH_MINWAVES) __attribute__((amdgpu_num_vgpr(RBS_DEPTH_VGPRS)))
void rbs_depth_kernel(const DevParams P)
{
    depth_kernel_body<MANY>(P);
}

// The likelihood half.  Block per work item (first round dealt statically, then tickets); the particle's sum as in
// the monolith: its only item's, or the items' partial sums added in item order by whoever finishes last.
template <bool UPDATE, bool SLAB, bool STP = false, bool EXACT = false>
__global__ __launch_bounds__(kBlock, RBS_EVAL_MINWAVES) void rbs_eval_kernel(const DevParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Smem m = carve(smem, 0, true);
    if (EXACT) {
        m.ageq = reinterpret_cast<int*>(smem + smem_bytes(0, true, false));
        m.agetab = reinterpret_cast<double*>(m.ageq + (kBlock / 64) * kEvalQueue);
        for (int i = threadIdx.x; i < 2 * min(kLdsAges, P.age_max + 1); i += kBlock) m.agetab[i] = P.ptab[i];
    }
    const int total = P.ctr_this[0];
    {
        constexpr int ne = rbsm::kErfcIntervals * rbsm::kErfcCoefs, nl = rbsm::kLogIntervals * 2;
        for (int i = threadIdx.x; i < ne; i += kBlock) m.mtab[i] = rbsm::kErfcTab[i];
        for (int i = threadIdx.x; i < nl; i += kBlock) m.mtab[ne + i] = rbsm::kLogTab[i];
    }
    __syncthreads();
    const int grid = (int)gridDim.x;
    int item = (int)blockIdx.x;
    for (;;) {
        if (item >= total) break;
        const int particle = __builtin_amdgcn_readfirstlane(P.item_particle[item]);
        const int2 range = P.item_range[particle];
        const int first = range.x;
        double part = 0.0;
        int ticket = -1;
        const unsigned* gtile = P.depth + (size_t)item * P.tile_px;
        if (item >= P.depth_items) {
            part = NAN;   // (the hand-over buffer was sized for fewer items: contained like a bad parent slot)
        } else if (P.groups == nullptr) {
            const int4 q = reinterpret_cast<const int4*>(P.rects)[particle];
            const Rect r = {q.x, q.y, q.z, q.w};
            if (r.x1 > r.x0) part = raster_eval_tile<UPDATE, 0, SLAB, false, 2, STP, EXACT>(P, particle, r, item - first, m, 0xffffffffu, false, ticket, gtile);
        } else {
            const Groups* G = P.groups + particle;
            const int k = item - first, ng = __builtin_amdgcn_readfirstlane(G->n);
            int g = 0;
            for (int c = 1; c < ng; ++c) g += (k >= __builtin_amdgcn_readfirstlane(G->first[c])) ? 1 : 0;
            if (ng > 0) {
                const int4 gq = G->rect[g];
                const Rect gr = {__builtin_amdgcn_readfirstlane(gq.x), __builtin_amdgcn_readfirstlane(gq.y),
                                 __builtin_amdgcn_readfirstlane(gq.z), __builtin_amdgcn_readfirstlane(gq.w)};
                part = raster_eval_tile<UPDATE, 0, SLAB, false, 2, STP, EXACT>(P, particle, gr, k - __builtin_amdgcn_readfirstlane(G->first[g]), m,
                                                                   (unsigned)__builtin_amdgcn_readfirstlane((int)G->mask[g]), false, ticket, gtile);
            }
        }
        if (threadIdx.x == 0) {
            const int cnt = range.y;
            if (cnt == 1) {
                P.out[particle] = part;
            } else {   // (see raster_kernel_body: partial sums through device-scope atomics, added in item order)
                unsigned long long* pp = reinterpret_cast<unsigned long long*>(P.partial);
                const unsigned long long old = atomicExch(pp + item, (unsigned long long)__double_as_longlong(part));
                const int bump = old == 0xfff8deadbeef0000ull ? 2 : 1;
                if (atomicAdd(&P.done[particle], bump) == cnt - 1) {
                    double sum = 0.0;
                    for (int k = 0; k < cnt; ++k) sum += __longlong_as_double((long long)atomicOr(pp + first + k, 0ull));
                    P.out[particle] = sum;
                }
            }
            *m.item = grid + atomicAdd(P.ctrb_this, 1);
        }
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(*m.item);
        __syncthreads();
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void rbs_copy_kernel(const DevParams P)
{
    const int items = P.n * P.bands;
    for (int w = (int)blockIdx.x; w < items; w += (int)gridDim.x) {
        const int particle = w / P.bands;
        const int band = w - particle * P.bands;
        const int4 q = reinterpret_cast<const int4*>(P.rects)[particle];
        const Rect r = {q.x, q.y, q.z, q.w};
        copy_band<VEC>(P, particle, band, r);
    }
}

// Lean row-structured stream (cols % 4 == 0): one block = one column segment (one wave = 64
// float4 by default) of ROWS image rows of one particle, one float4 column per lane, so there is
// no index arithmetic beyond the rectangle test.  Blocks are tiny (1-2 KB) and dispatched in
// address order (x = segment, then row group fastest), which keeps
// the chip-wide HBM access stream nearly sequential -- measured 6.2-6.6 TB/s for a plain copy
// of this shape vs 5.3-5.8 TB/s for 32-64 KB per block (tools/copybench2.hip).
// WIN: the planes are windowed but the windows have grown to a large part of the frame, where
// this streaming shape beats rbs_copy_window_kernel: a float4 outside the parent's window is the
// background instead of a load, and every block leaves a flag "wrote something that differs from
// the background" from which rbs_wide_window_kernel rebuilds the child's window afterwards (a
// million atomics on the windows themselves serialise: 19 ms).
template <int ROWS, bool WIN>
__global__ __launch_bounds__(1024) void rbs_copy_rows_kernel(const DevParams P, int nseg)
{
    const int particle = (int)blockIdx.y + (int)blockIdx.z * (int)gridDim.y;
    if (particle >= P.n) return;
    const int W4 = P.cols >> 2;
    const int4 q = reinterpret_cast<const int4*>(P.rects)[particle];
    const int parent = P.parents[particle];
    if ((unsigned)parent >= (unsigned)P.slots) return;
    int4 pw = make_int4(0, 0, P.cols, P.rows);
    if (WIN) pw = parent_window(P, parent);
    // blockIdx.x = row group * nseg + column segment (segment fastest: address order)
    const int rg = (int)blockIdx.x / nseg;
    const int seg = (int)blockIdx.x - rg * nseg;
    const int r0 = rg * ROWS;
    const int c4 = seg * (int)blockDim.x + (int)threadIdx.x;
    if (c4 >= W4) return;
    const floatx4* __restrict__ s4 = reinterpret_cast<const floatx4*>(parent_plane(P, parent));
    floatx4* __restrict__ d4 = reinterpret_cast<floatx4*>(P.occ_dst + (size_t)particle * P.plane_stride);
    const float alpha = P.alpha, beta = P.beta, bg_new = P.bg_new;
    const int col = c4 << 2;
    const bool in_cols = col >= q.x && col < q.z;
    floatx4 v[ROWS];
    bool ok[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int row = r0 + k;
        ok[k] = row < P.rows && !(in_cols && raster_writes(P, particle, q, row, col));
        const bool stored = !WIN || (col >= pw.x && col < pw.z && row >= pw.y && row < pw.w);
        v[k].x = v[k].y = v[k].z = v[k].w = P.bg_old;
        if (ok[k] && stored) v[k] = __builtin_nontemporal_load(&s4[(size_t)row * W4 + c4]);
    }
    int by0 = P.rows, by1 = 0;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        if (!ok[k]) continue;
        floatx4 w;
        w.x = occ_step(alpha, beta, v[k].x, bg_new);
        w.y = occ_step(alpha, beta, v[k].y, bg_new);
        w.z = occ_step(alpha, beta, v[k].z, bg_new);
        w.w = occ_step(alpha, beta, v[k].w, bg_new);
        __builtin_nontemporal_store(w, &d4[(size_t)(r0 + k) * W4 + c4]);
        if (WIN && (w.x != bg_new || w.y != bg_new || w.z != bg_new || w.w != bg_new)) {
            by0 = min(by0, r0 + k);
            by1 = max(by1, r0 + k + 1);
        }
    }
    if (WIN) {
        const unsigned long long any = __ballot(by1 > by0);
        if ((threadIdx.x & 63) == 0)
            P.wide_flags[(size_t)particle * gridDim.x + blockIdx.x] = any ? 1 : 0;
    }
}

// After a wide-window copy: the child's window = its rectangle (seeded by the rectangles kernel)
// grown over the flagged copy blocks (2 rows x 256 columns each).  One wave per particle.
__global__ __launch_bounds__(64) void rbs_wide_window_kernel(const DevParams P, int nseg, int nblk)
{
    const int i = (int)blockIdx.x;
    if (i >= P.n) return;
    const unsigned char* f = P.wide_flags + (size_t)i * nblk;
    int x0 = P.cols, y0 = P.rows, x1 = 0, y1 = 0;
    for (int b = (int)threadIdx.x; b < nblk; b += 64) {
        if (!f[b]) continue;
        const int rg = b / nseg, seg = b - rg * nseg;
        x0 = min(x0, seg * 256); x1 = max(x1, min(P.cols, seg * 256 + 256));
        y0 = min(y0, rg * 2);    y1 = max(y1, min(P.rows, rg * 2 + 2));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
        x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
    }
    if (threadIdx.x == 0 && x1 > x0) {
        const int4 w = P.win_dst[i];
        P.win_dst[i] = make_int4(min(w.x, x0), min(w.y, y0), max(w.z, x1), max(w.w, y1));
    }
}

// Windowed planes: the child's plane is written over win_used = bbox(parent window, rectangle)
// minus the rectangle (the raster kernel's share).  A float4 inside the parent's window is read
// and stepped; one outside it is the background.  blockIdx.x splits the region's rows into
// gridDim.x chunks; one wave per block, lanes run over the chunk's float4s row-major.  Every
// written float4 that differs from the background grows the child's window (atomic min/max once
// per wave), so a window shrinks again as soon as the values it held have decayed into the
// background snap.
// Two float4 loads in flight per lane = 32 VGPRs: exactly what three resident raster waves
// (160 each, by a stated budget) leave free on a SIMD, so one copy wave per SIMD runs BESIDE the
// persistent raster blocks instead of only before and after them (with four loads, 48 VGPRs, it
// could not: C2 3.33 -> 3.61 M/s, C3 slice 12.0 -> 12.8 M/s, C1 step 0.2055 -> 0.2015 ms).
#ifndef RBS_WIN_UNROLL
#define RBS_WIN_UNROLL 2
#endif
#ifndef RBS_COPY_STRIPS
#define RBS_COPY_STRIPS 1      // (0: one body takes the walk-everything form as well -- A/B)
#endif
constexpr int kWinUnroll = RBS_WIN_UNROLL;
// STRIPS: one rectangle per particle (P.groups == nullptr) -- only the cells outside it are enumerated; otherwise the
// whole window is walked and the cells the raster kernel writes are skipped (several bodies: up to kMaxGroups rectangles).
// STP: the shared background PLANE (DevParams.bgp_src / bgp_dst) stands where the scalar background stands -- a cell outside the
// parent's window holds bgp_src there, and the child's window grows over the cells that differ from bgp_dst.  One cell in flight
// per lane (its shared-plane values take the registers of the second).
// EXACT: stamped planes (DevParams.exact) -- a cell is four values and four ages; nothing is stepped: the values are copied, the
// ages advance by the call's elapsed frames, and the child's window grows over the cells that hold an age within age_max (STP:
// that differ from the shared plane's new cell).
// STRIPS == 2 (round 6): several bodies -- the cells come from the strip list the rectangles kernel made (DevParams::strips).
template <bool SLAB, int STRIPS, bool STP = false, bool EXACT = false>
__global__ __launch_bounds__(64) void rbs_copy_window_kernel(const DevParams P)
{
    constexpr int kU = STP ? 1 : kWinUnroll;
    const floatx4* __restrict__ bc4 = reinterpret_cast<const floatx4*>(P.bgp_src);
    const floatx4* __restrict__ bn4 = reinterpret_cast<const floatx4*>(P.bgp_dst);
    const uint2* __restrict__ bca = EXACT && STP ? reinterpret_cast<const uint2*>(P.bgp_src + P.npx) : nullptr;   // the shared plane's ages
    const uint2* __restrict__ bna = EXACT && STP ? reinterpret_cast<const uint2*>(P.bgp_dst + P.npx) : nullptr;
    const int particle = (int)blockIdx.y + (int)blockIdx.z * (int)gridDim.y;
    if (particle >= P.n) return;
    const int parent = P.parents[particle];
    if ((unsigned)parent >= (unsigned)P.slots) return;
    const int4 u = P.win_used[particle];
    if (u.z <= u.x || u.w <= u.y) return;
    const int4 q = reinterpret_cast<const int4*>(P.rects)[particle];
    if (q.z > q.x && q.x == u.x && q.y == u.y && q.z == u.z && q.w == u.w &&
        (P.groups == nullptr || (P.groups[particle].n == 1)))
        return;   // all raster's
    const int4 pw = parent_window(P, parent);
    const int w4 = (u.z - u.x) >> 2, ux4 = u.x >> 2, W4 = P.cols >> 2;
    const int rpc = (u.w - u.y + (int)gridDim.x - 1) / (int)gridDim.x;
    const int ry0 = u.y + (int)blockIdx.x * rpc, ry1 = min(u.w, ry0 + rpc);
    if (!STRIPS && ry0 >= ry1) return;   // (STRIPS: the blocks share CELLS, not rows -- below)
    const int n4 = max(ry1 - ry0, 0) * w4;
    const floatx4* __restrict__ s4 = reinterpret_cast<const floatx4*>(parent_plane(P, parent));
    floatx4* __restrict__ d4 = reinterpret_cast<floatx4*>(P.occ_dst + (size_t)particle * P.plane_stride);
    const uint2* __restrict__ sa = EXACT ? reinterpret_cast<const uint2*>(reinterpret_cast<const float*>(s4) + P.plane_px) : nullptr;   // the slots' ages,
    uint2* __restrict__ da = EXACT ? reinterpret_cast<uint2*>(reinterpret_cast<float*>(d4) + P.plane_px) : nullptr;                     // indexed like their float4s
    const unsigned amax = (unsigned)P.age_max;
    // one cell of the stamped form: advance the ages, store, report whether the child's window must cover it
#define RBS_EXACT_CELL(vv, aa, bnv, bnaa, didx)                                                          \
    ([&]() -> bool {                                                                                    \
        const uint2 ag_ = make_uint2(age_add2((aa).x, P.elapsed2), age_add2((aa).y, P.elapsed2));       \
        __builtin_nontemporal_store(vv, &d4[didx]);                                                     \
        da[didx] = ag_;                                                                                 \
        const unsigned a0_ = ag_.x & 0xffffu, a1_ = ag_.x >> 16, a2_ = ag_.y & 0xffffu, a3_ = ag_.y >> 16; \
        if (!STP) return a0_ <= amax || a1_ <= amax || a2_ <= amax || a3_ <= amax;                      \
        return exact_differs((vv).x, a0_, (bnv).x, (bnaa).x & 0xffffu, amax) || exact_differs((vv).y, a1_, (bnv).y, (bnaa).x >> 16, amax) || \
               exact_differs((vv).z, a2_, (bnv).z, (bnaa).y & 0xffffu, amax) || exact_differs((vv).w, a3_, (bnv).w, (bnaa).y >> 16, amax);   \
    }())
    // float4 index of pixel (col, row) in the parent's / the child's plane (whole planes: row W4 + col/4;
    // slabs: relative to the plane's stored region -- the child's region is u itself)
    const PlaneRef sref = SLAB ? parent_ref(P, parent) : PlaneRef{0, 0, P.cols};
    const int ss4 = SLAB ? sref.stride >> 2 : W4, sx4 = SLAB ? sref.x0 >> 2 : 0, sy0 = SLAB ? sref.y0 : 0;
    const int ds4 = SLAB ? w4 : W4, dx4 = SLAB ? ux4 : 0, dy0 = SLAB ? u.y : 0;
    const float alpha = P.alpha, beta = P.beta, bg_new = P.bg_new;
    const int lane = (int)threadIdx.x;
    int bx0 = P.cols, by0 = P.rows, bx1 = 0, by1 = 0;
    if (STRIPS) {
        // One rectangle: the region this kernel writes is u minus the rectangle -- a band of rows above it, one below
        // and two strips beside it, a tenth of u on a moving object (92 x 70 px of window around 88 x 65 of rectangle) --
        // and ONLY those cells are enumerated (the generic loop below walks all of u and skips nine lanes in ten: every
        // instruction of it is issued on SIMDs the persistent raster blocks are using).  Cell = one float4.
        const bool has = q.z > q.x;
        const int ty1 = has ? q.y : u.w, tby0 = has ? q.w : u.w;
        const int left4 = has ? (q.x - u.x) >> 2 : 0, right4 = has ? (u.z - q.z) >> 2 : 0, m = left4 + right4;
        const int rjump = has ? (q.z - u.x) >> 2 : 0;
        const Strips* __restrict__ SL = STRIPS == 2 ? P.strips + particle : nullptr;
        const int ns = STRIPS == 2 ? SL->n : 0;
        int sc = 0;                                      // (STRIPS == 2: this lane's cursor into the strip list: its cells only go forward)
        const int n_top = (ty1 - u.y) * w4, n_mid = (tby0 - ty1) * m, L = STRIPS == 2 ? SL->first[ns] : n_top + n_mid + (u.w - tby0) * w4;
        const int per = (L + (int)gridDim.x - 1) / (int)gridDim.x;
        const int lo = (int)blockIdx.x * per, hi = min(L, lo + per);
        for (int base = lo; base < hi; base += 64 * kU) {
            floatx4 v[kU], bn[kU];
            uint2 va[EXACT ? kU : 1], bna_[EXACT && STP ? kU : 1];
            int pk[kU];    // state << 28 | row << 14 | float4 column (rows and columns <= 8 192: create refuses more)
#pragma unroll
            for (int k = 0; k < kU; ++k) {
                const int idx = base + k * 64 + lane;
                const bool live = idx < hi;
                int row, c4;
                if (STRIPS == 2) {
                    row = u.y; c4 = 0;
                    if (live) {
                        while (idx >= SL->first[sc + 1]) ++sc;
                        const ushort4 B = SL->box[sc];
                        const int j = idx - SL->first[sc], sw = (int)B.y - (int)B.x, r = j / sw;
                        row = (int)B.z + r; c4 = (int)B.x + (j - r * sw) - ux4;
                    }
                } else
                if (idx < n_top) { const int r = idx / w4; row = u.y + r; c4 = idx - r * w4; }
                else if (idx < n_top + n_mid) {
                    const int j = idx - n_top, r = j / max(m, 1), kk = j - r * m;
                    row = ty1 + r; c4 = kk < left4 ? kk : kk - left4 + rjump;
                } else { const int j = idx - n_top - n_mid, r = j / w4; row = tby0 + r; c4 = j - r * w4; }
                const int col = (ux4 + c4) << 2;
                const bool stored = live && col >= pw.x && col < pw.z && row >= pw.y && row < pw.w;
                pk[k] = ((live ? (stored ? 2 : 1) : 0) << 28) | (row << 14) | (ux4 + c4);
                if (EXACT) va[EXACT ? k : 0] = make_uint2(kAgeBg2, kAgeBg2);
                if (stored) {
                    v[k] = __builtin_nontemporal_load(&s4[(row - sy0) * ss4 + (ux4 + c4 - sx4)]);
                    if (EXACT) va[EXACT ? k : 0] = sa[(row - sy0) * ss4 + (ux4 + c4 - sx4)];
                } else if (STP && live) {
                    v[k] = bc4[row * W4 + (ux4 + c4)];
                    if (EXACT) va[EXACT ? k : 0] = bca[row * W4 + (ux4 + c4)];
                }
                if (STP && live) {
                    bn[k] = bn4[row * W4 + (ux4 + c4)];
                    if (EXACT) bna_[EXACT && STP ? k : 0] = bna[row * W4 + (ux4 + c4)];
                }
            }
#pragma unroll
            for (int k = 0; k < kU; ++k) {
                const int st_ = pk[k] >> 28, row_ = (pk[k] >> 14) & 0x3fff, at_ = pk[k] & 0x3fff;
                if (!st_) continue;
                if (EXACT) {
                    if (st_ != 2 && !STP) v[k] = floatx4{bg_new, bg_new, bg_new, bg_new};
                    if (RBS_EXACT_CELL(v[k], va[EXACT ? k : 0], bn[k], bna_[EXACT && STP ? k : 0], (row_ - dy0) * ds4 + (at_ - dx4))) {
                        const int col = at_ << 2;
                        bx0 = min(bx0, col); bx1 = max(bx1, col + 4);
                        by0 = min(by0, row_); by1 = max(by1, row_ + 1);
                    }
                    continue;
                }
                floatx4 w;
                if (st_ == 2 || STP) {
                    w.x = occ_step(alpha, beta, v[k].x, bg_new);
                    w.y = occ_step(alpha, beta, v[k].y, bg_new);
                    w.z = occ_step(alpha, beta, v[k].z, bg_new);
                    w.w = occ_step(alpha, beta, v[k].w, bg_new);
                } else {
                    w.x = w.y = w.z = w.w = bg_new;
                }
                __builtin_nontemporal_store(w, &d4[(row_ - dy0) * ds4 + (at_ - dx4)]);
                if (STP ? (w.x != bn[k].x || w.y != bn[k].y || w.z != bn[k].z || w.w != bn[k].w)
                        : (w.x != bg_new || w.y != bg_new || w.z != bg_new || w.w != bg_new)) {
                    const int col = at_ << 2;
                    bx0 = min(bx0, col); bx1 = max(bx1, col + 4);
                    by0 = min(by0, row_); by1 = max(by1, row_ + 1);
                }
            }
        }
    } else {
    const int qstep = 64 / w4, rstep = 64 - qstep * w4;
    int row = ry0 + lane / w4;
    int c4 = lane - (lane / w4) * w4;
    for (int base = 0; base < n4; base += 64 * kU) {
        floatx4 v[kU], bn[kU];
        uint2 va[EXACT ? kU : 1], bna_[EXACT && STP ? kU : 1];
        int st[kU], at[kU], rr[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            const int idx = base + k * 64 + lane;
            const int col = (ux4 + c4) << 2;
            const bool live = idx < n4 && !raster_writes(P, particle, q, row, col);
            const bool stored = live && col >= pw.x && col < pw.z && row >= pw.y && row < pw.w;
            st[k] = live ? (stored ? 2 : 1) : 0;
            at[k] = ux4 + c4;
            rr[k] = row;
            if (EXACT) va[EXACT ? k : 0] = make_uint2(kAgeBg2, kAgeBg2);
            if (stored) {
                v[k] = __builtin_nontemporal_load(&s4[(row - sy0) * ss4 + (at[k] - sx4)]);
                if (EXACT) va[EXACT ? k : 0] = sa[(row - sy0) * ss4 + (at[k] - sx4)];
            } else if (STP && live) {
                v[k] = bc4[row * W4 + at[k]];
                if (EXACT) va[EXACT ? k : 0] = bca[row * W4 + at[k]];
            }
            if (STP && live) {
                bn[k] = bn4[row * W4 + at[k]];
                if (EXACT) bna_[EXACT && STP ? k : 0] = bna[row * W4 + at[k]];
            }
            c4 += rstep; row += qstep;
            if (c4 >= w4) { c4 -= w4; ++row; }
        }
#pragma unroll
        for (int k = 0; k < kU; ++k) {
            if (!st[k]) continue;
            if (EXACT) {
                if (st[k] != 2 && !STP) v[k] = floatx4{bg_new, bg_new, bg_new, bg_new};
                if (RBS_EXACT_CELL(v[k], va[EXACT ? k : 0], bn[k], bna_[EXACT && STP ? k : 0], (rr[k] - dy0) * ds4 + (at[k] - dx4))) {
                    const int col = at[k] << 2;
                    bx0 = min(bx0, col); bx1 = max(bx1, col + 4);
                    by0 = min(by0, rr[k]); by1 = max(by1, rr[k] + 1);
                }
                continue;
            }
            floatx4 w;
            if (st[k] == 2 || STP) {
                w.x = occ_step(alpha, beta, v[k].x, bg_new);
                w.y = occ_step(alpha, beta, v[k].y, bg_new);
                w.z = occ_step(alpha, beta, v[k].z, bg_new);
                w.w = occ_step(alpha, beta, v[k].w, bg_new);
            } else {
                w.x = w.y = w.z = w.w = bg_new;
            }
            __builtin_nontemporal_store(w, &d4[(rr[k] - dy0) * ds4 + (at[k] - dx4)]);
            if (STP ? (w.x != bn[k].x || w.y != bn[k].y || w.z != bn[k].z || w.w != bn[k].w)
                    : (w.x != bg_new || w.y != bg_new || w.z != bg_new || w.w != bg_new)) {
                const int col = at[k] << 2;
                bx0 = min(bx0, col); bx1 = max(bx1, col + 4);
                by0 = min(by0, rr[k]); by1 = max(by1, rr[k] + 1);
            }
        }
    }
    }
#undef RBS_EXACT_CELL
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, off)); by0 = min(by0, __shfl_xor(by0, off));
        bx1 = max(bx1, __shfl_xor(bx1, off)); by1 = max(by1, __shfl_xor(by1, off));
    }
    if (lane == 0 && bx1 > bx0) {
        int* w = reinterpret_cast<int*>(&P.win_dst[particle]);
        atomicMin(w + 0, bx0); atomicMin(w + 1, by0);
        atomicMax(w + 2, bx1); atomicMax(w + 3, by1);
    }
}

// One process per GPU (rbs_ipc_attach): pull the windows of parents that live in OTHER ranks' handles into
// local slots, one window-sized read over xGMI per staged plane -- for parents that several of this rank's
// children share (a parent with one child here is read in place by that child's raster block instead).
// Entry i: dst[i] < 0 -> nothing; else the plane of global slot src[i] -> local slot dst[i] of the CURRENT
// buffer (what the next call reads its parents from).  blockIdx.x = entry, blockIdx.y = row chunk.
__global__ __launch_bounds__(256) void rbs_stage_kernel(const DevParams P, const int* __restrict__ src, const int* __restrict__ dst,
                                                       float* __restrict__ occ_cur, int4* __restrict__ win_cur, int4* __restrict__ reg_cur)
{
    const int e = (int)blockIdx.x;
    const int d = dst[e];
    if (d < 0 || d >= P.shard_cap) return;
    const int g = src[e];
    if ((unsigned)g >= (unsigned)P.slots) return;
    const int4 w = parent_window(P, g);
    const bool empty = w.z <= w.x || w.w <= w.y;
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        const int4 r = empty ? make_int4(P.cols, P.rows, 0, 0) : w;
        win_cur[d] = r;
        if (P.slab_px) reg_cur[d] = r;   // the staged slab stores exactly the window, packed
    }
    if (empty) return;
    const PlaneRef sref = parent_ref(P, g);
    const float* __restrict__ sp = parent_plane(P, g);
    float* __restrict__ dp = occ_cur + (size_t)d * P.plane_stride;
    const int ww = w.z - w.x, hh = w.w - w.y, w4 = ww >> 2;
    const int dstride = P.slab_px ? ww : P.cols, dx0 = P.slab_px ? w.x : 0, dy0 = P.slab_px ? w.y : 0;
    const int rows_per = (hh + (int)gridDim.y - 1) / (int)gridDim.y;
    const int y0 = w.y + (int)blockIdx.y * rows_per, y1 = min(w.w, y0 + rows_per);
    for (int k = threadIdx.x; k < (y1 - y0) * w4; k += 256) {   // float4 columns: windows move in multiples of four pixels
        const int r = k / w4, y = y0 + r, x = w.x + ((k - r * w4) << 2);
        const floatx4 v = *reinterpret_cast<const floatx4*>(sp + (size_t)(y - sref.y0) * sref.stride + (x - sref.x0));
        *reinterpret_cast<floatx4*>(dp + (size_t)(y - dy0) * dstride + (x - dx0)) = v;
        if (P.exact)   // stamped planes: the four ages of the cell travel with it
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dp + P.plane_px) + (size_t)(y - dy0) * dstride + (x - dx0)) =
                *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(sp + P.plane_px) + (size_t)(y - sref.y0) * sref.stride + (x - sref.x0));
    }
}

// The largest region an updating call with these poses and parents would store (bbox of the parent's window and the
// particle's rectangle: what prep_particles records in err[1]) WITHOUT running the call: the library sizes slabs
// it chose itself with it before the first asynchronous call (which could not be repaired afterwards).
__global__ __launch_bounds__(64 * kPrepPerBlock) void rbs_region_probe_kernel(const DevParams P, int* __restrict__ out_max)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = (int)blockIdx.x * kPrepPerBlock + w;
    if (i >= P.n) return;
    const Rect r = particle_rect(P, P.poses + (size_t)i * 12 * P.n_bodies);
    if (lane != 0) return;
    int4 u = r.x1 > r.x0 ? make_int4(r.x0, r.y0, r.x1, r.y1) : make_int4(P.cols, P.rows, 0, 0);
    const int parent = P.indices[i];
    if ((unsigned)parent < (unsigned)P.slots) {
        const int4 pw = parent_window(P, parent);
        u = make_int4(min(pw.x, u.x), min(pw.y, u.y), max(pw.z, u.z), max(pw.w, u.w));
    }
    if (u.z > u.x && u.w > u.y) atomicMax(out_max, (u.z - u.x) * (u.w - u.y));
}

// The shared background plane's step: bgp_dst = occ_step(bgp_src) -- the same float operations as on any stored value, so a
// pixel outside a parent's window and the same pixel of the shared plane stay bit-identical.  rebase >= 0: the plane is
// RE-BASED on the plane of that slot first (inside its window the slot's stored values, outside the old shared plane);
// rebase == -2: the new plane is the scalar background everywhere (the handle leaves the shared-plane representation).
__global__ void rbs_bgp_step_kernel(const float* __restrict__ bgp_src, float* __restrict__ bgp_dst, const float* __restrict__ occ_src,
                                    const int4* __restrict__ win_src, const int4* __restrict__ reg_src, int plane_stride, int rebase,
                                    int rows, int cols, float alpha, float beta, float bg_new)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    if (rebase == -2) { bgp_dst[i] = bg_new; return; }   // back to the scalar background: the plane that IS it
    float v = bgp_src[i];
    if (rebase >= 0) {
        const int4 w = win_src[rebase];
        const int y = i / cols, x = i - y * cols;
        if (x >= w.x && x < w.z && y >= w.y && y < w.w) {
            if (reg_src) {   // slabs: the slot stores its region row-major from its first float
                const int4 g = reg_src[rebase];
                v = occ_src[(size_t)rebase * plane_stride + (size_t)(y - g.y) * (g.z - g.x) + (x - g.x)];
            } else {
                v = occ_src[(size_t)rebase * plane_stride + i];
            }
        }
    }
    bgp_dst[i] = occ_step(alpha, beta, v, bg_new);
}

// ---- stamped planes (DevParams.exact): the shared plane's step, and the hooks' view of a slot ----
// The shared plane on stamped planes: values copied, ages advanced (rebase >= 0: re-based on that slot's plane first; -2: all
// background again).  bgp = [npx floats][npx ages].
__global__ void rbs_bgp_step_exact_kernel(const float* __restrict__ bgp_src, float* __restrict__ bgp_dst, const float* __restrict__ occ_src,
                                          const int4* __restrict__ win_src, const int4* __restrict__ reg_src, int plane_stride, int plane_px,
                                          int rebase, int rows, int cols, unsigned elapsed2, float bg_new)
{
    const int npx = rows * cols;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    unsigned short* __restrict__ adst = reinterpret_cast<unsigned short*>(bgp_dst + npx);
    if (rebase == -2) { bgp_dst[i] = bg_new; adst[i] = 0xffffu; return; }
    float v = bgp_src[i];
    unsigned a = reinterpret_cast<const unsigned short*>(bgp_src + npx)[i];
    if (rebase >= 0) {
        const int4 w = win_src[rebase];
        const int y = i / cols, x = i - y * cols;
        if (x >= w.x && x < w.z && y >= w.y && y < w.w) {
            const float* slot = occ_src + (size_t)rebase * plane_stride;
            size_t at = (size_t)i;
            if (reg_src) { const int4 g = reg_src[rebase]; at = (size_t)(y - g.y) * (g.z - g.x) + (x - g.x); }
            v = slot[at];
            a = reinterpret_cast<const unsigned short*>(slot + plane_px)[at];
        }
    }
    bgp_dst[i] = v;
    adst[i] = (unsigned short)min(a + (elapsed2 & 0xffffu), 0xffffu);
}
// A slot as the hooks hand it out: the EFFECTIVE occlusion of every pixel as of the slot's epoch (the last updating call),
// out[npx] floats -- inside the slot's window propagate(value, age dt) rounded to float (age 0: the value itself), the
// background level where the age is beyond age_max or outside the window (bgp: the shared plane's pixel there).
// reg == nullptr: the slot is a whole plane.
__global__ void rbs_expand_exact_kernel(const float* __restrict__ slot, int plane_px, const int4* __restrict__ reg, const int4* __restrict__ win,
                                        int rows, int cols, float bg, const float* __restrict__ bgp, DevParams P, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int y = i / cols, x = i - y * cols;
    float v = bg;
    unsigned a = 0xffffu;
    bool have = false;
    if (win) {
        const int4 w = *win;
        if (x >= w.x && x < w.z && y >= w.y && y < w.w) {
            size_t at = (size_t)i;
            if (reg) { const int4 r = *reg; at = (size_t)(y - r.y) * (r.z - r.x) + (x - r.x); }
            v = slot[at];
            a = reinterpret_cast<const unsigned short*>(slot + plane_px)[at];
            have = true;
        }
    }
    if (!have && bgp) { v = bgp[i]; a = reinterpret_cast<const unsigned short*>(bgp + rows * cols)[i]; }
    P.bg_new = bg;
    out[i] = exact_prior(P, v, (int)a);
}
// A whole plane of effective values handed in from outside -> a slot with stored region r (whole planes: r spans the frame and
// `stride` is cols; slabs: packed, stride = r's width): values as given with age 0 ("as of the epoch", the rule of
// oracle orc_set_occlusion), background where the value equals the scalar background level.
__global__ void rbs_pack_exact_kernel(const float* __restrict__ full, const float* __restrict__ bgref, float bg, int4 r, int cols,
                                      int x0, int y0, int stride, float* __restrict__ slot, int plane_px)
{
    const int w = r.z - r.x, n = w * (r.w - r.y);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ly = i / w, lx = i - ly * w;
    const size_t src = (size_t)(r.y + ly) * cols + r.x + lx;
    const size_t at = (size_t)(r.y + ly - y0) * stride + (r.x + lx - x0);
    const float v = full[src];
    slot[at] = v;
    // (inside a window an age beyond age_max means the SCALAR background, shared plane or not: a value that merely equals the shared
    // plane's is stored like any other -- bgref only decided the window's extent)
    (void)bgref;
    reinterpret_cast<unsigned short*>(slot + plane_px)[at] = v == bg ? 0xffffu : 0u;
}
// Bounding box (float4-aligned) of the shared plane's pixels whose age is within age_max (stamped planes).
__global__ void rbs_bbox_age_kernel(const float* __restrict__ bgp, int rows, int cols, unsigned age_max, int* __restrict__ out4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    if (reinterpret_cast<const unsigned short*>(bgp + rows * cols)[i] > age_max) return;
    const int y = i / cols, x = i - y * cols;
    atomicMin(out4 + 0, x & ~3); atomicMin(out4 + 1, y);
    atomicMax(out4 + 2, (x & ~3) + 4); atomicMax(out4 + 3, y + 1);
}

// Make one windowed plane dense in place: pixels outside its window become the background.
// (The caller then marks the window full with rbs_set_window_kernel.)
__global__ void rbs_materialize_kernel(float* __restrict__ plane, const int4* __restrict__ win,
                                       int rows, int cols, float bg, const float* __restrict__ bgp = nullptr)
{
    const int4 w = *win;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int y = i / cols, x = i - y * cols;
    if (!(x >= w.x && x < w.z && y >= w.y && y < w.w)) plane[i] = bgp ? bgp[i] : bg;   // (bgp: the handle's shared background plane)
}

// Slabs: a slot's slab -> a whole plane (the background outside its window), a whole plane -> a
// slab with stored region r, and the float4-aligned bounding box of the values of a whole plane
// that differ from the background (out4 starts as (cols, rows, 0, 0)).
__global__ void rbs_expand_kernel(const float* __restrict__ slab, const int4* __restrict__ reg,
                                  const int4* __restrict__ win, int rows, int cols, float bg, float* __restrict__ out,
                                  const float* __restrict__ bgp = nullptr)
{
    const int4 r = *reg, w = *win;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int y = i / cols, x = i - y * cols;
    const bool in = x >= w.x && x < w.z && y >= w.y && y < w.w;
    out[i] = in ? slab[(size_t)(y - r.y) * (r.z - r.x) + (x - r.x)] : (bgp ? bgp[i] : bg);   // (bgp: the handle's shared background plane)
}
__global__ void rbs_pack_kernel(const float* __restrict__ full, int4 r, int cols, float* __restrict__ slab)
{
    const int w = r.z - r.x, n = w * (r.w - r.y);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ly = i / w, lx = i - ly * w;
    slab[i] = full[(size_t)(r.y + ly) * cols + r.x + lx];
}
__global__ void rbs_bbox_kernel(const float* __restrict__ full, int rows, int cols, float bg, int* __restrict__ out4,
                                const float* __restrict__ bgp = nullptr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    if (full[i] == (bgp ? bgp[i] : bg)) return;
    const int y = i / cols, x = i - y * cols;
    atomicMin(out4 + 0, x & ~3); atomicMin(out4 + 1, y);
    atomicMax(out4 + 2, (x & ~3) + 4); atomicMax(out4 + 3, y + 1);
}

__global__ void rbs_set_window_kernel(int4* __restrict__ win, int n, int4 value)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) win[i] = value;
}

// Inspection hook: depth image of one pose through the same raster_window path.
__global__ __launch_bounds__(kBlock) void rbs_render_kernel(const DevParams P, float* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Smem m = carve(smem, P.tile_px, false);
    const Rect r = particle_rect(P, P.poses);
    if (r.x1 <= r.x0) return;
    const TileGrid tg = tile_grid(r.x1 - r.x0, r.y1 - r.y0, P.tile_w, min(P.tile_w * P.tile_h, P.tile_px));
    for (int wy0 = r.y0; wy0 < r.y1; wy0 += tg.th)
        for (int wx0 = r.x0; wx0 < r.x1; wx0 += tg.tw) {
            const int wx1 = min(r.x1, wx0 + tg.tw), wy1 = min(r.y1, wy0 + tg.th);
            const int tw = wx1 - wx0, npx = tw * (wy1 - wy0);
            raster_window(P, P.poses, wx0, wy0, wx1, wy1, true, m.tile, m.big, m.nbig,
                          m.evalq + (threadIdx.x >> 6) * kQPlanes * kEvalQueue, 0xffffffffu);
            for (int p = threadIdx.x; p < npx; p += kBlock) {
                const int lr = p / tw;
                out[(wy0 + lr) * P.cols + wx0 + (p - lr * tw)] = __uint_as_float(m.tile[p]);
            }
            __syncthreads();
        }
}

__global__ void rbs_fill_kernel(float* __restrict__ p, size_t n, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

constexpr size_t smem_bytes(int tile_px, bool math_tables, bool many, bool exact)
{
    return sizeof(unsigned) * (size_t)tile_px + sizeof(int) * kBigCap + sizeof(double) * (kBlock / 64) + 16 + (many ? 16 * kCullSteps : 0) +
           sizeof(int) * kQPlanes * (kBlock / 64) * kEvalQueue + (math_tables ? sizeof(double) * kMathTabDoubles : 0) +
           (exact ? sizeof(int) * (kBlock / 64) * kEvalQueue + 16 * kLdsAges : 0);   // (the stamped planes' age ring and table head sit behind everything else)
}
constexpr int kExactRingPx = (kBlock / 64) * kEvalQueue + 4 * kLdsAges;   // the age ring's size in tile pixels: the stamped kernels' tile is that much smaller
constexpr int kTilePxExact = kTilePxF64 - kExactRingPx, kTilePxBigExact = kTilePxBigF64 - kExactRingPx;
static_assert(smem_bytes(kTilePxF64, true) <= smem_bytes(kTilePx, false) && smem_bytes(kTilePxBigF64, true) <= smem_bytes(kTilePxBig, false),
              "the F64 kernel's LDS block must not be larger than the F32 kernel's: the same number of blocks per CU");
static_assert(3 * ((smem_bytes(kTilePx, false, true) + 1279) / 1280 * 1280) <= 160 * 1024, "three raster blocks per CU: LDS is granted in 1 280-byte steps");
}  // namespace rbs
