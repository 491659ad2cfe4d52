/*
 * rbsensor_mi355x.h -- C-ABI of librbsensor_mi355x.so, the MI355X (gfx950) implementation of
 * dbot's Rao-Blackwellised depth-image observation model (RbSensor), i.e. the likelihood
 * evaluator that dbot_ros's particle tracker drives once per sampling block per frame.
 *
 * Drop-in boundary: each entry point replaces one virtual of the sensor object that
 * dbot::RbSensorBuilder<State>::build() returns.  The reference only names that object
 * through its builder, so the citations are the reference's own construction / call sites
 * (R: = bayesian-object-tracking/dbot_ros):
 *
 *   rbs_create            <- RbSensorBuilder<State>(object_model, camera_data, params)
 *                            R:source/dbot_ros/tracker/particle_tracker_node.cpp:164-203
 *                            R:source/dbot_ros/tracker/object_tracker_service_node.cpp:136-175
 *   rbs_reset             <- RbSensor::reset(), via tracker->initialize(...)
 *                            R:source/dbot_ros/tracker/particle_tracker_node.cpp:252
 *   rbs_set_observation   <- RbSensor::set_observation(image), via tracker_->track(image)
 *                            R:source/dbot_ros/object_tracker_ros.hpp:44-49
 *                            (layout R:source/dbot_ros/util/ros_interface.h:152-168)
 *   rbs_loglikes          <- RbSensor::loglikes(deltas, indices, update), once per sampling
 *                            block inside tracker_->track R:source/dbot_ros/object_tracker_ros.hpp:49
 *   rbs_destroy           <- ~RbSensor (tracker torn down per service session,
 *                            R:source/dbot_ros/tracker/object_tracker_service_node.cpp:233-238)
 *
 * Conventions: 0 = success, negative = error (message via rbs_last_error); the caller owns
 * every pointer it passes, the library copies what it keeps; a handle is driven by one
 * thread at a time but may be created/destroyed repeatedly from any thread; no exception
 * crosses this boundary.  There is NO CPU fallback: without a usable gfx950 device
 * rbs_create fails with RBS_ERR_NO_DEVICE.
 */
#ifndef RBSENSOR_MI355X_H
#define RBSENSOR_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBS_ABI_VERSION 2

enum {
    RBS_OK = 0,
    RBS_ERR_INVALID_ARGUMENT = -1,
    RBS_ERR_NO_DEVICE = -2,
    RBS_ERR_OUT_OF_MEMORY = -3,
    RBS_ERR_HIP = -4,
    RBS_ERR_UNSUPPORTED = -5
};

/* rbs_config.likelihood_precision: the arithmetic of the per-pixel Kinect likelihood (exp / erf /
 * log and the mixture algebra).  Coverage and depth do not depend on it (binary64 geometry, float
 * depth), nor does the occlusion process' time step; the per-particle sum is binary64 in both.
 *   F64  (the library default) binary64 with the reference CPU path's float rounding points
 *        (SURVEY A.4): agrees with the device-rule (EAGER) oracle to ~1e-15 -- its resampling
 *        (parent) indices are reproduced one for one at 20 000 particles -- and with the
 *        reference-semantics (LAZY: per-pixel time stamps, double propagation) oracle to ~1e-8 at
 *        640x480 and <= 5e-7 at 80x60, relative to max(1, |ll|): inside BASELINE.json north_star's
 *        1e-5 for EVERY particle of C1 on every frame of a 30-frame resampled sequence.  Parent
 *        indices against the LAZY oracle, same uniforms and history (measured,
 *        tests/test_gpu_reference_semantics.py): 0 of 60 000 children over 30 resamplings at 2 000
 *        particles / 640x480; 16 of 600 000 children over 30 resamplings at 20 000 particles /
 *        80x60 (0-5 per resampling), every one of them a uniform within 1e-7 of a step of the
 *        cumulative weights that drew the neighbouring parent.  Stored planes differ from the LAZY
 *        model's "as of now" values by <= 1.2e-6 after 30 frames (the 2^-18 snap bounds it).
 *   F32  OPT-IN.  float32 likelihood on the exp2 / log2 / rcp units, per-pixel terms derived from
 *        the observation on the fly.  It does NOT meet north_star's bar for every particle and does
 *        NOT reproduce parent indices at large particle counts: against the reference-semantics
 *        oracle it is within 1e-5 relative only for particles whose sum is well conditioned
 *        (|ll| >= 0.1 of the sum of the magnitudes of the per-pixel terms); for all particles the
 *        bound is 1e-6 of that sum of magnitudes, i.e. float32-level agreement, ~5e-4 absolute on
 *        log-likelihoods of magnitude 1e4.  The occlusion POSTERIOR (carried state) is computed in
 *        float32 as well: stored planes differ from the oracle's by up to 2e-6.
 * DEFAULT = RBS_PRECISION_LIBRARY_DEFAULT.  (Tooling only: RBS_PRECISION=f64|f32 in the environment
 * replaces DEFAULT; it never overrides a precision the caller named.) */
enum { RBS_PRECISION_DEFAULT = 0, RBS_PRECISION_F64 = 1, RBS_PRECISION_F32 = 2 };
#define RBS_PRECISION_LIBRARY_DEFAULT RBS_PRECISION_F64
/* rbs_config.state_layout: how occlusion planes are stored ("occlusion state layout" below).
 * DEFAULT = windowed (RBS_STATE=dense in the environment overrides DEFAULT only: tooling). */
enum { RBS_STATE_DEFAULT = 0, RBS_STATE_WINDOWED = 1, RBS_STATE_DENSE = 2 };
/* rbs_config.state_slab_px: see the field. */
#define RBS_SLAB_WHOLE_PLANES (-1)
/* rbs_config.occlusion_mode: how the occlusion process (OcclusionModel propagate, constants at
 * R:source/dbot_ros/tracker/particle_tracker_node.cpp:176-189) is carried between frames.
 *   REFERENCE   the reference CPU model's own bookkeeping: a pixel keeps the float posterior it was last UPDATED to and the
 *               time of that update, and its prior at a later call is propagate(value, elapsed time) evaluated in binary64
 *               and rounded once -- the oracle's mode LAZY, whose operations the device performs in the oracle's order, so
 *               priors are bit-identical and log-likelihoods agree to the last digits of the transcendentals (~1e-13
 *               relative; tests/test_gpu_exact_occlusion.py).  Storage: per pixel of a window the float and a 16-bit AGE
 *               (frames since the update, counted against the slot's epoch = the last updating call): 6 bytes instead of
 *               4; nothing is stepped between frames -- the copy kernel copies, ages advance by packed saturating adds.
 *               A pixel whose age exceeds the frame count at which (p_oo - p_ov)^(age * delta_time) <= 2^-40 (1 628
 *               frames with the defaults) is the background again: its value differs from a never-covered pixel's by
 *               less than that.  That count must fit the age: parameters with which a value has not decayed so far
 *               within 65 534 frames (p_oo - p_ov above ~0.987 at 30 frames/s) are RBS_ERR_UNSUPPORTED in this mode.
 *               Binary64 likelihood and the windowed layout only (RBS_ERR_UNSUPPORTED otherwise); every layout option
 *               of the windowed form works (slabs, several devices, attached ranks, the shared trail).
 *   DEVICE_RULE the float-stepped rule (oracle mode EAGER): every stored value advanced by one float FMA per updating
 *               call and snapped onto the background within 2^-18 of it.  Within ~1e-8 of REFERENCE (relative, typical);
 *               of 30 720 particle-frames one was 1.3e-5 away, and 16 of 600 000 resampled children drew a neighbouring
 *               parent at 20 000 particles (both gone with REFERENCE).  All precisions and layouts.
 * The hooks below that hand out or take whole planes (rbs_get/set_occlusion, rbs_export/import_plane/_window) deal in
 * EFFECTIVE values as of the last updating call in both modes; in REFERENCE mode a plane handed in is taken "as of now"
 * (ages 0), the rule of the oracle's set_occlusion.  rbs_occlusion_*device_ptr are RBS_ERR_UNSUPPORTED in REFERENCE mode
 * (a slot is not a float plane).  DEFAULT = RBS_OCC_LIBRARY_DEFAULT. */
enum { RBS_OCC_DEFAULT = 0, RBS_OCC_DEVICE_RULE = 1, RBS_OCC_REFERENCE = 2 };
#define RBS_OCC_LIBRARY_DEFAULT RBS_OCC_DEVICE_RULE

typedef struct rbs_handle rbs_handle;

/* kinect.tail_weight = 0 is accepted, as the reference accepts it (it passes the ROS parameter through
 * unchecked, R:source/dbot_ros/tracker/particle_tracker_node.cpp:183-184), but evaluated as this floor: the
 * library's binary64 erfc / exp are accurate to 1e-16 ABSOLUTE, which is exact next to tw / max_depth for any
 * tw >= 1e-9 and would not be in a mixture without a tail.  Observable difference: pixels the exact tw = 0
 * model prices at log(0) = -inf contribute log(1.7e-10 / p_bg) ~ -22 each instead. */
#define RBS_TAIL_WEIGHT_FLOOR 1e-9

typedef struct rbs_config {
    int32_t abi_version;            /* RBS_ABI_VERSION                                        */
    int32_t device_id;              /* HIP device ordinal                                      */
    int32_t rows, cols;             /* evaluated resolution (after down-sampling)             */
    double K[9];                    /* row-major 3x3 intrinsics, top two rows already divided
                                       by the down-sampling factor
                                       (R:source/dbot_ros/util/ros_camera_data_provider.cpp:72);
                                       zero skew, K[8] == 1                                    */
    int32_t max_particles;          /* occlusion slots to allocate (>= any n passed later)    */
    int32_t n_objects;              /* rigid bodies = meshes (object/meshes)                   */
    const double* vertices;         /* concatenated xyz per object, sum(vertex_counts)*3      */
    const int32_t* vertex_counts;   /* [n_objects]                                             */
    const int32_t* triangles;       /* concatenated vertex-index triples, local to each object*/
    const int32_t* triangle_counts; /* [n_objects]                                             */
    /* RbSensorBuilder<State>::Parameters, R:...particle_tracker_node.cpp:176-189 */
    double p_occluded_visible;
    double p_occluded_occluded;
    double initial_occlusion_prob;
    double tail_weight;             /* 0 <= tw < 1; below RBS_TAIL_WEIGHT_FLOOR it is evaluated AS the floor (see above) */
    double model_sigma;
    double sigma_factor;
    double delta_time;
    /* --- ABI version 2 --- */
    int32_t likelihood_precision;   /* RBS_PRECISION_*                                         */
    int32_t state_layout;           /* RBS_STATE_*                                             */
    /* Particle sharding over several devices of one node inside ONE handle (the reference's
     * tracker node is one process, R:source/dbot_ros/tracker/particle_tracker_node.cpp:277-284):
     * n_devices <= 1 -> device_id alone.  n_devices > 1 -> device_ids[n_devices] HIP ordinals;
     * max_particles is the TOTAL over all devices.  See "several devices" below. */
    int32_t n_devices;
    const int32_t* device_ids;
    /* Window-sized slabs for the occlusion state (windowed layout only).  A slot then holds
     * state_slab_px floats instead of a whole plane (rows*cols floats: 2 x 1.2 MB per particle at
     * 640x480, of which a tracked object's window touches ~3 %) and stores only the region an updating
     * call writes, bbox(parent's window, the particle's screen rectangle).  The numbers are those of
     * whole planes.
     *   0   the library's choice: whole planes up to 8 192 particles (per device), slabs of
     *       rows*cols/8 floats above (8x the particles in the same memory);
     *   >0  slabs of that many floats;   RBS_SLAB_WHOLE_PLANES (-1)  whole planes whatever the count.
     * Slabs GROW: at every synchronising call the library enlarges them once the largest region any
     * particle has asked for fills three quarters of a slab, and a synchronous rbs_loglikes whose
     * region does not fit all the same is taken back, the slabs enlarged, and run again -- the caller
     * sees the numbers of whole planes (state memory grows towards that of whole planes in the
     * worst case; enlarging needs the new buffers beside the old ones for a moment and fails with
     * RBS_ERR_OUT_OF_MEMORY if they do not fit).  Only a call that has already RETURNED cannot be
     * repaired -- rbs_loglikes_device, a frame of rbs_tracker_*: a region that outgrows its slab by
     * more than a quarter within one such call is contained (log-likelihood NaN, plane reset to the
     * background) and reported once, by the next synchronising call / that frame's result; the slabs
     * have been enlarged when that call returns.  rbs_tracker_initialize sizes the slabs for the object
     * at its default pose with a probe call, so a tracker does not start on slabs that are too small;
     * slabs the LIBRARY chose (0) are checked the same way in front of a handle's first updating
     * rbs_loglikes_device (one small kernel, one host synchronisation, once).  Planes handed in from
     * outside (rbs_set_occlusion, rbs_import_plane / _window) enlarge the slabs when they do not fit.
     * With slabs -- chosen by the caller or by the library -- two hooks are RBS_ERR_UNSUPPORTED, because a slot
     * is not a plane any more: rbs_occlusion_device_ptr and rbs_occlusion_next_device_ptr (use
     * rbs_export_plane / rbs_export_window); RBS_SLAB_WHOLE_PLANES keeps them.  The shards of a handle over
     * several devices and ranks attached with rbs_ipc_attach keep ONE slab size among them: a group grows all its
     * shards together (synchronous calls), attached ranks do not grow. */
    int32_t state_slab_px;
    int32_t occlusion_mode;         /* RBS_OCC_* (this field was reserved0 = 0 = DEFAULT before round 6: same layout, same ABI version) */
} rbs_config;

int32_t rbs_abi_version(void);
int32_t rbs_device_count(void);

int32_t rbs_create(const rbs_config* cfg, rbs_handle** out);
void rbs_destroy(rbs_handle* h);
/* Message of the last error on this handle; h == NULL gives the calling thread's last
 * rbs_create failure. Never NULL. */
const char* rbs_last_error(const rbs_handle* h);

int32_t rbs_reset(rbs_handle* h);

/* depth[n], n == rows*cols, row-major (row*cols+col), metres, NaN/inf = no reading. */
int32_t rbs_set_observation(rbs_handle* h, const double* depth, size_t n);
/* Same, from the float32 pixels the camera driver delivers (skips the double round trip). */
int32_t rbs_set_observation_f32(rbs_handle* h, const float* depth, size_t n);
/* rbs_set_observation WITHOUT the copy at call time (round 5): the library BORROWS `depth` -- it must stay valid and
 * unchanged until the next rbs_loglikes / rbs_loglikes_deltas / rbs_loglikes_prefetch or rbs_synchronize on the handle has
 * RETURNED (any other entry point that needs the observation stages it first as well; a later rbs_set_observation* or
 * rbs_reset simply drops it).  That next likelihood call launches its rectangles kernel and its GEOMETRY kernel -- which
 * need no frame -- first, converts (double -> float) and sends the frame while they run, and only then launches the
 * likelihood kernel: the two-kernel form of the raster launch (same work items, same summation order, same bits as the
 * one-kernel launch), so the frame's journey, 60-100 us of a synchronous 300 us step, hides behind the geometry.
 * This is what the dbot binding calls: inside tracker_->track(image) (R:source/dbot_ros/object_tracker_ros.hpp:49) the
 * image outlives the filter's set_observation / loglikes pair.  The occlusion clock advances at this call, as for
 * rbs_set_observation.  Handles over several devices, precision F32 and whole-plane handles copy at once (= rbs_set_observation).
 * "Returned" includes returned with an error: a likelihood call that is refused (bad indices, ...) or fails before it staged
 * the frame copies it on its way out -- the frame is the observation from then on, the buffer the caller's again.
 * MEMORY: the two-kernel launch hands its depth tiles from the first kernel to the second through device memory, one tile (44 KB)
 * per work item, allocated at the first such call for the worst case the call's particle count allows (every rectangle the whole
 * frame: 36 tiles per particle at 640x480 -- 2.7 GB at 2 000 particles; rbs_tracker_*_f64 below 5 000 evaluations likewise).  The
 * library takes it only if it is at most 8 GB AND at most half of the device's free memory; otherwise, or if the allocation fails,
 * the call is launched as one kernel with the frame staged first (= rbs_set_observation), same results. */
int32_t rbs_set_observation_borrowed(rbs_handle* h, const double* depth, size_t n);
int32_t rbs_set_observation_borrowed_f32(rbs_handle* h, const float* depth, size_t n);   /* the same for the driver's float pixels */

/* Frame ingest straight from the camera driver: `native` is the full-resolution
 * float32 image (width x height, metres, NaN = no reading); the evaluated image is its
 * sub-sampling by the reference's rule, eval(row, col) = native(row*f, col*f) with
 * rows = height/f, cols = width/f (ri::to_eigen_vector, R:source/dbot_ros/util/ros_interface.h:152-168),
 * applied while the frame is staged into pinned memory: ONE host pass over the rows*cols values the
 * sensor evaluates (19 KB of a 1.2 MB frame at the reference's default factor 8) and a transfer of
 * that size -- in place of the reference's three host copies of the whole frame
 * (R:source/dbot_ros/object_tracker_ros.hpp:82,119, ros_interface.h:156-165). */
int32_t rbs_set_observation_native_f32(rbs_handle* h, const float* native, int32_t width,
                                       int32_t height, int32_t downsampling_factor);
/* The same from DEVICE memory on the handle's device (float[rows*cols], evaluated resolution):
 * for frames that are already on the GPU (a driver / preprocessing stage there, or a caller
 * replaying a resident sequence).  Ordered on `stream` (NULL = the handle's own stream), no
 * host synchronisation.  The ingest kernel is launched by the next rbs_* call that needs the
 * frame (it shares a launch with the next rbs_loglikes_device on the same stream): the caller
 * keeps `d_depth` alive and unchanged until that call's work has run. */
int32_t rbs_set_observation_device(rbs_handle* h, const float* d_depth, void* stream);
/* Zero-copy hand-over of a host frame: rbs_acquire_frame_buffer gives the handle's pinned staging
 * buffer for the NEXT frame (float[rows*cols], evaluated resolution, valid until the matching
 * commit); the caller writes the frame straight into it -- the conversion loop of
 * ri::to_eigen_vector (R:source/dbot_ros/util/ros_interface.h:152-168) can target it -- and
 * rbs_commit_frame_buffer is then rbs_set_observation_f32 without the 1.2 MB host copy.  Two
 * buffers alternate: acquiring waits for the upload that used the buffer two frames ago. */
int32_t rbs_acquire_frame_buffer(rbs_handle* h, float** buf);
int32_t rbs_commit_frame_buffer(rbs_handle* h);
/* Current evaluated observation -> host float[rows*cols] (inspection). */
int32_t rbs_get_observation(rbs_handle* h, float* out);

/* poses:   [n][n_objects][12] doubles, R (row-major 3x3) then t: absolute camera-frame pose
 *          of each body (delta composed with the default pose by the caller).
 * indices: [n] in: occlusion slot each particle inherits from; out (update != 0): identity.
 * update:  non-zero -> write the posterior occlusion of particle i into slot i.
 * out_loglik: [n] doubles.  Host pointers; synchronous: the call returns when the log-likelihoods
 * are in out_loglik (the caller's buffers are staged through pinned memory that the kernels read
 * and write in place -- no copy-engine transfer besides the frame's); the occlusion planes of an
 * updating call are finished in the background and joined by the next call on the handle. */
int32_t rbs_loglikes(rbs_handle* h, const double* poses, int32_t* indices, int32_t n,
                     int32_t update, double* out_loglik);

/* rbs_loglikes with the NEXT frame travelling behind it (a recorded dataset, or a driver that is a frame ahead:
 * the reference's tracker takes its frames from a mailbox, R:source/dbot_ros/object_tracker_ros.hpp:69-84, :44-49).
 * Same call, same results; between launching its kernels and waiting for them the library stages next_depth
 * (float32 metres, rows*cols, the layout of rbs_set_observation_f32) into its other pinned image and sends it on
 * the upload stream, the per-pixel model terms behind it -- the 1.2 MB copy and transfer of a 640x480 frame pass
 * while the raster kernel runs.  The current observation is unchanged.  rbs_set_observation_prefetched then makes
 * that frame the observation at no cost (one frame per call, as rbs_set_observation_f32 does: the model clock
 * advances by delta_time); any other rbs_set_observation* in between abandons it.  One frame may be waiting at a time: a second
 * rbs_loglikes_prefetch with a next frame before the first has been installed (or abandoned) is RBS_ERR_INVALID_ARGUMENT -- a caller
 * with several sampling blocks per frame hands the next frame to ONE of them and calls rbs_loglikes for the others.
 * Single-device handles. */
int32_t rbs_loglikes_prefetch(rbs_handle* h, const double* poses, int32_t* indices, int32_t n, int32_t update,
                              double* out_loglik, const float* next_depth, size_t next_n);
int32_t rbs_set_observation_prefetched(rbs_handle* h);

/* Device-pointer variant: poses/indices/out_loglik are device memory on the handle's device,
 * work is enqueued on `stream` (a hipStream_t, NULL = the handle's own stream) and the call
 * returns without synchronising.  indices is read-only and read in `stream` order only (the
 * first kernel snapshots it for the library's second stream): the caller may overwrite it by work
 * enqueued on `stream` right after the call.  With update != 0 the caller must treat the slot
 * map as identity afterwards.  d_out_loglik is complete in `stream` order.  The
 * occlusion planes an updating call writes are finished by a second, internal stream: every
 * later rbs_* call on the handle orders itself after them (so back-to-back calls pipeline), and
 * rbs_synchronize / rbs_occlusion_*_device_ptr / any host-pointer entry point waits for them. */
int32_t rbs_loglikes_device(rbs_handle* h, const double* d_poses, const int32_t* d_indices,
                            int32_t n, int32_t update, double* d_out_loglik, void* stream);

/* RbSensor::loglikes(deltas, indices, update) WITH THE ARGUMENTS THE FILTER PASSES (round 5): the particles' state
 * DELTAS around the sensor's default ("integrated") poses, not absolute poses -- what dbot's filter hands the
 * sensor inside tracker_->track(image) (R:source/dbot_ros/object_tracker_ros.hpp:49; the State of
 * R:source/dbot_ros/object_tracker_ros.h:40-41, read per body through component(i) as at .hpp:54-60).  The
 * composition of SURVEY A.1,
 *     R = R(delta rotation vector) R(default rotation vector),   t = t(delta) + t(default),
 * rotation vector -> matrix through the unit quaternion, runs on the device inside the rectangles kernel (the wave that owns a
 * particle composes its poses first; the operations and their order are oracle/tracker_oracle.c's orc_compose_poses; sin / cos /
 * sqrt are the device library's, so a composed entry may differ from a host libm composition in its last bit:
 * rbs_get_poses returns exactly what was evaluated).  On the host it costs two sin / cos / sqrt and a 3x3 product
 * per particle and body -- 0.25 ms at 2 000 particles, more than the device step.
 *   deltas         [n][n_objects][body_stride] doubles; of each body's body_stride (>= 6) values the first six are
 *                  read: position (3), rotation ("Euler") vector (3).  body_stride = 12 takes an array of dbot
 *                  FreeFloatingRigidBodiesState storage as it is (position, rotation vector, velocities).
 *   default_poses  [n_objects][body_stride], same layout: the sensor's integrated_poses().
 * Everything else -- indices in / out, update, out_loglik, errors, slabs -- as rbs_loglikes.  Single-device handles
 * and handles over several devices. */
int32_t rbs_loglikes_deltas(rbs_handle* h, const double* deltas, const double* default_poses, int32_t body_stride,
                            int32_t* indices, int32_t n, int32_t update, double* out_loglik);
/* The pinned staging block rbs_loglikes_deltas reads the deltas from, for callers whose particles are separate vectors (dbot's
 * StateArray: one Eigen vector per particle) and who must gather them anyway: gather position + rotation vector of every body, six
 * doubles each, particle-major ([n][n_objects][6], n <= max_particles) straight into *buf and pass *buf itself as `deltas` with
 * body_stride = 6 -- rbs_loglikes_deltas then skips its own staging copy.  The block is the handle's; valid until rbs_destroy; its
 * contents are consumed by the call.  Single-device handles. */
int32_t rbs_deltas_buffer(rbs_handle* h, double** buf);
/* Test / inspection hook: the absolute poses ([n][n_objects][12] doubles: R row-major, t) the handle's LAST
 * host-pointer likelihood call evaluated -- for rbs_loglikes the caller's own, for rbs_loglikes_deltas the
 * device's compositions.  Synchronises.  Single-device handles. */
int32_t rbs_get_poses(rbs_handle* h, double* out, int32_t n);

/* Block until everything enqueued on the handle's own stream has finished. */
int32_t rbs_synchronize(rbs_handle* h);

/* --- several devices in one handle --------------------------------------------------------
 * rbs_config.n_devices > 1 (the reference's tracker node is ONE process,
 * R:source/dbot_ros/tracker/particle_tracker_node.cpp:277-284): the handle shards the particles over
 * device_ids[].  Occlusion slots are GLOBAL, 0 .. n_devices * cap - 1 with
 * cap = ceil(max_particles / n_devices); slot g lives on device_ids[g / cap].  rbs_loglikes
 * evaluates particle i on the device that owns slot i (an updating call writes slot i there);
 * indices[i] may name a plane on any device -- a remote parent's window is read in place over
 * xGMI (peer access), nothing migrates.  rbs_set_observation* upload the frame to every device.
 * rbs_tracker_* on such a handle runs the filter replicated on every device, shards the sensor
 * call by a parent-affine layout and exchanges the log-likelihoods with one RCCL all-gather per
 * sampling block (librccl is bound with dlopen when such a handle is created).  The zero-copy
 * entry points work on such a handle too: rbs_set_observation_device takes a frame in the memory of
 * the FIRST device (device_ids[0]) and every device ingests it from there over xGMI;
 * rbs_loglikes_device takes poses / parent slots / results as arrays on the first device in global
 * particle order -- device k pulls its slice [k cap, (k+1) cap) and stores its log-likelihoods in
 * place (peer access) -- with `stream` a stream of the first device (NULL = the library's): the
 * devices start after the work enqueued on it so far, and it is ordered after all of them, so
 * d_out_loglik is complete in `stream` order without a host synchronisation; slot-addressed hooks (rbs_get/set_occlusion, rbs_get_window,
 * rbs_export/import_plane, rbs_occlusion_*device_ptr) take global slots and act on the owning
 * device; rbs_render_depth, rbs_get_observation, the timing queries answer for device_ids[0].
 * An ordinal may repeat in device_ids (several shards on one GPU: functional tests). */

/* --- occlusion state layout -------------------------------------------------------------
 * A slot's plane is stored as a WINDOW (a pixel rectangle) plus the handle-wide background
 * level: outside its window a plane equals the value a never-covered pixel has reached
 * (initial_occlusion_prob stepped by the occlusion process at every updating call).  An
 * updating call writes only bbox(parent's window, the particle's screen rectangle) and
 * re-tightens the child's window to the values that still differ from the background; the
 * process snaps a value within 2^-18 of the background onto it, so a window follows the object
 * instead of growing for ever.  The numbers are those of whole planes (oracle mode EAGER);
 * only the bytes moved change.  rbs_config.state_layout = RBS_STATE_DENSE keeps whole
 * planes (every updating call then copies every plane in full).  Every entry point below
 * hands out / accepts whole planes in either layout. */
/* --- inspection hooks (tests, state migration between devices) --- */
/* Window (x0, y0, x1, y1) of a slot's current plane; (cols, rows, 0, 0) = empty (all
 * background); (0, 0, cols, rows) always with state_layout dense. */
int32_t rbs_get_window(rbs_handle* h, int32_t slot, int32_t out[4]);
/* Never-covered occlusion level of the current planes. */
/* Shared trail (round 5).  With a scalar background a pixel stays in a window until its value has relaxed to within 2^-18 of
 * that level, ~730 frames after the object left it, and every child inherits the whole trail from its parent: an object that
 * moves across the image leaves windows of most of the frame, stored, read and stepped once per particle per frame.  Once the
 * sampled window area exceeds a tenth of the frame, a handle on windowed planes (whole planes or slabs; either likelihood precision,
 * round 6) stores its planes against a handle-wide background PLANE instead: the plane is re-based on one particle's plane and every child is
 * re-measured against it, so that what the particles share from a common ancestor is stored once (after a resampling that is
 * nearly all of it).  Stored VALUES do not change by a bit -- the shared plane steps with the float operations of any stored
 * value -- and every entry point sees whole planes as before (rbs_get_occlusion, rbs_export_window: the slot is made dense first).
 * RBS_SHARED_TRAIL=0 in the environment disables it; handles over several devices and attached ranks: see rbs_shared_trail_rebase below.
 * Inspection: *active = the handle has switched, *rebases = how often it re-based so far. */
int32_t rbs_shared_trail_state(rbs_handle* h, int32_t* active, int32_t* rebases);
/* Round 6: the shared trail where OTHERS read a handle's planes in place.  A handle over several devices (rbs_config.n_devices) takes
 * ONE decision per call for all its shards -- every device keeps its own, identical copy of the shared plane, stepped alike and
 * re-based in the same call on the same global slot, whose plane the other devices read from its owner -- and needs nothing from the
 * caller.  Handles of one process per GPU (rbs_ipc_export / rbs_ipc_attach) cannot agree among themselves: the CALLER, who owns the
 * collective, tells every rank before the SAME updating call -- rbs_shared_trail_rebase(h, global_slot): at the next updating call the
 * handle enters the shared-trail representation if it has not yet and re-bases its shared plane on that global slot's plane (any
 * rank's; read in place); global_slot = -2: it returns to the scalar background at that call.  Same slot, same call, every rank --
 * dbot_ros_amd/dist.py PeerShardedStep does it (shared_trail=True: every 32 steps while any rank's windows exceed a tenth of the
 * frame, agreed by one all-reduce of a flag).  On a handle of its own the call forces a re-basing on one of its slots (tests).
 * Values are unchanged by a bit, as for a single handle; an exported handle never switches by itself. */
int32_t rbs_shared_trail_rebase(rbs_handle* h, int32_t global_slot);
/* Behaviour switches a caller may need, as an API (round 6; the environment variables of the same meaning are tooling and are read
 * once at rbs_create -- INTEGRATION.md section 6 lists them).  Values take effect from the next call on; results never depend on them
 * (they choose what is stored and how a call is launched).
 *   RBS_OPT_SHARED_TRAIL        0 / 1   may the handle store its planes against a shared background plane (default 1; a handle that
 *                                       has already switched stays so until rbs_reset)
 *   RBS_OPT_SHARED_TRAIL_ENTER  (0, 1]  sampled window fraction of the frame above which it switches / keeps re-basing (default 0.10)
 *   RBS_OPT_SHARED_TRAIL_EVERY  >= 1    updating calls between re-basings while windows stay large (default 32)
 *   RBS_OPT_TRACKER_SPLIT_MAX   >= 0    rbs_tracker_*: evaluations per frame up to which a frame handed over frame by frame travels
 *                                       behind the geometry kernel (two-kernel launch; default 5 000, 0 = never)
 *   RBS_OPT_TIMING_EVERY        >= 1    = rbs_set_timing_every
 * A handle over several devices applies them to all its shards. */
enum { RBS_OPT_SHARED_TRAIL = 1, RBS_OPT_SHARED_TRAIL_ENTER = 2, RBS_OPT_SHARED_TRAIL_EVERY = 3, RBS_OPT_TRACKER_SPLIT_MAX = 4, RBS_OPT_TIMING_EVERY = 5 };
int32_t rbs_set_option(rbs_handle* h, int32_t option, double value);
/* The window area the handle sampled last (mean over its particles of the region an updating call stored, as a fraction of the
 * frame; sampled every 8th updating call, read back without blocking): what the shared trail's policy looks at. */
int32_t rbs_window_fraction(rbs_handle* h, double* out);
int32_t rbs_get_background(rbs_handle* h, float* out);
/* Stored occlusion plane of a slot -> host float[rows*cols]. */
int32_t rbs_get_occlusion(rbs_handle* h, int32_t slot, float* out);
/* Overwrite a slot's plane from host float[rows*cols]. */
int32_t rbs_set_occlusion(rbs_handle* h, int32_t slot, const float* plane);
/* Device address of a slot's current plane, made whole first (valid until the next updating call). */
int32_t rbs_occlusion_device_ptr(rbs_handle* h, int32_t slot, void** out);
/* Device address of slot's plane in the buffer the NEXT updating call will write. */
int32_t rbs_occlusion_next_device_ptr(rbs_handle* h, int32_t slot, void** out);
/* Device-to-device copy of a slot's plane to / from caller-owned device memory (float[rows*cols]
 * on the handle's device), enqueued on `stream` (NULL = the handle's stream) after the planes of
 * the last updating call are complete: the migration path between GPUs (a torch.distributed /
 * RCCL send or recv of the caller's buffer), without staging through the host. */
int32_t rbs_export_plane(rbs_handle* h, int32_t slot, void* d_dst, void* stream);
int32_t rbs_import_plane(rbs_handle* h, int32_t slot, const void* d_src, void* stream);
/* The same transport, WINDOW-SIZED (round 4): a plane is its window plus the background level, so what
 * has to travel is the window's rectangle and its w x h values (about 3 % of a plane), not rows*cols
 * floats.  rbs_export_window: rect_out = (x0, y0, x1, y1) of the slot's window -- (cols, rows, 0, 0) when
 * the plane is all background -- and its values packed row-major, width x1 - x0, into d_payload
 * (device memory of the handle's device, capacity_floats floats; too small: RBS_ERR_INVALID_ARGUMENT
 * with nothing copied, rect_out still valid so the caller can size the buffer).  Waits for the
 * rectangle (one 16-byte read-back); the payload copy is enqueued on `stream`.
 * rbs_import_window: the slot's plane becomes "background everywhere, these values inside rect"
 * (x0 and x1 multiples of 4, inside the frame); a slab that is too small for rect is ENLARGED first, as
 * behind rbs_import_plane (one drain of the handle; every rank grows its slabs on its own schedule, so a
 * window from a rank that has grown must not be refused by one that has not) -- only the shards of a
 * multi-device handle and handles attached to other ranks (rbs_ipc_attach), whose slabs are addressed with
 * a fixed stride from outside, fail with RBS_ERR_OUT_OF_MEMORY.  Both layouts, whole planes and slabs.
 * Used by dbot_ros_amd/dist.py (one process per GPU) to migrate a parent's plane to the rank its
 * surplus children were placed on: an RCCL send / recv of rect + payload. */
int32_t rbs_export_window(rbs_handle* h, int32_t slot, int32_t rect_out[4], void* d_payload, size_t capacity_floats, void* stream);
int32_t rbs_import_window(rbs_handle* h, int32_t slot, const int32_t rect[4], const void* d_payload, void* stream);
/* `stream` waits (on the device, no host synchronisation) until the planes of the handle's last
 * updating call are complete, the side stream's copy kernel included: what a caller enqueues on
 * `stream` afterwards -- e.g. the collective that tells the other ranks "my step is done" -- is
 * ordered after them. */
int32_t rbs_stream_join(rbs_handle* h, void* stream);

/* --- particle sharding ACROSS PROCESSES: one process per GPU (torch.distributed / RCCL) ------------
 * SURVEY 8(e)'s partitioning with one handle per rank: rank r owns global slots
 * [r * max_particles, (r + 1) * max_particles).  After rbs_ipc_attach the parent indices of
 * rbs_loglikes / rbs_loglikes_device are GLOBAL slots: a parent that lives in another rank's handle
 * is read IN PLACE over xGMI -- its window's ~3 % of a plane, once -- through that rank's buffers
 * mapped into this process (hipIpcGetMemHandle / hipIpcOpenMemHandle), exactly as the shards of a
 * multi-device handle read each other (rbs_config.n_devices).  No plane ever migrates; the only
 * collective of a filter step is the all-gather of the log-likelihoods.
 *   rbs_ipc_export   this handle's description (RBS_IPC_BLOB_BYTES bytes): memory handles of its two
 *                    plane buffers, window and region tables, its device and geometry.  Single-device
 *                    handles only; synchronises.
 *   rbs_ipc_attach   blobs = world x RBS_IPC_BLOB_BYTES, rank-major, every rank's export (exchanged by
 *                    the caller: an all-gather of bytes); rank = this handle's.  All handles must have
 *                    the same geometry, layout, max_particles and slab size.  Afterwards children are
 *                    still written to LOCAL slots 0..n-1 (global rank * max_particles + i); an updating
 *                    rbs_loglikes returns indices[i] = rank * max_particles + i.
 * ORDERING is the caller's: every rank performs the same sequence of rbs_reset / updating calls, and a
 * rank may start step k+1 only when every rank has finished step k (it reads their planes and
 * overwrites the buffer they were reading) -- enqueue rbs_stream_join and then the step's all-gather
 * on the stream the next call is enqueued on (bench.py --gpus N, dbot_ros_amd/dist.py PeerShardedStep).
 * Slabs do not grow once attached (the other ranks address them with a fixed stride): a region that
 * does not fit is an error; size state_slab_px for the scene. */
#define RBS_IPC_BLOB_BYTES 512
int32_t rbs_ipc_export(rbs_handle* h, void* blob_out);
int32_t rbs_ipc_attach(rbs_handle* h, int32_t rank, int32_t world, const void* blobs);
/* Parents that SEVERAL of this rank's children share are better pulled once than read in place by every
 * child: for i in [0, n) with d_dst_local[i] >= 0, the plane of global slot d_src_global[i] (any rank's)
 * is copied -- its window only -- into local slot d_dst_local[i] of the current buffer, on `stream`, no
 * host synchronisation (entries with d_dst_local[i] < 0 are skipped, so the arrays can have a fixed
 * length).  The caller keeps local slots for this (max_particles > the particles it evaluates) and names
 * the staged copy -- global slot rank * max_particles + d_dst_local[i] -- as the children's parent.
 * Both arrays in device memory.  Works on an unattached handle as well (src = local slots). */
int32_t rbs_stage_windows(rbs_handle* h, const int32_t* d_src_global, const int32_t* d_dst_local, int32_t n, void* stream);
/* The resampling half of the filter step across processes in one call: three or four launches on `stream` (two or three over
 * the chip for the cdf and the children's parents, one block for the plan), no host synchronisation after the first call, which
 * allocates scratch: what dbot_ros_amd/dist.py global_resample + plan_shard
 * compute with ~45 tensor kernels.  d_loglik_all [n_total = world * n_local]: the all-gathered
 * log-likelihoods, rank-major.  d_uniforms_sorted [n_total]: the step's uniforms in [0, 1), identical on
 * every rank and ASCENDING -- multinomial resampling (parent = upper_bound(cumulative normalised
 * weights, u); weights exp((ll - max) / temperature), temperature 1 = the filter's own) draws exchangeable
 * children, and u -> parent is monotone, so sorted uniforms give the children sorted by parent: child g of
 * the job is evaluated by rank g / n_local, and in that order the children of rank r's particles are one
 * contiguous run that mostly coincides with rank r's slots.  A NaN log-likelihood (a contained particle)
 * weighs nothing.  Outputs, all in device memory, [n_local] each unless noted:
 *   d_parent_idx     the GLOBAL slot local child k inherits from -- owner * max_particles + local slot --
 *                    or, for a parent on another rank that at least min_share of this rank's children
 *                    share, the local staging slot rank * max_particles + n_local + j
 *   d_stage_src/dst  the arguments of rbs_stage_windows(h, src, dst, n_local, stream), which the caller
 *                    enqueues next: entry k pulls global slot src[k] into local slot dst[k] iff dst[k] >= 0
 *   d_parents_local  (may be null) each local child's parent as an index into d_loglik_all
 *   d_counts [4]     int64, ACCUMULATED: children with a parent on another rank, of them served from
 *                    staging, windows staged, distinct parents among this rank's children
 * The handle needs max_particles >= 2 * n_local (own slots + staging slots); an attached handle must be
 * rank `rank` of n_total / n_local.  Deterministic: every rank derives the same parents bit for bit.
 * Replaces nothing in the reference (its filter is one process: R:source/dbot_ros/tracker/
 * particle_tracker_node.cpp:277-284); it is the step north_star's "RCCL all-reduce of log-weights before
 * resampling" leaves to each rank. */
int32_t rbs_peer_resample(rbs_handle* h, const double* d_loglik_all, const double* d_uniforms_sorted, int32_t n_total, int32_t n_local,
                          int32_t rank, int32_t min_share, double temperature, int32_t* d_parent_idx, int32_t* d_stage_src,
                          int32_t* d_stage_dst, int32_t* d_parents_local, int64_t* d_counts, void* stream);

/* Rasterize one pose [n_objects][12] -> host float[rows*cols], +inf where uncovered. */
int32_t rbs_render_depth(rbs_handle* h, const double* pose, float* out);
/* Device time in milliseconds of the most recent rbs_loglikes* call (HIP events recorded on the
 * launch stream around its kernels); blocks until it finished. */
int32_t rbs_last_kernel_ms(rbs_handle* h, float* ms);
/* Averages over the timed calls among the last last_n rbs_loglikes* calls (the library brackets
 * every 8th call with HIP events -- rbs_set_timing_every changes that -- and keeps the last 256
 * timed calls), from events recorded on the streams the kernels
 * run on: call_ms = the launch-stream part of a call
 * (frame terms + rectangles kernel, raster kernel); copy_kernel_ms = the copy kernel alone on its own
 * stream (updating calls only, 0 if none).  Blocks until those calls finished. */
int32_t rbs_timing_summary(rbs_handle* h, int32_t last_n, float* call_ms, float* copy_kernel_ms,
                           int32_t* n_used);
/* Bracket every `every`-th rbs_loglikes* call with timing events from now on (default 8; the
 * library keeps the last 256 timed calls).  bench.py's kernel-timing pass uses 2. */
int32_t rbs_set_timing_every(rbs_handle* h, int32_t every);
/* Same window of calls: the raster kernel alone (HIP events around it on the launch stream). */
int32_t rbs_raster_kernel_ms(rbs_handle* h, int32_t last_n, float* raster_kernel_ms);

/* ------------------------------------------------------------------------------------------
 * Device-side tracker (the callers either side of the hot path): the object
 * state transition, the RBC particle-filter step (weights, KL test, multinomial resampling)
 * and the tracker's weighted mean, run on the sensor's device around rbs_loglikes_device with
 * one host synchronisation per frame.  Replaces, for use_gpu trackers,
 *   dbot::ObjectTransitionBuilder<State>  R:source/dbot_ros/tracker/particle_tracker_node.cpp:138-159
 *   dbot::ParticleTrackerBuilder<Tracker> R:source/dbot_ros/tracker/particle_tracker_node.cpp:208-218
 *   tracker->initialize / tracker_->track R:...particle_tracker_node.cpp:252,
 *                                         R:source/dbot_ros/object_tracker_ros.hpp:49
 * States are in MODEL coordinates (pose of the centred mesh frame); the host mirrors do the
 * center_object_frame conversion and the moving average. */
typedef struct rbs_tracker rbs_tracker;

typedef struct rbs_tracker_params {
    double linear_sigma[3];      /* object_transition/linear_sigma_{x,y,z}  */
    double angular_sigma[3];     /* object_transition/angular_sigma_{x,y,z} */
    double velocity_factor;      /* object_transition/velocity_factor       */
    double max_kl_divergence;    /* particle_filter/max_kl_divergence       */
    int32_t n_particles;         /* evaluation_count / #objects; <= the sensor's max_particles */
} rbs_tracker_params;

/* The tracker borrows `sensor` (must outlive it) and drives it on the sensor's own stream. */
int32_t rbs_tracker_create(rbs_handle* sensor, const rbs_tracker_params* params, rbs_tracker** out);
void rbs_tracker_destroy(rbs_tracker* t);
/* default_state: [n_objects*12] = per body position, rotation vector, linear + angular velocity.
 * Particles := zero deltas, weights := uniform, sensor reset. */
int32_t rbs_tracker_initialize(rbs_tracker* t, const double* default_state);
/* One frame.  frame: float32[rows*cols] evaluated-resolution depth (NULL: the caller has
 * already called an rbs_set_observation* variant for this frame).  normals: [n_objects][n][6]
 * standard normals, uniforms: [n_objects][n] in [0,1) -- host-supplied randomness (reproducible,
 * used by the parity tests); either may be NULL to draw on the device (Philox4x32-10 keyed by
 * `seed` and the frame counter).  Outputs: the updated default state [n_objects*12] and the
 * number of resamplings so far. */
int32_t rbs_tracker_track(rbs_tracker* t, const float* frame, const double* normals,
                          const double* uniforms, uint64_t seed, double* out_state,
                          int32_t* out_resamplings);
/* The same frame in two halves, for callers that have the NEXT frame before they need this frame's
 * estimate (dataset replay, a camera that outruns the consumer): rbs_tracker_submit enqueues a
 * frame's work and returns at once -- the caller's frame / normals / uniforms buffers are free on
 * return -- and rbs_tracker_result waits for the OLDEST submitted frame and hands out its state.
 * Up to two frames may be in flight: the second frame's host copy and upload run beside the first
 * frame's kernels (host frames travel on their own stream).  rbs_tracker_track == submit + result.
 * Frames are processed strictly in order; the numbers are those of rbs_tracker_track. */
int32_t rbs_tracker_submit(rbs_tracker* t, const float* frame, const double* normals,
                           const double* uniforms, uint64_t seed);
int32_t rbs_tracker_result(rbs_tracker* t, double* out_state, int32_t* out_resamplings);
/* The same two with the image as dbot's tracker receives it -- rows*cols DOUBLES (Obsrv of R:source/dbot_ros/object_tracker_ros.h:40-41,
 * filled by ri::to_eigen_vector, R:source/dbot_ros/util/ros_interface.h:152-168): converted to float while it is staged (AVX2), and, frame
 * by frame with at most 5 000 evaluations, staged BEHIND the sensor's geometry kernel like any borrowed frame (the buffer is free on return). */
int32_t rbs_tracker_track_f64(rbs_tracker* t, const double* frame, const double* normals, const double* uniforms, uint64_t seed,
                              double* out_state, int32_t* out_resamplings);
int32_t rbs_tracker_submit_f64(rbs_tracker* t, const double* frame, const double* normals, const double* uniforms, uint64_t seed);
/* Inspection: particle deltas [n][n_objects*12], log-weights [n], occlusion slot map [n]
 * (any pointer may be NULL). */
int32_t rbs_tracker_get(rbs_tracker* t, double* particles, double* log_weights, int32_t* indices);

#ifdef __cplusplus
}
#endif
#endif
