/*
 * rbsensor_oracle.h -- CPU restatement ("oracle") of the dbot Rao-Blackwellised
 * depth-image observation model that dbot_ros's particle tracker drives.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing on the product path (dbot_ros_amd/, the
 * C-ABI library, the C++ shim) may include, link or call this file.  It is
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the catkin packages
 * `dbot` and `fl`, which /root/reference only find_package()s
 * (R:CMakeLists.txt:34-47, R:package.xml:27-28, no version pin) and which are
 * absent from the build container; the reference ships no tests, fixtures or
 * golden vectors.  This file follows SURVEY.md Appendix A (recalled upstream
 * formulas) and the observable contract of the reference's call sites:
 *   parameters  R:source/dbot_ros/tracker/particle_tracker_node.cpp:164-199
 *   defaults    R:config/particle_tracker.yaml:38-49
 *   frame input R:source/dbot_ros/util/ros_interface.h:152-168 (row-major, metres, NaN)
 *   intrinsics  R:source/dbot_ros/util/ros_camera_data_provider.cpp:66-76
 *
 * The written-down rules below (coverage rule, rounding points) are THE
 * specification the HIP path is tested against.
 */
#ifndef RBSENSOR_ORACLE_H
#define RBSENSOR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* occlusion bookkeeping modes */
enum {
    ORC_OCC_LAZY = 0,  /* reference CPU semantics: per-pixel last-update stamp, propagate by
                          the pixel's own elapsed time in double            */
    ORC_OCC_EAGER = 1  /* the device rule: every pixel advanced on every updating call with
                          occ' = snap(fmaf(alpha_f, occ, beta_f)) (float), no stamps; snap(x)
                          = bg' if |x - bg'| <= ORC_SNAP_TAU, where bg' is the same step
                          applied to the never-covered level bg (initial_occlusion_prob at
                          reset)                                                              */
};
#define ORC_SNAP_TAU 0x1p-18f

typedef struct orc_config {
    int32_t rows, cols;
    double fx, fy, cx, cy;         /* K already divided by the down-sampling factor */
    int32_t max_particles;
    int32_t n_objects;
    const double* vertices;        /* concatenated xyz, sum(vertex_counts)*3          */
    const int32_t* vertex_counts;  /* [n_objects]                                     */
    const int32_t* triangles;      /* concatenated, indices local to each object      */
    const int32_t* triangle_counts;/* [n_objects]                                     */
    double p_occluded_visible, p_occluded_occluded, initial_occlusion_prob;
    double tail_weight, model_sigma, sigma_factor;
    double delta_time;
    int32_t occlusion_mode;        /* ORC_OCC_* */
} orc_config;

typedef struct orc_sensor orc_sensor;

orc_sensor* orc_create(const orc_config* cfg);
void orc_destroy(orc_sensor* s);

/* RbSensor::reset() -- every slot := initial_occlusion_prob, stamps := 0, clock := 0 */
void orc_reset(orc_sensor* s);

/* RbSensor::set_observation(image): depth[rows*cols] row-major metres, NaN = no reading.
 * Stored as float; advances the model clock by one delta_time. */
void orc_set_observation(orc_sensor* s, const double* depth);

/* RbSensor::loglikes(deltas, indices, update).
 * poses: [n][n_objects][12] = R (row-major 3x3) | t, absolute camera-frame pose per body.
 * indices: in = parent occlusion slot per particle; out = identity when update != 0.
 * out_loglik: [n] doubles. */
void orc_loglikes(orc_sensor* s, const double* poses, int32_t* indices, int32_t n,
                  int32_t update, double* out_loglik);

/* Same, with the per-particle loop spread over n_threads OpenMP threads (the reference's CPU
 * model is single-threaded; this is the "all host cores" baseline of BASELINE.md section 2). */
void orc_loglikes_mt(orc_sensor* s, const double* poses, int32_t* indices, int32_t n,
                     int32_t update, double* out_loglik, int32_t n_threads);

/* Stored occlusion plane of a slot (float[rows*cols]). */
void orc_get_occlusion(const orc_sensor* s, int32_t slot, float* out);
/* Overwrite a slot's stored plane (LAZY: stamps := current clock, i.e. the plane is taken to
 * be "as of now", which is what orc_get_occlusion_now of the sending side delivers). */
void orc_set_occlusion(orc_sensor* s, int32_t slot, const float* plane);
/* Occlusion plane of a slot advanced to the current clock (what the next evaluation
 * would use as its prior if set_observation were not called again). */
void orc_get_occlusion_now(const orc_sensor* s, int32_t slot, float* out);

/* RigidBodyRenderer::Render restated: depth_out[rows*cols], +inf where uncovered.
 * pose: [n_objects][12]. Returns number of covered pixels. */
int32_t orc_render(const orc_sensor* s, const double* pose, float* depth_out);

/* scalar pieces, exposed for unit tests */
double orc_prob_visible(const orc_sensor* s, double obs, double rendered);  /* rendered may be +inf */
double orc_prob_occluded(const orc_sensor* s, double obs, double rendered); /* rendered may be +inf */
double orc_propagate(const orc_sensor* s, double occ, double dt);
void orc_eager_coeffs(const orc_sensor* s, int32_t n_frames, float* alpha, float* beta);
float orc_eager_prior(float alpha, float beta, float occ, float bg_now);
/* One covered, observed pixel: its log-likelihood term and posterior occlusion. */
double orc_pixel_term(const orc_sensor* s, float obs, float rendered, float occ, float* posterior);
void orc_pixel_terms(const orc_sensor* s, const float* obs, const float* rendered, const float* occ, int64_t n,
                     double* term, float* posterior);
float orc_background(const orc_sensor* s);  /* EAGER: never-covered level at the last updating call */
/* Per particle of the last orc_loglikes* call: the sum of the MAGNITUDES of the per-pixel log
 * terms its log-likelihood adds up -- the conditioning of that sum, against which a float32
 * implementation's error is bounded (tests/test_gpu_f32.py). */
void orc_last_abs_sums(const orc_sensor* s, double* out, int32_t n);
/* orc_reset with NUMA-aware first touch for the n_threads baseline (see the .c file). */
void orc_reset_mt(orc_sensor* s, int32_t n_threads);

/* ---- tracker_oracle.c: transition + RBC filter step + tracker mean ---- */
typedef struct orc_tracker orc_tracker;
/* absolute poses from state deltas around default poses; see tracker_oracle.c */
void orc_compose_poses(const double* deltas, const double* deflt, int32_t stride, int32_t n, int32_t parts, double* out);
orc_tracker* orc_tracker_create(orc_sensor* s, int32_t parts, int32_t n, const double* sigma6,
                                double velocity_factor, double max_kl_divergence);
void orc_tracker_destroy(orc_tracker* t);
void orc_tracker_initialize(orc_tracker* t, const double* default_state);
/* frame: double[rows*cols]; normals [parts][n][6]; uniforms [parts][n]; out_state [parts*12] */
void orc_tracker_track(orc_tracker* t, const double* frame, const double* normals, const double* uniforms,
                       double* out_state, int32_t* out_resamplings);
void orc_tracker_get(const orc_tracker* t, double* particles, double* log_weights, int32_t* indices);

#ifdef __cplusplus
}
#endif
#endif
