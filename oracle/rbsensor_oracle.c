/*
 * rbsensor_oracle.c -- see rbsensor_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.
 *
 * Build with -ffp-contract=off (oracle/Makefile does): the geometry below is specified
 * as individually rounded IEEE-754 binary64 operations and the HIP path reproduces the
 * same operation order bit-for-bit.
 *
 * Each function cites the reference call site whose behaviour it restates
 * (R: = /root/reference/) and the SURVEY.md appendix paragraph it follows.
 */
#include "rbsensor_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* EXPOSURE VARIANTS (round 5).  The restatement is unpinned: where SURVEY Appendix A left a rule open, this
 * file chose one (divergence ledger).  These compile-time switches build the oracle with the rule
 * upstream MIGHT have chosen instead, so that tests/test_oracle_variants.py can print how far log-likelihoods and resampled
 * parents would move -- the size of the exposure, not a pin.  0 everywhere = the rule of record (what the device matches).
 *   ORC_VARIANT_COVERAGE   ledger L1: 0 closed triangle at integer sample points, either winding;
 *                                      1 top-left fill rule (a sample exactly ON an edge belongs to one of the two triangles);
 *                                      2 scan-line spans (per row the columns between the edges' intersections, inclusive:
 *                                        SURVEY A.2's own wording);
 *                                      3 sample points at pixel CENTRES (col + 0.5, row + 0.5): the GL convention (A.7)
 *   ORC_VARIANT_ROUNDING   ledger L6: 0 float temporaries (prior, a, b, p_bg, their sum and ratios);
 *                                      1 binary64 throughout, only the stored posterior rounded to float
 *   ORC_VARIANT_NONFINITE  ledger L4: 0 a pixel whose observation is not finite (NaN or +-inf) is skipped;
 *                                      1 only NaN is skipped, +-inf is evaluated like any other reading */
#ifndef ORC_VARIANT_COVERAGE
#define ORC_VARIANT_COVERAGE 0
#endif
#ifndef ORC_VARIANT_ROUNDING
#define ORC_VARIANT_ROUNDING 0
#endif
#ifndef ORC_VARIANT_NONFINITE
#define ORC_VARIANT_NONFINITE 0
#endif

#define ORC_MAX_DEPTH 6.0       /* hard-coded max_depth        */
#define ORC_HALF_LIFE_DEPTH 1.0 /* hard-coded half_life_depth  */

struct orc_sensor {
    orc_config cfg;
    int32_t n_tri_total;
    double* soup;        /* [n_tri_total][9] model-space triangle soup      */
    int32_t* tri_begin;  /* [n_objects+1] triangle range per body           */
    size_t npx;
    float* frame;        /* current observation, float metres               */
    /* occlusion state, double-buffered over slots */
    float* occ[2];       /* [max_particles][npx]                            */
    int32_t* stamp[2];   /* LAZY only: frame index of last update per pixel */
    int cur;
    int32_t clock;       /* frame counter: t_now = clock * delta_time       */
    int32_t last_update_clock; /* EAGER: clock at the last updating call    */
    float background;    /* EAGER: value of a never-covered pixel at the last updating call */
    /* scratch */
    float* depth;        /* [npx] */
    int32_t* covered;    /* [npx] list of covered pixel ids */
    /* per-thread scratch of the multi-threaded baseline: allocated once, kept */
    int n_scratch;
    float** tdepth;
    int32_t** tcov;
    double* abs_sum;     /* [max_particles] sum of |per-pixel log term| of the last loglikes call */
    double lambda;       /* ln 2 / half_life_depth */
};

/* ---------------------------------------------------------------- pixel model */

/* KinectPixelModel::Probability, visible branch.
 * Constants: R:source/dbot_ros/tracker/particle_tracker_node.cpp:183-188,
 * defaults R:config/particle_tracker.yaml:47-49. */
double orc_prob_visible(const orc_sensor* s, double o, double r)
{
    const double tw = s->cfg.tail_weight;
    const double sigma = s->cfg.model_sigma + s->cfg.sigma_factor * o * o;
    if (isinf(r)) return tw / ORC_MAX_DEPTH;
    const double d = r - o;
    return tw / ORC_MAX_DEPTH +
           (1.0 - tw) * exp(-(d * d) / (2.0 * sigma * sigma)) / (sqrt(2.0 * M_PI) * sigma);
}

/* KinectPixelModel::Probability, occluded branch; r = +inf gives the
 * background density p_bg(o) used as the denominator of the likelihood ratio. */
double orc_prob_occluded(const orc_sensor* s, double o, double r)
{
    const double tw = s->cfg.tail_weight;
    const double lam = s->lambda;
    const double sigma = s->cfg.model_sigma + s->cfg.sigma_factor * o * o;
    if (isinf(r))
        return tw / ORC_MAX_DEPTH +
               (1.0 - tw) * lam * exp(0.5 * lam * (-2.0 * o + lam * sigma * sigma));
    return tw / ORC_MAX_DEPTH +
           (1.0 - tw) * lam * exp(0.5 * lam * (2.0 * r - 2.0 * o + lam * sigma * sigma)) *
               (1.0 + erf((r - o + lam * sigma * sigma) / (sqrt(2.0) * sigma))) /
               (2.0 * (exp(r * lam) - 1.0));
}

/* OcclusionModel propagate.
 * Constants: R:source/dbot_ros/tracker/particle_tracker_node.cpp:176-181,
 * defaults R:config/particle_tracker.yaml:41-43; delta_time R:...particle_tracker_node.cpp:189. */
double orc_propagate(const orc_sensor* s, double occ, double dt)
{
    const double p_oo = s->cfg.p_occluded_occluded;
    const double c = p_oo - s->cfg.p_occluded_visible;
    const double pow_c = exp(dt * log(c));
    const double new_visible = pow_c * (1.0 - occ) + (1.0 - p_oo) * (pow_c - 1.0) / (c - 1.0);
    return 1.0 - new_visible;
}

/* The device rule's affine form of orc_propagate over n frames:
 * propagate(occ, n*dt) = alpha*occ + beta, coefficients rounded once to float. */
void orc_eager_coeffs(const orc_sensor* s, int32_t n_frames, float* alpha, float* beta)
{
    const double p_oo = s->cfg.p_occluded_occluded;
    const double c = p_oo - s->cfg.p_occluded_visible;
    const double a = exp(((double)n_frames * s->cfg.delta_time) * log(c));
    const double g = (1.0 - p_oo) * (a - 1.0) / (c - 1.0);
    *alpha = (float)a;
    *beta = (float)((1.0 - a) - g);
}

/* EAGER prior of one pixel: the affine step, then the background snap -- a value within
 * ORC_SNAP_TAU of the never-covered level `bg_now` (the same step applied to the background
 * scalar) IS the background.  The snap makes "differs from the background" a finite-time
 * property (a float contraction otherwise stalls up to 0.5 ulp / (1 - alpha) away from its
 * limit for ever), which is what lets an implementation keep only a window of each plane. */
float orc_eager_prior(float alpha, float beta, float occ, float bg_now)
{
    const float x = fmaf(alpha, occ, beta);
    return fabsf(x - bg_now) <= ORC_SNAP_TAU ? bg_now : x;
}

/* ---------------------------------------------------------------- renderer */

/* RigidBodyRenderer::Render restated with this repo's written coverage rule:
 *   - sample points are INTEGER pixel coordinates (col,row);
 *   - a pixel is covered by a triangle iff all three edge functions are >= 0 or all
 *     are <= 0 (closed triangle, either winding, no culling);
 *   - triangles with a vertex at Z <= 0 or with zero projected area are skipped;
 *   - depth = plane/ray intersection Z, rounded once to float, z-min across triangles.
 * Mesh source R:source/dbot_ros/tracker/particle_tracker_node.cpp:89-97; K and resolution
 * R:...particle_tracker_node.cpp:112-121. */
static void raster_triangle(const orc_sensor* s, const double* tri, const double* Rt,
                            float* depth, int32_t* covered, int32_t* n_covered)
{
    const double fx = s->cfg.fx, fy = s->cfg.fy, cx = s->cfg.cx, cy = s->cfg.cy;
    double X[3], Y[3], Z[3], u[3], v[3];
    for (int k = 0; k < 3; ++k) {
        const double vx = tri[3 * k + 0], vy = tri[3 * k + 1], vz = tri[3 * k + 2];
        X[k] = ((Rt[0] * vx + Rt[1] * vy) + Rt[2] * vz) + Rt[9];
        Y[k] = ((Rt[3] * vx + Rt[4] * vy) + Rt[5] * vz) + Rt[10];
        Z[k] = ((Rt[6] * vx + Rt[7] * vy) + Rt[8] * vz) + Rt[11];
    }
    if (!(Z[0] > 0.0 && Z[1] > 0.0 && Z[2] > 0.0)) return;
    for (int k = 0; k < 3; ++k) {
        const double iz = 1.0 / Z[k];
        u[k] = fx * (X[k] * iz) + cx;
        v[k] = fy * (Y[k] * iz) + cy;
    }
    const double e01u = u[1] - u[0], e01v = v[1] - v[0];
    const double e12u = u[2] - u[1], e12v = v[2] - v[1];
    const double e20u = u[0] - u[2], e20v = v[0] - v[2];
    const double area2 = e01u * (v[2] - v[0]) - e01v * (u[2] - u[0]);
    if (!(area2 != 0.0) || !(fabs(area2) < INFINITY)) return;

    const double ax = X[1] - X[0], ay = Y[1] - Y[0], az = Z[1] - Z[0];
    const double bx = X[2] - X[0], by = Y[2] - Y[0], bz = Z[2] - Z[0];
    const double nx = ay * bz - az * by;
    const double ny = az * bx - ax * bz;
    const double nz = ax * by - ay * bx;
    const double nv0 = (nx * X[0] + ny * Y[0]) + nz * Z[0];
    const double pa = nx / fx;
    const double pb = ny / fy;
    const double pc = (nz - pa * cx) - pb * cy;

    const double umin = fmin(fmin(u[0], u[1]), u[2]), umax = fmax(fmax(u[0], u[1]), u[2]);
    const double vmin = fmin(fmin(v[0], v[1]), v[2]), vmax = fmax(fmax(v[0], v[1]), v[2]);
#if ORC_VARIANT_COVERAGE == 3
    const double smp = 0.5;   /* (pixel centres: the candidate samples of the bounding box move with the grid) */
#else
    const double smp = 0.0;
#endif
    const double xlo_d = fmax(ceil(umin - smp), 0.0), xhi_d = fmin(floor(umax - smp), (double)(s->cfg.cols - 1));
    const double ylo_d = fmax(ceil(vmin - smp), 0.0), yhi_d = fmin(floor(vmax - smp), (double)(s->cfg.rows - 1));
    if (!(xlo_d <= xhi_d) || !(ylo_d <= yhi_d)) return;
    const int xlo = (int)xlo_d, xhi = (int)xhi_d, ylo = (int)ylo_d, yhi = (int)yhi_d;

    const double shift = smp;   /* the sample of pixel (col, row) sits at (col + shift, row + shift) */
#define ORC_XLO xlo
#define ORC_XHI xhi
#define ORC_YLO ylo
#define ORC_YHI yhi
    (void)xlo; (void)xhi; (void)ylo; (void)yhi; (void)e12u; (void)e12v; (void)e20u; (void)e20v;   /* (not every variant uses all of them) */
    for (int row = ORC_YLO; row <= ORC_YHI; ++row) {
        const double py = (double)row + shift;
#if ORC_VARIANT_COVERAGE == 2
        /* scan line: the span of this row between the edges' intersections, both ends inclusive */
        double xl = INFINITY, xr = -INFINITY;
        for (int k = 0; k < 3; ++k) {
            const int a = k, b = (k + 1) % 3;
            if (v[a] == v[b]) {
                if (v[a] == py) { xl = fmin(xl, fmin(u[a], u[b])); xr = fmax(xr, fmax(u[a], u[b])); }
                continue;
            }
            if (py >= fmin(v[a], v[b]) && py <= fmax(v[a], v[b])) {
                const double x = u[a] + (py - v[a]) * ((u[b] - u[a]) / (v[b] - v[a]));
                xl = fmin(xl, x); xr = fmax(xr, x);
            }
        }
#endif
        for (int col = ORC_XLO; col <= ORC_XHI; ++col) {
            const double px = (double)col + shift;
#if ORC_VARIANT_COVERAGE == 2
            const int in = px >= xl && px <= xr;
#else
            const double E0 = e01u * (py - v[0]) - e01v * (px - u[0]);
            const double E1 = e12u * (py - v[1]) - e12v * (px - u[1]);
            const double E2 = e20u * (py - v[2]) - e20v * (px - u[2]);
#if ORC_VARIANT_COVERAGE == 1
            /* top-left: inside = strictly on the interior side of every edge, or exactly on an edge that owns its samples
             * (one fixed half of the edge directions, so that a sample on a shared edge belongs to exactly one triangle) */
            const double sg = area2 > 0.0 ? 1.0 : -1.0;
            const double F[3] = {sg * E0, sg * E1, sg * E2};
            const double du[3] = {sg * e01u, sg * e12u, sg * e20u}, dv[3] = {sg * e01v, sg * e12v, sg * e20v};
            int in = 1;
            for (int k = 0; k < 3; ++k) {
                const int owns = dv[k] > 0.0 || (dv[k] == 0.0 && du[k] < 0.0);
                if (!(F[k] > 0.0 || (F[k] == 0.0 && owns))) in = 0;
            }
#else
            const int in = (E0 >= 0.0 && E1 >= 0.0 && E2 >= 0.0) ||
                           (E0 <= 0.0 && E1 <= 0.0 && E2 <= 0.0);
#endif
#endif
            if (!in) continue;
            const double den = (pa * px + pb * py) + pc;
            const float zf = (float)(nv0 / den);
            if (!(zf > 0.0f) || !(zf < INFINITY)) continue;
            const int32_t pid = row * s->cfg.cols + col;
            float* d = &depth[pid];
            if (zf < *d) {
                if (!(*d < INFINITY)) covered[(*n_covered)++] = pid; /* first hit */
                *d = zf;
            }
        }
    }
}

/* depth must be all +inf on entry; returns the list of covered pixel ids (the upstream
 * renderer's intersect_indices) in covered[0..n). */
static int32_t render_into(const orc_sensor* s, const double* pose, float* depth,
                           int32_t* covered)
{
    int32_t n = 0;
    for (int b = 0; b < s->cfg.n_objects; ++b) {
        const double* Rt = pose + 12 * b;
        for (int32_t t = s->tri_begin[b]; t < s->tri_begin[b + 1]; ++t)
            raster_triangle(s, s->soup + 9 * (size_t)t, Rt, depth, covered, &n);
    }
    return n;
}

int32_t orc_render(const orc_sensor* s, const double* pose, float* depth_out)
{
    for (size_t p = 0; p < s->npx; ++p) depth_out[p] = INFINITY;
    return render_into(s, pose, depth_out, s->covered);
}

/* ---------------------------------------------------------------- life cycle */

orc_sensor* orc_create(const orc_config* cfg)
{
    orc_sensor* s = (orc_sensor*)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->cfg = *cfg;
    s->npx = (size_t)cfg->rows * cfg->cols;
    s->lambda = -log(0.5) / ORC_HALF_LIFE_DEPTH;
    s->tri_begin = (int32_t*)calloc((size_t)cfg->n_objects + 1, sizeof(int32_t));
    for (int b = 0; b < cfg->n_objects; ++b)
        s->tri_begin[b + 1] = s->tri_begin[b] + cfg->triangle_counts[b];
    s->n_tri_total = s->tri_begin[cfg->n_objects];
    s->soup = (double*)malloc(sizeof(double) * 9 * (size_t)(s->n_tri_total > 0 ? s->n_tri_total : 1));
    size_t voff = 0, toff = 0;
    for (int b = 0; b < cfg->n_objects; ++b) {
        for (int32_t t = 0; t < cfg->triangle_counts[b]; ++t) {
            for (int k = 0; k < 3; ++k) {
                const int32_t vi = cfg->triangles[3 * (toff + t) + k];
                for (int c = 0; c < 3; ++c)
                    s->soup[9 * (toff + t) + 3 * k + c] = cfg->vertices[3 * (voff + vi) + c];
            }
        }
        voff += (size_t)cfg->vertex_counts[b];
        toff += (size_t)cfg->triangle_counts[b];
    }
    const size_t plane = s->npx * (size_t)cfg->max_particles;
    s->frame = (float*)malloc(sizeof(float) * s->npx);
    s->depth = (float*)malloc(sizeof(float) * s->npx);
    s->covered = (int32_t*)malloc(sizeof(int32_t) * s->npx);
    s->abs_sum = (double*)calloc((size_t)cfg->max_particles, sizeof(double));
    for (int k = 0; k < 2; ++k) {
        s->occ[k] = (float*)malloc(sizeof(float) * plane);
        s->stamp[k] = cfg->occlusion_mode == ORC_OCC_LAZY
                          ? (int32_t*)malloc(sizeof(int32_t) * plane)
                          : NULL;
    }
    /* copies of the caller's arrays are held in soup; drop the borrowed pointers */
    s->cfg.vertices = NULL;
    s->cfg.triangles = NULL;
    s->cfg.vertex_counts = NULL;
    s->cfg.triangle_counts = NULL;
    for (size_t p = 0; p < s->npx; ++p) s->frame[p] = NAN;
    for (size_t p = 0; p < s->npx; ++p) s->depth[p] = INFINITY;
    orc_reset(s);
    return s;
}

void orc_destroy(orc_sensor* s)
{
    if (!s) return;
    free(s->soup);
    free(s->tri_begin);
    free(s->frame);
    free(s->depth);
    free(s->covered);
    free(s->abs_sum);
    for (int t = 0; t < s->n_scratch; ++t) {
        free(s->tdepth[t]);
        free(s->tcov[t]);
    }
    free(s->tdepth);
    free(s->tcov);
    for (int k = 0; k < 2; ++k) {
        free(s->occ[k]);
        free(s->stamp[k]);
    }
    free(s);
}

/* RbSensor::reset(), triggered by tracker->initialize at
 * R:source/dbot_ros/tracker/particle_tracker_node.cpp:252 (SURVEY A.4 last line). */
void orc_reset(orc_sensor* s) { orc_reset_mt(s, 1); }

/* n_threads > 1 (the multi-threaded baseline): every slot of BOTH buffers is first touched by
 * the thread that will evaluate it under schedule(static), so its pages land on that thread's
 * NUMA node instead of all on the creating thread's. */
void orc_reset_mt(orc_sensor* s, int32_t n_threads)
{
    const float init = (float)s->cfg.initial_occlusion_prob;
    const int32_t slots = s->cfg.max_particles;
    s->cur = 0;
    s->clock = 0;
    s->last_update_clock = 0;
    s->background = init;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 1 ? n_threads : 1)
#endif
    for (int32_t i = 0; i < slots; ++i) {
        for (int k = 0; k < (n_threads > 1 ? 2 : 1); ++k) {
            float* o = s->occ[k] + (size_t)i * s->npx;
            for (size_t p = 0; p < s->npx; ++p) o[p] = init;
            if (s->stamp[k]) memset(s->stamp[k] + (size_t)i * s->npx, 0, sizeof(int32_t) * s->npx);
        }
    }
}

/* RbSensor::set_observation -- frame layout per R:source/dbot_ros/util/ros_interface.h:161-165;
 * consumed through tracker_->track at R:source/dbot_ros/object_tracker_ros.hpp:49. */
void orc_set_observation(orc_sensor* s, const double* depth)
{
    for (size_t p = 0; p < s->npx; ++p) s->frame[p] = (float)depth[p];
    s->clock += 1;
}

/* ---------------------------------------------------------------- the hot function */

/* One covered, observed pixel: the term log((a + b) / p_bg) it adds to the particle's
 * log-likelihood and its posterior occlusion b / (a + b).  Rounding points: a = p_vis (1 - occ),
 * b = p_occ occ and p_bg -> float ("float temporaries upstream"); a + b and both ratios in float;
 * log in double. */
#if ORC_VARIANT_ROUNDING == 1
/* binary64 throughout (the prior included): only the stored posterior is rounded to float */
static double pixel_term_f64(const orc_sensor* s, float o, float r, double occ, float* posterior)
{
    const double a = orc_prob_visible(s, (double)o, (double)r) * (1.0 - occ);
    const double b = orc_prob_occluded(s, (double)o, (double)r) * occ;
    const double pbg = orc_prob_occluded(s, (double)o, INFINITY);
    *posterior = (float)(b / (a + b));
    return log((a + b) / pbg);
}
#endif
double orc_pixel_term(const orc_sensor* s, float o, float r, float occ, float* posterior)
{
#if ORC_VARIANT_ROUNDING == 1
    return pixel_term_f64(s, o, r, (double)occ, posterior);
#endif
    const float a = (float)(orc_prob_visible(s, (double)o, (double)r) * (1.0 - (double)occ));
    const float b = (float)(orc_prob_occluded(s, (double)o, (double)r) * (double)occ);
    const float pbg = (float)orc_prob_occluded(s, (double)o, INFINITY);
    const float sum = a + b;
    *posterior = b / sum;
    return log((double)(sum / pbg));
}

/* The same for n pixels (tests: the device's own exp / erfc / log against libm's, pixel by pixel). */
void orc_pixel_terms(const orc_sensor* s, const float* o, const float* r, const float* occ, int64_t n,
                     double* term, float* posterior)
{
    for (int64_t i = 0; i < n; ++i) term[i] = orc_pixel_term(s, o[i], r[i], occ[i], &posterior[i]);
}

/* One particle of orc_loglikes; depth/covered are per-thread scratch (depth all +inf on entry
 * and on exit). */
static double loglik_one(orc_sensor* s, const double* pose, int32_t parent, int32_t child,
                         int32_t update, float alpha, float beta, float bg_now, float* depth,
                         int32_t* covered)
{
    const int lazy = s->cfg.occlusion_mode == ORC_OCC_LAZY;
    const int src = s->cur, dst = 1 - s->cur;
    const size_t poff = (size_t)parent * s->npx;
    const size_t coff = (size_t)child * s->npx;
    const float* pocc = s->occ[src] + poff;
    const int32_t* pstamp = lazy ? s->stamp[src] + poff : NULL;
    float* cocc = s->occ[dst] + coff;
    int32_t* cstamp = lazy ? s->stamp[dst] + coff : NULL;

    if (update) {
        if (lazy) {
            memcpy(cocc, pocc, sizeof(float) * s->npx);
            memcpy(cstamp, pstamp, sizeof(int32_t) * s->npx);
        } else {
            for (size_t p = 0; p < s->npx; ++p) cocc[p] = orc_eager_prior(alpha, beta, pocc[p], bg_now);
        }
    }

    const int32_t ncov = render_into(s, pose, depth, covered);

    double ll = 0.0, abs_ll = 0.0;
    for (int32_t k = 0; k < ncov; ++k) {
        const size_t p = (size_t)covered[k];
        const float r = depth[p];
        depth[p] = INFINITY; /* leave the scratch buffer clean for the next particle */
        const float o = s->frame[p];
#if ORC_VARIANT_NONFINITE == 1
        if (isnan(o)) continue;
#else
        if (!isfinite(o)) continue;
#endif
        float occ;
        float post;
#if ORC_VARIANT_ROUNDING == 1
        double occ_d;
        if (lazy) occ_d = orc_propagate(s, (double)pocc[p], (double)(s->clock - pstamp[p]) * s->cfg.delta_time);
        else occ_d = (double)orc_eager_prior(alpha, beta, pocc[p], bg_now);
        occ = (float)occ_d;
        (void)occ;
        const double term = pixel_term_f64(s, o, r, occ_d, &post);
#else
        if (lazy) {
            const double dt = (double)(s->clock - pstamp[p]) * s->cfg.delta_time;
            occ = (float)orc_propagate(s, (double)pocc[p], dt);
        } else {
            occ = orc_eager_prior(alpha, beta, pocc[p], bg_now);
        }
        const double term = orc_pixel_term(s, o, r, occ, &post);
#endif
        ll += term;
        abs_ll += fabs(term);
        if (update) {
            cocc[p] = post;
            if (lazy) cstamp[p] = s->clock;
        }
    }
    s->abs_sum[child] = abs_ll;
    return ll;
}

/* KinectImageModel::loglikes restated; selected by use_gpu=false at
 * R:source/dbot_ros/tracker/particle_tracker_node.cpp:165.  Rounding points:
 *   prior occlusion    -> float
 *   a = p_vis*(1-occ), b = p_occ*occ, p_bg -> float; a+b and (a+b)/p_bg in float
 *   log(...) in double, accumulated in double
 *   posterior occlusion b/(a+b) in float.
 * Pixels whose observation is not finite contribute 0 and are left untouched.
 * n_threads <= 1: the reference's single-threaded loop; > 1: the same per-particle work
 * spread over OpenMP threads (particles are independent; results are identical). */
void orc_loglikes_mt(orc_sensor* s, const double* poses, int32_t* indices, int32_t n,
                     int32_t update, double* out_loglik, int32_t n_threads)
{
    float alpha = 1.0f, beta = 0.0f;
    if (s->cfg.occlusion_mode != ORC_OCC_LAZY)
        orc_eager_coeffs(s, s->clock - s->last_update_clock, &alpha, &beta);
    const float bg_now = fmaf(alpha, s->background, beta);
    const size_t pstride = (size_t)12 * s->cfg.n_objects;
    if (n_threads <= 1) {
        for (int32_t i = 0; i < n; ++i)
            out_loglik[i] = loglik_one(s, poses + (size_t)i * pstride, indices[i], i, update, alpha, beta,
                                       bg_now, s->depth, s->covered);
    } else {
#ifdef _OPENMP
        /* per-thread scratch lives in the sensor: allocated (and first touched by its thread)
         * once, not mmap'ed and unmapped in every call */
        if (s->n_scratch < n_threads) {
            s->tdepth = (float**)realloc(s->tdepth, sizeof(float*) * (size_t)n_threads);
            s->tcov = (int32_t**)realloc(s->tcov, sizeof(int32_t*) * (size_t)n_threads);
            for (int t = s->n_scratch; t < n_threads; ++t) { s->tdepth[t] = NULL; s->tcov[t] = NULL; }
            s->n_scratch = n_threads;
        }
#pragma omp parallel num_threads(n_threads)
        {
            const int t = omp_get_thread_num();
            if (!s->tdepth[t]) {
                s->tdepth[t] = (float*)malloc(sizeof(float) * s->npx);
                s->tcov[t] = (int32_t*)malloc(sizeof(int32_t) * s->npx);
                for (size_t p = 0; p < s->npx; ++p) s->tdepth[t][p] = INFINITY;
            }
            float* depth = s->tdepth[t];
            int32_t* covered = s->tcov[t];
            /* static: particle i is always evaluated by the same thread, whose NUMA node holds
             * slot i of both buffers (orc_reset_mt) */
#pragma omp for schedule(static)
            for (int32_t i = 0; i < n; ++i)
                out_loglik[i] = loglik_one(s, poses + (size_t)i * pstride, indices[i], i, update, alpha,
                                           beta, bg_now, depth, covered);
        }
#else
        for (int32_t i = 0; i < n; ++i)
            out_loglik[i] = loglik_one(s, poses + (size_t)i * pstride, indices[i], i, update, alpha, beta,
                                       bg_now, s->depth, s->covered);
#endif
    }
    if (update) {
        s->cur = 1 - s->cur;
        s->last_update_clock = s->clock;
        if (s->cfg.occlusion_mode != ORC_OCC_LAZY) s->background = bg_now;
        for (int32_t i = 0; i < n; ++i) indices[i] = i;
    }
}

void orc_loglikes(orc_sensor* s, const double* poses, int32_t* indices, int32_t n,
                  int32_t update, double* out_loglik)
{
    orc_loglikes_mt(s, poses, indices, n, update, out_loglik, 1);
}

void orc_get_occlusion(const orc_sensor* s, int32_t slot, float* out)
{
    memcpy(out, s->occ[s->cur] + (size_t)slot * s->npx, sizeof(float) * s->npx);
}

void orc_set_occlusion(orc_sensor* s, int32_t slot, const float* plane)
{
    memcpy(s->occ[s->cur] + (size_t)slot * s->npx, plane, sizeof(float) * s->npx);
    if (s->stamp[s->cur]) {
        int32_t* st = s->stamp[s->cur] + (size_t)slot * s->npx;
        for (size_t p = 0; p < s->npx; ++p) st[p] = s->clock;
    }
}

void orc_get_occlusion_now(const orc_sensor* s, int32_t slot, float* out)
{
    const float* occ = s->occ[s->cur] + (size_t)slot * s->npx;
    if (s->cfg.occlusion_mode == ORC_OCC_LAZY) {
        const int32_t* st = s->stamp[s->cur] + (size_t)slot * s->npx;
        for (size_t p = 0; p < s->npx; ++p)
            out[p] = (float)orc_propagate(s, (double)occ[p],
                                          (double)(s->clock - st[p]) * s->cfg.delta_time);
    } else {
        float alpha, beta;
        orc_eager_coeffs(s, s->clock - s->last_update_clock, &alpha, &beta);
        const float bg_now = fmaf(alpha, s->background, beta);
        for (size_t p = 0; p < s->npx; ++p) out[p] = orc_eager_prior(alpha, beta, occ[p], bg_now);
    }
}

float orc_background(const orc_sensor* s) { return s->background; }

void orc_last_abs_sums(const orc_sensor* s, double* out, int32_t n)
{
    memcpy(out, s->abs_sum, sizeof(double) * (size_t)n);
}
