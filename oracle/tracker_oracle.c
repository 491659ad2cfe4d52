/*
 * tracker_oracle.c -- CPU restatement of the callers either side of the hot path (SURVEY 8
 * f1/f2): object state transition, Rao-Blackwellised coordinate particle filter step and the
 * tracker's weighted mean / re-centring, driving the sensor oracle of rbsensor_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as rbsensor_oracle.h).  PARITY UNPINNED: follows
 * SURVEY.md Appendix A.1 / A.6 (recalled upstream dbot behaviour) and the reference's call sites
 *   ObjectTransitionBuilder parameters   R:source/dbot_ros/tracker/particle_tracker_node.cpp:138-159
 *   ParticleTrackerBuilder parameters    R:source/dbot_ros/tracker/particle_tracker_node.cpp:208-218
 *   tracker->initialize / ->track        R:...particle_tracker_node.cpp:252, R:source/dbot_ros/object_tracker_ros.hpp:49
 * Plain sequential loops, double precision, host-supplied randomness.
 */
#include "rbsensor_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BODY 12

typedef struct orc_tracker {
    orc_sensor* s;
    int n, parts, D;
    double sigma[6], vf, max_kl;
    double *old, *cur, *noise;   /* [n][D], [n][D], [n][parts][6] */
    double *logw, *ll;           /* [n] */
    int32_t* idx;                /* [n] occlusion slot map */
    double* deflt;               /* [D] */
    int resamplings;
} orc_tracker;

static void rotvec_to_matrix(const double* rv, double* R)
{
    const double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    const double half = 0.5 * angle;
    const double k = angle < 1e-9 ? 0.5 - angle * angle / 48.0 : sin(half) / angle;
    const double w = cos(half), x = rv[0] * k, y = rv[1] * k, z = rv[2] * k;
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
}

static void matrix_to_rotvec(const double* R, double* rv)
{
    const double sx = 0.5 * (R[7] - R[5]), sy = 0.5 * (R[2] - R[6]), sz = 0.5 * (R[3] - R[1]);
    const double sn = sqrt(sx * sx + sy * sy + sz * sz);
    const double cs = 0.5 * ((R[0] + R[4] + R[8]) - 1.0);
    const double ang = atan2(sn, cs);
    if (sn > 1e-8 || cs > 0.0) {
        const double k = sn > 1e-8 ? ang / sn : 1.0;
        rv[0] = sx * k; rv[1] = sy * k; rv[2] = sz * k;
        return;
    }
    /* angle ~ pi: the antisymmetric part vanishes, the axis comes from the symmetric part
     * (column i of R + e_i is parallel to the axis; i = the largest diagonal entry), as in
     * dbot_ros_amd/pose.py matrix_to_rotvec */
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = sqrt(fmax((R[4 * k] + 1.0) * 0.5, 0.0));
    const int i = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
    double a[3];
    for (int k = 0; k < 3; ++k) a[k] = (R[3 * k + i] + (k == i ? 1.0 : 0.0)) / (2.0 * d[i]);
    const double an = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    for (int k = 0; k < 3; ++k) rv[k] = a[k] / an * ang;
}

static void matmul3(const double* A, const double* B, double* C)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

/* The sensor's pose composition on its own: absolute pose = delta (+) default,
 *   R = R(delta rotation vector) R(default rotation vector),  t = t(delta) + t(default),
 * for n particles x parts bodies; deltas [n][parts][stride], deflt [parts][stride] (position, rotation vector first),
 * out [n][parts][12] = R row-major, t.  What RbSensor::loglikes does with the deltas the filter hands it
 * (R:source/dbot_ros/object_tracker_ros.hpp:49); the checker of rbs_loglikes_deltas' device composition. */
void orc_compose_poses(const double* deltas, const double* deflt, int32_t stride, int32_t n, int32_t parts, double* out)
{
    for (int32_t i = 0; i < n; ++i)
        for (int32_t b = 0; b < parts; ++b) {
            const double* d = deltas + ((size_t)i * parts + b) * stride;
            const double* d0 = deflt + (size_t)b * stride;
            double Rd[9], R0[9];
            double* o = out + ((size_t)i * parts + b) * 12;
            rotvec_to_matrix(d + 3, Rd);
            rotvec_to_matrix(d0 + 3, R0);
            matmul3(Rd, R0, o);
            for (int k = 0; k < 3; ++k) o[9 + k] = d[k] + d0[k];
        }
}

orc_tracker* orc_tracker_create(orc_sensor* s, int32_t parts, int32_t n, const double* sigma6, double vf,
                                double max_kl)
{
    orc_tracker* t = (orc_tracker*)calloc(1, sizeof(*t));
    t->s = s; t->n = n; t->parts = parts; t->D = parts * BODY;
    memcpy(t->sigma, sigma6, sizeof(double) * 6);
    t->vf = vf; t->max_kl = max_kl;
    t->old = (double*)calloc((size_t)n * t->D, sizeof(double));
    t->cur = (double*)calloc((size_t)n * t->D, sizeof(double));
    t->noise = (double*)calloc((size_t)n * parts * 6, sizeof(double));
    t->logw = (double*)calloc(n, sizeof(double));
    t->ll = (double*)calloc(n, sizeof(double));
    t->idx = (int32_t*)calloc(n, sizeof(int32_t));
    t->deflt = (double*)calloc(t->D, sizeof(double));
    return t;
}

void orc_tracker_destroy(orc_tracker* t)
{
    if (!t) return;
    free(t->old); free(t->cur); free(t->noise); free(t->logw); free(t->ll); free(t->idx); free(t->deflt);
    free(t);
}

/* tracker->initialize: the first initial state becomes the default pose, particles are zero
 * deltas, weights uniform, sensor reset (SURVEY A.1, A.4 last line). */
void orc_tracker_initialize(orc_tracker* t, const double* default_state)
{
    memcpy(t->deflt, default_state, sizeof(double) * t->D);
    memset(t->old, 0, sizeof(double) * (size_t)t->n * t->D);
    memset(t->noise, 0, sizeof(double) * (size_t)t->n * t->parts * 6);
    memset(t->logw, 0, sizeof(double) * t->n);
    memset(t->ll, 0, sizeof(double) * t->n);
    memset(t->idx, 0, sizeof(int32_t) * t->n);
    t->resamplings = 0;
    orc_reset(t->s);
}

/* One frame: per sampling block b restart from the old particles, apply the
 * transition of bodies 0..b with their accumulated noise, evaluate, update log-weights, test
 * KL(belief || uniform) against max_kl and resample multinomially (upper_bound of the cumulative
 * weights at host-supplied uniforms).  Then fold the weighted mean into the default pose. */
void orc_tracker_track(orc_tracker* t, const double* frame, const double* normals, const double* uniforms,
                       double* out_state, int32_t* out_resamplings)
{
    const int n = t->n, parts = t->parts, D = t->D;
    double* poses = (double*)malloc(sizeof(double) * (size_t)n * parts * 12);
    double* ll_new = (double*)malloc(sizeof(double) * n);
    double* w = (double*)malloc(sizeof(double) * n);
    double* tmpD = (double*)malloc(sizeof(double) * (size_t)n * D);
    double* tmpN = (double*)malloc(sizeof(double) * (size_t)n * parts * 6);
    int32_t* parents = (int32_t*)malloc(sizeof(int32_t) * n);
    orc_set_observation(t->s, frame);
    for (int b = 0; b < parts; ++b) {
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 6; ++k) t->noise[((size_t)i * parts + b) * 6 + k] = normals[((size_t)b * n + i) * 6 + k];
            for (int bb = 0; bb < parts; ++bb) {
                double s[BODY];
                memcpy(s, t->old + (size_t)i * D + bb * BODY, sizeof(s));
                if (bb <= b) {
                    const double* nz = t->noise + ((size_t)i * parts + bb) * 6;
                    for (int k = 0; k < 6; ++k) {
                        s[6 + k] = t->vf * s[6 + k] + t->sigma[k] * nz[k];
                        s[k] = s[k] + s[6 + k];
                    }
                }
                memcpy(t->cur + (size_t)i * D + bb * BODY, s, sizeof(s));
                double Rd[9], R0[9], R[9];
                rotvec_to_matrix(s + 3, Rd);
                rotvec_to_matrix(t->deflt + bb * BODY + 3, R0);
                matmul3(Rd, R0, R);
                double* out = poses + ((size_t)i * parts + bb) * 12;
                memcpy(out, R, sizeof(R));
                for (int k = 0; k < 3; ++k) out[9 + k] = s[k] + t->deflt[bb * BODY + k];
            }
        }
        const int last = b == parts - 1;
        int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * n);
        memcpy(idx, t->idx, sizeof(int32_t) * n);
        orc_loglikes(t->s, poses, idx, n, last, ll_new);
        if (last) memcpy(t->idx, idx, sizeof(int32_t) * n);
        free(idx);
        double m = -INFINITY;
        for (int i = 0; i < n; ++i) {
            t->logw[i] += ll_new[i] - t->ll[i];
            t->ll[i] = ll_new[i];
            if (t->logw[i] > m) m = t->logw[i];
        }
        double S = 0.0;
        for (int i = 0; i < n; ++i) { w[i] = exp(t->logw[i] - m); S += w[i]; }
        double kl = log((double)n);
        for (int i = 0; i < n; ++i) { w[i] /= S; if (w[i] > 0.0) kl += w[i] * log(w[i]); }
        if (kl > t->max_kl) {
            t->resamplings += 1;
            double run = 0.0, total = 0.0;
            for (int i = 0; i < n; ++i) total += w[i];
            /* cumulative weights, normalised by their last entry */
            double* c = (double*)malloc(sizeof(double) * n);
            for (int i = 0; i < n; ++i) { run += w[i]; c[i] = run / total; }
            for (int j = 0; j < n; ++j) {
                const double u = uniforms[(size_t)b * n + j];
                int lo = 0, hi = n;   /* first index with c > u == std::upper_bound */
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (c[mid] > u) hi = mid; else lo = mid + 1; }
                parents[j] = lo < n ? lo : n - 1;
            }
            free(c);
            /* children inherit particle, noise, likelihood and occlusion slot; weights reset */
            for (int j = 0; j < n; ++j) memcpy(tmpD + (size_t)j * D, t->old + (size_t)parents[j] * D, sizeof(double) * D);
            memcpy(t->old, tmpD, sizeof(double) * (size_t)n * D);
            for (int j = 0; j < n; ++j) memcpy(tmpD + (size_t)j * D, t->cur + (size_t)parents[j] * D, sizeof(double) * D);
            memcpy(t->cur, tmpD, sizeof(double) * (size_t)n * D);
            for (int j = 0; j < n; ++j) memcpy(tmpN + (size_t)j * parts * 6, t->noise + (size_t)parents[j] * parts * 6, sizeof(double) * parts * 6);
            memcpy(t->noise, tmpN, sizeof(double) * (size_t)n * parts * 6);
            for (int j = 0; j < n; ++j) { ll_new[j] = t->ll[parents[j]]; }
            memcpy(t->ll, ll_new, sizeof(double) * n);
            int32_t* ni = (int32_t*)malloc(sizeof(int32_t) * n);
            for (int j = 0; j < n; ++j) ni[j] = t->idx[parents[j]];
            memcpy(t->idx, ni, sizeof(int32_t) * n);
            free(ni);
            memset(t->logw, 0, sizeof(double) * n);
        }
    }
    /* weighted mean delta -> default pose; particles re-centred */
    double m = -INFINITY, S = 0.0;
    for (int i = 0; i < n; ++i) if (t->logw[i] > m) m = t->logw[i];
    for (int i = 0; i < n; ++i) { w[i] = exp(t->logw[i] - m); S += w[i]; }
    double* mean = (double*)calloc(D, sizeof(double));
    for (int d = 0; d < D; ++d) {
        double a = 0.0;
        for (int i = 0; i < n; ++i) a += (w[i] / S) * t->cur[(size_t)i * D + d];
        mean[d] = a;
    }
    for (int b = 0; b < parts; ++b) {
        double* z = t->deflt + b * BODY;
        const double* mu = mean + b * BODY;
        double Rm[9], Rz[9], R[9], RmT[9];
        rotvec_to_matrix(mu + 3, Rm);
        rotvec_to_matrix(z + 3, Rz);
        matmul3(Rm, Rz, R);
        for (int k = 0; k < 3; ++k) z[k] += mu[k];
        matrix_to_rotvec(R, z + 3);
        for (int k = 6; k < 12; ++k) z[k] = mu[k];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) RmT[3 * r + c] = Rm[3 * c + r];
        for (int i = 0; i < n; ++i) {
            double* p = t->cur + (size_t)i * D + b * BODY;
            for (int k = 0; k < 3; ++k) p[k] -= mu[k];
            double Rd[9], Rn[9];
            rotvec_to_matrix(p + 3, Rd);
            matmul3(Rd, RmT, Rn);
            matrix_to_rotvec(Rn, p + 3);
        }
    }
    memcpy(t->old, t->cur, sizeof(double) * (size_t)n * D);
    memcpy(out_state, t->deflt, sizeof(double) * D);
    if (out_resamplings) *out_resamplings = t->resamplings;
    free(mean); free(poses); free(ll_new); free(w); free(tmpD); free(tmpN); free(parents);
}

void orc_tracker_get(const orc_tracker* t, double* particles, double* log_weights, int32_t* indices)
{
    if (particles) memcpy(particles, t->old, sizeof(double) * (size_t)t->n * t->D);
    if (log_weights) memcpy(log_weights, t->logw, sizeof(double) * t->n);
    if (indices) memcpy(indices, t->idx, sizeof(int32_t) * t->n);
}
