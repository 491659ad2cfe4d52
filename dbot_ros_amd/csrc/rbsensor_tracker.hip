// rbsensor_tracker.hip -- device-side callers of the hot path ("next" rows): the
// object state transition, the Rao-Blackwellised coordinate particle filter step (log-weight
// update, KL test, multinomial resampling, gather) and the tracker's weighted mean + re-centring,
// as the reference's filter drives them once per sampling block inside tracker_->track(image)
// (R:source/dbot_ros/object_tracker_ros.hpp:49; parameters
// R:source/dbot_ros/tracker/particle_tracker_node.cpp:138-159,208-218).  Semantics follow
// SURVEY.md Appendix A.1/A.6 and are mirrored on the host by dbot_ros_amd/tracker.py, which the
// parity tests compare against.  Everything is stream-ordered: one host synchronisation per frame.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace rbt {

constexpr int kBody = 12;   // position(3) rotation-vector(3) linear-velocity(3) angular-velocity(3)

struct TrackerDev {
    int n, parts, D;              // particles, bodies, state length = parts*12
    double sigma[6];              // linear xyz, angular xyz
    double vf;                    // velocity_factor
    double max_kl;
    double* part_old; double* part_new; double* noise;      // [n][D], [n][D], [n][parts][6]
    double* part_old2; double* part_new2; double* noise2;   // gather targets (ping-pong)
    double* logw; double* ll; double* ll2; double* ll_new;  // [n]
    int* idx; int* idx2; int* parents;                       // [n]
    double* cdf;                  // [n]
    double* deflt;                // [D] integrated (default) pose the deltas live around
    double* mean;                 // [D] + [parts][9] transposed mean rotations
    double* poses;                // [n][parts][12] absolute R|t handed to the sensor
    int* flag;                    // [0] resample decision of the last block, [1] resampling count
    const double* normals;        // host-supplied [parts][n][6] or nullptr (device RNG)
    const double* uniforms;       // host-supplied [parts][n] or nullptr
    unsigned long long seed, frame;
    // several devices: every device holds ALL particle states and runs every kernel of this file
    // redundantly (identical inputs, identical code: identical results); only the sensor call is
    // sharded.  layout[g] = particle evaluated at global slot g (device g / cap); *_sorted are the
    // sensor's inputs / outputs in slot order.
    double* red;                  // [kRedBlocks * (3 + D)] per-block partials of the multi-block filter kernels
    int* layout;                  // [n]
    double* ll_sorted;            // [n_dev * cap]
    double* poses_sorted;         // [cap][parts][12]   (this device's shard)
    int* idx_sorted;              // [cap]
    // pinned host memory the kernel that finishes the frame writes the estimate ([D]) and the flags
    // ([2]) into, or nullptr: no device-to-host copies after the last kernel
    double* host_state;
    int* host_flags;
};

// ------------------------------------------------------------------ rotations
// rotation vector -> row-major matrix through the unit quaternion; same formula as
// dbot_ros_amd/pose.py rotvec_to_matrix.
__device__ inline void rotvec_to_matrix(const double* rv, double* R)
{
    const double angle = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    const double half = 0.5 * angle;
    const double k = angle < 1e-9 ? 0.5 - angle * angle / 48.0 : sin(half) / angle;
    const double w = cos(half), x = rv[0] * k, y = rv[1] * k, z = rv[2] * k;
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y - w * z); R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z); R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y); R[7] = 2.0 * (y * z + w * x); R[8] = 1.0 - 2.0 * (x * x + y * y);
}

// matrix -> rotation vector, atan2 form (dbot_ros_amd/tracker.py _rotvecs / pose.py matrix_to_rotvec)
__device__ inline void matrix_to_rotvec(const double* R, double* rv)
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
    // angle ~ pi: the antisymmetric part vanishes, the axis comes from the symmetric part
    // (column i of R + e_i is parallel to the axis; i = the largest diagonal entry), as in
    // dbot_ros_amd/pose.py matrix_to_rotvec
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = sqrt(fmax((R[4 * k] + 1.0) * 0.5, 0.0));
    const int i = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
    double a[3];
    for (int k = 0; k < 3; ++k) a[k] = (R[3 * k + i] + (k == i ? 1.0 : 0.0)) / (2.0 * d[i]);
    const double an = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    for (int k = 0; k < 3; ++k) rv[k] = a[k] / an * ang;
}

__device__ inline void matmul3(const double* A, const double* B, double* C)
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// ------------------------------------------------------------------ random numbers
// Philox4x32-10 counter-based generator: (seed, frame, stream, index) -> 4 x 32 random bits.
__device__ inline uint4 philox(unsigned long long seed, unsigned long long ctr_hi, unsigned long long ctr_lo)
{
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    unsigned c0 = (unsigned)ctr_lo, c1 = (unsigned)(ctr_lo >> 32), c2 = (unsigned)ctr_hi, c3 = (unsigned)(ctr_hi >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
__device__ inline double u01(unsigned hi, unsigned lo)   // 53-bit uniform in [0,1)
{
    return (double)((((unsigned long long)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------ f2: transition + pose composition
// One thread per particle, sampling block b: restart from the OLD particle, apply the
// transition of bodies 0..b with their accumulated noise (vel' = vf vel + sigma o n,
// pose' = pose + vel'), write the new particle and its absolute poses
// R = R(delta) R(default), t = t(delta) + t(default)   (SURVEY A.1).
__device__ inline void recentre_body(const TrackerDev& T, double* __restrict__ p, int bb);

__device__ inline void propagate_body(const TrackerDev& T, int b, int i, int bb, bool recentre)
{
    double s[kBody];
    if (recentre) recentre_body(T, T.part_old + (size_t)i * T.D + bb * kBody, bb);   // last frame's re-centring, deferred into this launch
#pragma unroll
    for (int k = 0; k < kBody; ++k) s[k] = T.part_old[(size_t)i * T.D + bb * kBody + k];
    if (bb <= b) {
        double nz[6];
        if (bb == b) {
            if (T.normals) {
#pragma unroll
                for (int k = 0; k < 6; ++k) nz[k] = T.normals[((size_t)b * T.n + i) * 6 + k];
            } else {  // Box-Muller on Philox uniforms: 3 pairs
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
                    const uint4 r = philox(T.seed, (T.frame << 8) | (unsigned)b, ((unsigned long long)i << 2) | pr);
                    const double u1 = 1.0 - u01(r.x, r.y), u2 = u01(r.z, r.w);
                    const double rad = sqrt(-2.0 * log(u1));
                    nz[2 * pr] = rad * cos(6.283185307179586 * u2);
                    nz[2 * pr + 1] = rad * sin(6.283185307179586 * u2);
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) T.noise[((size_t)i * T.parts + bb) * 6 + k] = nz[k];
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) nz[k] = T.noise[((size_t)i * T.parts + bb) * 6 + k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            s[6 + k] = T.vf * s[6 + k] + T.sigma[k] * nz[k];
            s[k] = s[k] + s[6 + k];
        }
    }
#pragma unroll
    for (int k = 0; k < kBody; ++k) T.part_new[(size_t)i * T.D + bb * kBody + k] = s[k];
    double Rd[9], R0[9], R[9];
    rotvec_to_matrix(s + 3, Rd);
    rotvec_to_matrix(T.deflt + bb * kBody + 3, R0);
    matmul3(Rd, R0, R);
    double* out = T.poses + ((size_t)i * T.parts + bb) * 12;
#pragma unroll
    for (int k = 0; k < 9; ++k) out[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) out[9 + k] = s[k] + T.deflt[bb * kBody + k];
}

// recentre: the re-centring of the previous frame's particles (recentre_kernel's work) has been
// deferred into this launch -- one launch and one kernel boundary less per frame; the thread owns
// particle i, so it re-centres the old particle in place first (same operations, same bits).
__global__ __launch_bounds__(256) void propagate_kernel(const TrackerDev T, int b, int recentre)   // (launched with 256: without the bound the
                                                                                              //  compiler budgets for 1 024 and spills 20 registers)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T.n) return;
    for (int bb = 0; bb < T.parts; ++bb) propagate_body(T, b, i, bb, recentre != 0);
}

// ------------------------------------------------------------------ f1: weights, KL, resampling
__device__ inline double block_sum(double v, double* sh)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += sh[k];
    return s;
}
__device__ inline double block_max(double v, double* sh)
{
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double s = -INFINITY;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s = fmax(s, sh[k]);
    return s;
}

// Single block (1024 threads): log_w += new_ll - ll; ll = new_ll; w = softmax(log_w);
// KL(w || uniform) = log n + sum w log w; if KL > max_kl: cdf = cumsum(w)/sum, flag = 1.
// `updated`: the sensor call wrote slot i for particle i, so the slot map becomes identity.
__device__ inline void weights_body(const TrackerDev& T, int updated, double* sh)
{
    const int n = T.n;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    double m = -INFINITY;
    for (int i = lo; i < hi; ++i) {
        const double nl = T.ll_new[i];
        const double lw = T.logw[i] + (nl - T.ll[i]);
        T.logw[i] = lw;
        T.ll[i] = nl;
        if (updated) T.idx[i] = i;
        m = fmax(m, lw);
    }
    m = block_max(m, sh);
    double s = 0.0;
    for (int i = lo; i < hi; ++i) s += exp(T.logw[i] - m);
    const double local = s;
    const double S = block_sum(s, sh);
    double e = 0.0;
    for (int i = lo; i < hi; ++i) {
        const double w = exp(T.logw[i] - m) / S;
        if (w > 0.0) e += w * log(w);
    }
    const double kl = log((double)n) + block_sum(e, sh);
    const bool resample = kl > T.max_kl;
    if (threadIdx.x == 0) {
        T.flag[0] = resample ? 1 : 0;
        if (resample) T.flag[1] += 1;
    }
    if (!resample) return;   // block-uniform
    // scan of the per-thread sums -> exclusive offsets, then the running cdf: inside a wave by
    // shuffles, across the sixteen waves through LDS (two barriers; a Hillis-Steele scan over the
    // 1 024 threads took twenty)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double v = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    const double below = __shfl_up(v, 1, 64);   // the wave's sum up to the previous lane
    __syncthreads();
    if (lane == 63) sh[wv] = v;
    __syncthreads();
    double prefix = 0.0, total = 0.0;
    for (int k = 0; k < 16; ++k) {
        if (k < wv) prefix += sh[k];
        total += sh[k];
    }
    double run = prefix + (lane ? below : 0.0);
    for (int i = lo; i < hi; ++i) {
        run += exp(T.logw[i] - m);
        T.cdf[i] = run / total;
    }
}

__global__ __launch_bounds__(1024) void weights_kernel(const TrackerDev T, int updated)
{
    __shared__ double sh[1024];
    weights_body(T, updated, sh);
}

// ---- the same step for many particles: a single 1 024-thread block walks 200 000 particles in
// 2.6 ms (measured 1.3 ms at 100 000), more than an eight-way sharded sensor call takes.  Four
// grid launches over chunks of kChunk particles, every reduction in a fixed order (per-thread
// chunk order, shuffle tree, block order), so every device of a sharded tracker -- and every
// run -- computes the same bits:
//   w1  log_w += new_ll - ll; ll = new_ll; per-block max                       -> red[blk]
//   w2  m = max over blocks; e_i = exp(log_w_i - m) -> cdf[i]; per-block sum e, sum e (log_w - m)
//   w3  (one block) S, KL = log n + sum(e (log_w - m))/S - log S, flag, block offsets
//   w4  flag: cdf[i] = (offset[blk] + inclusive scan of e inside the block) / S
constexpr int kChunk = 4096;          // particles per block (1 024 threads x 4)
constexpr int kRedBlocks = 1024;      // at most this many chunks: 4 194 304 particles
constexpr int kMultiBlockFrom = 8192; // below it the single-block kernels (the parity tests' path) are faster

__global__ __launch_bounds__(1024) void weights_w1_kernel(const TrackerDev T, int updated)
{
    __shared__ double sh[16];
    const int base = blockIdx.x * kChunk;
    double m = -INFINITY;
    for (int k = 0; k < 4; ++k) {
        const int i = base + k * 1024 + (int)threadIdx.x;
        if (i < T.n) {
            const double nl = T.ll_new[i];
            const double lw = T.logw[i] + (nl - T.ll[i]);
            T.logw[i] = lw;
            T.ll[i] = nl;
            if (updated) T.idx[i] = i;
            m = fmax(m, lw);
        }
    }
    m = block_max(m, sh);
    if (threadIdx.x == 0) T.red[blockIdx.x] = m;
}

__global__ __launch_bounds__(1024) void weights_w2_kernel(const TrackerDev T)
{
    __shared__ double sh[16];
    double m = -INFINITY;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 1024) m = fmax(m, T.red[b]);
    m = block_max(m, sh);
    const int base = blockIdx.x * kChunk;
    double se = 0.0, sl = 0.0;
    for (int k = 0; k < 4; ++k) {
        const int i = base + k * 1024 + (int)threadIdx.x;
        if (i < T.n) {
            const double d = T.logw[i] - m;
            const double e = exp(d);
            T.cdf[i] = e;
            se += e;
            if (e > 0.0) sl += e * d;
        }
    }
    se = block_sum(se, sh);
    sl = block_sum(sl, sh);
    if (threadIdx.x == 0) {
        T.red[kRedBlocks + blockIdx.x] = se;
        T.red[2 * kRedBlocks + blockIdx.x] = sl;
    }
}

__global__ __launch_bounds__(1024) void weights_w3_kernel(const TrackerDev T, int blocks)
{
    if (threadIdx.x != 0) return;       // <= 1 024 partials: a serial, order-fixed sum
    double S = 0.0, L = 0.0;
    for (int b = 0; b < blocks; ++b) {
        const double se = T.red[kRedBlocks + b];
        T.red[kRedBlocks + b] = S;       // exclusive offset of the block
        S += se;
        L += T.red[2 * kRedBlocks + b];
    }
    const double kl = log((double)T.n) + L / S - log(S);
    const bool resample = kl > T.max_kl;
    T.flag[0] = resample ? 1 : 0;
    if (resample) T.flag[1] += 1;
    T.red[0] = S;
}

__global__ __launch_bounds__(1024) void weights_w4_kernel(const TrackerDev T)
{
    if (!T.flag[0]) return;
    __shared__ double sh[1024];
    const int base = blockIdx.x * kChunk + (int)threadIdx.x * 4;   // four consecutive particles per thread
    double e[4], run = 0.0;
    for (int k = 0; k < 4; ++k) { e[k] = base + k < T.n ? T.cdf[base + k] : 0.0; run += e[k]; }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const double v = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    double acc = T.red[kRedBlocks + blockIdx.x] + (threadIdx.x ? sh[threadIdx.x - 1] : 0.0);
    const double S = T.red[0];
    for (int k = 0; k < 4; ++k) {
        acc += e[k];
        if (base + k < T.n) T.cdf[base + k] = acc / S;
    }
}

// The frame's result for the host (called by a whole block after T.deflt is complete).
__device__ inline void publish_result(const TrackerDev& T)
{
    if (!T.host_state) return;
    __threadfence_block();
    __syncthreads();
    for (int k = threadIdx.x; k < T.D; k += blockDim.x) T.host_state[k] = T.deflt[k];
    if (threadIdx.x < 2) T.host_flags[threadIdx.x] = T.flag[threadIdx.x];
    // ... and the frame's number behind them: the host may take the estimate as soon as it sees the
    // number, without waiting for this kernel's completion to be signalled.  The block's stores are
    // ordered before thread 0 by the barrier (workgroup scope) and before the number by thread 0's
    // system-scope release (cumulative) -- ONE wave writes the L2 back, not all sixteen (a system
    // fence per thread here cost the kernel several microseconds behind a raster kernel's dirty L2)
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(T.host_flags + 2, (int)(T.frame + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- weighted mean for many particles: per-block partial sums of e_i * particle_i, then one
// block folds them into the default pose (mean_body's tail)
__global__ __launch_bounds__(1024) void mean_m1_kernel(const TrackerDev T)
{
    __shared__ double sh[16];
    double m = -INFINITY;
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * kChunk + k * 1024 + (int)threadIdx.x;
        if (i < T.n) m = fmax(m, T.logw[i]);
    }
    m = block_max(m, sh);
    if (threadIdx.x == 0) T.red[blockIdx.x] = m;
}

__global__ __launch_bounds__(1024) void mean_m2_kernel(const TrackerDev T)
{
    __shared__ double sh[16];
    __shared__ double shb[16][kBody];
    double m = -INFINITY;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 1024) m = fmax(m, T.red[b]);
    m = block_max(m, sh);
    double se = 0.0;
    double e[4];
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * kChunk + k * 1024 + (int)threadIdx.x;
        e[k] = i < T.n ? exp(T.logw[i] - m) : 0.0;
        se += e[k];
    }
    se = block_sum(se, sh);
    double* out = T.red + 3 * kRedBlocks + (size_t)blockIdx.x * T.D;
    if (threadIdx.x == 0) T.red[kRedBlocks + blockIdx.x] = se;
    for (int b = 0; b < T.parts; ++b) {
        double a[kBody];
#pragma unroll
        for (int c = 0; c < kBody; ++c) a[c] = 0.0;
        for (int k = 0; k < 4; ++k) {
            const int i = blockIdx.x * kChunk + k * 1024 + (int)threadIdx.x;
            if (i < T.n) {
                const double* p = T.part_new + (size_t)i * T.D + b * kBody;
#pragma unroll
                for (int c = 0; c < kBody; ++c) a[c] += e[k] * p[c];
            }
        }
#pragma unroll
        for (int c = 0; c < kBody; ++c)
            for (int off = 32; off > 0; off >>= 1) a[c] += __shfl_down(a[c], off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int c = 0; c < kBody; ++c) shb[threadIdx.x >> 6][c] = a[c];
        __syncthreads();
        if ((int)threadIdx.x < kBody) {
            double sum = 0.0;
            for (int w = 0; w < 16; ++w) sum += shb[w][threadIdx.x];
            out[b * kBody + threadIdx.x] = sum;
        }
    }
}

__global__ __launch_bounds__(64) void mean_m3_kernel(const TrackerDev T, int blocks)
{
    // component c of the mean = (sum over blocks, in block order) / S
    double S = 0.0;
    for (int b = 0; b < blocks; ++b) S += T.red[kRedBlocks + b];
    for (int c = threadIdx.x; c < T.D; c += 64) {
        double sum = 0.0;
        for (int b = 0; b < blocks; ++b) sum += T.red[3 * kRedBlocks + (size_t)b * T.D + c];
        T.mean[c] = sum / S;
    }
    __syncthreads();
    if ((int)threadIdx.x < T.parts) {
        const int b = threadIdx.x;
        double* z = T.deflt + b * kBody;
        const double* mu = T.mean + b * kBody;
        double Rm[9], Rz[9], R[9];
        rotvec_to_matrix(mu + 3, Rm);
        rotvec_to_matrix(z + 3, Rz);
        matmul3(Rm, Rz, R);
        for (int k = 0; k < 3; ++k) z[k] += mu[k];
        matrix_to_rotvec(R, z + 3);
        for (int k = 6; k < 12; ++k) z[k] = mu[k];
        double* RmT = T.mean + T.D + b * 9;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) RmT[3 * r + c] = Rm[3 * c + r];
    }
    publish_result(T);
}

// parents[j] = flag ? upper_bound(cdf, u_j) : j   (multinomial resampling)
__device__ inline void resample_one(const TrackerDev& T, int b, int j)
{
    int p = j;
    if (T.flag[0]) {
        double u;
        if (T.uniforms) u = T.uniforms[(size_t)b * T.n + j];
        else {
            const uint4 r = philox(T.seed ^ 0x5bd1e995ull, (T.frame << 8) | (unsigned)b, (unsigned long long)j);
            u = u01(r.x, r.y);
        }
        int lo = 0, hi = T.n;   // first index with cdf > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (T.cdf[mid] > u) hi = mid; else lo = mid + 1;
        }
        p = min(lo, T.n - 1);
    }
    T.parents[j] = p;
}

__global__ void resample_kernel(const TrackerDev T, int b)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < T.n) resample_one(T, b, j);
}

// children inherit particle, noise, likelihood and occlusion slot of their parent; weights reset
__device__ inline void gather_one(const TrackerDev& T, int j, int k0, int kstep)
{
    const int p = T.parents[j];
    for (int k = k0; k < T.D; k += kstep) {
        T.part_old2[(size_t)j * T.D + k] = T.part_old[(size_t)p * T.D + k];
        T.part_new2[(size_t)j * T.D + k] = T.part_new[(size_t)p * T.D + k];
    }
    for (int k = k0; k < T.parts * 6; k += kstep)
        T.noise2[(size_t)j * T.parts * 6 + k] = T.noise[(size_t)p * T.parts * 6 + k];
    if (k0 == 0) {
        T.ll2[j] = T.ll[p];
        T.idx2[j] = T.idx[p];
        if (T.flag[0]) T.logw[j] = 0.0;   // every child writes its own weight slot only
    }
}

__global__ void gather_kernel(const TrackerDev T)
{
    gather_one(T, (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x);
}

// Resampling and gather in one launch: the block that copies child j draws its parent first.
__global__ __launch_bounds__(64) void resample_gather_kernel(const TrackerDev T, int b)
{
    if (threadIdx.x == 0) resample_one(T, b, (int)blockIdx.x);
    __threadfence_block();
    __syncthreads();
    gather_one(T, (int)blockIdx.x, (int)threadIdx.x, 64);
}

// ------------------------------------------------------------------ tracker: mean + re-centring
// Single block: mean = sum_i softmax(log_w)_i * particle_i; fold it into the default pose.
// via != nullptr: particle i is part_new[via[i]] -- the mean of the RESAMPLED particles straight from
// the parents' rows, before (instead of after) the gather: the same values in the same order.
__device__ inline void mean_body(const TrackerDev& T, const double* __restrict__ part_new, double* sh,
                                 double (*shb)[kBody], const int* __restrict__ via = nullptr)
{
    const int n = T.n;
    // After a resampling every log-weight is 0 (gather_one): max 0, exp(0) = 1, their sum n -- the
    // same bits as the general route below, without its two block reductions and 2 n exponentials.
    const bool uniform = T.flag[0] != 0;   // block-uniform
    double m = 0.0, S = (double)n;
    if (!uniform) {
        m = -INFINITY;
        for (int i = threadIdx.x; i < n; i += 1024) m = fmax(m, T.logw[i]);
        m = block_max(m, sh);
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 1024) s += exp(T.logw[i] - m);
        S = block_sum(s, sh);
    }
    // the twelve components of a body are reduced together: each one's additions are exactly
    // those of block_sum (same per-thread order, same shuffle tree, same order over the waves),
    // with two barriers per body instead of twenty-four
    for (int b = 0; b < T.parts; ++b) {
        double a[kBody];
#pragma unroll
        for (int k = 0; k < kBody; ++k) a[k] = 0.0;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const double w = (uniform ? 1.0 : exp(T.logw[i] - m)) / S;
            const double* p = part_new + (size_t)(via ? via[i] : i) * T.D + b * kBody;
#pragma unroll
            for (int k = 0; k < kBody; ++k) a[k] += w * p[k];
        }
#pragma unroll
        for (int k = 0; k < kBody; ++k)
            for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int k = 0; k < kBody; ++k) shb[threadIdx.x >> 6][k] = a[k];
        __syncthreads();
        if ((int)threadIdx.x < kBody) {
            double sum = 0.0;
            for (int w = 0; w < 16; ++w) sum += shb[w][threadIdx.x];
            T.mean[b * kBody + threadIdx.x] = sum;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < T.parts) {
        const int b = threadIdx.x;
        double* z = T.deflt + b * kBody;
        const double* mu = T.mean + b * kBody;
        double Rm[9], Rz[9], R[9];
        rotvec_to_matrix(mu + 3, Rm);
        rotvec_to_matrix(z + 3, Rz);
        matmul3(Rm, Rz, R);
        for (int k = 0; k < 3; ++k) z[k] += mu[k];
        matrix_to_rotvec(R, z + 3);
        for (int k = 6; k < 12; ++k) z[k] = mu[k];
        double* RmT = T.mean + T.D + b * 9;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) RmT[3 * r + c] = Rm[3 * c + r];
    }
    publish_result(T);
}

__global__ __launch_bounds__(1024) void mean_kernel(const TrackerDev T)
{
    __shared__ double sh[1024];
    __shared__ double shb[16][kBody];
    mean_body(T, T.part_new, sh, shb);
}

// delta_i <- delta_i (-) mean:  t -= t_mean,  R(delta_i) <- R(delta_i) R(mean)^T
__device__ inline void recentre_body(const TrackerDev& T, double* __restrict__ p, int b)
{
    for (int k = 0; k < 3; ++k) p[k] -= T.mean[b * kBody + k];
    double Rd[9], R[9];
    rotvec_to_matrix(p + 3, Rd);
    matmul3(Rd, T.mean + T.D + b * 9, R);
    matrix_to_rotvec(R, p + 3);
}
__device__ inline void recentre_one(const TrackerDev& T, double* __restrict__ part_new, int i)
{
    for (int b = 0; b < T.parts; ++b) recentre_body(T, part_new + (size_t)i * T.D + b * kBody, b);
}

__global__ __launch_bounds__(256) void recentre_kernel(const TrackerDev T, double* __restrict__ particles)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T.n) recentre_one(T, particles, i);
}

// The LAST sampling block's filter step up to the frame's estimate in ONE single-block launch:
// weights + KL test, resampling (parents only), the weighted mean of the resampled particles read
// through their parents' rows, the new default pose, and the result stored into the caller's
// pinned slot.  The gather -- which the estimate does not need -- follows as its own launch, behind
// the event the host waits for: the host wakes up and stages the next frame while it runs.  Same
// operations in the same order as weights_kernel / resample_gather_kernel / mean_kernel (three
// launches, 32 us at 2 000 particles; this one 20).
__global__ __launch_bounds__(1024) void filter_tail_kernel(const TrackerDev T, int b, int updated)
{
    __shared__ double sh[1024];
    __shared__ double shb[16][kBody];
    weights_body(T, updated, sh);
    __threadfence_block();
    __syncthreads();
    for (int j = threadIdx.x; j < T.n; j += 1024) resample_one(T, b, j);
    __threadfence_block();
    __syncthreads();
    mean_body(T, T.part_new, sh, shb, T.parents);
}

// Few particles: the whole filter step after the sensor call -- weights and KL test, resampling,
// gather, and after the last sampling block the mean and the re-centring -- in ONE single-block
// launch instead of five.  Same code, same order of additions as the separate kernels; the
// phases are separated by block barriers (every phase reads what the previous one wrote to
// global memory from other threads of this block).
constexpr int kFusedFilterMax = 512;    // measured: +4 % frames/s at 200 particles, -8 % at 2 000
__global__ __launch_bounds__(1024) void filter_step_kernel(const TrackerDev T, int b, int updated, int last)
{
    __shared__ double sh[1024];
    __shared__ double shb[16][kBody];
    weights_body(T, updated, sh);
    __threadfence_block();
    __syncthreads();
    for (int j = threadIdx.x; j < T.n; j += 1024) resample_one(T, b, j);
    __threadfence_block();
    __syncthreads();
    // 16 threads per particle walk its components, as 16 of the gather kernel's 64 would
    for (int j = threadIdx.x >> 4; j < T.n; j += 64) gather_one(T, j, threadIdx.x & 15, 16);
    if (!last) return;
    __threadfence_block();
    __syncthreads();
    mean_body(T, T.part_new2, sh, shb);      // the gathered particles: the host swaps the buffers afterwards
    __threadfence_block();
    __syncthreads();
    for (int i = threadIdx.x; i < T.n; i += 1024) recentre_one(T, T.part_new2, i);
}

// ------------------------------------------------------------------ several devices: slot layout
// Particles are laid out over the devices' slots by the device that holds the plane they inherit
// (T.idx[j] / cap): a stable n_dev-way partition of the particle ids, bucket after bucket, cut
// into the devices' fixed slot ranges [k cap, (k+1) cap).  A particle whose bucket spills over a
// boundary is evaluated next door and reads its parent's window over xGMI; typically (a few
// survivors with many children each) almost every particle stays with its parent.  Single block,
// deterministic: every device computes the same layout.
constexpr int kMaxDev = 8;
__global__ __launch_bounds__(1024) void layout_kernel(const TrackerDev T, int n_dev, int cap)
{
    __shared__ int cnt[kMaxDev][1024 + 1];
    __shared__ int bucket_start[kMaxDev + 1];
    const int n = T.n, t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    int c[kMaxDev];
#pragma unroll
    for (int b = 0; b < kMaxDev; ++b) c[b] = 0;
    for (int j = lo; j < hi; ++j) {
        const int b = min(max(T.idx[j] / cap, 0), n_dev - 1);
#pragma unroll
        for (int k = 0; k < kMaxDev; ++k) c[k] += (k == b);
    }
#pragma unroll
    for (int b = 0; b < kMaxDev; ++b) cnt[b][t] = c[b];
    __syncthreads();
    // exclusive scan over the threads, one wave per bucket pair (n_dev <= 8, 1 024 threads: cheap)
    if (t < kMaxDev) {
        int run = 0;
        for (int k = 0; k < 1024; ++k) { const int v = cnt[t][k]; cnt[t][k] = run; run += v; }
        cnt[t][1024] = run;
    }
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int b = 0; b < kMaxDev; ++b) { bucket_start[b] = run; run += cnt[b][1024]; }
        bucket_start[kMaxDev] = run;
    }
    __syncthreads();
    int pos[kMaxDev];
#pragma unroll
    for (int b = 0; b < kMaxDev; ++b) pos[b] = bucket_start[b] + cnt[b][t];
    for (int j = lo; j < hi; ++j) {
        const int b = min(max(T.idx[j] / cap, 0), n_dev - 1);
        int p = 0;
#pragma unroll
        for (int k = 0; k < kMaxDev; ++k) if (k == b) { p = pos[k]; pos[k] += 1; }
        T.layout[p] = j;
    }
}

// this device's shard of the sensor call: poses and parent slots of the particles at slots lo..lo+cnt
__global__ void shard_gather_kernel(const TrackerDev T, int lo, int cnt)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= cnt) return;
    const int j = T.layout[lo + g];
    const int w = T.parts * 12;
    for (int k = 0; k < w; ++k) T.poses_sorted[(size_t)g * w + k] = T.poses[(size_t)j * w + k];
    T.idx_sorted[g] = T.idx[j];
}

// after the all-gather: log-likelihoods back to particle order; an updating call wrote particle
// layout[g]'s plane to global slot g
__global__ void shard_scatter_kernel(const TrackerDev T, int updated)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= T.n) return;
    const int j = T.layout[g];
    T.ll_new[j] = T.ll_sorted[g];
    if (updated) T.idx[j] = g;
}

__global__ void init_kernel(const TrackerDev T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T.n) { T.logw[i] = 0.0; T.ll[i] = 0.0; T.idx[i] = 0; }
    if (i < 2) T.flag[i] = 0;
    for (size_t k = i; k < (size_t)T.n * T.D; k += (size_t)gridDim.x * blockDim.x) T.part_old[k] = 0.0;
    for (size_t k = i; k < (size_t)T.n * T.parts * 6; k += (size_t)gridDim.x * blockDim.x) T.noise[k] = 0.0;
}

// The default pose of every body as particle 0's absolute pose (zero delta): rbs_tracker_initialize's
// probe call that sizes window-sized slabs before the first frame.
__global__ void default_pose_kernel(const TrackerDev T)
{
    const int bb = threadIdx.x;
    if (bb >= T.parts) return;
    double R0[9];
    rotvec_to_matrix(T.deflt + bb * kBody + 3, R0);
    double* out = T.poses + (size_t)bb * 12;
    for (int k = 0; k < 9; ++k) out[k] = R0[k];
    for (int k = 0; k < 3; ++k) out[9 + k] = T.deflt[bb * kBody + k];
}

}  // namespace rbt
