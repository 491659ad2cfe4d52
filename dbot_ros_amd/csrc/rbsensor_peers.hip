// rbsensor_peers.hip -- the resampling half of the filter step across PROCESSES (one rank per GPU,
// SURVEY 8(e); dbot_ros_amd/dist.py PeerShardedStep): everything between the all-gather of the
// log-likelihoods and the next step's parent indices in ONE launch.
//
// Every rank holds the gathered log-likelihoods of all N = world * n particles and the same N uniforms,
// SORTED ascending (children are exchangeable: sorting the uniforms is sorting the children by parent,
// because u -> upper_bound(cdf, u) is monotone).  Child g's parent is upper_bound(cdf, u[g]) (SURVEY A.6:
// multinomial resampling at host-supplied uniforms); rank r evaluates children [r n, (r + 1) n) -- in
// parent order the children of rank r's particles are one contiguous run that mostly coincides with
// those slots, so most children find their parent's plane on their own GPU.  What the kernel leaves
// behind is rank r's plan (the arithmetic of dist.global_resample + dist.plan_shard, which runs the same
// step as ~45 small tensor kernels):
//   parent_idx[k]   the GLOBAL slot child k inherits from: owner * cap + local slot, or -- a parent on
//                   another rank that >= min_share of this rank's children share -- the local staging
//                   slot rank * cap + n + j its window is pulled into once (rbs_stage_windows)
//   stage_src / stage_dst[k]   entry k staged iff stage_dst[k] >= 0
//   counts[4] += children with a remote parent, of them served from staging, windows staged, distinct
//                   parents among this rank's children
// Deterministic (fixed summation order, no floating-point atomics): every rank computes the same cdf
// bit for bit, which is what makes the ranks' plans consistent without another exchange.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace rbp {

constexpr int kThreads = 1024;

struct PeerPlan {
    const double* ll_all;     // [N] gathered log-likelihoods, rank-major
    const double* uniforms;   // [N] ascending
    int N, n, rank, cap, min_share;
    double temperature;
    double* cdf;              // [N] scratch
    int* mine;                // [n] scratch: this rank's children's parents (indices into ll_all)
    int* aux;                 // [n] scratch
    int32_t* parent_idx;      // [n]
    int32_t* stage_src;       // [n]
    int32_t* stage_dst;       // [n]
    int32_t* parents_local;   // [n] or null: a copy of `mine`
    long long* counts;        // [4], accumulated
};

struct OpAdd { template <class T> __device__ T operator()(T a, T b) const { return a + b; } };
struct OpMax { template <class T> __device__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct OpMin { template <class T> __device__ T operator()(T a, T b) const { return a < b ? a : b; } };

// Exclusive scan of one value per thread in thread order (wave scans by shuffles, the sixteen wave
// totals combined in order); *total = the reduction over all threads.  sh holds >= kThreads / 64 values.
template <class T, class Op>
__device__ inline T block_exscan(T v, T ident, Op op, T* sh, T* total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    T inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(inc, off, 64);
        if (lane >= off) inc = op(o, inc);
    }
    __syncthreads();
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    T base = ident, tot = ident;
    for (int k = 0; k < nw; ++k) {
        if (k == w) base = tot;
        tot = op(tot, sh[k]);
    }
    if (total) *total = tot;
    T exc = __shfl_up(inc, 1, 64);
    if (lane == 0) exc = ident;
    return op(base, exc);
}

__global__ __launch_bounds__(kThreads) void peer_resample_kernel(const PeerPlan Q)
{
    __shared__ double shd[kThreads / 64];
    __shared__ int shi[kThreads / 64];
    const int t = (int)threadIdx.x, N = Q.N, n = Q.n;

    // ---- weights exp((ll - max) / T), their cumulative sums, normalised: the cdf
    double m = -INFINITY;
    for (int i = t; i < N; i += kThreads) m = fmax(m, Q.ll_all[i]);      // (fmax drops NaN: a contained particle weighs nothing)
    {
        double tot;
        (void)block_exscan(m, -(double)INFINITY, OpMax(), shd, &tot);
        m = tot;
    }
    const int L = (N + kThreads - 1) / kThreads;
    const int lo = min(N, t * L), hi = min(N, lo + L);
    double run = 0.0;
    for (int i = lo; i < hi; ++i) {
        const double l = Q.ll_all[i];
        const double w = l == l ? exp((l - m) / Q.temperature) : 0.0;
        run += w;
        Q.cdf[i] = run;
    }
    double total;
    const double before = block_exscan(run, 0.0, OpAdd(), shd, &total);
    for (int i = lo; i < hi; ++i) Q.cdf[i] = (before + Q.cdf[i]) / total;
    __syncthreads();

    // ---- this rank's children: parent = upper_bound(cdf, u), clamped (the last cdf value is 1 up to rounding)
    const int Ln = (n + kThreads - 1) / kThreads;
    const int klo = min(n, t * Ln), khi = min(n, klo + Ln);
    for (int k = klo; k < khi; ++k) {
        const double u = Q.uniforms[(size_t)Q.rank * n + k];
        int a = 0, b = N;
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (Q.cdf[mid] <= u) a = mid + 1; else b = mid;
        }
        const int p = min(a, N - 1);
        Q.mine[k] = p;
        if (Q.parents_local) Q.parents_local[k] = p;
    }
    __syncthreads();

    // ---- runs of equal parents: where each child's run starts ...
    int last_start = -1;
    for (int k = klo; k < khi; ++k) {
        const bool nw = k == 0 || Q.mine[k] != Q.mine[k - 1];
        if (nw) last_start = k;
        Q.aux[k] = last_start;                       // (-1: the run began in an earlier thread's chunk)
    }
    int dummy;
    const int start_before = block_exscan(last_start, -1, OpMax(), shi, &dummy);
    // ... and where it ends: the same scan over the chunks in reverse order
    const int rt = kThreads - 1 - t;                 // the chunk this thread owns in the reverse pass
    const int rlo = min(n, rt * Ln), rhi = min(n, rlo + Ln);
    int first_end = n;                               // smallest end position at or after the chunk's first child
    for (int k = rhi - 1; k >= rlo; --k)
        if (k == n - 1 || Q.mine[k + 1] != Q.mine[k]) first_end = k;
    const int end_after = block_exscan(first_end, n, OpMin(), shi, &dummy);   // over the chunks that FOLLOW this thread's reverse chunk
    // (hand the reverse pass's result to the thread that owns the chunk in the forward pass)
    __shared__ int end_of_chunk[kThreads];
    end_of_chunk[rt] = end_after;
    __syncthreads();
    const int my_end_after = end_of_chunk[t];

    // ---- shared remote parents are staged once; everything else is read where it lives
    long long c_remote = 0, c_shared = 0, c_start = 0, c_runs = 0;
    int starts = 0;
    {
        int next_end = my_end_after;
        for (int k = khi - 1; k >= klo; --k) {       // backwards: the end of k's run
            if (k == n - 1 || Q.mine[k + 1] != Q.mine[k]) next_end = k;
            const int s = Q.aux[k] >= 0 ? Q.aux[k] : start_before;
            const int len = next_end - s + 1;
            const int p = Q.mine[k], owner = p / n;
            const bool shared = owner != Q.rank && len >= Q.min_share;
            Q.aux[k] = shared ? (s == k ? 2 : 1) : 0;   // 2: the run's first child (stages the window)
            starts += shared && s == k;
        }
    }
    const int starts_before = block_exscan(starts, 0, OpAdd(), shi, &dummy);
    int sidx = starts_before - 1;
    for (int k = klo; k < khi; ++k) {
        const int p = Q.mine[k], owner = p / n;
        const int pg = owner * Q.cap + (p - owner * n);
        const int f = Q.aux[k];
        if (f == 2) ++sidx;
        const bool nw = k == 0 || Q.mine[k] != Q.mine[k - 1];
        Q.parent_idx[k] = f ? Q.rank * Q.cap + n + sidx : pg;
        Q.stage_src[k] = f == 2 ? pg : -1;
        Q.stage_dst[k] = f == 2 ? n + sidx : -1;
        c_remote += owner != Q.rank;
        c_shared += f != 0;
        c_start += f == 2;
        c_runs += nw;
    }
    // (a shared child whose run began in an earlier chunk: sidx = starts_before - 1 is that run's index -- the
    // run's first child is the last start before this chunk, since the run reaches into it)
    long long tot;
    (void)block_exscan(c_remote, 0ll, OpAdd(), reinterpret_cast<long long*>(shd), &tot);
    if (t == 0) Q.counts[0] += tot;
    (void)block_exscan(c_shared, 0ll, OpAdd(), reinterpret_cast<long long*>(shd), &tot);
    if (t == 0) Q.counts[1] += tot;
    (void)block_exscan(c_start, 0ll, OpAdd(), reinterpret_cast<long long*>(shd), &tot);
    if (t == 0) Q.counts[2] += tot;
    (void)block_exscan(c_runs, 0ll, OpAdd(), reinterpret_cast<long long*>(shd), &tot);
    if (t == 0) Q.counts[3] += tot;
}

}  // namespace rbp
