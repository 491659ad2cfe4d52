// rbsensor_peers.hip -- the resampling half of the filter step across PROCESSES (one rank per GPU,
// SURVEY 8(e); dbot_ros_amd/dist.py PeerShardedStep): everything between the all-gather of the
// log-likelihoods and the next step's parent indices in three or four launches: the largest log-likelihood per tile
// (large jobs only), weights + cumulative sums per tile, the children's parents -- all over the chip -- and one block
// for the plan.
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
constexpr int kTileShift = 11, kTile = 1 << kTileShift;   // particles per block of the two elementwise passes: two per thread
constexpr int kMaxTiles = 1024;                            // (their totals sit in the plan kernel's LDS: N <= 2 M particles)
constexpr int kOwnMaxTiles = 16;                           // up to here peer_weights_kernel's blocks find the maximum themselves
constexpr int kSamples = 2048;                             // cdf values kept in LDS to start a search from
constexpr int kLdsChildren = 4096;                         // the general plan (min_share != 2) keeps a rank's children's parents in LDS up to this many

struct PeerPlan {
    const double* ll_all;     // [N] gathered log-likelihoods, rank-major
    const double* uniforms;   // [N] ascending
    int N, n, rank, cap, min_share;
    double temperature;
    double* cdf;              // [N] scratch: cumulative weights WITHIN a tile of kTile particles (peer_weights_kernel)
    double* tile_max;         // [tiles] scratch: the largest log-likelihood of a tile (peer_max_kernel)
    double* tile_total;       // [tiles] scratch: the weight of a tile
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

// The cdf over all N particles, spread over the chip (one CU took 3.6 ns per particle for the exponentials and the
// scan: 57 us of N = 16 000, 720 us of N = 200 000): the largest log-likelihood per tile ...
__global__ __launch_bounds__(kThreads) void peer_max_kernel(const PeerPlan Q)
{
    __shared__ double shd[kThreads / 64];
    const int i0 = (int)blockIdx.x * kTile + 2 * (int)threadIdx.x;
    double m = -INFINITY;
    if (i0 < Q.N) m = fmax(m, Q.ll_all[i0]);              // (fmax drops NaN: a contained particle weighs nothing)
    if (i0 + 1 < Q.N) m = fmax(m, Q.ll_all[i0 + 1]);
    double tot;
    (void)block_exscan(m, -(double)INFINITY, OpMax(), shd, &tot);
    if (threadIdx.x == 0) Q.tile_max[blockIdx.x] = tot;
}

// ... then the weights exp((ll - max) / T) and their cumulative sums inside each tile, and each tile's total.
__global__ __launch_bounds__(kThreads) void peer_weights_kernel(const PeerPlan Q)
{
    __shared__ double shd[kThreads / 64];
    const int t = (int)threadIdx.x, tiles = (int)gridDim.x;
    double m = -INFINITY;
    if (tiles <= kOwnMaxTiles) {   // few tiles: every block finds the largest log-likelihood itself (a launch less: ~4.5 us)
        for (int i = t; i < Q.N; i += kThreads) m = fmax(m, Q.ll_all[i]);
    } else {
        for (int k = t; k < tiles; k += kThreads) m = fmax(m, Q.tile_max[k]);
    }
    {
        double tot;
        (void)block_exscan(m, -(double)INFINITY, OpMax(), shd, &tot);
        m = tot;
    }
    __syncthreads();
    const int i0 = (int)blockIdx.x * kTile + 2 * t;
    double w0 = 0.0, w1 = 0.0;
    if (i0 < Q.N)     { const double l = Q.ll_all[i0];     w0 = l == l ? exp((l - m) / Q.temperature) : 0.0; }
    if (i0 + 1 < Q.N) { const double l = Q.ll_all[i0 + 1]; w1 = l == l ? exp((l - m) / Q.temperature) : 0.0; }
    double tile_total;
    const double before = block_exscan(w0 + w1, 0.0, OpAdd(), shd, &tile_total);
    if (i0 < Q.N)     Q.cdf[i0] = before + w0;
    if (i0 + 1 < Q.N) Q.cdf[i0 + 1] = before + (w0 + w1);
    if (t == 0) Q.tile_total[blockIdx.x] = tile_total;
}

// This rank's children's parents, a child per thread over ceil(n / kThreads) blocks: each block scans the tiles' totals
// in LDS (cdf value i = (weight of the tiles before i's + cumulative weight inside the tile) / total, formed where it
// is read, so the cdf is never rewritten) and keeps kSamples evenly spaced cdf values there -- a search narrows to one
// stretch of `stride` particles in LDS and only its last log2(stride) probes go to memory.  parent =
// upper_bound(cdf, u), clamped (the last cdf value is 1 up to rounding).  (One block walking the cdf for all of the
// rank's children was bound by its CU's load path: 190 us of 25 000 children with flat weights.)
__global__ __launch_bounds__(kThreads) void peer_search_kernel(const PeerPlan Q)
{
    __shared__ double shd[kThreads / 64];
    __shared__ double tile_before[kMaxTiles];
    const int t = (int)threadIdx.x, N = Q.N, n = Q.n;
    const int tiles = (N + kTile - 1) >> kTileShift;
    double total;
    {
        double carry = 0.0;
        for (int base = 0; base < tiles; base += kThreads) {
            const int k = base + t;
            const double v = k < tiles ? Q.tile_total[k] : 0.0;
            double part;
            const double before = block_exscan(v, 0.0, OpAdd(), shd, &part);
            if (k < tiles) tile_before[k] = carry + before;
            carry += part;
        }
        total = carry;
    }
    __syncthreads();
    auto cdf_at = [&](int i) { return (tile_before[i >> kTileShift] + Q.cdf[i]) / total; };
    __shared__ double sample[kSamples];
    const int stride = (N + kSamples - 1) / kSamples, stretches = (N + stride - 1) / stride;
    for (int j = t; j < stretches; j += kThreads) sample[j] = cdf_at(min(N, (j + 1) * stride) - 1);
    __syncthreads();
    const int k = (int)blockIdx.x * kThreads + t;
    if (k >= n) return;
    const double u = Q.uniforms[(size_t)Q.rank * n + k];
    // cdf_at(i) <= u without the division where the answer is clear: the quotient's rounding (and the product's, u x
    // total) moves either side by parts in 1e16, so outside a band of 1e-15 around u x total the sums decide
    const bool weighable = total > 0.0 && total < INFINITY;   // (no particle weighs anything: every comparison takes the division)
    const double ut = u * total;
    const double ut_lo = weighable ? ut * (1.0 - 1e-15) : -INFINITY, ut_hi = weighable ? ut * (1.0 + 1e-15) : INFINITY;
    int a = 0, b = stretches;
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (sample[mid] <= u) a = mid + 1; else b = mid;
    }
    int p = N;                                                   // first particle with cdf > u (N: none)
    if (a < stretches) {
        int lo = a * stride, hi = min(N, (a + 1) * stride) - 1;  // cdf(hi) > u, and cdf(lo - 1) <= u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const double x = tile_before[mid >> kTileShift] + Q.cdf[mid];
            const bool le = x <= ut_lo ? true : x >= ut_hi ? false : x / total <= u;
            if (le) lo = mid + 1; else hi = mid;
        }
        p = lo;
    }
    const int pc = min(p, N - 1);
    Q.mine[k] = pc;
    if (Q.parents_local) Q.parents_local[k] = pc;
}

// One block: the plan of this rank's children from their parents (peer_search_kernel).
__global__ __launch_bounds__(kThreads) void peer_resample_kernel(const PeerPlan Q)
{
    __shared__ double shd[kThreads / 64];
    __shared__ int shi[kThreads / 64];
    const int t = (int)threadIdx.x, n = Q.n;
    // ---- min_share == 2 (the default): "at least two children share the parent" is a question to a
    // child's two neighbours, so ONE forward sweep does the plan -- tiles of 4 x kThreads children, four consecutive ones
    // per thread, everything read and written in order (the general plan below, on more children than LDS holds, walks a
    // thread's scattered children four times: 200 us of 25 000 children, one CU reading 64 cache lines per wave load)
    if (Q.min_share == 2) {
        long long c_remote = 0, c_shared = 0, c_start = 0, c_runs = 0;
        int carry = 0, dummy;                            // windows staged before this tile
        for (int base = 0; base < n; base += 4 * kThreads) {
            const int k0 = base + 4 * t;
            int pm[6];                                   // parents of children k0 - 1 ... k0 + 4 (-1 / -2: no such child)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int kk = k0 - 1 + j;
                pm[j] = kk < 0 ? -1 : kk >= n ? -2 : Q.mine[kk];
            }
            int f[4], starts = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[j] = 0;
                if (k0 + j >= n) continue;
                const int p = pm[j + 1], owner = p / n;
                const bool nw = p != pm[j], last = p != pm[j + 2], remote = owner != Q.rank;
                const bool shared = remote && !(nw && last);
                f[j] = shared ? (nw ? 2 : 1) : 0;       // 2: the run's first child (stages the window)
                starts += f[j] == 2;
                c_remote += remote; c_shared += shared; c_start += f[j] == 2; c_runs += nw;
            }
            int tile_total;
            const int before = block_exscan(starts, 0, OpAdd(), shi, &tile_total);
            int sidx = carry + before - 1;              // (a shared child whose run began earlier: the last start before it is its run's)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j;
                if (k >= n) continue;
                const int p = pm[j + 1], owner = p / n;
                const int pg = owner * Q.cap + (p - owner * n);
                if (f[j] == 2) ++sidx;
                Q.parent_idx[k] = f[j] ? Q.rank * Q.cap + n + sidx : pg;
                Q.stage_src[k] = f[j] == 2 ? pg : -1;
                Q.stage_dst[k] = f[j] == 2 ? n + sidx : -1;
            }
            carry += tile_total;
        }
        (void)dummy;
        // (four counts, each at most n < 2^31, in two sums)
        long long tot;
        (void)block_exscan((c_remote << 32) | c_shared, 0ll, OpAdd(), reinterpret_cast<long long*>(shd), &tot);
        if (t == 0) { Q.counts[0] += tot >> 32; Q.counts[1] += tot & 0xffffffffll; }
        (void)block_exscan((c_start << 32) | c_runs, 0ll, OpAdd(), reinterpret_cast<long long*>(shd), &tot);
        if (t == 0) { Q.counts[2] += tot >> 32; Q.counts[3] += tot & 0xffffffffll; }
        return;
    }

    // ---- any other min_share: run lengths.  (The plan's two work arrays: in LDS when this rank's children fit.)
    __shared__ int mine_s[kLdsChildren], aux_s[kLdsChildren];
    int* __restrict__ const mine = n <= kLdsChildren ? mine_s : Q.mine;
    int* __restrict__ const aux = n <= kLdsChildren ? aux_s : Q.aux;
    if (n <= kLdsChildren)
        for (int k = t; k < n; k += kThreads) mine_s[k] = Q.mine[k];
    const int Ln = (n + kThreads - 1) / kThreads;
    const int klo = min(n, t * Ln), khi = min(n, klo + Ln);
    __syncthreads();

    // runs of equal parents: where each child's run starts ...
    int last_start = -1;
    for (int k = klo; k < khi; ++k) {
        const bool nw = k == 0 || mine[k] != mine[k - 1];
        if (nw) last_start = k;
        aux[k] = last_start;                       // (-1: the run began in an earlier thread's chunk)
    }
    int dummy;
    const int start_before = block_exscan(last_start, -1, OpMax(), shi, &dummy);
    // ... and where it ends: the same scan over the chunks in reverse order
    const int rt = kThreads - 1 - t;                 // the chunk this thread owns in the reverse pass
    const int rlo = min(n, rt * Ln), rhi = min(n, rlo + Ln);
    int first_end = n;                               // smallest end position at or after the chunk's first child
    for (int k = rhi - 1; k >= rlo; --k)
        if (k == n - 1 || mine[k + 1] != mine[k]) first_end = k;
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
            if (k == n - 1 || mine[k + 1] != mine[k]) next_end = k;
            const int s = aux[k] >= 0 ? aux[k] : start_before;
            const int len = next_end - s + 1;
            const int p = mine[k], owner = p / n;
            const bool shared = owner != Q.rank && len >= Q.min_share;
            aux[k] = shared ? (s == k ? 2 : 1) : 0;   // 2: the run's first child (stages the window)
            starts += shared && s == k;
        }
    }
    const int starts_before = block_exscan(starts, 0, OpAdd(), shi, &dummy);
    int sidx = starts_before - 1;
    for (int k = klo; k < khi; ++k) {
        const int p = mine[k], owner = p / n;
        const int pg = owner * Q.cap + (p - owner * n);
        const int f = aux[k];
        if (f == 2) ++sidx;
        const bool nw = k == 0 || mine[k] != mine[k - 1];
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
