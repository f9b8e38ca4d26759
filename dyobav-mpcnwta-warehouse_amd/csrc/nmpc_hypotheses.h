// nmpc_hypotheses.h -- multi-hypothesis motion predictions -> obstacle ellipses on the device ("next" row f2).
//
// Replaces, for a whole batch (reference = /root/reference/src):
//   utils_test.py:133-143  fit_DBSCAN(data, eps=1, min_sample=2)  (sklearn): with min_samples = 2 every point with another
//                          point within eps is a core point, so clusters = connected components of the "distance <= eps"
//                          graph, numbered by their smallest point index; isolated points are noise
//   utils_test.py:145-151  fit_cluster2gaussian(clusters, enlarge, extra_margin): mean and population std per cluster
//   main_base.py:293-302   obstacle list: row [mu_x, mu_y, std_x, std_y, 0, 1] per cluster and time offset, the current
//                          positions with HUMAN_SIZE at offset 0, [0,0,0,0,0,1] where a used slot has no cluster
//
// One wavefront per instance; a pass handles G = 64 / P time offsets at once, lane = g * P + i holding point i of offset
// t0 + g (P = 20: three offsets per pass). The point coordinates of the pass are staged in LDS (one ds_read_b64 broadcasts
// point j of the lane's own group), adjacency and reachability are P-bit masks relative to the group, the transitive
// closure is Warshall's algorithm on the mask rows (P steps, row j fetched with ds_bpermute), and every lane sums the
// statistics of ITS cluster over the group's points (centred on the cluster's first point, single pass) -- the first lane
// of each cluster then stores the row. The kernel is instruction-issue bound: everything above is what keeps the count
// per time offset at ~180 VALU/LDS instructions for P = 20 (the one-offset-per-pass version with v_readlane loops, closure
// by repeated squaring and one wave-wide DPP reduction per cluster needed ~1500).
#pragma once

#include <hip/hip_runtime.h>

#include "wave_ops.h"

namespace nmpc {

template <typename T>
struct HypParams {
    int B, N, P, H, Ndyn;
    T eps, human_size, enlarge, extra_margin;
    const T* hypos; // [B][N][P][2]
    const T* cur;   // [B][H][2]
    T* dyn;         // [B][Ndyn][N+1][6]
    int* n_obs;     // [B] (may be null)
};

__device__ __forceinline__ float hsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double hsqrt(double x) { return sqrt(x); }

// value of v in lane (addr4 / 4) of the wavefront
__device__ __forceinline__ unsigned bperm(unsigned v, int addr4) { return (unsigned)__builtin_amdgcn_ds_bpermute(addr4, (int)v); }
__device__ __forceinline__ unsigned long long bperm(unsigned long long v, int addr4)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(addr4, (int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(addr4, (int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int popc(unsigned v) { return __popc(v); }
__device__ __forceinline__ int popc(unsigned long long v) { return __popcll(v); }
__device__ __forceinline__ int ctz(unsigned v) { return (int)__builtin_ctz(v); }
__device__ __forceinline__ int ctz(unsigned long long v) { return (int)__builtin_ctzll(v); }

// Behind the time offsets: n_obs = the largest cluster count (main_base.py:294-297), the rows of offset 0 (current
// positions), [0,0,0,0,0,1] where a used obstacle has no cluster at an offset, zeros in the unused obstacles.
template <typename T>
__device__ __forceinline__ void hyp_finish(const HypParams<T>& a, const int b, const int lane, const int* counts, T* out)
{
    const int NP1 = a.N + 1;
    __syncthreads();
    int max_cl = a.H; // main_base.py:294-297
    for (int t = lane; t < a.N; t += 64) max_cl = counts[t + 1] > max_cl ? counts[t + 1] : max_cl;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const int o = __shfl_xor(max_cl, s);
        max_cl = o > max_cl ? o : max_cl;
    }
    const int n_obs = max_cl;
    const int used = n_obs < a.Ndyn ? n_obs : a.Ndyn;
    if (lane == 0 && a.n_obs) a.n_obs[b] = n_obs;
    // offset 0: current positions; empty slots of used obstacles: [0,0,0,0,0,1]; unused obstacles: zeros
    for (int i = lane; i < a.Ndyn * NP1; i += 64) {
        const int c = i / NP1, t = i - c * NP1;
        const int have = counts[t] < a.Ndyn ? counts[t] : a.Ndyn;
        if (c < have && t > 0) continue; // written above
        T* o = out + (size_t)i * 6;
        if (c < have) { // t == 0
            const T* p = a.cur + ((size_t)b * a.H + c) * 2;
            o[0] = p[0];
            o[1] = p[1];
            o[2] = a.human_size;
            o[3] = a.human_size;
            o[4] = T(0);
            o[5] = T(1);
        } else {
            o[0] = o[1] = o[2] = o[3] = o[4] = T(0);
            o[5] = c < used ? T(1) : T(0);
        }
    }
}

// M: mask type of one group (unsigned for P <= 32, unsigned long long for P <= 64)
template <typename T, typename M>
__global__ __launch_bounds__(64) void hypotheses_kernel(HypParams<T> a)
{
    __shared__ int counts[65]; // clusters per time offset (index 0 = current positions)
    __shared__ T pts[64 * 2];  // (x, y) of the pass, by lane
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int NP1 = a.N + 1, P = a.P;
    const int G = 64 / P;              // time offsets per pass
    const int g = lane / P, i = lane - g * P;
    const bool in_group = g < G;
    const int gbase = in_group ? g * P : 0;
    const M one = 1;
    const M all = P == (int)(8 * sizeof(M)) ? ~M(0) : (one << P) - one;
    T* out = a.dyn + (size_t)b * a.Ndyn * NP1 * 6;
    const T eps2 = a.eps * a.eps;
    if (lane == 0) counts[0] = a.H;
    for (int t0 = 0; t0 < a.N; t0 += G) {
        const int t = t0 + g;
        const bool on = in_group && t < a.N;
        const T* pt = a.hypos + (((size_t)b * a.N + (on ? t : 0)) * P + i) * 2;
        T x = 0, y = 0;
        if (on) {
            x = pt[0];
            y = pt[1];
        }
        __syncthreads(); // (one wavefront: orders the LDS reads of the previous pass before these writes)
        pts[2 * lane] = x;
        pts[2 * lane + 1] = y;
        __syncthreads();
        const T* gp = pts + 2 * gbase;
        // adjacency: bit j = point j of this group within eps of this lane's point (incl. itself)
        M adj = 0;
#pragma clang loop vectorize(disable) unroll_count(4)
        for (int j = 0; j < P; ++j) {
            const T dx = x - gp[2 * j], dy = y - gp[2 * j + 1];
            if (dx * dx + dy * dy <= eps2) adj |= one << j;
        }
        if (!on) adj = 0;
        const bool noise = popc(adj) < 2; // min_samples = 2 counts the point itself
        // transitive closure (Warshall): after step j every row holds the points reachable through points 0..j
        M reach = adj;
        const int ga4 = gbase * 4;
#pragma clang loop vectorize(disable) unroll_count(4)
        for (int j = 0; j < P; ++j) {
            const M rj = bperm(reach, ga4 + 4 * j);
            if ((reach >> j) & one) reach |= rj;
        }
        const int label = noise ? 0 : ctz(reach); // smallest index of the component
        const bool leader = !noise && label == i;
        const M leaders = (M)(__ballot(leader) >> gbase) & all;
        const int ncl = popc(leaders);
        const int cid = popc(leaders & ((one << label) - one));
        if (on && i == 0) counts[t + 1] = ncl;
        // statistics of this lane's cluster, centred on its first point
        const T xr = gp[2 * label], yr = gp[2 * label + 1];
        T cnt = 0, sx = 0, sy = 0, sxx = 0, syy = 0;
#pragma clang loop vectorize(disable) unroll_count(4)
        for (int j = 0; j < P; ++j) {
            const bool in = (reach >> j) & one;
            const T dx = in ? gp[2 * j] - xr : T(0), dy = in ? gp[2 * j + 1] - yr : T(0);
            cnt += in ? T(1) : T(0);
            sx += dx;
            sy += dy;
            sxx += dx * dx;
            syy += dy * dy;
        }
        if (leader && cid < a.Ndyn) {
            const T inv = T(1) / cnt;
            const T mx = sx * inv, my = sy * inv;
            const T vx = sxx * inv - mx * mx, vy = syy * inv - my * my;
            T* o = out + ((size_t)cid * NP1 + (t + 1)) * 6;
            o[0] = xr + mx;
            o[1] = yr + my;
            o[2] = hsqrt(vx > T(0) ? vx : T(0)) * a.enlarge + a.extra_margin;
            o[3] = hsqrt(vy > T(0) ? vy : T(0)) * a.enlarge + a.extra_margin;
            o[4] = T(0);
            o[5] = T(1);
        }
    }
    hyp_finish(a, b, lane, counts, out);
}

// More than 64 points per time offset (the reference clusters the hypotheses of ALL pedestrians together,
// main_base.py:192-196: 8 pedestrians x 20 hypotheses = 160 points for BASELINE configs[4]): one time offset per pass, lane l
// holds points l, l + 64, ... (PPL per lane), masks are W = 2 * PPL 32-bit words per point in registers. Warshall's
// closure runs over all P rows: the lane that owns row k publishes it through LDS, every point that reaches k ORs it in.
// Only the first point of a cluster needs the statistics (it is their centre of expansion and the lane that stores the
// row), so the sums are skipped for register slots without such a point.
template <typename T, int PPL>
__global__ __launch_bounds__(64) void hypotheses_wide_kernel(HypParams<T> a)
{
    constexpr int W = 2 * PPL;
    __shared__ int counts[65];
    __shared__ T pts[64 * PPL * 2];
    __shared__ __attribute__((aligned(16))) unsigned rowk[W];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int NP1 = a.N + 1, P = a.P;
    T* out = a.dyn + (size_t)b * a.Ndyn * NP1 * 6;
    const T eps2 = a.eps * a.eps;
    if (lane == 0) counts[0] = a.H;
    for (int t = 0; t < a.N; ++t) {
        T x[PPL], y[PPL];
        bool on[PPL];
        __syncthreads(); // (one wavefront: orders the LDS reads of the previous offset before these writes)
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int p = lane + 64 * q;
            on[q] = p < P;
            const T* pt = a.hypos + (((size_t)b * a.N + t) * P + (on[q] ? p : 0)) * 2;
            x[q] = on[q] ? pt[0] : T(0);
            y[q] = on[q] ? pt[1] : T(0);
            pts[2 * p] = x[q];
            pts[2 * p + 1] = y[q];
        }
        __syncthreads();
        // adjacency, one mask word (32 points) at a time
        unsigned reach[PPL][W];
        int deg[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) deg[q] = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            unsigned word[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) word[q] = 0;
            const int j0 = 32 * w, jn = P - j0 < 32 ? P - j0 : 32;
#pragma clang loop vectorize(disable) unroll_count(4)
            for (int jb = 0; jb < jn; ++jb) {
                const T xj = pts[2 * (j0 + jb)], yj = pts[2 * (j0 + jb) + 1];
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    const T dx = x[q] - xj, dy = y[q] - yj;
                    if (dx * dx + dy * dy <= eps2) word[q] |= 1u << jb;
                }
            }
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                reach[q][w] = on[q] ? word[q] : 0u;
                deg[q] += __popc(reach[q][w]);
            }
        }
        // transitive closure (Warshall over the P rows)
#pragma unroll
        for (int q0 = 0; q0 < PPL; ++q0) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int w0 = 2 * q0 + half, k0 = 64 * q0 + 32 * half;
                const int kn = P - k0 < 32 ? P - k0 : 32;
                for (int kb = 0; kb < kn; ++kb) {
                    __syncthreads();
                    if (lane == 32 * half + kb) {
#pragma unroll
                        for (int w = 0; w < W; ++w) rowk[w] = reach[q0][w];
                    }
                    __syncthreads();
                    unsigned rk[W];
#pragma unroll
                    for (int w = 0; w < W; ++w) rk[w] = rowk[w];
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        const unsigned m = 0u - ((reach[q][w0] >> kb) & 1u);
#pragma unroll
                        for (int w = 0; w < W; ++w) reach[q][w] |= rk[w] & m;
                    }
                }
            }
        }
        // clusters = components with more than one point, numbered by their smallest point index
        bool leader[PPL];
        unsigned long long leadmask[PPL];
        int ncl = 0;
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            int label = 0;
#pragma unroll
            for (int w = W - 1; w >= 0; --w)
                if (reach[q][w]) label = 32 * w + (int)__builtin_ctz(reach[q][w]);
            leader[q] = on[q] && deg[q] >= 2 && label == lane + 64 * q; // (min_samples = 2 counts the point itself)
            leadmask[q] = __ballot(leader[q]);
            ncl += __popcll(leadmask[q]);
        }
        if (lane == 0) counts[t + 1] = ncl;
        int before = 0; // clusters whose first point sits in an earlier register slot
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            if (leadmask[q] != 0ull) { // (wave-uniform)
                const int cid = before + __popcll(leadmask[q] & ((1ull << lane) - 1ull));
                const T xr = x[q], yr = y[q];
                T cnt = 0, sx = 0, sy = 0, sxx = 0, syy = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const int j0 = 32 * w, jn = P - j0 < 32 ? P - j0 : 32;
#pragma clang loop vectorize(disable) unroll_count(4)
                    for (int jb = 0; jb < jn; ++jb) {
                        const bool in = (reach[q][w] >> jb) & 1u;
                        const T dx = in ? pts[2 * (j0 + jb)] - xr : T(0), dy = in ? pts[2 * (j0 + jb) + 1] - yr : T(0);
                        cnt += in ? T(1) : T(0);
                        sx += dx;
                        sy += dy;
                        sxx += dx * dx;
                        syy += dy * dy;
                    }
                }
                if (leader[q] && cid < a.Ndyn) {
                    const T inv = T(1) / cnt;
                    const T mx = sx * inv, my = sy * inv;
                    const T vx = sxx * inv - mx * mx, vy = syy * inv - my * my;
                    T* o = out + ((size_t)cid * NP1 + (t + 1)) * 6;
                    o[0] = xr + mx;
                    o[1] = yr + my;
                    o[2] = hsqrt(vx > T(0) ? vx : T(0)) * a.enlarge + a.extra_margin;
                    o[3] = hsqrt(vy > T(0) ? vy : T(0)) * a.enlarge + a.extra_margin;
                    o[4] = T(0);
                    o[5] = T(1);
                }
            }
            before += __popcll(leadmask[q]);
        }
    }
    hyp_finish(a, b, lane, counts, out);
}

} // namespace nmpc
