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
// One wavefront per instance, one lane per hypothesis point (P <= 64): the adjacency and reachability sets are 64-bit
// lane masks, transitive closure by repeated squaring with v_readlane broadcasts, cluster statistics by DPP wave sums.
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

__device__ __forceinline__ unsigned long long read_lane_u64(unsigned long long v, int lane)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ float hsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double hsqrt(double x) { return sqrt(x); }

template <typename T>
__global__ __launch_bounds__(64) void hypotheses_kernel(HypParams<T> a)
{
    __shared__ int counts[65]; // clusters per time offset (index 0 = current positions)
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int NP1 = a.N + 1;
    T* out = a.dyn + (size_t)b * a.Ndyn * NP1 * 6;
    const bool on = lane < a.P;
    int max_cl = a.H;
    if (lane == 0) counts[0] = a.H;
    for (int t = 0; t < a.N; ++t) {
        const T* pt = a.hypos + (((size_t)b * a.N + t) * a.P + (on ? lane : 0)) * 2;
        const T x = on ? pt[0] : T(0), y = on ? pt[1] : T(0);
        // adjacency: bit j of adj = point j within eps of this lane's point (incl. itself)
        unsigned long long adj = 0;
        const T eps2 = a.eps * a.eps;
        for (int j = 0; j < a.P; ++j) {
            const T dx = x - read_lane(x, j), dy = y - read_lane(y, j);
            if (dx * dx + dy * dy <= eps2) adj |= 1ull << j;
        }
        if (!on) adj = 0;
        const bool noise = __popcll(adj) < 2; // min_samples = 2 counts the point itself
        // transitive closure: reach <- reach o reach, 6 squarings cover paths of length 64
        unsigned long long reach = adj;
        for (int it = 0; it < 6; ++it) {
            unsigned long long nw = reach;
            for (int j = 0; j < a.P; ++j) {
                const unsigned long long rj = read_lane_u64(reach, j);
                if ((reach >> j) & 1ull) nw |= rj;
            }
            const bool changed = __ballot(nw != reach) != 0ull;
            reach = nw;
            if (!changed) break;
        }
        const int label = noise ? 64 : (int)__builtin_ctzll(reach); // smallest index of the component
        const unsigned long long leaders = __ballot(!noise && label == lane);
        const int ncl = __popcll(leaders);
        const int cid = noise ? -1 : __popcll(leaders & ((1ull << label) - 1ull));
        if (lane == 0) counts[t + 1] = ncl;
        max_cl = ncl > max_cl ? ncl : max_cl;
        const int nstore = ncl < a.Ndyn ? ncl : a.Ndyn;
        for (int c = 0; c < nstore; ++c) { // wave-uniform
            const bool in = cid == c;
            T cnt, sx, sy;
            wave_sum3(in ? T(1) : T(0), in ? x : T(0), in ? y : T(0), cnt, sx, sy);
            const T mx = sx / cnt, my = sy / cnt;
            T vx, vy;
            wave_sum2(in ? (x - mx) * (x - mx) : T(0), in ? (y - my) * (y - my) : T(0), vx, vy);
            if (lane == 0) {
                T* o = out + ((size_t)c * NP1 + (t + 1)) * 6;
                o[0] = mx;
                o[1] = my;
                o[2] = hsqrt(vx / cnt) * a.enlarge + a.extra_margin;
                o[3] = hsqrt(vy / cnt) * a.enlarge + a.extra_margin;
                o[4] = T(0);
                o[5] = T(1);
            }
        }
    }
    __syncthreads();
    const int n_obs = max_cl; // main_base.py:294-297
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

} // namespace nmpc
