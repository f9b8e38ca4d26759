// nmpc_assemble.h -- device-side assembly of the parameter vectors P[B][np] from structured inputs ("next" row f1).
//
// Replaces, for a whole batch and without a host round trip (reference = /root/reference/src):
//   interfaces/mpc_interface.py:90-100   get_closest_n_stc_obstacles  (distance of the robot to every map polygon,
//                                        the Nstcobs closest)  + pkg_mpc_tracker/utils_geo.py:6-33 lineseg_dists
//   interfaces/mpc_interface.py:73-80    get_stc_constraints -> utils_geo.py:35-62 polygon_halfspace_representation
//   interfaces/mpc_interface.py:82-88    get_dyn_constraints  (flatten + zero-pad to Ndynobs)
//   pkg_mpc_tracker/trajectory_tracker.py:291-317  list concatenation into the flat parameter vector
//
// Two kernels: select_static_kernel (one wavefront per instance; compute-light, latency-bound) writes the o_s block,
// fill_kernel (one dword per lane, 256 B per wave-instruction) writes everything else -- a pure HBM byte mover:
// algorithmic bytes per instance = 2 * sizeof(T) * np (every element written once, read or generated once).
#pragma once

#include <hip/hip_runtime.h>

namespace nmpc {

template <typename T>
struct AsmParams {
    int N, Nother, Nstc, Ndyn, np;
    int off_rs, off_rv, off_c0, off_os, off_od, off_qstc, off_qdyn;
    int B, M, n_dyn;
    const T* last_u;       // [B][2]
    const T* state;        // [B][3]
    const T* ref_states;   // [B][N][3]
    const T* speed_ref;    // [B]
    const T* tuning;       // [10]
    const T* other_robots; // [B][3*(N+1)*Nother] or nullptr
    const T* map_polygons; // [M][4][2]
    const T* dyn;          // [B][n_dyn][N+1][6] or nullptr
    const T* stc_weights;  // [N]
    const T* dyn_weights;  // [N]
    T* P;                  // [B][np]
    int* selected;         // [B][Nstc] indices of the chosen map polygons, nearest first, -1 = none (may be null)
};

__device__ __forceinline__ float thypot(float a, float b) { return hypotf(a, b); }
__device__ __forceinline__ double thypot(double a, double b) { return hypot(a, b); }

// distance from p to the boundary of the quadrilateral q[4][2] (utils_geo.py:6-33 applied to the 4 edges, min)
template <typename T>
__device__ __forceinline__ T quad_distance(T px, T py, const T* q)
{
    T best = T(3.0e38);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const T ax = q[2 * e], ay = q[2 * e + 1], bx = q[2 * ((e + 1) & 3)], by = q[2 * ((e + 1) & 3) + 1];
        const T dbx = bx - ax, dby = by - ay;
        const T len = thypot(dbx, dby);
        const T dx = dbx / len, dy = dby / len;
        const T s = (ax - px) * dx + (ay - py) * dy;
        const T t = (px - bx) * dx + (py - by) * dy;
        T h = s > t ? s : t;
        h = h > T(0) ? h : T(0);
        const T c = (px - ax) * dy - (py - ay) * dx;
        const T d = thypot(h, c);
        best = d < best ? d : best;
    }
    return best;
}

// One wavefront per instance: distances to all M map polygons (LDS), Nstc rounds of wave arg-min, then one lane
// per (slot, edge) converts the selected quadrilaterals to half-space rows (b, a0, a1).
template <typename T>
__global__ __launch_bounds__(64) void select_static_kernel(AsmParams<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* dist = reinterpret_cast<T*>(smem_raw);            // [M]
    int* sel = reinterpret_cast<int*>(dist + a.M);       // [Nstc]
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const T px = a.state[3 * b], py = a.state[3 * b + 1];
    for (int m = lane; m < a.M; m += 64) dist[m] = quad_distance(px, py, a.map_polygons + 8 * m);
    __syncthreads();
    const T BIG = T(3.0e38);
    for (int s = 0; s < a.Nstc; ++s) {
        T bv = BIG;
        int bi = 0x7fffffff;
        for (int m = lane; m < a.M; m += 64) {
            const T d = dist[m];
            if (d < bv) { // strict: lowest index wins ties within a lane (indices ascend)
                bv = d;
                bi = m;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const T ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov < bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            sel[s] = bv < BIG ? bi : -1;
            if (bv < BIG) dist[bi] = BIG;
            if (a.selected) a.selected[(size_t)b * a.Nstc + s] = sel[s];
        }
        __syncthreads();
    }
    // half-space rows: lane -> (slot, edge)
    for (int i = lane; i < a.Nstc * 4; i += 64) {
        const int s = i >> 2, e = i & 3, m = sel[s];
        T bb = 0, a0 = 0, a1 = 0;
        if (m >= 0) {
            const T* q = a.map_polygons + 8 * m;
            const T cx = (q[0] + q[2] + q[4] + q[6]) * T(0.25), cy = (q[1] + q[3] + q[5] + q[7]) * T(0.25);
            const T v1x = q[2 * e] - cx, v1y = q[2 * e + 1] - cy;
            const T v2x = q[2 * ((e + 1) & 3)] - cx, v2y = q[2 * ((e + 1) & 3) + 1] - cy;
            const T det = v1x * v2y - v1y * v2x;
            if (det != T(0)) { // [v1; v2] a = [1; 1]
                a0 = (v2y - v1y) / det;
                a1 = (v1x - v2x) / det;
                bb = a0 * cx + a1 * cy + T(1);
            }
        }
        T* o = a.P + (size_t)b * a.np + a.off_os + 12 * s;
        o[e] = bb;
        o[4 + e] = a0;
        o[8 + e] = a1;
    }
}

#ifndef NMPC_FILL_PER_LANE
#define NMPC_FILL_PER_LANE 4
#endif
constexpr unsigned kFillPerLane = NMPC_FILL_PER_LANE; // elements (independent loads in flight) per lane of the byte mover

template <typename T>
__device__ __forceinline__ T asm_element(const AsmParams<T>& a, unsigned b, unsigned e, bool& skip)
{
    skip = false;
    // largest blocks first: o_d (68 % of the vector at the yaml dimensions), then c_0/c (23 %)
    if (e >= (unsigned)a.off_od && e < (unsigned)a.off_qstc) {
        const unsigned i = e - a.off_od, per = 6 * (a.N + 1);
        return (a.dyn && i < a.n_dyn * per) ? a.dyn[(size_t)b * (a.n_dyn * per) + i] : T(0);
    }
    if (e >= (unsigned)a.off_c0 && e < (unsigned)a.off_os)
        return a.other_robots ? a.other_robots[(size_t)b * (a.off_os - a.off_c0) + (e - a.off_c0)] : T(0);
    if (e >= (unsigned)a.off_os && e < (unsigned)a.off_od) {
        skip = true; // written by select_static_kernel
        return T(0);
    }
    if (e >= (unsigned)a.off_qdyn) return a.dyn_weights[e - a.off_qdyn];
    if (e >= (unsigned)a.off_qstc) return a.stc_weights[e - a.off_qstc];
    if (e >= (unsigned)a.off_rv) return a.speed_ref[b];
    if (e >= (unsigned)a.off_rs) return a.ref_states[(size_t)b * a.N * 3 + (e - a.off_rs)];
    if (e >= 8u) return a.tuning[e - 8];
    if (e >= 5u) return a.ref_states[((size_t)b * a.N + (a.N - 1)) * 3 + (e - 5)]; // goal = last reference row
    if (e >= 2u) return a.state[3 * b + (e - 2)];
    return a.last_u[2 * b + e];
}

// Byte mover: block = (instance, 256 * kFillPerLane-element chunk of its row) flattened into blockIdx.x; consecutive lanes handle
// consecutive elements (256 B per wave-instruction on both the load and the store side; the only index division is
// one scalar division per block), kFillPerLane independent loads in flight per lane.
template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(AsmParams<T> a, unsigned nchunk)
{
    const unsigned b = blockIdx.x / nchunk, chunk = blockIdx.x - b * nchunk, np = (unsigned)a.np;
    T* row = a.P + (size_t)b * np;
    T v[kFillPerLane];
    bool skip[kFillPerLane];
#pragma unroll
    for (int j = 0; j < kFillPerLane; ++j) {
        const unsigned e = (chunk * kFillPerLane + j) * 256u + threadIdx.x;
        skip[j] = true;
        v[j] = 0;
        if (e < np) v[j] = asm_element(a, b, e, skip[j]);
    }
#pragma unroll
    for (int j = 0; j < kFillPerLane; ++j) {
        const unsigned e = (chunk * kFillPerLane + j) * 256u + threadIdx.x;
        if (!skip[j]) row[e] = v[j];
    }
}

} // namespace nmpc
