// nmpc_assemble.h -- device-side assembly of the parameter vectors P[B][np] from structured inputs ("next" row f1).
//
// Replaces, for a whole batch and without a host round trip (reference = /root/reference/src):
//   interfaces/mpc_interface.py:90-100   get_closest_n_stc_obstacles  (distance of the robot to every map polygon,
//                                        the Nstcobs closest)  + pkg_mpc_tracker/utils_geo.py:6-33 lineseg_dists
//   interfaces/mpc_interface.py:73-80    get_stc_constraints -> utils_geo.py:35-62 polygon_halfspace_representation
//   interfaces/mpc_interface.py:82-88    get_dyn_constraints  (flatten + zero-pad to Ndynobs)
//   pkg_mpc_tracker/trajectory_tracker.py:291-317  list concatenation into the flat parameter vector
//
// One fused kernel, one workgroup per instance: wavefront 0 (select_static; compute-light, latency-bound) writes the
// o_s block, wavefronts 1..3 (fill_row; element pairs) write everything else -- a pure HBM byte mover:
// algorithmic bytes per instance = 2 * sizeof(T) * np (every element written once, read or generated once).
#pragma once

#include <hip/hip_runtime.h>

#include "wave_ops.h"

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
// Runs on ONE wavefront (the LDS it uses is private to that wavefront, whose LDS operations execute in program order:
// wave_barrier() only keeps the compiler from moving them).
template <typename T>
__device__ __forceinline__ void select_static(const AsmParams<T>& a, unsigned char* smem_raw)
{
    T* dist = reinterpret_cast<T*>(smem_raw);            // [M]
    int* sel = reinterpret_cast<int*>(dist + a.M);       // [Nstc]
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const T px = a.state[3 * b], py = a.state[3 * b + 1];
    const T BIG = T(3.0e38);
    if (a.M <= 64) {
        // Up to one polygon per lane: no selection rounds. Every lane counts the lanes whose (distance, index) key is
        // smaller than its own -- 64 scalar-indexed v_readlane broadcasts -- and the lanes of rank < Nstc write their
        // polygon to slot `rank`: the Nstc closest, nearest first, lowest index first among equals, like the rounds below.
        const T d = lane < a.M ? quad_distance(px, py, a.map_polygons + 8 * lane) : BIG;
        int rank = 0;
        for (int j = 0; j < a.M; ++j) {
            const T dj = read_lane(d, j);
            rank += (dj < d || (dj == d && j < lane)) ? 1 : 0;
        }
        for (int s = lane; s < a.Nstc; s += 64) sel[s] = -1;
        __builtin_amdgcn_wave_barrier();
        if (lane < a.M && rank < a.Nstc) sel[rank] = lane;
        __builtin_amdgcn_wave_barrier();
        if (a.selected)
            for (int s = lane; s < a.Nstc; s += 64) a.selected[(size_t)b * a.Nstc + s] = sel[s];
    } else {
    for (int m = lane; m < a.M; m += 64) dist[m] = quad_distance(px, py, a.map_polygons + 8 * m);
    __builtin_amdgcn_wave_barrier();
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
        __builtin_amdgcn_wave_barrier();
    }
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

// One element of the small blocks of the parameter vector: everything but c_0/c, o_s and o_d
// (u_m1, s_0, s_N, q, r_s, r_v in front of them; q_stc, q_dyn behind).
template <typename T>
__device__ __forceinline__ T asm_small_element(const AsmParams<T>& a, unsigned b, unsigned e)
{
    if (e >= (unsigned)a.off_qdyn) return a.dyn_weights[e - a.off_qdyn];
    if (e >= (unsigned)a.off_qstc) return a.stc_weights[e - a.off_qstc];
    if (e >= (unsigned)a.off_rv) return a.speed_ref[b];
    if (e >= (unsigned)a.off_rs) return a.ref_states[(size_t)b * a.N * 3 + (e - a.off_rs)];
    if (e >= 8u) return a.tuning[e - 8];
    if (e >= 5u) return a.ref_states[((size_t)b * a.N + (a.N - 1)) * 3 + (e - 5)]; // goal = last reference row
    if (e >= 2u) return a.state[3 * b + (e - 2)];
    return a.last_u[2 * b + e];
}

template <typename T>
struct alignas(2 * sizeof(T)) Pair {
    T a, b;
};

// Byte mover: ONE workgroup per instance. The two big blocks of the row -- o_d (68 % at the yaml dimensions: a straight
// copy of the instance's obstacle rows followed by the zero padding) and c_0/c (23 %: the other robots, or zeros) -- move
// as element PAIRS (8 B per lane in fp32, 16 B in fp64: 512 B / 1 KB per wave-instruction) whenever the block
// boundaries are pair-aligned, the ~160 elements of the small blocks one per lane. Every load of a thread is issued
// before its first store, so a resident workgroup keeps its whole share of the row (5-7 KB) in flight.
constexpr int kAsmThreads = 256; // workgroup of the fused kernel: wavefront 0 selects, wavefronts 1..3 move bytes
constexpr int kFillMaxPairs = 8; // pairs per thread and block held in registers (rows up to 2 * 8 * 256 elements per block)

template <typename T, int kFillThreads>
__device__ __forceinline__ void fill_row(const AsmParams<T>& a, const unsigned vec_ok, const unsigned tid)
{
    const unsigned b = blockIdx.x;
    T* __restrict__ row = a.P + (size_t)b * a.np;
    const unsigned per = 6u * (a.N + 1), n_src = a.dyn ? (unsigned)a.n_dyn * per : 0u;
    const unsigned n_od = (unsigned)(a.off_qstc - a.off_od), n_c = (unsigned)(a.off_os - a.off_c0);
    const T* __restrict__ src_od = a.dyn ? a.dyn + (size_t)b * n_src : nullptr;
    const T* __restrict__ src_c = a.other_robots ? a.other_robots + (size_t)b * n_c : nullptr;
    // small blocks: elements [0, off_c0) and [off_qstc, np)
    const unsigned n_head = (unsigned)a.off_c0, n_tail = (unsigned)(a.np - a.off_qstc);
    if (vec_ok) {
        using V = Pair<T>;
        const V* vod = reinterpret_cast<const V*>(src_od);
        const V* vc = reinterpret_cast<const V*>(src_c);
        V* dod = reinterpret_cast<V*>(row + a.off_od);
        V* dc = reinterpret_cast<V*>(row + a.off_c0);
        const unsigned p_od = n_od / 2, p_src = n_src / 2, p_c = n_c / 2;
        V r_od[kFillMaxPairs], r_c[kFillMaxPairs];
        T r_small[2];
#pragma unroll
        for (int j = 0; j < kFillMaxPairs; ++j) { // loads first ...
            const unsigned i = j * kFillThreads + tid;
            r_od[j] = V{T(0), T(0)};
            if (i < p_src) r_od[j] = vod[i];
            r_c[j] = V{T(0), T(0)};
            if (src_c && i < p_c) r_c[j] = vc[i];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned i = j * kFillThreads + tid;
            r_small[j] = T(0);
            if (i < n_head + n_tail) r_small[j] = asm_small_element(a, b, i < n_head ? i : i - n_head + a.off_qstc);
        }
#pragma unroll
        for (int j = 0; j < kFillMaxPairs; ++j) { // ... then stores
            const unsigned i = j * kFillThreads + tid;
            if (i < p_od) dod[i] = r_od[j];
            if (i < p_c) dc[i] = r_c[j];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned i = j * kFillThreads + tid;
            if (i < n_head + n_tail) row[i < n_head ? i : i - n_head + a.off_qstc] = r_small[j];
        }
        // (rows larger than the register budget above: the remainder element by element)
        for (unsigned i = kFillMaxPairs * kFillThreads + tid; i < p_od; i += kFillThreads) dod[i] = i < p_src ? vod[i] : V{T(0), T(0)};
        for (unsigned i = kFillMaxPairs * kFillThreads + tid; i < p_c; i += kFillThreads) dc[i] = src_c ? vc[i] : V{T(0), T(0)};
        for (unsigned i = 2 * kFillThreads + tid; i < n_head + n_tail; i += kFillThreads) {
            const unsigned e = i < n_head ? i : i - n_head + a.off_qstc;
            row[e] = asm_small_element(a, b, e);
        }
    } else { // unaligned block boundaries (odd N ...): element by element
        for (unsigned i = tid; i < n_od; i += kFillThreads) row[a.off_od + i] = i < n_src ? src_od[i] : T(0);
        for (unsigned i = tid; i < n_c; i += kFillThreads) row[a.off_c0 + i] = src_c ? src_c[i] : T(0);
        for (unsigned i = tid; i < n_head + n_tail; i += kFillThreads) {
            const unsigned e = i < n_head ? i : i - n_head + a.off_qstc;
            row[e] = asm_small_element(a, b, e);
        }
    }
}

// Fused assembly kernel, one workgroup of four wavefronts per instance: wavefront 0 runs the (latency-bound) static-
// obstacle selection while wavefronts 1..3 move the bytes of the rest of the row -- the two parts write disjoint blocks
// of the row, so they simply overlap.
template <typename T>
__global__ __launch_bounds__(kAsmThreads) void assemble_kernel(AsmParams<T> a, unsigned vec_ok)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (threadIdx.x < 64)
        select_static(a, smem_raw);
    else
        fill_row<T, kAsmThreads - 64>(a, vec_ok, threadIdx.x - 64);
}

} // namespace nmpc
