// nmpc_step.h -- the closed-loop time step around the solve ("next" row f3): two kernels, one wavefront per scenario.
//
// Replaces, for B warehouse scenarios in lock-step and without leaving HBM (reference = /root/reference/src):
//   loop_pre  (before nmpc_assemble_params + nmpc_solve_batch)
//     main_base.py:238-264 run_cv_prediction + interfaces/cvmp_interface.py:24-57  constant-velocity prediction from
//                           the last <= 5 positions; obstacle rows [mu_x, mu_y, std_x, std_y, 0, 1]  (main_base.py:293-302)
//     pkg_mpc_tracker/trajectory_tracker.py:242-270 get_ref_states   sliding-window closest index + N reference rows
//     trajectory_tracker.py:304-310                                  speed reference (the `max` quirk included)
//   loop_post (after the solve)
//     main_base.py:320-324    no-backward clip of the first action
//     basic_agent.py:52-82    Robot.one_step (unicycle RK4, closed form) / Human.run_step (way-point following,
//                             stagger, past trajectory grows only while moving)
//     main_pre.py:34-53       clearance to pedestrians / static polygons, deviation from the reference trajectory
//     main_base.py:326-335, 366-371, 407-410   collision / completion flags
// In the evaluator these were ~100 small torch launches per time step (75 % of a B = 1 step was host-side launch time);
// the arithmetic below follows evaluate.py's torch expressions term by term (pinned by tests/golden/evaluate_cases.json).
#pragma once

#include <hip/hip_runtime.h>

#include "nmpc_assemble.h"
#include "wave_ops.h"

namespace nmpc {

template <typename T>
struct LoopParams {
    int B, n_run, N, H, W, Lmax, M, step, max_steps;
    const long long* run; // [n_run] indices of the running scenarios (ascending) or nullptr = all B
    T ts, base_speed, lin_vel_max, human_size, human_vmax;
    // state, leading dimension B
    T* robot;             // [B][3]
    T* last_u;            // [B][2]
    T* humans;            // [B][H][2]
    T* hist;              // [B][H][5][2] last <= 5 positions, newest last
    long long* hcount;    // [B][H]
    long long* hidx;      // [B][H]
    const T* hpath;       // [B][H][W][2]
    const T* ref_traj;    // [B][Lmax][3]
    const long long* ref_len; // [B]
    long long* idx_ref;   // [B]
    const T* goal;        // [B][2]
    const T* polys;       // [M][4][2]
    const T* stagger;     // [B][H] or nullptr
    unsigned char *alive, *collision, *complete; // [B]
    long long* steps;     // [B]
    T *clr_dyn, *clr_stc, *dev_sum, *dev_max, *n_traj; // [B]
    T* traj;              // [B][max_steps + 1][3]
    T* acts;              // [B][max_steps][2]
    // compact per-step buffers, leading dimension n_run
    T *state_c, *last_u_c, *refs_c, *speed_c, *dyn_c; // -> nmpc_assemble_params
    T *U_c, *y_c;         // solver in / out of the running scenarios
    T *U, *y;             // [B][2N] full copies
    int gather_y;         // pre: y_c[a] = y[b]
    // hypothesis fan (nmpc_loop_args::n_hyp): n_hyp <= 1 = the reference's one row per pedestrian
    int n_hyp;
    T hyp_fan, hyp_r0, hyp_grow;
};

// ---- before the solve -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void loop_pre_kernel(LoopParams<T> p)
{
    const int a = blockIdx.x, lane = threadIdx.x;
    const int b = p.run ? (int)p.run[a] : a;
    const int N = p.N, H = p.H;
    const T rx = p.robot[3 * b], ry = p.robot[3 * b + 1], rth = p.robot[3 * b + 2];
    // constant-velocity prediction: mean of the last <= 4 position differences, extrapolated 0..N steps
    // (n_hyp > 1: each pedestrian fans out into n_hyp rows around that step -- SURVEY.md 8(d) configs[2], nmpc_hip.h)
    const int nh = p.n_hyp > 1 ? p.n_hyp : 1;
    for (int e = lane; e < H * nh * (N + 1); e += 64) {
        const int row_i = e / (N + 1), t = e - row_i * (N + 1);
        const int h = row_i / nh, j = row_i - h * nh;
        const T* hs = p.hist + ((size_t)b * H + h) * 10;
        long long nd = p.hcount[(size_t)b * H + h] - 1;
        nd = nd < 0 ? 0 : nd > 4 ? 4 : nd;
        T vx = 0, vy = 0;
        for (int k = 0; k < 4; ++k) {
            const T m = k >= 4 - (int)nd ? T(1) : T(0);
            vx += (hs[2 * (k + 1)] - hs[2 * k]) * m;
            vy += (hs[2 * (k + 1) + 1] - hs[2 * k + 1]) * m;
        }
        const T den = T(nd > 1 ? nd : 1);
        vx /= den;
        vy /= den;
        T* row = p.dyn_c + (((size_t)a * H * nh + row_i) * (N + 1) + t) * 6;
        if (nh > 1) {
            T sa, ca;
            tsincos((T(j) - T(0.5) * T(nh - 1)) * p.hyp_fan, sa, ca);
            const T wx = ca * vx - sa * vy, wy = sa * vx + ca * vy;
            row[0] = p.humans[((size_t)b * H + h) * 2] + wx * T(t);
            row[1] = p.humans[((size_t)b * H + h) * 2 + 1] + wy * T(t);
            row[2] = row[3] = p.hyp_r0 + p.hyp_grow * T(t);
        } else {
            row[0] = p.humans[((size_t)b * H + h) * 2] + vx * T(t);
            row[1] = p.humans[((size_t)b * H + h) * 2 + 1] + vy * T(t);
            row[2] = row[3] = t == 0 ? p.human_size : T(1);
        }
        row[4] = 0;
        row[5] = 1;
    }
    // reference window: first minimum of the distance over [idx - N, min(len, idx + 5N))
    const long long idx0 = p.idx_ref[b], len = p.ref_len[b];
    const long long lo = idx0 - N > 0 ? idx0 - N : 0, hi = len < idx0 + 5 * N ? len : idx0 + 5 * N;
    const T* rt = p.ref_traj + (size_t)b * p.Lmax * 3;
    T best = T(INFINITY);
    int bj = 0x7fffffff;
    for (int j = lane; j < p.Lmax; j += 64) {
        if (j >= lo && j < hi) {
            const T d = thypot(rx - rt[3 * j], ry - rt[3 * j + 1]);
            if (d < best) {
                best = d;
                bj = j;
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) { // (value, index) minimum, lowest index among equals
        const T ob = __shfl_xor(best, s, 64);
        const int oj = __shfl_xor(bj, s, 64);
        if (ob < best || (ob == best && oj < bj)) {
            best = ob;
            bj = oj;
        }
    }
    const long long idx = bj == 0x7fffffff ? 0 : bj; // (empty window: torch.argmin of an all-inf row is 0)
    if (lane == 0) p.idx_ref[b] = idx;
    for (int e = lane; e < 3 * N; e += 64) {
        const int i = e / 3, c = e - 3 * i;
        long long r = idx + i;
        r = r < len - 1 ? r : len - 1;
        r = r < 0 ? 0 : r < p.Lmax ? r : p.Lmax - 1; // (device-side ref_len outside [1, Lmax]: stay inside the row)
        p.refs_c[(size_t)a * 3 * N + e] = rt[3 * r + c];
    }
    if (lane == 0) {
        const T dg = thypot(rx - p.goal[2 * b], ry - p.goal[2 * b + 1]);
        const bool near = dg < p.base_speed * T(N) * p.ts;
        const T sp = dg / T(N) / p.ts;
        p.speed_c[a] = near ? (sp > p.lin_vel_max ? sp : p.lin_vel_max) : p.base_speed; // sic: max(), trajectory_tracker.py:308-309
        p.state_c[3 * a] = rx, p.state_c[3 * a + 1] = ry, p.state_c[3 * a + 2] = rth;
        p.last_u_c[2 * a] = p.last_u[2 * b], p.last_u_c[2 * a + 1] = p.last_u[2 * b + 1];
    }
    if (p.gather_y && p.run)
        for (int e = lane; e < 2 * N; e += 64) p.y_c[(size_t)a * 2 * N + e] = p.y[(size_t)b * 2 * N + e];
}

// ---- after the solve --------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_min(T x)
{
    for (int s = 32; s > 0; s >>= 1) {
        const T o = __shfl_xor(x, s, 64);
        x = o < x ? o : x;
    }
    return x;
}

template <typename T>
__global__ __launch_bounds__(64) void loop_post_kernel(LoopParams<T> p)
{
    const int a = blockIdx.x, lane = threadIdx.x;
    const int b = p.run ? (int)p.run[a] : a;
    const int N = p.N, H = p.H, W = p.W;
    const bool alive = p.alive[b] != 0;
    // results of the running scenarios back into the full arrays (finished scenarios keep their last rows)
    if (p.run)
        for (int e = lane; e < 2 * N; e += 64) {
            p.U[(size_t)b * 2 * N + e] = p.U_c[(size_t)a * 2 * N + e];
            p.y[(size_t)b * 2 * N + e] = p.y_c[(size_t)a * 2 * N + e];
        }
    const T rv = p.U_c[(size_t)a * 2 * N], rw = p.U_c[(size_t)a * 2 * N + 1];
    const T av = rv < T(0) ? T(0) : rv, aw = rv < T(0) ? T(0) : rw; // no-backward, main_base.py:320-321
    T x = p.robot[3 * b], y = p.robot[3 * b + 1], th = p.robot[3 * b + 2];
    if (alive) { // Robot.one_step: unicycle RK4 in closed form
        const T hh = T(0.5) * p.ts * aw;
        T s0, c0, s1, c1, s2, c2;
        tsincos(th, s0, c0);
        tsincos(th + hh, s1, c1);
        tsincos(th + T(2) * hh, s2, c2);
        const T cc = (c0 + T(4) * c1 + c2) / T(6), ss = (s0 + T(4) * s1 + s2) / T(6);
        x = x + p.ts * av * cc;
        y = y + p.ts * av * ss;
        th = th + p.ts * aw;
    }
    // pedestrians (one lane each): way-point following with stagger; the past trajectory grows only while moving
    T dd = T(INFINITY);
    if (lane < H) {
        const size_t hb = (size_t)b * H + lane;
        T hx = p.humans[2 * hb], hy = p.humans[2 * hb + 1];
        long long hi = p.hidx[hb];
        const T* path = p.hpath + hb * W * 2;
        {
            const long long c = hi < W - 1 ? hi : W - 1;
            const T d = thypot(path[2 * c] - hx, path[2 * c + 1] - hy);
            if (d < p.human_vmax * p.ts && hi < W) hi += 1; // basic_agent.py:57-59
        }
        const bool moving = hi < W;
        if (moving) {
            const T tx = path[2 * hi] - hx, ty = path[2 * hi + 1] - hy;
            T d = thypot(tx, ty);
            d = d > T(1e-12) ? d : T(1e-12);
            const T st = p.stagger ? p.stagger[hb] : T(0);
            hx = hx + p.ts * (tx / d * p.human_vmax + st);
            hy = hy + p.ts * (ty / d * p.human_vmax + st);
            T* hs = p.hist + hb * 10;
            for (int k = 0; k < 8; ++k) hs[k] = hs[k + 2];
            hs[8] = hx, hs[9] = hy;
            p.hcount[hb] += 1;
            p.humans[2 * hb] = hx, p.humans[2 * hb + 1] = hy;
        }
        p.hidx[hb] = hi;
        dd = thypot(x - hx, y - hy); // main_pre.calc_minimal_dynamic_obstacle_distance
    }
    dd = wave_min(dd);
    // static polygons (one lane per polygon, strided): distance (0 inside) and strict containment
    T dpoly = T(INFINITY);
    bool inside_any = false;
    for (int m = lane; m < p.M; m += 64) {
        const T* q = p.polys + (size_t)m * 8;
        T dmin = T(INFINITY);
        bool allp = true, alln = true;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const T ax = q[2 * e], ay = q[2 * e + 1], bx = q[2 * ((e + 1) & 3)], by = q[2 * ((e + 1) & 3) + 1];
            const T abx = bx - ax, aby = by - ay;
            T t = ((x - ax) * abx + (y - ay) * aby) / (abx * abx + aby * aby);
            t = t < T(0) ? T(0) : t > T(1) ? T(1) : t;
            const T ex = ax + t * abx - x, ey = ay + t * aby - y;
            const T d = sqrt(ex * ex + ey * ey);
            dmin = d < dmin ? d : dmin;
            const T cr = abx * (y - ay) - aby * (x - ax);
            allp = allp && cr > T(0);
            alln = alln && cr < T(0);
        }
        const bool ins = allp || alln;
        inside_any = inside_any || ins;
        const T dm = ins ? T(0) : dmin;
        dpoly = dm < dpoly ? dm : dpoly;
    }
    dpoly = wave_min(dpoly);
    inside_any = __ballot(inside_any) != 0ull;
    // deviation from the reference trajectory (main_pre.calc_deviation_distance: closest point)
    const long long len = p.ref_len[b];
    const T* rt = p.ref_traj + (size_t)b * p.Lmax * 3;
    T dref = T(INFINITY);
    for (int j = lane; j < p.Lmax && j < len; j += 64) {
        const T ex = x - rt[3 * j], ey = y - rt[3 * j + 1];
        const T d = sqrt(ex * ex + ey * ey);
        dref = d < dref ? d : dref;
    }
    dref = wave_min(dref);
    if (lane == 0) {
        T* tr = p.traj + ((size_t)b * (p.max_steps + 1) + p.step + 1) * 3;
        T* ac = p.acts + ((size_t)b * p.max_steps + p.step) * 2;
        tr[0] = x, tr[1] = y, tr[2] = th;
        if (alive) {
            p.robot[3 * b] = x, p.robot[3 * b + 1] = y, p.robot[3 * b + 2] = th;
            p.last_u[2 * b] = rv, p.last_u[2 * b + 1] = rw;
            ac[0] = rv, ac[1] = rw;
            p.steps[b] += 1;
            p.clr_dyn[b] = dd < p.clr_dyn[b] ? dd : p.clr_dyn[b];
            p.clr_stc[b] = dpoly < p.clr_stc[b] ? dpoly : p.clr_stc[b];
            p.dev_sum[b] += dref;
            p.dev_max[b] = dref > p.dev_max[b] ? dref : p.dev_max[b];
            p.n_traj[b] += T(1);
            const bool col = inside_any || dd <= p.human_size;
            const T gx = p.goal[2 * b], gy = p.goal[2 * b + 1];
            const bool done = !col && tabs(x - gx) <= T(0.5) && tabs(y - gy) <= T(0.5) && tabs(av) < T(0.4);
            if (col) p.collision[b] = 1;
            if (done) p.complete[b] = 1;
            if (col || done) p.alive[b] = 0;
        }
    }
}

} // namespace nmpc
