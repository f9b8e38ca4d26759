// nmpc_spec.h -- latency mode of the solver: W wavefronts per MPC instance, speculative line search.
//
// Small batches (BASELINE configs[1]: B = 1024 = one wavefront per SIMD of an MI355X) are bound by the LATENCY of the
// slowest instance, not by throughput: a PANOC iteration is a chain of dependent psi evaluations -- one cost-only
// evaluation for the Lipschitz test and then the line-search candidates tau = 1, 1/2, 1/4, ... one after another
// (measured on configs[1]: 3.8 candidates per iteration on average, 4.9 for the slowest instances) -- while three
// quarters of the chip idle once the easy instances have finished.
//
// Here a workgroup of W wavefronts owns one instance. One of them -- the MASTER, role 0 -- runs the solver state machine;
// the others are WORKERS that evaluate the line-search candidates the master hands them (the tables in LDS are shared,
// every wavefront keeps its own copy of the register-resident obstacle rows) and sleep at a barrier in between. In the
// first round of an iteration the master evaluates psi at the half step (the Lipschitz test) while roles 1..W-1 already
// evaluate the first W-1 line-search candidates, computed under the assumption that the test passes (it does unless
// gamma has to be halved, which resets the L-BFGS buffer anyway); later rounds evaluate W candidates at a time. A round:
// the master writes every role's request to the command area, barrier (A), everybody evaluates (ONE evaluation site in
// the code, so every candidate goes through the same instructions whoever takes it), results (psi, gradient) to the
// result rows, barrier (B), the master replays the sequential acceptance logic on them in order. Each candidate is the
// same function of the same inputs as in the sequential order, the acceptance tests run in the same order on the same
// numbers, so the results do not depend on W: launched with one wavefront this kernel IS the sequential algorithm, and
// 2, 3 or 4 wavefronts give BIT-IDENTICAL results (tested); evaluations whose results are never looked at are the price
// (info[6] = exchange rounds, info[7] = W). solve_instance() -- the throughput kernel for large batches -- is a separate
// compilation of the same algorithm and agrees with this one to rounding only.
// (Rounds 2-3 ran the whole state machine in every wavefront on identical data -- W copies of the two-loop recursion, the
//  L-BFGS update and the step head per iteration, about half of a workgroup's instructions issued W times over. With the
//  roles configs[1] went from 44.8 to 47.4 k solves/s, same bits. What bounds that batch now is the serial chain of its
//  slowest instance: B = 64 ... 1024 all take 18-23 ms, profiles/r04_exp_cfg1_batch_size_and_up_to_8_wavefronts.txt.)
//
// Restates the same OpEn pieces as solve_instance (core::panoc, lbfgs, alm); see the notes there.
#pragma once

#include "nmpc_device.h"

namespace nmpc {

enum SpecPhase : int { SP_INIT_A, SP_INIT_B, SP_LIP, SP_SPEC0, SP_NOLS, SP_LSN, SP_OUTER, SP_RESUME };

// FLAT: how the obstacle passes of an evaluation are written. true (the latency family proper): the pair form WITHOUT its
// early-outs -- straight-line code whose LDS reads and DPP chains overlap, what a lone instance wants. false (round 6, the
// TAIL member): the throughput kernels' gated form, i.e. the very evaluation solve_instance() runs. That form is the ONLY
// numerical difference between the two families: with it this kernel returns the throughput kernels' bits (every row of
// profiles/r06_family_bits.txt, fp32 and fp64), which is what lets a launch of the throughput kernel hand its longest
// instances -- parked in its drain phase -- to the speculative line search without an instance's result depending on what
// else is in its batch (nmpc_config.tail_latency; nmpc_capi.hip, run_solve).
template <typename T, int LPS, bool GLB, int RS = 0, bool AXIS = false, bool FLAT = true>
__device__ __forceinline__ void solve_instance_spec(const KParams<T>& kp, const int inst, T* lds)
{
    // W is a run-time value (workgroup size / 64, at most kMaxSpecWaves) so that every choice of W executes the very
    // same machine code: with the compiler's default contraction (-ffp-contract=fast) two instantiations of this
    // function would fuse multiplies and adds differently and agree only to rounding.
    const int W = (int)(blockDim.x >> 6);
    const T GAMMA_L = T(0.95), LIP_EPS_UPD = T(1e-6), MIN_L = T(1e-10), MAX_L = T(1e9);
    const int MAX_LIP = 10, MAX_LS = 10;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // ROLES (round 4). One wavefront of the workgroup -- the master, role 0 -- runs the solver; the others are workers
    // that evaluate the line-search candidates it hands them and sleep at a barrier in between. (Rounds 2-3: every
    // wavefront ran the whole state machine on identical data and replayed the acceptance logic itself -- W copies of
    // the two-loop recursion, the L-BFGS update, the step head: half of a workgroup's instructions, issued W times. A
    // batch that oversubscribes the SIMDs pays for them with the critical path of every other instance.) The master is
    // wavefront inst % W, so that the always-busy wavefronts of co-resident workgroups do not pile up on one SIMD.
    int role = wave - inst % W;
    if (role < 0) role += W;
    role = __builtin_amdgcn_readfirstlane(role);

    Instance<T, LPS, GLB, RS, false, false, AXIS> I(kp, kp.P + (size_t)inst * kp.np, lds, GLB ? kp.ws + (long long)inst * kp.ws_stride : nullptr);
    if (const int bad = I.load()) { // NMPC_CAPACITY_EXCEEDED / NMPC_NOT_AXIS_ALIGNED (uniform over the workgroup)
        const T nan = __builtin_nanf("");
        const auto* kc = cold_args<T>();
        if (wave == 0) {
            if (I.lead) {
                kc->U[(size_t)inst * 2 * kp.N + 2 * I.k] = nan;
                kc->U[(size_t)inst * 2 * kp.N + 2 * I.k + 1] = nan;
            }
            if (I.lane == 0) {
                if (kc->cost) kc->cost[inst] = nan;
                if (kc->status) kc->status[inst] = bad;
                if (kc->iters) kc->iters[2 * inst] = kc->iters[2 * inst + 1] = 0;
                if (kc->info)
                    for (int i = 0; i < 8 + kProfSlots; ++i) kc->info[(size_t)inst * (8 + kProfSlots) + i] = 0;
            }
        }
        return;
    }
    const int N = kp.N, n2 = kp.Ndyn;
    const int mem = kp.mem;
    const int kk = I.act ? I.k : 0;
    const bool lead = I.lead;

    T uv = 0, uw = 0, yv = 0, yw = 0;
    T c;
    {
        const auto* kc = cold_args<T>();
        if (kc->u0) {
            uv = kc->u0[(size_t)inst * 2 * N + 2 * kk];
            uw = kc->u0[(size_t)inst * 2 * N + 2 * kk + 1];
        }
        if (kc->y && kc->y_is_input) {
            yv = kc->y[(size_t)inst * 2 * N + kk];
            yw = kc->y[(size_t)inst * 2 * N + N + kk];
        }
        c = kc->c0v ? kc->c0v[inst] : kc->c_init;
    }
    if (!I.act) uv = uw = yv = yw = 0;
    const int resuming = __builtin_amdgcn_readfirstlane(cold_args<T>()->stage_in);
    if (resuming && cold_args<T>()->status[inst] != -1) return; // finished in an earlier launch (uniform)

    // PANOC cache (same names as solve_instance)
    T gv = 0, gw = 0, gpv = 0, gpw = 0, hv = 0, hw = 0, sv = 0, sw = 0, dv = 0, dw = 0, fv = 0, fw = 0, pv = 0, pw = 0;
    T gamma = 0, inv_gamma = 0, L = 0, sigma = 0, tau = 1, cost_value = 0, norm_fpr = 0, rhs_ls = 0;
    T akkt_tol = cold_args<T>()->init_tol;
    int iteration = 0;
    Quad<T>* hist = reinterpret_cast<Quad<T>*>(lds + kp.lds_hist);
    LaneVec<T> rho; // rho_i by physical slot (every wavefront keeps its own, identical copy)
    T osv = 0, osw = 0, ogv = 0, ogw = 0, lb_gamma = 1;
    int lb_active = 0, lb_head = 0;
    int lb_first = 1; // (the wave-uniform flags are ints: a bool would live in an SGPR PAIR as a lane mask)
    T dyn = 0, dyn_plus = 0, f2n = 0, f2n_plus = 0;
    int alm_iter = 0, inner_total = 0, outer = 1, status = 0;
    int num_iter = 0, lip_it = 0, nls = 0;
    int cont = 1;
    T cost_half = 0, normh = 0, fbe_cur = 0;
    int fbe_valid = 0;
    // Evaluations the sequential algorithm performs (gradient evaluations; exchange rounds). The psi evaluations are counted
    // DOWN from the evaluation budget (nmpc_config.max_evaluations; none = INT_MAX): the budget test of every inner iteration
    // is then a compare against zero and needs no scalar register of its own round the loop (a live budget next to the
    // count cost the configs[1] kernel 1-3 %: profiles/r06_ab_evaluation_budget.jsonl). alg_psi = budget - evals_left.
    auto evals_budget = [&]() {
        const int me = cold_args<T>()->max_evals;
        return me > 0 ? me : 0x7fffffff;
    };
    int evals_left = evals_budget();
    int alg_grad = 0, rounds = 0;
    // max_solver_time: wavefront 0 publishes its elapsed real-time ticks with every exchange round, so that all
    // wavefronts of the workgroup take the same decision (see solve_instance)
    const long long time_budget = cold_args<T>()->time_budget;
    const long long t_start = time_budget > 0 ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
    int t_now = 0;
    int cont_time = 1;
    int deep_slot = 0; // > 0: the instance was parked inside an inner solve (KParams::deep)

    // exchange area: 2 buffers x W wavefronts x (64 lanes x (g_v, g_w) + psi, padded to 4). Per lane, not per step:
    // the LPS lanes of a step hold copies of the step's scalars that may differ in the last bit (each lane's
    // suffix sum associates differently), and every lane must get exactly what it would have computed itself.
    const int XS = 2 * 64 + 4;
    T* xch = lds + kp.lds_xch;    // results: one row per role
    // commands of the master: y (2 x 64), then (ev, ew) of roles 1..W-1 (2 x 64 each), then the scalars ec, 1/max(c,1) and
    // two ints: bit r = role r evaluates, exit flag. (Addresses formed where they are used, from an opaque W: kept in
    // scalar registers for the whole solve they displaced solver state into v_writelane / v_readlane traffic.)
    auto cmd_area = [&]() {
        int Wv = W;
        asm volatile("" : "+s"(Wv));
        return xch + Wv * XS;
    };

    const T vlo = kp.vmin, vhi = kp.vmax, wlo = -kp.wmax, whi = kp.wmax;
    auto project = [&](T& a, T& b) {
        a = tclamp(a, vlo, vhi);
        b = tclamp(b, wlo, whi);
    };
    auto grad_step_half = [&](T bv, T bw) {
        sv = bv - gamma * gv;
        sw = bw - gamma * gw;
        hv = sv;
        hw = sw;
        project(hv, hw);
    };
    auto reset_cache = [&]() {
        lb_active = 0;
        lb_first = true;
        rhs_ls = 0;
        tau = 1;
        L = 0;
        sigma = 0;
        cost_value = 0;
        iteration = 0;
        gamma = 0;
        inv_gamma = 0;
        fbe_valid = false;
    };

    if (resuming) {
        // resumable solve (KParams::resume): the state the instance parked behind the ALM update of its last outer
        // iteration; every wavefront of the workgroup loads the same values
        const T* rs = cold_args<T>()->resume + (size_t)inst * kResumeStride;
        uv = rs[0 * 64 + I.lane], uw = rs[1 * 64 + I.lane], yv = rs[2 * 64 + I.lane], yw = rs[3 * 64 + I.lane];
        gpv = rs[4 * 64 + I.lane], gpw = rs[5 * 64 + I.lane];
        const T* sc = rs + 6 * 64;
        c = sc[0];
        akkt_tol = sc[1];
        dyn = sc[2];
        f2n = sc[3];
        alm_iter = __builtin_amdgcn_readfirstlane((int)sc[4]);
        outer = __builtin_amdgcn_readfirstlane((int)sc[5]) + 1;
        inner_total = __builtin_amdgcn_readfirstlane((int)sc[6]);
        evals_left -= __builtin_amdgcn_readfirstlane((int)sc[7]);
        alg_grad = __builtin_amdgcn_readfirstlane((int)sc[8]);
        rounds = __builtin_amdgcn_readfirstlane((int)sc[9]);
        yv = tclamp(yv, T(-1e12), T(1e12));
        yw = tclamp(yw, T(-1e12), T(1e12));
        if constexpr (!FLAT) deep_slot = __builtin_amdgcn_readfirstlane((int)sc[10]); // (the tail member only: nothing else is handed deep parks)
        if (deep_slot > 0) {
            // parked INSIDE an inner solve by the throughput kernel (tail hand-off, KParams::deep): the whole solver state as it
            // stood between two PANOC iterations; the loop below is entered at the head of the iteration that comes next
            // (SP_RESUME). Every wavefront restores its registers; the ring in LDS belongs to the master.
            const T* g = cold_args<T>()->deep + (size_t)(deep_slot - 1) * deep_park_stride(N);
            const Quad<T>* pq = reinterpret_cast<const Quad<T>*>(g) + I.lane;   // park() layout: [quad][lane]
            const Quad<T> q0 = pq[0 * 64], q1 = pq[1 * 64], q3 = pq[3 * 64], q4 = pq[4 * 64];
            osv = q0.a, osw = q0.b, ogv = q0.c, ogw = q0.d;
            sv = q1.c, sw = q1.d;
            hv = q3.a, hw = q3.b;
            uv = q4.a, uw = q4.b, gv = q4.c, gw = q4.d;
            g += kParkQuads * 64 * 4;
            const int nr = deep_ring_elems(N);
            if (role == 0) {
                T* ring = reinterpret_cast<T*>(hist);
                for (int i = I.lane; i < nr; i += 64) ring[i] = g[i];
            }
            g += nr;
            yv = g[I.lane], yw = g[64 + I.lane], rho.v = g[128 + I.lane];
            const T* ds = g + 192;
            gamma = ds[kDeepScGamma], inv_gamma = ds[kDeepScInvGamma], L = ds[kDeepScL], cost_value = ds[kDeepScCost], fbe_cur = ds[kDeepScFbe];
            akkt_tol = ds[kDeepScAkkt], c = ds[kDeepScC], lb_gamma = ds[kDeepScLbGamma], dyn = ds[kDeepScDyn], f2n = ds[kDeepScF2n];
            auto rd = [&](int j) { return __builtin_amdgcn_readfirstlane((int)ds[j]); };
            iteration = rd(kDeepScIteration), num_iter = rd(kDeepScNumIter), cont = rd(kDeepScCont), cont_time = rd(kDeepScContTime);
            fbe_valid = rd(kDeepScFbeValid), lb_active = rd(kDeepScLbActive), lb_head = rd(kDeepScLbHead), lb_first = rd(kDeepScLbFirst);
            alm_iter = rd(kDeepScAlmIter), inner_total = rd(kDeepScInnerTotal), outer = rd(kDeepScOuter), alg_grad = rd(kDeepScNGrad);
            rounds = 0;
            evals_left = evals_budget() - rd(kDeepScNPsi);
        }
    }
    // The solver options the master consults in EVERY iteration, read once (round 5): re-read from the kernel argument at the
    // point of use -- what the throughput kernels do to keep their scalar registers free -- each of them is a scalar load
    // whose ~200-cycle round trip the lone master wavefront waits for with nobody to fill it (three per iteration); a
    // scalar register that ends up spilled to a VGPR lane costs one v_readlane instead.
    const T o_tol = kp.tol, o_sy_eps = kp.sy_eps, o_cbfgs_eps = kp.cbfgs_eps, o_cbfgs_alpha = kp.cbfgs_alpha;
    const int o_max_inner = kp.max_inner, o_akkt_form = kp.akkt_form;
    int phase = deep_slot > 0 ? SP_RESUME : SP_INIT_A;
    T ev = uv, ew = uw, ec = c;
    T inv_cdiv = T(1) / (c > T(1) ? c : T(1));
    int want_grad = 1, do_eval = deep_slot > 0 ? 0 : 1, exchange = 0;
    T r_psi = 0, r_f2 = 0, r_gv = 0, r_gw = 0;

    // every wavefront evaluates the same request (results used locally, no exchange)
    auto request_uniform = [&](T a, T b, T cc, bool grad) {
        ev = a;
        ew = b;
        ec = cc;
        want_grad = grad;
        do_eval = true;
        exchange = false;
    };
    // line-search candidates nls, nls+1, ... for roles first, first+1, ... (tau is the candidate nls's step): the master's
    // own request, the workers' through the command area, then the barrier the workers wait at
    auto request_candidates = [&](int first) {
        exchange = true;
        if (first == 0) {
            do_eval = nls <= MAX_LS; // candidate MAX_LS is accepted unconditionally, nothing lies beyond it
            ev = ls_point(uv, fv, dv, tau);
            ew = ls_point(uw, fw, dw, tau);
            ec = c;
            want_grad = true;
        }
        T* const cmd = cmd_area();
        T* const cmds = cmd + 2 * 64 * W;
        T tw = tau;
        int flags = 0;
        for (int r = 1; r < W; ++r) {
            if (r > first) tw *= T(0.5);
            cmd[2 * 64 * r + I.lane] = ls_point(uv, fv, dv, tw);
            cmd[2 * 64 * r + 64 + I.lane] = ls_point(uw, fw, dw, tw);
            if (nls + (r - first) <= MAX_LS) flags |= 1 << r;
        }
        cmd[I.lane] = yv;
        cmd[64 + I.lane] = yw;
        if (I.lane == 0) {
            cmds[0] = c;
            cmds[1] = inv_cdiv;
            reinterpret_cast<int*>(cmds + 2)[0] = flags;
            reinterpret_cast<int*>(cmds + 2)[1] = 0;
            cmds[4] = gamma;     // (the candidates' FBE is formed by whoever evaluates them, below)
            cmds[5] = inv_gamma;
        }
        __syncthreads(); // (A) the workers start
    };
    // the master is done (finished, or parked at the stage boundary): release the workers
    auto dismiss_workers = [&]() {
        T* const cmds = cmd_area() + 2 * 64 * W;
        if (I.lane == 0) reinterpret_cast<int*>(cmds + 2)[1] = 1;
        __syncthreads();
    };

    // The integer / boolean solver state is the same in every lane by construction; the tests on floating-point values
    // are declared uniform where they are made (one ballot each), so that it lives in scalar registers all the way round
    // the loop -- see solve_instance (nmpc_device.h).
    auto anyb = [](bool cond) { return __builtin_amdgcn_ballot_w64(cond) != 0ull; };
    // bit 0: wall-clock budget (max_solver_time_us), bit 1: evaluation budget (max_evaluations) -- one flag, so that the
    // default path (neither) carries no further scalar register round the loop and reads nothing per iteration
    const int timed_ = __builtin_amdgcn_readfirstlane(time_budget > 0 ? 1 : 0);
    for (;;) {
        // wavefront index / budget flag as opaque per-round values: conditions on them are then evaluated where they
        // are used (s_cmp) instead of living in SGPR pairs as loop-invariant lane masks for the whole solve
        int wv = role, timed = timed_;
        asm volatile("" : "+s"(wv), "+s"(timed));
        if (wv != 0) { // worker: sleep until the master has published a round, take this role's request
            __syncthreads(); // (A)
            const T* const cmd = cmd_area();
            const T* const cmds = cmd + 2 * 64 * W;
            if (__builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(cmds + 2)[1]) != 0) return;
            do_eval = (__builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(cmds + 2)[0]) >> wv) & 1;
            ev = cmd[2 * 64 * wv + I.lane];
            ew = cmd[2 * 64 * wv + 64 + I.lane];
            yv = cmd[I.lane];
            yw = cmd[64 + I.lane];
            ec = c = cmds[0];
            inv_cdiv = cmds[1];
            gamma = cmds[4];
            inv_gamma = cmds[5];
            want_grad = 1;
            exchange = 1;
        }
        const T e_icd = ec == c ? inv_cdiv : T(1);
        if (do_eval) {
            if (want_grad)
                I.template eval<true, FLAT>(ev, ew, ec, e_icd, yv, yw, r_psi, r_f2, r_gv, r_gw);
            else
                I.template eval<false, FLAT>(ev, ew, ec, e_icd, yv, yw, r_psi, r_f2, r_gv, r_gw);
        }
        // The forward-backward envelope of a line-search candidate -- the left-hand side of its acceptance test -- by the
        // wavefront that evaluated it (round 5): psi - gamma/2 ||grad||^2 + ||gradient_step - Proj_U(gradient_step)||^2 / (2 gamma)
        // at the candidate (ev, ew). ONE code site for every role, the master included, so a candidate's value does not
        // depend on who evaluated it (nor on W). Rounds 2-4 had the master recompute this for every candidate it replayed:
        // two LDS reads, the projection and two wave reductions per candidate, ~4.5 candidates per iteration on its serial path.
        T fbe = 0;
        if (exchange && do_eval && want_grad) {
            T s1, s2;
            if constexpr (sizeof(T) == 4) {
                s1 = __builtin_fmaf(-gamma, r_gv, ev);
                s2 = __builtin_fmaf(-gamma, r_gw, ew);
            } else {
                s1 = __builtin_fma(-gamma, r_gv, ev);
                s2 = __builtin_fma(-gamma, r_gw, ew);
            }
            T h1 = s1, h2 = s2;
            project(h1, h2);
            const T t1 = s1 - h1, t2 = s2 - h2;
            T d2, gg;
            I.dot2x2(t1, t2, t1, t2, r_gv, r_gw, r_gv, r_gw, d2, gg);
            fbe = r_psi - T(0.5) * gamma * gg + T(0.5) * d2 * inv_gamma;
        }
        NMPC_STAMP(I, 7); // (eval epilogue)
        const T* xr = xch; // results of this round, one row per role
        if (exchange) {
            T* xw = xch + wv * XS;
            if (do_eval) {
                if (want_grad) {
                    xw[2 * I.lane] = r_gv;
                    xw[2 * I.lane + 1] = r_gw;
                }
                if (I.lane == 0) {
                    xw[2 * 64] = r_psi;
                    xw[2 * 64 + 1] = fbe;
                }
            }
            __syncthreads(); // (B) the round's results are in; the workers go back to (A), which also keeps them from
                             //     overwriting these rows before the master has replayed the round
            if (timed) { // the clock is read once per exchange round, as before
                long long el = (long long)__builtin_amdgcn_s_memrealtime() - t_start;
                if (el > 0x7fffffffll) el = 0x7fffffffll;
                t_now = (int)el;
            }
            rounds++;
        }
        if (wv != 0) continue;
        NMPC_STAMP(I, 8); // exchange: LDS writes + barrier (= waiting for the slowest wavefront of the round)

        // sequential line-search logic replayed on the exchanged results of wavefronts first..W-1
        auto process_candidates = [&](int first) -> bool {
            for (int w = first; w < W; ++w) {
                evals_left--;
                alg_grad++;
                const T lhs = xr[w * XS + 2 * 64 + 1]; // the candidate's FBE, formed by the wavefront that evaluated it
                if (anyb(lhs > rhs_ls) && nls < MAX_LS) {
                    tau *= T(0.5);
                    nls++;
                    continue;
                }
                // accepted: the candidate becomes the iterate -- its point, cost, gradient, gradient step and half step
                pv = ls_point(uv, fv, dv, tau);
                pw = ls_point(uw, fw, dw, tau);
                cost_value = xr[w * XS + 2 * 64];
                gv = xr[w * XS + 2 * I.lane];
                gw = xr[w * XS + 2 * I.lane + 1];
                grad_step_half(pv, pw);
                fbe_cur = lhs;
                fbe_valid = true;
                uv = pv;
                uw = pw;
                return true;
            }
            return false;
        };

        bool step_head = false, step_done = false;
        bool need_dir = false, spec = false; // run the L-BFGS update / direction block below

        if (phase == SP_INIT_A) {
            evals_left--;
            alg_grad++;
            cost_value = r_psi;
            gv = r_gv;
            gw = r_gw;
            const auto* kc = cold_args<T>();
            const T lip_eps = kc->lip_eps, lip_delta = kc->lip_delta;
            const T e1 = lip_eps * uv, e2 = lip_eps * uw;
            const T h1 = e1 > lip_delta ? e1 : lip_delta, h2 = e2 > lip_delta ? e2 : lip_delta;
            normh = tsqrt(I.dot2(h1, h2, h1, h2));
            if (I.act) {
                uv += h1;
                uw += h2;
            }
            request_uniform(uv, uw, c, true);
            phase = SP_INIT_B;
            continue;
        } else if (phase == SP_INIT_B) {
            evals_left--;
            alg_grad++;
            const T d1 = r_gv - gv, d2 = r_gw - gw;
            L = tsqrt(I.dot2(d1, d2, d1, d2)) / normh;
            gamma = GAMMA_L / tmax(L, MIN_L);
            inv_gamma = T(1) / gamma;
            sigma = (T(1) - GAMMA_L) * T(0.25) * inv_gamma;
            cost_value = r_psi;
            grad_step_half(uv, uw);
            step_head = true;
        } else if (phase == SP_LIP || phase == SP_SPEC0) {
            // SP_LIP: every wavefront evaluated psi(u_half) itself; SP_SPEC0: wavefront 0 did
            cost_half = phase == SP_SPEC0 ? xr[2 * 64] : r_psi;
            evals_left--;
            const T ip = I.dot2(gv, gw, fv, fw);
            const T rhs = cost_value + LIP_EPS_UPD * tabs(cost_value) - ip +
                          (GAMMA_L * T(0.5) * inv_gamma) * (norm_fpr * norm_fpr);
            if (anyb(cost_half > rhs && L < MAX_L) && lip_it < MAX_LIP) {
                // gamma is halved: the speculative candidates (if any) are void, continue sequentially
                lb_active = 0;
                lb_first = true;
                L *= T(2);
                gamma *= T(0.5);
                inv_gamma *= T(2);
                fbe_valid = false;
                grad_step_half(uv, uw);
                fv = uv - hv;
                fw = uw - hw;
                norm_fpr = tsqrt(I.dot2(fv, fw, fv, fw));
                lip_it++;
                request_uniform(hv, hw, c, false);
                phase = SP_LIP;
                continue;
            }
            NMPC_STAMP(I, 9); // Lipschitz test
            if (phase == SP_SPEC0) {
                // the direction and rhs_ls were computed before the round; candidates 0..W-2 are on wavefronts 1..
                if (process_candidates(1)) {
                    step_done = true;
                } else {
                    request_candidates(0);
                    NMPC_STAMP(I, 15); // (diagnostic: a round without an acceptable candidate -- replay + the next requests + barrier A)
                    phase = SP_LSN;
                    continue;
                }
            } else {
                need_dir = true; // sequential path: L-BFGS update and direction now
            }
        } else if (phase == SP_NOLS) {
            evals_left--;
            alg_grad++;
            fbe_valid = false;
            cost_value = r_psi;
            gv = r_gv;
            gw = r_gw;
            grad_step_half(uv, uw);
            step_done = true;
        } else if (phase == SP_RESUME) {
            step_head = true; // (deep resume: nothing was evaluated; the head of the next iteration on the restored state)
        } else if (phase == SP_LSN) {
            if (process_candidates(0)) {
                step_done = true;
            } else {
                request_candidates(0);
                NMPC_STAMP(I, 15);
                continue; // stay in SP_LSN
            }
        } else { // SP_OUTER
            evals_left--;
            const auto* kc = cold_args<T>();
            const T f_u = r_psi;
            f2n_plus = tsqrt(r_f2);
            const T v_up = wave_shift_up<LPS>(uv), w_up = wave_shift_up<LPS>(uw);
            const T vprev = I.k == 0 ? I.p[0] : v_up;
            const T wprev = I.k == 0 ? I.p[1] : w_up;
            const T acc = (uv - vprev) * kc->inv_ts, wacc = (uw - wprev) * kc->inv_ts;
            const T za = acc + yv / c, zw = wacc + yw / c;
            T ypv = yv + c * (acc - tclamp(za, kc->amin, kc->amax));
            T ypw = yw + c * (wacc - tclamp(zw, -kc->wamax, kc->wamax));
            if (!I.act) ypv = ypw = 0;
            const T e1 = ypv - yv, e2 = ypw - yw;
            dyn_plus = tsqrt(I.dot2(e1, e2, e1, e2));
            const T SMALL = Lim<T>::eps;
            const int single = kc->single_inner; // (polish: see KParams::single_inner; `status` = how the inner solve ended)
            const bool c1 = single || (alm_iter > 0 && dyn_plus <= c * kc->delta_tol + SMALL); // (single: y is not updated)
            const bool c2 = n2 == 0 || f2n_plus <= kc->delta_tol + SMALL;
            const bool c3 = akkt_tol <= kc->tol + SMALL;
            bool finished = false, converged = false, out_of_time = false;
            if (anyb(c1 && c2 && c3) && (!single || status == 0)) {
                finished = converged = true;
            } else {
                const bool stall = alm_iter == 0 || anyb((dyn_plus <= kc->suff_dec * dyn + SMALL) &&
                                                         (n2 == 0 || f2n_plus <= kc->suff_dec * f2n + SMALL));
                if (!stall && !single) { // (single_inner: penalty and multipliers are returned as they came, whatever the outcome)
                    c *= kc->pen_update;
                    inv_cdiv = T(1) / (c > T(1) ? c : T(1));
                }
                akkt_tol = tmax(akkt_tol * kc->tol_update, kc->tol);
                alm_iter++;
                dyn = dyn_plus;
                f2n = f2n_plus;
                if (!single) {
                    yv = ypv;
                    yw = ypw;
                }
                reset_cache();
                if (outer >= kc->max_outer) finished = true;
                else if (!cont_time || evals_left <= 0) { // (nmpc_config.max_evaluations, see solve_instance)
                    finished = true;
                    out_of_time = true;
                }
            }
            if (finished) {
                if (!converged) status = (out_of_time || status == 2) ? 2 : 1; // (status = how the last inner solve ended)
                bool finite = tfinite(uv) && tfinite(uw) && tfinite(f_u);
                if (__ballot(!finite) != 0ull) status = 3;
                dismiss_workers();
                {
                    int ri = inst;
                    asm volatile("" : "+s"(ri));
                    if (lead) {
                        int ko = I.k; // (opaque: the lane's offsets are formed here, not hoisted out of the solve and spilled)
                        asm volatile("" : "+v"(ko));
                        kc->U[(size_t)ri * 2 * N + 2 * ko] = uv;
                        kc->U[(size_t)ri * 2 * N + 2 * ko + 1] = uw;
                        if (kc->y) {
                            kc->y[(size_t)ri * 2 * N + ko] = yv;
                            kc->y[(size_t)ri * 2 * N + N + ko] = yw;
                        }
                    }
                    if (I.lane == 0) {
                        if (kc->cost) kc->cost[ri] = f_u;
                        if (kc->status) kc->status[ri] = status;
                        if (kc->iters) {
                            kc->iters[2 * ri] = outer;
                            kc->iters[2 * ri + 1] = inner_total;
                        }
                        if (kc->info) {
                            T* o = kc->info + (size_t)ri * (8 + kProfSlots);
                            o[0] = norm_fpr;
                            o[1] = f2n_plus;
                            o[2] = dyn_plus;
                            o[3] = c;
                            o[4] = T(evals_budget() - evals_left);
                            o[5] = T(alg_grad);
                            o[6] = T(rounds); // exchange rounds (each = up to W evaluations in parallel)
                            o[7] = T(W);
#ifdef NMPC_PROFILE
                            for (int i = 0; i < kProfSlots; ++i) o[8 + i] = T(I.prof_value(i));
#endif
                        }
                    }
                }
                return;
            }
            {
                const int cap = kc->stage_outer_cap;
                if (cap > 0 && outer >= cap) { // park the instance here (KParams::resume); the master writes
                    dismiss_workers();
                    {
                        int ri = inst;
                        asm volatile("" : "+s"(ri));
                        T* rs = kc->resume + (size_t)ri * kResumeStride;
                        rs[0 * 64 + I.lane] = uv, rs[1 * 64 + I.lane] = uw, rs[2 * 64 + I.lane] = yv, rs[3 * 64 + I.lane] = yw;
                        rs[4 * 64 + I.lane] = gpv, rs[5 * 64 + I.lane] = gpw;
                        if (I.lane == 0) {
                            T* sc = rs + 6 * 64;
                            sc[0] = c;
                            sc[1] = akkt_tol;
                            sc[2] = dyn;
                            sc[3] = f2n;
                            sc[4] = T(alm_iter);
                            sc[5] = T(outer);
                            sc[6] = T(inner_total);
                            sc[7] = T(evals_budget() - evals_left);
                            sc[8] = T(alg_grad);
                            sc[9] = T(rounds);
                            sc[10] = T(0);
                            kc->status[ri] = -1;
                        }
                    }
                    return;
                }
            }
            request_uniform(uv, uw, c, true);
            phase = SP_INIT_A;
            num_iter = 0;
            cont = true;
            outer++;
            yv = tclamp(yv, T(-1e12), T(1e12));
            yw = tclamp(yw, T(-1e12), T(1e12));
            continue;
        }

        NMPC_STAMP(I, 10); // phase code (mostly: replaying the line-search tests on the exchanged candidates)
        if (step_done) {
            iteration++;
            if (!cont || !cont_time) {
                step_head = false;
            } else {
                num_iter++;
                cont = num_iter < o_max_inner;
                if (timed) cont_time = (long long)t_now <= time_budget;
                if (evals_left <= 0) cont_time = 0; // (max_evaluations: the count of the sequential algorithm, independent of W)
                step_head = true;
            }
        }

        bool inner_exit = !step_head && !need_dir;
        if (step_head) {
            if (iteration >= 1) {
                gpv = gv;
                gpw = gw;
            }
            fv = uv - hv;
            fw = uw - hw;
            const T a1 = fv + gamma * (gv - gpv), a2 = fw + gamma * (gw - gpw);
            T ff, aa;
            I.dot2x2(fv, fw, fv, fw, a1, a2, a1, a2, ff, aa);
            norm_fpr = tsqrt(ff);
            const T akkt = o_akkt_form ? tsqrt(aa) * inv_gamma : tsqrt(aa); // (nmpc_config.akkt_form)
            if (anyb(norm_fpr < o_tol && akkt < akkt_tol)) {
                inner_exit = true;
            } else {
                lip_it = 0;
                if (iteration == 0) { // first iteration of an inner solve: no line search, nothing to speculate on
                    request_uniform(hv, hw, c, false);
                    phase = SP_LIP;
                    continue;
                }
                need_dir = true;
                spec = true;
            }
        }
        NMPC_STAMP(I, 11); // step head
        if (inner_exit) {
            status = !cont ? 1 : !cont_time ? 2 : 0;
            inner_total += num_iter;
            uv = hv;
            uw = hw;
            request_uniform(uv, uw, T(0), false);
            phase = SP_OUTER;
            continue;
        }

        // ---- need_dir: lbfgs.update_hessian(gamma_fpr, u), then (iteration > 0) apply_hessian + rhs of the line
        //      search. spec = true: done BEFORE the Lipschitz test, assuming gamma survives it (otherwise the buffer
        //      is reset and everything computed here is discarded by the failure branch above).
        sigma = (T(1) - GAMMA_L) * T(0.25) * inv_gamma;
        if (lb_first) {
            lb_first = false;
            osv = uv;
            osw = uw;
            ogv = fv;
            ogw = fw;
        } else {
            const T nsv = uv - osv, nsw = uw - osw, nyv = fv - ogv, nyw = fw - ogw;
            T ys, ss;
            I.dot2x2(nsv, nsw, nyv, nyw, nsv, nsw, nsv, nsw, ys, ss);
            bool ok = true;
            const T sy_eps = o_sy_eps, cbfgs_eps = o_cbfgs_eps, cbfgs_alpha = o_cbfgs_alpha;
            if (ss <= Lim<T>::min_pos || (sy_eps > T(0) && ys <= sy_eps)) {
                ok = false;
            } else if (cbfgs_eps > T(0) && cbfgs_alpha > T(0)) {
                const T lhs = ys / ss;
                const T rhs_c = cbfgs_eps * (cbfgs_alpha == T(1) ? norm_fpr : tpow(norm_fpr, cbfgs_alpha));
                ok = lhs > rhs_c && tfinite(lhs) && tfinite(rhs_c);
            }
            ok = anyb(ok);
            if (ok) {
                osv = uv;
                osw = uw;
                ogv = fv;
                ogw = fw;
                lb_head = lb_head == 0 ? mem - 1 : lb_head - 1;
                if (lead) {
                    int ko = I.k; // (opaque, as above)
                    asm volatile("" : "+v"(ko));
                    hist[lb_head * lbfgs_slot_stride(N) + ko] = Quad<T>{nsv, nsw, nyv, nyw};
                }
                rho.set(lb_head, T(1) / ys);
                lb_gamma = ys / I.dot2(nyv, nyw, nyv, nyw);
                lb_active = lb_active + 1 < mem ? lb_active + 1 : mem;
                // (no barrier: the master alone writes and reads the ring, and a wavefront's LDS accesses stay in order)
            }
        }
        NMPC_STAMP(I, 12); // L-BFGS update
        if (iteration == 0) { // update_no_linesearch (only reached from SP_LIP)
            // (no pair can have been accepted here: iteration 0 follows a reset of the buffer, whose first update only records the point)
            uv = hv;
            uw = hw;
            request_uniform(uv, uw, c, true);
            phase = SP_NOLS;
            continue;
        }
        lbfgs_apply(I, hist, rho, N, kk, mem, lb_head, lb_active, lb_gamma, fv, fw, dv, dw);
        NMPC_STAMP(I, 13); // two-loop recursion
        if (!fbe_valid) {
            const T t1 = sv - hv, t2 = sw - hw;
            T dist2, gg;
            I.dot2x2(t1, t2, t1, t2, gv, gw, gv, gw, dist2, gg);
            fbe_cur = cost_value - T(0.5) * gamma * gg + T(0.5) * dist2 * inv_gamma;
        }
        rhs_ls = fbe_cur - sigma * norm_fpr * norm_fpr;
        tau = 1;
        nls = 0;
        if (spec) {
            // the master: the Lipschitz test's psi(u_half); roles 1.. : candidates 0..W-2
            ev = hv;
            ew = hw;
            ec = c;
            want_grad = false;
            do_eval = true;
            request_candidates(1);
            NMPC_STAMP(I, 14); // (diagnostic: forming the requests + barrier A)
            phase = SP_SPEC0;
        } else {
            request_candidates(0);
            phase = SP_LSN;
        }
    }
}

} // namespace nmpc
