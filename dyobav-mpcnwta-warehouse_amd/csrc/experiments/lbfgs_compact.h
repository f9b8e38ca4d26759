// NOT PART OF THE PRODUCT BUILD -- kept for the record (rounds 5-6).
//
// The L-BFGS direction in compact (Byrd-Nocedal-Schnabel) form, built twice in round 5, parity-green both times, measured
// slower both times (v1: configs[1] -17 %, configs[2] -8 %; v2: -10 % / -2 %;
// profiles/r05_ab_lbfgs_compact_v{1,2}_rejected.jsonl) and therefore never shipped. Round 6 moved it out of
// nmpc_device.h / nmpc_spec.h together with its plumbing (NMPC_LBFGS_COMPACT, KParams::lds_lbc, the `lb_new` flag of the
// two solver loops, the zeroing of R / Y'Y in Instance::load and the workspace reservation in make_layout); the shipped
// code object is bit-identical before and after (profiles/r06_hygiene_code_object_identity.txt).
// To revive it: include this header behind nmpc_device.h's lbfgs_apply, give KParams an `lds_lbc` offset to
// lbfgs_compact_elems(N) elements of LDS (zero the 2 * kMem * kMem matrix entries in Instance::load), set a flag `lb_new`
// where a pair is accepted (cleared on reset) and call
//     lbfgs_apply_compact(I, hist, lds + lds_lbc, rho, N, kk, mem, lb_head, lb_active, lb_new, lb_gamma, fv, fw, dv, dw); lb_new = 0;
// in place of lbfgs_apply(...) -- git show 77b6b5d:dyobav-mpcnwta-warehouse_amd/csrc/nmpc_device.h has the wiring.
#pragma once

template <typename T>
__host__ __device__ constexpr int lbfgs_compact_elems(int N) { return 4 * N + 2 * kMem * kMem; }

// lbfgs.apply_hessian in COMPACT form (Byrd, Nocedal & Schnabel 1994, eq. 3.1) -- the same operator H as the two-loop
// recursion above, H = g I + [S gY] [ R^-T (D + g Y'Y) R^-1 , -R^-T ; -R^-1 , 0 ] [S' ; gY'] with R_ij = s_i . y_j for a pair i
// older than or equal to j, D = its diagonal, g = lb_gamma -- arranged so that NOTHING is a chain of dependent wave
// reductions (round 5; VERDICT r4 item 3: the recursion is 2 x lb_active inner products, each a 6-step DPP reduction that
// has to finish before the next can start -- ~220 cycles per step on the lone master wavefront of the latency kernel):
//   0. a new pair: the stored matrices R and Y'Y -- kept in LOGICAL order, index 0 = newest -- move one down the diagonal;
//   1. all inner products at once, one LANE per inner product: lane 10 g + j forms, by itself, the 2 N-term sum
//        g = 0: s_j . q     g = 1: y_j . q     g = 2: s_j . y_0     g = 3: y_j . y_0      (j = logical index, 0 = the newest
//      pair) from the ring in LDS -- 2 N reads + 2 N FMAs per lane, no cross-lane step. Groups 2 and 3 are the new column
//      of R and the new row / column of Y'Y (used when a pair was accepted in this iteration), groups 0 and 1 are p and r;
//   2. R w = p by substitution, newest pair first; lane j <-> logical index j holds row j of R in registers: per step one
//      broadcast (v_readlane) and one FMA; steps beyond lb_active run on a zero;
//   3. t = D w + g Y'Y w - g r, likewise;
//   4. R' z = t, oldest pair first (column j of R in registers);
//   5. d = g q + sum_j z_j s_j - g w_j y_j over the ring, every lane for its own horizon step.
// Every LDS read of steps 2-4 is issued before the first substitution step (the first version read inside the loops: one
// exposed LDS round trip per step, -17 % on configs[1] -- profiles/r05_ab_lbfgs_compact_v1_rejected.jsonl).
// Same mathematics, another rounding: like every other pair of kernel families, the results agree with the two-loop
// kernels to rounding, not bit for bit.   lb_ws (LDS, KParams::lds_lbc): Quad qtab[N] | R[kMem][kMem] | YY[kMem][kMem]

template <typename T, typename Inst>
__device__ __forceinline__ void lbfgs_apply_compact(const Inst& I, const Quad<T>* hist, T* lb_ws, const LaneVec<T>& rho, const int N,
                                                    const int kk, const int mem_, const int lb_head, const int lb_active,
                                                    const int new_pair, const T lb_gamma, const T fv, const T fw, T& dv, T& dw)
{
    const int head = __builtin_amdgcn_readfirstlane(lb_head), nact = __builtin_amdgcn_readfirstlane(lb_active);
    const int mem = __builtin_amdgcn_readfirstlane(mem_);
    if (nact == 0) { // (wave-uniform) H = I
        dv = fv;
        dw = fw;
        return;
    }
    const int NS = lbfgs_slot_stride(N);
    Quad<T>* const qtab = reinterpret_cast<Quad<T>*>(lb_ws);
    T* const Rm = lb_ws + 4 * N;
    T* const YYm = Rm + kMem * kMem;
    int ln = I.lane; // (opaque: nothing derived from it is hoisted out of the solver loop into registers held for the whole solve)
    asm volatile("" : "+v"(ln));
    // ---- 0. a new pair is logical index 0: entry (i, j) of both matrices becomes (i + 1, j + 1); the last row / column drop out
    if (new_pair) { // (wave-uniform)
        const int e0 = ln, e1 = ln + 64;                                   // old entries (row-major, kMem x kMem)
        const bool m0 = (e0 / kMem) < kMem - 1 && (e0 % kMem) < kMem - 1;  // (e0 < 64 < kMem * kMem)
        const bool m1 = e1 < kMem * kMem && (e1 / kMem) < kMem - 1 && (e1 % kMem) < kMem - 1;
        const T r0 = Rm[e0], y0 = YYm[e0], r1 = Rm[m1 ? e1 : 0], y1 = YYm[m1 ? e1 : 0];
        asm volatile("" ::: "memory"); // (every read before the first write: a wavefront's LDS accesses execute in order)
        if (m0) {
            Rm[e0 + kMem + 1] = r0;
            YYm[e0 + kMem + 1] = y0;
        }
        if (m1) {
            Rm[e1 + kMem + 1] = r1;
            YYm[e1 + kMem + 1] = y1;
        }
    }
    // ---- 1. the inner products, one lane each
    if (I.lead) qtab[I.k] = Quad<T>{fv, fw, T(0), T(0)};
    asm volatile("" ::: "memory");
    const int grp = ln < kMem ? 0 : ln < 2 * kMem ? 1 : ln < 3 * kMem ? 2 : 3;
    const int jl = ln - kMem * grp;                                        // logical index of this lane's pair (lanes >= 4 kMem: idle)
    int sl = head + (jl < mem ? jl : 0);                                   // its physical slot
    sl = sl >= mem ? sl - mem : sl;
    const T* pa = reinterpret_cast<const T*>(hist + sl * NS) + ((grp & 1) ? 2 : 0);                      // s_j or y_j
    const T* pb = grp < 2 ? reinterpret_cast<const T*>(qtab) : reinterpret_cast<const T*>(hist + head * NS) + 2; // q or y_0
    T acc0 = 0, acc1 = 0;
    auto fma_ = [](T a, T b, T c) { return sizeof(T) == 4 ? (T)__builtin_fmaf((float)a, (float)b, (float)c) : (T)__builtin_fma((double)a, (double)b, (double)c); };
#pragma unroll 10
    for (int k = 0; k < N; ++k) {
        acc0 = fma_(pa[4 * k], pb[4 * k], acc0);
        acc1 = fma_(pa[4 * k + 1], pb[4 * k + 1], acc1);
    }
    const T acc = acc0 + acc1;
    const bool row_on = ln < 4 * kMem && jl < nact;
    if (new_pair) { // the new pair's column of R (s_j . y_0: j older or equal) and row / column of Y'Y
        if (row_on && grp == 2) Rm[jl * kMem] = acc;
        if (row_on && grp == 3) {
            YYm[jl * kMem] = acc;
            YYm[jl] = acc;
        }
    }
    asm volatile("" ::: "memory");
    // ---- lanes 0 .. kMem-1 from here on: lane j <-> logical index j. p = s_j . q is in place, r = y_j . q comes from lane
    //      kMem + j, 1 / (s_j . y_j) from the lane of the pair's physical slot. Row j of R, column j of R and row j of Y'Y -> registers.
    const int lj = ln < kMem ? ln : 0;                                      // (idle lanes mimic lane 0: values never used)
    int slj = head + (lj < mem ? lj : 0);
    slj = slj >= mem ? slj - mem : slj;
    const T r_j = bperm(acc, 4 * (ln + kMem < 64 ? ln + kMem : 63));
    const T rinv = bperm(rho.v, 4 * slj);
    T Rr[kMem], Rc[kMem], Yr[kMem];
    const T diag = Rm[lj * (kMem + 1)];                                      // D_jj = s_j . y_j
#pragma unroll
    for (int j = 0; j < kMem; ++j) {
        Rr[j] = Rm[lj * kMem + j];   // s_lj . y_j : valid for j newer or equal (j <= lj)
        Rc[j] = Rm[j * kMem + lj];   // s_j . y_lj : valid for j older or equal (j >= lj)
        Yr[j] = YYm[lj * kMem + j];
    }
    // ---- 2. R w = p, newest pair first: w_j = (p_j - sum_{i < j} R[j][i] w_i) / R[j][j]
    T pacc = acc, w = 0;
#pragma unroll
    for (int j = 0; j < kMem; ++j) {
        T wb = read_lane(pacc * rinv, j);
        wb = j < nact ? wb : T(0);                                         // (wave-uniform select: steps beyond the buffer add zeros)
        w = ln == j ? wb : w;
        pacc -= Rr[j] * wb;                                                 // (lanes of pairs already solved: garbage, not read again)
    }
    // ---- 3. t = D w + g Y'Y w - g r
    T t = diag * w - lb_gamma * r_j;
#pragma unroll
    for (int j = 0; j < kMem; ++j) {
        T gw = lb_gamma * read_lane(w, j);
        gw = j < nact ? gw : T(0);
        t += Yr[j] * gw;
    }
    // ---- 4. R' z = t, oldest pair first: z_j = (t_j - sum_{i > j} R[i][j] z_i) / R[j][j]
    T z = 0;
#pragma unroll
    for (int j = kMem - 1; j >= 0; --j) {
        T zb = read_lane(t * rinv, j);
        zb = j < nact ? zb : T(0);
        z = ln == j ? zb : z;
        t -= Rc[j] * zb;
    }
    // ---- 5. d = g q + S z - g Y w, every lane for its own step
    T qv = lb_gamma * fv, qw = lb_gamma * fw;
    int sb = head;
#pragma unroll
    for (int j = 0; j < kMem; ++j) {
        if (j < nact) { // (wave-uniform)
            const T zs = read_lane(z, j), ws = lb_gamma * read_lane(w, j);
            const Quad<T> hq = hist[sb * NS + kk];
            qv += zs * hq.a - ws * hq.c;
            qw += zs * hq.b - ws * hq.d;
            sb = sb + 1 == mem ? 0 : sb + 1;
        }
    }
    dv = qv;
    dw = qw;
}
