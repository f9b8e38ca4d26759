/*
 * nmpc_oracle_impl.h -- body of the CPU oracle, included twice by nmpc_oracle.c with
 *   REAL = double / float and SUF(x) = x##_f64 / x##_f32.
 * TEST INFRASTRUCTURE ONLY (see nmpc_oracle.h for the parity status: problem functions pinned,
 * solver algorithm "parity unpinned").
 *
 * All "ref:" citations are relative to /root/reference/src/.
 * All "OpEn:" citations name the upstream (un-vendored) optimization_engine / lbfgs / opengen item that
 * the block restates.
 */

#ifndef REAL
#error "include from nmpc_oracle.c"
#endif

/* ORC_REASSOC (third instantiation, suffix _r64): THE SAME ALGORITHM IN THE SAME PRECISION WITH ITS SUMS ASSOCIATED
 * DIFFERENTLY -- inner products and the accumulations over obstacles / polygons / robots run in reverse index order, the
 * two obstacle snapshots swap places, the rollout positions are X0 + (running sum of the increments) instead of a chain
 * of additions onto X. No decision rule, constant or tie-break differs (the segment minimum keeps "first wins"). It
 * exists to measure how far two correct fp64 implementations of this solver end up from each other (the oracle's own
 * noise floor under re-association: tests/accuracy_protocol.py). */
#ifdef ORC_REASSOC
#define ORC_IDX(i, n) ((n) - 1 - (i))
#else
#define ORC_IDX(i, n) (i)
#endif

#define ORC_MAXN 64                 /* max horizon supported by the oracle   */
#define ORC_MAXNV (2 * ORC_MAXN)    /* max number of decision variables      */
#define ORC_MAXDYN 512

typedef struct {
    int um1, s0, sN, q, rs, rv, c0, c, os, od, qstc, qdyn, np;
} SUF(offs);

static void SUF(make_offs)(const orc_problem *pr, SUF(offs) * o)
{
    /* ref: pkg_mpc_tracker/solver_build/mpc_builder.py:47-60 (z = vertcat(...)),
     *      pkg_mpc_tracker/trajectory_tracker.py:315-317 */
    const int N = pr->N;
    o->um1 = 0;
    o->s0 = 2;
    o->sN = 5;
    o->q = 8;
    o->rs = 18;
    o->rv = o->rs + 3 * N;
    o->c0 = o->rv + N;
    o->c = o->c0 + 3 * pr->Nother;
    o->os = o->c + 3 * N * pr->Nother;
    o->od = o->os + 12 * pr->Nstc;
    o->qstc = o->od + 6 * (N + 1) * pr->Ndyn;
    o->qdyn = o->qstc + N;
    o->np = o->qdyn + N;
}

/* orc_options.hoist_trig: cos / sin of every ellipse angle computed ONCE per solve into this per-thread table instead of on
 * every evaluation. Same values, same bits -- a speed option for the CPU baseline only: CasADi's generated code (what the
 * reference executes) has no notion of per-solve subexpressions and recomputes them on every call, which is what the
 * default does; a hand-tuned CPU solver would hoist them, and VERDICT r3 called a baseline without that pessimistic.
 * Layout [j * (N + 1) + t][2] = (cos, sin); NULL = compute on the fly. */
static __thread const REAL *SUF(tl_trig) = 0;

static inline REAL SUF(rmax)(REAL a, REAL b) { return a > b ? a : b; }
static inline REAL SUF(rmin)(REAL a, REAL b) { return a < b ? a : b; }

/* ref: pkg_mpc_tracker/solver_build/mpc_helper.py:17-36 (dist_to_lineseg), squared.
 * Deliberate deviation: the reference takes norm_2 and squares it again (mpc_cost.py:92-93); the value is
 * identical, the derivative of the squared form is finite on the segment itself. */
static inline REAL SUF(seg_d2)(REAL px, REAL py, REAL ax, REAL ay, REAL bx, REAL by, REAL *tx, REAL *ty)
{
    REAL dx = bx - ax, dy = by - ay;
    REAL t_hat = ((px - ax) * dx + (py - ay) * dy) / (dx * dx + dy * dy + (REAL)1e-16);
    REAL t_star = SUF(rmin)(SUF(rmax)(t_hat, (REAL)0), (REAL)1);
    REAL vx = ax + t_star * dx - px, vy = ay + t_star * dy - py; /* vector to the closest point */
    *tx = vx;
    *ty = vy;
    return vx * vx + vy * vy;
}

/* per-ellipse constants shared by the plain and the margin-inflated indicator
 * ref: mpc_helper.py:38-52 (inside_ellipses) */
static inline REAL SUF(ell_ind)(REAL a2, REAL b2, REAL rx, REAL ry)
{
    REAL ex = rx + (REAL)1e-6, ey = ry + (REAL)1e-6;
    return (REAL)1 - a2 / (ex * ex) - b2 / (ey * ey);
}

/*
 * One stage (the body of the horizon loop, mpc_builder.py:79-143) evaluated at the post-step position
 * (x, y) of step k.
 *   forward  (cost != NULL): adds the stage cost to *cost, adds the scalar polygon penalty to *pen_s and the
 *                            per-obstacle ellipse penalties to F2[j].
 *   backward (gx != NULL)  : returns d/d(x,y) of [stage cost + sum_j W[j]*F2_j-contributions of this stage],
 *                            W[j] = c*F2_j, Wsum = sum_j W[j].
 */
static void SUF(stage)(const orc_problem *pr, const SUF(offs) * o, const REAL *p, int k, REAL x, REAL y,
                       REAL *cost, REAL *pen_s, REAL *F2, const REAL *W, REAL Wsum, REAL *gx, REAL *gy)
{
    const int N = pr->N;
    const REAL qrpd = p[o->q + 7];
    const REAL vm = (REAL)pr->vehicle_margin, sm = (REAL)pr->social_margin;
    const REAL safe2 = (REAL)(pr->vehicle_width * pr->vehicle_width);
    REAL c_acc = 0, g_x = 0, g_y = 0;

    /* --- reference path deviation: min over segments k..N-1 of the (N+1)-row reference whose last row
     *     duplicates row N-1 (mpc_builder.py:68-69, :81; mpc_cost.py:84-95) */
    {
        REAL best = 0, btx = 0, bty = 0;
        for (int i = k; i < N; ++i) {
            int i2 = (i + 1 < N) ? i + 1 : N - 1;
            REAL tx, ty;
            REAL d2 = SUF(seg_d2)(x, y, p[o->rs + 3 * i], p[o->rs + 3 * i + 1], p[o->rs + 3 * i2],
                                  p[o->rs + 3 * i2 + 1], &tx, &ty);
            if (i == k || d2 < best) {
                best = d2;
                btx = tx;
                bty = ty;
            }
        }
        c_acc += qrpd * best;
        g_x += qrpd * (REAL)(-2) * btx;
        g_y += qrpd * (REAL)(-2) * bty;
    }

    /* --- fleet collision, other robots at t=0: robots 1..Nother-1 only, weight 1000
     *     (mpc_builder.py:86-90: the strided slice starts at ns; mpc_cost.py:65-76) */
    for (int jj = 1; jj < pr->Nother; ++jj) {
        const int j = 1 + ORC_IDX(jj - 1, pr->Nother - 1);
        REAL dx = x - p[o->c0 + 3 * j], dy = y - p[o->c0 + 3 * j + 1];
        REAL h = safe2 - (dx * dx + dy * dy);
        if (h > 0) {
            c_acc += (REAL)1000 * h;
            g_x += (REAL)1000 * (REAL)(-2) * dx;
            g_y += (REAL)1000 * (REAL)(-2) * dy;
        }
    }
    /* --- fleet collision, predictive: all Nother robots at step k, weight 10 (mpc_builder.py:93-97);
     *     c is laid out [j*3N + k*3 + f] */
    for (int jj = 0; jj < pr->Nother; ++jj) {
        const int j = ORC_IDX(jj, pr->Nother);
        const REAL *cj = p + o->c + j * 3 * N + k * 3;
        REAL dx = x - cj[0], dy = y - cj[1];
        REAL h = safe2 - (dx * dx + dy * dy);
        if (h > 0) {
            c_acc += (REAL)10 * h;
            g_x += (REAL)10 * (REAL)(-2) * dx;
            g_y += (REAL)10 * (REAL)(-2) * dy;
        }
    }

    /* --- static obstacles (mpc_builder.py:100-108; mpc_helper.py:54-75; mpc_cost.py:6-24) */
    {
        const REAL qs = p[o->qstc + k];
        for (int ii = 0; ii < pr->Nstc; ++ii) {
            const int i = ORC_IDX(ii, pr->Nstc);
            const REAL *b = p + o->os + 12 * i, *a0 = b + 4, *a1 = b + 8;
            REAL h[4], ind = 1;
            for (int e = 0; e < 4; ++e) {
                h[e] = SUF(rmax)((REAL)0, b[e] - a0[e] * x - a1[e] * y);
                ind *= h[e];
            }
            if (cost) {
                c_acc += qs * ind * ind;
                *pen_s += SUF(rmax)((REAL)0, ind);
            }
            if (gx && ind > 0) {
                /* d ind / d(x,y) = sum_e (-a_e) * prod_{e' != e} h_e'   (all h_e > 0 here) */
                REAL dix = 0, diy = 0;
                for (int e = 0; e < 4; ++e) {
                    REAL pr_o = 1;
                    for (int e2 = 0; e2 < 4; ++e2)
                        if (e2 != e) pr_o *= h[e2];
                    dix += -a0[e] * pr_o;
                    diy += -a1[e] * pr_o;
                }
                REAL w = (REAL)2 * qs * ind + Wsum; /* cost term + scalar-broadcast penalty term */
                g_x += w * dix;
                g_y += w * diy;
            }
        }
    }

    /* --- dynamic obstacles: snapshot t=0 (mpc_builder.py:111-125) and t=k+1 (:129-143)
     *     o_d is laid out [j*6(N+1) + t*6 + f], f = (x, y, rx, ry, angle, alpha) */
    for (int sn = 0; sn < 2; ++sn) {
        const int snap = ORC_IDX(sn, 2);
        const int t = snap == 0 ? 0 : k + 1;
        const REAL marg = snap == 0 ? vm + sm : vm;
        const REAL wgt = snap == 0 ? (REAL)1000 : p[o->qdyn + k];
        for (int jj = 0; jj < pr->Ndyn; ++jj) {
            const int j = ORC_IDX(jj, pr->Ndyn);
            const REAL *e = p + o->od + j * 6 * (N + 1) + t * 6;
            REAL dx = x - e[0], dy = y - e[1];
            REAL ca, sa;
            if (SUF(tl_trig)) {
                ca = SUF(tl_trig)[2 * (j * (N + 1) + t)];
                sa = SUF(tl_trig)[2 * (j * (N + 1) + t) + 1];
            } else {
                ca = SUF(rcos)(e[4]);
                sa = SUF(rsin)(e[4]);
            }
            REAL a = dx * ca + dy * sa, b = dx * sa - dy * ca;
            REAL a2 = a * a, b2 = b * b;
            REAL ind_p = SUF(ell_ind)(a2, b2, e[2], e[3]);               /* hard (penalty) ellipse */
            REAL ind_m = SUF(ell_ind)(a2, b2, e[2] + marg, e[3] + marg); /* soft (cost) ellipse    */
            REAL hp = SUF(rmax)((REAL)0, ind_p), hm = SUF(rmax)((REAL)0, ind_m);
            if (cost) {
                F2[j] += hp;
                c_acc += wgt * e[5] * hm * hm;
            }
            if (gx) {
                /* d ind/d(x,y) = -2 a/ex^2 (ca, sa) - 2 b/ey^2 (sa, -ca) */
                if (hm > 0) {
                    REAL ex = e[2] + marg + (REAL)1e-6, ey = e[3] + marg + (REAL)1e-6;
                    REAL pa = a / (ex * ex), pb = b / (ey * ey);
                    REAL w = (REAL)2 * wgt * e[5] * hm;
                    g_x += w * (REAL)(-2) * (pa * ca + pb * sa);
                    g_y += w * (REAL)(-2) * (pa * sa - pb * ca);
                }
                if (hp > 0 && W) {
                    REAL ex = e[2] + (REAL)1e-6, ey = e[3] + (REAL)1e-6;
                    REAL pa = a / (ex * ex), pb = b / (ey * ey);
                    g_x += W[j] * (REAL)(-2) * (pa * ca + pb * sa);
                    g_y += W[j] * (REAL)(-2) * (pa * sa - pb * ca);
                }
            }
        }
    }

    if (cost) *cost += c_acc;
    if (gx) {
        *gx = g_x;
        *gy = g_y;
    }
}

/*
 * Everything in one place: f, F1, F2, psi and (optionally) grad psi.
 *   psi(u; c, y) = f + c/2 * [ dist^2_C(F1 + y/max(c,1)) + ||F2||^2 ]
 * OpEn: opengen/builder/optimizer_builder.py (construction of psi; note the max(c,1) so that c = 0 returns f).
 * Gradient: hand-written adjoint of the rollout, replacing CasADi reverse AD.
 */
static void SUF(core)(const orc_problem *pr, const REAL *u, REAL c, const REAL *ymul, const REAL *p, REAL *f_out,
                      REAL *F1_out, REAL *F2_out, REAL *psi_out, REAL *grad)
{
    SUF(offs) o;
    SUF(make_offs)(pr, &o);
    const int N = pr->N, Nd = pr->Ndyn;
    const REAL ts = (REAL)pr->ts;
    const REAL *q = p + o.q;
    const REAL qvel = q[1], rv = q[3], rw = q[4], qN = q[5], qthN = q[6], accp = q[8], waccp = q[9];

    REAL X[ORC_MAXN + 1], Y[ORC_MAXN + 1], TH[ORC_MAXN + 1];
    REAL Ck[ORC_MAXN], Sk[ORC_MAXN], dCw[ORC_MAXN], dSw[ORC_MAXN];
    REAL F1[ORC_MAXNV], F2[ORC_MAXDYN];
    REAL f = 0, pen_s = 0;
    for (int j = 0; j < Nd; ++j) F2[j] = 0;

    /* ---- forward rollout + stage costs (mpc_builder.py:74-143) */
    X[0] = p[o.s0];
    Y[0] = p[o.s0 + 1];
    TH[0] = p[o.s0 + 2];
#ifdef ORC_REASSOC
    REAL sumx = 0, sumy = 0, sumt = 0;
#endif
    for (int k = 0; k < N; ++k) {
        const REAL v = u[2 * k], w = u[2 * k + 1];
        /* unicycle RK4 in closed form (ref: basic_motion_model/motion_model.py:141-163):
         * k1..k4 only differ in the heading theta, theta+h, theta+h, theta+2h with h = ts*w/2 */
        const REAL th = TH[k], h = ts * w / (REAL)2;
        const REAL c0 = SUF(rcos)(th), c1 = SUF(rcos)(th + h), c2 = SUF(rcos)(th + (REAL)2 * h);
        const REAL s0 = SUF(rsin)(th), s1 = SUF(rsin)(th + h), s2 = SUF(rsin)(th + (REAL)2 * h);
        Ck[k] = (c0 + (REAL)4 * c1 + c2) / (REAL)6;
        Sk[k] = (s0 + (REAL)4 * s1 + s2) / (REAL)6;
        dCw[k] = -ts * ((REAL)2 * s1 + s2) / (REAL)6;
        dSw[k] = ts * ((REAL)2 * c1 + c2) / (REAL)6;
#ifdef ORC_REASSOC
        sumx += ts * v * Ck[k];
        sumy += ts * v * Sk[k];
        sumt += ts * w;
        X[k + 1] = X[0] + sumx;
        Y[k + 1] = Y[0] + sumy;
        TH[k + 1] = TH[0] + sumt;
#else
        X[k + 1] = X[k] + ts * v * Ck[k];
        Y[k + 1] = Y[k] + ts * v * Sk[k];
        TH[k + 1] = th + ts * w;
#endif

        SUF(stage)(pr, &o, p, k, X[k + 1], Y[k + 1], &f, &pen_s, F2, 0, 0, 0, 0);
        f += qvel * (v - p[o.rv + k]) * (v - p[o.rv + k]); /* mpc_cost.py:78-79 */
        f += rv * v * v + rw * w * w;                     /* mpc_cost.py:46-53 */
    }
    /* scalar (polygon) part of the penalty constraint is broadcast onto every component
     * (mpc_builder.py:72,106,119,137) */
    for (int j = 0; j < Nd; ++j) F2[j] += pen_s;

    /* ---- terminal cost (mpc_builder.py:148) */
    {
        REAL ex = X[N] - p[o.sN], ey = Y[N] - p[o.sN + 1], et = TH[N] - p[o.sN + 2];
        f += qN * (ex * ex + ey * ey) + qthN * et * et;
    }
    /* ---- accelerations (mpc_builder.py:156-169): F1 = (acc; w_acc) */
    for (int k = 0; k < N; ++k) {
        REAL vp = k ? u[2 * (k - 1)] : p[o.um1], wp = k ? u[2 * (k - 1) + 1] : p[o.um1 + 1];
        F1[k] = (u[2 * k] - vp) / ts;
        F1[N + k] = (u[2 * k + 1] - wp) / ts;
        f += accp * F1[k] * F1[k] + waccp * F1[N + k] * F1[N + k];
    }

    if (f_out) *f_out = f;
    if (F1_out)
        for (int i = 0; i < 2 * N; ++i) F1_out[i] = F1[i];
    if (F2_out)
        for (int j = 0; j < Nd; ++j) F2_out[j] = F2[j];
    if (!psi_out && !grad) return;

    /* ---- augmented-Lagrangian + penalty terms */
    REAL dC[ORC_MAXNV]; /* z - Proj_C(z), z = F1 + y/max(c,1) */
    REAL psi = f, d2 = 0, f2s = 0;
    const REAL cdiv = c > (REAL)1 ? c : (REAL)1;
    for (int ii = 0; ii < 2 * N; ++ii) {
        const int i = ORC_IDX(ii, 2 * N);
        REAL lo = i < N ? (REAL)pr->lin_acc_min : (REAL)(-pr->ang_acc_max);
        REAL hi = i < N ? (REAL)pr->lin_acc_max : (REAL)pr->ang_acc_max;
        REAL z = F1[i] + (ymul ? ymul[i] : (REAL)0) / cdiv;
        REAL pz = SUF(rmin)(SUF(rmax)(z, lo), hi);
        dC[i] = z - pz;
        d2 += dC[i] * dC[i];
    }
    for (int j = 0; j < Nd; ++j) f2s += F2[ORC_IDX(j, Nd)] * F2[ORC_IDX(j, Nd)];
    psi += c * (d2 + f2s) / (REAL)2;
    if (psi_out) *psi_out = psi;
    if (!grad) return;

    /* ---- backward sweep: lambda = dJ/ds_{k+1}; g_k = dl/du_k + B_k^T lambda; lambda <- A_k^T lambda */
    REAL W[ORC_MAXDYN], Wsum = 0;
    for (int jj = 0; jj < Nd; ++jj) {
        const int j = ORC_IDX(jj, Nd);
        W[j] = c * F2[j];
        Wsum += W[j];
    }
    REAL lx = (REAL)2 * qN * (X[N] - p[o.sN]), ly = (REAL)2 * qN * (Y[N] - p[o.sN + 1]),
         lt = (REAL)2 * qthN * (TH[N] - p[o.sN + 2]);
    for (int k = N - 1; k >= 0; --k) {
        const REAL v = u[2 * k], w = u[2 * k + 1];
        REAL gx, gy;
        SUF(stage)(pr, &o, p, k, X[k + 1], Y[k + 1], 0, 0, 0, W, Wsum, &gx, &gy);
        lx += gx;
        ly += gy;
        grad[2 * k] = ts * (Ck[k] * lx + Sk[k] * ly) + (REAL)2 * qvel * (v - p[o.rv + k]) + (REAL)2 * rv * v;
        grad[2 * k + 1] = ts * v * (dCw[k] * lx + dSw[k] * ly) + ts * lt + (REAL)2 * rw * w;
        lt += ts * v * (-Sk[k] * lx + Ck[k] * ly);
    }
    /* acceleration cost + ALM term: d/dF1_i = 2*pen*F1_i + c*dC_i */
    for (int k = 0; k < N; ++k) {
        REAL da = (REAL)2 * accp * F1[k] + c * dC[k];
        REAL dw = (REAL)2 * waccp * F1[N + k] + c * dC[N + k];
        grad[2 * k] += da / ts;
        grad[2 * k + 1] += dw / ts;
        if (k) {
            grad[2 * (k - 1)] -= da / ts;
            grad[2 * (k - 1) + 1] -= dw / ts;
        }
    }
}

void SUF(orc_eval)(const orc_problem *pr, const REAL *u, const REAL *p, REAL *f, REAL *F1, REAL *F2)
{
    SUF(core)(pr, u, 0, 0, p, f, F1, F2, 0, 0);
}

void SUF(orc_psi)(const orc_problem *pr, const REAL *u, REAL c, const REAL *y, const REAL *p, REAL *psi, REAL *grad)
{
    SUF(core)(pr, u, c, y, p, 0, 0, 0, psi, grad);
}

/* =====================================================================================================
 *  Solver: PANOC (L-BFGS directions) inside ALM / quadratic-penalty outer loop.
 *  Restated from the upstream OpEn project (NOT in the reference tree; "parity unpinned").
 * ===================================================================================================== */

typedef struct {
    const orc_problem *pr;
    const orc_options *op;
    const REAL *p;
    int n;
    REAL c;
    const REAL *y;
    int n_cost, n_grad;
    int n_points; /* distinct arguments evaluated so far (orc_options.max_evals) */
    /* optional iteration trace (orc_solve_trace_*): one record of ORC_TRACE_HEAD + n doubles per completed inner
     * iteration -- see nmpc_oracle.h */
    double *trace;
    int max_rec, n_rec, outer;
    /* scratch of the iteration in progress */
    int t_lip, t_nls, t_cbfgs, t_kind;
    double t_margin, t_steep;
    double outer_margin; /* smallest relative margin of the outer loop's decisions that preceded this inner solve (kind 5) */
    int outer_margin_set;
} SUF(ctx);

/* smallest relative margin of the discrete decisions taken in the iteration in progress, and which one it was
 * (1 Lipschitz backtracking test, 2 line-search test, 3 C-BFGS / s'y acceptance, 4 exit test) */
static void SUF(note_margin)(SUF(ctx) * cx, double m, int kind)
{
    if (!cx->trace) return;
    if (m < 0) m = -m;
    if (cx->t_kind == 0 || m < cx->t_margin) {
        cx->t_margin = m;
        cx->t_kind = kind;
    }
}

static REAL SUF(cost)(SUF(ctx) * cx, const REAL *u)
{
    REAL v;
    SUF(core)(cx->pr, u, cx->c, cx->y, cx->p, 0, 0, 0, &v, 0);
    cx->n_cost++;
    return v;
}
static void SUF(gradf)(SUF(ctx) * cx, const REAL *u, REAL *g)
{
    REAL v;
    SUF(core)(cx->pr, u, cx->c, cx->y, cx->p, 0, 0, 0, &v, g);
    cx->n_grad++;
    cx->n_points++; /* (every gradient is taken at a new point; the cost call that goes with it is the same point) */
}
static REAL SUF(dot)(const REAL *a, const REAL *b, int n)
{
    REAL s = 0;
    for (int i = 0; i < n; ++i) s += a[ORC_IDX(i, n)] * b[ORC_IDX(i, n)];
    return s;
}
static REAL SUF(norm2)(const REAL *a, int n) { return SUF(rsqrt)(SUF(dot)(a, a, n)); }

static void SUF(project_U)(const orc_problem *pr, REAL *u, int n)
{
    /* OpEn: constraints::Rectangle::project; set U of mpc_builder.py:151-153 */
    for (int i = 0; i < n; ++i) {
        REAL lo = (i & 1) ? (REAL)(-pr->ang_vel_max) : (REAL)pr->lin_vel_min;
        REAL hi = (i & 1) ? (REAL)pr->ang_vel_max : (REAL)pr->lin_vel_max;
        u[i] = SUF(rmin)(SUF(rmax)(u[i], lo), hi);
    }
}

/* ---- L-BFGS buffer (OpEn: crate lbfgs, struct Lbfgs; index 0 = newest pair) */
typedef struct {
    int n, mem, active, first_old;
    REAL gamma;
    REAL s[ORC_MAX_MEM + 1][ORC_MAXNV], y[ORC_MAX_MEM + 1][ORC_MAXNV];
    REAL rho[ORC_MAX_MEM + 1], alpha[ORC_MAX_MEM];
    REAL old_state[ORC_MAXNV], old_g[ORC_MAXNV];
} SUF(lbfgs);

static void SUF(lbfgs_reset)(SUF(lbfgs) * L)
{
    L->active = 0;
    L->first_old = 1;
}

/* OpEn: Lbfgs::update_hessian(g, state) incl. new_s_and_y_valid (C-BFGS test of Li & Fukushima) */
static void SUF(lbfgs_update)(SUF(lbfgs) * L, const orc_options *op, const REAL *g, const REAL *state, int *flag,
                              double *margin)
{
    *flag = -1; /* -1: first call / nothing tested, 0: pair rejected, 1: pair accepted */
    *margin = 1e300;
    const int n = L->n, m = L->mem;
    if (L->first_old) {
        L->first_old = 0;
        for (int i = 0; i < n; ++i) {
            L->old_state[i] = state[i];
            L->old_g[i] = g[i];
        }
        return;
    }
    REAL *sn = L->s[m], *yn = L->y[m]; /* temporary slot */
    for (int i = 0; i < n; ++i) {
        sn[i] = state[i] - L->old_state[i];
        yn[i] = g[i] - L->old_g[i];
    }
    REAL ys = SUF(dot)(sn, yn, n), ss = SUF(dot)(sn, sn, n);
    int ok = 1;
    if (op->sy_eps > 0) *margin = ((double)ys - op->sy_eps) / ((double)(ys < 0 ? -ys : ys) + 1e-300);
    if (ss <= (REAL)REAL_MIN_POS || (op->sy_eps > 0 && ys <= (REAL)op->sy_eps)) {
        ok = 0;
    } else if (op->cbfgs_eps > 0 && op->cbfgs_alpha > 0) {
        REAL lhs = ys / ss;
        REAL rhs = (REAL)op->cbfgs_eps * SUF(rpow)(SUF(norm2)(g, n), (REAL)op->cbfgs_alpha);
        ok = (lhs > rhs) && SUF(risfinite)(lhs) && SUF(risfinite)(rhs);
        double m2 = ((double)lhs - (double)rhs) / ((double)(lhs < 0 ? -lhs : lhs) + 1e-300);
        if ((m2 < 0 ? -m2 : m2) < (*margin < 0 ? -*margin : *margin)) *margin = m2;
    }
    *flag = ok;
    if (!ok) return; /* rejection: old point kept */
    for (int i = 0; i < n; ++i) {
        L->old_state[i] = state[i];
        L->old_g[i] = g[i];
    }
    /* rotate_right(1): temp slot becomes index 0 */
    REAL ts_[ORC_MAXNV], ty_[ORC_MAXNV];
    for (int i = 0; i < n; ++i) {
        ts_[i] = sn[i];
        ty_[i] = yn[i];
    }
    for (int k = m; k >= 1; --k) {
        for (int i = 0; i < n; ++i) {
            L->s[k][i] = L->s[k - 1][i];
            L->y[k][i] = L->y[k - 1][i];
        }
        L->rho[k] = L->rho[k - 1];
    }
    for (int i = 0; i < n; ++i) {
        L->s[0][i] = ts_[i];
        L->y[0][i] = ty_[i];
    }
    L->rho[0] = (REAL)1 / SUF(dot)(L->s[0], L->y[0], n);
    L->gamma = ((REAL)1 / L->rho[0]) / SUF(dot)(L->y[0], L->y[0], n);
    L->active = L->active + 1 < m ? L->active + 1 : m;
}

/* OpEn: Lbfgs::apply_hessian (two-loop recursion, newest pair first) */
static void SUF(lbfgs_apply)(SUF(lbfgs) * L, REAL *q)
{
    const int n = L->n;
    if (L->active == 0) return;
    for (int k = 0; k < L->active; ++k) {
        REAL a = L->rho[k] * SUF(dot)(L->s[k], q, n);
        L->alpha[k] = a;
        for (int i = 0; i < n; ++i) q[i] -= a * L->y[k][i];
    }
    for (int i = 0; i < n; ++i) q[i] *= L->gamma;
    for (int k = L->active - 1; k >= 0; --k) {
        REAL beta = L->rho[k] * SUF(dot)(L->y[k], q, n);
        for (int i = 0; i < n; ++i) q[i] += (L->alpha[k] - beta) * L->s[k][i];
    }
}

/* ---- PANOC cache (OpEn: core::panoc::PANOCCache) */
typedef struct {
    SUF(lbfgs) lb;
    REAL grad[ORC_MAXNV], grad_prev[ORC_MAXNV], u_half[ORC_MAXNV], gstep[ORC_MAXNV], dir[ORC_MAXNV],
        gfpr[ORC_MAXNV], u_plus[ORC_MAXNV];
    REAL gamma, L, sigma, tau, cost_value, norm_gfpr, rhs_ls, lhs_ls;
    REAL tol, akkt_tol;
    int iteration;
} SUF(pcache);

static void SUF(pc_reset)(SUF(pcache) * pc)
{
    /* OpEn: PANOCCache::reset (gradient_u_previous is intentionally not cleared) */
    SUF(lbfgs_reset)(&pc->lb);
    pc->lhs_ls = pc->rhs_ls = 0;
    pc->tau = 1;
    pc->L = 0;
    pc->sigma = 0;
    pc->cost_value = 0;
    pc->iteration = 0;
    pc->gamma = 0;
}

#define GAMMA_L_COEFF ((REAL)0.95)
#define LIP_UPDATE_EPS ((REAL)1e-6)
#define MIN_L ((REAL)1e-10)
#define MAX_L ((REAL)1e9)
#define MAX_LIP_ITERS 10
#define MAX_LS_ITERS 10

static void SUF(gradient_step)(SUF(pcache) * pc, const REAL *u, int n)
{
    for (int i = 0; i < n; ++i) pc->gstep[i] = u[i] - pc->gamma * pc->grad[i];
}
static void SUF(half_step)(SUF(ctx) * cx, SUF(pcache) * pc)
{
    for (int i = 0; i < cx->n; ++i) pc->u_half[i] = pc->gstep[i];
    SUF(project_U)(cx->pr, pc->u_half, cx->n);
}

/* OpEn: PANOCEngine::init + LipschitzEstimator::estimate_local_lipschitz.
 * Note: the estimator leaves u perturbed by h (u <- u + h); restated faithfully. */
static void SUF(panoc_init)(SUF(ctx) * cx, SUF(pcache) * pc, REAL *u)
{
    const int n = cx->n;
    SUF(pc_reset)(pc);
    pc->cost_value = SUF(cost)(cx, u);
    SUF(gradf)(cx, u, pc->grad);
    REAL h[ORC_MAXNV], g2[ORC_MAXNV];
    for (int i = 0; i < n; ++i) {
        REAL e = (REAL)cx->op->lip_eps * u[i];
        h[i] = e > (REAL)cx->op->lip_delta ? e : (REAL)cx->op->lip_delta;
    }
    REAL norm_h = SUF(norm2)(h, n);
    for (int i = 0; i < n; ++i) u[i] += h[i];
    SUF(gradf)(cx, u, g2);
    for (int i = 0; i < n; ++i) g2[i] -= pc->grad[i];
    pc->L = SUF(norm2)(g2, n) / norm_h;
    pc->gamma = GAMMA_L_COEFF / SUF(rmax)(pc->L, MIN_L);
    pc->sigma = ((REAL)1 - GAMMA_L_COEFF) / ((REAL)4 * pc->gamma);
    SUF(gradient_step)(pc, u, n);
    SUF(half_step)(cx, pc);
}

/* OpEn: PANOCEngine::step; returns 1 to continue, 0 when the exit condition holds */
static int SUF(panoc_step)(SUF(ctx) * cx, SUF(pcache) * pc, REAL *u)
{
    const int n = cx->n;
    /* cache_previous_gradient */
    if (pc->iteration >= 1)
        for (int i = 0; i < n; ++i) pc->grad_prev[i] = pc->grad[i];
    /* compute_fpr */
    for (int i = 0; i < n; ++i) pc->gfpr[i] = u[i] - pc->u_half[i];
    pc->norm_gfpr = SUF(norm2)(pc->gfpr, n);
    /* exit_condition: ||gamma*fpr|| < tol  &&  AKKT residual < akkt_tol.
     * OpEn: PANOCCache::akkt_residual -- source (recalled): sum (gamma_fpr_i + gamma*(df_i - dfp_i))^2, its doc comment:
     * ||gamma^{-1}(u - u_plus) + df(u) - df(u_plus)||. The two differ by the factor gamma; op->akkt_form selects.
     * Note: cache_previous_gradient() has just copied df into df_prev (iteration >= 1), so the gradient difference is
     * zero after the first iteration and the source form reduces to ||gamma*fpr||. */
    {
        REAL r = 0;
        for (int ii = 0; ii < n; ++ii) {
            const int i = ORC_IDX(ii, n);
            REAL t = pc->gfpr[i] + pc->gamma * (pc->grad[i] - pc->grad_prev[i]);
            r += t * t;
        }
        r = SUF(rsqrt)(r);
        if (cx->op->akkt_form) r /= pc->gamma; /* documented form */
        if (pc->norm_gfpr < pc->tol && r < pc->akkt_tol) return 0;
        cx->t_kind = 0;
        cx->t_lip = cx->t_nls = 0;
        cx->t_cbfgs = -1;
        if (cx->trace) cx->t_steep = (double)SUF(norm2)(pc->grad, n) / ((double)SUF(rabs)(pc->cost_value) + 1e-300);
        if (cx->outer_margin_set && pc->iteration == 0) { /* first iteration of an outer iteration: the ALM decisions before it */
            SUF(note_margin)(cx, cx->outer_margin, 5);
            cx->outer_margin_set = 0;
        }
        {   /* the exit test said "continue": by how much (the test that binds: the larger of the two ratios) */
            double m1 = ((double)pc->norm_gfpr - (double)pc->tol) / (double)pc->tol;
            double m2 = ((double)r - (double)pc->akkt_tol) / (double)pc->akkt_tol;
            SUF(note_margin)(cx, m1 > m2 ? m1 : m2, 4);
        }
    }
    /* update_lipschitz_constant */
    {
        REAL cost_half = SUF(cost)(cx, pc->u_half);
        cx->n_points++;
        pc->cost_value = SUF(cost)(cx, u); /* (same point as the last gradient: not a new one) */
        int it = 0;
        for (;;) {
            REAL ip = SUF(dot)(pc->grad, pc->gfpr, n);
            REAL rhs = pc->cost_value + LIP_UPDATE_EPS * SUF(rabs)(pc->cost_value) - ip +
                       (GAMMA_L_COEFF / ((REAL)2 * pc->gamma)) * (pc->norm_gfpr * pc->norm_gfpr);
            SUF(note_margin)(cx, ((double)cost_half - (double)rhs) / ((double)SUF(rabs)(pc->cost_value) + 1e-300), 1);
            if (!(cost_half > rhs && it < MAX_LIP_ITERS && pc->L < MAX_L)) break;
            SUF(lbfgs_reset)(&pc->lb);
            pc->L *= (REAL)2;
            pc->gamma /= (REAL)2;
            SUF(gradient_step)(pc, u, n);
            SUF(half_step)(cx, pc);
            cost_half = SUF(cost)(cx, pc->u_half);
            cx->n_points++;
            for (int i = 0; i < n; ++i) pc->gfpr[i] = u[i] - pc->u_half[i];
            pc->norm_gfpr = SUF(norm2)(pc->gfpr, n);
            ++it;
        }
        cx->t_lip = it;
        pc->sigma = ((REAL)1 - GAMMA_L_COEFF) / ((REAL)4 * pc->gamma);
    }
    /* lbfgs_direction */
    {
        double mg;
        SUF(lbfgs_update)(&pc->lb, cx->op, pc->gfpr, u, &cx->t_cbfgs, &mg);
        if (cx->t_cbfgs >= 0) SUF(note_margin)(cx, mg, 3);
    }
    if (pc->iteration > 0) {
        for (int i = 0; i < n; ++i) pc->dir[i] = pc->gfpr[i];
        SUF(lbfgs_apply)(&pc->lb, pc->dir);
    }
    if (pc->iteration == 0) {
        /* update_no_linesearch */
        for (int i = 0; i < n; ++i) u[i] = pc->u_half[i];
        pc->cost_value = SUF(cost)(cx, u);
        SUF(gradf)(cx, u, pc->grad);
        SUF(gradient_step)(pc, u, n);
        SUF(half_step)(cx, pc);
    } else {
        /* linesearch: compute_rhs_ls */
        REAL dist2 = 0;
        for (int ii = 0; ii < n; ++ii) {
            const int i = ORC_IDX(ii, n);
            REAL t = pc->gstep[i] - pc->u_half[i];
            dist2 += t * t;
        }
        REAL fbe = pc->cost_value - (REAL)0.5 * pc->gamma * SUF(dot)(pc->grad, pc->grad, n) +
                   (REAL)0.5 * dist2 / pc->gamma;
        pc->rhs_ls = fbe - pc->sigma * pc->norm_gfpr * pc->norm_gfpr;
        pc->tau = 1;
        int nls = 0;
        for (;;) {
            /* line_search_condition */
            for (int i = 0; i < n; ++i)
                pc->u_plus[i] = u[i] - ((REAL)1 - pc->tau) * pc->gfpr[i] - pc->tau * pc->dir[i];
            pc->cost_value = SUF(cost)(cx, pc->u_plus);
            SUF(gradf)(cx, pc->u_plus, pc->grad);
            SUF(gradient_step)(pc, pc->u_plus, n);
            SUF(half_step)(cx, pc);
            REAL d2 = 0;
            for (int ii = 0; ii < n; ++ii) {
                const int i = ORC_IDX(ii, n);
                REAL t = pc->gstep[i] - pc->u_half[i];
                d2 += t * t;
            }
            pc->lhs_ls = pc->cost_value - (REAL)0.5 * pc->gamma * SUF(dot)(pc->grad, pc->grad, n) +
                         (REAL)0.5 * d2 / pc->gamma;
            SUF(note_margin)(cx, ((double)pc->lhs_ls - (double)pc->rhs_ls) / ((double)SUF(rabs)(pc->rhs_ls) + 1e-300), 2);
            if (!(pc->lhs_ls > pc->rhs_ls && nls < MAX_LS_ITERS)) break;
            pc->tau /= (REAL)2;
            ++nls;
        }
        cx->t_nls = nls;
        /* (OpEn sets tau = 0 / u <- u_half when nls == MAX but then overwrites u with u_plus) */
        for (int i = 0; i < n; ++i) u[i] = pc->u_plus[i];
    }
    if (cx->trace && cx->n_rec < cx->max_rec) {
        double *t = cx->trace + (size_t)cx->n_rec * (ORC_TRACE_HEAD + n);
        t[0] = cx->outer;
        t[1] = pc->iteration;            /* index of the iteration just completed, within its outer iteration */
        t[2] = cx->t_lip;                /* Lipschitz doublings (gamma halvings) */
        t[3] = cx->t_nls;                /* line-search halvings of tau (0 on iteration 0: no line search) */
        t[4] = cx->t_cbfgs;              /* -1 nothing tested, 0 pair rejected, 1 pair accepted */
        t[5] = pc->lb.active;
        t[6] = (double)pc->gamma;
        t[7] = (double)pc->norm_gfpr;    /* ||gamma * fpr|| at the head of this iteration */
        t[8] = (double)pc->cost_value;   /* psi at the new iterate */
        t[9] = pc->iteration == 0 ? 0.0 : (double)pc->tau;
        t[10] = cx->n_cost;
        t[11] = cx->n_grad;
        t[12] = cx->t_margin;            /* smallest relative margin of this iteration's decisions ... */
        t[13] = cx->t_kind;              /* ... and which decision it was (see note_margin) */
        t[14] = (double)cx->c;
        t[15] = cx->t_steep;             /* ||grad psi|| / |psi| at the head of this iteration: what a distance in u is worth
                                            in relative psi -- at penalties of 1e6 a 1e-13 apart is a 1e-7 apart in psi */
        for (int i = 0; i < n; ++i) t[ORC_TRACE_HEAD + i] = (double)u[i];
        cx->n_rec++;
    }
    pc->iteration++;
    return 1;
}

/* OpEn: PANOCOptimizer::solve. Returns inner exit status (0 converged, 1 max iterations). */
static int SUF(panoc_solve)(SUF(ctx) * cx, SUF(pcache) * pc, REAL *u, int max_iter, int *iters, REAL *last_fpr,
                            double t_end)
{
    SUF(panoc_init)(cx, pc, u);
    int num_iter = 0, cont_iters = 1, cont_time = 1;
    int flag = SUF(panoc_step)(cx, pc, u);
    while (flag && cont_iters && cont_time) {
        num_iter++;
        cont_iters = num_iter < max_iter;
        if (t_end > 0) cont_time = orc_now() <= t_end;   /* (remaining time checked once per iteration, like OpEn) */
        if (cx->op->max_evals > 0 && cx->n_points >= cx->op->max_evals) cont_time = 0; /* the same test on the count */
        flag = SUF(panoc_step)(cx, pc, u);
    }
    for (int i = 0; i < cx->n; ++i) u[i] = pc->u_half[i];
    *iters = num_iter;
    *last_fpr = pc->norm_gfpr;
    return !cont_iters ? 1 : !cont_time ? 2 : 0;
}

/* OpEn: alm::AlmOptimizer::solve / step */
static int SUF(orc_solve_impl)(const orc_problem *pr, const orc_options *op, const REAL *p, REAL *u, REAL *y,
                               orc_result *res, double *trace, int max_rec, int *n_rec)
{
    if (pr->N > ORC_MAXN || pr->Ndyn > ORC_MAXDYN || op->lbfgs_mem > ORC_MAX_MEM) return -1;
    const int N = pr->N, n = 2 * N, n1 = 2 * N, n2 = pr->Ndyn;
    SUF(pcache) *pc = (SUF(pcache) *)calloc(1, sizeof(SUF(pcache)));
    if (!pc) return -2;
    pc->lb.n = n;
    pc->lb.mem = op->lbfgs_mem;
    pc->lb.gamma = 1;
    pc->tol = (REAL)op->tolerance;
    pc->akkt_tol = (REAL)op->initial_tolerance;
    SUF(ctx) cx = {pr, op, p, n, (REAL)op->initial_penalty, y, 0, 0, 0, trace, max_rec, 0, 0, 0, 0, -1, 0, 0.0, 0.0, 0.0, 0};

    REAL y_plus[ORC_MAXNV], F1[ORC_MAXNV], F2[ORC_MAXDYN];
    REAL dyn = 0, dyn_plus = 0, f2n = 0, f2n_plus = 0, last_fpr = 0;
    int alm_iter = 0, inner_total = 0, outer = 0, status = 0, converged = 0;
    const REAL SMALL = (REAL)REAL_EPS;

    const double t_end = op->max_time_s > 0 ? orc_now() + op->max_time_s : 0.0;
    int out_of_time = 0;
    REAL *trig = 0;
    if (op->hoist_trig) {
        SUF(offs) o_;
        SUF(make_offs)(pr, &o_);
        trig = (REAL *)malloc(sizeof(REAL) * 2 * (size_t)pr->Ndyn * (size_t)(N + 1));
        if (trig) {
            for (int j = 0; j < pr->Ndyn; ++j)
                for (int t = 0; t <= N; ++t) {
                    const REAL ang = p[o_.od + j * 6 * (N + 1) + t * 6 + 4];
                    trig[2 * (j * (N + 1) + t)] = SUF(rcos)(ang);
                    trig[2 * (j * (N + 1) + t) + 1] = SUF(rsin)(ang);
                }
            SUF(tl_trig) = trig;
        }
    }
    for (int it = 0; it < op->max_outer; ++it) {
        if (it > 0 && ((t_end > 0 && orc_now() > t_end) || (op->max_evals > 0 && cx.n_points >= op->max_evals))) {
            /* no time (no evaluation budget) left for another outer iteration */
            out_of_time = 1;
            break;
        }
        outer++;
        cx.outer = outer;
        /* project y on Y = [-1e12, 1e12]^n1 */
        for (int i = 0; i < n1; ++i) y[i] = SUF(rmin)(SUF(rmax)(y[i], (REAL)-1e12), (REAL)1e12);
        int inner_iters;
        int inner_status = SUF(panoc_solve)(&cx, pc, u, op->max_inner, &inner_iters, &last_fpr, t_end);
        inner_total += inner_iters;
        status = inner_status;
        if (inner_status == 2) out_of_time = 1; /* the clock ran out inside this inner solve (also when it was the last
                                                   outer iteration: the exit status is OutOfTime, not Iterations) */
        /* update_lagrange_multipliers: y+ = y + c [F1(u) - Proj_C(F1(u) + y/c)] ; F2 norm */
        SUF(core)(pr, u, 0, 0, p, 0, F1, F2, 0, 0);
        cx.n_points++;
        for (int i = 0; i < n1; ++i) {
            REAL lo = i < N ? (REAL)pr->lin_acc_min : (REAL)(-pr->ang_acc_max);
            REAL hi = i < N ? (REAL)pr->lin_acc_max : (REAL)pr->ang_acc_max;
            REAL z = F1[i] + y[i] / cx.c;
            REAL pz = SUF(rmin)(SUF(rmax)(z, lo), hi);
            y_plus[i] = y[i] + cx.c * (F1[i] - pz);
        }
        f2n_plus = SUF(norm2)(F2, n2);
        {
            REAL s = 0;
            for (int ii = 0; ii < n1; ++ii) {
                const int i = ORC_IDX(ii, n1);
                s += (y_plus[i] - y[i]) * (y_plus[i] - y[i]);
            }
            dyn_plus = SUF(rsqrt)(s);
        }
        /* is_exit_criterion_satisfied */
        int c1 = (alm_iter > 0) && (dyn_plus <= cx.c * (REAL)op->delta_tolerance + SMALL);
        int c2 = (n2 == 0) || (f2n_plus <= (REAL)op->delta_tolerance + SMALL);
        int c3 = pc->akkt_tol <= (REAL)op->tolerance + SMALL;
        if (c1 && c2 && c3) {
            converged = 1;
            break;
        }
        /* is_penalty_stall_criterion */
        int stall = (alm_iter == 0) || ((dyn_plus <= (REAL)op->sufficient_decrease * dyn + SMALL) &&
                                        (n2 == 0 || f2n_plus <= (REAL)op->sufficient_decrease * f2n + SMALL));
        if (trace) { /* how close the outer loop's decisions were: exit criteria 1 and 2, the two halves of the stall test */
            double m = 1e300, t;
            const double tiny = 1e-300;
            if (alm_iter > 0) {
                t = ((double)dyn_plus - (double)cx.c * op->delta_tolerance) / ((double)cx.c * op->delta_tolerance + tiny);
                if (fabs(t) < fabs(m)) m = t;
                t = ((double)dyn_plus - op->sufficient_decrease * (double)dyn) / ((double)dyn_plus + tiny);
                if (fabs(t) < fabs(m)) m = t;
                if (n2 > 0 && (f2n_plus != 0 || f2n != 0)) { /* (0 <= 0.1 * 0 is not a near-tie: both are exact zeros of max(0, .)) */
                    t = ((double)f2n_plus - op->sufficient_decrease * (double)f2n) / ((double)f2n_plus + tiny);
                    if (fabs(t) < fabs(m)) m = t;
                }
            }
            if (n2 > 0) {
                t = ((double)f2n_plus - op->delta_tolerance) / (op->delta_tolerance + tiny);
                if (fabs(t) < fabs(m)) m = t;
            }
            cx.outer_margin = m;
            cx.outer_margin_set = 1;
        }
        if (!stall) cx.c *= (REAL)op->penalty_update;
        /* update_inner_akkt_tolerance */
        pc->akkt_tol = SUF(rmax)(pc->akkt_tol * (REAL)op->inner_tol_update, (REAL)op->tolerance);
        /* final_cache_update */
        alm_iter++;
        dyn = dyn_plus;
        f2n = f2n_plus;
        for (int i = 0; i < n1; ++i) y[i] = y_plus[i];
        SUF(pc_reset)(pc);
    }
    if (!converged) status = out_of_time ? 2 : 1; /* out of time / outer iterations exhausted */
    if (res) {
        REAL f;
        SUF(core)(pr, u, 0, 0, p, &f, 0, 0, 0, 0);
        res->cost = (double)f;
        res->status = status;
        res->outer_iters = outer;
        res->inner_iters = inner_total;
        res->n_cost_evals = cx.n_cost;
        res->n_grad_evals = cx.n_grad;
        res->last_fpr = (double)last_fpr;
        res->delta_y_norm = (double)dyn_plus;
        res->f2_norm = (double)f2n_plus;
        res->penalty = (double)cx.c;
        res->n_points = cx.n_points;
        res->reserved_ = 0;
    }
    if (n_rec) *n_rec = cx.n_rec;
    SUF(tl_trig) = 0;
    free(trig);
    free(pc);
    return 0;
}

int SUF(orc_solve)(const orc_problem *pr, const orc_options *op, const REAL *p, REAL *u, REAL *y, orc_result *res)
{
    return SUF(orc_solve_impl)(pr, op, p, u, y, res, 0, 0, 0);
}

int SUF(orc_solve_trace)(const orc_problem *pr, const orc_options *op, const REAL *p, REAL *u, REAL *y, orc_result *res,
                         double *trace, int max_rec, int *n_rec)
{
    return SUF(orc_solve_impl)(pr, op, p, u, y, res, trace, max_rec, n_rec);
}

int SUF(orc_solve_batch)(const orc_problem *pr, const orc_options *op, const REAL *P, int B, REAL *U, orc_result *res,
                         int nthreads)
{
    SUF(offs) o;
    SUF(make_offs)(pr, &o);
    const int n = 2 * pr->N;
    int err = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int b = 0; b < B; ++b) {
        REAL y[ORC_MAXNV];
        for (int i = 0; i < n; ++i) {
            U[(size_t)b * n + i] = 0;
            y[i] = 0;
        }
        int e = SUF(orc_solve)(pr, op, P + (size_t)b * o.np, U + (size_t)b * n, y, res ? res + b : 0);
        if (e) {
#pragma omp atomic write
            err = e;
        }
    }
    return err;
}

#undef GAMMA_L_COEFF
#undef LIP_UPDATE_EPS
#undef MIN_L
#undef MAX_L
#undef MAX_LIP_ITERS
#undef MAX_LS_ITERS
#undef ORC_IDX
