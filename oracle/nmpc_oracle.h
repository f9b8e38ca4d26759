/*
 * nmpc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the NMPC solve path of Woodenonez/DyObAv-MPCnWTA-Warehouse:
 *
 *   problem definition  : src/pkg_mpc_tracker/solver_build/mpc_builder.py:45-174,
 *                         mpc_cost.py, mpc_helper.py, basic_motion_model/motion_model.py:141-163
 *   algorithm           : third-party, NOT vendored in the reference tree:
 *                         opengen==0.6.13 (requirements.txt) -> Rust crate `optimization_engine`
 *                         (0.7.x line: core::panoc, alm) + crate `lbfgs` (0.2.x). Its published
 *                         algorithm (PANOC with L-BFGS directions inside an ALM/penalty outer loop)
 *                         is restated here from the upstream project; reference call sites:
 *                         mpc_builder.py:171-203 (build), trajectory_tracker.py:362-367 (run).
 *
 * PARITY STATUS
 *   - problem functions f, F1, F2 : PINNED against golden vectors produced by running the reference's own
 *     mpc_builder/mpc_cost/mpc_helper/motion_model Python code in the authoring container
 *     (tests/golden/make_golden.py -> tests/golden/problem_*.npz) and against the known answers of
 *     src/tests/test_mpc_builder.py:16-253.
 *   - solver algorithm (PANOC/L-BFGS/ALM iterate path) : *** PARITY UNPINNED ***. The OpEn crate cannot be
 *     built here (no cargo/rustc, no casadi/opengen, no network) and the reference's tests pin no solver
 *     output. The restatement is checked only through solver-independent properties (KKT residual,
 *     agreement with an independent scipy L-BFGS-B minimisation of the same psi).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef NMPC_ORACLE_H
#define NMPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Problem dimensions and robot constants (config/mpc_default.yaml:6-31 of the reference). */
typedef struct {
    int32_t N;        /* N_hor                      */
    int32_t Nother;   /* other robots               */
    int32_t Nstc;     /* static obstacles (4 edges, nstcobs = 12) */
    int32_t Ndyn;     /* dynamic obstacles (ndynobs = 6)          */
    double ts;
    double lin_vel_min, lin_vel_max, ang_vel_max;   /* set U (mpc_builder.py:151-153) */
    double lin_acc_min, lin_acc_max, ang_acc_max;   /* set C (mpc_builder.py:160-166) */
    double vehicle_width, vehicle_margin, social_margin;
} orc_problem;

/* Solver options (opengen SolverConfiguration defaults + mpc_builder.py:187-195). */
typedef struct {
    double tolerance;            /* epsilon: PANOC FPR tolerance and final AKKT tolerance (1e-4) */
    double initial_tolerance;    /* initial inner AKKT tolerance (1e-4)                         */
    double delta_tolerance;      /* ALM/PM infeasibility tolerance (1e-4)                        */
    int32_t max_outer;           /* 10  */
    int32_t max_inner;           /* 500 */
    int32_t lbfgs_mem;           /* 10 (<= ORC_MAX_MEM) */
    double initial_penalty;      /* 10 (mpc_builder.py:188) */
    double penalty_update;       /* 5   */
    double inner_tol_update;     /* 0.1 */
    double sufficient_decrease;  /* 0.1 */
    double lip_delta;            /* 1e-12 (Lipschitz estimator delta)   */
    double lip_eps;              /* 1e-6  (Lipschitz estimator epsilon) */
    double cbfgs_alpha;          /* 1.0  */
    double cbfgs_eps;            /* 1e-8 */
    double sy_eps;               /* 1e-10 */
    int32_t akkt_form;           /* AKKT residual of the inner exit test: 0 = OpEn source form (recalled)
                                    ||gamma*fpr + gamma*(df - df_prev)||, 1 = OpEn documentation ||fpr + df - df_prev||
                                    (= the former / gamma). See DESIGN.md "AKKT residual". */
    int32_t hoist_trig;          /* 0 (default): cos / sin of the ellipse angles on every evaluation, as the reference's
                                    CasADi-generated code does; 1: once per solve (same bits; CPU-baseline speed option) */
    double max_time_s;           /* wall-clock budget of one solve (the reference: with_max_duration_micros = 0.1 s,
                                    mpc_builder.py:189): the inner loop stops once it is used up and no further outer
                                    iteration is started (status 2); 0 = none (default: iteration caps only) */
    int32_t max_evals;           /* the same cap as a COUNT (deterministic, what nmpc_config.max_evaluations does on the
                                    device): budget of "points" -- distinct arguments at which psi (and possibly grad psi)
                                    is evaluated: every gradient call, every psi(u_half) of the Lipschitz test, the
                                    F1 / F2 evaluation behind each inner solve (orc_result.n_points; the kernels'
                                    info[4]). Tested where OpEn reads its clock: after every inner iteration the loop
                                    goes on only while n_points < max_evals, and no further outer iteration is started
                                    once n_points >= max_evals (status 2). As in OpEn's `while step() && flags` loop the step
                                    that follows a failed test still runs: overshoot <= 2 x 22 + 1 points. 0 = none */
    int32_t reserved_;
} orc_options;

typedef struct {
    double cost;                 /* f(u*) : psi with c = 0 */
    int32_t status;              /* 0 Converged, 1 NotConvergedIterations, 2 NotConvergedOutOfTime */
    int32_t outer_iters;
    int32_t inner_iters;
    int32_t n_cost_evals;
    int32_t n_grad_evals;
    double last_fpr;
    double delta_y_norm;
    double f2_norm;
    double penalty;
    int32_t n_points;            /* points evaluated (see orc_options.max_evals) = what the HIP kernels report as info[4] */
    int32_t reserved_;
} orc_result;

#define ORC_MAX_MEM 32

int orc_np(const orc_problem *pr);   /* length of the parameter vector p */

void orc_default_options(orc_options *o);

/* f, F1[2N], F2[Ndyn] of mpc_builder.py:45-174 */
void orc_eval_f64(const orc_problem *pr, const double *u, const double *p, double *f, double *F1, double *F2);
void orc_eval_f32(const orc_problem *pr, const float *u, const float *p, float *f, float *F1, float *F2);

/* psi(u; xi=(c,y), p) and (if grad != NULL) its gradient by the hand-written adjoint */
void orc_psi_f64(const orc_problem *pr, const double *u, double c, const double *y, const double *p,
                 double *psi, double *grad);
void orc_psi_f32(const orc_problem *pr, const float *u, float c, const float *y, const float *p,
                 float *psi, float *grad);

/* One full ALM/PANOC solve. u: in = initial guess, out = solution. y: in/out multipliers (length 2N). */
int orc_solve_f64(const orc_problem *pr, const orc_options *op, const double *p, double *u, double *y,
                  orc_result *res);
int orc_solve_f32(const orc_problem *pr, const orc_options *op, const float *p, float *u, float *y,
                  orc_result *res);

/* The same double-precision solver with its sums associated differently (reverse accumulation order, rollout as
 * X0 + running sum): same algorithm, same decisions rules, different rounding -- the oracle's own noise floor under
 * re-association (tests/accuracy_protocol.py). */
void orc_psi_r64(const orc_problem *pr, const double *u, double c, const double *y, const double *p,
                 double *psi, double *grad);
int orc_solve_r64(const orc_problem *pr, const orc_options *op, const double *p, double *u, double *y,
                  orc_result *res);
int orc_solve_batch_r64(const orc_problem *pr, const orc_options *op, const double *P, int B, double *U,
                        orc_result *res, int nthreads);

/* Iteration trace of one solve (first-divergence audit, tests/accuracy_protocol.py): like orc_solve_*, plus one record
 * of ORC_TRACE_HEAD + 2N doubles per completed inner iteration, up to max_rec records:
 *   [0] outer iteration (1-based)  [1] inner iteration index within it  [2] Lipschitz doublings  [3] line-search halvings
 *   [4] L-BFGS pair: -1 not tested / 0 rejected / 1 accepted  [5] pairs in the buffer  [6] gamma  [7] ||gamma fpr||
 *   [8] psi at the new iterate  [9] tau  [10] cost evaluations so far  [11] gradient evaluations so far
 *   [12] smallest relative margin of the iteration's discrete decisions  [13] which: 1 Lipschitz test, 2 line-search
 *   test, 3 pair acceptance, 4 exit test, 5 the outer loop's tests before this inner solve (exit criteria, penalty
 *   stall test; noted on the first record of an outer iteration)  [14] penalty c  [15] ||grad psi|| / |psi| at the head of the iteration
 *   [16..] u after the iteration */
#define ORC_TRACE_HEAD 16
int orc_solve_trace_f64(const orc_problem *pr, const orc_options *op, const double *p, double *u, double *y,
                        orc_result *res, double *trace, int max_rec, int *n_rec);
int orc_solve_trace_r64(const orc_problem *pr, const orc_options *op, const double *p, double *u, double *y,
                        orc_result *res, double *trace, int max_rec, int *n_rec);
int orc_solve_trace_f32(const orc_problem *pr, const orc_options *op, const float *p, float *u, float *y,
                        orc_result *res, double *trace, int max_rec, int *n_rec);

/* Batch of independent solves (zero initial guess, zero multipliers), OpenMP over instances. */
int orc_solve_batch_f64(const orc_problem *pr, const orc_options *op, const double *P, int B, double *U,
                        orc_result *res, int nthreads);
int orc_solve_batch_f32(const orc_problem *pr, const orc_options *op, const float *P, int B, float *U,
                        orc_result *res, int nthreads);

/* primitives exposed for the known-answer tests (tests/test_mpc_builder.py of the reference) */
double orc_dist2_to_lineseg(double px, double py, double ax, double ay, double bx, double by);
double orc_inside_ellipse(double px, double py, double cx, double cy, double rx, double ry, double ang);
double orc_inside_cvx_polygon(double px, double py, const double *b, const double *a0, const double *a1, int ne);
void orc_unicycle_rk4(double ts, const double *s, const double *a, double *s_next);

#ifdef __cplusplus
}
#endif
#endif
