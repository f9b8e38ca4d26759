/*
 * nmpc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE). See nmpc_oracle.h.
 * Instantiates nmpc_oracle_impl.h for double (the checker) and float (to study fp32 behaviour of the
 * same algorithm on the CPU).
 */
#include "nmpc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double orc_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_np(const orc_problem *pr)
{
    /* ref: SURVEY.md 8a; mpc_builder.py:47-60 */
    return 18 + 6 * pr->N + 3 * pr->Nother * (pr->N + 1) + 12 * pr->Nstc + 6 * pr->Ndyn * (pr->N + 1);
}

void orc_default_options(orc_options *o)
{
    /* OpEn: opengen.config.SolverConfiguration defaults; initial penalty from mpc_builder.py:188 */
    o->tolerance = 1e-4;
    o->initial_tolerance = 1e-4;
    o->delta_tolerance = 1e-4;
    o->max_outer = 10;
    o->max_inner = 500;
    o->lbfgs_mem = 10;
    o->initial_penalty = 10.0;
    o->penalty_update = 5.0;
    o->inner_tol_update = 0.1;
    o->sufficient_decrease = 0.1;
    o->lip_delta = 1e-12;
    o->lip_eps = 1e-6;
    o->cbfgs_alpha = 1.0;
    o->cbfgs_eps = 1e-8;
    o->sy_eps = 1e-10;
    o->akkt_form = 0;
    o->hoist_trig = 0;
    o->max_time_s = 0.0;
    o->max_evals = 0;
    o->reserved_ = 0;
}

/* ---------------- double ---------------- */
#define REAL double
#define SUF(x) x##_f64
#define REAL_MIN_POS DBL_MIN
#define REAL_EPS DBL_EPSILON
static inline double rcos_f64(double x) { return cos(x); }
static inline double rsin_f64(double x) { return sin(x); }
static inline double rsqrt_f64(double x) { return sqrt(x); }
static inline double rabs_f64(double x) { return fabs(x); }
static inline double rpow_f64(double x, double y) { return pow(x, y); }
static inline int risfinite_f64(double x) { return isfinite(x); }
#include "nmpc_oracle_impl.h"
#undef REAL
#undef SUF
#undef REAL_MIN_POS
#undef REAL_EPS

/* ---------------- float ---------------- */
#define REAL float
#define SUF(x) x##_f32
#define REAL_MIN_POS FLT_MIN
#define REAL_EPS FLT_EPSILON
static inline float rcos_f32(float x) { return cosf(x); }
static inline float rsin_f32(float x) { return sinf(x); }
static inline float rsqrt_f32(float x) { return sqrtf(x); }
static inline float rabs_f32(float x) { return fabsf(x); }
static inline float rpow_f32(float x, float y) { return powf(x, y); }
static inline int risfinite_f32(float x) { return isfinite(x); }
#include "nmpc_oracle_impl.h"
#undef REAL
#undef SUF
#undef REAL_MIN_POS
#undef REAL_EPS

/* ---------------- double, sums re-associated (see ORC_REASSOC in nmpc_oracle_impl.h) ---------------- */
#define ORC_REASSOC 1
#define REAL double
#define SUF(x) x##_r64
#define REAL_MIN_POS DBL_MIN
#define REAL_EPS DBL_EPSILON
static inline double rcos_r64(double x) { return cos(x); }
static inline double rsin_r64(double x) { return sin(x); }
static inline double rsqrt_r64(double x) { return sqrt(x); }
static inline double rabs_r64(double x) { return fabs(x); }
static inline double rpow_r64(double x, double y) { return pow(x, y); }
static inline int risfinite_r64(double x) { return isfinite(x); }
#include "nmpc_oracle_impl.h"
#undef REAL
#undef SUF
#undef REAL_MIN_POS
#undef REAL_EPS
#undef ORC_REASSOC

/* ---------------- primitives for the known-answer tests ---------------- */
double orc_dist2_to_lineseg(double px, double py, double ax, double ay, double bx, double by)
{
    double tx, ty;
    return seg_d2_f64(px, py, ax, ay, bx, by, &tx, &ty);
}

double orc_inside_ellipse(double px, double py, double cx, double cy, double rx, double ry, double ang)
{
    /* ref: pkg_mpc_tracker/solver_build/mpc_helper.py:38-52 */
    double dx = px - cx, dy = py - cy, ca = cos(ang), sa = sin(ang);
    double a = dx * ca + dy * sa, b = dx * sa - dy * ca;
    return ell_ind_f64(a * a, b * b, rx, ry);
}

double orc_inside_cvx_polygon(double px, double py, const double *b, const double *a0, const double *a1, int ne)
{
    /* ref: pkg_mpc_tracker/solver_build/mpc_helper.py:54-75 */
    double ind = 1.0;
    for (int e = 0; e < ne; ++e) {
        double h = b[e] - a0[e] * px - a1[e] * py;
        ind *= h > 0 ? h : 0.0;
    }
    return ind;
}

void orc_unicycle_rk4(double ts, const double *s, const double *a, double *s_next)
{
    /* literal RK4 as written in ref: basic_motion_model/motion_model.py:141-163 (not the closed form),
     * so that the closed form used in core() is checked against it */
    double k[4][3], st[3];
    memcpy(st, s, sizeof st);
    for (int i = 0; i < 4; ++i) {
        k[i][0] = ts * a[0] * cos(st[2]);
        k[i][1] = ts * a[0] * sin(st[2]);
        k[i][2] = ts * a[1];
        double f = i < 2 ? 0.5 : 1.0;
        if (i < 3)
            for (int d = 0; d < 3; ++d) st[d] = s[d] + f * k[i][d];
    }
    for (int d = 0; d < 3; ++d) s_next[d] = s[d] + (k[0][d] + 2 * k[1][d] + 2 * k[2][d] + k[3][d]) / 6.0;
}
