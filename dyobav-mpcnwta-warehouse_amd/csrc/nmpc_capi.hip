// nmpc_capi.hip -- C ABI (include/nmpc_hip.h) over the gfx950 kernels in nmpc_device.h.
// Host side of the drop-in boundary for `solver.run(p)` of the reference
// (/root/reference/src/pkg_mpc_tracker/trajectory_tracker.py:54-66, :362). No CPU fallback: every entry point
// either launches the HIP kernels or returns an error code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/nmpc_hip.h"
#include "nmpc_assemble.h"
#include "nmpc_device.h"
#include "nmpc_spec.h"
#include "nmpc_hypotheses.h"
#include "nmpc_step.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(e_ == hipErrorOutOfMemory ? NMPC_ERR_OUT_OF_MEMORY : NMPC_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                              \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(NMPC_ERR_OUT_OF_MEMORY, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

bool is_device_ptr(const void* ptr)
{
    if (!ptr) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // plain host memory: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

struct Layout {
    int np, off_rs, off_rv, off_c0, off_c, off_os, off_od, off_qstc, off_qdyn;
    int lds_alpha, lds_poly, lds_seg, lds_seginv, lds_fl0, lds_fl, lds_iflag, lds_hist, lds_rho, lds_total;
    int lds_xch, lds_total_spec; // latency mode: exchange area + the other wavefronts' parking areas behind lds_total
    int lds_park, lds_deepsc;
    int coop_lanes, lds_t0c; // cooperative kernels: lanes per plane of the exchange area; t = 0 rows of the compressed global table
    int lds_xch_coop, lds_total_coop; // cooperative mode: two shared parking areas, then the partial-sum exchange area
    int lds_left, lds_left_alpha;     // LDS table of the rows beyond the register-resident ones (cooperative register kernel)
    int dyn_cap;          // obstacle rows provisioned per instance
    int table_entries;    // entries of the obstacle table the kernels index in LDS / the workspace
    int rs;               // > 0: register-resident obstacle table with this many slots per lane (LDS keeps t = 0 only)
    bool glb;             // obstacle table streamed from a global workspace instead of LDS
    long long ws_stride;  // workspace elements per instance (glb only)
};

#ifndef NMPC_MID_SLOTS
#define NMPC_MID_SLOTS 1 // offer the 6-slot register-table kernels (13..18 provisioned rows)
#endif
#ifndef NMPC_SPEC_WPE_F32
#define NMPC_SPEC_WPE_F32 3 // wavefronts per SIMD the fp32 latency kernel is compiled for (caps VGPRs at 168)
#endif
constexpr int kSpecWaves = 4; // wavefronts per instance in latency mode (nmpc_spec.h): automatic choice for batches up to one
                              // workgroup per SIMD; wavefronts of the cooperative kernels
constexpr int kSpecWavesWide = 6; // ... for batches of at most one workgroup per CU (the master + five workers: LIP + 5 candidates a round)
constexpr int kSpecWavesMax = 8;  // most that nmpc_config.latency_waves may ask for
constexpr size_t kLdsLimit = 160 * 1024; // bytes of LDS one workgroup may use on gfx950

int round4(int x) { return (x + 3) & ~3; }
// exchange area of a latency-kernel workgroup of W wavefronts (nmpc_spec.h: xch + command area)
int spec_xch_elems(int W) { return round4(W * (2 * 64 + 4) + 2 * 64 * W + 8); } // (+ 8 command scalars: c, 1/max(c,1), flags, exit, gamma, 1/gamma)

constexpr int kRegSlotsSmall = 4, kRegSlotsMid = 6, kRegSlotsLarge = 14; // compiled register-table sizes (rows = 3 x slots)
// (Mid, round 5: 13..18 provisioned rows -- the reference's shipped yaml provisions 15 -- at the 168-register budget of the 4-slot
//  kernels, three wavefronts per SIMD / four per instance in latency mode, instead of the 256-register 14-slot kernels)
constexpr int kRegSlotsCoop = 12; // one lane per step: 8 cooperating wavefronts (2 per SIMD, 256 registers each) x 12 slots in
                                  // registers (96 rows; 144 with helper lanes, below); the rows beyond those in LDS
constexpr int kCoopRegWaves = 8;
// Horizons of 33..42 steps leave 22..31 lanes of every wavefront without a step: there the kernel is compiled with helper
// lanes (nmpc_device.h, HLP) that take a third row in each pass of two slots -- 8 x 18 = 144 rows in registers.
bool coop_helper_lanes(int N) { return N >= 33 && N <= 42; }
int coop_reg_rows(int N) { return kCoopRegWaves * (coop_helper_lanes(N) ? 3 * (kRegSlotsCoop / 2) : kRegSlotsCoop); }

// coop_rs: layout of the cooperative register-table kernel (fp32, one lane per step): 4 x kRegSlotsCoop rows in the
// registers of the four wavefronts, the t = 0 snapshot of all rows and the full table of the remaining rows in LDS --
// nothing is streamed from global memory. L.rs = 0 on return if the configuration does not qualify.
// reg64: layout of the fp64 register-table kernel (one wavefront per SIMD, 512 registers: 14 slots of 9 doubles = 252 of
// them) -- offered for 13..42 provisioned rows and N <= 21; used for large batches where the 72-byte entries of the fp64
// LDS table leave room for fewer than four instances per CU (nmpc_create decides).
Layout make_layout(const nmpc_config& c, size_t elem_size, bool coop_rs = false, bool reg64 = false)
{
    Layout L;
    const int N = c.N_hor;
    L.off_rs = 18;
    L.off_rv = L.off_rs + 3 * N;
    L.off_c0 = L.off_rv + N;
    L.off_c = L.off_c0 + 3 * c.Nother;
    L.off_os = L.off_c + 3 * N * c.Nother;
    L.off_od = L.off_os + 12 * c.Nstcobs;
    L.off_qstc = L.off_od + 6 * (N + 1) * c.Ndynobs;
    L.off_qdyn = L.off_qstc + N;
    L.np = L.off_qdyn + N;
    const int cap = c.max_active_dynobs > 0 && c.max_active_dynobs < c.Ndynobs ? c.max_active_dynobs : c.Ndynobs;
    // fp32, three lanes per horizon step: the table entries of t >= 1 live in the registers of the one lane that reads
    // them (nmpc_device.h, RS > 0) when the provisioned rows fit 3 x 4 or 3 x 14
    L.rs = 0;
    if (elem_size == 4 && c.reg_table >= 0 && N <= 21 && 64 / N >= 3 && cap > 0) {
        if (cap <= 3 * kRegSlotsSmall) L.rs = kRegSlotsSmall;
        else if (cap <= 3 * kRegSlotsMid && NMPC_MID_SLOTS) L.rs = kRegSlotsMid;
        else if (cap <= 3 * kRegSlotsLarge) L.rs = kRegSlotsLarge;
    }
    if (elem_size == 8 && reg64 && c.reg_table >= 0 && N <= 21 && 64 / N >= 3 && cap > 3 * kRegSlotsSmall && cap <= 3 * kRegSlotsLarge)
        L.rs = kRegSlotsLarge;
    int left_ne = 0; // entries of the LDS table of the rows beyond the register-resident ones (cooperative register kernel)
    if (coop_rs) {
        L.rs = 0;
        if (elem_size == 4 && c.reg_table >= 0 && 64 / N == 1 && cap > 0) {
            L.rs = kRegSlotsCoop;
            left_ne = std::max(0, cap - coop_reg_rows(N)) * (N + 1);
        }
    }
    // table entries provisioned in LDS / the workspace. Register table: the t = 0 rows + the dummy row(s) -- three lanes per
    // step: every row a pass can address (3 x slots), so that the passes read at fixed offsets without a clamp
    int ne = L.rs ? std::max(cap + 1, coop_rs ? 0 : 3 * L.rs) : cap * (N + 1);
    // three lanes per step, register table: the table of groups of rows with identical t = 0 snapshots (nmpc_device.h,
    // load()) takes the place of the cooperative kernel's left-over table
    if (L.rs && !coop_rs) left_ne = 3 * L.rs;
    L.dyn_cap = cap;
    L.glb = false;
    L.ws_stride = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
    L.lds_alpha = L.glb ? 0 : nmpc::kEllStride * ne;
    L.lds_left = L.lds_alpha + (L.glb ? 0 : round4(ne));
    L.lds_left_alpha = L.lds_left + nmpc::kEllStride * left_ne;
    L.lds_poly = L.lds_left_alpha + round4(left_ne);
    L.lds_seg = L.lds_poly + 12 * (c.Nstcobs + 3); // (+3: dummy polygons behind the stored ones, nmpc_device.h load())
    // path segments: N real ones + far-away dummies up to 2 N + 4, so that every lane can run the same number of loop trips
    // over `first segment + 3 j` without a bound (nmpc_device.h, eval(): the segment loop)
    const int nseg = nmpc::seg_table_len(N);
    L.lds_seginv = L.lds_seg + 4 * nseg;
    L.lds_fl0 = L.lds_seginv + round4(nseg);
    L.lds_fl = L.lds_fl0 + round4(c.Nother);          // int list: robots with a non-zero t=0 position
    L.lds_iflag = L.lds_fl + round4(c.Nother);         // int list: robots with a non-zero predicted position
    L.lds_hist = L.lds_iflag + round4(c.Ndynobs);  // int flags / compaction map (an int fits in a T)
    L.lds_rho = L.lds_hist + 4 * nmpc::kMem * nmpc::lbfgs_slot_stride(N); // L-BFGS ring: kMem slots x (N | 1) x (s_v, s_w, y_v, y_w)
    L.lds_park = L.lds_rho + round4(2 * nmpc::kMem);   // rho[kMem], alpha[kMem]; then the parking area(s) (16-B aligned)
    const int park_one = nmpc::kParkQuads * 4 * 64;    // elements per wavefront
    L.lds_deepsc = L.lds_park + park_one;              // scalar block of a deep park (tail hand-off), throughput kernels only
    L.lds_total = L.lds_deepsc + nmpc::kDeepScalars;
    // latency kernel: the exchange area of W wavefronts (nmpc_spec.h) in place of the parking area (its solver vectors stay
    // in registers): W result rows of 64 x 2 gradient entries + psi (padded to 132), the master's command area of
    // 2 x 64 x W + 4 scalars
    L.lds_xch = L.lds_park;
    L.lds_total_spec = L.lds_xch + spec_xch_elems(kSpecWavesMax);
    const int cw = coop_rs ? kCoopRegWaves : kSpecWaves;
    L.lds_xch_coop = L.lds_park + 2 * park_one; // cooperative kernels: two shared parking areas, used alternately
    // exchange area: 2 buffers x cw wavefronts x 2 planes x coop_lanes Quads. Global-table kernels with one lane per step keep
    // only the lanes that carry a step (nmpc_device.h) and put the t = 0 rows of the compressed table behind it
    L.coop_lanes = (L.glb && 64 / N == 1) ? std::min(64, round4(N)) : 64;
    L.lds_t0c = L.lds_xch_coop + 2 * cw * 2 * 4 * L.coop_lanes;
    L.lds_total_coop = L.lds_t0c + (L.glb ? round4((nmpc::kEllStride + 1) * cap) : 0);
    if (coop_rs) { // no global fallback for this variant: it either fits LDS or is not offered
        if ((size_t)L.lds_total_coop * elem_size > kLdsLimit) L.rs = 0;
        break;
    }
    if (L.glb || (size_t)L.lds_total * elem_size <= kLdsLimit) break;
    L.glb = true; // second attempt: everything but the ellipse table in LDS
    L.rs = 0;     // (the GLB kernels index the full [row][t] table: the register-table layout does not apply)
    left_ne = 0;
    ne = cap * (N + 1);
    // (room for the general table, 9 values per entry, and for the compressed one: 5 per entry + the expanded t = 0 rows)
    L.ws_stride = (long long)(nmpc::kEllStride + 1) * ne; // (the compressed table -- 5 values per entry -- uses a prefix of it)
    }
    L.table_entries = ne;
    return L;
}

} // namespace

struct nmpc_handle_s {
    nmpc_config cfg;
    Layout lay32, lay64;
    Layout lay32c; // cooperative register-table kernel (fp32, one lane per step); rs = 0 if not available
    Layout lay64r; // fp64 register-table kernel (three lanes per step, one wavefront per SIMD); rs = 0 if not available
    bool use64r_auto = false; // ... chosen automatically for large batches (the LDS table allows < 4 instances per CU)
    int lps;
    int n_simd = 0; // SIMDs of the device (4 per CU): latency_waves = 0 picks the wavefront count from B / n_simd
    bool spec_ok[2] = {true, true}; // [f32, f64]
    bool coop_ok[2] = {true, true};
    int last_mode = 0;              // kernel of the last solve: 0 throughput, 1 latency (speculative), 2 cooperative
    int last_axis = -1;             // axis-aligned variant: -1 not applicable, 0 general only, 1 AXIS only, 2 decided on the device
    int last_staged = 0;            // pilot outer iterations of the last solve (0 = one launch)
    int last_polish_selected = 0, last_polish_converged = -1;
    template <typename T>
    const Layout& lay() const
    {
        return sizeof(T) == 4 ? lay32 : lay64;
    }
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int last_tail = 0;              // tail hand-off of the last solve: the parking threshold it ran with (0 = none)
    bool timed = false;
    int ptr_mode = NMPC_PTR_DETECT;
    DevBuf dP, dU, dcost, dstatus, diters, du0, dy, dc0, dinfo, dY2, dC2, dpsi, dgrad, df2, dws;
    DevBuf dorder;   // dispatch order of the next solves (nmpc_set_dispatch_order), order_B entries; 0 = none
    int order_B = 0;
    DevBuf dflag;    // [0]: epoch of the last call whose batch had an ellipse with angle != 0 (KParams::axis_flag)
    int axis_epoch = 0;
    DevBuf dresume, dorder2, dhist; // resumable solve: parked states, ranked order of the second launch, bucket counters
    DevBuf ddeep;                   // tail hand-off: solver states parked inside an inner solve (KParams::deep)
    DevBuf dproxy;                  // dispatch order from one evaluation (run_solve): nominal controls, zeros, penalties, psi, ||F2||^2
    // polish: compact fp64 copies of the selected instances and their results
    DevBuf psel, pP, pU0, pY, pC, pU, pcost, pstatus, piters, pinfo;
    std::vector<int32_t> host_status, host_sel;
};

namespace {

// second launch-bound argument = minimum waves per SIMD; it caps the register allocation (512 / waves)
#ifndef NMPC_WPE_F32
#define NMPC_WPE_F32 3 // throughput kernel, table in LDS / global memory (133 VGPRs; one spill short of fitting 128)
#endif
#ifndef NMPC_WPE_F64
#define NMPC_WPE_F64 2
#endif
// waves per SIMD a kernel variant is compiled for: the large register table needs the 256-register budget
template <typename T, int RS>
constexpr int wpe(int f32_default)
{
    return sizeof(T) == 8 ? (RS > 0 ? 1 : NMPC_WPE_F64) : RS >= kRegSlotsLarge ? 2 : RS > 0 ? 3 : f32_default;
}
// (register-table kernels exist per code path -- general / axis-aligned -- see KParams::axis_mode)
template <typename T, int LPS, int RS>
constexpr bool kHasAxisVariant = LPS == 3 && RS > 0;

// ONLY: 1 / 2 = the axis-aligned / the general path alone, returning at once when the launch takes the other one -- how every
// register-table kernel is built and launched since round 4 (pick_solve below). (0 = both paths inlined into one kernel,
// chosen per workgroup: round 3's fp32 form, no longer instantiated. In fp64 that form never worked: the 512-register
// kernel computed garbage on the general path, nondeterministically, while each path compiled alone is right -- it was
// 142 KB of code, beyond the +-128 KB reach of s_cbranch, its far branches relaxed into s_getpc / s_setpc sequences;
// tests/test_kernel_resources_cpu.py keeps every kernel below that size.)
template <typename T, int LPS, bool GLB, int RS = 0, int ONLY = 0>
__global__ __launch_bounds__(64, (wpe<T, RS>(NMPC_WPE_F32))) void solve_kernel(nmpc::KParams<T> kp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (kHasAxisVariant<T, LPS, RS> && !GLB) {
        const bool axis = nmpc::axis_path(kp);
        if (ONLY != 0 && axis != (ONLY == 1)) return; // (a pair: the twin of the path this launch takes returns at once)
        const int inst = nmpc::dispatch_index(kp);
        if (nmpc::finished_in_pilot<T>(inst)) return;
        if (ONLY != 2 && axis) {
            nmpc::solve_instance<T, LPS, GLB, RS, false, false, true>(kp, inst, reinterpret_cast<T*>(smem));
            return;
        }
        if constexpr (ONLY != 1) nmpc::solve_instance<T, LPS, GLB, RS>(kp, inst, reinterpret_cast<T*>(smem));
    } else {
        const int inst = nmpc::dispatch_index(kp);
        if (nmpc::finished_in_pilot<T>(inst)) return;
        nmpc::solve_instance<T, LPS, GLB, RS>(kp, inst, reinterpret_cast<T*>(smem));
    }
}

// diagnostic: the one-wavefront fp64 solver (LDS / global table) writing one record per inner iteration (KParams::trace)
template <int LPS, bool GLB>
__global__ __launch_bounds__(64, NMPC_WPE_F64) void trace_kernel(nmpc::KParams<double> kp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    nmpc::solve_instance<double, LPS, GLB, 0, false, false, false, true>(kp, (int)blockIdx.x, reinterpret_cast<double*>(smem));
}

// cooperative mode: up to kSpecWaves wavefronts per instance share every evaluation (nmpc_device.h, COOP)
// GLB: a PAIR like the register-table kernels -- ONLY = 1: the compressed table of axis-aligned ellipses (5 instead of 9
// values streamed per entry, nmpc_device.h Instance::CMP), ONLY = 2: the general table; the member the launch does not take
// returns at once (KParams::axis_mode). ONLY = 0: no variants (table in LDS).
template <typename T, int LPS, bool GLB, int ONLY = 0>
__global__ __launch_bounds__(64 * kSpecWaves, (sizeof(T) == 4 ? NMPC_SPEC_WPE_F32 : NMPC_WPE_F64)) void solve_coop_kernel(nmpc::KParams<T> kp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (ONLY != 0) {
        static_assert(GLB, "the compressed table is a variant of the global one");
        if (nmpc::axis_path(kp) != (ONLY == 1)) return;
    }
    const int inst = nmpc::dispatch_index(kp);
    if (nmpc::finished_in_pilot<T>(inst)) return;
    nmpc::solve_instance<T, LPS, GLB, 0, true, false, ONLY == 1>(kp, inst, reinterpret_cast<T*>(smem));
}
// ... with the obstacle table on chip instead of in global memory, for one lane per horizon step (N > 32), where it does
// not fit LDS: EIGHT wavefronts (two per SIMD) keep 12 rows each in registers, the remaining rows live in LDS
template <bool HLP>
__global__ __launch_bounds__(64 * kCoopRegWaves, 2) void solve_coop_reg_kernel(nmpc::KParams<float> kp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int inst = nmpc::dispatch_index(kp);
    if (nmpc::finished_in_pilot<float>(inst)) return;
    nmpc::solve_instance<float, 1, false, kRegSlotsCoop, true, HLP>(kp, inst, reinterpret_cast<float*>(smem));
}

// latency mode: kSpecWaves wavefronts per instance, speculative line search (nmpc_spec.h)
// FLAT = false: the TAIL member (nmpc_spec.h) -- this kernel with the throughput kernels' evaluation, bit-identical to them
template <typename T, int LPS, bool GLB, int RS = 0, int ONLY = 0, bool FLAT = true>
__global__ __launch_bounds__(64 * kSpecWavesMax, (wpe<T, RS>(NMPC_SPEC_WPE_F32))) void solve_spec_kernel(nmpc::KParams<T> kp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (kHasAxisVariant<T, LPS, RS> && !GLB) {
        const bool axis = nmpc::axis_path(kp);
        if (ONLY != 0 && axis != (ONLY == 1)) return; // (a pair: the twin of the path this launch takes returns at once)
        const int inst = nmpc::dispatch_index(kp);
        if (nmpc::finished_in_pilot<T>(inst)) return;
        if (ONLY != 2 && axis) {
            nmpc::solve_instance_spec<T, LPS, GLB, RS, true, FLAT>(kp, inst, reinterpret_cast<T*>(smem));
            return;
        }
        if constexpr (ONLY != 1) nmpc::solve_instance_spec<T, LPS, GLB, RS, false, FLAT>(kp, inst, reinterpret_cast<T*>(smem));
    } else {
        const int inst = nmpc::dispatch_index(kp);
        if (nmpc::finished_in_pilot<T>(inst)) return;
        nmpc::solve_instance_spec<T, LPS, GLB, RS, false, FLAT>(kp, inst, reinterpret_cast<T*>(smem));
    }
}

template <typename T, int LPS, bool GLB, int RS, bool AXIS>
__device__ __forceinline__ void eval_instance(const nmpc::KParams<T>& kp, const nmpc::EvalParams<T>& ep, T* lds)
{
    const int inst = blockIdx.x, N = kp.N;
    nmpc::Instance<T, LPS, GLB, RS, false, false, AXIS> I(kp, kp.P + (size_t)inst * kp.np, lds,
                                                        GLB ? kp.ws + (long long)inst * kp.ws_stride : nullptr);
    if (I.load()) {
        if (I.lane == 0) ep.psi[inst] = __builtin_nanf("");
        return;
    }
    const int kk = I.act ? I.k : 0;
    T v = ep.U[(size_t)inst * 2 * N + 2 * kk], w = ep.U[(size_t)inst * 2 * N + 2 * kk + 1];
    T yv = ep.Y[(size_t)inst * 2 * N + kk], yw = ep.Y[(size_t)inst * 2 * N + N + kk];
    if (!I.act) v = w = yv = yw = 0;
    const T c = ep.C[inst];
    const T icd = T(1) / (c > T(1) ? c : T(1));
    T psi, f2, gv, gw;
    if (ep.grad)
        I.template eval<true>(v, w, c, icd, yv, yw, psi, f2, gv, gw);
    else
        I.template eval<false>(v, w, c, icd, yv, yw, psi, f2, gv, gw);
    if (I.lead && ep.grad) {
        ep.grad[(size_t)inst * 2 * N + 2 * I.k] = gv;
        ep.grad[(size_t)inst * 2 * N + 2 * I.k + 1] = gw;
    }
    if (I.lane == 0) {
        ep.psi[inst] = psi;
        if (ep.f2sq) ep.f2sq[inst] = f2;
    }
}
template <typename T, int LPS, bool GLB, int RS = 0, int ONLY = 0>
__global__ __launch_bounds__(64, 1) void eval_kernel(nmpc::KParams<T> kp, nmpc::EvalParams<T> ep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (kHasAxisVariant<T, LPS, RS> && !GLB) {
        const bool axis = nmpc::axis_path(kp);
        if (ONLY != 0 && axis != (ONLY == 1)) return;
        if (ONLY != 2 && axis) {
            eval_instance<T, LPS, GLB, RS, true>(kp, ep, reinterpret_cast<T*>(smem));
            return;
        }
    }
    if constexpr (ONLY != 1) eval_instance<T, LPS, GLB, RS, false>(kp, ep, reinterpret_cast<T*>(smem));
}

// ---- small data kernels around the solves ------------------------------------------------------------------------
// Is some ellipse of the batch not axis-aligned (angle != 0)? One workgroup per instance over o_d[j][t][4]; any
// workgroup that finds a non-zero angle stores the call's epoch in *flag (plain store: all writers agree).
template <typename T>
__global__ __launch_bounds__(256) void axis_scan_kernel(const T* P, int np, int off_od, int n_entries, int* flag, int epoch)
{
    const T* s = P + (size_t)blockIdx.x * np + off_od;
    bool skew = false;
    for (int e = threadIdx.x; e < n_entries; e += 256) skew = skew || s[6 * e + 4] != T(0);
    if (skew) *flag = epoch; // (this call's epoch: the flag needs no reset between calls)
}

// Resumable solve, ranking of the next launch: counting sort, descending, of a key the parked state holds -- key 0: the
// hard-constraint violation ||F2|| (bucket = the top 10 bits of the float: monotonic for values >= 0), key 1: the psi
// evaluations used so far (64 per bucket). Finished instances -- and only they -- go to bucket 0.
constexpr int kRankBuckets = 1024;
template <typename T>
__device__ __forceinline__ int rank_bucket(const T* resume, const int* status, int b, int key)
{
    if (status[b] != -1) return 0;
    const T* sc = resume + (size_t)b * nmpc::kResumeStride + 6 * 64;
    unsigned bits;
    if (key == 0) {
        const float f = (float)sc[3];
        bits = __float_as_uint(f >= 0.0f ? f : 0.0f) >> 21;
    } else {
        bits = (unsigned)((int)sc[7]) >> 6;
    }
    // (an unfinished instance never shares bucket 0 with the finished ones -- ||F2|| = 0 is common: no obstacle in reach --:
    //  the launch that follows counts on the unfinished instances standing FIRST in its order, and on their number, offs[0])
    return (int)(bits < 1u ? 1u : bits < (unsigned)kRankBuckets ? bits : (unsigned)kRankBuckets - 1u);
}
template <typename T>
__global__ __launch_bounds__(256) void rank_hist_kernel(const T* resume, const int* status, int B, int* hist, int key)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int q = b < B ? rank_bucket(resume, status, b, key) : -1;
    // Bucket 0 -- the instances that finished in the pilot, half of a `passing`-type batch -- is counted once per wavefront:
    // 34 k atomics on one address took 0.41 ms here and again in the scatter (profiles/r05_cfg2_passing_kernel_timeline.txt).
    const unsigned long long m0 = __ballot(q == 0);
    if (q > 0) atomicAdd(&hist[q], 1);
    else if (q == 0 && (int)(threadIdx.x & 63) == __ffsll((long long)m0) - 1) atomicAdd(&hist[0], __popcll(m0));
}
// (tail hand-off under a caller's dispatch order, no ranking before the launch: all B instances are to be solved)
__global__ void dyn_init_kernel(int* dyn_ctr, int total) { dyn_ctr[0] = 0, dyn_ctr[1] = 0, dyn_ctr[2] = total, dyn_ctr[3] = 0; }
// offs[q] = number of instances in buckets above q (one workgroup of kRankBuckets threads; reversed inclusive scan)
// (the counters are left zeroed for the next ranking: no memset between launches)
__global__ __launch_bounds__(kRankBuckets) void rank_scan_kernel(int* hist, int* offs, int* dyn_ctr)
{
    __shared__ int sh[kRankBuckets];
    const int t = threadIdx.x;                 // t = 0 is the top bucket
    const int mine = hist[kRankBuckets - 1 - t];
    hist[kRankBuckets - 1 - t] = 0;
    sh[t] = mine;
    __syncthreads();
    for (int d = 1; d < kRankBuckets; d <<= 1) {
        const int add = t >= d ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    offs[kRankBuckets - 1 - t] = sh[t] - mine;
    // (tail hand-off, KParams::dyn_ctr: the launch that follows has offs[0] unfinished instances to solve -- everything above bucket 0)
    if (dyn_ctr && t == kRankBuckets - 1) dyn_ctr[0] = 0, dyn_ctr[1] = 0, dyn_ctr[2] = sh[t] - mine, dyn_ctr[3] = 0;
}
template <typename T>
__global__ __launch_bounds__(256) void rank_scatter_kernel(const T* resume, const int* status, int B, int* offs, int* order, int key)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    const int q = b < B ? rank_bucket(resume, status, b, key) : -1;
    const unsigned long long m0 = __ballot(q == 0);
    if (q > 0) {
        order[atomicAdd(&offs[q], 1)] = b;
    } else if (q == 0) { // one atomic per wavefront for the finished instances (see rank_hist_kernel)
        const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)m0) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&offs[0], __popcll(m0));
        base = __shfl(base, leader);
        order[base + __popcll(m0 & ((1ull << lane) - 1ull))] = b;
    }
}

// Dispatch order of a latency-plan batch from ONE evaluation instead of a pilot launch (run_solve): the nominal controls
// (v_nom, 0) at every step, zero multipliers, the initial penalty; and the key -- ||F2||^2 there -- written where rank_bucket
// reads it (every instance counts as unfinished: nothing has been solved yet).
template <typename T>
__global__ __launch_bounds__(256) void proxy_fill_kernel(T* U, T* Y, T* C, int B, int n, T v_nom, T c_init)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)B * n) {
        U[i] = (i & 1) ? T(0) : v_nom;
        Y[i] = T(0);
    }
    if (i < (size_t)B) C[i] = c_init;
}
template <typename T>
__global__ __launch_bounds__(256) void proxy_key_kernel(const T* f2sq, T* resume, int* status, int B)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) {
        status[b] = -1;
        resume[(size_t)b * nmpc::kResumeStride + 6 * 64 + 3] = f2sq[b];
    }
}

// Polish: fp64 copies of the selected instances' parameters and of the main solve's (u, y, c) / results back into the
// caller's arrays where the continuation converged (nmpc_config.polish)
template <typename T>
__global__ __launch_bounds__(256) void polish_gather_kernel(const T* P, const T* U, const T* y, const T* info, const int* sel,
                                                            int np, int n2, double* P64, double* U64, double* Y64, double* C64)
{
    const int i = blockIdx.x, b = sel[i];
    const T* src = P + (size_t)b * np;
    double* dst = P64 + (size_t)i * np;
    for (int j = threadIdx.x; j < np; j += 256) dst[j] = (double)src[j];
    if ((int)threadIdx.x < n2) {
        U64[(size_t)i * n2 + threadIdx.x] = (double)U[(size_t)b * n2 + threadIdx.x];
        Y64[(size_t)i * n2 + threadIdx.x] = (double)y[(size_t)b * n2 + threadIdx.x];
    }
    if (threadIdx.x == 0) C64[i] = (double)info[(size_t)b * 8 + 3];
}
// (info[:, 6] is the polish outcome when the polish is on: 0 for the instances it does not touch)
template <typename T>
__global__ __launch_bounds__(256) void polish_clear_flag_kernel(T* info, int B)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) info[(size_t)b * 8 + 6] = T(0);
}
template <typename T>
__global__ __launch_bounds__(128) void polish_scatter_kernel(const int* sel, const double* U64, const double* Y64, const double* cost64,
                                                            const int* status64, const int* iters64, const double* info64, int n2,
                                                            T* U, T* y, bool y_user, T* cost, int* iters, T* info, bool info_user)
{
    const int i = blockIdx.x, b = sel[i], t = threadIdx.x;
    const bool ok = status64[i] == 0;
    if (ok && t < n2) {
        U[(size_t)b * n2 + t] = (T)U64[(size_t)i * n2 + t];
        if (y_user) y[(size_t)b * n2 + t] = (T)Y64[(size_t)i * n2 + t];
    }
    if (t == 0) {
        if (ok && cost) cost[b] = (T)cost64[i];
        if (iters) {
            iters[2 * b] += iters64[2 * i];
            iters[2 * b + 1] += iters64[2 * i + 1];
        }
        if (info_user) {
            T* o = info + (size_t)b * 8;
            const double* q = info64 + (size_t)i * 8;
            if (ok) o[0] = (T)q[0], o[1] = (T)q[1], o[2] = (T)q[2], o[3] = (T)q[3];
            o[4] += (T)q[4];
            o[5] += (T)q[5];
            o[6] = ok ? T(1) : T(2);
        }
    }
}

// wave-primitive self test: integer-valued data so that every summation order gives the same float
template <typename T>
__global__ __launch_bounds__(64) void selftest_kernel(int* fails)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    for (int round = 0; round < 8; ++round) {
        const T x = T(((lane * 37 + round * 11) % 23) - 9);
        if (nmpc::wave_sum(x) != nmpc::ref_wave_sum(x)) bad |= 1;
        if (nmpc::wave_scan_incl(x) != nmpc::ref_wave_scan_incl(x)) bad |= 2;
        if (nmpc::wave_scan_suffix_incl(x) != nmpc::ref_wave_scan_suffix_incl(x)) bad |= 4;
        const T up3 = nmpc::wave_shift_up<3>(x), up3r = __shfl_up(x, 3, 64);
        if (up3 != (lane >= 3 ? up3r : T(0))) bad |= 8;
        const T dn2 = nmpc::wave_shift_down<2>(x), dn2r = __shfl_down(x, 2, 64);
        if (dn2 != (lane < 62 ? dn2r : T(0))) bad |= 16;
        if (nmpc::wave_reverse(x) != __shfl(x, 63 - lane, 64)) bad |= 32;
        {
            T a = x, b = T(3) - x;
            nmpc::wave_scan_incl2(a, b);
            if (a != nmpc::ref_wave_scan_incl(x) || b != nmpc::ref_wave_scan_incl(T(3) - x)) bad |= 128;
            T c = x, d = T(2) * x;
            nmpc::wave_scan_suffix_incl2(c, d);
            if (c != nmpc::ref_wave_scan_suffix_incl(x) || d != nmpc::ref_wave_scan_suffix_incl(T(2) * x)) bad |= 256;
            T s1, s2, s3;
            nmpc::wave_sum2(x, T(1) + x, s1, s2);
            if (s1 != nmpc::ref_wave_sum(x) || s2 != nmpc::ref_wave_sum(T(1) + x)) bad |= 512;
            nmpc::wave_sum3(x, T(2) - x, T(5) * x, s1, s2, s3);
            if (s1 != nmpc::ref_wave_sum(x) || s2 != nmpc::ref_wave_sum(T(2) - x) || s3 != nmpc::ref_wave_sum(T(5) * x)) bad |= 1024;
        }
        if (nmpc::read_lane(x, 17) != __shfl(x, 17, 64)) bad |= 64;
        {
            int a[4];
            nmpc::class3_addresses(lane % 3, a);
            T c1 = x, c2 = T(4) - x;
            nmpc::class3_sum2(c1, c2, a);
            if (c1 != nmpc::ref_class3_sum(x) || c2 != nmpc::ref_class3_sum(T(4) - x)) bad |= 2048;
        }
    }
    if (sizeof(T) == 4) { // sincos_medium against the library on its whole range, sign and quadrant boundaries included
        for (int round = 0; round < 64; ++round) {
            const int i = round * 64 + lane;
            float x = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f);            // [0, 1)
            x = (i & 3) == 0 ? x * 8.0f - 4.0f : (i & 3) == 1 ? (x - 0.5f) * 200.0f : (i & 3) == 2 ? (x - 0.5f) * 2.6e5f
                                                                                                : 0.78539816f * (float)((i >> 2) % 17 - 8) + (x - 0.5f) * 1e-4f;
            float s1, c1, s2, c2;
            sincosf(x, &s1, &c1);
            nmpc::sincos_medium(x, s2, c2);
            if (__builtin_bit_cast(unsigned, s1) != __builtin_bit_cast(unsigned, s2) ||
                __builtin_bit_cast(unsigned, c1) != __builtin_bit_cast(unsigned, c2)) bad |= 4096;
        }
    }
    if (bad) atomicOr(fails, bad);
}

// The layout-dependent fields only (parameter offsets, LDS map): what a kernel choice that brings its own layout re-fills,
// leaving the batch buffers and -- ADVICE r3 -- the solver options of the caller (the polish's tolerances and caps) alone.
template <typename T>
void fill_layout(nmpc::KParams<T>& k, const Layout& L)
{
    k.np = L.np;
    k.dyn_cap = L.dyn_cap;
    k.off_rs = L.off_rs;
    k.off_rv = L.off_rv;
    k.off_c0 = L.off_c0;
    k.off_c = L.off_c;
    k.off_os = L.off_os;
    k.off_od = L.off_od;
    k.off_qstc = L.off_qstc;
    k.off_qdyn = L.off_qdyn;
    k.lds_alpha = L.lds_alpha;
    k.lds_poly = L.lds_poly;
    k.lds_seg = L.lds_seg;
    k.lds_seginv = L.lds_seginv;
    k.lds_fl0 = L.lds_fl0;
    k.lds_fl = L.lds_fl;
    k.lds_iflag = L.lds_iflag;
    k.lds_hist = L.lds_hist;
    k.lds_rho = L.lds_rho;
    k.lds_total = L.lds_total;
    k.lds_xch = L.lds_xch;
    k.lds_park = L.lds_park;
    k.lds_deepsc = L.lds_deepsc;
    k.coop_lanes = L.coop_lanes;
    k.lds_t0c = L.lds_t0c;
    k.lds_left = L.lds_left;
    k.lds_left_alpha = L.lds_left_alpha;
}

template <typename T>
void fill_kparams(const nmpc_handle_s* h, nmpc::KParams<T>& k, const Layout* layout = nullptr)
{
    const nmpc_config& c = h->cfg;
    std::memset(&k, 0, sizeof k);
    fill_layout(k, layout ? *layout : h->lay<T>());
    k.N = c.N_hor;
    k.Nother = c.Nother;
    k.Nstc = c.Nstcobs;
    k.Ndyn = c.Ndynobs;
    k.ts = (T)c.ts;
    k.inv_ts = (T)(1.0 / c.ts);
    k.vmin = (T)c.lin_vel_min;
    k.vmax = (T)c.lin_vel_max;
    k.wmax = (T)c.ang_vel_max;
    k.amin = (T)c.lin_acc_min;
    k.amax = (T)c.lin_acc_max;
    k.wamax = (T)c.ang_acc_max;
    k.safe2 = (T)(c.vehicle_width * c.vehicle_width);
    k.vm = (T)c.vehicle_margin;
    k.sm = (T)c.social_margin;
    k.tol = (T)c.tolerance;
    k.init_tol = (T)c.initial_tolerance;
    k.delta_tol = (T)c.delta_tolerance;
    k.c_init = (T)c.initial_penalty;
    k.pen_update = (T)c.penalty_update_factor;
    k.tol_update = (T)c.inner_tolerance_update_factor;
    k.suff_dec = (T)c.sufficient_decrease_coeff;
    const bool f32 = sizeof(T) == 4;
    k.lip_eps = (T)(f32 ? c.lip_eps_f32 : c.lip_eps_f64);
    k.lip_delta = (T)(f32 ? c.lip_delta_f32 : c.lip_delta_f64);
    k.cbfgs_alpha = (T)c.cbfgs_alpha;
    k.cbfgs_eps = (T)c.cbfgs_epsilon;
    k.sy_eps = (T)c.sy_epsilon;
    k.max_outer = c.max_outer_iterations;
    k.max_inner = c.max_inner_iterations;
    k.mem = c.lbfgs_memory;
    k.akkt_form = c.akkt_form;
    k.time_budget = c.max_solver_time_us > 0 ? (long long)(c.max_solver_time_us * 100.0 + 0.5) : 0; // 100 MHz ticks
    k.max_evals = c.max_evaluations > 0 ? c.max_evaluations : 0;
}

// The same evaluation through the cooperative kernels' code path (W wavefronts share it; wavefront 0 writes): what
// nmpc_eval_batch_* launches when the handle's coop_waves asks for the cooperative mode, so that psi / grad psi of the
// row split, the partial-sum exchange and the helper lanes can be compared with the oracle directly.
template <typename T, int LPS, bool GLB, int RS, bool HLP, int WAVES, int ONLY = 0>
__global__ __launch_bounds__(64 * WAVES, (RS > 0 ? 2 : (sizeof(T) == 4 ? NMPC_SPEC_WPE_F32 : NMPC_WPE_F64)))
void eval_coop_kernel(nmpc::KParams<T> kp, nmpc::EvalParams<T> ep)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (ONLY != 0)
        if (nmpc::axis_path(kp) != (ONLY == 1)) return; // (global table: compressed / general member of the pair)
    const int inst = blockIdx.x, N = kp.N;
    T* lds = reinterpret_cast<T*>(smem);
    nmpc::Instance<T, LPS, GLB, RS, true, HLP, ONLY == 1> I(kp, kp.P + (size_t)inst * kp.np, lds,
                                                 GLB ? kp.ws + (long long)inst * kp.ws_stride : nullptr);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    I.cw_ = wave;
    I.CW_ = (int)(blockDim.x >> 6);
    I.coop_x = lds + kp.lds_xch;
    if (I.load()) {
        if (threadIdx.x == 0) ep.psi[inst] = __builtin_nanf("");
        return;
    }
    const int kk = I.act ? I.k : 0;
    T v = ep.U[(size_t)inst * 2 * N + 2 * kk], w = ep.U[(size_t)inst * 2 * N + 2 * kk + 1];
    T yv = ep.Y[(size_t)inst * 2 * N + kk], yw = ep.Y[(size_t)inst * 2 * N + N + kk];
    if (!I.act) v = w = yv = yw = 0;
    const T c = ep.C[inst];
    const T icd = T(1) / (c > T(1) ? c : T(1));
    T psi, f2, gv, gw;
    if (ep.grad)
        I.template eval<true>(v, w, c, icd, yv, yw, psi, f2, gv, gw);
    else
        I.template eval<false>(v, w, c, icd, yv, yw, psi, f2, gv, gw);
    if (wave != 0) return;
    if (I.lead && ep.grad) {
        ep.grad[(size_t)inst * 2 * N + 2 * I.k] = gv;
        ep.grad[(size_t)inst * 2 * N + 2 * I.k + 1] = gw;
    }
    if (I.lane == 0) {
        ep.psi[inst] = psi;
        if (ep.f2sq) ep.f2sq[inst] = f2;
    }
}

template <typename T>
using SolveFn = void (*)(nmpc::KParams<T>);
template <typename T>
using EvalFn = void (*)(nmpc::KParams<T>, nmpc::EvalParams<T>);

// the register-table variants exist for float with three lanes per step only
void (*pick_solve_coop_reg(int N))(nmpc::KParams<float>)
{
    return coop_helper_lanes(N) ? solve_coop_reg_kernel<true> : solve_coop_reg_kernel<false>;
}

// (the register-table variants exist for float with three lanes per step only)
// The register-table kernels come as PAIRS since round 4 -- <.., 1> = the axis-aligned path alone, <.., 2> = the general
// (rotated-ellipse) path alone, launched one behind the other, each workgroup of the twin that the device-side flag does
// not pick returning before it touches anything. One kernel with both paths inlined (round 3) reported its resources as
// the maximum over both -- 19-22 spilled VGPRs and 52-88 B of scratch that only the general path has -- and had grown to
// 125 KB of code, against the 128 KB reach of s_cbranch. `only` = 1 / 2 picks the member.
template <typename T>
SolveFn<T> pick_solve(int lps, bool glb, int rs = 0, int only = 1)
{
    if constexpr (sizeof(T) == 4)
    {
        if (rs == kRegSlotsSmall && lps == 3 && !glb)
            return only == 2 ? solve_kernel<T, 3, false, kRegSlotsSmall, 2> : solve_kernel<T, 3, false, kRegSlotsSmall, 1>;
        if (rs == kRegSlotsMid && lps == 3 && !glb)
            return only == 2 ? solve_kernel<T, 3, false, kRegSlotsMid, 2> : solve_kernel<T, 3, false, kRegSlotsMid, 1>;
    }
    if (rs == kRegSlotsLarge && lps == 3 && !glb)
        return only == 2 ? solve_kernel<T, 3, false, kRegSlotsLarge, 2> : solve_kernel<T, 3, false, kRegSlotsLarge, 1>;
    if (glb) return lps == 3 ? solve_kernel<T, 3, true> : lps == 2 ? solve_kernel<T, 2, true> : solve_kernel<T, 1, true>;
    return lps == 3 ? solve_kernel<T, 3, false> : lps == 2 ? solve_kernel<T, 2, false> : solve_kernel<T, 1, false>;
}
template <typename T>
SolveFn<T> pick_solve_spec(int lps, bool glb, int rs = 0, int only = 1)
{
    if constexpr (sizeof(T) == 4) {
        if (rs == kRegSlotsSmall && lps == 3 && !glb)
            return only == 2 ? solve_spec_kernel<T, 3, false, kRegSlotsSmall, 2> : solve_spec_kernel<T, 3, false, kRegSlotsSmall, 1>;
        if (rs == kRegSlotsMid && lps == 3 && !glb)
            return only == 2 ? solve_spec_kernel<T, 3, false, kRegSlotsMid, 2> : solve_spec_kernel<T, 3, false, kRegSlotsMid, 1>;
        if (rs == kRegSlotsLarge && lps == 3 && !glb)
            return only == 2 ? solve_spec_kernel<T, 3, false, kRegSlotsLarge, 2> : solve_spec_kernel<T, 3, false, kRegSlotsLarge, 1>;
    }
    if (glb) return lps == 3 ? solve_spec_kernel<T, 3, true> : lps == 2 ? solve_spec_kernel<T, 2, true> : solve_spec_kernel<T, 1, true>;
    return lps == 3 ? solve_spec_kernel<T, 3, false> : lps == 2 ? solve_spec_kernel<T, 2, false> : solve_spec_kernel<T, 1, false>;
}

// The tail member of the fp32 register-table kernels (nmpc_config.tail_latency): the latency kernel with the throughput
// kernels' evaluation; nullptr where there is none.
template <typename T>
SolveFn<T> pick_solve_tail(int lps, bool glb, int rs, int only)
{
    if constexpr (sizeof(T) == 4) {
        if (lps != 3 || glb) return nullptr;
        if (rs == kRegSlotsSmall)
            return only == 2 ? solve_spec_kernel<T, 3, false, kRegSlotsSmall, 2, false> : solve_spec_kernel<T, 3, false, kRegSlotsSmall, 1, false>;
        if (rs == kRegSlotsMid)
            return only == 2 ? solve_spec_kernel<T, 3, false, kRegSlotsMid, 2, false> : solve_spec_kernel<T, 3, false, kRegSlotsMid, 1, false>;
        if (rs == kRegSlotsLarge)
            return only == 2 ? solve_spec_kernel<T, 3, false, kRegSlotsLarge, 2, false> : solve_spec_kernel<T, 3, false, kRegSlotsLarge, 1, false>;
    }
    return nullptr;
}

// (global table: `only` = 1 the compressed-table member of the pair, 2 the general one)
template <typename T>
SolveFn<T> pick_solve_coop(int lps, bool glb, int only = 2)
{
    if (glb && only == 1) return lps == 3 ? solve_coop_kernel<T, 3, true, 1> : lps == 2 ? solve_coop_kernel<T, 2, true, 1> : solve_coop_kernel<T, 1, true, 1>;
    if (glb) return lps == 3 ? solve_coop_kernel<T, 3, true, 2> : lps == 2 ? solve_coop_kernel<T, 2, true, 2> : solve_coop_kernel<T, 1, true, 2>;
    return lps == 3 ? solve_coop_kernel<T, 3, false> : lps == 2 ? solve_coop_kernel<T, 2, false> : solve_coop_kernel<T, 1, false>;
}

template <typename T>
EvalFn<T> pick_eval(int lps, bool glb, int rs = 0, int only = 1)
{
    if constexpr (sizeof(T) == 4)
    {
        if (rs == kRegSlotsSmall && lps == 3 && !glb)
            return only == 2 ? eval_kernel<T, 3, false, kRegSlotsSmall, 2> : eval_kernel<T, 3, false, kRegSlotsSmall, 1>;
        if (rs == kRegSlotsMid && lps == 3 && !glb)
            return only == 2 ? eval_kernel<T, 3, false, kRegSlotsMid, 2> : eval_kernel<T, 3, false, kRegSlotsMid, 1>;
    }
    if (rs == kRegSlotsLarge && lps == 3 && !glb)
        return only == 2 ? eval_kernel<T, 3, false, kRegSlotsLarge, 2> : eval_kernel<T, 3, false, kRegSlotsLarge, 1>;
    if (glb) return lps == 3 ? eval_kernel<T, 3, true> : lps == 2 ? eval_kernel<T, 2, true> : eval_kernel<T, 1, true>;
    return lps == 3 ? eval_kernel<T, 3, false> : lps == 2 ? eval_kernel<T, 2, false> : eval_kernel<T, 1, false>;
}

template <typename T>
EvalFn<T> pick_eval_coop(int lps, bool glb, int only = 2)
{
    constexpr int W = kSpecWaves;
    if (glb && only == 1) return lps == 3 ? eval_coop_kernel<T, 3, true, 0, false, W, 1> : lps == 2 ? eval_coop_kernel<T, 2, true, 0, false, W, 1> : eval_coop_kernel<T, 1, true, 0, false, W, 1>;
    if (glb) return lps == 3 ? eval_coop_kernel<T, 3, true, 0, false, W, 2> : lps == 2 ? eval_coop_kernel<T, 2, true, 0, false, W, 2> : eval_coop_kernel<T, 1, true, 0, false, W, 2>;
    return lps == 3 ? eval_coop_kernel<T, 3, false, 0, false, W> : lps == 2 ? eval_coop_kernel<T, 2, false, 0, false, W> : eval_coop_kernel<T, 1, false, 0, false, W>;
}
EvalFn<float> pick_eval_coop_reg(int N)
{
    return coop_helper_lanes(N) ? eval_coop_kernel<float, 1, false, kRegSlotsCoop, true, kCoopRegWaves>
                                : eval_coop_kernel<float, 1, false, kRegSlotsCoop, false, kCoopRegWaves>;
}

// stage `count` elements: returns the device pointer to use (src itself if already on the device)
template <typename T>
int stage_in(nmpc_handle_s* h, DevBuf& buf, const T* src, size_t count, const T** out)
{
    if (!src) {
        *out = nullptr;
        return 0;
    }
    if (h->ptr_mode == NMPC_PTR_DEVICE || (h->ptr_mode == NMPC_PTR_DETECT && is_device_ptr(src))) {
        *out = src;
        return 0;
    }
    if (int rc = buf.reserve(count * sizeof(T))) return rc;
    HIP_TRY(hipMemcpyAsync(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
    *out = static_cast<const T*>(buf.p);
    return 0;
}
template <typename T>
int stage_out(nmpc_handle_s* h, DevBuf& buf, T* dst, size_t count, T** dev, bool* is_host)
{
    *is_host = false;
    if (!dst) {
        *dev = nullptr;
        return 0;
    }
    if (h->ptr_mode == NMPC_PTR_DEVICE || (h->ptr_mode == NMPC_PTR_DETECT && is_device_ptr(dst))) {
        *dev = dst;
        return 0;
    }
    if (int rc = buf.reserve(count * sizeof(T))) return rc;
    *dev = static_cast<T*>(buf.p);
    *is_host = true;
    return 0;
}

// ---- kernel choice and launch ---------------------------------------------------------------------------------------
// (development builds, -DNMPC_DEV_ENV: the batch-size thresholds of the resumable solve / the tail hand-off, in device fills, from
//  the environment -- tools/exp_mid_batches.py; the shipped library has the constants)
#ifdef NMPC_DEV_ENV
static double dev_factor(const char* name, double dflt)
{
    const char* s = getenv(name);
    return s ? atof(s) : dflt;
}
#else
static constexpr double dev_factor(const char*, double dflt) { return dflt; }
#endif
// Batch size, in device fills of the planned kernel, from which the resumable solve (pilot + ranking) and the tail hand-off are
// used. fp32 register-table kernels: ONE fill -- measured at configs[1]'s and configs[2]'s dimensions on three families
// (tools/exp_mid_batches.py, profiles/r06_exp_mid_batches.txt: batches of 1-4 fills, what a closed-loop evaluation sends once
// most scenarios have finished, -15..-35 %; until round 6 both started at four fills). Everything else: four, as measured
// in rounds 3-4 (fp64 and the LDS-table kernels have no tail member and were not re-measured).
constexpr double kStageFills = 4, kTailFills = 4, kStageFillsReg = 1, kTailFillsReg = 1;
// 14-slot kernels (2 048 resident wavefronts): already from 0.7 fills on -- B = 1 500 at configs[2]'s dimensions: `passing` 42.0 ->
// 27.2 ms, reference scenarios 31.0 -> 32.5, contract family 62.2 -> 50.5; at 1 024 the contract family still loses 20 % to two
// wavefronts per instance, and at configs[1]'s dimensions (4-slot kernels) it does so up to a full fill (same record).
constexpr double kFillsLarge = 0.7;
constexpr double kProxyFills = 8;     // throughput plans below this many fills: dispatch order from one evaluation instead of a pilot launch
constexpr double kTailMinParks = 5;   // the hand-off needs a batch of at least this many parking thresholds (8 until the above)

template <typename T>
struct Plan {
    SolveFn<T> fn = nullptr, fn2 = nullptr; // fn2: the general-path member of a register-table kernel pair (fn = the axis-aligned one)
    bool has_axis = false; // the kernel contains the axis-aligned variant (KParams::axis_mode)
    int threads = 64;
    size_t lds_bytes = 0;
    int mode = 0;          // 0 throughput, 1 latency (speculative), 2 cooperative
    bool uses_ws = false;  // the variant reads the global obstacle workspace
    bool stageable = false;
    int resident = 0;      // cooperative kernels: workgroups resident at once (0 = derived from the one-wavefront layout)
};

// Which solve kernel runs a batch of B instances (measured crossovers). May re-fill `k` with the layout of the
// on-chip cooperative kernel (the batch buffers are kept).
template <typename T>
Plan<T> plan_solve(nmpc_handle_s* h, int B, nmpc::KParams<T>& k)
{
    const Layout& L = h->lay<T>();
    Plan<T> pl;
    // wavefronts per instance: 0 = throughput kernel; > 0 = latency kernel (pays off while the batch leaves SIMDs idle)
    int lw = h->cfg.latency_waves;
    // fp64 runs 2 wavefronts per SIMD (256 VGPRs) against 3 in fp32, so fewer 4-wavefront workgroups are resident
    const int cap = sizeof(T) == 4 ? h->n_simd : h->n_simd / 2;
    if (lw == 0) {
        // one workgroup per SIMD or less: four wavefronts per instance. (Round 2 measured W = 3 -- what stays resident
        // together at 168 registers -- ahead of W = 4, 28.5 k against 25.1 k solves/s on configs[1]; since the resumable
        // solve starts the long instances first, the quarter of the workgroups that has to wait for a slot is the short
        // ones and the faster line search of the long ones wins: 41.8 k (W = 4) against 40.0 k (W = 3) and 35.4 k (W = 2),
        // `passing` 46.6 / 41.8 / 34.7 k -- profiles/r04_exp_cfg1_waves.txt. Results do not depend on W, bit for bit.)
        lw = B <= cap ? (sizeof(T) == 4 ? (L.rs >= kRegSlotsLarge ? (4 * B <= 3 * h->n_simd ? kSpecWaves : 2) : kSpecWaves) : kSpecWaves) : B <= 4 * cap ? 2 : 1;
        // (14-slot kernels, 2 048 resident wavefronts: until round 6 two wavefronts per instance at every size. Measured at
        //  configs[2]'s dimensions on three families, W = 2 / 3 / 4 / 6 -- profiles/r06_exp_mid_batches.txt: B = 64 / 256 six
        //  wavefronts -32..-35 %; B = 512 / 768 four -25..-30 % on `passing` and the reference scenarios, -31 % / +3 % on the
        //  contract family; at B = 1 024 four still win 27-30 % on the first two but lose 14 % on the contract family: two.)
        // (14-slot kernels, two wavefronts per SIMD: from 0.7 device fills on -- 1 434 instances, kFillsLarge -- the throughput kernels with
        //  the resumable solve and the tail hand-off are ahead of two wavefronts per instance: configs[2]'s dimensions,
        //  B = 2 100 / 3 200 / 4 096: `passing` 34.5 / 50.7 / 43.8 -> 22.3 / 30.3 / 28.3 ms, reference scenarios 40.0 / 49.7 /
        //  50.3 -> 37.4 / 38.6 / 39.5, contract family 73.6 / 94.2 / ~100 -> 56.2 / 70.2 / 87.1; below a fill the two are
        //  level. The 4- and 6-slot kernels keep the latency plan up to 4 096: level or ahead there. tools/exp_mid_batches.py)
        if (sizeof(T) == 4 && L.rs >= kRegSlotsLarge && !L.glb && B >= kFillsLarge * 2 * cap) lw = 1;
        // At most one workgroup per CU (what a fleet's real-time loop sends: a handful of robots): six wavefronts -- the master
        // and five workers, the Lipschitz evaluation + five candidates in the first round of an iteration (1.3 instead of 1.6-1.8
        // rounds per iteration on the long instances). B = 64 / 256: 23.4 -> 21.8 / 18.5 -> 17.3 ms; from two workgroups per CU
        // on (B = 512) six or eight wavefronts cost more than they bring
        // (profiles/r04_exp_cfg1_batch_size_and_up_to_8_wavefronts.txt). Same results, bit for bit.
        // (the 4-slot register table only: what was measured, and what nmpc_hip.h documents -- ADVICE r4)
        // (the 6-slot table, same register budget: measured too -- B = 64 / 256: 22.98 -> 21.37 / 18.12 -> 16.96 ms, tools/exp_mid_w6.py;
        //  the 14-slot table: round 6, below)
        if (sizeof(T) == 4 && L.rs > 0 && !L.glb && 4 * B <= h->n_simd) lw = kSpecWavesWide;
        // Large batches whose LDS tables allow only a few workgroups per CU (e.g. 40 active obstacle rows: 35 KB,
        // 4 per CU = one wavefront per SIMD): the wavefronts of a latency-kernel workgroup SHARE the instance's
        // tables, so W of them fill the SIMDs that the throughput kernel leaves empty (measured on configs[2]:
        // 16.4 k -> 24.0 k solves/s with W = 3). Smallest W that reaches the resident-wavefront limit, if that is
        // at least 1.5x what the throughput kernel gets.
        // Obstacle table streamed from the global workspace (GLB): the wavefronts of a workgroup read the same
        // 236 KB at about the same time, so speculation rides on cache hits (configs[4]: 1.31 k -> 1.58 k solves/s)
        if (L.glb) lw = kSpecWaves;
        // resident wavefronts per CU: register budget of the kernel variant (wpe<>) x 4 SIMDs, capped by LDS
        const bool f32 = sizeof(T) == 4;
        const int wpe_tp = !f32 ? NMPC_WPE_F64 : L.rs >= kRegSlotsLarge ? 2 : L.rs > 0 ? 3 : NMPC_WPE_F32;
        const int wpe_sp = !f32 ? NMPC_WPE_F64 : L.rs >= kRegSlotsLarge ? 2 : NMPC_SPEC_WPE_F32;
        const size_t elem = sizeof(T);
        const int tp = std::min<int>(4 * wpe_tp, (int)(kLdsLimit / ((size_t)L.lds_total * elem)));
        const int wg_spec = (int)(kLdsLimit / ((size_t)(L.lds_xch + spec_xch_elems(kSpecWaves)) * elem));
        int best = tp;
        for (int w = std::max(lw, 2); w <= kSpecWaves; ++w) {
            const int res = std::min(4 * wpe_sp / w, wg_spec) * w;
            if (2 * res >= 3 * tp && res > best) {
                best = res;
                lw = w;
            }
        }
    }
    int waves = lw == 1 ? 0 : lw < 0 ? 1 : lw > kSpecWavesMax ? kSpecWavesMax : lw;
    if (!h->spec_ok[sizeof(T) == 4 ? 0 : 1]) waves = 0;
    // cooperative evaluation (nmpc_config.coop_waves): explicit request, or automatic where the obstacle table is streamed
    // from global memory (configs[4]: the obstacle loop is 94 % of the time and its rows split cleanly over the
    // wavefronts). Needs the LDS / global table (not the register table), room for the exchange area and no wall-clock
    // budget (each wavefront would read its own clock).
    int coop = h->cfg.coop_waves > kSpecWaves ? kSpecWaves : h->cfg.coop_waves;
    if (coop == 0) coop = (L.glb && h->cfg.latency_waves == 0) ? kSpecWaves : 1;
    if (L.rs > 0 || h->cfg.max_solver_time_us > 0 || !h->coop_ok[sizeof(T) == 4 ? 0 : 1]) coop = 1;
    // (latency kernel: the exchange area by the W actually launched -- about half of the 8-wavefront maximum at W = 4)
    pl.lds_bytes = (size_t)(waves ? L.lds_xch + spec_xch_elems(waves) : L.lds_total) * sizeof(T);
    pl.fn = waves ? pick_solve_spec<T>(h->lps, L.glb, L.rs) : pick_solve<T>(h->lps, L.glb, L.rs);
    pl.has_axis = L.rs > 0 && h->lps == 3 && !L.glb;
    if (pl.has_axis) pl.fn2 = waves ? pick_solve_spec<T>(h->lps, L.glb, L.rs, 2) : pick_solve<T>(h->lps, L.glb, L.rs, 2);
    // nmpc_config.batch_invariant: the latency plan on the TAIL members -- the throughput kernels' evaluation, hence their bits.
    // Automatic (0): the 14-slot kernels. There the gated form is also the faster one on everything but the contract family
    // (configs[2]'s dimensions, B = 1 .. 1 300: `passing` -8..-20 %, the reference scenarios -26..+1 %, the corridor family
    // -18..+19 %, one instance alone -7..-17 % on all four families; the contract family -15..+20 % --
    // profiles/r06_exp_mid_batches.txt); the 4- / 6-slot kernels keep the flat form configs[1] is quoted on.
    if (waves && (h->cfg.batch_invariant > 0 || (h->cfg.batch_invariant == 0 && L.rs >= kRegSlotsLarge)))
        if (SolveFn<T> tf = pick_solve_tail<T>(h->lps, L.glb, L.rs, 1)) {
            pl.fn = tf;
            if (pl.has_axis) pl.fn2 = pick_solve_tail<T>(h->lps, L.glb, L.rs, 2);
        }
    pl.uses_ws = L.glb;
    if (coop > 1) {
        // global table: the pair (compressed table of axis-aligned ellipses / general table); LDS table: one kernel
        pl.fn = pick_solve_coop<T>(h->lps, L.glb, L.glb ? 1 : 2);
        pl.fn2 = L.glb ? pick_solve_coop<T>(h->lps, true, 2) : nullptr;
        pl.has_axis = L.glb;
        pl.lds_bytes = (size_t)L.lds_total_coop * sizeof(T);
        k.lds_xch = L.lds_xch_coop;
        waves = coop;
        if constexpr (sizeof(T) == 4) {
            // the register-table variant is available (automatic / 4-wavefront request): eight wavefronts hold the table,
            // nothing is streamed from global memory
            if (coop == kSpecWaves && h->lay32c.rs > 0 && h->lps == 1) {
                waves = kCoopRegWaves;
                const Layout& C = h->lay32c;
                fill_layout(k, C);
                k.lds_xch = C.lds_xch_coop;
                pl.fn = pick_solve_coop_reg(h->cfg.N_hor);
                pl.fn2 = nullptr;
                pl.has_axis = false;
                pl.lds_bytes = (size_t)C.lds_total_coop * sizeof(T);
                pl.uses_ws = false;
            }
        }
    }
    if constexpr (sizeof(T) == 8) {
        // fp64, three lanes per step, 13..42 rows: the register-table kernel (one wavefront per SIMD, four instances per CU)
        // where the LDS table leaves room for fewer -- measured on configs[2]'s dimensions: ahead of the latency kernel
        // from B = 1024 on (97 vs 122 ms; 500 vs 696 ms at 8192), behind it below; never ahead for <= 12 rows, whose LDS
        // table is small. reg_table = 1 forces it (tests), -1 switches it off.
        const bool wanted = h->cfg.reg_table > 0 || (h->use64r_auto && B >= 2 * cap && h->cfg.latency_waves <= 1);
        if (h->lay64r.rs > 0 && coop <= 1 && h->cfg.latency_waves <= 1 && wanted) {
            const Layout& R = h->lay64r;
            fill_layout(k, R);
            waves = 0;
            pl.fn = pick_solve<T>(h->lps, false, R.rs);
            pl.fn2 = pick_solve<T>(h->lps, false, R.rs, 2);
            pl.lds_bytes = (size_t)R.lds_total * sizeof(T);
            pl.has_axis = true;
            pl.uses_ws = false;
            pl.resident = h->n_simd;
        }
    }
    pl.threads = waves ? 64 * waves : 64;
    pl.mode = coop > 1 ? 2 : waves ? 1 : 0;
    pl.stageable = h->cfg.max_solver_time_us <= 0; // (every kernel family parks / resumes; a wall-clock budget does not survive it)
    if (pl.mode == 2) { // cooperative kernels: workgroups resident on the device (LDS-bound; one per CU for the on-chip variant)
        const int per_cu = std::max<int>(1, (int)(kLdsLimit / std::max<size_t>(pl.lds_bytes, 1)));
        pl.resident = std::min(per_cu, std::max(1, 8 / (pl.threads / 64))) * (h->n_simd / 4);
    }
    return pl;
}

// one launch of the planned kernel over `grid` workgroups
template <typename T>
int launch_plan(nmpc_handle_s* h, const Plan<T>& pl, const nmpc::KParams<T>& k, int grid, hipStream_t stream = nullptr, bool own = true)
{
    if (own) stream = h->stream;
    if (grid <= 0) return 0;
    if (!pl.fn2 || k.axis_mode != 0) { // (axis_mode 0 = the general path only: the axis-only kernel of a pair has nothing to do)
        hipLaunchKernelGGL(pl.fn, dim3(grid), dim3(pl.threads), pl.lds_bytes, stream, k);
        HIP_TRY(hipGetLastError());
    }
    if (pl.fn2 && k.axis_mode != 1) {
        hipLaunchKernelGGL(pl.fn2, dim3(grid), dim3(pl.threads), pl.lds_bytes, stream, k);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// the tail member that goes with a throughput plan (fn == nullptr: none -- other kernel family, fp64, LDS / global table)
template <typename T>
Plan<T> plan_tail(nmpc_handle_s* h, const Plan<T>& pl, const Layout& L, int waves)
{
    Plan<T> t;
    if (pl.mode != 0 || pl.uses_ws || !h->spec_ok[sizeof(T) == 4 ? 0 : 1]) return t;
    t.fn = pick_solve_tail<T>(h->lps, L.glb, L.rs, 1);
    if (!t.fn) return t;
    t.fn2 = pl.has_axis ? pick_solve_tail<T>(h->lps, L.glb, L.rs, 2) : nullptr;
    t.has_axis = pl.has_axis;
    t.threads = 64 * waves;
    t.lds_bytes = (size_t)(L.lds_xch + spec_xch_elems(waves)) * sizeof(T);
    t.mode = 1;
    return t;
}

// How the axis-aligned variant takes part in a call over B instances at P (device): sets k.axis_mode to 0 (general only),
// 1 (the caller's promise) or 2 (decided on the device: the scan of the batch's angle entries is enqueued here).
template <typename T>
int prepare_axis(nmpc_handle_s* h, bool has_axis, nmpc::KParams<T>& k, int B)
{
    k.axis_mode = 0;
    if (!has_axis || h->cfg.axis_aligned < 0) return 0;
    if (h->cfg.axis_aligned > 0) {
        k.axis_mode = 1;
        return 0;
    }
    const Layout& L = h->lay<T>();
    const int n_entries = h->cfg.Ndynobs * (h->cfg.N_hor + 1);
    h->axis_epoch = h->axis_epoch == 0x7ffffff0 ? 1 : h->axis_epoch + 1;
    k.axis_mode = 2;
    k.axis_epoch = h->axis_epoch;
    k.axis_flag = static_cast<const int*>(h->dflag.p);
    if (n_entries > 0) {
        hipLaunchKernelGGL(axis_scan_kernel<T>, dim3(B), dim3(256), 0, h->stream, k.P, L.np, L.off_od, n_entries,
                           static_cast<int*>(h->dflag.p), h->axis_epoch);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// Solve B instances whose buffers (all on the device) are in `k`: kernel choice, the axis-aligned twin, the two-launch
// resumable solve. `allow_staging`: the caller's status array may be used for the in-progress marker.
template <typename T>
int run_solve(nmpc_handle_s* h, nmpc::KParams<T>& k, int B, bool allow_staging)
{
    const Layout& L = h->lay<T>();
    Plan<T> pl = plan_solve<T>(h, B, k);
    if (pl.uses_ws) {
        if (int rc_ = h->dws.reserve((size_t)B * L.ws_stride * sizeof(T))) return rc_;
        k.ws = static_cast<T*>(h->dws.p);
        k.ws_stride = L.ws_stride;
    } else {
        k.ws = nullptr;
        k.ws_stride = 0;
    }
    if (int rc = prepare_axis<T>(h, pl.has_axis, k, B)) return rc;
    h->last_mode = pl.mode;
    h->last_axis = pl.has_axis ? k.axis_mode : -1;
    // Resumable solve: up to two stage boundaries (outer-iteration count, ranking key of the launch that follows).
    //  * first (nmpc_config.staged, ranked by ||F2||) -- the instances whose hard constraints are still violated after the
    //    first inner solve are the ones that will run into the iteration caps. Automatic (one outer iteration) for
    //    batches that fill the device at least four times over with the one-wavefront kernel: started first, the long
    //    solves no longer end the launch alone (configs[2] `passing`: 156 -> 121 ms); and for the latency kernel with
    //    about one workgroup per SIMD, where every workgroup is resident at once and the order decides which long solves
    //    share a SIMD to the end (configs[1]: 32.6 -> 28.5 ms; costs ~1.5 ms where most instances converge early).
    //  * second (nmpc_config.staged_evals, ranked by the evaluations used so far) -- explicit only: every boundary is a
    //    barrier (each stage ends with ITS longest instance), and late boundaries lost more to that than the better
    //    ranking returned in every measurement (tools/exp_cfg1_order2.py, tools/sim_stages.py).
    int caps[2] = {h->cfg.staged, h->cfg.staged_evals};
    const int wpe_tp = sizeof(T) == 8 ? NMPC_WPE_F64 : L.rs >= kRegSlotsLarge ? 2 : L.rs > 0 ? 3 : NMPC_WPE_F32;
    const int resident = std::max(1, std::min<int>(wpe_tp * h->n_simd,
                                                   (int)(kLdsLimit / ((size_t)L.lds_total * sizeof(T))) * (h->n_simd / 4)));
    const int lat_cap = sizeof(T) == 4 ? h->n_simd : h->n_simd / 2;
    const bool reg32 = sizeof(T) == 4 && L.rs > 0 && !L.glb;
    // Latency plan at about one workgroup per SIMD (configs[1]): the order decides which long solves share a SIMD to the end.
    // Until round 6 a pilot launch ranked them (32.6 -> 28.5 ms then); the barrier and the second launch cost ~1.3 ms of
    // 20, and an order from ONE evaluation -- ||F2||^2 at the nominal controls (2/3 v_max, 0), ~20 us -- does as well on the
    // contract family below full oversubscription (B = 600 / 800: 21.5 -> 20.1 / 20.4 ms) and needs no barrier: `passing` -5..-7 %,
    // the reference scenarios -6.5 %, configs[2]'s dimensions at 1 024: -2..-9 % (tools/exp_cfg1_proxy_order.py,
    // profiles/r06_exp_proxy_order.txt; at 65 536 instances the pilot stays ahead).
    // One measured exception keeps the pilot: the 4- / 6-slot kernels at full oversubscription (B > 7/8 of the SIMD count, four
    // wavefronts per instance on three slots per SIMD -- configs[1] itself), where the order decides which quarter of the
    // workgroups waits for a slot and the pilot's better key is worth its barrier on the contract family (21.1 against 21.6 ms;
    // `passing` would gain 7 % there too: 20.6 -> 19.2).
    // The two-wavefront plans of the 4- / 6-slot kernels (1 024 < B <= 4 096, up to 2.7x oversubscribed) had no ranking at all: the
    // same evaluation order there -- B = 2 500 / 4 096: contract family -10 / -12 %, corridor -9 %, `passing` -4 / -6 %, the
    // reference scenarios +-2 %; a pilot launch loses or ties everywhere (same record). 14-slot kernels, 769..1 433: neutral, none.
    const double fills_ = !reg32 ? kStageFills : L.rs >= kRegSlotsLarge ? kFillsLarge : kStageFillsReg;
    const int res_ = pl.resident ? pl.resident : resident;
    // ... and the throughput plans up to eight device fills (tools/exp_cfg1_proxy_order.py at 3 000 .. 40 000 instances, both
    // dimensions, same record): one launch in the evaluation order + the tail hand-off against pilot + ranking + hand-off --
    // `passing` / the reference scenarios -8..-11 % up to six fills, the contract family -3..+1 %; level at eight fills; from ten
    // on the pilot's key wins (65 536: `passing` 85.7 against 91.3 ms) and stays.
    const bool proxy = reg32 && caps[0] == 0 && allow_staging && pl.stageable && !k.order && k.status && dev_factor("NMPC_PROXY_ORDER", 1) > 0 &&
                       (pl.mode == 1 ? (B > lat_cap / 2 && (B <= lat_cap ? (L.rs >= kRegSlotsLarge || 8 * B <= 7 * lat_cap)
                                                                         : (L.rs < kRegSlotsLarge && B <= 4 * lat_cap)))
                                     : (pl.mode == 0 && B >= fills_ * res_ && B < dev_factor("NMPC_PROXY_FILLS", kProxyFills) * res_));
    if (proxy) caps[0] = -1;
    if (caps[0] == 0)
        caps[0] = ((pl.mode == 0 && B >= dev_factor("NMPC_STAGE_FILLS", !reg32 ? kStageFills : L.rs >= kRegSlotsLarge ? kFillsLarge : kStageFillsReg) * (pl.resident ? pl.resident : resident)) || (pl.mode == 1 && B > lat_cap / 2 && B <= lat_cap) ||
                   (pl.mode == 2 && !pl.uses_ws && B >= 4 * pl.resident)) ? 1 : -1; // (configs[4] fp32: 2 116 -> 2 087 ms;
                                                                                   //  streamed table, fp64: -2 %, off)
    if (caps[1] == 0) caps[1] = -1;
    const bool stageable = allow_staging && pl.stageable && !k.order && k.status;
    int n_stage = 0, stage_cap[2], stage_key[2];
    for (int i = 0; i < 2; ++i)
        if (stageable && caps[i] > 0 && caps[i] < h->cfg.max_outer_iterations && (n_stage == 0 || caps[i] > stage_cap[n_stage - 1])) {
            stage_cap[n_stage] = caps[i];
            stage_key[n_stage] = i;
            ++n_stage;
        }
    h->last_staged = n_stage == 0 ? 0 : n_stage == 1 ? stage_cap[0] : 100 * stage_cap[0] + stage_cap[1];
    h->last_tail = 0;
    // (the evaluation order below goes into k.order for the rest of this function only: the caller's k gets its own back)
    struct OrderGuard {
        nmpc::KParams<T>& kk;
        decltype(nmpc::KParams<T>::order) saved;
        ~OrderGuard() { kk.order = saved; }
    } order_guard{k, k.order};
    if (proxy) {
        const size_t n = 2 * (size_t)h->cfg.N_hor;
        if (int rc = h->dresume.reserve((size_t)B * nmpc::kResumeStride * sizeof(T))) return rc;
        if (int rc = h->dorder2.reserve((size_t)B * sizeof(int))) return rc;
        if (int rc = h->dproxy.reserve((2 * (size_t)B * n + 3 * (size_t)B) * sizeof(T))) return rc;
        T* const Un = static_cast<T*>(h->dproxy.p);
        nmpc::EvalParams<T> ep;
        ep.U = Un, ep.Y = Un + (size_t)B * n, ep.C = Un + 2 * (size_t)B * n;
        ep.psi = Un + 2 * (size_t)B * n + B, ep.grad = nullptr, ep.f2sq = Un + 2 * (size_t)B * n + 2 * (size_t)B;
        const int nbf = (int)(((size_t)B * n + 255) / 256), nb = (B + 255) / 256;
        hipLaunchKernelGGL(proxy_fill_kernel<T>, dim3(nbf), dim3(256), 0, h->stream, Un, Un + (size_t)B * n, Un + 2 * (size_t)B * n, B, (int)n,
                           T(dev_factor("NMPC_PROXY_VNOM", 2.0 / 3.0)) * k.vmax, k.c_init);
        const size_t lds_eval = (size_t)L.lds_total * sizeof(T);
        if (k.axis_mode != 0) hipLaunchKernelGGL(pick_eval<T>(h->lps, L.glb, L.rs, 1), dim3(B), dim3(64), lds_eval, h->stream, k, ep);
        if (k.axis_mode != 1) hipLaunchKernelGGL(pick_eval<T>(h->lps, L.glb, L.rs, 2), dim3(B), dim3(64), lds_eval, h->stream, k, ep);
        T* const resume = static_cast<T*>(h->dresume.p);
        int* const hist = static_cast<int*>(h->dhist.p);
        int* const order2 = static_cast<int*>(h->dorder2.p);
        hipLaunchKernelGGL(proxy_key_kernel<T>, dim3(nb), dim3(256), 0, h->stream, ep.f2sq, resume, k.status, B);
        hipLaunchKernelGGL(rank_hist_kernel<T>, dim3(nb), dim3(256), 0, h->stream, resume, k.status, B, hist, 0);
        hipLaunchKernelGGL(rank_scan_kernel, dim3(1), dim3(kRankBuckets), 0, h->stream, hist, hist + kRankBuckets, (int*)nullptr);
        hipLaunchKernelGGL(rank_scatter_kernel<T>, dim3(nb), dim3(256), 0, h->stream, resume, k.status, B, hist + kRankBuckets, order2, 0);
        HIP_TRY(hipGetLastError());
        k.order = order2;   // (from here on as under a caller's order: one launch, the tail hand-off where the batch is big enough)
        if (pl.mode == 1) return launch_plan<T>(h, pl, k, B);
    }
    // Tail hand-off (nmpc_config.tail_latency; round 6). A launch of the throughput kernel ends with
    // whatever long solves are still running -- one wavefront each, alone on its SIMD -- while the rest of the chip idles: on
    // batches with a skewed distribution of solve lengths the launch IS its longest instance
    // (profiles/r05_cfg2_passing_kernel_timeline.txt). The latency family's TAIL member (solve_spec_kernel<.., FLAT = false>:
    // speculative line search over six wavefronts, the throughput kernels' own evaluation) computes the throughput kernels'
    // bits and solves a long instance ~1.8x faster on an idle chip (tools/exp_tail_solo.py). So the LAST throughput launch
    // of a solve parks whatever is still running once it is in its drain phase -- every workgroup dispatched, at most
    // `park` instances left (KParams::dyn_ctr) -- at the instance's next outer-iteration boundary, and one more launch
    // -- the tail member over the parked instances, found by the same ranking kernels -- finishes them. Who solves which
    // part of an instance depends on timing; the results do not (tests/test_gpu_tail.py). (Running the two families side by
    // side on two streams does not work: with tens of thousands of one-wavefront workgroups pending, a four-wavefront
    // workgroup never finds its four slots on one CU and runs after the throughput launch -- profiles/r06_ab_tail_handoff.jsonl.)
    h->last_tail = 0;
    int park = h->cfg.tail_latency;
    if (park == 0) park = std::max(32, h->n_simd / 4);      // automatic: one tail workgroup per CU
    // (six wavefronts per parked instance while they are all resident at two per SIMD, else four: `passing` 87.0 -> 84.4 ms per
    //  call, batches of 3 000 / 6 000: -8 / -6 %; eight bring nothing more -- profiles/r06_exp_mid_batches.txt)
    const int tail_waves = (int)dev_factor("NMPC_TAIL_WAVES", park * kSpecWavesWide <= 2 * h->n_simd ? kSpecWavesWide : kSpecWaves);
    const bool big = B >= dev_factor("NMPC_TAIL_FILLS", !reg32 ? kTailFills : L.rs >= kRegSlotsLarge ? kFillsLarge : kTailFillsReg) * (pl.resident ? pl.resident : resident);
    const Plan<T> tail = (park > 0 && allow_staging && k.status && pl.stageable && (n_stage > 0 || (k.order && big)) && B >= dev_factor("NMPC_TAIL_MINB", kTailMinParks) * park)
                             ? plan_tail<T>(h, pl, L, tail_waves) : Plan<T>();
    if (n_stage == 0 && !tail.fn) return launch_plan<T>(h, pl, k, B);
    if (tail.fn) {
        if (int rc = h->ddeep.reserve((size_t)park * nmpc::deep_park_stride(h->cfg.N_hor) * sizeof(T))) return rc;
        k.deep = static_cast<T*>(h->ddeep.p);   // (read by the tail launch; handed to the LAST throughput launch only)
    }

    if (int rc = h->dresume.reserve((size_t)B * nmpc::kResumeStride * sizeof(T))) return rc;
    if (int rc = h->dorder2.reserve((size_t)B * sizeof(int))) return rc;
    k.resume = static_cast<T*>(h->dresume.p);
    int* hist = static_cast<int*>(h->dhist.p);
    int* offs = hist + kRankBuckets;
    int* dctr = offs + kRankBuckets;             // (KParams::dyn_ctr: four counters behind the bucket tables)
    int* order2 = static_cast<int*>(h->dorder2.p);
    const int nb = (B + 255) / 256;
    auto rank = [&](int key, bool publish) {
        hipLaunchKernelGGL(rank_hist_kernel<T>, dim3(nb), dim3(256), 0, h->stream, k.resume, k.status, B, hist, key);
        hipLaunchKernelGGL(rank_scan_kernel, dim3(1), dim3(kRankBuckets), 0, h->stream, hist, offs, publish ? dctr : nullptr);
        hipLaunchKernelGGL(rank_scatter_kernel<T>, dim3(nb), dim3(256), 0, h->stream, k.resume, k.status, B, offs, order2, key);
        return hipGetLastError();
    };
    for (int i = 0; i <= n_stage; ++i) {
        nmpc::KParams<T> ki = k;
        ki.deep = nullptr;
        ki.stage_in = i > 0;
        ki.stage_outer_cap = i < n_stage ? stage_cap[i] : 0;
        if (i > 0) ki.order = order2;
        if (i == n_stage && tail.fn) { // the last throughput launch parks its drain phase (the counters: set by the ranking before it)
            if (n_stage == 0) {        // (one launch under the caller's order: all B instances are to be solved)
                hipLaunchKernelGGL(dyn_init_kernel, dim3(1), dim3(1), 0, h->stream, dctr, B);
                HIP_TRY(hipGetLastError());
            }
            ki.dyn_ctr = dctr;
            ki.dyn_park = park;
            ki.deep = k.deep;       // (parking inside an inner solve: the slots reserved below)
            ki.deep_slots = park;
        }
        if (int rc = launch_plan<T>(h, pl, ki, B)) return rc;
        if (i == n_stage) break;
        HIP_TRY(rank(stage_key[i], i + 1 == n_stage && tail.fn != nullptr));
    }
    if (tail.fn) {
        // the parked instances (status -1) first in order2 -- at most `park` of them by construction -- and the tail member over them
        HIP_TRY(rank(0, false));
        nmpc::KParams<T> kt = k;
        kt.stage_in = 1;
        kt.stage_outer_cap = 0;
        kt.order = order2;
        if (int rc = launch_plan<T>(h, tail, kt, std::min(B, park))) return rc;
        h->last_tail = park;
    }
    return 0;
}

template <typename T>
int polish_batch(nmpc_handle_s* h, const nmpc::KParams<T>& k, int B, bool y_user, bool info_user);

template <typename T>
int solve_batch(nmpc_handle_s* h, const T* P, int32_t B, T* U, T* cost, int32_t* status, int32_t* iters,
                const T* u0, T* y, int32_t y_is_input, const T* c0, T* info, int32_t sync)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    if (!P || !U) return fail(NMPC_ERR_INVALID_ARGUMENT, "P and U must not be NULL");
    if (B < 0) return fail(NMPC_ERR_INVALID_ARGUMENT, "B = %d < 0", B);
    if (B == 0) return 0;
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    const Layout& L = h->lay<T>();
    const size_t n = 2 * (size_t)h->cfg.N_hor, np = L.np;
    nmpc::KParams<T> k;
    fill_kparams(h, k);
    k.B = B;
    k.y_is_input = y_is_input;
    bool hU, hcost, hstatus, hiters, hy, hinfo;
    int rc;
    if ((rc = stage_in(h, h->dP, P, (size_t)B * np, &k.P))) return rc;
    if ((rc = stage_in(h, h->du0, u0, (size_t)B * n, &k.u0))) return rc;
    if ((rc = stage_in(h, h->dc0, c0, (size_t)B, &k.c0v))) return rc;
    if ((rc = stage_out(h, h->dU, U, (size_t)B * n, &k.U, &hU))) return rc;
    if ((rc = stage_out(h, h->dcost, cost, (size_t)B, &k.cost, &hcost))) return rc;
    if ((rc = stage_out(h, h->dstatus, status, (size_t)B, &k.status, &hstatus))) return rc;
    if ((rc = stage_out(h, h->diters, iters, (size_t)B * 2, &k.iters, &hiters))) return rc;
    if ((rc = stage_out(h, h->dy, y, (size_t)B * n, &k.y, &hy))) return rc;
    const size_t info_row = 8 + nmpc::kProfSlots; // 8 in the shipped library (kProfSlots = 0)
    if ((rc = stage_out(h, h->dinfo, info, (size_t)B * info_row, &k.info, &hinfo))) return rc;
    if (hy && y_is_input) HIP_TRY(hipMemcpyAsync(k.y, y, (size_t)B * n * sizeof(T), hipMemcpyHostToDevice, h->stream));
    // arrays the caller did not ask for but the resumable solve (status) / the polish (status, y, info) need: the
    // handle's own buffers stand in
    const bool polish = h->cfg.polish > 0 && nmpc::kProfSlots == 0;
    const bool y_user = y != nullptr, info_user = info != nullptr;
    if (!k.status) {
        if ((rc = h->dstatus.reserve((size_t)B * sizeof(int32_t)))) return rc;
        k.status = static_cast<int*>(h->dstatus.p);
    }
    if (polish && !k.y) {
        if ((rc = h->dy.reserve((size_t)B * n * sizeof(T)))) return rc;
        k.y = static_cast<T*>(h->dy.p);
    }
    if (polish && !k.info) {
        if ((rc = h->dinfo.reserve((size_t)B * info_row * sizeof(T)))) return rc;
        k.info = static_cast<T*>(h->dinfo.p);
    }
    k.order = h->order_B == B ? static_cast<const int*>(h->dorder.p) : nullptr;

    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if ((rc = run_solve<T>(h, k, B, true))) return rc;
    h->last_polish_selected = 0;
    h->last_polish_converged = -1;
    if (polish && (rc = polish_batch<T>(h, k, B, y_user, info_user))) return rc;
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true;

    const bool any_host = hU || hcost || hstatus || hiters || hy || hinfo;
    if (hU) HIP_TRY(hipMemcpyAsync(U, k.U, (size_t)B * n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (hcost) HIP_TRY(hipMemcpyAsync(cost, k.cost, (size_t)B * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (hstatus)
        HIP_TRY(hipMemcpyAsync(status, k.status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (hiters)
        HIP_TRY(hipMemcpyAsync(iters, k.iters, (size_t)B * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (hy) HIP_TRY(hipMemcpyAsync(y, k.y, (size_t)B * n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (hinfo)
        HIP_TRY(hipMemcpyAsync(info, k.info, (size_t)B * info_row * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (any_host || sync) HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// nmpc_config.polish: fp64 continuation of the instances the main solve flagged Converged (results of the main solve
// in k.U / k.y / k.info / k.status on the device). One stream synchronisation: the selection is made on the host.
template <typename T>
int polish_batch(nmpc_handle_s* h, const nmpc::KParams<T>& k, int B, bool y_user, bool info_user)
{
    const int n2 = 2 * h->cfg.N_hor;
    h->host_status.resize((size_t)B);
    HIP_TRY(hipMemcpyAsync(h->host_status.data(), k.status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->host_sel.clear();
    for (int b = 0; b < B; ++b)
        if (h->host_status[(size_t)b] == NMPC_CONVERGED) h->host_sel.push_back(b);
    const int ns = (int)h->host_sel.size();
    h->last_polish_selected = ns;
    h->last_polish_converged = 0;
    if (info_user) {
        hipLaunchKernelGGL(polish_clear_flag_kernel<T>, dim3((B + 255) / 256), dim3(256), 0, h->stream, k.info, B);
        HIP_TRY(hipGetLastError());
    }
    if (ns == 0) return 0;
    const Layout& L64 = h->lay64;
    const size_t np = (size_t)L64.np;
    int rc;
    if ((rc = h->psel.reserve((size_t)ns * sizeof(int)))) return rc;
    if ((rc = h->pP.reserve((size_t)ns * np * sizeof(double)))) return rc;
    if ((rc = h->pU0.reserve((size_t)ns * n2 * sizeof(double)))) return rc;
    if ((rc = h->pY.reserve((size_t)ns * n2 * sizeof(double)))) return rc;
    if ((rc = h->pC.reserve((size_t)ns * sizeof(double)))) return rc;
    if ((rc = h->pU.reserve((size_t)ns * n2 * sizeof(double)))) return rc;
    if ((rc = h->pcost.reserve((size_t)ns * sizeof(double)))) return rc;
    if ((rc = h->pstatus.reserve((size_t)ns * sizeof(int)))) return rc;
    if ((rc = h->piters.reserve((size_t)ns * 2 * sizeof(int)))) return rc;
    if ((rc = h->pinfo.reserve((size_t)ns * 8 * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpyAsync(h->psel.p, h->host_sel.data(), (size_t)ns * sizeof(int), hipMemcpyHostToDevice, h->stream));
    const int* sel = static_cast<const int*>(h->psel.p);
    hipLaunchKernelGGL(polish_gather_kernel<T>, dim3(ns), dim3(256), 0, h->stream, k.P, (const T*)k.U, (const T*)k.y,
                       (const T*)k.info, sel, (int)np, n2, static_cast<double*>(h->pP.p), static_cast<double*>(h->pU0.p),
                       static_cast<double*>(h->pY.p), static_cast<double*>(h->pC.p));
    HIP_TRY(hipGetLastError());
    nmpc::KParams<double> q;
    fill_kparams(h, q);
    const nmpc_config& c = h->cfg;
    // The exit test bounds ||gamma fpr||, which pins u only to ~tol / gamma, and the curvature of the path term grows with
    // the lever arm of the horizon: at N = 40 a tolerance of 1e-6 leaves the controls 2-3e-4 from the fixed point where
    // N = 20 gets 2e-5. Beyond the reference's horizon the continuation's tolerance therefore shrinks with (20 / N)^3
    // (N = 40: 1.25e-7) and its iteration cap grows with (N / 20)^2.
    const double hscale = c.N_hor > 20 ? 20.0 / c.N_hor : 1.0;
    q.tol = q.init_tol = c.polish_tolerance * hscale * hscale * hscale;
    q.max_inner = (int)(c.polish_max_inner_iterations / (hscale * hscale) + 0.5);
    if (c.polish == 1) {
        // ONE inner solve at the penalty and multipliers the main solve ended with (KParams::single_inner): what the
        // continuation needs is stationarity to polish_tolerance; the hard-constraint criterion stays at the main solve's delta.
        // Measured (tools/exp_polish_stats.py, configs[2] `passing`): the first inner solve of the ALM continuation takes
        // 123 evaluations and leaves the controls 1.8e-5 (median) from the 1e-8 fixed point, 86.6 % below 1e-4; the outer
        // iterations behind it -- the second one is mandatory in OpEn's loop -- add 131 evaluations for 1.1e-5 / 86.9 %.
        q.single_inner = 1;
        q.max_outer = 1;
        q.delta_tol = c.delta_tolerance;
    } else {
        q.delta_tol = c.polish_delta_tolerance;
        q.max_outer = c.polish_max_outer_iterations;
    }
    q.time_budget = 0;
    q.max_evals = 0; // (the continuation of a converged instance is not budgeted)
    q.B = ns;
    q.P = static_cast<const double*>(h->pP.p);
    q.u0 = static_cast<const double*>(h->pU0.p);
    q.y = static_cast<double*>(h->pY.p);
    q.y_is_input = 1;
    q.c0v = static_cast<const double*>(h->pC.p);
    q.U = static_cast<double*>(h->pU.p);
    q.cost = static_cast<double*>(h->pcost.p);
    q.status = static_cast<int*>(h->pstatus.p);
    q.iters = static_cast<int*>(h->piters.p);
    q.info = static_cast<double*>(h->pinfo.p);
    const int keep_mode = h->last_mode, keep_axis = h->last_axis, keep_staged = h->last_staged, keep_tail = h->last_tail;
    rc = run_solve<double>(h, q, ns, false);
    h->last_mode = keep_mode, h->last_axis = keep_axis, h->last_staged = keep_staged, h->last_tail = keep_tail;
    if (rc) return rc;
    hipLaunchKernelGGL(polish_scatter_kernel<T>, dim3(ns), dim3(128), 0, h->stream, sel, (const double*)q.U, (const double*)q.y,
                       (const double*)q.cost, (const int*)q.status, (const int*)q.iters, (const double*)q.info, n2, k.U, k.y,
                       y_user, k.cost, k.iters, k.info, info_user);
    HIP_TRY(hipGetLastError());
    return 0;
}

// nmpc_solve_trace_f64: ONE instance (host pointers) through trace_kernel; see include/nmpc_hip.h
int solve_trace(nmpc_handle_s* h, const double* p, const double* u0, const double* y0, double c0, double* U, double* y,
                int32_t* status, int32_t* iters, double* info, double* trace, int32_t max_records, int32_t* n_records)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    if (!p || !U || !trace || !n_records || max_records < 1)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "p, U, trace, n_records must not be NULL and max_records >= 1");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    const Layout& L = h->lay64;
    const size_t n = 2 * (size_t)h->cfg.N_hor, np = L.np;
    const size_t rec = (size_t)nmpc::kTraceHead + n, tlen = (size_t)nmpc::kTraceHead + (size_t)max_records * rec;
    nmpc::KParams<double> k;
    fill_kparams(h, k);
    k.B = 1;
    int rc;
    struct Scoped : DevBuf { // (local buffers: freed on every path out of this function)
        ~Scoped() { release(); }
    } dtrace, dc0;
    if ((rc = dtrace.reserve(tlen * sizeof(double)))) return rc;
    if ((rc = h->dP.reserve(np * sizeof(double)))) return rc;
    if ((rc = h->dU.reserve(n * sizeof(double)))) return rc;
    if ((rc = h->dy.reserve(n * sizeof(double)))) return rc;
    if ((rc = h->du0.reserve(n * sizeof(double)))) return rc;
    if ((rc = h->dcost.reserve(sizeof(double)))) return rc;
    if ((rc = h->dstatus.reserve(sizeof(int32_t)))) return rc;
    if ((rc = h->diters.reserve(2 * sizeof(int32_t)))) return rc;
    if ((rc = h->dinfo.reserve((8 + nmpc::kProfSlots) * sizeof(double)))) return rc; // (a -DNMPC_PROFILE build writes its slots behind the 8)
    if ((rc = dc0.reserve(sizeof(double)))) return rc;
    // the handle's staging buffers are shared with nmpc_solve_batch_*: a solve enqueued with sync = 0 must have finished
    // with them (its results copied out) before this call overwrites them
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpyAsync(h->dP.p, p, np * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (u0) HIP_TRY(hipMemcpyAsync(h->du0.p, u0, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (y0) HIP_TRY(hipMemcpyAsync(h->dy.p, y0, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (c0 > 0) HIP_TRY(hipMemcpyAsync(dc0.p, &c0, sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(dtrace.p, 0, nmpc::kTraceHead * sizeof(double), h->stream));
    k.P = static_cast<const double*>(h->dP.p);
    k.u0 = u0 ? static_cast<const double*>(h->du0.p) : nullptr;
    k.y = static_cast<double*>(h->dy.p);
    k.y_is_input = y0 != nullptr;
    k.c0v = c0 > 0 ? static_cast<const double*>(dc0.p) : nullptr;
    k.U = static_cast<double*>(h->dU.p);
    k.cost = static_cast<double*>(h->dcost.p);
    k.status = static_cast<int*>(h->dstatus.p);
    k.iters = static_cast<int*>(h->diters.p);
    k.info = static_cast<double*>(h->dinfo.p);
    k.trace = static_cast<double*>(dtrace.p);
    k.trace_cap = max_records;
    k.time_budget = 0;
    if (L.glb) {
        if ((rc = h->dws.reserve((size_t)L.ws_stride * sizeof(double)))) return rc;
        k.ws = static_cast<double*>(h->dws.p);
        k.ws_stride = L.ws_stride;
    }
    using Fn = void (*)(nmpc::KParams<double>);
    const Fn fn = L.glb ? (h->lps == 3 ? (Fn)trace_kernel<3, true> : h->lps == 2 ? (Fn)trace_kernel<2, true> : (Fn)trace_kernel<1, true>)
                        : (h->lps == 3 ? (Fn)trace_kernel<3, false> : h->lps == 2 ? (Fn)trace_kernel<2, false> : (Fn)trace_kernel<1, false>);
    const size_t lds_bytes = (size_t)L.lds_total * sizeof(double);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(fn, dim3(1), dim3(64), lds_bytes, h->stream, k);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(U, k.U, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (y) HIP_TRY(hipMemcpyAsync(y, k.y, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (status) HIP_TRY(hipMemcpyAsync(status, k.status, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (iters) HIP_TRY(hipMemcpyAsync(iters, k.iters, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (info) HIP_TRY(hipMemcpyAsync(info, k.info, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    std::vector<double> host(tlen);
    HIP_TRY(hipMemcpyAsync(host.data(), dtrace.p, tlen * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int nr = (int)host[0];
    *n_records = nr;
    std::memcpy(trace, host.data() + nmpc::kTraceHead, (size_t)nr * rec * sizeof(double));
    return 0;
}

template <typename T>
int eval_batch(nmpc_handle_s* h, const T* P, const T* U, const T* Y, const T* C, int32_t B, T* psi, T* grad,
               T* f2sq)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    if (!P || !U || !Y || !C || !psi) return fail(NMPC_ERR_INVALID_ARGUMENT, "P, U, Y, C, psi must not be NULL");
    if (B <= 0) return B == 0 ? 0 : fail(NMPC_ERR_INVALID_ARGUMENT, "B = %d < 0", B);
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    const Layout& L = h->lay<T>();
    const size_t n = 2 * (size_t)h->cfg.N_hor, np = L.np;
    nmpc::KParams<T> k;
    fill_kparams(h, k);
    k.B = B;
    nmpc::EvalParams<T> ep;
    bool hpsi, hgrad, hf2;
    int rc;
    if ((rc = stage_in(h, h->dP, P, (size_t)B * np, &k.P))) return rc;
    if ((rc = stage_in(h, h->du0, U, (size_t)B * n, &ep.U))) return rc;
    if ((rc = stage_in(h, h->dY2, Y, (size_t)B * n, &ep.Y))) return rc;
    if ((rc = stage_in(h, h->dC2, C, (size_t)B, &ep.C))) return rc;
    if ((rc = stage_out(h, h->dpsi, psi, (size_t)B, &ep.psi, &hpsi))) return rc;
    if ((rc = stage_out(h, h->dgrad, grad, (size_t)B * n, &ep.grad, &hgrad))) return rc;
    if ((rc = stage_out(h, h->df2, f2sq, (size_t)B, &ep.f2sq, &hf2))) return rc;
    size_t lds_bytes = (size_t)L.lds_total * sizeof(T);
    EvalFn<T> fn = pick_eval<T>(h->lps, L.glb, L.rs), fn2 = nullptr;
    bool has_axis = L.rs > 0 && h->lps == 3 && !L.glb;
    if (has_axis) fn2 = pick_eval<T>(h->lps, L.glb, L.rs, 2); // (register-table kernels are pairs: the general-path member)
    if constexpr (sizeof(T) == 8) { // (what solves of large fp64 batches run: the register-table kernel where it is offered)
        if (h->lay64r.rs > 0 && h->cfg.coop_waves <= 1 && h->cfg.latency_waves <= 1 && (h->cfg.reg_table > 0 || h->use64r_auto)) {
            const Layout& R = h->lay64r;
            fill_layout(k, R);
            fn = pick_eval<T>(h->lps, false, R.rs);
            fn2 = eval_kernel<T, 3, false, kRegSlotsLarge, 2>;
            lds_bytes = (size_t)R.lds_total * sizeof(T);
            has_axis = true;
        }
    }
    bool uses_ws = L.glb;
    int waves = 1;
    // coop_waves > 1: evaluate through the cooperative kernels' code path (same variant choice as solve_batch)
    if (h->cfg.coop_waves > 1 && L.rs == 0 && h->coop_ok[sizeof(T) == 4 ? 0 : 1]) {
        waves = std::min<int>(h->cfg.coop_waves, kSpecWaves);
        fn = pick_eval_coop<T>(h->lps, L.glb, L.glb ? 1 : 2);
        fn2 = L.glb ? pick_eval_coop<T>(h->lps, true, 2) : nullptr;
        has_axis = L.glb;
        lds_bytes = (size_t)L.lds_total_coop * sizeof(T);
        k.lds_xch = L.lds_xch_coop;
        if constexpr (sizeof(T) == 4) {
            if (waves == kSpecWaves && h->lay32c.rs > 0 && h->lps == 1) {
                const Layout& C = h->lay32c;
                fill_layout(k, C);
                k.lds_xch = C.lds_xch_coop;
                waves = kCoopRegWaves;
                fn = pick_eval_coop_reg(h->cfg.N_hor);
                fn2 = nullptr;
                has_axis = false;
                lds_bytes = (size_t)C.lds_total_coop * sizeof(T);
                uses_ws = false;
            }
        }
    }
    if (uses_ws) { // (only the variants that stream the obstacle table reserve the global workspace)
        if (int rc_ = h->dws.reserve((size_t)B * L.ws_stride * sizeof(T))) return rc_;
        k.ws = static_cast<T*>(h->dws.p);
        k.ws_stride = L.ws_stride;
    }
    if ((rc = prepare_axis<T>(h, has_axis, k, B))) return rc;
    h->last_mode = waves > 1 ? 2 : 0;
    h->last_axis = has_axis ? k.axis_mode : -1;
    h->last_staged = 0;
    h->last_polish_selected = 0;
    if (!fn2 || k.axis_mode != 0) {
        hipLaunchKernelGGL(fn, dim3(B), dim3(64 * waves), lds_bytes, h->stream, k, ep);
        HIP_TRY(hipGetLastError());
    }
    if (fn2 && k.axis_mode != 1) {
        hipLaunchKernelGGL(fn2, dim3(B), dim3(64 * waves), lds_bytes, h->stream, k, ep);
        HIP_TRY(hipGetLastError());
    }
    if (hpsi) HIP_TRY(hipMemcpyAsync(psi, ep.psi, (size_t)B * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (hgrad) HIP_TRY(hipMemcpyAsync(grad, ep.grad, (size_t)B * n * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    if (hf2) HIP_TRY(hipMemcpyAsync(f2sq, ep.f2sq, (size_t)B * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

template <typename T>
int assemble_params(nmpc_handle_s* h, const nmpc_assemble_args* g, int32_t B, T* P)
{
    if (!h || !g || !P) return fail(NMPC_ERR_INVALID_ARGUMENT, "null argument");
    if (B <= 0) return B == 0 ? 0 : fail(NMPC_ERR_INVALID_ARGUMENT, "B = %d < 0", B);
    const nmpc_config& c = h->cfg;
    if (g->n_dyn < 0 || g->n_dyn > c.Ndynobs)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "n_dyn = %d outside [0, Ndynobs = %d]", g->n_dyn, c.Ndynobs);
    if (g->n_map_polygons < 0) return fail(NMPC_ERR_INVALID_ARGUMENT, "n_map_polygons < 0");
    const void* required[] = {g->last_u, g->state, g->ref_states, g->speed_ref, g->tuning, g->stc_weights,
                              g->dyn_weights, P};
    for (const void* q : required)
        if (!is_device_ptr(q))
            return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_assemble_params: every array must be a device pointer");
    if ((g->n_map_polygons > 0 && !is_device_ptr(g->map_polygons)) || (g->n_dyn > 0 && !is_device_ptr(g->dyn_obstacles)) ||
        (g->other_robots && !is_device_ptr(g->other_robots)))
        return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_assemble_params: every array must be a device pointer");
    HIP_TRY(hipSetDevice(c.device_id));
    const Layout& L = h->lay<T>();
    nmpc::AsmParams<T> a;
    std::memset(&a, 0, sizeof a);
    a.N = c.N_hor;
    a.Nother = c.Nother;
    a.Nstc = c.Nstcobs;
    a.Ndyn = c.Ndynobs;
    a.np = L.np;
    a.off_rs = L.off_rs;
    a.off_rv = L.off_rv;
    a.off_c0 = L.off_c0;
    a.off_os = L.off_os;
    a.off_od = L.off_od;
    a.off_qstc = L.off_qstc;
    a.off_qdyn = L.off_qdyn;
    a.B = B;
    a.M = g->n_map_polygons;
    a.n_dyn = g->n_dyn;
    a.last_u = static_cast<const T*>(g->last_u);
    a.state = static_cast<const T*>(g->state);
    a.ref_states = static_cast<const T*>(g->ref_states);
    a.speed_ref = static_cast<const T*>(g->speed_ref);
    a.tuning = static_cast<const T*>(g->tuning);
    a.other_robots = static_cast<const T*>(g->other_robots);
    a.map_polygons = static_cast<const T*>(g->map_polygons);
    a.dyn = g->n_dyn > 0 ? static_cast<const T*>(g->dyn_obstacles) : nullptr;
    a.stc_weights = static_cast<const T*>(g->stc_weights);
    a.dyn_weights = static_cast<const T*>(g->dyn_weights);
    a.P = P;
    a.selected = g->selected;
    if (g->selected && !is_device_ptr(g->selected))
        return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_assemble_params: every array must be a device pointer");
    const size_t lds = (size_t)a.M * sizeof(T) + (size_t)a.Nstc * sizeof(int) + 16;
    if (lds > kLdsLimit) return fail(NMPC_ERR_UNSUPPORTED, "%d map polygons do not fit the selection kernel's LDS", a.M);
    // (every argument is validated before the first launch)
    // element pairs (8 / 16 B per lane) need every block boundary and every source row pair-aligned
    const unsigned per = 6u * (a.N + 1);
    const bool even = (a.np % 2 == 0) && (a.off_od % 2 == 0) && (a.off_qstc % 2 == 0) && (a.off_c0 % 2 == 0) &&
                      (a.off_os % 2 == 0) && ((a.n_dyn * per) % 2 == 0);
    const auto aligned = [](const void* q, size_t al) { return q == nullptr || reinterpret_cast<uintptr_t>(q) % al == 0; };
    const unsigned vec_ok = even && aligned(a.P, 2 * sizeof(T)) && aligned(a.dyn, 2 * sizeof(T)) &&
                            aligned(a.other_robots, 2 * sizeof(T));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(nmpc::assemble_kernel<T>, dim3((unsigned)B), dim3(nmpc::kAsmThreads), lds, h->stream, a, vec_ok);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}

template <typename T>
int hypotheses_to_ellipses(nmpc_handle_s* h, const T* hypos, int32_t P, const T* cur, int32_t H, double human_size,
                           double eps, double enlarge, double extra_margin, int32_t B, T* dyn, int32_t* n_obs)
{
    if (!h || !hypos || !cur || !dyn) return fail(NMPC_ERR_INVALID_ARGUMENT, "null argument");
    if (B <= 0) return B == 0 ? 0 : fail(NMPC_ERR_INVALID_ARGUMENT, "B = %d < 0", B);
    if (P < 1 || P > 256) return fail(NMPC_ERR_UNSUPPORTED, "P = %d hypothesis points per time offset outside [1, 256]", P);
    if (h->cfg.N_hor > 64) return fail(NMPC_ERR_UNSUPPORTED, "nmpc_hypotheses_to_ellipses: N_hor = %d > 64", h->cfg.N_hor);
    if (H < 0 || H > h->cfg.Ndynobs) return fail(NMPC_ERR_INVALID_ARGUMENT, "H = %d outside [0, Ndynobs]", H);
    if (!is_device_ptr(hypos) || !is_device_ptr(cur) || !is_device_ptr(dyn) || (n_obs && !is_device_ptr(n_obs)))
        return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_hypotheses_to_ellipses: every array must be a device pointer");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    nmpc::HypParams<T> a;
    a.B = B;
    a.N = h->cfg.N_hor;
    a.P = P;
    a.H = H;
    a.Ndyn = h->cfg.Ndynobs;
    a.eps = (T)eps;
    a.human_size = (T)human_size;
    a.enlarge = (T)enlarge;
    a.extra_margin = (T)extra_margin;
    a.hypos = hypos;
    a.cur = cur;
    a.dyn = dyn;
    a.n_obs = n_obs;
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if (P <= 32)
        hipLaunchKernelGGL((nmpc::hypotheses_kernel<T, unsigned>), dim3(B), dim3(64), 0, h->stream, a);
    else if (P <= 64)
        hipLaunchKernelGGL((nmpc::hypotheses_kernel<T, unsigned long long>), dim3(B), dim3(64), 0, h->stream, a);
    else if (P <= 128)
        hipLaunchKernelGGL((nmpc::hypotheses_wide_kernel<T, 2>), dim3(B), dim3(64), 0, h->stream, a);
    else if (P <= 192)
        hipLaunchKernelGGL((nmpc::hypotheses_wide_kernel<T, 3>), dim3(B), dim3(64), 0, h->stream, a);
    else
        hipLaunchKernelGGL((nmpc::hypotheses_wide_kernel<T, 4>), dim3(B), dim3(64), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return 0;
}

template <typename T>
int loop_step(nmpc_handle_s* h, const nmpc_loop_args* g, bool post)
{
    if (!h || !g) return fail(NMPC_ERR_INVALID_ARGUMENT, "null argument");
    if (g->n_run <= 0) return g->n_run == 0 ? 0 : fail(NMPC_ERR_INVALID_ARGUMENT, "n_run < 0");
    if (g->B < g->n_run || g->H < 1 || g->H > 64 || g->W < 1 || g->Lmax < 1 || g->M < 0 || g->step < 0 || g->step >= g->max_steps)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_loop_*: bad dimensions (B %d, n_run %d, H %d, W %d, Lmax %d, M %d, step %d of %d)",
                    g->B, g->n_run, g->H, g->W, g->Lmax, g->M, g->step, g->max_steps);
    if (!g->run && g->n_run != g->B) return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_loop_*: run = NULL needs n_run = B");
    if (g->n_hyp < 0 || (g->n_hyp > 1 && (long long)g->H * g->n_hyp > h->cfg.Ndynobs))
        return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_loop_*: H * n_hyp = %d x %d rows exceed Ndynobs = %d", g->H, g->n_hyp, h->cfg.Ndynobs);
    const void* need[] = {g->robot, g->last_u, g->humans, g->hist, g->hcount, g->hidx, g->hpath, g->ref_traj, g->ref_len,
                          g->idx_ref, g->goal, g->alive, g->collision, g->complete, g->steps, g->clr_dyn, g->clr_stc,
                          g->dev_sum, g->dev_max, g->n_traj, g->traj, g->acts, g->state_c, g->last_u_c, g->refs_c,
                          g->speed_c, g->dyn_c, g->U_c, g->y_c, g->U, g->y};
    for (const void* q : need)
        if (!q) return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_loop_*: a required array is NULL");
    if (g->M > 0 && !g->polys) return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_loop_*: polys is NULL");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    // (a host pointer here would fault inside the kernel: three samples of the argument block are looked up)
    if (h->ptr_mode != NMPC_PTR_DEVICE && (!is_device_ptr(g->robot) || !is_device_ptr(g->U_c) || !is_device_ptr(g->dyn_c)))
        return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_loop_*: every array must be a device pointer");
    nmpc::LoopParams<T> p;
    std::memset(&p, 0, sizeof p);
    p.B = g->B, p.n_run = g->n_run, p.N = h->cfg.N_hor, p.H = g->H, p.W = g->W, p.Lmax = g->Lmax, p.M = g->M;
    p.step = g->step, p.max_steps = g->max_steps;
    p.run = reinterpret_cast<const long long*>(g->run);
    p.ts = (T)h->cfg.ts, p.base_speed = (T)g->base_speed, p.lin_vel_max = (T)g->lin_vel_max;
    p.human_size = (T)g->human_size, p.human_vmax = (T)g->human_vmax;
    p.robot = static_cast<T*>(g->robot), p.last_u = static_cast<T*>(g->last_u), p.humans = static_cast<T*>(g->humans);
    p.hist = static_cast<T*>(g->hist);
    p.hcount = reinterpret_cast<long long*>(g->hcount), p.hidx = reinterpret_cast<long long*>(g->hidx);
    p.hpath = static_cast<const T*>(g->hpath), p.ref_traj = static_cast<const T*>(g->ref_traj);
    p.ref_len = reinterpret_cast<const long long*>(g->ref_len), p.idx_ref = reinterpret_cast<long long*>(g->idx_ref);
    p.goal = static_cast<const T*>(g->goal), p.polys = static_cast<const T*>(g->polys);
    p.stagger = static_cast<const T*>(g->stagger);
    p.alive = g->alive, p.collision = g->collision, p.complete = g->complete;
    p.steps = reinterpret_cast<long long*>(g->steps);
    p.clr_dyn = static_cast<T*>(g->clr_dyn), p.clr_stc = static_cast<T*>(g->clr_stc), p.dev_sum = static_cast<T*>(g->dev_sum);
    p.dev_max = static_cast<T*>(g->dev_max), p.n_traj = static_cast<T*>(g->n_traj);
    p.traj = static_cast<T*>(g->traj), p.acts = static_cast<T*>(g->acts);
    p.state_c = static_cast<T*>(g->state_c), p.last_u_c = static_cast<T*>(g->last_u_c), p.refs_c = static_cast<T*>(g->refs_c);
    p.speed_c = static_cast<T*>(g->speed_c), p.dyn_c = static_cast<T*>(g->dyn_c);
    p.U_c = static_cast<T*>(g->U_c), p.y_c = static_cast<T*>(g->y_c), p.U = static_cast<T*>(g->U), p.y = static_cast<T*>(g->y);
    p.gather_y = g->gather_y;
    p.n_hyp = g->n_hyp, p.hyp_fan = (T)g->hyp_fan_rad, p.hyp_r0 = (T)g->hyp_radius0, p.hyp_grow = (T)g->hyp_radius_growth;
    if (post)
        hipLaunchKernelGGL(nmpc::loop_post_kernel<T>, dim3(g->n_run), dim3(64), 0, h->stream, p);
    else
        hipLaunchKernelGGL(nmpc::loop_pre_kernel<T>, dim3(g->n_run), dim3(64), 0, h->stream, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
int set_lds_limit(nmpc_handle_s* h)
{
    const Layout& L = h->lay<T>();
    const size_t lds_bytes = (size_t)L.lds_total * sizeof(T);
    if (lds_bytes > 48 * 1024)
        for (int only = 1; only <= 2; ++only) { // (register-table kernels: both members of the pair; else the same kernel twice)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_solve<T>(h->lps, L.glb, L.rs, only)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_eval<T>(h->lps, L.glb, L.rs, only)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        }
    if (sizeof(T) == 4 && h->lay32c.rs > 0) {
        const size_t cb = (size_t)h->lay32c.lds_total_coop * sizeof(float);
        if (cb > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_solve_coop_reg(h->cfg.N_hor)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)cb));
        if (cb > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_eval_coop_reg(h->cfg.N_hor)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)cb));
    }
    const size_t coop_bytes = (size_t)L.lds_total_coop * sizeof(T);
    if (coop_bytes > kLdsLimit) {
        h->coop_ok[sizeof(T) == 4 ? 0 : 1] = false;
    } else if (coop_bytes > 48 * 1024) {
        for (int only = 1; only <= 2; ++only) { // (global table: both members of the pair; else the same kernel twice)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_solve_coop<T>(h->lps, L.glb, only)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_bytes));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_eval_coop<T>(h->lps, L.glb, only)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_bytes));
        }
    }
    const size_t spec_bytes = (size_t)L.lds_total_spec * sizeof(T);
    if (spec_bytes > kLdsLimit) {
        h->spec_ok[sizeof(T) == 4 ? 0 : 1] = false; // no room for the exchange area: latency mode unavailable
    } else if (spec_bytes > 48 * 1024) {
        for (int only = 1; only <= 2; ++only) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pick_solve_spec<T>(h->lps, L.glb, L.rs, only)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)spec_bytes));
            if (auto* tf = pick_solve_tail<T>(h->lps, L.glb, L.rs, only))
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)spec_bytes));
        }
    }
    return 0;
}

} // namespace

extern "C" {

const char* nmpc_last_error(void) { return g_err.c_str(); }

int nmpc_default_config(nmpc_config* c)
{
    if (!c) return fail(NMPC_ERR_INVALID_ARGUMENT, "null config");
    std::memset(c, 0, sizeof *c);
    c->abi_version = NMPC_ABI_VERSION;
    c->device_id = 0;
    // config/mpc_fast.yaml == config/mpc_default.yaml for everything the solver consumes
    c->N_hor = 20;
    c->Nother = 10;
    c->Nstcobs = 10;
    c->Ndynobs = 15;
    c->ts = 0.2;
    c->lin_vel_min = -0.5;
    c->lin_vel_max = 1.5;
    c->ang_vel_max = 0.5;
    c->lin_acc_min = -1.0;
    c->lin_acc_max = 1.0;
    c->ang_acc_max = 3.0;
    c->vehicle_width = 0.5;
    c->vehicle_margin = 0.2;
    c->social_margin = 0.2;
    c->tolerance = 1e-4;
    c->initial_tolerance = 1e-4;
    c->delta_tolerance = 1e-4;
    c->max_outer_iterations = 10;
    c->max_inner_iterations = 500;
    c->lbfgs_memory = 10;
    c->max_active_dynobs = 0;
    c->initial_penalty = 10.0;
    c->penalty_update_factor = 5.0;
    c->inner_tolerance_update_factor = 0.1;
    c->sufficient_decrease_coeff = 0.1;
    c->lip_eps_f64 = 1e-6;
    c->lip_delta_f64 = 1e-12;
    c->lip_eps_f32 = 1e-4;
    c->lip_delta_f32 = 1e-4;
    c->cbfgs_alpha = 1.0;
    c->cbfgs_epsilon = 1e-8;
    c->sy_epsilon = 1e-10;
    c->latency_waves = 0;
    c->akkt_form = 0;
    c->max_solver_time_us = 0.0;
    c->coop_waves = 0;
    c->axis_aligned = 0;
    c->reg_table = 0;
    c->staged = 0;
    c->polish = 0;
    c->polish_max_outer_iterations = 4;
    c->polish_max_inner_iterations = 150; // (round 4: 300 -> 150, see nmpc_hip.h)
    c->staged_evals = 0;
    c->polish_tolerance = 1e-6;
    c->polish_delta_tolerance = 1e-5;
    c->max_evaluations = 0;
    c->tail_latency = 0;
    c->batch_invariant = 0;
    return 0;
}

int nmpc_layout(const nmpc_config* cfg, nmpc_layout_info* out)
{
    if (!cfg || !out) return fail(NMPC_ERR_INVALID_ARGUMENT, "null argument");
    if (cfg->abi_version != NMPC_ABI_VERSION)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "abi_version %d != %d", cfg->abi_version, NMPC_ABI_VERSION);
    if (cfg->N_hor < 1 || cfg->N_hor > NMPC_MAX_HORIZON || cfg->Nother < 1 || cfg->Nstcobs < 0 || cfg->Ndynobs < 0 ||
        cfg->max_active_dynobs < 0)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "bad dimensions");
    const Layout a = make_layout(*cfg, sizeof(float)), b = make_layout(*cfg, sizeof(double));
    std::memset(out, 0, sizeof *out);
    out->np = a.np;
    out->lds_bytes_f32 = (int32_t)((size_t)a.lds_total * sizeof(float));
    out->lds_bytes_f64 = (int32_t)((size_t)b.lds_total * sizeof(double));
    out->reg_slots_f32 = a.rs;
    out->global_table_f32 = a.glb ? 1 : 0;
    out->global_table_f64 = b.glb ? 1 : 0;
    out->ws_elems_f32 = a.ws_stride;
    out->ws_elems_f64 = b.ws_stride;
    out->table_entries_f32 = a.table_entries;
    out->table_entries_f64 = b.table_entries;
    out->dyn_cap = a.dyn_cap;
    return 0;
}

int nmpc_create(const nmpc_config* cfg, nmpc_handle* out)
{
    if (!cfg || !out) return fail(NMPC_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (cfg->abi_version != NMPC_ABI_VERSION)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "abi_version %d != %d", cfg->abi_version, NMPC_ABI_VERSION);
    if (cfg->N_hor < 1 || cfg->N_hor > NMPC_MAX_HORIZON)
        return fail(NMPC_ERR_UNSUPPORTED, "N_hor = %d outside [1, %d] (one horizon step per lane group)", cfg->N_hor,
                    NMPC_MAX_HORIZON);
    if (cfg->Nother < 1 || cfg->Nstcobs < 0 || cfg->Ndynobs < 0)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "bad dimensions Nother=%d Nstcobs=%d Ndynobs=%d", cfg->Nother,
                    cfg->Nstcobs, cfg->Ndynobs);
    if (cfg->max_active_dynobs < 0)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "max_active_dynobs = %d < 0", cfg->max_active_dynobs);
    if (cfg->lbfgs_memory < 1 || cfg->lbfgs_memory > NMPC_LBFGS_MAX_MEMORY)
        return fail(NMPC_ERR_UNSUPPORTED, "lbfgs_memory = %d outside [1, %d]", cfg->lbfgs_memory,
                    NMPC_LBFGS_MAX_MEMORY);
    if (cfg->latency_waves < -1)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "latency_waves = %d < -1", cfg->latency_waves);
    if (cfg->coop_waves < 0 || cfg->coop_waves > kSpecWaves)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "coop_waves = %d outside [0, %d]", cfg->coop_waves, kSpecWaves);
    if (cfg->akkt_form != 0 && cfg->akkt_form != 1)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "akkt_form = %d (0 = OpEn source form, 1 = documented form)", cfg->akkt_form);
    if (!(cfg->max_solver_time_us >= 0))
        return fail(NMPC_ERR_INVALID_ARGUMENT, "max_solver_time_us < 0");
    if (cfg->max_evaluations < 0) return fail(NMPC_ERR_INVALID_ARGUMENT, "max_evaluations = %d < 0", cfg->max_evaluations);
    if (cfg->tail_latency < -1) return fail(NMPC_ERR_INVALID_ARGUMENT, "tail_latency = %d < -1", cfg->tail_latency);
    if (cfg->batch_invariant < -1 || cfg->batch_invariant > 1)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "batch_invariant = %d (0 automatic, 1 always, -1 never)", cfg->batch_invariant);
    if (cfg->staged_evals < -1) return fail(NMPC_ERR_INVALID_ARGUMENT, "staged_evals = %d < -1", cfg->staged_evals);
    if (cfg->axis_aligned < -1 || cfg->axis_aligned > 1)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "axis_aligned = %d (0 automatic, 1 promised, -1 never)", cfg->axis_aligned);
    if (cfg->staged < -1) return fail(NMPC_ERR_INVALID_ARGUMENT, "staged = %d < -1", cfg->staged);
    if (cfg->polish < 0 || cfg->polish > 2) return fail(NMPC_ERR_INVALID_ARGUMENT, "polish = %d (0, 1 or 2)", cfg->polish);
    if (cfg->polish && (!(cfg->polish_tolerance > 0) || !(cfg->polish_delta_tolerance > 0) ||
                        cfg->polish_max_outer_iterations < 1 || cfg->polish_max_inner_iterations < 1))
        return fail(NMPC_ERR_INVALID_ARGUMENT, "bad polish tolerances / iteration caps");
    if (!(cfg->ts > 0) || cfg->max_outer_iterations < 1 || cfg->max_inner_iterations < 1 ||
        !(cfg->initial_penalty > 0))
        return fail(NMPC_ERR_INVALID_ARGUMENT, "bad ts / iteration caps / initial penalty");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(NMPC_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(NMPC_ERR_INVALID_ARGUMENT, "device_id %d not in [0, %d)", cfg->device_id, ndev);
    nmpc_handle_s* h = new (std::nothrow) nmpc_handle_s();
    if (!h) return fail(NMPC_ERR_OUT_OF_MEMORY, "host allocation failed");
    h->cfg = *cfg;
    h->lay32 = make_layout(*cfg, sizeof(float));
    h->lay64 = make_layout(*cfg, sizeof(double));
    h->lay32c = make_layout(*cfg, sizeof(float), true);
    h->lay64r = make_layout(*cfg, sizeof(double), false, true);
    h->use64r_auto = h->lay64r.rs > 0 && !h->lay64.glb &&
                     std::min<size_t>(4 * NMPC_WPE_F64, kLdsLimit / ((size_t)h->lay64.lds_total * sizeof(double))) < 4;
    h->lps = 64 / cfg->N_hor;
    if (h->lps > 3) h->lps = 3;
    if (h->lps < 1) h->lps = 1;
    if ((size_t)h->lay64.lds_total * sizeof(double) > kLdsLimit || (size_t)h->lay32.lds_total * sizeof(float) > kLdsLimit) {
        const size_t need = (size_t)h->lay64.lds_total * sizeof(double);
        delete h;
        return fail(NMPC_ERR_UNSUPPORTED,
                    "polygon / robot / path tables alone need %zu B of LDS per instance (> 160 KiB)", need);
    }
    hipError_t e = hipSetDevice(cfg->device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) {
        nmpc_destroy(h);
        return fail(NMPC_ERR_HIP, "handle setup: %s", hipGetErrorString(e));
    }
    h->stream = h->own_stream;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device_id) == hipSuccess)
            h->n_simd = prop.multiProcessorCount * 4; // 4 SIMDs per CU
    }
    int rc = h->dflag.reserve(4 * sizeof(int));
    if (rc == 0 && hipMemset(h->dflag.p, 0, 4 * sizeof(int)) != hipSuccess) rc = fail(NMPC_ERR_HIP, "hipMemset failed");
    if (rc == 0) rc = h->dhist.reserve((2 * kRankBuckets + 4) * sizeof(int)); // (bucket counters of the resumable solve, kept zeroed; + the tail hand-off's three)
    if (rc == 0 && hipMemset(h->dhist.p, 0, (2 * kRankBuckets + 4) * sizeof(int)) != hipSuccess) rc = fail(NMPC_ERR_HIP, "hipMemset failed");
    if (rc == 0) rc = set_lds_limit<float>(h);
    if (rc == 0) rc = set_lds_limit<double>(h);
    if (rc) {
        nmpc_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int nmpc_destroy(nmpc_handle h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device_id);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    for (DevBuf* b : {&h->dP, &h->dU, &h->dcost, &h->dstatus, &h->diters, &h->du0, &h->dy, &h->dc0, &h->dinfo,
                      &h->dY2, &h->dC2, &h->dpsi, &h->dgrad, &h->df2, &h->dws, &h->dorder, &h->dflag, &h->dresume,
                      &h->dorder2, &h->dhist, &h->ddeep, &h->psel, &h->pP, &h->pU0, &h->pY, &h->pC, &h->pU, &h->pcost, &h->pstatus,
                      &h->piters, &h->pinfo})
        b->release();
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return 0;
}

int nmpc_param_len(nmpc_handle h)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    return h->lay32.np;
}

int nmpc_set_stream(nmpc_handle h, void* s)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    h->stream = static_cast<hipStream_t>(s); // NULL is a stream too: the HIP null stream (torch's default stream)
    return 0;
}

int nmpc_set_pointer_mode(nmpc_handle h, int32_t mode)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    if (mode < NMPC_PTR_DETECT || mode > NMPC_PTR_DEVICE) return fail(NMPC_ERR_INVALID_ARGUMENT, "pointer mode %d", mode);
    h->ptr_mode = mode;
    return 0;
}

int nmpc_set_dispatch_order(nmpc_handle h, const int32_t* order, int32_t B)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    if (!order || B <= 0) {
        h->order_B = 0;
        return 0;
    }
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    if (int rc = h->dorder.reserve((size_t)B * sizeof(int32_t))) return rc;
    if (is_device_ptr(order)) {
        // (a device-resident order is taken as it is: it must be a permutation of 0..B-1)
        HIP_TRY(hipMemcpyAsync(h->dorder.p, order, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    } else {
        std::vector<unsigned char> seen((size_t)B, 0);
        for (int32_t i = 0; i < B; ++i) {
            const int32_t v = order[i];
            if (v < 0 || v >= B || seen[(size_t)v])
                return fail(NMPC_ERR_INVALID_ARGUMENT, "nmpc_set_dispatch_order: order[%d] = %d: not a permutation of 0..%d", i, v, B - 1);
            seen[(size_t)v] = 1;
        }
        HIP_TRY(hipMemcpyAsync(h->dorder.p, order, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream)); // (the host array may be released by the caller on return)
    }
    h->order_B = B;
    return 0;
}

int nmpc_use_own_stream(nmpc_handle h)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    h->stream = h->own_stream;
    return 0;
}

int nmpc_solve_batch_f32(nmpc_handle h, const float* P, int32_t B, float* U, float* cost, int32_t* status,
                         int32_t* iters, const float* u0, float* y, int32_t y_is_input, const float* c0, float* info,
                         int32_t sync)
{
    return solve_batch<float>(h, P, B, U, cost, status, iters, u0, y, y_is_input, c0, info, sync);
}

int nmpc_solve_batch_f64(nmpc_handle h, const double* P, int32_t B, double* U, double* cost, int32_t* status,
                         int32_t* iters, const double* u0, double* y, int32_t y_is_input, const double* c0,
                         double* info, int32_t sync)
{
    return solve_batch<double>(h, P, B, U, cost, status, iters, u0, y, y_is_input, c0, info, sync);
}

int nmpc_solve_trace_f64(nmpc_handle h, const double* p, const double* u0, const double* y0, double c0, double* U, double* y,
                         int32_t* status, int32_t* iters, double* info, double* trace, int32_t max_records,
                         int32_t* n_records)
{
    return solve_trace(h, p, u0, y0, c0, U, y, status, iters, info, trace, max_records, n_records);
}

int nmpc_eval_batch_f32(nmpc_handle h, const float* P, const float* U, const float* Y, const float* C, int32_t B,
                        float* psi, float* grad, float* f2sq)
{
    return eval_batch<float>(h, P, U, Y, C, B, psi, grad, f2sq);
}

int nmpc_eval_batch_f64(nmpc_handle h, const double* P, const double* U, const double* Y, const double* C, int32_t B,
                        double* psi, double* grad, double* f2sq)
{
    return eval_batch<double>(h, P, U, Y, C, B, psi, grad, f2sq);
}

int nmpc_assemble_params_f32(nmpc_handle h, const nmpc_assemble_args* args, int32_t B, float* P)
{
    return assemble_params<float>(h, args, B, P);
}

int nmpc_assemble_params_f64(nmpc_handle h, const nmpc_assemble_args* args, int32_t B, double* P)
{
    return assemble_params<double>(h, args, B, P);
}

int nmpc_hypotheses_to_ellipses_f32(nmpc_handle h, const float* hypos, int32_t P, const float* cur, int32_t H,
                                    double human_size, double eps, double enlarge, double extra_margin, int32_t B,
                                    float* dyn, int32_t* n_obs)
{
    return hypotheses_to_ellipses<float>(h, hypos, P, cur, H, human_size, eps, enlarge, extra_margin, B, dyn, n_obs);
}

int nmpc_hypotheses_to_ellipses_f64(nmpc_handle h, const double* hypos, int32_t P, const double* cur, int32_t H,
                                    double human_size, double eps, double enlarge, double extra_margin, int32_t B,
                                    double* dyn, int32_t* n_obs)
{
    return hypotheses_to_ellipses<double>(h, hypos, P, cur, H, human_size, eps, enlarge, extra_margin, B, dyn, n_obs);
}

int nmpc_loop_pre_f32(nmpc_handle h, const nmpc_loop_args* a) { return loop_step<float>(h, a, false); }
int nmpc_loop_pre_f64(nmpc_handle h, const nmpc_loop_args* a) { return loop_step<double>(h, a, false); }
int nmpc_loop_post_f32(nmpc_handle h, const nmpc_loop_args* a) { return loop_step<float>(h, a, true); }
int nmpc_loop_post_f64(nmpc_handle h, const nmpc_loop_args* a) { return loop_step<double>(h, a, true); }

int nmpc_last_kernel_ms(nmpc_handle h, float* ms)
{
    if (!h || !ms) return fail(NMPC_ERR_INVALID_ARGUMENT, "null argument");
    if (!h->timed) return fail(NMPC_ERR_INVALID_ARGUMENT, "no solve has been launched on this handle yet");
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return 0;
}

int nmpc_last_launch_info(nmpc_handle h, int32_t out[8])
{
    if (!h || !out) return fail(NMPC_ERR_INVALID_ARGUMENT, "null argument");
    std::memset(out, 0, 8 * sizeof(int32_t));
    out[0] = h->last_mode;
    out[1] = h->last_axis;
    out[2] = h->last_staged;
    out[3] = h->last_polish_selected;
    out[4] = h->last_tail;
    return 0;
}

int nmpc_kernel_info(nmpc_handle h, int32_t* lds_bytes_f32, int32_t* lds_bytes_f64, int32_t* lanes_per_step,
                     int32_t* waves_per_cu_f32, int32_t* waves_per_cu_f64)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    const size_t l32 = (size_t)h->lay32.lds_total * 4, l64 = (size_t)h->lay64.lds_total * 8;
    if (lds_bytes_f32) *lds_bytes_f32 = (int32_t)l32;
    if (lds_bytes_f64) *lds_bytes_f64 = (int32_t)l64;
    if (lanes_per_step) *lanes_per_step = h->lps;
    if (waves_per_cu_f32) {
        int nb = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &nb, reinterpret_cast<const void*>(pick_solve<float>(h->lps, h->lay32.glb, h->lay32.rs)), 64, l32));
        *waves_per_cu_f32 = nb;
    }
    if (waves_per_cu_f64) {
        int nb = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &nb, reinterpret_cast<const void*>(pick_solve<double>(h->lps, h->lay64.glb, h->lay64.rs)), 64, l64));
        *waves_per_cu_f64 = nb;
    }
    return 0;
}

int nmpc_selftest(nmpc_handle h)
{
    if (!h) return fail(NMPC_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device_id));
    DevBuf buf; // freed on every path below
    if (int rc = buf.reserve(2 * sizeof(int))) return rc;
    int* d = static_cast<int*>(buf.p);
    int res[2] = {-1, -1};
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(int), h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(selftest_kernel<float>, dim3(1), dim3(64), 0, h->stream, d);
        hipLaunchKernelGGL(selftest_kernel<double>, dim3(1), dim3(64), 0, h->stream, d + 1);
        e = hipMemcpyAsync(res, d, sizeof res, hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    buf.release();
    if (e != hipSuccess) return fail(NMPC_ERR_HIP, "selftest: %s", hipGetErrorString(e));
    if (res[0] || res[1]) {
        g_err = "wave primitive self-test failed: f32 mask " + std::to_string(res[0]) + ", f64 mask " +
                std::to_string(res[1]);
        return __builtin_popcount(res[0]) + __builtin_popcount(res[1]);
    }
    return 0;
}

} // extern "C"
