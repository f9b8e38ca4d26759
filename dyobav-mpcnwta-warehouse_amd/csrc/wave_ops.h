// wave_ops.h -- 64-lane wavefront primitives for gfx950 (CDNA4): DPP reductions / scans / shifts.
//
// One MPC instance lives in one wavefront; every inner product, prefix sum of the rollout and suffix sum of
// the adjoint is a cross-lane operation done in registers with DPP (no LDS round trip).
// All functions must be called with EXEC = all 64 lanes (wave-uniform control flow).
#pragma once

#include <hip/hip_runtime.h>

namespace nmpc {

// DPP control encodings (GFX9 family)
enum : int {
    DPP_QUAD_1032 = 0xB1,    // quad_perm:[1,0,3,2]
    DPP_QUAD_2301 = 0x4E,    // quad_perm:[2,3,0,1]
    DPP_ROW_SHR0 = 0x110,    // row_shr:n = 0x110 + n
    DPP_WAVE_SHL1 = 0x130,   // lane i <- lane i+1
    DPP_WAVE_SHR1 = 0x138,   // lane i <- lane i-1
    DPP_ROW_MIRROR = 0x140,
    DPP_ROW_HALF_MIRROR = 0x141,
    DPP_ROW_BCAST15 = 0x142,
    DPP_ROW_BCAST31 = 0x143,
};

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = false>
__device__ __forceinline__ float dpp_mov(float old, float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                  __builtin_bit_cast(int, x), CTRL, ROW_MASK,
                                                                  BANK_MASK, BOUND_CTRL));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = false>
__device__ __forceinline__ double dpp_mov(double old, double x)
{
    const unsigned long long o = __builtin_bit_cast(unsigned long long, old);
    const unsigned long long v = __builtin_bit_cast(unsigned long long, x);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)v, CTRL, ROW_MASK, BANK_MASK,
                                               BOUND_CTRL);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(v >> 32), CTRL,
                                               ROW_MASK, BANK_MASK, BOUND_CTRL);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// value of x in lane addr4 / 4 (ds_bpermute_b32: any lane pattern, through the LDS crossbar, no memory touched)
__device__ __forceinline__ float bperm(float x, int addr4)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr4, __builtin_bit_cast(int, x)));
}
__device__ __forceinline__ double bperm(double x, int addr4)
{
    const unsigned long long v = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(addr4, (int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(addr4, (int)(unsigned)(v >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float read_lane(float x, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
__device__ __forceinline__ double read_lane(double x, int lane)
{
    const unsigned long long v = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// ---- f32 reductions / scans: each step is x += dpp(x) as ONE VALU instruction (v_add_f32_dpp). Steps with full row and
//      bank masks go through update_dpp builtins: the compiler folds mov + add into v_add_f32_dpp (with
//      -fno-slp-vectorize) and fills the two wait states a DPP read needs after a VALU write of the same register with
//      the caller's independent instructions, inserting s_nop only where none is at hand. Steps with partial row masks
//      (row_bcast) would come out as v_mov (old) + v_mov_dpp + v_add: those are asm blocks with their own s_nop.

// Sum over the 64 lanes; the result is wave-uniform (read back from lane 63 into scalar registers).
__device__ __forceinline__ float wave_sum(float x)
{
    // The four in-row steps (full row / bank masks) through builtins: the compiler folds each into one v_add_f32_dpp and
    // fills the two wait states a dependent DPP read needs with whatever independent instructions the caller has (the
    // two-loop recursion: the next pair's LDS read, slot arithmetic, the alpha bookkeeping) instead of s_nop. The two
    // cross-row steps have partial row masks -- there it would emit v_mov + v_mov_dpp + v_add -- and stay an asm block.
    x += dpp_mov<DPP_QUAD_1032, 0xf, 0xf, true>(0.0f, x);
    x += dpp_mov<DPP_QUAD_2301, 0xf, 0xf, true>(0.0f, x);
    x += dpp_mov<DPP_ROW_HALF_MIRROR, 0xf, 0xf, true>(0.0f, x);
    x += dpp_mov<DPP_ROW_MIRROR, 0xf, 0xf, true>(0.0f, x);                                   // row sums everywhere
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"    // rows 1,3 += rows 0,2
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"        // rows 2,3 += rows 0+1
        : "+v"(x));
    return read_lane(x, 63);
}
__device__ __forceinline__ double wave_sum(double x)
{
    x += dpp_mov<DPP_QUAD_1032>(x, x);
    x += dpp_mov<DPP_QUAD_2301>(x, x);
    x += dpp_mov<DPP_ROW_HALF_MIRROR>(x, x);
    x += dpp_mov<DPP_ROW_MIRROR>(x, x);
    x += dpp_mov<DPP_ROW_BCAST15, 0xa>(0.0, x);
    x += dpp_mov<DPP_ROW_BCAST31, 0xc>(0.0, x);
    return read_lane(x, 63);
}

// Two / three independent sums at once with interleaved chains.
#define NMPC_P2(ctrl) "s_nop 0\n\tv_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\t"
#define NMPC_P3(ctrl) \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl "\n\t"
__device__ __forceinline__ void wave_sum2(float x, float y, float& sx, float& sy)
{
    // (in-row steps through builtins, cross-row steps in one asm block: see wave_sum)
    x += dpp_mov<DPP_QUAD_1032, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<DPP_QUAD_1032, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<DPP_QUAD_2301, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<DPP_QUAD_2301, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<DPP_ROW_HALF_MIRROR, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<DPP_ROW_HALF_MIRROR, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<DPP_ROW_MIRROR, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<DPP_ROW_MIRROR, 0xf, 0xf, true>(0.0f, y);
    asm("s_nop 0\n\t" NMPC_P2("row_bcast:15 row_mask:0xa bank_mask:0xf") NMPC_P2("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(x), "+v"(y));
    sx = read_lane(x, 63);
    sy = read_lane(y, 63);
}
__device__ __forceinline__ void wave_sum3(float x, float y, float z, float& sx, float& sy, float& sz)
{
    asm("s_nop 1\n\t" NMPC_P3("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
            NMPC_P3("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                NMPC_P3("row_half_mirror row_mask:0xf bank_mask:0xf") NMPC_P3("row_mirror row_mask:0xf bank_mask:0xf")
                    NMPC_P3("row_bcast:15 row_mask:0xa bank_mask:0xf")
                        NMPC_P3("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(x), "+v"(y), "+v"(z));
    sx = read_lane(x, 63);
    sy = read_lane(y, 63);
    sz = read_lane(z, 63);
}
// six independent sums: with six chains in flight no DPP wait state is left to pad
#define NMPC_P6(ctrl)                                                                                         \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl \
    "\n\tv_add_f32_dpp %3, %3, %3 " ctrl "\n\tv_add_f32_dpp %4, %4, %4 " ctrl "\n\tv_add_f32_dpp %5, %5, %5 " ctrl "\n\t"
__device__ __forceinline__ void wave_sum6(float (&x)[6])
{
    asm("s_nop 1\n\t" NMPC_P6("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
            NMPC_P6("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                NMPC_P6("row_half_mirror row_mask:0xf bank_mask:0xf") NMPC_P6("row_mirror row_mask:0xf bank_mask:0xf")
                    NMPC_P6("row_bcast:15 row_mask:0xa bank_mask:0xf")
                        NMPC_P6("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = read_lane(x[i], 63);
}
#undef NMPC_P6
__device__ __forceinline__ void wave_sum6(double (&x)[6])
{
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = wave_sum(x[i]);
}
__device__ __forceinline__ void wave_sum2(double x, double y, double& sx, double& sy)
{
    sx = wave_sum(x);
    sy = wave_sum(y);
}
__device__ __forceinline__ void wave_sum3(double x, double y, double z, double& sx, double& sy, double& sz)
{
    sx = wave_sum(x);
    sy = wave_sum(y);
    sz = wave_sum(z);
}

// Inclusive prefix sum over lanes 0..63 (lane i gets x_0 + ... + x_i).
__device__ __forceinline__ float wave_scan_incl(float x)
{
    x += dpp_mov<DPP_ROW_SHR0 + 1, 0xf, 0xf, true>(0.0f, x);
    x += dpp_mov<DPP_ROW_SHR0 + 2, 0xf, 0xf, true>(0.0f, x);
    x += dpp_mov<DPP_ROW_SHR0 + 4, 0xf, 0xf, true>(0.0f, x);
    x += dpp_mov<DPP_ROW_SHR0 + 8, 0xf, 0xf, true>(0.0f, x);
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(x));
    return x;
}
// two prefix sums with interleaved chains
__device__ __forceinline__ void wave_scan_incl2(float& x, float& y)
{
    x += dpp_mov<DPP_ROW_SHR0 + 1, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<DPP_ROW_SHR0 + 1, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<DPP_ROW_SHR0 + 2, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<DPP_ROW_SHR0 + 2, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<DPP_ROW_SHR0 + 4, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<DPP_ROW_SHR0 + 4, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<DPP_ROW_SHR0 + 8, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<DPP_ROW_SHR0 + 8, 0xf, 0xf, true>(0.0f, y);
    asm("s_nop 0\n\t" NMPC_P2("row_bcast:15 row_mask:0xa bank_mask:0xf") NMPC_P2("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(x), "+v"(y));
}
__device__ __forceinline__ double wave_scan_incl(double x)
{
    x += dpp_mov<DPP_ROW_SHR0 + 1>(0.0, x);
    x += dpp_mov<DPP_ROW_SHR0 + 2>(0.0, x);
    x += dpp_mov<DPP_ROW_SHR0 + 4>(0.0, x);
    x += dpp_mov<DPP_ROW_SHR0 + 8>(0.0, x);
    x += dpp_mov<DPP_ROW_BCAST15, 0xa>(0.0, x);
    x += dpp_mov<DPP_ROW_BCAST31, 0xc>(0.0, x);
    return x;
}
__device__ __forceinline__ void wave_scan_incl2(double& x, double& y)
{
    x = wave_scan_incl(x);
    y = wave_scan_incl(y);
}

// Inclusive suffix sum (lane i gets x_i + ... + x_63): in-row suffix with row_shl, then the totals of the rows
// above (they sit in the first lane of each row) are added from scalar registers. No LDS crossbar involved.
template <typename T>
__device__ __forceinline__ T wave_suffix_fix_rows(T x)
{
    const T t1 = read_lane(x, 16), t2 = read_lane(x, 32), t3 = read_lane(x, 48);
    int l = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(l)); // (opaque: keeps the three row masks from being hoisted into long-lived SGPR pairs)
    const int row = l >> 4;
    // The three candidates as finished per-lane values before the selection: left to itself the compiler turns the
    // chain of conditions into divergent branches (exec masking, ~20 instructions) to avoid computing sums a row does
    // not need; as plain selects it is three compares and three v_cndmask.
    T c3 = t3, c23 = t2 + t3, c123 = t1 + c23;
    asm volatile("" : "+v"(c3), "+v"(c23), "+v"(c123));
    T add = row == 2 ? c3 : T(0);
    add = row == 1 ? c23 : add;
    add = row == 0 ? c123 : add;
    return x + add;
}
__device__ __forceinline__ float wave_scan_suffix_incl(float x)
{
    x += dpp_mov<0x100 + 1, 0xf, 0xf, true>(0.0f, x); // row_shl:n = 0x100 + n
    x += dpp_mov<0x100 + 2, 0xf, 0xf, true>(0.0f, x);
    x += dpp_mov<0x100 + 4, 0xf, 0xf, true>(0.0f, x);
    x += dpp_mov<0x100 + 8, 0xf, 0xf, true>(0.0f, x);
    return wave_suffix_fix_rows(x);
}
__device__ __forceinline__ void wave_scan_suffix_incl2(float& x, float& y)
{
    x += dpp_mov<0x100 + 1, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<0x100 + 1, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<0x100 + 2, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<0x100 + 2, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<0x100 + 4, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<0x100 + 4, 0xf, 0xf, true>(0.0f, y);
    x += dpp_mov<0x100 + 8, 0xf, 0xf, true>(0.0f, x);
    y += dpp_mov<0x100 + 8, 0xf, 0xf, true>(0.0f, y);
    x = wave_suffix_fix_rows(x);
    y = wave_suffix_fix_rows(y);
}
__device__ __forceinline__ double wave_scan_suffix_incl(double x)
{
    x += dpp_mov<0x100 + 1, 0xf, 0xf, true>(0.0, x); // row_shl:n = 0x100 + n
    x += dpp_mov<0x100 + 2, 0xf, 0xf, true>(0.0, x);
    x += dpp_mov<0x100 + 4, 0xf, 0xf, true>(0.0, x);
    x += dpp_mov<0x100 + 8, 0xf, 0xf, true>(0.0, x);
    return wave_suffix_fix_rows(x);
}
__device__ __forceinline__ void wave_scan_suffix_incl2(double& x, double& y)
{
    x = wave_scan_suffix_incl(x);
    y = wave_scan_suffix_incl(y);
}
// Stride-3 all-reduce of two values at once: every lane gets the sums of x and of y over the lanes of its own residue
// class lane % 3 (three lanes per horizon step: the class = the obstacle row a lane owns in a slot of the register
// table, the sum runs over the horizon steps). In-row partial sums with three DPP steps (row_shr 3 / 6 / 12: the last
// lane of each class in a 16-lane row ends up with the class's sum over that row), then each lane fetches the four row
// partials of its class through the LDS crossbar (ds_bpermute, no memory touched) -- 10 instructions per value for all
// three classes, against three masked full-wave reductions. `a[q]` = 4 * (last lane of this lane's class in row q),
// see class3_addresses().
__device__ __forceinline__ void class3_sum2(float& x, float& y, const int (&a)[4])
{
    asm("s_nop 0\n\t" NMPC_P2("row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0")
            NMPC_P2("row_shr:6 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                NMPC_P2("row_shr:12 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        : "+v"(x), "+v"(y));
    const float x0 = bperm(x, a[0]), x1 = bperm(x, a[1]), x2 = bperm(x, a[2]), x3 = bperm(x, a[3]);
    const float y0 = bperm(y, a[0]), y1 = bperm(y, a[1]), y2 = bperm(y, a[2]), y3 = bperm(y, a[3]);
    x = (x0 + x1) + (x2 + x3);
    y = (y0 + y1) + (y2 + y3);
}
// One value (the register-table passes gate their two slots separately since round 4): the same DPP steps and the same
// association of the four row partials as class3_sum2 -- a slot's E_j comes out bit-identical either way.
template <typename T>
__device__ __forceinline__ void class3_sum1(T& x, const int (&a)[4])
{
    x += dpp_mov<DPP_ROW_SHR0 + 3, 0xf, 0xf, true>(T(0), x);
    x += dpp_mov<DPP_ROW_SHR0 + 6, 0xf, 0xf, true>(T(0), x);
    x += dpp_mov<DPP_ROW_SHR0 + 12, 0xf, 0xf, true>(T(0), x);
    const T x0 = bperm(x, a[0]), x1 = bperm(x, a[1]), x2 = bperm(x, a[2]), x3 = bperm(x, a[3]);
    x = (x0 + x1) + (x2 + x3);
}
template <typename T>
struct Class3Pending1 {
    T x[4];
};
template <typename T>
__device__ __forceinline__ void class3_issue1(T x, const int (&a)[4], Class3Pending1<T>& p)
{
    x += dpp_mov<DPP_ROW_SHR0 + 3, 0xf, 0xf, true>(T(0), x);
    x += dpp_mov<DPP_ROW_SHR0 + 6, 0xf, 0xf, true>(T(0), x);
    x += dpp_mov<DPP_ROW_SHR0 + 12, 0xf, 0xf, true>(T(0), x);
#pragma unroll
    for (int q = 0; q < 4; ++q) p.x[q] = bperm(x, a[q]);
}
template <typename T>
__device__ __forceinline__ T class3_finish1(const Class3Pending1<T>& p)
{
    return (p.x[0] + p.x[1]) + (p.x[2] + p.x[3]);
}
// The same in two halves, so that independent work can sit between the crossbar requests and their results (the
// round trip is ~130 cycles; a wavefront that waits for it does not issue, and with two wavefronts per SIMD nobody else
// takes its slots): class3_issue() leaves the eight row partials in flight, class3_finish() adds them up.
template <typename T>
struct Class3Pending {
    T x[4], y[4];
};
__device__ __forceinline__ void class3_issue(float x, float y, const int (&a)[4], Class3Pending<float>& p)
{
    // (builtins here, not an asm block: with full row / bank masks the compiler folds each step into one v_add_f32_dpp
    //  and fills the DPP wait states with the caller's independent instructions instead of s_nop)
    auto rows = [](float v) {
        v += dpp_mov<DPP_ROW_SHR0 + 3, 0xf, 0xf, true>(0.0f, v);
        v += dpp_mov<DPP_ROW_SHR0 + 6, 0xf, 0xf, true>(0.0f, v);
        v += dpp_mov<DPP_ROW_SHR0 + 12, 0xf, 0xf, true>(0.0f, v);
        return v;
    };
    x = rows(x);
    y = rows(y);
#pragma unroll
    for (int q = 0; q < 4; ++q) p.x[q] = bperm(x, a[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) p.y[q] = bperm(y, a[q]);
}
__device__ __forceinline__ void class3_issue(double x, double y, const int (&a)[4], Class3Pending<double>& p)
{
    auto rows = [](double v) {
        v += dpp_mov<DPP_ROW_SHR0 + 3, 0xf, 0xf, true>(0.0, v);
        v += dpp_mov<DPP_ROW_SHR0 + 6, 0xf, 0xf, true>(0.0, v);
        v += dpp_mov<DPP_ROW_SHR0 + 12, 0xf, 0xf, true>(0.0, v);
        return v;
    };
    x = rows(x);
    y = rows(y);
#pragma unroll
    for (int q = 0; q < 4; ++q) p.x[q] = bperm(x, a[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) p.y[q] = bperm(y, a[q]);
}
template <typename T>
__device__ __forceinline__ void class3_finish(const Class3Pending<T>& p, T& x, T& y)
{
    x = (p.x[0] + p.x[1]) + (p.x[2] + p.x[3]);
    y = (p.y[0] + p.y[1]) + (p.y[2] + p.y[3]);
}
__device__ __forceinline__ void class3_sum2(double& x, double& y, const int (&a)[4])
{
    auto rows = [](double v) {
        v += dpp_mov<DPP_ROW_SHR0 + 3, 0xf, 0xf, true>(0.0, v);
        v += dpp_mov<DPP_ROW_SHR0 + 6, 0xf, 0xf, true>(0.0, v);
        v += dpp_mov<DPP_ROW_SHR0 + 12, 0xf, 0xf, true>(0.0, v);
        return v;
    };
    x = rows(x);
    y = rows(y);
    x = (bperm(x, a[0]) + bperm(x, a[1])) + (bperm(x, a[2]) + bperm(x, a[3]));
    y = (bperm(y, a[0]) + bperm(y, a[1])) + (bperm(y, a[2]) + bperm(y, a[3]));
}
// byte addresses (ds_bpermute) of the last lane of residue class r = lane % 3 in each of the four 16-lane rows:
// rows start at lanes 0, 16, 32, 48 = classes 0, 1, 2, 0, so the last lanes are 15/13/14, 30/31/29, 45/46/47, 63/61/62.
// Three byte lookups in packed constants (a shift and three bit-field extracts) + one OR: cheap enough to be redone
// where the addresses are needed instead of occupying four registers for a whole evaluation.
__device__ __forceinline__ void class3_addresses(int r, int (&a)[4])
{
    const unsigned sh = (unsigned)r << 3;
    a[0] = (int)((0x38343Cu >> sh) & 0xffu); // 4 * {15, 13, 14}
    a[1] = (int)((0x747C78u >> sh) & 0xffu); // 4 * {30, 31, 29}
    a[2] = (int)((0xBCB8B4u >> sh) & 0xffu); // 4 * {45, 46, 47}
    a[3] = a[0] | 0xC0;                      // 4 * {63, 61, 62} = a[0] + 4 * 48
}
#undef NMPC_P2
#undef NMPC_P3

// lane i <- lane 63-i
template <typename T>
__device__ __forceinline__ T wave_reverse(T x)
{
    return __shfl(x, 63 - (int)(threadIdx.x & 63), 64);
}

// lane i <- lane i-S (zeros shifted in at the bottom)
template <int S, typename T>
__device__ __forceinline__ T wave_shift_up(T x)
{
#pragma unroll
    for (int s = 0; s < S; ++s) x = dpp_mov<DPP_WAVE_SHR1, 0xf, 0xf, true>(T(0), x);
    return x;
}
// lane i <- lane i+S (zeros shifted in at the top)
template <int S, typename T>
__device__ __forceinline__ T wave_shift_down(T x)
{
#pragma unroll
    for (int s = 0; s < S; ++s) x = dpp_mov<DPP_WAVE_SHL1, 0xf, 0xf, true>(T(0), x);
    return x;
}

// ---- shuffle-based reference versions (used only by the device self-test) ---------------------------------
template <typename T>
__device__ inline T ref_wave_sum(T x)
{
    T s = 0;
    for (int i = 0; i < 64; ++i) s += __shfl(x, i, 64);
    return s;
}
template <typename T>
__device__ inline T ref_class3_sum(T x)
{
    const int lane = threadIdx.x & 63;
    T s = 0;
    for (int i = 0; i < 64; ++i) {
        T v = __shfl(x, i, 64);
        if (i % 3 == lane % 3) s += v;
    }
    return s;
}
template <typename T>
__device__ inline T ref_wave_scan_incl(T x)
{
    const int lane = threadIdx.x & 63;
    T s = 0;
    for (int i = 0; i < 64; ++i) {
        T v = __shfl(x, i, 64);
        if (i <= lane) s += v;
    }
    return s;
}
template <typename T>
__device__ inline T ref_wave_scan_suffix_incl(T x)
{
    const int lane = threadIdx.x & 63;
    T s = 0;
    for (int i = 63; i >= 0; --i) {
        T v = __shfl(x, i, 64);
        if (i >= lane) s += v;
    }
    return s;
}

} // namespace nmpc
