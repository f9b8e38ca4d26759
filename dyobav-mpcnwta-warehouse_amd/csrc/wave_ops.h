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

// ---- f32: x += dpp(x) as ONE VALU instruction (v_add_f32_dpp). hipcc lowers update_dpp + add to
//      v_mov_b32 (old) + s_nop + v_mov_b32_dpp + v_add (3-4 issue slots per step); the fused form is one.
//      "s_nop 1" covers the 2 wait states a DPP read needs after a VALU write of the same VGPR.
#define NMPC_DPP_ADD(x, ctrl) asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " ctrl : "+v"(x))

// Sum over the 64 lanes; the result is wave-uniform (read back from lane 63 into scalar registers).
__device__ __forceinline__ float wave_sum(float x)
{
    NMPC_DPP_ADD(x, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    NMPC_DPP_ADD(x, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    NMPC_DPP_ADD(x, "row_half_mirror row_mask:0xf bank_mask:0xf");
    NMPC_DPP_ADD(x, "row_mirror row_mask:0xf bank_mask:0xf");        // every lane of a row holds the row sum
    NMPC_DPP_ADD(x, "row_bcast:15 row_mask:0xa bank_mask:0xf");      // rows 1,3 += rows 0,2
    NMPC_DPP_ADD(x, "row_bcast:31 row_mask:0xc bank_mask:0xf");      // rows 2,3 += rows 0+1
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

// Inclusive prefix sum over lanes 0..63 (lane i gets x_0 + ... + x_i).
__device__ __forceinline__ float wave_scan_incl(float x)
{
    NMPC_DPP_ADD(x, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    NMPC_DPP_ADD(x, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    NMPC_DPP_ADD(x, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    NMPC_DPP_ADD(x, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    NMPC_DPP_ADD(x, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    NMPC_DPP_ADD(x, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    return x;
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

// lane i <- lane 63-i
template <typename T>
__device__ __forceinline__ T wave_reverse(T x)
{
    return __shfl(x, 63 - (int)(threadIdx.x & 63), 64);
}

// Inclusive suffix sum (lane i gets x_i + ... + x_63).
template <typename T>
__device__ __forceinline__ T wave_scan_suffix_incl(T x)
{
    return wave_reverse(wave_scan_incl(wave_reverse(x)));
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
