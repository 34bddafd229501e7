// sr_dtw_dev.h -- get_dis (DTW.C:45-62) and dtw_limit (DTW.C:76-109) as device functions, feature rows in registers; shared by the DTW kernels and the scalar diagnostics.
// gfx950 (MI355X, CDNA4) only; wave = 64 lanes; no MFMA (the path has no dense contraction), integer VALU + LDS.
// Every kernel reproduces the reference's integer arithmetic bit for bit; cited lines are relative to the reference tree.
#pragma once
#include "sr_dev.h"

namespace sr {

struct Frame12 {
    uint32_t w[6];
};
__device__ __forceinline__ Frame12 load_frame(const int16_t *p)
{
    const uint2 *q = (const uint2 *)p;  // rows are 24 bytes, 8-byte aligned
    const uint2 a = q[0], b = q[1], c = q[2];
    Frame12 f;
    f.w[0] = a.x;
    f.w[1] = a.y;
    f.w[2] = b.x;
    f.w[3] = b.y;
    f.w[4] = c.x;
    f.w[5] = c.y;
    return f;
}
__device__ __forceinline__ uint32_t norm2(const Frame12 &f)
{
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) acc = sdot2(f.w[i], f.w[i], acc);
    return (uint32_t)acc;
}
// get_dis (DTW.C:45-62): sum (a-b)^2 in u32 wrap = |a|^2 + |b|^2 - 2 a.b in the same ring
__device__ __forceinline__ uint32_t get_dis_dev(const Frame12 &fa, uint32_t na, const Frame12 &fb, uint32_t nb)
{
    int dot = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) dot = sdot2(fa.w[i], fb.w[i], dot);
    const uint32_t d = na + nb - 2u * (uint32_t)dot;
    return cvt_u32(sqrt_rn_int((float)d));
}
// dtw_limit (DTW.C:76-109); returns true when (x, y) is OUTSIDE the relaxed parallelogram
__device__ __forceinline__ bool dtw_out(int x, int y, int X1, int X2, int in_n, int mdl_n)
{
    const bool o1 = (x < X1) ? (y >= 2 * x + 2) : (2 * y + in_n - 2 * mdl_n >= x + 4);
    const bool o2 = (x < X2) ? (2 * y + 2 <= x) : (y + 4 <= 2 * x + mdl_n - 2 * in_n);
    return o1 || o2;
}

__device__ __forceinline__ uint32_t dis_from(uint32_t na, uint32_t nb, int dot)
{
    const uint32_t d = na + nb - 2u * (uint32_t)dot;
    return cvt_u32(sqrt_rn_int((float)d));
}

// floor(sqrtf((float)d)) from BELOW: v_sqrt_f32 is within 1 ulp of the correctly rounded root r, so pred(s0) <= r and
// floor(pred(s0)) is g = floor(r) or g - 1 (2 ulp < 1 for every u32 input) -- never above.  k_dtw_lds takes this value
// and learns from its own tie threshold T(mn) (the first squared distance whose root is mn + 1, exact by table) whether it
// is one short: d >= T(mn) <=> g = mn + 1.  Validated for all 2^32 inputs by tests/exhaustive_math_sweep.py through
// sr_math_diag (the diagnostic poisons its output word if the value is ever anything but g or g - 1).
__device__ __forceinline__ uint32_t sqrt_floor_low(uint32_t d)
{
    const float s0 = __builtin_amdgcn_sqrtf((float)d);
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(__int_as_float(__float_as_int(s0) - 1)));  // NaN (d = 0) -> 0
    return r;
}
// a - b, saturating at 0
__device__ __forceinline__ uint32_t sub_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// 32-byte feature row: w[0..5] = 12 x s16, w[6] = squared norm (u32 wrap), w[7] unused
struct Row32 {
    uint32_t w[8];
};
__device__ __forceinline__ Row32 row_from(const u32x4 lo, const u32x4 hi)
{
    Row32 r;
    r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
    r.w[4] = hi.x; r.w[5] = hi.y; r.w[6] = hi.z; r.w[7] = hi.w;
    return r;
}
__device__ __forceinline__ Row32 row_from2(const u32x2 a, const u32x2 b, const u32x2 c, uint32_t nrm)
{
    Row32 r;
    r.w[0] = a.x; r.w[1] = a.y; r.w[2] = b.x; r.w[3] = b.y; r.w[4] = c.x; r.w[5] = c.y;
    r.w[6] = nrm; r.w[7] = 0;
    return r;
}
// dst = src as four 64-bit moves (v_pk_mov_b32 moves a register pair per issue slot)
__device__ __forceinline__ void copy_row(Row32 &dst, const Row32 &src)
{
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        u32x2 d;
        const u32x2 v = {src.w[i], src.w[i + 1]};
        asm("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(d) : "v"(v));
        dst.w[i] = d.x;
        dst.w[i + 1] = d.y;
    }
}
__device__ __forceinline__ int dot_rows(const Row32 &a, const Row32 &b)
{
    int acc = sdot2z(a.w[0], b.w[0]);
#pragma unroll
    for (int i = 1; i < 6; i++) acc = sdot2(a.w[i], b.w[i], acc);
    return acc;
}

// c + a.b over the 12 coefficients: the accumulator input carries the norm sum
__device__ __forceinline__ int dot_rows_acc(const Row32 &a, const Row32 &b, int c)
{
    int acc = sdot2a(a.w[0], b.w[0], c);
#pragma unroll
    for (int i = 1; i < 6; i++) acc = sdot2(a.w[i], b.w[i], acc);
    return acc;
}

}  // namespace sr
