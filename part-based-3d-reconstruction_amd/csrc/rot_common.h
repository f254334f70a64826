// Shared by the generic-angle kernels (rotate.hip) and the bit-sliced chain (sliced.hip): SciPy's coordinate / weight
// arithmetic of one output cell and the 16-entry result table it collapses to on 0/1 data.
//
// Bit-exact restatement of scipy.ndimage.affine_transform(order=1, mode="constant", cval=0) as the
// reference calls it (reference utils/voxel_carving_utils.py:116-123), for matrices whose row 1 is
// [+-0, 1, +-0] with off[1] == 0 (every Rinv(angle) the reference can produce).  Then cc1 == y
// exactly, the y-weights are {1, 0}, and per output voxel the arithmetic is
//     cc_h = (((0 + x*M[h][0]) + y*M[h][1]) + z*M[h][2]) + off[h]       h = 0, 2   (IEEE, NO fma)
//     outside [0, n_h - 1]  -> 0
//     t = cc - floor(cc);  w0 = 1 - t;  w1 = 1 - w0
//     acc = ((v00*wx0)*wz0 + (v01*wx0)*wz1) + (v10*wx1)*wz0 + (v11*wx1)*wz1    (left to right)
//     out = acc > 0 ? (uint8)min(acc + 0.5, 255) : 0
// All doubles are evaluated with __dmul_rn/__dadd_rn so that no contraction can occur; the files are
// also compiled with -ffp-contract=off.
#pragma once
#include "pb3d_internal.h"

namespace {

struct RotParams {
    double m00, m01, m02, off0;
    double m20, m21, m22, off2;
};

struct Cell {
    double wx0, wx1, wz0, wz1;
    int s0, s2;  // floor of the source coordinate; s0 < 0 marks "outside -> 0"
};

__device__ __forceinline__ double coord(double x, double z, double ma, double mb, double mc, double off) {
    // (((0 + x*ma) + y*mb) + z*mc) + off with y*mb == +-0 for every y >= 0 (mb is +-0)
    double c = __dadd_rn(0.0, __dmul_rn(x, ma));
    c = __dadd_rn(c, __dmul_rn(0.0, mb));
    c = __dadd_rn(c, __dmul_rn(z, mc));
    return __dadd_rn(c, off);
}

__device__ __forceinline__ Cell make_cell(const RotParams& p, i64 x, i64 z, i64 W, i64 D) {
    Cell c;
    const double cc0 = coord((double)x, (double)z, p.m00, p.m01, p.m02, p.off0);
    const double cc2 = coord((double)x, (double)z, p.m20, p.m21, p.m22, p.off2);
    if (cc0 < 0.0 || cc0 > (double)(W - 1) || cc2 < 0.0 || cc2 > (double)(D - 1)) {
        c.s0 = -1; c.s2 = 0; c.wx0 = c.wx1 = c.wz0 = c.wz1 = 0.0;
        return c;
    }
    const double f0 = floor(cc0), f2 = floor(cc2);
    c.s0 = (int)f0; c.s2 = (int)f2;
    c.wx0 = __dsub_rn(1.0, __dsub_rn(cc0, f0));
    c.wx1 = __dsub_rn(1.0, c.wx0);
    c.wz0 = __dsub_rn(1.0, __dsub_rn(cc2, f2));
    c.wz1 = __dsub_rn(1.0, c.wz0);
    return c;
}

// Bit b of the table is SciPy's result for the tap pattern b = v00 | v01<<1 | v10<<2 | v11<<3 on 0/1 data (partial sums formed
// in SciPy's tap order from the exact products w_x*w_z; 1.0*w == w and a zero tap adds +0.0, so this IS the reference arithmetic).
// A tap whose weight is exactly 0 does not influence the table.
__device__ __forceinline__ u32 lut_of(const Cell& c) {
    const double p00 = __dmul_rn(c.wx0, c.wz0), p01 = __dmul_rn(c.wx0, c.wz1), p10 = __dmul_rn(c.wx1, c.wz0),
                 p11 = __dmul_rn(c.wx1, c.wz1);
    u32 lut = 0;
#pragma unroll
    for (int b = 1; b < 16; ++b) {
        double acc = 0.0;
        if (b & 1) acc = __dadd_rn(acc, p00);
        if (b & 2) acc = __dadd_rn(acc, p01);
        if (b & 4) acc = __dadd_rn(acc, p10);
        if (b & 8) acc = __dadd_rn(acc, p11);
        // uint8 store rule: acc > 0 ? trunc(acc + 0.5) : 0 ; the weights sum to ~1 so the value is 0 or 1
        if (acc > 0.0 && __dadd_rn(acc, 0.5) >= 1.0) lut |= 1u << b;
    }
    return lut;
}

__device__ __forceinline__ u32 bsel(u32 sel, u32 a, u32 b) { return (sel & a) | (~sel & b); }   // v_bfi_b32

// the table applied to 32 bit-sliced planes at once: a 4-level multiplexer tree of bitwise selects.  Written as the instructions
// themselves (16 v_bfe_i32 + 15 v_bfi_b32 per cell): from the C form the compiler builds compare / cndmask / and / or chains with
// the table bits as lane masks in scalar registers -- three times the instructions (and 460 spilled SGPRs when the table is loop-invariant).
__device__ __forceinline__ u32 sbit(u32 x, int k) { u32 r; asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(x), "n"(k)); return r; }   // bit k -> 0 / ~0
__device__ __forceinline__ u32 vbfi(u32 s, u32 a, u32 b) { u32 r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(s), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ u32 lut_apply32(u32 lut, u32 t00, u32 t01, u32 t10, u32 t11) {
    u32 g[8], h[4];
    g[0] = vbfi(t00, sbit(lut, 1), sbit(lut, 0));   g[1] = vbfi(t00, sbit(lut, 3), sbit(lut, 2));
    g[2] = vbfi(t00, sbit(lut, 5), sbit(lut, 4));   g[3] = vbfi(t00, sbit(lut, 7), sbit(lut, 6));
    g[4] = vbfi(t00, sbit(lut, 9), sbit(lut, 8));   g[5] = vbfi(t00, sbit(lut, 11), sbit(lut, 10));
    g[6] = vbfi(t00, sbit(lut, 13), sbit(lut, 12)); g[7] = vbfi(t00, sbit(lut, 15), sbit(lut, 14));
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = vbfi(t01, g[2 * j + 1], g[2 * j]);
    const u32 m0 = vbfi(t10, h[1], h[0]), m1 = vbfi(t10, h[3], h[2]);
    return vbfi(t11, m1, m0);
}

__device__ __forceinline__ u32 pperm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// a[g].byte v  ->  v[v].byte g   (4 x 4 byte transpose)
__device__ __forceinline__ void tr4x4(u32 a0, u32 a1, u32 a2, u32 a3, u32 v[4]) {
    const u32 l01 = pperm(a1, a0, 0x05010400u), h01 = pperm(a1, a0, 0x07030602u);
    const u32 l23 = pperm(a3, a2, 0x05010400u), h23 = pperm(a3, a2, 0x07030602u);
    v[0] = pperm(l23, l01, 0x05040100u); v[1] = pperm(l23, l01, 0x07060302u);
    v[2] = pperm(h23, h01, 0x05040100u); v[3] = pperm(h23, h01, 0x07060302u);
}

}  // namespace
