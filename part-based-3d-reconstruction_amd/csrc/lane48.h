// Shared device helpers of the 16-voxels-per-lane sweeps (csrc/carve.hip, csrc/bits90.hip): 16-byte vector types and accessors, the
// wave-private LDS window that turns a lane's 48 contiguous bytes into 1 KB-contiguous wave stores / loads, keep-bit expansion.
#pragma once
#include "pb3d_internal.h"

namespace {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));     // 16-byte access at any byte alignment (odd-sized grids, buffer views)

__device__ __forceinline__ u32x4 ld_nt(const u32x4_u* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt(u32x4_u* p, u32x4 v) { __builtin_nontemporal_store(v, p); }
// RGB side of the 16-voxels-per-lane kernels: a lane owns 48 contiguous bytes, so every 128-byte line is touched by three
// instructions of the wave.  Nontemporal hints make the line leave the L1 between them (measured 1.4x slower on the
// point sweep of project.hip), so these accesses stay plain.
#ifndef PB3D_STRIDED_NT
#define PB3D_STRIDED_NT 0
#endif
__device__ __forceinline__ u32x4 ld_s(const u32x4* p) { return PB3D_STRIDED_NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st_s(u32x4* p, u32x4 v) { if (PB3D_STRIDED_NT) __builtin_nontemporal_store(v, p); else *p = v; }

// RGB stores of those kernels: the wave's 64 groups form 3 KiB contiguous in the output (vector index 3 * gw0 ...), but a lane
// holds vectors 3*lane .. 3*lane+2 of it.  Route them through a wave-private LDS window (192 vectors) so that every store
// instruction writes 1 KiB contiguous (whole 128-byte lines) instead of 16 bytes every 48.  Called by all 64 lanes.
__device__ __forceinline__ void store48_wave(u32x4_u* __restrict__ out, i64 gw0, i64 ngroups, const u32x4 r[3], u32x4* lds_w) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; ++k) lds_w[3 * lane + k] = r[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const i64 left = ngroups - gw0;
    const int nvec = 3 * (int)(left < 64 ? left : 64);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = lane + 64 * k;
        if (idx < nvec) __builtin_nontemporal_store(lds_w[idx], out + 3 * gw0 + idx);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void load48_wave(const u32x4* __restrict__ in, i64 gw0, i64 ngroups, u32x4 r[3], u32x4* lds_w) {
    const int lane = threadIdx.x & 63;
    const i64 left = ngroups - gw0;
    const int nvec = 3 * (int)(left < 64 ? left : 64);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = lane + 64 * k;
        lds_w[idx] = idx < nvec ? ld_nt(in + 3 * gw0 + idx) : (u32x4)(0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = lds_w[3 * lane + k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ u32 byte_of(const u32* w, int idx) { return (w[idx >> 2] >> ((idx & 3) * 8)) & 0xffu; }

__device__ __forceinline__ void expand16(const u32 keep16, const u32 r, const u32 g, const u32 b, u32 w[12]) {
    // byte j of the 48 output bytes belongs to voxel j/3 and channel j%3.  The RGB stream repeats every 3 dwords (RGBR GBRG
    // BRGB); a dword covers parts of two voxels, whose keep masks are joined with a constant byte mask per dword phase.
    const u32 S[3] = {r | (g << 8) | (b << 16) | (r << 24), g | (b << 8) | (r << 16) | (g << 24), b | (r << 8) | (g << 16) | (b << 24)};
    u32 m[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) m[v] = (u32)__builtin_amdgcn_sbfe((int)keep16, v, 1);      // 0 or ~0
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int v0 = (4 * k) / 3, ph = (4 * k) % 3;          // first voxel of the dword and its channel phase
        // ph 0: bytes v0,v0,v0,v0+1 ; ph 1: v0,v0,v0+1,v0+1 ; ph 2: v0,v0+1,v0+1,v0+1
        const u32 lo = ph == 0 ? 0x00ffffffu : (ph == 1 ? 0x0000ffffu : 0x000000ffu);
        w[k] = S[k % 3] & ((m[v0] & lo) | (m[v0 + 1 < 16 ? v0 + 1 : 15] & ~lo));
    }
}

// bit i = (byte i of the 16 bytes == 1)
__device__ __forceinline__ u32 ones16(const u32 cw[4]) {
    u32 bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 t = cw[j] ^ 0x01010101u;                                     // a byte equal to 1 becomes 0
        const u32 z = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu | t) & 0x80808080u;       // bit 7 of a byte set <=> the byte is non-zero
        const u32 e = (~z & 0x80808080u) >> 7;                                  // bit 0 of a byte set <=> the byte was 1
        bits |= ((e * 0x01020408u) >> 24) << (4 * j);                            // gather bits 0, 8, 16, 24 into 4 bits
    }
    return bits;
}

__device__ __forceinline__ u32 spread4(u32 b4) { return ((b4 * 0x00204081u) & 0x01010101u) * 0xffu; }      // bit j -> byte j = 0xff
__device__ __forceinline__ u32 nonzero16(const u32 cw[4]) {      // bit i = (byte i of the 16 bytes != 0)
    u32 bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 t = cw[j];
        const u32 z = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;
        bits |= (((z >> 7) * 0x01020408u) >> 24) << (4 * j);
    }
    return bits;
}

}  // namespace
