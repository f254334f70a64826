// Seeded synthetic inputs generated directly in HBM: no multi-GiB H2D copies.
// Every formula is integer-only so that tests can regenerate the same bytes on the host in NumPy.
#include "pb3d_internal.h"

namespace {

__host__ __device__ inline u64 splitmix64(u64 z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

struct Palette16 {
    u8 rgb[48];
};

Palette16 make_palette() {
    // label 0 = background, 1..9 = the nine non-background part colours of reference utils/config.py:29-40
    // (in that file's order), 10..15 = (16k, 255-16k, 8k+7).
    static const u8 base[10][3] = {{216, 224, 251}, {253, 248, 96}, {1, 220, 5},    {63, 138, 173},  {190, 0, 255},
                                   {0, 0, 255},     {5, 223, 223},  {255, 180, 80}, {180, 140, 255}, {255, 120, 230}};
    Palette16 p;
    for (int k = 0; k < 10; ++k)
        for (int c = 0; c < 3; ++c) p.rgb[3 * k + c] = base[k][c];
    for (int k = 10; k < 16; ++k) {
        p.rgb[3 * k] = (u8)(16 * k);
        p.rgb[3 * k + 1] = (u8)(255 - 16 * k);
        p.rgb[3 * k + 2] = (u8)(8 * k + 7);
    }
    return p;
}

// label of pixel (row y, column x) of the S x S synthetic part mask (formula stated for S = 1024 and
// applied to xn = x*1024/S, yn = y*1024/S).
__device__ inline int mask16_label(i64 x, i64 y, i64 S) {
    const i64 xn = (x * 1024) / S, yn = (y * 1024) / S;
    const i64 dx2 = 2 * xn - 1023;                 // 2*(x - 511.5)
    const i64 adx2 = dx2 < 0 ? -dx2 : dx2;
    const bool body = adx2 < 840 && yn >= 256;     // |x-511.5| < 420, y >= 256
    const i64 dy2 = 2 * (yn - 256);
    const bool dome = dx2 * dx2 * 40000 + dy2 * dy2 * 90000 < 4ll * 90000 * 40000;  // ellipse 300 x 200
    const bool towers = adx2 > 880 && adx2 < 960 && yn >= 96;                       // |x-511.5| in (440,480)
    if (!(body || dome || towers)) return 0;
    return 1 + (int)(((xn >> 6) + 3 * (yn >> 7)) % 15);
}

__global__ __launch_bounds__(256) void k_synth_mask16(i64 S, Palette16 pal, u8* __restrict__ label_hw, u8* __restrict__ bin_hw,
                                                      u8* __restrict__ rgb_hw3, u8* __restrict__ bin_wh) {
    const i64 n = S * S;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 y = i / S, x = i - y * S;
        const int lab = mask16_label(x, y, S);
        if (label_hw) label_hw[i] = (u8)lab;
        if (bin_hw) bin_hw[i] = lab ? 1 : 0;
        if (bin_wh) bin_wh[x * S + y] = lab ? 1 : 0;
        if (rgb_hw3) {
            rgb_hw3[3 * i] = pal.rgb[3 * lab];
            rgb_hw3[3 * i + 1] = pal.rgb[3 * lab + 1];
            rgb_hw3[3 * i + 2] = pal.rgb[3 * lab + 2];
        }
    }
}

// 4 voxels (12 bytes = 3 dwords) per thread; voxel v of the FULL grid gets palette[splitmix64(seed ^ v) & 15]
__global__ __launch_bounds__(256) void k_synth_sem(i64 v_begin, i64 nvox, u64 seed, Palette16 pal, u8* __restrict__ out) {
    const i64 ngroups = (nvox + 3) / 4;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        u8 b[12];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lab = (int)(splitmix64(seed ^ (u64)(v_begin + 4 * g + q)) & 15);
            b[3 * q] = pal.rgb[3 * lab]; b[3 * q + 1] = pal.rgb[3 * lab + 1]; b[3 * q + 2] = pal.rgb[3 * lab + 2];
        }
        if (4 * g + 4 <= nvox) {
            u32* o = (u32*)(out + 12 * g);
#pragma unroll
            for (int k = 0; k < 3; ++k) o[k] = b[4 * k] | (b[4 * k + 1] << 8) | (b[4 * k + 2] << 16) | ((u32)b[4 * k + 3] << 24);
        } else {
            for (i64 q = 0; 4 * g + q < nvox; ++q)
                for (int c = 0; c < 3; ++c) out[12 * g + 3 * q + c] = b[3 * q + c];
        }
    }
}

__global__ __launch_bounds__(256) void k_synth_occ(i64 v_begin, i64 nvox, u64 seed, u8* __restrict__ out) {
    for (i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x; v < nvox; v += (i64)gridDim.x * blockDim.x)
        out[v] = (u8)(splitmix64(seed ^ (u64)(v_begin + v)) & 1);
}

}  // namespace

extern "C" {

int pb3d_synth_palette16(uint8_t palette[48]) {
    PB3D_REQUIRE(palette != nullptr, "pb3d_synth_palette16: null output");
    const Palette16 p = make_palette();
    memcpy(palette, p.rgb, 48);
    return PB3D_OK;
}

int pb3d_synth_mask16_dev(pb3d_ctx* ctx, int64_t S, uint8_t* d_label_hw, uint8_t* d_binary_hw, uint8_t* d_rgb_hw3,
                          uint8_t* d_binary_wh) {
    PB3D_REQUIRE(ctx != nullptr && S > 0 && S <= (1 << 20), "pb3d_synth_mask16: bad size");
    hipLaunchKernelGGL(k_synth_mask16, dim3(pb3d_stream_blocks(ctx, S * S, 256, 8)), dim3(256), 0, ctx->stream, S, make_palette(),
                       d_label_hw, d_binary_hw, d_rgb_hw3, d_binary_wh);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_synth_sem_dev(pb3d_ctx* ctx, int64_t x0, int64_t x1, int64_t H, int64_t D, uint64_t seed, uint8_t* d_slab_rgb) {
    PB3D_REQUIRE(ctx != nullptr && x0 >= 0 && x1 >= x0 && H >= 0 && D >= 0, "pb3d_synth_sem: bad slab");
    const i64 nvox = (x1 - x0) * H * D;
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_slab_rgb != nullptr && ((uintptr_t)d_slab_rgb & 3u) == 0, "pb3d_synth_sem: null or unaligned buffer");
    hipLaunchKernelGGL(k_synth_sem, dim3(pb3d_stream_blocks(ctx, (nvox + 3) / 4, 256, 8)), dim3(256), 0, ctx->stream, x0 * H * D,
                       nvox, seed, make_palette(), d_slab_rgb);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_synth_occ_dev(pb3d_ctx* ctx, int64_t x0, int64_t x1, int64_t H, int64_t D, uint64_t seed, uint8_t* d_slab) {
    PB3D_REQUIRE(ctx != nullptr && x0 >= 0 && x1 >= x0 && H >= 0 && D >= 0, "pb3d_synth_occ: bad slab");
    const i64 nvox = (x1 - x0) * H * D;
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_slab != nullptr, "pb3d_synth_occ: null buffer");
    hipLaunchKernelGGL(k_synth_occ, dim3(pb3d_stream_blocks(ctx, nvox, 256, 8)), dim3(256), 0, ctx->stream, x0 * H * D, nvox, seed,
                       d_slab);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

}  // extern "C"
