// Row N3, device half: the 1-byte LABEL form of a semantic grid, and the sharded entry points built on it.
//
// A semantic grid only ever holds (0,0,0) and the colours of a small palette (reference utils/config.py:29-43; the masks come
// from reference utils/mask_utils.py:14-87 with the same palette), so a voxel is fully described by one byte:
//     label 0 <-> (0,0,0),   label k <-> palette[k-1]   (1 <= k <= npal <= 254)
// The carve ops act on labels exactly as on colours (zero where the mask drops, the column's own value elsewhere), at a third
// of the traffic; the label volume is also what a multi-GPU run should reassemble (1 B/voxel over xGMI instead of 3) -- every
// rank expands to RGB locally, and only if a consumer needs RGB at all.
#include <vector>

#include "pb3d_internal.h"

namespace {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct LabelHash {
    u32 K;            // odd 24-bit multiplier; entry = bits 31..24 of mul24(colour, K)
    u32 tab[256];     // colour << 8 | label ; empty entries hold a colour that hashes elsewhere (can never compare equal)
};

// colours of the palette + black, perfectly hashed.  false: two DIFFERENT labels share a colour (ambiguous palette).
bool build_hash(const u8* palette, int npal, LabelHash* h) {
    u32 cols[256];
    cols[0] = 0;
    for (int k = 0; k < npal; ++k) cols[k + 1] = (u32)palette[3 * k] | ((u32)palette[3 * k + 1] << 8) | ((u32)palette[3 * k + 2] << 16);
    for (int a = 0; a <= npal; ++a)
        for (int b = a + 1; b <= npal; ++b)
            if (cols[a] == cols[b]) return false;
    u32 K = 0x9e3779u;
    for (int attempt = 0; attempt < (1 << 18); ++attempt) {
        K = (K * 1664525u + 1013904223u) & 0xffffffu;
        const u32 Ko = K | 1u;
        int owner[256];
        for (int e = 0; e < 256; ++e) owner[e] = -1;
        bool ok = true;
        for (int k = 0; k <= npal && ok; ++k) {
            const u32 e = (u32)(((unsigned long long)cols[k] * Ko) & 0xffffffffull) >> 24;
            if (owner[e] >= 0) ok = false; else owner[e] = k;
        }
        if (!ok) continue;
        h->K = Ko;
        for (u32 e = 0; e < 256; ++e) {
            if (owner[e] >= 0) { h->tab[e] = (cols[owner[e]] << 8) | (u32)owner[e]; continue; }
            u32 c = 0;                                   // any colour whose own entry is not e
            while (((u32)(((unsigned long long)c * Ko) & 0xffffffffull) >> 24) == e) ++c;
            h->tab[e] = (c << 8) | 0xffu;
        }
        return true;
    }
    return false;
}

// 16 voxels (48 B of RGB) per thread -> 16 label bytes.  An RGB value outside palette + black raises *unknown (label 255).
__global__ __launch_bounds__(256) void k_rgb_to_label(const u8* __restrict__ rgb, i64 nvox, LabelHash H, u8* __restrict__ label,
                                                      int* __restrict__ unknown) {
    __shared__ u32 htab[256];
    htab[threadIdx.x] = H.tab[threadIdx.x];
    __syncthreads();
    bool bad = false;
    const i64 ngroups = (nvox + 15) / 16;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        const i64 v0 = 16 * g;
        u32 w[12];
        if (v0 + 16 <= nvox) {
            const u32x4* p = (const u32x4*)(rgb + 3 * v0);
#pragma unroll
            for (int k = 0; k < 3; ++k) { const u32x4 t = p[k]; w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w; }
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                u32 t = 0;
                for (int b = 0; b < 4; ++b) { const i64 o = 3 * v0 + 4 * k + b; if (o < 3 * nvox) t |= (u32)rgb[o] << (8 * b); }
                w[k] = t;
            }
        }
        u32 out[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = (3 * i) >> 2, sh = (3 * i) & 3;
            const u32 v = __builtin_amdgcn_alignbyte(j + 1 < 12 ? w[j + 1] : 0u, w[j], (u32)sh);     // top byte: the next voxel's red
            const u32 t = *(const u32*)((const u8*)htab + ((__umul24(v, H.K) >> 22) & 0x3fcu));      // mul24 ignores the top byte
            const bool hit = (t >> 8) == (v & 0x00ffffffu);
            if (!hit && v0 + i < nvox) bad = true;
            out[i >> 2] |= (hit ? (t & 0xffu) : 0xffu) << (8 * (i & 3));
        }
        if (v0 + 16 <= nvox) { u32x4 r; r.x = out[0]; r.y = out[1]; r.z = out[2]; r.w = out[3]; *(u32x4*)(label + v0) = r; }
        else for (i64 i = 0; v0 + i < nvox; ++i) label[v0 + i] = (u8)(out[i >> 2] >> (8 * (i & 3)));
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(unknown, 1);
}

// 16 labels per thread -> 48 B of RGB (pal[k] = colour of label k, pal[0] = 0; labels above npal give black and raise *unknown)
struct Pal256 { u32 c[256]; };
__global__ __launch_bounds__(256) void k_label_to_rgb(const u8* __restrict__ label, i64 nvox, Pal256 P, int npal, u8* __restrict__ rgb,
                                                      int* __restrict__ unknown) {
    __shared__ u32 pal[256];
    pal[threadIdx.x] = P.c[threadIdx.x];
    __syncthreads();
    bool bad = false;
    const i64 ngroups = (nvox + 15) / 16;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        const i64 v0 = 16 * g;
        u32 l[4] = {0, 0, 0, 0};
        if (v0 + 16 <= nvox) { const u32x4 t = *(const u32x4*)(label + v0); l[0] = t.x; l[1] = t.y; l[2] = t.z; l[3] = t.w; }
        else for (i64 i = 0; v0 + i < nvox; ++i) l[i >> 2] |= (u32)label[v0 + i] << (8 * (i & 3));
        u32 c[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const u32 k = (l[i >> 2] >> (8 * (i & 3))) & 0xffu;
            if ((int)k > npal) bad = true;
            c[i] = pal[k];                               // 0x00bbggrr
        }
        u32 w[12];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // four voxels -> three dwords
            const u32 a = c[4 * q], b = c[4 * q + 1], d = c[4 * q + 2], e = c[4 * q + 3];
            w[3 * q] = a | (b << 24);
            w[3 * q + 1] = (b >> 8) | (d << 16);
            w[3 * q + 2] = (d >> 16) | (e << 8);
        }
        if (v0 + 16 <= nvox) {
            u32x4* o = (u32x4*)(rgb + 3 * v0);
#pragma unroll
            for (int k = 0; k < 3; ++k) { u32x4 r; r.x = w[4 * k]; r.y = w[4 * k + 1]; r.z = w[4 * k + 2]; r.w = w[4 * k + 3]; o[k] = r; }
        } else {
            for (i64 b = 0; 3 * v0 + b < 3 * nvox; ++b) rgb[3 * v0 + b] = (u8)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(unknown, 1);
}

// apply_colored_mask_to_voxel_grid on labels: out[x,y,z] = carved[x,y,z] == 1 ? label_hw[y,x] : 0
__global__ __launch_bounds__(256) void k_label_apply(const u8* __restrict__ carved, const u8* __restrict__ label_hw, i64 W, i64 H, i64 D,
                                                     u8* __restrict__ out, i64 v_first = 0) {
    const i64 nvox = W * H * D;
    for (i64 i = v_first + (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (i64)gridDim.x * blockDim.x) {
        const i64 col = i / D, x = col / H, y = col - x * H;
        out[i] = carved[i] == 1 ? label_hw[y * W + x] : (u8)0;
    }
}


// the same, 16 voxels of the flat stream per thread (fewer than 2^32 voxels, D >= 16, 16-byte aligned output): a group lies in one
// (x,y) column or straddles two; exact u32 divisions (the per-voxel form spends two 64-bit divisions per BYTE)
__global__ __launch_bounds__(256) void k_label_apply16(const u8* __restrict__ carved, const u8* __restrict__ label_hw, i64 W, i64 H, i64 ngroups,
                                                       pb3d_magic mD, pb3d_magic mH, u8* __restrict__ out) {
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        const u32 v0 = (u32)(16 * g);
        const u32 col = pb3d_div(v0, mD), x = pb3d_div(col, mH), y = col - x * mH.d;
        const u32 bnd = (col + 1) * mD.d - v0;                      // voxels of the group in column `col` (>= 1)
        const u32 L0 = label_hw[(i64)y * W + x];
        u32 L1 = L0;
        if (bnd < 16u) { const u32 x1 = y + 1 < mH.d ? x : x + 1, y1 = y + 1 < mH.d ? y + 1 : 0u; L1 = label_hw[(i64)y1 * W + x1]; }
        const u32x4 cv = *(const u32x4*)(carved + 16 * g);
        const u32 cw[4] = {cv.x, cv.y, cv.z, cv.w};
        u32 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 t = cw[j] ^ 0x01010101u;                                          // a byte equal to 1 becomes 0
            const u32 z = (((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) & 0x80808080u;          // bit 7 set <=> the byte is non-zero
            const u32 e = ((~z & 0x80808080u) >> 7) * 0xffu;                             // 0xff where carved == 1
            u32 lab = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) lab |= ((u32)(4 * j + b) < bnd ? L0 : L1) << (8 * b);
            o[j] = e & lab;
        }
        u32x4 r; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
        *(u32x4*)(out + 16 * g) = r;
    }
}

// part_carve on labels (reference utils/voxel_carving_utils.py:139-160 with the colour grid in label form), per job:
//   occ[v]  = mask_sub[xy] && label[v] != 0 ; carved = process_voxel_grid(occ, mask_carve, angle) ; keep[v] |= mask_sub[xy] && carved[v]
//   out[v]  = keep[v] ? label[v] : 0
// G = 16 voxels per thread when a column is a whole number of 16-byte groups, else 1.
template <int G>
__global__ __launch_bounds__(256) void k_label_occ(const u8* __restrict__ label, const u8* __restrict__ mask_sub, u8* __restrict__ occ, i64 nvox, i64 D,
                                                   pb3d_magic mD, int small) {
    const i64 ngroups = nvox / G;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        const bool m = mask_sub[small ? (i64)pb3d_div((u32)(g * G), mD) : (g * G) / D] != 0;
        if (G == 16) {
            u32x4 r = (u32x4)(0u);
            if (m) {
                const u32x4 t = *(const u32x4*)(label + 16 * g);
                const u32 w[4] = {t.x, t.y, t.z, t.w};
                u32 o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { u32 v = w[k]; v |= v >> 4; v |= v >> 2; v |= v >> 1; o[k] = v & 0x01010101u; }
                r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
            }
            *(u32x4*)(occ + 16 * g) = r;
        } else {
            occ[g] = (m && label[g]) ? 1 : 0;
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void k_label_keep_or(const u8* __restrict__ carved, const u8* __restrict__ mask_sub, u8* __restrict__ keep, i64 nvox,
                                                       i64 D, int first, pb3d_magic mD, int small) {
    const i64 ngroups = nvox / G;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        const bool m = mask_sub[small ? (i64)pb3d_div((u32)(g * G), mD) : (g * G) / D] != 0;
        if (G == 16) {
            u32x4 r = (u32x4)(0u);
            if (m) {
                const u32x4 t = *(const u32x4*)(carved + 16 * g);
                const u32 w[4] = {t.x, t.y, t.z, t.w};
                u32 o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { u32 v = w[k]; v |= v >> 4; v |= v >> 2; v |= v >> 1; o[k] = v & 0x01010101u; }
                r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
            }
            if (!first) { const u32x4 k0 = *(const u32x4*)(keep + 16 * g); r.x |= k0.x; r.y |= k0.y; r.z |= k0.z; r.w |= k0.w; }
            *(u32x4*)(keep + 16 * g) = r;
        } else {
            const u8 k = (m && carved[g]) ? 1 : 0;
            keep[g] = first ? k : (u8)(keep[g] | k);
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void k_label_final(const u8* __restrict__ label, const u8* __restrict__ keep, u8* __restrict__ out, i64 nvox) {
    const i64 ngroups = nvox / G;
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (i64)gridDim.x * blockDim.x) {
        if (G == 16) {
            const u32x4 k = *(const u32x4*)(keep + 16 * g);
            u32x4 r = (u32x4)(0u);
            if (k.x | k.y | k.z | k.w) {
                const u32x4 t = *(const u32x4*)(label + 16 * g);
                r.x = t.x & (k.x * 0xffu); r.y = t.y & (k.y * 0xffu); r.z = t.z & (k.z * 0xffu); r.w = t.w & (k.w * 0xffu);   // keep bytes are 0/1
            }
            *(u32x4*)(out + 16 * g) = r;
        } else {
            out[g] = keep[g] ? label[g] : (u8)0;
        }
    }
}

int read_flag(pb3d_ctx* ctx, const int* d_flag, int* out) {
    PB3D_HIP(hipMemcpyAsync(ctx->pinned, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PB3D_TRY(pb3d_stream_sync(ctx));
    *out = *(const int*)ctx->pinned;
    return PB3D_OK;
}

int palette_words(const u8* palette, int npal, u32* pal) {
    PB3D_REQUIRE(npal >= 0 && npal <= 254 && (npal == 0 || palette), "pb3d label ops: palette of 0..254 colours");
    for (int k = 0; k < 256; ++k) pal[k] = 0;
    for (int k = 0; k < npal; ++k) pal[k + 1] = (u32)palette[3 * k] | ((u32)palette[3 * k + 1] << 8) | ((u32)palette[3 * k + 2] << 16);
    return PB3D_OK;
}

}  // namespace

extern "C" {

int pb3d_rgb_to_label_dev(pb3d_ctx* ctx, const uint8_t* d_rgb, int64_t nvox, const uint8_t* palette, int npal, uint8_t* d_label) {
    PB3D_REQUIRE(ctx && nvox >= 0, "pb3d_rgb_to_label: bad argument");
    u32 pal[256];
    PB3D_TRY(palette_words(palette, npal, pal));
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_rgb && d_label, "pb3d_rgb_to_label: null buffer");
    LabelHash h;
    PB3D_REQUIRE(build_hash(palette, npal, &h), "pb3d_rgb_to_label: the palette repeats a colour (or black): labels would be ambiguous");
    void* flag;
    PB3D_TRY(pb3d_scratch(ctx, 23, 64, &flag));
    PB3D_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_rgb_to_label, dim3(pb3d_stream_blocks(ctx, (nvox + 15) / 16, 256, 8)), dim3(256), 0, ctx->stream, d_rgb, nvox, h, d_label,
                       (int*)flag);
    PB3D_CHECK_LAUNCH();
    int bad = 0;
    PB3D_TRY(read_flag(ctx, (const int*)flag, &bad));
    PB3D_REQUIRE(!bad, "pb3d_rgb_to_label: the grid holds a colour that is neither black nor in the palette");
    return PB3D_OK;
}

// enqueue only; *flag (device) is raised when a label exceeds the palette
static int label_to_rgb_launch(pb3d_ctx* ctx, const uint8_t* d_label, int64_t nvox, const uint8_t* palette, int npal, uint8_t* d_rgb, int** flag_out) {
    Pal256 P;
    PB3D_TRY(palette_words(palette, npal, P.c));
    void* flag;
    PB3D_TRY(pb3d_scratch(ctx, 23, 64, &flag));
    PB3D_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_label_to_rgb, dim3(pb3d_stream_blocks(ctx, (nvox + 15) / 16, 256, 8)), dim3(256), 0, ctx->stream, d_label, nvox, P, npal,
                       d_rgb, (int*)flag);
    PB3D_CHECK_LAUNCH();
    if (flag_out) *flag_out = (int*)flag;
    return PB3D_OK;
}

int pb3d_label_to_rgb_dev(pb3d_ctx* ctx, const uint8_t* d_label, int64_t nvox, const uint8_t* palette, int npal, uint8_t* d_rgb) {
    PB3D_REQUIRE(ctx && nvox >= 0, "pb3d_label_to_rgb: bad argument");
    PB3D_REQUIRE(npal >= 0 && npal <= 254 && (npal == 0 || palette), "pb3d_label_to_rgb: palette of 0..254 colours");
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_label && d_rgb, "pb3d_label_to_rgb: null buffer");
    int* flag;
    PB3D_TRY(label_to_rgb_launch(ctx, d_label, nvox, palette, npal, d_rgb, &flag));
    int bad = 0;
    PB3D_TRY(read_flag(ctx, flag, &bad));
    PB3D_REQUIRE(!bad, "pb3d_label_to_rgb: a label exceeds the palette size");
    return PB3D_OK;
}

int pb3d_global_carve_label_dev(pb3d_ctx* ctx, const uint8_t* d_bin_hw, const uint8_t* d_label_hw, int64_t h, int64_t w, int angle_interval,
                                uint8_t* d_out) {
    PB3D_REQUIRE(ctx != nullptr && h >= 0 && w >= 0, "pb3d_global_carve_label: bad shape");
    PB3D_REQUIRE(angle_interval > 0, "pb3d_global_carve_label: angle_interval must be a positive integer (got %d)", angle_interval);
    const i64 W = w, H = h, D = w, nvox = W * H * D;
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_bin_hw && d_label_hw && d_out, "pb3d_global_carve_label: null buffer");
    if (angle_interval == 90 && ctx->tune_per_job != 1) {       // the write-only stream kernel of the RGB form on label bytes (csrc/bits90.hip, k_global_carve90s<., 1>)
        const i64 shape[3] = {W, H, D};
        double M[9], off[3];
        PB3D_TRY(pb3d_rotinv(90, M));
        PB3D_TRY(pb3d_offset(M, shape, off));
        if (pb3d_is_perm_step(M, off, W, D)) return pb3d_launch_global_carve90(ctx, d_bin_hw, d_label_hw, 1, h, w, M, off, 0, W, d_out);
    }
    void *ones, *carved, *tmp, *mwh;
    PB3D_TRY(pb3d_scratch(ctx, 4, (size_t)nvox, &ones));
    PB3D_TRY(pb3d_scratch(ctx, 5, (size_t)nvox, &carved));
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)nvox, &tmp));
    PB3D_TRY(pb3d_scratch(ctx, 7, (size_t)(W * H), &mwh));
    PB3D_HIP(hipMemsetAsync(ones, 1, (size_t)nvox, ctx->stream));
    PB3D_TRY(pb3d_transpose_mask_dev(ctx, d_bin_hw, H, W, (u8*)mwh));
    PB3D_TRY(pb3d_process_grid_binary_dev(ctx, (const u8*)ones, W, H, D, (const u8*)mwh, angle_interval, (u8*)carved, (u8*)tmp));
    if (nvox < (1ll << 32) && D >= 16 && (((uintptr_t)d_out) & 15u) == 0) {
        const i64 ngroups = nvox / 16;
        hipLaunchKernelGGL(k_label_apply16, dim3(pb3d_stream_blocks(ctx, ngroups, 256, 16)), dim3(256), 0, ctx->stream, (const u8*)carved, d_label_hw, W, H,
                           ngroups, pb3d_make_magic((u32)D), pb3d_make_magic((u32)H), d_out);
        if (16 * ngroups < nvox)
            hipLaunchKernelGGL(k_label_apply, dim3(1), dim3(256), 0, ctx->stream, (const u8*)carved, d_label_hw, W, H, D, d_out, 16 * ngroups);
    } else
        hipLaunchKernelGGL(k_label_apply, dim3(pb3d_stream_blocks(ctx, nvox, 256, 16)), dim3(256), 0, ctx->stream, (const u8*)carved, d_label_hw, W, H, D,
                           d_out);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

int pb3d_part_carve_label_dev(pb3d_ctx* ctx, const uint8_t* d_label, int64_t W, int64_t H, int64_t D, const uint8_t* d_mask_sub,
                              const uint8_t* d_mask_carve, const int* job_angle, const int* job_skip, int njobs, uint8_t* d_out) {
    PB3D_REQUIRE(ctx != nullptr && W >= 0 && H >= 0 && D >= 0 && njobs >= 0, "pb3d_part_carve_label: bad shape");
    const i64 nvox = W * H * D;
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_label && d_out && d_label != d_out && (njobs == 0 || (d_mask_sub && d_mask_carve && job_angle && job_skip)),
                 "pb3d_part_carve_label: null or aliased buffer");
    for (int j = 0; j < njobs; ++j)
        PB3D_REQUIRE(job_skip[j] || job_angle[j] > 0, "pb3d_part_carve_label: job %d has angle %d (must be > 0)", j, job_angle[j]);
    // all live jobs at 90 degrees (the notebook's group_jobs): ONE launch of the plane-local kernel on the label bytes (csrc/bits90.hip,
    // k_part90_plane<1>: occupancy = label != 0) instead of occupancy / process / keep passes per job
    {
        bool all90 = njobs > 0 && njobs <= 32, any_live = false;
        for (int j = 0; j < njobs; ++j)
            if (!job_skip[j]) { any_live = true; all90 = all90 && job_angle[j] == 90; }
        if (all90 && any_live && ctx->tune_per_job != 1) {
            const int rc = pb3d_try_part_carve90(ctx, d_label, 1, W, H, D, d_mask_sub, d_mask_carve, job_angle, job_skip, njobs, d_out);
            if (rc != PB3D_EUNSUPPORTED) return rc;
        }
    }
    void *occ, *carved, *tmp, *keep;
    PB3D_TRY(pb3d_scratch(ctx, 4, (size_t)nvox, &occ));
    PB3D_TRY(pb3d_scratch(ctx, 5, (size_t)nvox, &carved));
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)nvox, &tmp));
    PB3D_TRY(pb3d_scratch(ctx, 7, (size_t)nvox, &keep));
    const bool wide = D % 16 == 0 && ((((uintptr_t)d_label) | ((uintptr_t)d_out)) & 15u) == 0;
    const unsigned blocks = pb3d_stream_blocks(ctx, wide ? nvox / 16 : nvox, 256, 8);
    bool any = false;
    const pb3d_magic mDl = pb3d_make_magic((u32)(D > 0 && D < (1ll << 31) ? D : 1));
    const int smalll = (nvox < (1ll << 32) && D < (1ll << 31)) ? 1 : 0;
    for (int j = 0; j < njobs; ++j) {
        if (job_skip[j]) continue;
        const u8* ms = d_mask_sub + (i64)j * W * H;
        const u8* mc = d_mask_carve + (i64)j * W * H;
        if (wide) hipLaunchKernelGGL(k_label_occ<16>, dim3(blocks), dim3(256), 0, ctx->stream, d_label, ms, (u8*)occ, nvox, D, mDl, smalll);
        else hipLaunchKernelGGL(k_label_occ<1>, dim3(blocks), dim3(256), 0, ctx->stream, d_label, ms, (u8*)occ, nvox, D, mDl, smalll);
        PB3D_CHECK_LAUNCH();
        PB3D_TRY(pb3d_process_grid_binary_dev(ctx, (const u8*)occ, W, H, D, mc, job_angle[j], (u8*)carved, (u8*)tmp));
        if (wide) hipLaunchKernelGGL(k_label_keep_or<16>, dim3(blocks), dim3(256), 0, ctx->stream, (const u8*)carved, ms, (u8*)keep, nvox, D, any ? 0 : 1, mDl, smalll);
        else hipLaunchKernelGGL(k_label_keep_or<1>, dim3(blocks), dim3(256), 0, ctx->stream, (const u8*)carved, ms, (u8*)keep, nvox, D, any ? 0 : 1, mDl, smalll);
        PB3D_CHECK_LAUNCH();
        any = true;
    }
    if (!any) { PB3D_HIP(hipMemsetAsync(d_out, 0, (size_t)nvox, ctx->stream)); return PB3D_OK; }
    if (wide) hipLaunchKernelGGL(k_label_final<16>, dim3(blocks), dim3(256), 0, ctx->stream, d_label, (const u8*)keep, d_out, nvox);
    else hipLaunchKernelGGL(k_label_final<1>, dim3(blocks), dim3(256), 0, ctx->stream, d_label, (const u8*)keep, d_out, nvox);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

// ---- sharded forms: one process per GPU, communicator from pb3d_comm_init; rank r of n owns the X-planes [r W/n, (r+1) W/n) ----------
int pb3d_carve_mask_sharded_dev(pb3d_ctx* ctx, const uint8_t* d_grid_slab, int64_t W, int64_t H, int64_t D, int C, const uint8_t* d_mask_wh,
                                uint8_t* d_out_full) {
    PB3D_REQUIRE(ctx && W >= 0 && H >= 0 && D >= 0 && (C == 1 || C == 3), "pb3d_carve_mask_sharded: bad shape");
    PB3D_REQUIRE(ctx->rccl_comm != nullptr, "pb3d_carve_mask_sharded: call pb3d_comm_init first");
    PB3D_REQUIRE(W % ctx->nranks == 0, "pb3d_carve_mask_sharded: %lld planes do not split evenly over %d ranks", (long long)W, ctx->nranks);
    const i64 planes = W / ctx->nranks, x0 = planes * ctx->rank;
    const size_t slab = (size_t)(planes * H * D * C);
    if (slab == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid_slab && d_mask_wh && d_out_full, "pb3d_carve_mask_sharded: null buffer");
    u8* mine = d_out_full + slab * (size_t)ctx->rank;
    PB3D_TRY(pb3d_carve_mask_dev(ctx, d_grid_slab, planes, H, D, C, d_mask_wh + x0 * H, mine));
    return pb3d_allgather_dev(ctx, mine, d_out_full, slab);                  // in place: ONE collective
}

int pb3d_global_carve_sharded_dev(pb3d_ctx* ctx, const uint8_t* d_bin_hw, const uint8_t* d_rgb_hw3, int64_t h, int64_t w, int angle_interval,
                                  uint8_t* d_out_full) {
    PB3D_REQUIRE(ctx && h >= 0 && w >= 0, "pb3d_global_carve_sharded: bad shape");
    PB3D_REQUIRE(ctx->rccl_comm != nullptr, "pb3d_global_carve_sharded: call pb3d_comm_init first");
    PB3D_REQUIRE(w % ctx->nranks == 0, "pb3d_global_carve_sharded: %lld planes do not split evenly over %d ranks", (long long)w, ctx->nranks);
    const i64 planes = w / ctx->nranks, x0 = planes * ctx->rank;
    const size_t slab = (size_t)(planes * h * w * 3);
    if (slab == 0) return PB3D_OK;
    PB3D_REQUIRE(d_out_full != nullptr, "pb3d_global_carve_sharded: null buffer");
    u8* mine = d_out_full + slab * (size_t)ctx->rank;
    PB3D_TRY(pb3d_global_carve_dev(ctx, d_bin_hw, d_rgb_hw3, h, w, angle_interval, x0, x0 + planes, mine));
    return pb3d_allgather_dev(ctx, mine, d_out_full, slab);
}

int pb3d_carve_labels_sharded_dev(pb3d_ctx* ctx, const uint8_t* d_label_slab, int64_t W, int64_t H, int64_t D, const uint8_t* d_mask_wh,
                                  uint8_t* d_label_full, const uint8_t* palette, int npal, uint8_t* d_rgb_full) {
    PB3D_TRY(pb3d_carve_mask_sharded_dev(ctx, d_label_slab, W, H, D, 1, d_mask_wh, d_label_full));
    if (!d_rgb_full || W * H * D == 0) return PB3D_OK;
    PB3D_REQUIRE(npal >= 0 && npal <= 254 && (npal == 0 || palette), "pb3d_carve_labels_sharded: palette of 0..254 colours");
    // enqueue only (no flag read-back: this entry stays asynchronous); labels above npal expand to black
    return label_to_rgb_launch(ctx, d_label_full, W * H * D, palette, npal, d_rgb_full, nullptr);
}

}  // extern "C"
