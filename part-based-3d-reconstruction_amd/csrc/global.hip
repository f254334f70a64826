// global_carve (reference utils/voxel_carving_utils.py:269-298): ones((w,h,w)) -> process -> colour.
#include <cmath>

#include "pb3d_internal.h"

namespace {

__global__ __launch_bounds__(256) void k_transpose_mask(const u8* __restrict__ hw, u8* __restrict__ wh, i64 h, i64 w) {
    const i64 n = h * w;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        const i64 x = i / h, y = i - x * h;
        wh[i] = hw[y * w + x] ? 1 : 0;
    }
}

}  // namespace

// (h,w) truthiness image -> (w,h) 0/1 image (the carving mask in the grid's own axis order)
int pb3d_transpose_mask_dev(pb3d_ctx* ctx, const u8* d_hw, i64 h, i64 w, u8* d_wh) {
    hipLaunchKernelGGL(k_transpose_mask, dim3(pb3d_stream_blocks(ctx, w * h, 256, 8)), dim3(256), 0, ctx->stream, d_hw, d_wh, h, w);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

extern "C" {

int pb3d_global_carve_dev(pb3d_ctx* ctx, const uint8_t* d_bin_hw, const uint8_t* d_rgb_hw3, int64_t h, int64_t w,
                          int angle_interval, int64_t x0, int64_t x1, uint8_t* d_out_slab) {
    PB3D_REQUIRE(ctx != nullptr && h >= 0 && w >= 0, "pb3d_global_carve: bad shape");
    PB3D_REQUIRE(angle_interval > 0, "pb3d_global_carve: angle_interval must be a positive integer (got %d)", angle_interval);
    PB3D_REQUIRE(x0 >= 0 && x0 <= x1 && x1 <= w, "pb3d_global_carve: bad slab [%lld,%lld) of %lld", (long long)x0, (long long)x1,
                 (long long)w);
    const i64 W = w, H = h, D = w, nvox = W * H * D;
    if (nvox == 0 || x1 == x0) return PB3D_OK;
    PB3D_REQUIRE(d_bin_hw && d_rgb_hw3 && d_out_slab, "pb3d_global_carve: null buffer");
    const i64 shape[3] = {W, H, D};
    if (angle_interval == 90) {
        // angles [0, 90]: both steps are exact permutations on a (w,h,w) grid -> one write-only kernel (k_global_carve90s, csrc/bits90.hip)
        double M[9], off[3];
        PB3D_TRY(pb3d_rotinv(90, M));
        PB3D_TRY(pb3d_offset(M, shape, off));
        if (pb3d_is_perm_step(M, off, W, D)) return pb3d_launch_global_carve90(ctx, d_bin_hw, d_rgb_hw3, 3, h, w, M, off, x0, x1, d_out_slab);
    }
    // Other angle steps: the chain of process_voxel_grid on a grid that never exists as bytes before its final (W,H,D,3) form -- the mask
    // bits are written as the bit-sliced volume, the rotation steps run on it, the last one writes the colours (csrc/sliced.hip).
    // Slab outputs need the fused 90-degree path above.
    PB3D_REQUIRE(x0 == 0 && x1 == W, "pb3d_global_carve: slab output needs the fused 90-degree path (angle_interval=90)");
    const int nsteps = 90 / angle_interval + 1;  // len(range(0, 91, k))
    void* mwh;
    PB3D_TRY(pb3d_scratch(ctx, 7, (size_t)(W * H), &mwh));
    PB3D_TRY(pb3d_transpose_mask_dev(ctx, d_bin_hw, H, W, (u8*)mwh));
    if (nsteps >= 2 && ctx->tune_global_composed != 1) {             // knob global_composed = 1: the composed pipeline (parity tests run both)
        int took = 0;
        PB3D_TRY(pb3d_global_carve_sliced(ctx, (const u8*)mwh, d_rgb_hw3, W, H, D, angle_interval, d_out_slab, &took));
        if (took) return PB3D_OK;
    }
    // composed pipeline: ones -> process_voxel_grid -> colour
    void *ones, *carved, *tmp;
    PB3D_TRY(pb3d_scratch(ctx, 4, (size_t)nvox, &ones));
    PB3D_TRY(pb3d_scratch(ctx, 5, (size_t)nvox, &carved));
    PB3D_TRY(pb3d_scratch(ctx, 6, (size_t)nvox, &tmp));
    PB3D_HIP(hipMemsetAsync(ones, 1, (size_t)nvox, ctx->stream));
    PB3D_TRY(pb3d_process_grid_binary_dev(ctx, (const u8*)ones, W, H, D, (const u8*)mwh, angle_interval, (u8*)carved, (u8*)tmp));
    return pb3d_color_apply_dev(ctx, (const u8*)carved, W, H, D, d_rgb_hw3, d_out_slab);
}

}  // extern "C"
