// process_voxel_grid for grids that are not uint8.
//
// reference utils/voxel_carving_utils.py:104-126 never looks at the grid's dtype: scipy.ndimage.affine_transform(order=1, mode="constant",
// cval=0) returns the dtype it was given (int8 .. int64, uint8 .. uint64, float32 / 64, complex64 / 128; float16 raises "data type
// not supported") and carve_voxel_grid_with_masks (np.where(mask, grid, 0)) keeps it -- except bool, which NumPy promotes to int64 with the
// Python 0: a bool grid is an int64 grid from the first step on (the host side converts it; the 0-degree step is the identity).  The notebooks only ever pass uint8 occupancy, which has its own
// kernels (rotate.hip, sliced.hip, rotate_tiled.hip); this file is the plain restatement for everything else -- one thread per (x, z) cell,
// the cell's f64 coordinates and weights formed once (rot_common.h: the same arithmetic as the uint8 kernels), all planes of the cell in a loop:
//     acc = (((v00*wx0)*wz0 + (v01*wx0)*wz1) + (v10*wx1)*wz0) + (v11*wx1)*wz1        taps with an exactly-zero weight skipped (they add +-0)
//     store (scipy/ndimage/src/ni_interpolation.c, CASE_INTERP_OUT*):
//         float32 / float64:  (T)acc
//         unsigned:           acc > 0 ? acc + 0.5 : 0, clipped to [0, MAX], truncated
//         signed:             acc > 0 ? acc + 0.5 : acc - 0.5, clipped to [MIN, MAX], truncated
//         complex:            real and imaginary parts separately (scipy/ndimage/_interpolation.py transforms them one after the other)
// NON-FINITE VALUES are outside the parity contract: SciPy multiplies every one of its (order + 1)^3 taps by its weight, the zero weights
// and the y + 1 plane included, so an inf or NaN next to a cell turns the cell into NaN (inf * 0); this kernel skips zero-weight taps and
// never touches the y + 1 plane, so non-finite values go only where taps with NON-ZERO weight carry them (the 0-degree step moves them
// like any value; at 90 degrees cos = 6e-17 gives the second taps tiny weights, and a non-finite neighbour reaches the cell as in SciPy).  Pinned by tests/test_gpu_parity.py::test_process_typed_non_finite_values_are_carried_not_spread; finite data is byte-exact.
// and the carve of the step (np.where(mask, g, 0)) in the same store.  64-bit integers beyond 2^53 go through a double exactly as in SciPy;
// a value that clips to 2^63 / 2^64 is converted out of range there (C leaves the result to the platform) -- not reproduced.
#include "rot_common.h"

namespace {

enum { K_FLOAT = 0, K_UNSIGNED = 1, K_SIGNED = 2 };

template <typename T, int KIND>
__device__ __forceinline__ T store_of(double acc) {
    if constexpr (KIND == K_FLOAT) return (T)acc;
    else if constexpr (KIND == K_UNSIGNED) {
        double t = acc > 0.0 ? __dadd_rn(acc, 0.5) : 0.0;
        const double mx = (double)(T)~(T)0;
        if (t > mx) t = mx;
        if (t < 0.0) t = 0.0;
        return (T)t;
    } else {
        double t = acc > 0.0 ? __dadd_rn(acc, 0.5) : __dsub_rn(acc, 0.5);
        const T tmax = (T)(((unsigned long long)1 << (8 * sizeof(T) - 1)) - 1), tmin = (T)(-(long long)tmax - 1);
        const double mx = (double)tmax, mn = (double)tmin;
        if (t > mx) t = mx;
        if (t < mn) t = mn;
        return (T)t;
    }
}

// grid as (W, H, D, NC) elements of T (NC = 2: complex, parts interleaved); out likewise; mask_wh (W, H) bytes, != 0 keeps
template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_rotate_typed(const T* __restrict__ in, T* __restrict__ out, const u8* __restrict__ mask_wh, RotParams p,
                                                      i64 W, i64 H, i64 D, int NC) {
    const i64 z = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 x = blockIdx.y;
    if (z >= D) return;
    const Cell c = make_cell(p, x, z, W, D);
    const bool x1 = c.wx1 != 0.0, z1 = c.wz1 != 0.0;
    for (i64 y = 0; y < H; ++y) {
        T* o = out + ((x * H + y) * D + z) * NC;
        const bool keep = mask_wh[x * H + y] != 0;
        for (int k = 0; k < NC; ++k) {
            double acc = 0.0;
            if (keep && c.s0 >= 0) {
                const T* r0 = in + (((i64)c.s0 * H + y) * D + c.s2) * NC + k;
                const T* r1 = r0 + H * D * NC;
                acc = __dadd_rn(acc, __dmul_rn(__dmul_rn((double)r0[0], c.wx0), c.wz0));
                if (z1) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn((double)r0[NC], c.wx0), c.wz1));
                if (x1) {
                    acc = __dadd_rn(acc, __dmul_rn(__dmul_rn((double)r1[0], c.wx1), c.wz0));
                    if (z1) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn((double)r1[NC], c.wx1), c.wz1));
                }
            }
            // (a dropped or outside cell stores the type's zero: acc == +0.0 goes through the same rule)
            o[k] = keep ? store_of<T, KIND>(acc) : (T)0;
        }
    }
}

template <typename T, int KIND>
int launch_typed(pb3d_ctx* ctx, const void* in, void* out, const u8* mask, const RotParams& p, i64 W, i64 H, i64 D, int NC) {
    dim3 grid((unsigned)((D + 255) / 256), (unsigned)W);
    hipLaunchKernelGGL((k_rotate_typed<T, KIND>), grid, dim3(256), 0, ctx->stream, (const T*)in, (T*)out, mask, p, W, H, D, NC);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

}  // namespace

extern "C" {

size_t pb3d_dtype_bytes(int dtype) {
    switch (dtype) {
        case PB3D_I8: case PB3D_U8: return 1;
        case PB3D_I16: case PB3D_U16: return 2;
        case PB3D_I32: case PB3D_U32: case PB3D_F32: return 4;
        case PB3D_I64: case PB3D_U64: case PB3D_F64: case PB3D_C64: return 8;
        case PB3D_C128: return 16;
        default: return 0;
    }
}

int pb3d_process_grid_typed_dev(pb3d_ctx* ctx, const void* d_grid, int dtype, int64_t W, int64_t H, int64_t D, const uint8_t* d_mask_wh,
                                int angle_interval, void* d_out, void* d_tmp) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_process_grid_typed: null context");
    const size_t es = pb3d_dtype_bytes(dtype);
    PB3D_REQUIRE(es != 0, "pb3d_process_grid_typed: data type not supported (code %d)", dtype);
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0 && W <= 65535 && W * H * D < (1ll << 40), "pb3d_process_grid_typed: bad shape (%lld,%lld,%lld)", (long long)W,
                 (long long)H, (long long)D);
    PB3D_REQUIRE(angle_interval >= 1, "pb3d_process_grid_typed: angle_interval must be >= 1");
    if (W * H * D == 0) return PB3D_OK;
    PB3D_REQUIRE(d_grid && d_mask_wh && d_out && d_tmp && d_out != d_grid && d_tmp != d_grid && d_out != d_tmp, "pb3d_process_grid_typed: null or aliased buffer");
    int nsteps = 0;
    for (int a = 0; a < 91; a += angle_interval) ++nsteps;
    const void* src = d_grid;
    const i64 shape[3] = {W, H, D};
    int step = 0;
    for (int a = 0; a < 91; a += angle_interval, ++step) {
        double M[9], off[3];
        PB3D_TRY(pb3d_rotinv(a, M));
        PB3D_TRY(pb3d_offset(M, shape, off));
        const RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
        void* dst = ((nsteps - 1 - step) & 1) ? d_tmp : d_out;              // the last step writes d_out
        int rc;
        switch (dtype) {
            case PB3D_I8: rc = launch_typed<signed char, K_SIGNED>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_U8: rc = launch_typed<u8, K_UNSIGNED>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_I16: rc = launch_typed<short, K_SIGNED>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_U16: rc = launch_typed<unsigned short, K_UNSIGNED>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_I32: rc = launch_typed<int, K_SIGNED>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_U32: rc = launch_typed<u32, K_UNSIGNED>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_I64: rc = launch_typed<long long, K_SIGNED>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_U64: rc = launch_typed<unsigned long long, K_UNSIGNED>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_F32: rc = launch_typed<float, K_FLOAT>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_F64: rc = launch_typed<double, K_FLOAT>(ctx, src, dst, d_mask_wh, p, W, H, D, 1); break;
            case PB3D_C64: rc = launch_typed<float, K_FLOAT>(ctx, src, dst, d_mask_wh, p, W, H, D, 2); break;
            default: rc = launch_typed<double, K_FLOAT>(ctx, src, dst, d_mask_wh, p, W, H, D, 2); break;
        }
        if (rc != PB3D_OK) return rc;
        src = dst;
    }
    return PB3D_OK;
}

}  // extern "C"
