// K2: rotate-about-Y (trilinear, uint8 rounding) + mask carve -- the ARITHMETIC kernel -- and the process_voxel_grid loop.
//
// Round 4: this file held five generations of table-driven byte-tile kernels for 0/1 data (k_rotate_bits, _bits32, _bits16w, _bits8p and
// their table builders, 1 400 lines).  Every rotation step on 0/1 data -- chains and single steps alike -- now runs bit-sliced
// (csrc/sliced.hip: a single 60-degree step through slice -> table step -> un-slice measured equal or faster at every size from 512-class
// grids up, 5 us slower at 128^3; tools/singlestep_ab.py, profiles/r04_singlestep_ab.jsonl), single 90-degree steps on the permutation
// kernels (csrc/rotate_tiled.hip).  What remains here is the kernel that evaluates SciPy's arithmetic voxel by voxel: grids with values
// other than 0/1, shapes the sliced chain does not take, and the pinned reference of the parity tests (tune sliced = 1).
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
// All doubles are evaluated with __dmul_rn/__dadd_rn so that no contraction can occur; the file is
// also compiled with -ffp-contract=off.  The coordinates depend on (x,z) only, so a thread computes
// them once and re-uses them for every Y-plane of its chunk (the FP64 work is amortised over y).
#include "rot_common.h"

namespace {

// value of one output voxel from plane y of `in`; taps with an exactly-zero weight add +0.0 and are
// skipped (this also keeps the index in range: a tap beyond n-1 only ever occurs with weight 0).
// mask_src (optional): the input is read as carve(in, mask_src) -- a source column (n0, y) with mask_src[n0*H + y] == 0 reads as zeros
// (the 0-degree carve of process_voxel_grid folded into its first rotation step).
__device__ __forceinline__ u32 sample(const u8* __restrict__ in, const Cell& c, i64 y, i64 H, i64 D, const u8* __restrict__ mask_src = nullptr) {
    if (c.s0 < 0) return 0;
    const u8* r0 = in + ((i64)c.s0 * H + y) * D + c.s2;
    const u8* r1 = r0 + H * D;
    double acc = 0.0;
    const bool x1 = c.wx1 != 0.0, z1 = c.wz1 != 0.0;
    const double k0 = (!mask_src || mask_src[(i64)c.s0 * H + y]) ? 1.0 : 0.0;            // v * 1.0 == v, v * 0.0 == +0.0: exact
    const double k1 = (x1 && (!mask_src || mask_src[((i64)c.s0 + 1) * H + y])) ? 1.0 : 0.0;
    // weights are >= 0; a zero wx0/wz0 (never happens: w0 = 1 - t > 0) needs no special case
    acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(__dmul_rn((double)r0[0], k0), c.wx0), c.wz0));
    if (z1) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(__dmul_rn((double)r0[1], k0), c.wx0), c.wz1));
    if (x1) {
        acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(__dmul_rn((double)r1[0], k1), c.wx1), c.wz0));
        if (z1) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(__dmul_rn((double)r1[1], k1), c.wx1), c.wz1));
    }
    if (!(acc > 0.0)) return 0;
    acc = __dadd_rn(acc, 0.5);
    if (acc > 255.0) acc = 255.0;
    return (u32)acc;  // truncation
}

// Tile: 4 x-rows (one per wavefront) by 256 z (4 consecutive z per lane), swept over TY planes.
template <bool PACK>
__global__ __launch_bounds__(256) void k_rotate_generic(const u8* __restrict__ in, u8* __restrict__ out,
                                                        const u8* __restrict__ mask_wh, RotParams p, i64 W, i64 H, i64 D,
                                                        int TY, const u8* __restrict__ mask_src) {
    const int lane = threadIdx.x & 63;
    const i64 x = (i64)blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const i64 z0 = ((i64)blockIdx.x * 64 + lane) * 4;
    if (x >= W || z0 >= D) return;
    const i64 y_beg = (i64)blockIdx.z * TY;
    const i64 y_end = y_beg + TY < H ? y_beg + TY : H;
    Cell c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (z0 + q < D) c[q] = make_cell(p, x, z0 + q, W, D);
        else { c[q].s0 = -1; c[q].s2 = 0; c[q].wx0 = c[q].wx1 = c[q].wz0 = c[q].wz1 = 0.0; }
    }
    for (i64 y = y_beg; y < y_end; ++y) {
        const bool keep = mask_wh ? mask_wh[x * H + y] != 0 : true;  // wave-uniform
        u32 r[4] = {0, 0, 0, 0};
        if (keep) {
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = sample(in, c[q], y, H, D, mask_src);
        }
        u8* o = out + (x * H + y) * D + z0;
        if (PACK) {
            *(u32*)o = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (z0 + q < D) o[q] = (u8)r[q];
        }
    }
}

bool is_zero(double v) { return v == 0.0; }

}  // namespace

int pb3d_launch_rotate_generic(pb3d_ctx* ctx, const u8* d_in, i64 W, i64 H, i64 D, const double M[9],
                               const double off[3], const u8* d_mask_wh, u8* d_out, const u8* d_mask_src) {
    PB3D_REQUIRE(is_zero(M[3]) && M[4] == 1.0 && is_zero(M[5]) && is_zero(M[1]) && is_zero(M[7]) && is_zero(off[1]),
                 "pb3d_rotate_carve: matrix is not a rotation about Y (row 1 must be [0,1,0], M[0][1]=M[2][1]=0, off[1]=0)");
    PB3D_REQUIRE(W < (1ll << 30) && D < (1ll << 30), "pb3d_rotate_carve: axis too long");
    if (W * H * D == 0) return PB3D_OK;
    RotParams p = {M[0], M[1], M[2], off[0], M[6], M[7], M[8], off[2]};
    int TY = 16;
    // keep at least ~8 blocks per CU in flight for small grids
    const i64 tiles_xz = ((D + 255) / 256) * ((W + 3) / 4);
    while (TY > 1 && tiles_xz * ((H + TY - 1) / TY) < (i64)ctx->cus * 8) TY >>= 1;
    const dim3 grid((unsigned)((D + 255) / 256), (unsigned)((W + 3) / 4), (unsigned)((H + TY - 1) / TY));
    PB3D_REQUIRE(grid.y <= 65535u && grid.z <= 65535u, "pb3d_rotate_carve: grid too large");
    const bool pack = (D % 4 == 0) && (((uintptr_t)d_out & 3u) == 0);
    if (pack) hipLaunchKernelGGL(k_rotate_generic<true>, grid, dim3(256), 0, ctx->stream, d_in, d_out, d_mask_wh, p, W, H, D, TY, d_mask_src);
    else hipLaunchKernelGGL(k_rotate_generic<false>, grid, dim3(256), 0, ctx->stream, d_in, d_out, d_mask_wh, p, W, H, D, TY, d_mask_src);
    PB3D_CHECK_LAUNCH();
    return PB3D_OK;
}

static int process_grid_impl(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D, const uint8_t* d_mask_wh,
                             int angle_interval, uint8_t* d_out, uint8_t* d_tmp, int known_binary);

extern "C" {

int pb3d_rotate_carve_dev(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D,
                          const double M[9], const double off[3], const uint8_t* d_mask_wh, uint8_t* d_out) {
    PB3D_REQUIRE(ctx != nullptr && M && off, "pb3d_rotate_carve: null argument");
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0, "pb3d_rotate_carve: bad shape");
    if (W * H * D == 0) return PB3D_OK;
    PB3D_REQUIRE(d_occ && d_out && d_occ != d_out, "pb3d_rotate_carve: null or aliased buffer");
    const bool row1 = M[3] == 0.0 && M[4] == 1.0 && M[5] == 0.0 && M[1] == 0.0 && M[7] == 0.0 && off[1] == 0.0;
    if (row1 && pb3d_perm_step_ok(M, off, W, D, d_occ, d_out))
        return pb3d_launch_rotate_perm(ctx, d_occ, W, H, D, M, off, nullptr, d_mask_wh, d_out);
    if (row1) {         // 0/1 data: one bit-sliced table step (one host wait for the slice pass's "is it 0/1" verdict)
        int took = 0;
        PB3D_TRY(pb3d_rotate_step_sliced(ctx, d_occ, W, H, D, M, off, d_mask_wh, d_out, &took));
        if (took) return PB3D_OK;
    }
    return pb3d_launch_rotate_generic(ctx, d_occ, W, H, D, M, off, d_mask_wh, d_out, nullptr);
}

int pb3d_process_grid_dev(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D,
                          const uint8_t* d_mask_wh, int angle_interval, uint8_t* d_out, uint8_t* d_tmp) {
    return process_grid_impl(ctx, d_occ, W, H, D, d_mask_wh, angle_interval, d_out, d_tmp, 0);
}

}  // extern "C"

int pb3d_process_grid_binary_dev(pb3d_ctx* ctx, const u8* d_occ, i64 W, i64 H, i64 D, const u8* d_mask_wh, int angle_interval, u8* d_out,
                                 u8* d_tmp) {
    return process_grid_impl(ctx, d_occ, W, H, D, d_mask_wh, angle_interval, d_out, d_tmp, 1);
}

static int process_grid_impl(pb3d_ctx* ctx, const uint8_t* d_occ, int64_t W, int64_t H, int64_t D,
                             const uint8_t* d_mask_wh, int angle_interval, uint8_t* d_out, uint8_t* d_tmp, int known_binary) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_process_grid: null context");
    PB3D_REQUIRE(W >= 0 && H >= 0 && D >= 0, "pb3d_process_grid: bad shape");
    PB3D_REQUIRE(angle_interval > 0, "pb3d_process_grid: angle_interval must be a positive integer (got %d)", angle_interval);
    const i64 nvox = W * H * D;
    if (nvox == 0) return PB3D_OK;
    PB3D_REQUIRE(d_occ && d_mask_wh && d_out && d_tmp, "pb3d_process_grid: null buffer");
    PB3D_REQUIRE(d_out != d_occ && d_tmp != d_occ && d_tmp != d_out, "pb3d_process_grid: buffers must not alias");
    const int nsteps = 90 / angle_interval + 1;  // len(range(0, 91, k))
    const i64 shape[3] = {W, H, D};
    // Step 0 is Rinv(0) = I with zero offset: coordinates are the integers themselves, weights {1,0},
    // acc == v and (uint8)(v + 0.5) == v, i.e. the step is the mask carve alone.
    double M0[9], off0[3];
    PB3D_TRY(pb3d_rotinv(0, M0));
    PB3D_TRY(pb3d_offset(M0, shape, off0));
    bool ident = off0[0] == 0.0 && off0[1] == 0.0 && off0[2] == 0.0;
    for (int k = 0; k < 9; ++k) ident = ident && M0[k] == ((k % 4 == 0) ? 1.0 : 0.0);
    PB3D_REQUIRE(ident, "pb3d_process_grid: internal error, Rinv(0) is not the identity");
    if (nsteps == 1) return pb3d_carve_mask_dev(ctx, d_occ, W, H, D, 1, d_mask_wh, d_out);
    {   // rotation steps on 0/1 data run bit-sliced (csrc/sliced.hip; a single 90-degree step stays on the permutation kernels below);
        // data that is not 0 / 1 comes back here
        int took = 0;
        PB3D_TRY(pb3d_process_grid_sliced(ctx, d_occ, W, H, D, d_mask_wh, angle_interval, d_out, known_binary, &took));
        if (took) return PB3D_OK;
    }
    // The byte chain.  The first rotation takes the 0-degree carve as its SOURCE mask (one pass over the volume less): a
    // permutation-like step (90 degrees, W + D even) when it can run in both directions of the ping-pong, the arithmetic kernel always.
    double M1[9], off1[3];
    PB3D_TRY(pb3d_rotinv(angle_interval, M1));
    PB3D_TRY(pb3d_offset(M1, shape, off1));
    const bool perm1 = pb3d_perm_step_ok(M1, off1, W, D, d_occ, d_out) && pb3d_perm_step_ok(M1, off1, W, D, d_occ, d_tmp);
    const bool fuse_first = perm1 || !pb3d_is_perm_step(M1, off1, W, D);
    const int nlaunch = fuse_first ? nsteps - 1 : nsteps;
    const u8* src = d_occ;
    int li = 0;
    for (int s = fuse_first ? 1 : 0; s < nsteps; ++s, ++li) {
        u8* dst = ((nlaunch - 1 - li) % 2 == 0) ? d_out : d_tmp;  // ping-pong so that the last step lands in d_out
        if (s == 0) {
            PB3D_TRY(pb3d_carve_mask_dev(ctx, src, W, H, D, 1, d_mask_wh, dst));
        } else {
            double M[9], off[3];
            PB3D_TRY(pb3d_rotinv(s * angle_interval, M));
            PB3D_TRY(pb3d_offset(M, shape, off));
            if (pb3d_perm_step_ok(M, off, W, D, src, dst))
                PB3D_TRY(pb3d_launch_rotate_perm(ctx, src, W, H, D, M, off, (fuse_first && s == 1) ? d_mask_wh : nullptr, d_mask_wh, dst));
            else
                PB3D_TRY(pb3d_launch_rotate_generic(ctx, src, W, H, D, M, off, d_mask_wh, dst, (fuse_first && s == 1) ? d_mask_wh : nullptr));
        }
        src = dst;
    }
    return PB3D_OK;
}
