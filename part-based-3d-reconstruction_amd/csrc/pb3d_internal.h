// Internal declarations shared by the translation units of libpb3d.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "pb3d.h"

typedef uint8_t u8;
typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;

#define PB3D_NSCRATCH 48
#define PB3D_POOL_SLOTS 64
#define PB3D_POOL_LIVE 4096

struct pb3d_event {
    hipEvent_t ev;
};

struct pb3d_ctx {
    int device;
    int cus;
    hipStream_t stream;
    bool orient_lds_set;        // ... and for the 128-pixel orientation kernel (csrc/components.hip)
    bool guided_lds_set;        // ... and for the crop-chain kernel of left_right_guided_carve (csrc/guided.hip)
    // development knobs, read from the environment ONCE in pb3d_create (never on a launch path)
    // (round 4: every knob has a name -- the numbered "misc" switches of rounds 1-3 are gone; PB3D_KNOBS="name=value,..." sets any of them at pb3d_create)
    int tune_rot90_fill;        // knob "rot90_fill": workgroups per CU in the grid of the 90-degree kernels (0 = the built-in rule)
    int tune_rot90_order;       // knob "rot90_order": 1 = tiles in plain x-fastest order instead of one plane chunk per XCD
    int tune_rot90_flat;        // knob "rot90_flat": 0 = choose, 1 = never the flat (stream) forms: the row-wise tile kernel, 2 = flat only on whole-line streams
    int tune_rot90_mask_block;  // knob "rot90_mask_block": 1 = the 90-degree kernels fetch their mask bytes per plane / segment instead of once into LDS
    int tune_points_onepass;    // knob "points_onepass": 1 = the host entry of the point extraction uses the one-pass (look-back) form
    int tune_orient_tile;       // knob "orient_tile": 1 = pb3d_orient_dev without its 128-pixel tile kernel
    int tune_global_composed;   // knob "global_composed": 1 = global_carve with other angle steps as ones -> process -> colour (parity tests run both)
    int tune_per_job;           // knob "per_job": 1 = part_carve's non-90 jobs one by one (no merged pass); label forms of global_carve / part_carve by their per-job passes
    int tune_no_table_cache;    // knob "no_table_cache": 1 = validity tables and tile programs are rebuilt on every call
    int tune_part90_inflight;   // knob "part90_inflight": 10 UA + UE, items in flight per thread in the source / output pass of k_part90_plane (0 = 2 / 4)
    int tune_crop_ablate;       // knob "crop_ablate": ablation switches of k_crop_chain (tools/cropabl.py)
    int tune_uncap;             // PB3D_UNCAP=1: every grid-stride kernel gets one workgroup per tile (A/B of the persistent grids)
    int tune_sliced;            // PB3D_SLICED: 0 = rotation steps on 0/1 data run bit-sliced (csrc/sliced.hip), 1 = never (byte chain, arithmetic kernel)
    int tune_rot90_wide;        // PB3D_ROT90_WIDE: 1 = the 256 x 256-tile form of the 90-degree step (development A/B)
    bool rot90w_lds_set;
    bool part90_lds_set;        // k_part90_plane has been given its large dynamic LDS limit
    bool rot90wf_lds_set;       // ... and for its form on the rows' (y, z) streams (odd row lengths)
    int tune_s32_order;         // knob "s32_order": 1 = the slice / un-slice passes walk x fastest (round 3's order; development A/B)
    int tune_s32_fuse_last;     // knob "s32_fuse_last": 1 = the chain's last 90-degree step stays a table step (development A/B)
    int tune_s32_gpw;           // PB3D_S32_GPW: plane groups per workgroup of the sliced step kernel (0 = choose)
    int tune_ccl_blocks;        // knob "ccl_blocks": workgroups per CU of the labelling's last pass (0 = default)
    int tune_ccl_init_blocks;   // knob "ccl_init_blocks": workgroups per CU of the labelling's first pass (0 = 16)
    int tune_ccl_tilecols;      // knob "ccl_tilecols": windows per level of a plane-to-plane merge tile (0 = 32)
    int tune_points_fill;       // knob "points_fill": 1 = the block form of the two-pass fill (k_points_fill16; development A/B)
    int tune_ccl_merge;         // knob "ccl_merge": 0 = tile kernels where the rows fit, 1 = always the pairwise kernel (development A/B)
    // Growable device scratch slots used by the host-pointer entry points (no hipMalloc /
    // hipFree per call once warm).
    void* scratch[PB3D_NSCRATCH];
    size_t scratch_bytes[PB3D_NSCRATCH];
    // small pinned host area for counters read back from the device
    void* pinned;
    size_t pinned_bytes;
    // pinned staging ring of pb3d_h2d_async: small host inputs (2-D masks, descriptors) go up without a host wait
    void* stage;
    size_t stage_bytes, stage_head;
    u64 sync_count;             // host waits on the context's stream so far (pb3d_sync_count: tests bound the waits of a pipeline)
    // state kept between pb3d_points_count and pb3d_points_fill (host-pointer flavour)
    struct {
        i64 A0, A1, A2, n;
        int C, ncolors, stride;
        u8 colors[3 * 32];
        bool valid;
        bool extracted;     // the points already sit in scratch 1 / 3 (single-pass extraction during the count)
    } pts;
    // state kept between pb3d_deform_count and pb3d_deform_fill
    struct {
        int ox, oy, oz;
        i64 X, Y, Z, n;
        bool valid;
    } deform;
    struct ValidCache { void* buf; u64 gen; i64 W, D; double p[8]; } valid_cache;   // validity bit table of the last 90-degree step (scratch slot 10)
    // Tile programs of the bit-sliced chain (csrc/sliced.hip, scratch slot 32): valid for these steps on this (W, D)
    struct S32Cache { bool valid; u64 gen; i64 W, D; int ns; double p[32 * 8]; } s32_cache;
    hipEvent_t s32_ev;          // recorded behind the slice kernel's "not 0/1" flag copy
    // membership bits of the last labelled volume (csrc/ccl.hip, scratch slot 42): word row * P + t, bit = voxel is a member
    // of colour k (K colours labelled together, csrc/ccl.hip: their numbering is per colour, so a label needs its colour's bits to mean anything
    // once K > 1).  gen = scratch_slot_gen[42] when the bits were written: only a reallocation of THAT slot invalidates them.
    struct CclLast { bool valid; const void* labels; const void* bits; i64 rows, A2, P; u64 gen; bool members_only; int K, C; u32 colors[PB3D_CCL_MAX_COLORS]; } ccl_last;
    // Device block pool behind pb3d_dev_alloc / pb3d_dev_free: a freed block is kept (no hipFree, no stream synchronisation) and handed
    // to the next request of about its size.  Everything that touches such a block runs on ctx->stream, in order, so a re-used block
    // is never written before its previous reader has finished.  The NumPy-signature API allocates and frees a volume-sized buffer
    // or two per call: with hipMalloc / hipFree (which waits for the device) that was most of a notebook-1 run's wall time.
    struct PoolBlock { void* p; size_t bytes; u64 stamp; };
    PoolBlock pool_free[PB3D_POOL_SLOTS];
    int pool_nfree;
    PoolBlock pool_live[PB3D_POOL_LIVE];
    int pool_nlive;
    size_t pool_cached, pool_cap;       // bytes held in pool_free; its limit (PB3D_DEVICE_POOL_MB, default a quarter of the HBM, 0 = off)
    u64 pool_stamp;
    u64 scratch_gen;            // bumped whenever ANY scratch slot is reallocated (cached tables in other slots may have moved)
    u64 scratch_slot_gen[PB3D_NSCRATCH];    // ... and per slot
    // RCCL (loaded lazily with dlopen; see comm.hip)
    void* rccl_lib;
    void* rccl_comm;
    int rank, nranks;
};

// ---- error plumbing -------------------------------------------------------------------------
void pb3d_set_error(const char* fmt, ...);

#define PB3D_HIP(call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            pb3d_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? PB3D_ENOMEM : PB3D_ENODEVICE;                    \
        }                                                                                       \
    } while (0)

#define PB3D_TRY(call)                 \
    do {                               \
        int rc_ = (call);              \
        if (rc_ != PB3D_OK) return rc_; \
    } while (0)

#define PB3D_REQUIRE(cond, ...)         \
    do {                                \
        if (!(cond)) {                  \
            pb3d_set_error(__VA_ARGS__); \
            return PB3D_EINVAL;         \
        }                               \
    } while (0)

#define PB3D_CHECK_LAUNCH() PB3D_HIP(hipGetLastError())

// scratch slot `slot` grown to at least `bytes`
int pb3d_scratch(pb3d_ctx* ctx, int slot, size_t bytes, void** out);
// hipStreamSynchronize(ctx->stream) + bookkeeping (the staging ring is empty afterwards)
int pb3d_stream_sync(pb3d_ctx* ctx);

// grid size for grid-stride streaming kernels: enough blocks to fill 256 CUs, capped
// blocks_per_cu <= 0: no cap -- one workgroup per tile of the stream.  For write-heavy or latency-heavy streams the dispatcher balances
// many small workgroups better than a persistent grid-stride loop does (colour apply 0.84 -> 0.70 ms, float32 projection 1.03 -> 0.90 ms
// at 1024^3); kernels with per-workgroup set-up or flush (component statistics) want the cap.
static inline unsigned pb3d_stream_blocks(const pb3d_ctx* ctx, i64 work_items, int per_block, int blocks_per_cu) {
    i64 need = (work_items + per_block - 1) / per_block;
    i64 cap = (blocks_per_cu > 0 && !ctx->tune_uncap) ? (i64)(ctx->cus > 0 ? ctx->cus : 256) * blocks_per_cu : 0x7fffffffll;
    if (need < 1) need = 1;
    return (unsigned)(need < cap ? need : cap);
}

// ---- exact u32 division by a run-time constant (Granlund-Montgomery round-up form): q = (t + ((n - t) >> sa)) >> sb, t = mulhi(m, n).
// A 64-bit integer division is ~100 vector instructions on gfx950 and the VALU issues one wave instruction per four cycles: kernels
// that turn a linear voxel index into coordinates with / and % are bound by exactly that.
struct pb3d_magic { u32 m; int sa, sb; u32 d; };
static inline pb3d_magic pb3d_make_magic(u32 d) {     // 1 <= d < 2^31
    int L = 0;
    while ((1ull << L) < d) ++L;
    pb3d_magic g;
    g.m = (u32)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
    g.sa = L < 1 ? L : 1;
    g.sb = L > 1 ? L - 1 : 0;
    g.d = d;
    return g;
}
__device__ __forceinline__ u32 pb3d_div(u32 n, const pb3d_magic g) {
    const u32 t = __umulhi(g.m, n);
    return (t + ((n - t) >> g.sa)) >> g.sb;
}

// what a labelling leaves on the device (csrc/ccl.hip): total[k] = components of colour k, records[(k * dcap + c) * 64] = 64-byte statistics
// record of component c + 1 {int lo[3], hi[3] (inclusive), pad[2]; u64 count, sum[3]}, valid for c < min(total[k], dcap)
struct pb3d_ccl_dev { const i64* total; const char* records; int dcap; };
int pb3d_ccl_label_on_device(pb3d_ctx* ctx, const uint8_t* d_grid, int64_t A0, int64_t A1, int64_t A2, const uint8_t color[3], int channels, int32_t* d_labels,
                             int64_t cap, pb3d_ccl_dev* dev);

// ---- kernels' host launchers used across translation units ---------------------------------
// process_voxel_grid through the bit-sliced chain (csrc/sliced.hip); *took = 0: not applicable, nothing written
int pb3d_process_grid_sliced(pb3d_ctx* ctx, const u8* d_occ, i64 W, i64 H, i64 D, const u8* d_mask_wh, int angle_interval, u8* d_out,
                             int known_binary, int* took);
int pb3d_global_carve_sliced(pb3d_ctx* ctx, const u8* d_mask_wh, const u8* d_rgb_hw3, i64 W, i64 H, i64 D, int angle_interval, u8* d_out_rgb,
                             int* took);
// pb3d_process_grid_dev for callers whose grid is 0/1 by construction (occupancy of a colour grid, all-ones): no host wait
int pb3d_process_grid_binary_dev(pb3d_ctx* ctx, const u8* d_occ, i64 W, i64 H, i64 D, const u8* d_mask_wh, int angle_interval, u8* d_out,
                                 u8* d_tmp);
// one rotation step with the caller's matrix through the bit-sliced path (csrc/sliced.hip); *took = 0: not applicable
int pb3d_rotate_step_sliced(pb3d_ctx* ctx, const u8* d_occ, i64 W, i64 H, i64 D, const double M[9], const double off[3], const u8* d_mask_wh, u8* d_out,
                            int* took);
// the arithmetic kernel (csrc/rotate.hip): any uint8 data, any rotation about Y
int pb3d_launch_rotate_generic(pb3d_ctx* ctx, const u8* d_in, i64 W, i64 H, i64 D, const double M[9],
                               const double off[3], const u8* d_mask_wh, u8* d_out, const u8* d_mask_src);
bool pb3d_is_perm_step(const double M[9], const double off[3], i64 W, i64 D);
int pb3d_perm_valid_table(pb3d_ctx* ctx, const double M[9], const double off[3], i64 W, i64 D, u32** bits, int* nw, int* c0, int* c2, bool* rot90);
bool pb3d_perm_step_ok(const double M[9], const double off[3], i64 W, i64 D, const void* a, const void* b);
int pb3d_launch_rotate_perm(pb3d_ctx* ctx, const u8* d_in, i64 W, i64 H, i64 D, const double M[9], const double off[3],
                            const u8* d_mask_src, const u8* d_mask_dst, u8* d_out);
int pb3d_try_part_carve90(pb3d_ctx* ctx, const u8* d_colored, int C, i64 W, i64 H, i64 D, const u8* d_mask_sub, const u8* d_mask_carve,
                          const int* job_angle, const int* job_skip, int njobs, u8* d_out);
int pb3d_transpose_mask_dev(pb3d_ctx* ctx, const u8* d_hw, i64 h, i64 w, u8* d_wh);
int pb3d_part_carve90_planes(pb3d_ctx* ctx, const u8* d_colored, int C, i64 W, i64 H, i64 D, const u32* d_A, const u32* d_AT, int njobs, const u32* d_vbits,
                             int nwv, int c0, int c2, u8* d_out, int* took);
int pb3d_launch_gc90_stream(pb3d_ctx* ctx, const u8* d_bin_hw, const u8* d_rgb_hw3, int C, const u32* d_vbits, int nw, int c0, i64 W, i64 H, i64 D, i64 x0,
                            i64 x1, u8* d_out_slab);
int pb3d_launch_global_carve90(pb3d_ctx* ctx, const u8* d_bin_hw, const u8* d_rgb_hw3, int C, i64 h, i64 w, const double M[9],
                               const double off[3], i64 x0, i64 x1, u8* d_out_slab);
